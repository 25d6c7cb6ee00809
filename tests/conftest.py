import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running check, excluded unless PEMAP_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("PEMAP_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set PEMAP_SLOW=1 to run")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)
