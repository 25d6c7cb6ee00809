"""Start-up cost of the look-up replicas: index build without them, their allocation + build, a second index on the same object."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from pecaller_amd import PemapDev
ix = fixtures.index()
dev = PemapDev(0)
dev.set_lookup_replicas(0)
t=time.time(); dev.build_index(ix["genome"], ix["contig_len"]); print("build without replicas %.2f s" % (time.time()-t))
t=time.time(); dev.set_lookup_replicas(8); print("replicas (alloc + build) %.2f s" % (time.time()-t))
t=time.time(); dev.build_index(ix["genome"], ix["contig_len"]); print("second build, replicas kept allocated %.2f s" % (time.time()-t))
dev.close()
