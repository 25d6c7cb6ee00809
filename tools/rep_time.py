import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import fixtures
from pecaller_amd import PemapDev
ix = fixtures.index()
dev = PemapDev(0)
dev.set_lookup_replicas(0)
t=time.time(); dev.build_index(ix["genome"], ix["contig_len"]); print("build without replicas %.2f s" % (time.time()-t))
t=time.time(); dev.set_lookup_replicas(8); print("replicas (alloc + build) %.2f s" % (time.time()-t))
t=time.time(); dev.build_index(ix["genome"], ix["contig_len"]); print("second build, replicas kept allocated %.2f s" % (time.time()-t))
dev.close()
