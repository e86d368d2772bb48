import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)
x0 = np.ascontiguousarray(pp.x0[pp.comp_free_vid])
for rep in range(4):
    t = time.perf_counter()
    r = g.cgd_batch(pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id, x0, 25, 3e-8)
    print("one-shot call %.3f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
