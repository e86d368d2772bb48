import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
from rdis_amd import capi, problems as P
lib = C.CDLL("tests/cpp/libfactors_host.so")
pp = P.load_bal().single_component()
ctx = capi.Context(0); g = capi.Problem(ctx, pp)
rng = np.random.default_rng(1)
for k in range(3):
    x = pp.x0 if k == 0 else pp.x0 * (1 + 1e-3 * rng.standard_normal(pp.nvars))
    g.set_x(x)
    fd, gd = g.eval_each(), g.grad_each_ba()
    x12 = np.concatenate([x[pp.cam_vid0[:, None] + np.arange(9)], x[pp.pt_vid0[:, None] + np.arange(3)]], axis=1).copy()
    f = np.empty(pp.nfac); g12 = np.empty((pp.nfac, 12))
    v = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.fh_eval_grad_each(C.c_longlong(pp.nfac), v(x12), v(np.ascontiguousarray(pp.obs)), v(f), v(g12))
    print(k, "values differ:", int(np.sum(f != fd)), "partials differ:", int(np.sum(g12 != gd)), "max rel", float(np.max(np.abs(f - fd) / np.abs(fd))))
