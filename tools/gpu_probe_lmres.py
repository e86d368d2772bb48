import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
ctx = capi.Context(0)
for nc, npt in ((5, 30), (49, 300), (49, 2000), (0, 0)):
    pp = P.load_bal(ncams=nc, npts=npt) if nc else P.load_bal()
    g = capi.Problem(ctx, pp)
    r1 = g.lm_optimize(maxiters=1)
    o = O.OracleProblem(pp, emulate_stale_cache=False)
    grad = o.gradient()
    dp = r1.x - pp.x0
    mu = r1.history[-1, 0]
    E, G = o.eval_each(np.arange(pp.nfac)), o.grad_each_ba(np.arange(pp.nfac))
    Jd = np.zeros(pp.nvars)
    vids = np.concatenate([pp.cam_vid0[:, None] + np.arange(9), pp.pt_vid0[:, None] + np.arange(3)], axis=1)
    rows = G / np.sqrt(2.0 * E)[:, None]
    np.add.at(Jd, vids.reshape(-1), (rows * np.sum(rows * dp[vids], axis=1)[:, None]).reshape(-1))
    res = Jd + mu * dp + grad
    nc9 = 9 * int(pp.meta["ncams"])
    print(nc, npt, "hist", r1.history[:, [0, 3]].tolist(), "res/grad %.3e cams %.3e pts %.3e" % (np.linalg.norm(res) / np.linalg.norm(grad), np.linalg.norm(res[:nc9]) / np.linalg.norm(grad), np.linalg.norm(res[nc9:]) / np.linalg.norm(grad)),
          "clamped", int(np.sum((r1.x <= pp.lo) | (r1.x >= pp.hi))))
