"""ladybug 5 / 30 and 12 / 300 as alternation plans (cameras against fixed points, points against fixed cameras): which solver the
dispatcher picks for these small components with constants, and whether the oracle's LDS topology reproduces it"""
import sys; sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
kinds = ("components_cooperative", "components_grid_stream", "components_tiny", "components_lds", "components_point_major", "components_plain")
for nc, npt in ((5, 30), (12, 300), (49, 1000)):
    pp = P.load_bal(ncams=nc, npts=npt)
    for name, dec in zip(("cameras", "points"), P.ba_alternation_plans(pp)):
        fp, fv, cp, ci = dec
        g = capi.Problem(ctx, pp); plan = capi.Plan(g, *dec)
        plan.set_start(pp.x0[fv]); plan.solve(25, 3e-8); r = plan.fetch()
        ncomp = len(fp) - 1
        mf = int(np.diff(cp).max())
        same = 0; tried = 0
        for c in range(0, ncomp, max(1, ncomp // 8)):
            v, f = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
            w = O.OracleProblem.device_lds_default(pp, free_vid=v, fac=f, threads=(64 if mf <= 64 else 128 if mf <= 128 else 256 if mf <= 256 else 512 if mf <= 512 else 768)).cgd(free_vid=v, fac=f, x=pp.x0[v], maxiters=25)
            tried += 1; same += int(float(r.fret[c]) == w.fret and int(r.nfeval[c]) == w.nfeval)
        print(nc, npt, name, ncomp, "max factors", mf, {k[11:]: plan.info(k) for k in kinds if plan.info(k)}, "== oracle (lds topology):", same, "of", tried)
