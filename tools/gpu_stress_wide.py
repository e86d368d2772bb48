"""stress of the wide point-major groups (plain and with local camera numbering): many solves of several shapes in turn, every
repetition must give the first one's bits; run under `timeout` (a hung exchange would otherwise hang the box)

    timeout 600 python tools/gpu_stress_wide.py [rounds]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = capi.Context(0)
shapes = [(24, 30000, 4, {}), (64, 200000, 4, {}), (300, 60000, 4, {}), (24, 30000, 4, {"ptm_local_cameras": 1}), (1778, 400000, 5, {}), (12, 40000, 3, {"ptm_group": 200})]
plans = []
for (C_, Pn, K_, opts) in shapes:
    pp = P.make_synthetic_ba(1, C_, Pn, obs_per_pt=K_).single_component()
    g = capi.Problem(ctx, pp)
    plan = capi.Plan(g)
    for k, v in opts.items(): plan.set_option(k, v)
    plans.append((pp, g, plan, None))
t0 = time.time()
for rnd in range(rounds):
    for i, (pp, g, plan, first) in enumerate(plans):
        plan.set_start(pp.x0)
        plan.solve(5, 3e-8)
        r = plan.fetch()
        sig = (float(r.fret[0]), r.x.tobytes(), int(r.nfeval[0]), int(r.status[0]))
        assert (r.status[0] & 0xFF) != 7, "exchange timed out"
        if first is None:
            plans[i] = (pp, g, plan, sig)
            print("shape", shapes[i][:3], shapes[i][3], "K", plan.info("point_major_group"), "wide", plan.info("point_major_wide"), "local", plan.info("point_major_local_cameras"),
                  "fret %.9g" % r.fret[0], flush=True)
        else:
            assert sig == first, ("bits changed", i, rnd)
print("%d rounds x %d shapes: the same bits every time, %.1f s" % (rounds, len(shapes), time.time() - t0))
