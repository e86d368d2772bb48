"""exploratory: long CGD runs on full ladybug (device), end values by iteration budget"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp); plan = capi.Plan(g)
for mi in (25, 100, 400, 1000, 3000):
    plan.set_start(pp.x0)
    t = time.time(); plan.solve(mi, 3e-8); r = plan.fetch(); dt = time.time() - t
    print("device maxiters %4d: fret %.6f iters %d nfeval %d status %d (%.1f ms)" % (mi, r.fret[0], r.iters[0] + 1, r.nfeval[0], r.status[0], dt * 1e3))
