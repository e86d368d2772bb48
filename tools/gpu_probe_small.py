"""exploratory: many small components (synthetic-S, ladybug point components), kernel time"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
def run(name, pp, plans=None):
    g = capi.Problem(ctx, pp)
    plan = capi.Plan(g) if plans is None else capi.Plan(g, *plans)
    best = 1e9
    for rep in range(5):
        g.set_x(pp.x0)
        plan.set_start(None)
        plan.solve(25, 3e-8); r = plan.fetch()
        ms, nl = plan.last_kernel_ms(); best = min(best, ms)
    print("%s %-20s kernel %.3f ms  iters %d  fret sum %.9g" % (os.environ.get("RDIS_PROBE_LIB", "cur"), name, best, int((r.iters + 1).sum()), r.fret.sum()))
run("synthetic-S", P.make_synthetic_ba(1000, 3, 40))
pp = P.load_bal()
cams, pts = P.ba_alternation_plans(pp)
run("ladybug points", pp, pts)
run("ladybug cameras", pp, cams)
run("synthetic 4000x(2,12)", P.make_synthetic_ba(4000, 2, 12))
