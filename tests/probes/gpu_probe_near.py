import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
if os.environ.get("RDIS_PROBE_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["RDIS_PROBE_LIB"])
ctx = capi.Context(0)
pp = P.load_bal(ncams=49, npts=500)
g = capi.Problem(ctx, pp)
free_ptr, free_vid, fac_ptr, fac_id = P.ba_alternation_plans(pp)[0]
plan = capi.Plan(g, free_ptr, free_vid, fac_ptr, fac_id)
plan.set_option("trace_records", 4096); plan.set_option("dump_iters", 25)
plan.set_start(None)
x_before = g.get_x()
plan.solve(25, 3e-8); r = plan.fetch()
worst = []
for c in range(len(free_ptr) - 1):
    fv, fc = free_vid[free_ptr[c]:free_ptr[c + 1]], fac_id[fac_ptr[c]:fac_ptr[c + 1]]
    q = P.load_bal(ncams=49, npts=500); q.x0 = x_before
    tr, n = plan.get_trace(c, 4096)
    rep = O.OracleProblem(q).replay(tr, free_vid=fv, fac=fc, x=x_before[fv], maxiters=25, vdump=plan.get_vectors(c, 25)[:int(r.iters[c]) + 1])
    worst.append((rep.max_f_rel_near, rep.max_slope_rel_near, c, rep.step_mismatches, rep.max_f_bound, rep.max_slope_bound))
worst.sort(reverse=True)
print(os.environ.get("RDIS_PROBE_LIB"), "worst near f_rel:", ["%.2e (c%d)" % (w[0], w[2]) for w in worst[:5]], "slope:", "%.2e" % max(w[1] for w in worst), "mismatches", sum(w[3] for w in worst), "| bounds: f %.3f slope %.3f" % (max(w[4] for w in worst), max(w[5] for w in worst)))
