"""Exercises the solver / evaluation kernels the headline bench does not reach, a few launches each,
so that one `rocprofv3 --kernel-trace --stats` run yields a statistics row per kernel:
cgd_stream_kernel (ladybug forced onto the streaming grid solver), cgd_group_kernel<16> (ladybug's
7776 point components), cgd_group_kernel<4> (31104 synthetic point components), cc_* (component
labelling), eval_sum_kernel / gather_grad_kernel / eval_each_kernel / partials_kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi

ctx = capi.Context(0)
REPS = 5
out = {}
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)
plan = capi.Plan(g)
plan.set_option("force_stream", 1)
plan.set_start(pp.x0)
for _ in range(REPS):
    plan.solve(25, 3e-8); r = plan.fetch()
out["cgd_stream_kernel ladybug full"] = (plan.last_kernel_ms()[0], int(r.nfeval[0]))
plan.close()
for _ in range(REPS):
    g.set_x(pp.x0); g.eval(); g.eval_grad(); g.eval_each(); g.grad_each_ba()
pts = P.ba_alternation_plans(pp)[1]
plan = capi.Plan(g, *pts)
for _ in range(REPS):
    g.set_x(pp.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
out["cgd_group_kernel<16> ladybug 7776 points"] = (plan.last_kernel_ms()[0], int(r.nfeval.sum()))
plan.close()
a = np.zeros(pp.nvars, np.uint8); a[:441] = 1
for _ in range(REPS):
    comps = g.components(a)
g.close()
big = P.make_synthetic_ba(4, 49, 7776, obs_per_pt=4)
g = capi.Problem(ctx, big)
a = np.zeros(big.nvars, np.uint8); a[np.arange(big.nvars) % 23769 < 441] = 1
for _ in range(REPS):
    comps = g.components(a)
plan = capi.Plan(g, *comps)
for _ in range(REPS):
    g.set_x(big.x0); plan.set_start(None); plan.solve(25, 3e-8); r = plan.fetch()
out["cgd_group_kernel<4> %d synthetic points" % (len(comps[0]) - 1)] = (plan.last_kernel_ms()[0], int(r.nfeval.sum()))
for k, (ms, nf) in out.items():
    print("%s: %.3f ms per launch (HIP events), %d f-evaluations" % (k, ms, nf))
