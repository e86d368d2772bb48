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
plan.close(); g.close()
# the evaluation entry points in their streaming regime: 256 x (49 cameras, 7776 points) = 8.0e6 factors, 6.1e6 variables
# (OptimizableFunction::evalFactors / computeGradient; algorithmic bytes 24 F + 8 N and 24 F + 16 N)
import time
huge = P.make_synthetic_ba(256, 49, 7776, obs_per_pt=4)
g = capi.Problem(ctx, huge)
g.set_x(huge.x0); g.eval(); g.eval_grad()
for name, fn, nbytes in (("eval_sum_kernel (value) 8.0e6 factors", g.eval, 24 * huge.nfac + 8 * huge.nvars),
                         ("eval_sum_kernel<grad> + gather_grad_kernel 8.0e6 factors", g.eval_grad, 24 * huge.nfac + 16 * huge.nvars)):
    t0 = time.perf_counter()
    for _ in range(REPS): fn()
    dt = (time.perf_counter() - t0) / REPS
    print("%s: %.3f ms per call (host clock, result fetched), algorithmic %.0f MB -> %.0f GB/s" % (name, dt * 1e3, nbytes / 1e6, nbytes / dt / 1e9))
for k, (ms, nf) in out.items():
    print("%s: %.3f ms per launch (HIP events), %d f-evaluations" % (k, ms, nf))
