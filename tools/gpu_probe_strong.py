"""exploratory: the strong-scaling block's per-rank launch at 1, 2, 4, 8 ranks, emulated on ONE GPU (a rank's
shard is what that GPU would run), and a single 2048-factor component on the streaming grid solver with 1..4
workgroups: would a component spread over several workgroups shorten a rank's launch at high rank counts?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
for world in (1, 2, 4, 8):
    pp, csr, mine, loads = bench.strong_scaling_shard(0, world)
    g = capi.Problem(ctx, pp)
    plan = capi.Plan(g, *csr)
    for a in sys.argv[1:]:
        if a.startswith("--opt="):
            k, v = a[6:].split(":"); plan.set_option(k, int(v))
    plan.set_start(pp.x0[csr[1]])
    best = 1e9
    for rep in range(3):
        plan.solve(25, 3e-8); r = plan.fetch(); best = min(best, plan.last_kernel_ms()[0])
    print("world %d: rank 0 has %d components, kernel %.3f ms, %d launches, objective %.6g, evals/component %.0f" % (
        world, len(mine), best, plan.last_kernel_ms()[1], r.fret.sum(), r.nfeval.mean()))
    plan.close(); g.close()
one = P.make_synthetic_ba(1, 8, 512, obs_per_pt=4)
g = capi.Problem(ctx, one)
for nwg in (1, 2, 3, 4, 8):
    plan = capi.Plan(g)
    plan.set_option("force_stream", 1); plan.set_option("coop_min_factors", 1024); plan.set_option("coop_workgroups", nwg)
    plan.set_start(one.x0)
    best = 1e9
    for rep in range(3):
        plan.solve(25, 3e-8); r = plan.fetch(); best = min(best, plan.last_kernel_ms()[0])
    print("one component (%d factors) on the streaming solver with %d workgroup(s) of 512: %.3f ms, %d evaluations, fret %.6g" % (
        one.nfac, nwg, best, r.nfeval[0], r.fret[0]))
    plan.close()
plan = capi.Plan(g); plan.set_start(one.x0)
for rep in range(3):
    plan.solve(25, 3e-8); r = plan.fetch()
print("the same component, one workgroup of the batched solver: %.3f ms, %d evaluations" % (plan.last_kernel_ms()[0], r.nfeval[0]))
