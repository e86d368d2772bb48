"""the strong-scaling block's per-rank launch at 1, 2, 4, 8 ranks, emulated on ONE GPU (a rank's shard is what that GPU
would run): kernel time of rank 0's launch per world size, for both component sizes of bench.STRONG_SIZES -- what
sharding alone can give (the 8-byte all-reduce of the objective excluded)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
sizes = [a for a in sys.argv[1:] if a in bench.STRONG_SIZES] or ["L", "small"]
for size in sizes:
    bench.STRONG.clear(); bench.STRONG.update(bench.STRONG_SIZES[size])
    t1 = None
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
        t1 = t1 or best
        print("size %-5s world %d: rank 0 has %4d components (%d factors each), kernel %8.3f ms (%.2f x world 1), %d launch(es), "
              "%d workgroup(s) per component, objective %.8g, evaluations per component %.0f (max %d)" % (
                  size, world, len(mine), pp.nfac // pp.ncomp, best, t1 / best, plan.last_kernel_ms()[1], plan.info("point_major_group"),
                  r.fret.sum(), r.nfeval.mean(), r.nfeval.max()), flush=True)
        plan.close(); g.close()
        del pp
