"""profile target for counter passes: ONE workload of the batch solvers, solved three times.
    python tools/pmc_target.py strong1|strong8|synthL|synthS|ladybug [--opt=name:value ...]
(rocprofv3 --pmc ... -- python tools/pmc_target.py strong8; tools/pmc_summary.py turns the CSV into per-kernel figures)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
bench.STRONG.update(bench.STRONG_SIZES["small"])   # (round 2's block: 2048-factor components, the LDS-resident solver's case)
from rdis_amd import capi, problems as P
what = sys.argv[1]
ctx = capi.Context(0)
if what.startswith("strong"):
    pp, csr, mine, loads = bench.strong_scaling_shard(0, int(what[6:]))
elif what == "synthL":
    pp = P.make_synthetic_ba(int(os.environ.get("RDIS_SYNTHL_COMPONENTS", "256")), 49, 7776, obs_per_pt=4)
    csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
elif what == "synthS":
    pp = P.make_synthetic_ba(1000, 3, 40)
    csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
else:
    pp = P.load_bal().single_component()
    csr = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
g = capi.Problem(ctx, pp)
plan = capi.Plan(g, *csr)
for a in sys.argv[2:]:
    if a.startswith("--opt="):
        k, v = a[6:].split(":"); plan.set_option(k, int(v))
plan.set_start(pp.x0[csr[1]])
for _ in range(3):
    plan.solve(25, 3e-8); r = plan.fetch()
ms, nl = plan.last_kernel_ms()
F = np.diff(csr[2]); N = np.diff(csr[0])
ab = float(np.sum((r.nfeval - r.ngeval).clip(0) * (24 * F + 8 * N + 8) + r.ngeval * (24 * F + 16 * N + 8)))
fe = float(np.sum(r.nfeval * F))
print("%s: kernel %.3f ms, %d launch(es), %d components, %.4g factor evaluations (%.3g /s), algorithmic bytes %.4g (%.1f GB/s), objective %.9g" % (
    what, ms, nl, len(F), fe, fe / (ms * 1e-3), ab, ab / (ms * 1e-3) / 1e9, r.fret.sum()))
