"""one large bundle-adjustment component on the wide point-major group (solver_ptm.hpp, cgd_ptmg_kernel<512, ., true>) and, with
ptm_stream = 0, on the grid solver it replaces (solver_stream.hpp): kernel time per evaluation, replay of a short solve.

    python tools/gpu_probe_wide.py [cameras] [points] [obs_per_pt] [iters]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
C_, Pn, K_, IT = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 64), (2, 2000000), (3, 4), (4, 3)))
ctx = capi.Context(0)
t = time.time(); big = P.make_synthetic_ba(1, C_, Pn, obs_per_pt=K_); print("generated in %.1f s" % (time.time() - t), flush=True)
g = capi.Problem(ctx, big)
res = {}
for name, opts in (("wide point-major group", {}), ("grid solver (ptm_stream = 0)", {"ptm_stream": 0})):
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(big.x0)
    t = time.time(); plan.solve(IT, 3e-8); r = plan.fetch(want_x=True); first = time.time() - t
    plan.set_start(big.x0)
    t = time.time(); plan.solve(IT, 3e-8); r = plan.fetch(want_x=True); second = time.time() - t
    ms = plan.last_kernel_ms()[0]
    nf, ng = int(r.nfeval[0]), int(r.ngeval[0])
    F, N = big.nfac, big.nvars
    abytes = (nf - ng) * (24 * F + 8 * N + 8) + ng * (24 * F + 16 * N + 8)
    print("%s: %d factors, %d variables: first solve %.2f s (tables), second %.3f s; kernel %.3f ms, %d evaluations (%d gradients): %.1f us per evaluation, "
          "%.2f TB/s algorithmic (%.3f of 8); fret %.9g, status %d; K = %d, wide %d, point-major %d, grid %d" % (
              name, F, N, first, second, ms, nf, ng, ms * 1e3 / nf, abytes / (ms * 1e-3) / 1e12, abytes / (ms * 1e-3) / 8e12, r.fret[0], int(r.status[0]),
              plan.info("point_major_group"), plan.info("point_major_wide"), plan.info("components_point_major"), plan.info("components_grid_stream")), flush=True)
    res[name] = r
    plan.close()
a, b = res.values()
print("end values: %.12g against %.12g (relative difference %.2e)" % (a.fret[0], b.fret[0], abs(a.fret[0] - b.fret[0]) / abs(b.fret[0])))
