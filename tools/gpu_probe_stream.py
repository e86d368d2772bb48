"""exploratory: streaming grid solver -- ladybug (forced) and one large synthetic component"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import problems as P, capi
ctx = capi.Context(0)
def run(pp, opts, mit, label):
    g = capi.Problem(ctx, pp)
    plan = capi.Plan(g)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_start(pp.x0)
    for rep in range(2):
        t = time.time(); plan.solve(mit, 3e-8); r = plan.fetch(want_x=False); dt = time.time() - t
    ms, nl = plan.last_kernel_ms()
    F, N = pp.nfac, pp.nvars
    nf, ng = int(r.nfeval[0]), int(r.ngeval[0])
    abytes = max(nf - ng, 0) * (24 * F + 8 * N + 8) + ng * (24 * F + 16 * N + 8)
    print("%s: kernel %.3f ms, iters %d, nf/ng %d/%d, %.2f us/eval, algorithmic %.1f GB/s (%.1f%% of 8 TB/s), fret %.6g status %d" % (
        label, ms, r.iters[0] + 1, nf, ng, ms * 1e3 / nf, abytes / (ms * 1e-3) / 1e9, abytes / (ms * 1e-3) / 8e12 * 100, r.fret[0], r.status[0]))
    plan.close(); g.close()
pp = P.load_bal().single_component()
run(pp, {}, 25, "ladybug full, register-resident")
run(pp, {"force_stream": 1}, 25, "ladybug full, streaming")
for (C, Pn) in [(64, 250000), (64, 2000000)]:
    t = time.time(); big = P.make_synthetic_ba(1, C, Pn, obs_per_pt=4); print("generated %d factors, %d vars in %.1f s" % (big.nfac, big.nvars, time.time() - t))
    run(big, {}, 3, "synthetic %dx%d (%d factors), streaming" % (C, Pn, big.nfac))
