"""exploratory: the call shapes RDIS makes on full ladybug: (i) 46 cameras + 1 point
free, every factor of those cameras (the reference spends 113 s per such call), (ii) 3 cameras + their
points, (iii) the single-point components, in one launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rdis_amd import problems as P, capi
from oracle import oracle as O
ctx = capi.Context(0)
pp = P.load_bal()
g = capi.Problem(ctx, pp)
nc = 49
cam_of = pp.cam_vid0 // 9
# (i)
free = np.concatenate([np.arange(9 * 46), np.arange(441, 444)]).astype(np.int64)
fac = np.where((cam_of < 46) | (pp.pt_vid0 == 441))[0].astype(np.int64)
plan = capi.Plan(g, np.array([0, len(free)]), free, np.array([0, len(fac)]), fac)
plan.set_option("trace_records", 8192); plan.set_option("dump_iters", 25)
for rep in range(3):
    g.set_x(pp.x0); plan.set_start(None)
    t = time.perf_counter(); plan.solve(25, 3e-8); r = plan.fetch(); dt = time.perf_counter() - t
ms, nl = plan.last_kernel_ms()
print("(i) %d free vars, %d factors: kernel %.3f ms wall %.3f ms, f %.6g -> %.6g, iters %d nfeval %d status %d" % (
    len(free), len(fac), ms, dt * 1e3, r.fret[0] - r.delta[0], r.fret[0], r.iters[0] + 1, r.nfeval[0], r.status[0]))
tr, n = plan.get_trace(0, 8192)
q = P.load_bal()
t = time.perf_counter()
rep = O.OracleProblem(q).replay(tr, free_vid=free, fac=fac, x=pp.x0[free], maxiters=25, vdump=plan.get_vectors(0, 25)[:int(r.iters[0]) + 1])
print("    replay: mismatches %d/%d consumed %d of %d, f_rel ordinary %.2e slope %.2e (oracle replay took %.1f s)" % (
    rep.step_mismatches, rep.tag_mismatches, rep.consumed, n, rep.max_f_rel_near, rep.max_slope_rel_near, time.perf_counter() - t))
t = time.perf_counter(); ro = O.OracleProblem(q).cgd(free_vid=free, fac=fac, x=pp.x0[free], maxiters=25); dto = time.perf_counter() - t
print("    CPU oracle (dense gradient): %.2f s, fret %.6g" % (dto, ro.fret))
# (ii)+(iii): fix the 46 cameras, label the rest
a = np.zeros(pp.nvars, np.uint8); a[:9 * 46] = 1
g.set_x(pp.x0)
t = time.perf_counter(); comps = g.components(a); dtc = time.perf_counter() - t
sizes = np.diff(comps[0])
for overlap in (0, 1):
    plan2 = capi.Plan(g, *comps)
    plan2.set_option("overlap_batch", overlap)
    for rep in range(3):
        g.set_x(pp.x0); plan2.set_start(None)
        t = time.perf_counter(); plan2.solve(25, 3e-8); r2 = plan2.fetch(); dt = time.perf_counter() - t
    ms2, nl2 = plan2.last_kernel_ms()
    print("(ii+iii) 46 cameras fixed: %d components (labelling %.2f ms), largest %d vars / %d factors; overlap %d: solve kernel %.3f ms in %d launches, wall %.3f ms, sum f %.6g -> %.9g" % (
        len(sizes), dtc * 1e3, sizes[-1], np.diff(comps[2])[-1], overlap, ms2, nl2, dt * 1e3, (r2.fret - r2.delta).sum(), r2.fret.sum()))
    plan2.close()
