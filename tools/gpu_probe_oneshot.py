"""One-shot rdis_hip_cgd_batch calls (= one SubspaceOptimizer::optimize each: id lists in, results out, nothing
kept between calls) on the decompositions of ladybug DESIGN.md section 4 quotes.  RDIS_HIP_TIMING=1 prints the
phases of every call on stderr."""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)


def run(name, fp, fv, cp, ci, reps=6):
    x0 = np.ascontiguousarray(pp.x0[fv])
    best = 1e9
    for rep in range(reps):
        t = time.perf_counter()
        g.cgd_batch(fp, fv, cp, ci, x0, 25, 3e-8)
        best = min(best, (time.perf_counter() - t) * 1e3)
    print("one-shot call, %s (%d components, %d variables, %d factors): best of %d %.3f ms" % (name, len(fp) - 1, len(fv), len(ci), reps, best), file=sys.stderr)


run("ladybug as one component", pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
cams, pts = P.ba_alternation_plans(pp)
run("49 camera components", *cams)
run("7776 point components", *pts)
g.set_x(pp.x0)
# the level RDIS reaches with a 46-camera separator assigned: what is left falls apart on the device
assigned = np.zeros(pp.nvars, np.uint8)
deg = np.bincount(pp.cam_vid0 // 9, minlength=49)
for c in np.argsort(-deg, kind="stable")[:46]:
    assigned[9 * c:9 * c + 9] = 1
assigned[441:444] = 1
lists = g.components(assigned)
run("the level below a 46-camera separator", *lists)
