"""Soak of the one-shot path: rdis_hip_cgd_batch over four decompositions of ladybug in turn, many times -- the results of
every shape must be the same bits every time (the plans' host arrays come from a cache of reused blocks) and the
process must not grow."""
import sys, os, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
pp = P.load_bal().single_component()
g = capi.Problem(ctx, pp)
cams, pts = P.ba_alternation_plans(pp)
whole = (pp.comp_free_ptr, pp.comp_free_vid, pp.comp_fac_ptr, pp.comp_fac_id)
sub = tuple(np.ascontiguousarray(a) for a in (cams[0][:11], cams[1][:90], cams[2][:11], cams[3][:cams[2][10]]))
shapes = [("whole", whole), ("cameras", cams), ("points", pts), ("ten cameras", sub)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ref, bad = {}, 0
rss0 = None
for rep in range(reps):
    for name, (fp, fv, cp, ci) in shapes:
        g.set_x(pp.x0)
        r = g.cgd_batch(fp, fv, cp, ci, np.ascontiguousarray(pp.x0[fv]), 5, 3e-8)
        key = (r.fret.tobytes(), r.x.tobytes(), r.iters.tobytes(), r.status.tobytes())
        if name not in ref: ref[name] = key
        elif ref[name] != key: bad += 1; print("MISMATCH", name, rep)
    if rep == 20: rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("%d rounds of %d shapes: %d mismatches; peak RSS after 20 rounds %.1f MB, at the end %.1f MB" % (reps, len(shapes), bad, rss0 / 1024, rss1 / 1024))
sys.exit(1 if bad or rss1 > rss0 * 1.05 + 8192 else 0)
