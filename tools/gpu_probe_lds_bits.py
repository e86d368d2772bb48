"""exploratory: where do the LDS-resident batch solver (lds_camera_sums = 0) and the plain one part?  Both run
with a trace; the first record that differs is printed per workgroup size / rotation mode, and every run is
repeated to see whether it is reproducible in itself."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rdis_amd import capi, problems as P
ctx = capi.Context(0)
syn = P.make_synthetic_ba(40, 5, 64, obs_per_pt=3)
csr = (syn.comp_free_ptr, syn.comp_free_vid, syn.comp_fac_ptr, syn.comp_fac_id)
g = capi.Problem(ctx, syn)
TAGS = {1: "F", 2: "FD", 3: "ITER", 4: "START", 5: "LINMIN"}


def run(opts, iters=25, trace=4096):
    g.set_x(syn.x0)
    plan = capi.Plan(g, *csr)
    for k, v in opts.items():
        plan.set_option(k, v)
    plan.set_option("trace_records", trace)
    plan.set_start(None)
    plan.solve(iters, 3e-8)
    r = plan.fetch()
    tr = [plan.get_trace(c, trace) for c in range(len(csr[0]) - 1)]
    plan.close()
    return r, tr


for threads in (128, 256, 768):
    for rot in (0, 2):
        a = {"lds_resident": 0, "block_threads": threads, "camera_records": rot}
        b = {"lds_resident": 1, "block_threads": threads, "camera_records": rot, "lds_rot": 1 if rot else 0, "lds_camera_sums": 0}
        ra, ta = run(a); ra2, ta2 = run(a)
        rb, tb = run(b); rb2, tb2 = run(b)
        print("threads %d rot %d: plain reproducible %s, lds reproducible %s, equal components %d / %d" % (
            threads, rot, np.array_equal(ra.fret, ra2.fret), np.array_equal(rb.fret, rb2.fret),
            int(np.sum(ra.fret == rb.fret)), len(ra.fret)), flush=True)
        shown = 0
        for c in range(len(ra.fret)):
            if ra.fret[c] == rb.fret[c] or shown >= 2: continue
            (xa, na), (xb, nb) = ta[c], tb[c]
            n = min(na, nb)
            d = np.nonzero(np.any(xa[:n] != xb[:n], axis=1))[0]
            if len(d) == 0:
                print("  comp %d: traces equal over %d records (lengths %d / %d)" % (c, n, na, nb)); shown += 1; continue
            k = d[0]
            print("  comp %d: first difference at record %d of %d: plain %s %r | lds %s %r" % (
                c, k, n, TAGS.get(int(xa[k, 0]), "?"), xa[k, 1:].tolist(), TAGS.get(int(xb[k, 0]), "?"), xb[k, 1:].tolist()))
            if k > 0: print("     previous record: %s %r" % (TAGS.get(int(xa[k - 1, 0]), "?"), xa[k - 1, 1:].tolist()))
            shown += 1
