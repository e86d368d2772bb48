"""where the subspace-optimizer calls of rdis_optba_run go (include/rdis_optba.h: rdis_optba_run_hist), per depth of the tree and
kind of step, on ladybug 5 / 30 and full ladybug, next to the reference's own 731 / 13 850 calls

    python tools/gpu_probe_optba_hist.py
"""
import ctypes as C, gzip, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdis_amd import problems as P
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "rdis_amd", "lib", "librdis_host.so"))
lib.rdis_optba_run_hist.restype = C.c_int
lib.rdis_optba_run_hist.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                    C.c_int32, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double), C.c_int32]
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "ladybug.txt")
    with gzip.open(P.LADYBUG_PATH, "rb") as src, open(path, "wb") as dst:
        shutil.copyfileobj(src, dst)
    for label, nc, npnt in (("ladybug 5 / 30", 5, 30), ("ladybug full", 0, 0)):
        for extra in ({}, {"maxCalls": 731.0 if nc else 13850.0}, {"minRR": 0.0, "nRRperLvl": 0.0}, {"steptol": 1e-2}):
            names = [b"SSmaxit", b"SSftol", b"maxCalls"]
            vals = [25.0, 3e-8, 200000.0]
            for k, v in extra.items():
                if k.encode() in names: vals[names.index(k.encode())] = v
                else: names.append(k.encode()); vals.append(v)
            out = (C.c_double * 10)()
            hist = (C.c_double * (7 * 8))()
            for _ in range(2):
                t = time.perf_counter()
                rc = lib.rdis_optba_run_hist(path.encode(), nc, npnt, 1, len(names), (C.c_char_p * len(names))(*names),
                                             (C.c_double * len(vals))(*vals), 0, out, None, hist, 8)
                wall = time.perf_counter() - t
            assert rc == 0, rc
            o = list(out)
            print("%s %s: %.6g -> %.9g; %d calls, %d CG iterations, %d launches (%.1f calls each), %.3f s (%.3f s wall)" % (
                label, extra or "", o[1], o[0], o[2], o[3], o[4], o[2] / max(o[4], 1), o[5], wall))
            for d in range(8):
                h = hist[7 * d:7 * d + 7]
                if h[0]:
                    print("    depth %d: %d nodes, %d free variables; steps: %d initial values, %d iterative improvement, %d random restart; "
                          "%d made no progress beyond steptol, %d new minima" % (d, h[0], h[1], h[2], h[3], h[4], h[5], h[6]))
