"""Which of Brent's steps (nrc Dbrent, minimize_nrc.h:284-404) can be guessed before the reply to the pending trial is
known?  Replays a recorded trace of the solver's requests ([n, 4] records: tag, a, f, slope; tags 1 = value, 2 =
value + slope, 5 = end of a line minimisation -- e.g. numpy.save of plan.get_trace(...) or of the CPU restatement's
record) through a Python copy of Brent's bookkeeping and counts, per kind of reply, how often the bisection rule
("the trial becomes the second-best point, Brent bisects") and the second rule ("the trial is worse than x, w and v:
the state stands, the next step follows exactly") name the step the method really takes next.
usage: python predictor_rules.py trace.npy"""
import sys
import numpy as np, math
tr = np.load(sys.argv[1] if len(sys.argv) > 1 else 'trace.npy')
TOL = 3.0e-8; ZEPS = np.finfo(float).eps * 1e-3
lines = []; cur = []
for r in tr:
    tag = int(r[0])
    if tag in (1, 2): cur.append((tag, r[1], r[2], r[3]))
    if tag == 5: lines.append(cur); cur = []
def head(a, b, x, w, v, dx, dw, dv, d, e):
    """S_DB_HEAD: returns (converged, u, d, e)"""
    xm = 0.5 * (a + b); tol1 = TOL * abs(x) + ZEPS; tol2 = 2.0 * tol1
    if abs(x - xm) <= (tol2 - 0.5 * (b - a)): return True, None, d, e
    bisect = True
    if abs(e) > tol1:
        d1 = 2.0 * (b - a); d2 = d1
        if dw != dx: d1 = (w - x) * dx / (dx - dw)
        if dv != dx: d2 = (v - x) * dx / (dx - dv)
        u1 = x + d1; u2 = x + d2
        ok1 = (a - u1) * (u1 - b) > 0.0 and dx * d1 <= 0.0
        ok2 = (a - u2) * (u2 - b) > 0.0 and dx * d2 <= 0.0
        olde = e; e = d
        if ok1 or ok2:
            if ok1 and ok2: d = d1 if abs(d1) < abs(d2) else d2
            elif ok1: d = d1
            else: d = d2
            if abs(d) <= abs(0.5 * olde):
                ut = x + d
                if ut - a < tol2 or b - ut < tol2: d = math.copysign(tol1, xm - x)
                bisect = False
    if bisect:
        e = (a - x) if dx >= 0.0 else (b - x); d = 0.5 * e
    if abs(d) >= tol1: u = x + d
    else: u = x + math.copysign(tol1, d)
    return False, u, d, e
tot = hit_old = hit_new = 0
kinds = {}
for l in lines:
    fd = [(a, f, s) for (t, a, f, s) in l if t == 2]
    first_fd = next(i for i,(t,_,_,_) in enumerate(l) if t == 2) if any(t==2 for (t,_,_,_) in l) else 0
    fs = [(a, f) for (t, a, f, s) in l[:first_fd] if t == 1]
    if len(fd) < 2: continue
    # bracket: from F records: ax, bx, cx (we only need a = min(ax,cx), b = max)
    # reconstruct: points evaluated in bracketing (F tags) excluding the last F (line end re-evaluation?)
    xs = [p[0] for p in fs]
    x0 = fd[0][0]
    # a,b = nearest bracketing points around x0 among F points
    lo = max([p for p in xs if p < x0], default=None); hi = min([p for p in xs if p > x0], default=None)
    if lo is None or hi is None: continue
    a, b = lo, hi
    x = w = v = x0; fx = fw = fv = fd[0][1]; dx = dw = dv = fd[0][2]; d = e = 0.0
    conv, u, d, e = head(a, b, x, w, v, dx, dw, dv, d, e)
    for k in range(1, len(fd)):
        uu, fu, du = fd[k]
        if conv or uu != u:
            break  # reconstruction lost
        # predictions for the NEXT step, made before the reply is known
        # old predictor: assume worse + bisect
        na, nb = (uu, b) if uu < x else (a, uu)
        xm = 0.5 * (na + nb); tol1 = TOL * abs(x) + ZEPS
        eb = (na - x) if dx >= 0.0 else (nb - x); dd = 0.5 * eb
        p_old = x + dd if abs(dd) >= tol1 else x + math.copysign(tol1, dd)
        # new predictor: assume worse than x, w, v (state unchanged but bracket) and run the real head
        near = (w != x) and ((uu - x) * (w - x) > 0.0) and abs(uu - x) < abs(w - x)
        if w != x and v != x and v != w and not near:
            c2, p_new, _, _ = head(na, nb, x, w, v, dx, dw, dv, d, e)
        else:
            p_new = p_old
        # actual update
        if fu <= fx:
            if uu >= x: a = x
            else: b = x
            v, fv, dv = w, fw, dw; w, fw, dw = x, fx, dx; x, fx, dx = uu, fu, du
            kind = 'better'
        else:
            if uu < x: a = uu
            else: b = uu
            if fu <= fw or w == x:
                v, fv, dv = w, fw, dw; w, fw, dw = uu, fu, du; kind = 'worse,w<-u'
            elif fu < fv or v == x or v == w:
                v, fv, dv = uu, fu, du; kind = 'worse,v<-u'
            else: kind = 'worse'
        conv, u, d, e = head(a, b, x, w, v, dx, dw, dv, d, e)
        if conv: break
        tot += 1
        ho = (u == p_old); hn = (u == p_new)
        hit_old += ho; hit_new += hn
        kk = kind + (' old+new' if ho and hn else ' old only' if ho else ' new only' if hn else ' miss')
        kinds[kk] = kinds.get(kk, 0) + 1
print("brent steps", tot, "old predictor hits", hit_old, "new", hit_new)
for k in sorted(kinds): print("  ", k, kinds[k])
