"""oracle/levels.py -- TEST INFRASTRUCTURE (see oracle/README.md): an independent restatement, in Python, of the
decomposition the level driver computes (rdis_amd/host/rdis_levels.{h,cpp}; what RDISOptimizer's recursion does
around ssopt.optimize, reference src/RDISOptimizer.cpp:253-334), so that the C++ tree -- nodes, separators,
children, the per-depth launch lists -- can be compared bit for bit, and a sweep's launches re-run on the CPU oracle.

Only tests/ may import this.  Parity status: the recursion's shape follows the reference (choose a block of
variables, assign it, the rest falls apart into the connected components of the residual factor graph, children
ordered by number of variables, src/Component.cpp:508-549, :603-608; a component of at most AVblkpct x N
variables is optimised as a whole, src/RDISOptimizer.cpp:342, :1761); the separator RULE is this build's own
(the reference calls PaToH, a binary-only library, src/RDISOptimizer.cpp:779-865), so what is pinned here is
"the C++ implements the rule as stated", not "the rule is PaToH".  The rule, stated once (rdis_levels.h):

  * variable blocks (a camera's 9, a point's 3; every variable of a polynomial its own, getBlockRangeByVid) of a
    component are put back into an empty graph in order of ascending degree -- the number of the component's
    factors that read the block -- ties by ascending block id;
  * a block is put back if the connected piece it would join up (its own variables + the pieces it touches
    through a factor) has at most max_piece variables; the first block that does not fit, and every block
    after it, is a separator block;
  * then, like ensureFactorWillBeAssigned (src/RDISOptimizer.cpp:412-458): among the factors that read a
    separator block and still have other variables, the first (in list order) with the fewest such variables
    gives its remaining blocks to the separator -- unless some factor already lies entirely in the separator.

Written from that statement with sets and a fresh component search per step (quadratic, fine for test sizes),
not from the C++ (incremental union-find).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

from rdis_amd import problems as P


@dataclass
class Node:
    depth: int
    parent: int
    leaf: bool
    vars: np.ndarray
    factors: np.ndarray
    separator: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    sep_factors: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))


def factor_variables(pp):
    """list of int64 arrays: the variables each factor reads"""
    if pp.kind == P.KIND_BA:
        cam, pt = pp.cam_vid0, pp.pt_vid0
        return [np.concatenate([np.arange(cam[j], cam[j] + 9), np.arange(pt[j], pt[j] + 3)]) for j in range(pp.nfac)]
    return [pp.vid[pp.rowptr[j]:pp.rowptr[j + 1]] for j in range(pp.nfac)]


def block_start(pp, ncam_vars):
    """first variable id of the block of every variable (getBlockRangeByVid: BundleAdjustmentFunction.h:98-123 --
    a camera's nine, a point's three; a polynomial's variable is its own block)"""
    v = np.arange(pp.nvars, dtype=np.int64)
    if pp.kind != P.KIND_BA:
        return v
    return np.where(v < ncam_vars, v - v % 9, ncam_vars + (v - ncam_vars) - (v - ncam_vars) % 3)


def choose_separator(pp, fvars, bstart, comp_vars, comp_factors, max_piece):
    comp_vars = np.asarray(comp_vars, dtype=np.int64)
    inside = set(int(v) for v in comp_vars)
    blocks = sorted(set(int(bstart[v]) for v in comp_vars))               # ascending block id
    size = {b: 0 for b in blocks}
    for v in comp_vars:
        size[int(bstart[v])] += 1
    # the component's blocks each listed factor reads, in the order its variables name them
    fblocks = []
    for j in comp_factors:
        seen = []
        for v in fvars[int(j)]:
            if int(v) in inside:
                b = int(bstart[v])
                if b not in seen:
                    seen.append(b)
        fblocks.append(seen)
    degree = {b: 0 for b in blocks}
    for fb in fblocks:
        for b in fb:
            degree[b] += 1
    order = sorted(blocks, key=lambda b: (degree[b], b))
    back = set()            # blocks put back so far
    piece_of = {}           # block -> id of its piece; pieces: id -> set of blocks
    pieces = {}
    cut = len(order)
    for k, b in enumerate(order):
        touched = set()
        for fb in fblocks:
            if b in fb:
                for o in fb:
                    if o != b and o in back:
                        touched.add(piece_of[o])
        total = size[b] + sum(sum(size[o] for o in pieces[t]) for t in touched)
        if total > max_piece:
            cut = k
            break
        new_id = k
        merged = {b}
        for t in touched:
            merged |= pieces.pop(t)
        pieces[new_id] = merged
        for o in merged:
            piece_of[o] = new_id
        back.add(b)
    sep = set(order[cut:])
    if sep:
        best, best_j, whole = None, None, False
        for j, fb in enumerate(fblocks):
            if not any(b in sep for b in fb):
                continue
            left = sum(size[b] for b in fb if b not in sep)
            if left == 0:
                whole = True
                break
            if best is None or left < best:
                best, best_j = left, j
        if not whole and best is not None:
            sep |= set(fblocks[best_j])
    return np.array([int(v) for v in comp_vars if int(bstart[v]) in sep], dtype=np.int64)


def build_tree(pp, oracle_problem, blkpct=0.2, seppct=0.0) -> List[Node]:
    """the decomposition tree, nodes in the level driver's order (depth by depth; the children of all nodes split at
    a depth in the order one labelling of the residual graph returns them: by number of variables, then by
    smallest variable id, Component.cpp:603-608)"""
    fvars = factor_variables(pp)
    ncam_vars = int(pp.pt_vid0.min()) if pp.kind == P.KIND_BA else 0
    bstart = block_start(pp, ncam_vars)
    N = pp.nvars
    leaf_max = max(1, int(np.floor(blkpct * N + 0.5)))
    nodes: List[Node] = []
    fp, fv, cp, ci = oracle_problem.components(np.zeros(N, np.uint8))
    frontier = []
    for c in range(len(fp) - 1):
        nodes.append(Node(0, -1, True, fv[fp[c]:fp[c + 1]].copy(), ci[cp[c]:cp[c + 1]].copy()))
        frontier.append(len(nodes) - 1)
    depth = 0
    while frontier:
        assigned = np.ones(N, np.uint8)
        owner = {}
        split = []
        for ni in frontier:
            nd = nodes[ni]
            if len(nd.factors) == 0 or len(nd.vars) <= leaf_max:
                continue
            piece = max(leaf_max, int(np.floor(seppct * len(nd.vars) + 0.5))) if seppct > 0 else leaf_max
            sep = choose_separator(pp, fvars, bstart, nd.vars, nd.factors, piece)
            if len(sep) == 0 or len(sep) >= len(nd.vars):
                continue
            nd.leaf = False
            nd.separator = sep
            sset = set(int(v) for v in sep)
            nd.sep_factors = np.array([int(j) for j in nd.factors if any(int(v) in sset for v in fvars[int(j)])], dtype=np.int64)
            for v in nd.vars:
                if int(v) not in sset:
                    assigned[int(v)] = 0
                    owner[int(v)] = ni
            split.append(ni)
        frontier = []
        if not split:
            break
        fp, fv, cp, ci = oracle_problem.components(assigned)
        for c in range(len(fp) - 1):
            vs = fv[fp[c]:fp[c + 1]].copy()
            nodes.append(Node(depth + 1, owner[int(vs[0])], True, vs, ci[cp[c]:cp[c + 1]].copy()))
            frontier.append(len(nodes) - 1)
        depth += 1
    return nodes


def level_plans(nodes: List[Node]):
    """the launches of a sweep: per depth the separators of its split nodes (kind 0), then its leaves (kind 1);
    each as (depth, kind, node indices, free_ptr, free_vid, fac_ptr, fac_id)"""
    out = []
    for d in range(max(nd.depth for nd in nodes) + 1):
        for kind in (0, 1):
            idx = [i for i, nd in enumerate(nodes) if nd.depth == d and len(nd.factors) > 0 and (nd.leaf == (kind == 1))]
            if not idx:
                continue
            vs = [nodes[i].separator if kind == 0 else nodes[i].vars for i in idx]
            fs = [nodes[i].sep_factors if kind == 0 else nodes[i].factors for i in idx]
            out.append((d, kind, idx,
                        np.concatenate([[0], np.cumsum([len(v) for v in vs])]).astype(np.int64), np.concatenate(vs).astype(np.int64),
                        np.concatenate([[0], np.cumsum([len(f) for f in fs])]).astype(np.int64), np.concatenate(fs).astype(np.int64)))
    return out


def sweep(pp, oracle_problem, plans, x, maxiters=25, ftol=3e-8, oracle_for=None):
    """one sweep on the CPU oracle: every launch's components one after the other (they are independent), from x;
    returns (objective after every launch, x after the sweep).  oracle_for(plan) -> (free_vid, fac) -> OracleProblem: the oracle
    that stands for the device solver the dispatcher gives that launch's components (its arithmetic and sums), default: the
    reference-order oracle passed in"""
    x = np.array(x, dtype=np.float64)
    allv = np.arange(pp.nvars, dtype=np.int64)
    oracle_problem.assign(allv, x)
    f = oracle_problem.eval()
    objectives = []
    for plan in plans:
        (_d, _k, _idx, fp, fv, cp, ci) = plan
        make = oracle_for(plan) if oracle_for is not None else None
        for c in range(len(fp) - 1):
            v, fc = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
            if make is not None:
                oracle_problem = make(v, fc)
            oracle_problem.assign(allv, x)
            r = oracle_problem.cgd(free_vid=v, fac=fc, x=x[v], maxiters=maxiters, ftol=ftol)
            x[v] = r.x
            f += r.delta
        objectives.append(f)
    return np.array(objectives), x


# ---------------------------------------------------------------------------------------------------------------
# The reference's schedule at a node -- doOptimization / getValueFromDomain / getSSInitialVal / updateDomain -- restated
# from src/RDISOptimizer.cpp (not from rdis_levels.cpp), as a CHECKER: it walks the tree depth first, node by node and
# child by child, the way the reference does (:253-334), takes every decision itself, and where the reference would
# call the subspace optimizer (:1067) or read a component's value (:1514-1515) it consumes the next record of the
# device run's trace for that node.  It asserts that each record is the step it expects (kind of start, restart
# count, assignments since the last restart, and -- for a random restart -- the start vector bit for bit, through its
# hash) and that updateDomain's verdict is the recorded one; at the end every record must have been consumed.
# Parity status: the schedule is pinned by the reference's lines cited below; END-TO-END parity with the reference's
# optBA is NOT pinned and cannot be here -- its cut comes from PaToH (binary-only), its restart values from Boost's
# mt19937 in visiting order (:37, :1139, :1223-1225); this build draws them per (node, restart, variable).
M64 = (1 << 64) - 1


def splitmix_restart_value(seed, node, restart, vid, slo, shi, lo, hi):
    """sampleRandomState (:1196-1216): uniform over the sampling interval, then VariableDomain::closestVal; the
    uniform number from splitmix64 of (seed, node, restart, variable) -- rdis_levels.cpp: restartValue"""
    z = (seed + 0x9E3779B97F4A7C15 * (node + 1) + 0xBF58476D1CE4E5B9 * (restart + 1) + 0x94D049BB133111EB * (vid + 1)) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    z ^= z >> 31
    u = float(z >> 11) * (1.0 / 9007199254740992.0)
    v = slo + u * (shi - slo)
    return min(max(v, lo), hi)


def fnv_of_doubles(xs):
    h = 1469598103934665603
    for b in np.asarray(xs, dtype=np.float64).view(np.uint64):
        h = ((h ^ int(b)) * 1099511628211) & M64
    return h


def ba_sampling_intervals(pp, ncams):
    """per variable (sampling lo, sampling hi): BundleAdjustmentFunction::setDomain
    (src/bundleadjust/BundleAdjustmentFunction.cpp:402-477) from the file's initial values"""
    x0 = pp.x0
    typ = np.concatenate([np.arange(9 * ncams) % 9, 9 + np.arange(pp.nvars - 9 * ncams) % 3])
    slo, shi = np.empty(pp.nvars), np.empty(pp.nvars)
    rot = typ < 3
    slo[rot], shi[rot] = -1.0 * np.pi, 1.0 * np.pi                                    # :419-424
    pos = ((typ >= 3) & (typ <= 6)) | (typ >= 9)                                       # translation, focal, point: init +- 100 (:426-451)
    slo[pos], shi[pos] = x0[pos] + -100.0, x0[pos] + 100.0
    k1 = typ == 7
    slo[k1], shi[k1] = x0[k1] + -1e-4, x0[k1] + 1e-4                                   # :453-457
    k2 = typ == 8
    slo[k2], shi[k2] = x0[k2] + -1e-6, x0[k2] + 1e-6                                   # :458-462
    return slo, shi


def replay_reference_schedule(nodes: List[Node], trace, pp, slo, shi, steptol=1e-4, nrr_per_lvl=2, nrr_at_top=None, min_rr=1,
                              max_na_to_rr=10, no_assign_limit_at_top=True, seed=0x5D15, max_calls=100000):
    """trace: rows (node, kind, nrr, va, fret, delta, value, newMin, startHash) in the order the device run made them.
    Returns the number of subspace-optimizer calls checked."""
    if nrr_at_top is None:
        nrr_at_top = nrr_per_lvl                                                      # :1784-1785
    queues = {}
    for row in trace:
        queues.setdefault(int(row[0]), []).append(row)
    pos = {n: 0 for n in queues}
    children = [[] for _ in nodes]
    roots = []
    for i, nd in enumerate(nodes):
        if len(nd.factors) == 0:
            continue                                                                   # checkEmpty (:262)
        (children[nd.parent] if nd.parent >= 0 else roots).append(i)
    calls = [0]

    class TimedOut(Exception):
        pass

    def take(n):
        # the reference's time limit (checkTimedOut, :317-321; here a budget of subspace-optimizer calls): the device run stopped
        # asking once the budget was spent, and so do we -- what was recorded up to there must have been asked for in order
        if pos.get(n, 0) >= len(queues.get(n, [])) and sum(len(q) for q in queues.values()) >= max_calls:
            raise TimedOut()
        assert pos.get(n, 0) < len(queues.get(n, [])), "the device made fewer steps at node %d than the reference's rules ask for" % n
        row = queues[n][pos[n]]
        pos[n] += 1
        calls[0] += 1
        return row

    def do_optimization(n, random_init):                                               # :253-334, one visit of one component
        nd = nodes[n]
        top = nd.depth == 0                                                            # isTopCComp (:977-978): its parent has nothing assigned
        num_restarts = max(min_rr, nrr_at_top if top else (nrr_per_lvl >> min(31, nd.depth)))   # :979-983, level = depth + 1
        valued = nd.vars if nd.leaf else nd.separator
        nrr, va = 0, 0                                                                 # a fresh component (src/Component.cpp:183-184)
        assigned, last_opt, have_prev, have_opt, opt = False, False, False, False, 0.0
        while True:                                                                    # while ( getValueFromDomain(...) ) (:279)
            force_rr = (not (top and no_assign_limit_at_top)) and va >= max_na_to_rr   # :992-994
            if (force_rr or not last_opt) and nrr > num_restarts:                      # :997-999: done with this component
                return
            if not assigned and not random_init:
                kind = 0                                                               # xvalinit, "initial values" (:1127-1130)
            elif assigned and not force_rr:
                kind = 1                                                               # "iterative improvement" (:1131-1133)
            else:
                kind = 2                                                               # sampleRandomState (:1134-1136)
            success, redo = False, False
            while True:                                                                # do ... while ( redoGD && !success ) (:1032-1106)
                restart_no = nrr
                if kind != 1:
                    nrr += 1                                                           # incrementNumRandomRestarts (:1047) ...
                    va = 0                                                             # ... which also clears the counter (Component.h:190-194)
                row = take(n)                                                          # ssopt.optimize (:1067)
                assert (int(row[1]), int(row[2])) == (kind, nrr), ("node %d: step kind / restart count" % n, row[:4], kind, nrr)
                if kind == 2:
                    x0 = [splitmix_restart_value(seed, n, restart_no, int(v), slo[v], shi[v], pp.lo[v], pp.hi[v]) for v in valued]
                    assert fnv_of_doubles(x0) == int(row[8]), "node %d: the restart's start vector differs" % n
                fret, delta = float(row[4]), float(row[5])
                if have_prev and delta >= 0.0 - steptol:                               # approxgeq( deltafval, 0.0, ftol ) (:1086; common.h:74-76)
                    assert np.isnan(row[6]) and int(row[3]) == va, row
                    if nrr < num_restarts:                                             # :1087-1094: try again from a random position
                        kind, redo = 2, True                                           # doAlternatingMin = false (:1089)
                        continue
                    return                                                             # :1095-1099: failure -> getValueFromDomain returns false
                success = True
                break
            assert success
            assigned = True                                                            # assign (:282)
            va += 1                                                                    # Component::onVarsAssigned (src/Component.cpp:221)
            assert int(row[3]) == va, ("node %d: assignments since the last restart" % n, row[:4], va)
            # success && doAlternatingMin && sdprev -> setInitialValFromChildren( *sdprev ) (:1112-1114, 1713-1724): the children
            # get the values of the previous subdomain as their initial values -- also after a FORCED restart that made
            # progress at once (forceRR leaves doAlternatingMin set); only a restart after no progress (:1088-1091) or a
            # first visit from a random state leaves them with random initial values (:1162-1171)
            inherit = have_prev and not redo
            for c in children[n]:                                                      # decompose, children in their order (:289-314)
                do_optimization(c, kind == 2 and not inherit)
            value = float(row[6])                                                      # newsd->fx (:1515)
            if nd.leaf:
                assert value == fret
            is_new_min = False
            if not have_opt or value < opt:                                            # :1521-1525
                is_new_min = (not have_opt) or not (abs(value - opt) < steptol)         # approxeq (common.h:66-68)
                opt, have_opt = value, True
            assert int(row[7]) == int(is_new_min), ("node %d: updateDomain's verdict" % n, row, opt)
            last_opt, have_prev = is_new_min, True                                     # :1554, :1576

    try:
        for r in roots:
            do_optimization(r, False)
    except TimedOut:
        return -1   # (the budget cut the run short: a depth-first walk cannot tell which nodes the lock-step run still reached)
    for n, q in queues.items():
        assert pos[n] == len(q), "node %d: %d records of the device run were not asked for" % (n, len(q) - pos[n])
    return calls[0]
