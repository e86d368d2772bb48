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


def sweep(pp, oracle_problem, plans, x, maxiters=25, ftol=3e-8):
    """one sweep on the CPU oracle: every launch's components one after the other (they are independent), from x;
    returns (objective after every launch, x after the sweep)"""
    x = np.array(x, dtype=np.float64)
    allv = np.arange(pp.nvars, dtype=np.int64)
    oracle_problem.assign(allv, x)
    f = oracle_problem.eval()
    objectives = []
    for (_d, _k, _idx, fp, fv, cp, ci) in plans:
        for c in range(len(fp) - 1):
            v, fc = fv[fp[c]:fp[c + 1]], ci[cp[c]:cp[c + 1]]
            oracle_problem.assign(allv, x)
            r = oracle_problem.cgd(free_vid=v, fac=fc, x=x[v], maxiters=maxiters, ftol=ftol)
            x[v] = r.x
            f += r.delta
        objectives.append(f)
    return np.array(objectives), x
