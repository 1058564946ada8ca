"""Host mirror of the seed-extension DPs of the graph aligner ("next" row N2):
genomeGraph.LeftDynamicAln / RightDynamicAln (/root/reference/genomeGraph/search.go:234-321), batched on the GPU through
gnx_gsw_extend_batch.  The DP and the traceback run on the device; what stays here is Go slice bookkeeping.

The reference's `dynamicScore` argument: resetDynamicScore (search.go:104-107) gets it by value, so it resets nothing -- the
route the caller passes in is kept and the traced ops are merged into it with a `routeIdx` that restarts at 0 (so the first
traced op is compared with route[0], the next new one with route[1], ...).  `route` below is that incoming slice; None / empty
gives the plain run-length encoding in traceback order, which is what GraphSmithWatermanToGiraf's top-level calls see.
"""
from . import _lib, cigar


def _merge_route(route_in, runs):
    """The route-building loop of search.go:252-262 / 298-308 applied to the traced ops `runs` = [(len, op)], traceback order."""
    route = [cigar.Cigar(c.RunLength, c.Op) for c in (route_in or [])]
    idx = 0
    for run, op in runs:
        for _ in range(int(run)):
            if len(route) == 0:
                route.append(cigar.Cigar(1, op))
            elif route[idx].Op == op:
                route[idx].RunLength += 1
            else:
                route.append(cigar.Cigar(1, op))
                idx += 1
    return route


def DynamicAlnBatch(side, alphas, betas, scores, gapPen, routes=None):
    """side "left" / "right"; returns a list of (score, route, i, j) like the two Go functions."""
    s, ei, ej, ops, off = _lib.gsw_extend_batch(_lib.GNX_GSW_LEFT if side == "left" else _lib.GNX_GSW_RIGHT, scores, gapPen, alphas, betas)
    out = []
    for p in range(len(alphas)):
        runs = [(int(ops["run_length"][k]), cigar.from_col(ops["op"][k])) for k in range(int(off[p]), int(off[p + 1]))]
        rin = routes[p] if routes is not None else None
        if rin:
            route = _merge_route(rin, runs)
        else:
            route = [cigar.Cigar(r, o) for r, o in runs]
        out.append((int(s[p]), route, int(ei[p]), int(ej[p])))
    return out


def LeftDynamicAln(alpha, beta, scores, gapPen, route=None):
    """(score, route, i, j) -- search.go:234-276"""
    return DynamicAlnBatch("left", [alpha], [beta], scores, gapPen, [route])[0]


def RightDynamicAln(alpha, beta, scores, gapPen, route=None):
    """(score, route, maxI, maxJ) -- search.go:278-321"""
    return DynamicAlnBatch("right", [alpha], [beta], scores, gapPen, [route])[0]
