"""ORACLE (test infrastructure only) -- restatement of the parts of
scikit-learn's KDTree that decide which of several EXACTLY equidistant points
`NearestNeighbors(n_neighbors=1, algorithm='kd_tree')` returns, i.e. the
reference's keypoint in every voxel whose centroid ties (graph_gen.py:84-88).

Only `tests/` may import this module.

Pinned: scikit-learn itself is installed (1.7.2), so every function here is
checked against the real thing -- `build` against `KDTree.get_arrays()`,
`nearest_with_ties` against `NearestNeighbors.kneighbors`
(tests/test_oracle_cpu.py).  The restatement exists to (a) document the rule
the HIP kernels implement (csrc/kdtree.hip, kdtree.h) and (b) give tests the
intermediate arrays.

Rule (sklearn/neighbors/_binary_tree.pxi.tp, _kd_tree.pyx.tp,
_partition_nodes.pyx, utils/_heap.pyx):
  * build: node i covers idx_array[s:e]; split dimension = first dimension of
    largest spread; `std::nth_element(idx+s, idx+s+n/2, idx+e)` with the
    strict total order (value, index); children (s, s+n/2), (s+n/2, e);
    n_levels = int(log2(max(1, (n-1)/leaf_size)) + 1), leaf_size = 30.
    `std::nth_element` = libstdc++ introselect (median-of-3 to `first`,
    unguarded Hoare partition, keep the side holding nth, insertion sort below
    4 elements).
  * query (depth-first, k = 1): a leaf is scanned in idx_array order; of two
    children the one with the smaller min_rdist is entered first (left on a
    tie); the heap replaces its entry only on a strictly smaller distance --
    the first point met at the minimum distance wins.
NOTE the version dependence: the reference does not pin scikit-learn
(README.md:27-31), and before scikit-learn 1.0 (PR #19473, cited in
_partition_nodes.pyx) `partition_node_indices` was a different quickselect, so
the tie winners are a property of the installed scikit-learn, not of the
reference.
"""
import numpy as np

LEAF_SIZE = 30


def _lt(data, d, a, b):
    va, vb = data[a, d], data[b, d]
    return (va < vb) | ((va == vb) & (a < b))


HEAP_SELECT_CALLS = [0]   # how often the fallback ran (tests look at it)


def _adjust_heap(data, d, idx, first, hole, length, value):
    """libstdc++ std::__adjust_heap (bits/stl_heap.h) followed by its
    __push_heap, on the max-heap idx[first:first+length] under _lt."""
    top = hole
    child = hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if _lt(data, d, idx[first + child], idx[first + child - 1]):
            child -= 1
        idx[first + hole] = idx[first + child]
        hole = child
    if (length & 1) == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        idx[first + hole] = idx[first + child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and _lt(data, d, idx[first + parent], value):
        idx[first + hole] = idx[first + parent]
        hole = parent
        parent = (hole - 1) // 2
    idx[first + hole] = value


def heap_select(data, d, idx, first, middle, last):
    """libstdc++ std::__heap_select(first, middle, last): what introselect
    falls back to when its depth limit (2 * floor(log2 n) partition rounds) is
    used up: a max-heap of [first, middle) swallows every later element that
    is smaller than its top."""
    HEAP_SELECT_CALLS[0] += 1
    length = middle - first
    if length >= 2:                      # __make_heap
        parent = (length - 2) // 2
        while True:
            _adjust_heap(data, d, idx, first, parent, length,
                         idx[first + parent])
            if parent == 0:
                break
            parent -= 1
    for i in range(middle, last):
        if _lt(data, d, idx[i], idx[first]):      # __pop_heap(first, middle, i)
            value = idx[i]
            idx[i] = idx[first]
            _adjust_heap(data, d, idx, first, 0, length, value)


def nth_element(data, d, idx, first, nth, last):
    """libstdc++ std::nth_element on idx[first:last] (in place)."""
    n = last - first
    if n == 0 or nth == last:
        return
    depth = 2 * (int(n).bit_length() - 1)
    while last - first > 3:
        if depth == 0:
            # __introselect: heap-select, then the nth element to its place
            heap_select(data, d, idx, first, nth + 1, last)
            idx[first], idx[nth] = idx[nth], idx[first]
            return
        depth -= 1
        mid = first + (last - first) // 2
        a, b, c = first + 1, mid, last - 1
        ia, ib, ic = idx[a], idx[b], idx[c]
        if _lt(data, d, ia, ib):            # __move_median_to_first
            if _lt(data, d, ib, ic):
                m = b
            elif _lt(data, d, ia, ic):
                m = c
            else:
                m = a
        elif _lt(data, d, ia, ic):
            m = a
        elif _lt(data, d, ib, ic):
            m = c
        else:
            m = b
        idx[first], idx[m] = idx[m], idx[first]
        piv = idx[first]
        seg = idx[first + 1:last]            # view
        less = _lt(data, d, seg, piv)
        c_less = int(less.sum())
        # __unguarded_partition in closed form: the k-th element > pivot from
        # the left is swapped with the k-th element < pivot from the right
        # while the former lies left of the latter
        g = np.flatnonzero(~less)
        s = np.flatnonzero(less)[::-1]
        k = min(len(g), len(s))
        sw = g[:k] < s[:k]
        gi, si = g[:k][sw], s[:k][sw]
        tmp = seg[gi].copy()
        seg[gi] = seg[si]
        seg[si] = tmp
        cut = first + 1 + c_less
        if cut <= nth:
            first = cut
        else:
            last = cut
    seg = idx[first:last]
    order = sorted(range(len(seg)), key=lambda i: (data[seg[i], d], seg[i]))
    idx[first:last] = seg[order]


def tree_shape(n, leaf_size=LEAF_SIZE):
    n_levels = int(np.log2(max(1, (n - 1) / leaf_size)) + 1)
    return n_levels, 2 ** n_levels - 1


def build(points, leaf_size=LEAF_SIZE):
    """-> (idx_array int64 [n], ranges {node: (s, e)}, bounds [n_nodes, 6])."""
    data = np.asarray(points, np.float64)
    n = data.shape[0]
    _, n_nodes = tree_shape(n, leaf_size)
    idx = np.arange(n)
    ranges = {}
    bounds = np.zeros((n_nodes, 6))
    stack = [(0, 0, n)]
    while stack:
        node, s, e = stack.pop()
        ranges[node] = (s, e)
        pts = data[idx[s:e]]
        bounds[node, :3] = pts.min(0)
        bounds[node, 3:] = pts.max(0)
        if 2 * node + 1 >= n_nodes or e - s < 2:
            continue
        spread = bounds[node, 3:] - bounds[node, :3]
        d, mx = 0, 0.0
        for j in range(data.shape[1]):
            if spread[j] > mx:
                mx, d = spread[j], j
        n_mid = (e - s) // 2
        nth_element(data, d, idx, s, s + n_mid, e)
        stack.append((2 * node + 2, s + n_mid, e))
        stack.append((2 * node + 1, s, s + n_mid))
    return idx, ranges, bounds


def min_rdist(b, pt):
    r = 0.0
    for j in range(3):
        d_lo = b[j] - pt[j]
        d_hi = pt[j] - b[3 + j]
        d = (d_lo + abs(d_lo)) + (d_hi + abs(d_hi))
        r += (0.5 * d) ** 2
    return r


def met_before(pa, pb, c, n, n_nodes, bounds):
    """idx_array slot pa is visited before slot pb by the depth-first query."""
    node, s, e = 0, 0, n
    while 2 * node + 1 < n_nodes:
        m = s + (e - s) // 2
        a_left, b_left = pa < m, pb < m
        if a_left == b_left:
            node = 2 * node + (1 if a_left else 2)
            if a_left:
                e = m
            else:
                s = m
            continue
        left_first = min_rdist(bounds[2 * node + 1], c) <= \
            min_rdist(bounds[2 * node + 2], c)
        return a_left == left_first
    return pa < pb


def nearest_with_ties(points, queries, radius):
    """Index of the nearest point per query, exact ties resolved by the rule
    above.  `radius`: any bound on the nearest distance (candidate filter).
    -> (indices [Q], number of queries that had an exact tie)."""
    from scipy.spatial import cKDTree
    data = np.asarray(points, np.float64)
    idx, _, bounds = build(data)
    n = len(idx)
    _, n_nodes = tree_shape(n)
    pos = np.empty(n, np.int64)
    pos[idx] = np.arange(n)
    t = cKDTree(data)
    out = np.empty(len(queries), np.int64)
    ties = 0
    for q, c in enumerate(np.asarray(queries, np.float64)):
        cand = np.array(t.query_ball_point(c, radius))
        diff = c[None] - data[cand]
        d = (diff[:, 0] ** 2 + diff[:, 1] ** 2) + diff[:, 2] ** 2
        tied = cand[d == d.min()]
        best = tied[0]
        ties += len(tied) > 1
        for p in tied[1:]:
            if met_before(pos[p], pos[best], c, n, n_nodes, bounds):
                best = p
        out[q] = best
    return out, ties
