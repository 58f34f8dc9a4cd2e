"""TEST INFRASTRUCTURE ONLY: exhaustive k-nearest-neighbours, the restatement of the reference's own test oracle
`exhaustive_knn` (scan-rs/src/nn.rs:112-137): distances sqrt(sum (o_i - v_i)^2) accumulated left to right (nn.rs:101-108),
candidates sorted as (distance, index) tuples. The reference's production path asks a ball tree (`ball_tree` crate, a
crates.io dependency absent from /root/reference) for the same neighbours; among exactly equidistant points the tree's
traversal order decides (nn.rs:213-229, `test_symmetry`), which is unpinned here — parity is on the distance classes."""
import numpy as np


def _dist_rows(v, q):
    # left-to-right accumulation like the Rust loop (not numpy's pairwise sum)
    d = np.zeros(v.shape[0])
    for j in range(v.shape[1]):
        t = v[:, j] - q[j]
        d = d + t * t
    return np.sqrt(d)


def exhaustive_knn(v: np.ndarray, k: int) -> np.ndarray:
    v = np.asarray(v, dtype=np.float64)
    n = v.shape[0]
    assert k < n
    out = np.zeros((n, k), dtype=np.int64)
    for c in range(n):
        d = _dist_rows(v, v[c])
        order = np.lexsort((np.arange(n), d))
        order = order[order != c]
        out[c] = order[:k]
    return out


def exhaustive_find_nn(queries: np.ndarray, points: np.ndarray, k: int, include_self: bool) -> np.ndarray:
    """nn.rs:63-83 by exhaustive search; missing neighbours are u32::MAX like `T::max_value()`."""
    queries, points = np.asarray(queries, dtype=np.float64), np.asarray(points, dtype=np.float64)
    out = np.full((queries.shape[0], k), np.iinfo(np.uint32).max, dtype=np.int64)
    for c in range(queries.shape[0]):
        d = _dist_rows(points, queries[c])
        order = np.lexsort((np.arange(points.shape[0]), d))
        if not include_self:
            order = order[order != c]
        out[c, : min(k, len(order))] = order[:k]
    return out
