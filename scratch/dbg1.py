import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import numpy as np
import scanrs_amd as sa, scanrs_oracle as so
from test_gpu_parity import random_counts, pair, _norm_pair
rng = np.random.default_rng(5)
dense = random_counts(rng, 150, 400, 0.1, 30)
dense[:, 0] += 1; dense[0, :] += 1
for storage in (0,1):
    g, o = _norm_pair(sa, dense, storage, "cellranger")
    a, b = g.to_dense(), o.to_dense()
    print(storage, "dense", np.abs(a-b).max())
    for l in (1,3,50,100):
        q = rng.standard_normal((400, l))
        a, b = g.dot(q), o.dot(q)
        d = np.abs(a-b)
        print(" l", l, "dot maxdiff", d.max(), "rel", (d/(np.abs(b)+1e-300)).max(), np.unravel_index(d.argmax(), d.shape))
        ql = rng.standard_normal((l, 150))
        a, b = g.rdot(ql), o.rdot(ql)
        d = np.abs(a-b)
        print(" l", l, "rdot maxdiff", d.max(), np.unravel_index(d.argmax(), d.shape))
print("---- transposed views")
for storage in (0,1):
    g, o = _norm_pair(sa, dense, storage, "cellranger")
    gt, ot = g.t(), o.t()
    print(storage, "dense_t", np.abs(gt.to_dense()-ot.to_dense()).max())
    for l in (1,3):
        q = rng.standard_normal((400, l)); ql = rng.standard_normal((l, 150))
        for name, a, b in (("t.dot", gt.dot(ql.T.copy()), ot.dot(ql.T.copy())), ("t.rdot", gt.rdot(q.T.copy()), ot.rdot(q.T.copy()))):
            d = np.abs(a-b); print(" l", l, name, d.max(), np.unravel_index(d.argmax(), d.shape), a.shape)
