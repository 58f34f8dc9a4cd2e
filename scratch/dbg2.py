import sys, os, time; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
import scanrs_amd as sa, scanrs_oracle as so
from scanrs_amd.synth import synth_counts
from threadpoolctl import threadpool_limits
nc, ng, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = synth_counts(nc, ng, 0.03, 5)
o = so.AdaptiveMat(ng, nc, so.CSC, m.indptr, m.indices, m.data)
t=time.time(); uo, s_o, vo = so.BkSvd().run_pca(so.normalize(o, "cellranger"), k); print("oracle", time.time()-t)
for lim in ("1e5", "1e300"):
    os.environ["SCANRS_REUSE_CMAX"] = lim
    g = sa.AdaptiveMat.from_csmat(ng, nc, sa.CSC, m.indptr, m.indices, m.data)
    g = sa.normalize(g, sa.Normalization.CellRanger)
    u, s, v = sa.BkSvd().run_pca(g, k)
    rel = np.abs(s - s_o)/s_o
    sign = np.sign(np.sum(u*uo, axis=0))
    du = np.abs(u*sign-uo).max(axis=0); dv = np.abs(v*sign-vo).max(axis=0)
    print("limit", lim, "sigma rel max %.2e"%rel.max(), "at", rel.argmax(), " u maxabs by comp:", np.array2string(du[[0,5,10,19,20,25,30,40,k-1]], precision=1), " v:", np.array2string(dv[[0,10,19,20,30,k-1]], precision=1))
print(s_o[[0,18,19,20,30,k-1]])
