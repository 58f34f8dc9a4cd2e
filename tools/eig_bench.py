import numpy as np, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
rng=np.random.default_rng(0)
for n,k in ((500,50),(1000,100)):
    b=rng.standard_normal((n,n)); a=b@b.T
    for rep in range(3):
        t=time.perf_counter(); w,z=sa.host_sym_eig_topk(a,k); dt=(time.perf_counter()-t)*1e3
        print(n,k,'%.2f ms'%dt, w[0], flush=True)
