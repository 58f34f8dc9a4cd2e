"""Device-memory stability: repeated normalize + BkSvd on fresh handles; free memory must not drift (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
dev = torch.device("cuda", 0)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
f0 = free()
for rep in range(3):
    ip, ix, vv = synth_counts_torch(100000, 33000, 0.03, rep, dev)
    m = sa.AdaptiveMat.from_device(33000, 100000, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
    del ip, ix, vv; torch.cuda.empty_cache()
    s = np.zeros(50)
    marks = []
    for it in range(12):
        m.reset_map(); sa.normalize(m, sa.Normalization.CellRanger)
        sa._check(sa._lib.scanrs_pca_bk(m._h, ctypes.c_uint32(50), ctypes.c_double(2.0), ctypes.c_uint32(5), ctypes.c_uint64(0), None, None, None, s.ctypes.data_as(ctypes.c_void_p), None))
        if it in (1, 11): marks.append(free())
    t = m.t(); v = m.view(); del t, v
    print(f"rep {rep}: free after PCA#2 {marks[0]:.0f} MiB, after PCA#12 {marks[1]:.0f} MiB")
    del m; torch.cuda.empty_cache()
    print(f"   after free: {free():.0f} MiB (start {f0:.0f})")
