"""Timeline of the last PCA step from a rocprofv3 kernel trace: which kernels fill the time between the persistent tile
kernels. usage: timeline.py <dir with *_kernel_trace.csv> [n_tile_kernels_per_step=11]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
per = int(sys.argv[2]) if len(sys.argv) > 2 else 11
tiles = [i for i, r in enumerate(rows) if ("spmm_tile_kernel" in r[2] or "spmm_tile_dense_kernel" in r[2])]
first = tiles[-per]
# start at the normalize in front of the step: look back for the row_reduce (library sizes)
i0 = first
while i0 > 0 and "row_reduce_kernel<0>" not in rows[i0][2] and "row_reduceILi0" not in rows[i0][2]:
    i0 -= 1
t0 = rows[i0][0]
def short(n):
    for p in ("void ", "scanrs::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.split("(")[0][:60]
print(f"step: {(rows[-1][1] - t0) / 1e6:.2f} ms from the library-size pass to the last kernel")
cur_end = t0
busy_tile = 0.0
out = []
for s, e, n, q in rows[i0:]:
    out.append((s, e, short(n), q))
# merge runs of the same kernel name on the same queue
merged = []
for s, e, n, q in out:
    if merged and merged[-1][2] == n and merged[-1][3] == q and s - merged[-1][1] < 200_000:
        merged[-1] = (merged[-1][0], max(e, merged[-1][1]), n, q, merged[-1][4] + 1, merged[-1][5] + (e - s))
    else:
        merged.append((s, e, n, q, 1, e - s))
for s, e, n, q, c, busy in merged:
    if "gather_ov" in n:
        continue
    print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f}  q{q:>3}  x{c:<4d} busy {busy / 1e6:8.3f} ms  {n}")
# union of tile-kernel intervals vs the rest
tile_iv = [(s, e) for s, e, n, q in out if "spmm_tile_kernel" in n or "spmm_tile_dense_kernel" in n]
tt = sum(e - s for s, e in tile_iv)
print(f"tile kernels: {tt / 1e6:.2f} ms in {len(tile_iv)} launches; everything else on the critical path: {(rows[-1][1] - t0 - tt) / 1e6:.2f} ms")
gaps = []
for (s0, e0), (s1, e1) in zip(tile_iv, tile_iv[1:]):
    gaps.append((s1 - e0) / 1e6)
print("gaps between consecutive tile kernels (ms):", " ".join(f"{g:.2f}" for g in gaps))
