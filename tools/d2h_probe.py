"""Host-link probe: device -> host copy rates of this box (pinned, two streams, pageable) and a single-thread host memcpy."""
import torch, time
d = torch.randn(50_000_000, device="cuda", dtype=torch.float64)  # 400 MB
h = torch.empty(50_000_000, dtype=torch.float64, pin_memory=True)
for _ in range(2): h.copy_(d); torch.cuda.synchronize()
t=time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t
print("1 stream pinned D2H: %.1f GB/s" % (0.4/dt))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t=time.perf_counter()
with torch.cuda.stream(s1): h[:25_000_000].copy_(d[:25_000_000], non_blocking=True)
with torch.cuda.stream(s2): h[25_000_000:].copy_(d[25_000_000:], non_blocking=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t
print("2 streams pinned D2H: %.1f GB/s" % (0.4/dt))
p = torch.empty(50_000_000, dtype=torch.float64)
t=time.perf_counter(); p.copy_(d); torch.cuda.synchronize(); dt=time.perf_counter()-t
print("pageable D2H: %.1f GB/s" % (0.4/dt))
import numpy as np
a = h.numpy(); b = np.empty_like(a)
t=time.perf_counter(); np.copyto(b, a); dt=time.perf_counter()-t
print("host memcpy 1 thread: %.1f GB/s" % (0.4/dt))
