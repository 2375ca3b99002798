"""HBM calibration on the box: write-only, read-only and copy bandwidth with torch kernels."""
import torch, time
dev = torch.device("cuda:0")
N = 1_200_000_000  # doubles: 9.6 GB
x = torch.empty(N, dtype=torch.float64, device=dev)
y = torch.empty(N, dtype=torch.float64, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(1.5)); print(f"fill 9.6GB: {ms:.3f} ms  write {9.6/ms*1e3:.0f} GB/s")
ms = t(lambda: x.zero_()); print(f"zero 9.6GB: {ms:.3f} ms  write {9.6/ms*1e3:.0f} GB/s")
ms = t(lambda: y.copy_(x)); print(f"copy 9.6GB: {ms:.3f} ms  r+w {19.2/ms*1e3:.0f} GB/s")
ms = t(lambda: x.sum()); print(f"sum 9.6GB: {ms:.3f} ms  read {9.6/ms*1e3:.0f} GB/s")
i32 = torch.empty(600_000_000, dtype=torch.int32, device=dev)
ms = t(lambda: i32.fill_(7)); print(f"fill 2.4GB i32: {ms:.3f} ms  write {2.4/ms*1e3:.0f} GB/s")
small = torch.empty(12_000_000, dtype=torch.float64, device=dev)  # 96 MB (fits MALL)
ms = t(lambda: small.fill_(1.0), 50); print(f"fill 96MB: {ms:.4f} ms  write {0.096/ms*1e3:.0f} GB/s")
