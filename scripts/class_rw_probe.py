"""Do the write classes of multi-GiB allocations show in sequential READS as well?  K buffers of 12 GiB alive
at once; per buffer: a streaming write (fill_), a streaming read (sum over int64 words) and a copy INTO it from one fixed
source, best of 3 each, two rounds.   python scripts/class_rw_probe.py [K=6] [GiB=12]"""
import sys, torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = (G << 30) // 8
bufs = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(K)]
src = torch.empty(n, dtype=torch.int64, device="cuda"); src.fill_(1)
def timed(fn):
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for b in bufs: b.fill_(3)
for rnd in range(2):
    for k, b in enumerate(bufs):
        w = timed(lambda: b.fill_(5))
        r = timed(lambda: b.sum())
        c = timed(lambda: b.copy_(src))
        print(f"round {rnd} buffer {k} at {b.data_ptr():#x}: write {G * 1.0737 / w * 1e3:7.0f} GB/s  read {G * 1.0737 / r * 1e3:7.0f} GB/s  copy-into {2 * G * 1.0737 / c * 1e3:7.0f} GB/s (r+w)", flush=True)
