"""Follow-up to class_rw_probe.py: does the write class of a big allocation come from what was allocated BEFORE it?
mode fresh: K buffers of `bytes` straight away.   mode frag: first 400 x 64 MiB tensors, every other one freed (torch's
cache emptied so that the holes go back to the driver), then the K buffers.   mode small: 3 GiB of small live tensors first.
    python scripts/class_rw_probe2.py fresh|frag|small [K=8] [bytes=12000000000]"""
import sys, torch
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nbytes = int(sys.argv[3]) if len(sys.argv) > 3 else 12_000_000_000
keep = []
if mode == "frag":
    tmp = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(400)]
    keep = tmp[::2]
    del tmp
    torch.cuda.empty_cache()
elif mode == "small":
    keep = [torch.empty(3 << 20, dtype=torch.uint8, device="cuda") for _ in range(1000)]
n = nbytes // 8
bufs = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(K)]
def timed(fn):
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for b in bufs: b.fill_(3)
for rnd in range(2):
    print(f"{mode} round {rnd}: " + " ".join(f"{nbytes / timed(lambda: b.fill_(5)) / 1e6:5.0f}" for b in bufs) + " GB/s written", flush=True)
