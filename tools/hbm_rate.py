"""Achievable HBM bandwidth on the box, for the roofline discussion in DESIGN.md section 6: device-to-device copy (read + write),
read-only reduction and fill (write-only) of buffers far larger than the 256 MB MALL.  usage: python tools/hbm_rate.py"""
import json
import torch

n = 2 << 30  # 2 GiB per buffer
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(3)
out = {}


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: b.copy_(a))
out["copy_read_plus_write_TBps"] = round(2 * n / t / 1e12, 3)
ai = a.view(torch.int64)
t = timed(lambda: ai.sum())
out["read_only_sum_TBps"] = round(n / t / 1e12, 3)
t = timed(lambda: b.fill_(7))
out["write_only_fill_TBps"] = round(n / t / 1e12, 3)
print(json.dumps(out))
