#!/usr/bin/env python3
"""Achievable HBM bandwidth on this GPU for plain streaming kernels (torch copy / fill / sum)."""
import torch, time
n = 1 << 28  # 2 GiB of u64
a = torch.empty(n, dtype=torch.int64, device="cuda")
b = torch.empty(n, dtype=torch.int64, device="cuda")
a.fill_(1)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    s = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - s) / reps
dt = t(lambda: b.copy_(a)); print(f"copy  : {2*n*8/dt/1e12:.2f} TB/s (read+write)")
dt = t(lambda: b.fill_(3)); print(f"fill  : {n*8/dt/1e12:.2f} TB/s (write)")
dt = t(lambda: a.sum());    print(f"sum   : {n*8/dt/1e12:.2f} TB/s (read)")
