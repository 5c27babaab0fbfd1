"""Host <-> device transfer rates on the box, for the design of the host-buffer path
(bz2hip_compress, streaming): pageable vs pinned vs registered, and host memcpy rates."""
import ctypes as C
import time

import numpy as np
import torch

N = 256 << 20
a = np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8)
dev = torch.device("cuda", 0)
d = torch.empty(N, dtype=torch.uint8, device=dev)
hip = C.CDLL("libamdhip64.so")


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


ta = torch.from_numpy(a)
print("H2D pageable  GB/s", N / t(lambda: d.copy_(ta)) / 1e9)
tp = torch.empty(N, dtype=torch.uint8).pin_memory()
print("host memcpy pageable->pinned 1 thread GB/s", N / t(lambda: tp.copy_(ta)) / 1e9)
print("H2D pinned    GB/s", N / t(lambda: d.copy_(tp, non_blocking=True)) / 1e9)
back = torch.empty(N, dtype=torch.uint8)
print("D2H pageable  GB/s", N / t(lambda: back.copy_(d)) / 1e9)
print("D2H pinned    GB/s", N / t(lambda: tp.copy_(d, non_blocking=True)) / 1e9)
# register / unregister cost of a pageable buffer
ptr = a.ctypes.data
t0 = time.perf_counter()
rc = hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(N), 0)
t1 = time.perf_counter()
print("hipHostRegister 256 MiB rc", rc, "ms", (t1 - t0) * 1e3)
if rc == 0:
    print("H2D registered GB/s", N / t(lambda: d.copy_(ta, non_blocking=True)) / 1e9)
    t0 = time.perf_counter()
    hip.hipHostUnregister(C.c_void_p(ptr))
    print("hipHostUnregister ms", (time.perf_counter() - t0) * 1e3)
# threaded memcpy into pinned
import threading
for nt in (2, 4, 8, 16):
    chunks = np.array_split(np.arange(N), nt)
    def work(i):
        lo, hi = (N * i) // nt, (N * (i + 1)) // nt
        C.memmove(tp.data_ptr() + lo, ptr + lo, hi - lo)
    def run():
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        [x.start() for x in th]
        [x.join() for x in th]
    print(f"host memcpy pageable->pinned {nt} threads GB/s", N / t(run) / 1e9)
