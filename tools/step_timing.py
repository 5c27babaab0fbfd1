import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bz2
from bzip2_amd import Context, corpus
data = corpus.mixed(256, corpus.MIXED256_SEED)
d_in = torch.from_numpy(data).cuda()
c = Context(device=0)
cap = c.L.bz2hip_compress_bound(data.size)
out = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
for it in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    n = c.compress_device(d_in.data_ptr(), data.size, 9, out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(it, round(dt * 1e3, 2), "ms", c.last_stats())
ok = bz2.decompress(out[:n].cpu().numpy().tobytes()) == data.tobytes()
print("roundtrip", ok)
