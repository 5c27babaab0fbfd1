import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bzip2_amd import Context, corpus
rng = np.random.default_rng(1)
for name, data in (("random", rng.integers(0, 256, 899000, dtype=np.uint8)), ("text", corpus.text(899000, 1))):
    c = Context(device=0)
    z = np.frombuffer(c.compress(data, 9), dtype=np.uint8)
    d_z = torch.from_numpy(z.copy()).cuda(); d_o = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc, n = c.decompress_device(d_z.data_ptr(), z.size, d_o.data_ptr(), data.size)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, "one block:", round(dt * 1e3, 1), "ms rc", rc, flush=True)
