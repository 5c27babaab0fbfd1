#!/usr/bin/env python3
"""Experiment: K contexts (own stream each) on one GPU, each compressing 1/K of mixed256 from its
own host thread; shows how much the latency-bound stages overlap with the bandwidth-bound ones."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bzip2_amd import Context, corpus

data = corpus.mixed(256, corpus.MIXED256_SEED)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
for K in (1, 2, 3, 4):
    # split on block boundaries so every part is a whole number of blocks
    ctxs = [Context(device=0) for _ in range(K)]
    starts = ctxs[0].split(data, 9)
    nb = len(starts)
    cuts = [int(starts[(nb * j) // K]) for j in range(K)] + [data.size]
    outs = [torch.empty(ctxs[0].L.bz2hip_compress_bound(cuts[j + 1] - cuts[j]) + 64, dtype=torch.uint8, device=dev) for j in range(K)]
    def work(j, reps):
        for _ in range(reps):
            ctxs[j].compress_device(d_in.data_ptr() + cuts[j], cuts[j + 1] - cuts[j], 9, outs[j].data_ptr(), outs[j].numel() - 64)
    def run(reps):
        th = [threading.Thread(target=work, args=(j, reps)) for j in range(K)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
    run(1)
    t = time.perf_counter(); run(3); dt = (time.perf_counter() - t) / 3
    print(f"K={K}: {dt*1e3:.1f} ms/step  {data.size/dt/1e6:.0f} MB/s", flush=True)
    for c in ctxs: c.close()
