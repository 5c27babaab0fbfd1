"""One compress_device of mixed256; with BZ2HIP_DEBUG=2 the library prints the S- and L-list sizes of every
doubling round to stderr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bzip2_amd import Context, corpus
data = corpus.mixed(256, corpus.MIXED256_SEED)
d_in = torch.from_numpy(data).cuda()
c = Context(device=0)
out = torch.empty(c.L.bz2hip_compress_bound(data.size) + 64, dtype=torch.uint8, device="cuda")
c.compress_device(d_in.data_ptr(), data.size, 9, out.data_ptr(), out.numel() - 64)
