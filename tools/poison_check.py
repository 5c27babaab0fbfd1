#!/usr/bin/env python3
"""Uninitialised-read hunt: fill free device memory with a pattern, release it, then compress through
several entry paths and compare with the oracle.  A workspace word that is read before it is written
then holds the pattern instead of whatever the last run left there.  usage: poison_check.py [rounds]"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bzip2_amd import corpus  # noqa: E402
import bzip2_amd  # noqa: E402
import orc  # noqa: E402

orc.lib()


def poison(value, gib=24):
    bufs = []
    for _ in range(gib):
        t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        t.fill_(value)
        bufs.append(t)
    torch.cuda.synchronize()
    del bufs
    torch.cuda.empty_cache()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    d3 = np.concatenate((corpus.mixed(6, 91, seg_bytes=700000), np.zeros(5_000_000, dtype=np.uint8),
                         corpus.mixed(5, 92, seg_bytes=600000)))
    want3, _ = orc.compress(d3, 3, threads=8)
    d9 = corpus.mixed(20, corpus.STREAM1G_SEED)
    want9, _ = orc.compress(d9, 9, threads=8)
    d1 = corpus.mixed(12, 5)
    want1, _ = orc.compress(d1, 1, threads=8)
    bad = 0
    for r in range(rounds):
        for value in (0xFF, 0x00, 0xA5, 0x01):
            for devices in (None, [0, 0], [0, 0, 0]):
                poison(value)
                os.environ["BZ2HIP_STREAM_CHUNK_MIB"] = "2"
                try:
                    c = bzip2_amd.Context(devices=devices, max_batch_blocks=16) if devices else bzip2_amd.Context()
                finally:
                    del os.environ["BZ2HIP_STREAM_CHUNK_MIB"]
                try:
                    for name, d, lv, want in (("d3", d3, 3, want3), ("d9", d9, 9, want9), ("d1", d1, 1, want1),
                                              ("d9b", d9, 9, want9)):
                        got = c.compress(d, lv)
                        if got != want:
                            bad += 1
                            k = next(i for i in range(min(len(got), len(want))) if got[i] != want[i]) if got[:len(want)] != want[:len(got)] else min(len(got), len(want))
                            print(f"MISMATCH round {r} fill {value:#x} devices {devices} case {name}: first diff at {k} of {len(want)}", flush=True)
                finally:
                    c.close()
        print(f"round {r} done, {bad} bad", flush=True)
    print(f"poison_check: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
