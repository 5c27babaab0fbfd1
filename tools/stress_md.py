#!/usr/bin/env python3
"""Flake hunt for the device-list path: the multi-device test's two inputs compressed over and over on
fresh and reused contexts; a mismatch is described (first differing byte, whether the streams still
decode, block counts).  usage: stress_md.py [iterations]"""
import bz2
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bzip2_amd  # noqa: E402
from bzip2_amd import corpus  # noqa: E402
import orc  # noqa: E402

orc.lib()


def describe(got, want, raw):
    k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
    try:
        ok = bz2.decompress(got) == raw
    except Exception as e:  # noqa: BLE001
        ok = repr(e)
    nd = sum(1 for i in range(min(len(got), len(want))) if got[i] != want[i])
    return f"first diff at {k} of {len(want)} (got {len(got)} bytes), {nd} bytes differ, decodes to the input: {ok}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    d3 = np.concatenate((corpus.mixed(6, 91, seg_bytes=700000), np.zeros(5_000_000, dtype=np.uint8),
                         corpus.mixed(5, 92, seg_bytes=600000)))
    want3, _ = orc.compress(d3, 3, threads=8)
    d9 = corpus.mixed(20, corpus.STREAM1G_SEED)
    want9, _ = orc.compress(d9, 9, threads=8)
    raw3, raw9 = d3.tobytes(), d9.tobytes()
    bad = 0
    c = None
    for it in range(n):
        if it % 5 == 0:
            if c is not None:
                c.close()
            devices = ([0, 0], [0, 0, 0], [0] * 5)[(it // 5) % 3]
            os.environ["BZ2HIP_STREAM_CHUNK_MIB"] = "2"
            try:
                c = bzip2_amd.Context(devices=devices, max_batch_blocks=16)
            finally:
                del os.environ["BZ2HIP_STREAM_CHUNK_MIB"]
        g3 = c.compress(d3, 3)
        if g3 != want3:
            bad += 1
            print(f"it {it} devices {devices} d3: {describe(g3, want3, raw3)}", flush=True)
        g9 = c.compress(d9, 9)
        if g9 != want9:
            bad += 1
            print(f"it {it} devices {devices} d9: {describe(g9, want9, raw9)}", flush=True)
        if it % 20 == 19:
            print(f"{it + 1} iterations, {bad} bad", flush=True)
    if c is not None:
        c.close()
    print(f"stress_md: {n} iterations, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
