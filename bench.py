#!/usr/bin/env python3
"""bench.py -- compress MB/s (uncompressed input) at -9 on N MI355X GPUs.

One step = one pass of the whole hot path (block split + RLE1 + CRC, BWT, MTF/RLE2, Huffman,
bit packing, framing) over the synthetic `mixed256` corpus (256 x 1 MiB segments, SURVEY §8(d)
config 3), input and output resident in HBM.  With N > 1 every rank compresses its own 256 MiB
shard of the `stream1g` generator (weak scaling, no collective on the data path).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel = the LSD radix scatter pass of the BWT) and `cpu_baseline` (the CPU oracle, timed on a
bounded sample of the same workload on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--segments", type=int, default=256, help="1 MiB segments per GPU")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--batch", type=int, default=0, help="blocks per batch (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample-mib", type=int, default=48)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP events around every kernel class (adds ~2 ms per step of event overhead)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal on a one-GPU box: BZ2_BENCH_BACKEND=gloo BZ2_BENCH_ONE_GPU=1 puts every rank on cuda:0
    backend = os.environ.get("BZ2_BENCH_BACKEND", "nccl")
    if os.environ.get("BZ2_BENCH_ONE_GPU"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from bzip2_amd import Context, corpus

    if world == 1:
        data = corpus.mixed(args.segments, corpus.MIXED256_SEED)
        workload = f"mixed{args.segments} ({args.segments} x 1 MiB segments, 8 kinds cycling)"
    else:
        data = corpus.mixed(args.segments, corpus.STREAM1G_SEED + rank * args.segments)
        workload = (f"stream1g shard: {args.segments} x 1 MiB segments per GPU, "
                    f"{world * args.segments} MiB total")
    n_in = int(data.size)
    d_in = torch.from_numpy(data).to(dev)
    ctx = Context(device=local_rank, max_batch_blocks=args.batch, profile=0 if args.no_profile else (2 if args.profile_all else 1))
    L = ctx.L
    cap = L.bz2hip_compress_bound(n_in)
    d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)

    def step():
        return ctx.compress_device(d_in.data_ptr(), n_in, args.level, d_out.data_ptr(), cap)

    out_len = 0
    for _ in range(args.warmup):
        out_len = step()
    ctx.profile_reset()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_len = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness guard on rank 0 (outside the timed region): the stream must round-trip
    ok = True
    if rank == 0:
        import bz2
        comp = d_out[:out_len].cpu().numpy().tobytes()
        try:
            ok = bz2.decompress(comp) == data.tobytes()
        except (OSError, ValueError):
            ok = False

    stats = ctx.last_stats()
    prof = ctx.profile_get() if not args.no_profile else {}

    res = None
    if rank == 0:
        total_bytes = n_in * world * args.steps
        value = total_bytes / dt / 1e6
        res = {
            "metric": "compress MB/s (uncompressed input) at -9",
            "value": round(value, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": workload, "level": args.level, "input_bytes_per_gpu": n_in,
                       "blocks_per_gpu": stats["blocks"], "compressed_bytes_rank0": out_len,
                       "roundtrip_ok": ok, "parallelism": f"blocks-sharded x{world}"},
        }
        if prof:
            # dominant kernel class = the one with the most GPU time among those with a defined
            # algorithmic byte count
            KERNEL = {"radix_scatter": "k2_radix_scatter", "radix_hist": "k2_radix_hist",
                      "bwt_keys": "k2_keys", "bwt_ssort": "k2_ssort", "mtf_replay": "k3_mtf_replay"}
            dom = max(KERNEL, key=lambda k: prof[k]["ms"])
            sc = prof[dom]
            launches = max(sc["launches"], 1)
            avg_ms = sc["ms"] / launches
            alg = sc["alg_bytes"] / launches
            achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            # HBM bytes per launch from the committed PMC passes of this same command
            # (profiles/r1_pmc_traffic.json, made by tools/profile_round.sh)
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
                    pmc = json.load(f)
                pmc_name = {"k2_ssort": "void k2_ssort<0>",
                            "k2_radix_scatter": "void k2_radix_scatter<false>",
                            "k2_radix_hist": "void k2_radix_hist<false>"}
                traffic = pmc[pmc_name.get(KERNEL[dom], KERNEL[dom])]["hbm_bytes_per_call"]
            except (OSError, ValueError, KeyError):
                traffic = None
            res["roofline"] = {"bound": "hbm", "kernel": KERNEL[dom],
                               "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                               "launches": launches, "alg_bytes_per_launch": round(alg)}
            res["kernel_rooflines"] = {
                KERNEL[k]: {"ms_per_step": round(prof[k]["ms"] / args.steps, 3),
                            "alg_GBs": round(prof[k]["alg_bytes"] / max(prof[k]["ms"], 1e-9) / 1e6, 1)}
                for k in KERNEL}
            # elapsed time of the BWT stage in the last step (host clock around its round loop, which
            # ends on a device round trip); the per-class event times overlap (two streams)
            t_bwt = stats["bwt_seconds"]
            n_rle = stats["rle_bytes"]
            # B_bwt = 15 n + 140 R n; our first sort reaches depth d0 = 5 symbols (byte
            # alphabets), which plain doubling from depth 1 reaches in log2(5) = 2.32 rounds
            b_bwt = 15.0 * n_rle + 140.0 * (stats["n_times_rounds"] + 2.32 * n_rle)
            res["bwt_stage"] = {"seconds_per_step": round(t_bwt, 5),
                                "alg_GBs": round(b_bwt / t_bwt / 1e9, 1) if t_bwt > 0 else None,
                                "frac_of_hbm_peak": round(b_bwt / t_bwt / 1e9 / HBM_PEAK_GBS, 4) if t_bwt > 0 else None,
                                "mean_doubling_rounds": round(stats["rounds_sum"] / max(stats["blocks"], 1), 2),
                                "periodic_blocks": stats["periodic_blocks"]}
            res["stage_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}
        # the decoder on the stream just written: informational, one pass, untimed
        # region of the headline metric
        d_back = torch.empty(n_in + 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        td = time.perf_counter()
        drc, dn = ctx.decompress_device(d_out.data_ptr(), out_len, d_back.data_ptr(), n_in)
        torch.cuda.synchronize()
        td = time.perf_counter() - td
        res["decode"] = {"value": round(n_in / td / 1e6, 1), "unit": "MB/s", "ms": round(td * 1e3, 1),
                         "ok": bool(drc == 0 and dn == n_in and torch.equal(d_back[:dn], d_in))}
        del d_back
        if not args.no_cpu and world == 1:  # the CPU baseline is timed at N = 1 only
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orc  # CPU oracle: test infrastructure, used here only as the timed baseline
            sample = data[: min(n_in, args.cpu_sample_mib << 20)]
            tc = time.perf_counter()
            o1, _ = orc.compress(sample, args.level, threads=1)
            t1 = time.perf_counter() - tc
            cores = min(os.cpu_count() or 1, 16)
            tc = time.perf_counter()
            o2, _ = orc.compress(sample, args.level, threads=cores)
            tm = time.perf_counter() - tc
            # the same sample through the GPU path must give the oracle's bytes (checker, not timed)
            ln = ctx.compress_device(d_in.data_ptr(), int(sample.size), args.level, d_out.data_ptr(), cap)
            same = d_out[:ln].cpu().numpy().tobytes() == bytes(o1) and bytes(o1) == bytes(o2)
            res["cpu_baseline"] = {"value": round(sample.size / t1 / 1e6, 2), "unit": "MB/s",
                                   "cores": 1, "kind": "port",
                                   "sample": f"first {sample.size >> 20} MiB of the same corpus, "
                                             f"C oracle -O2, 1 thread",
                                   "all_cores": {"value": round(sample.size / tm / 1e6, 2),
                                                 "cores": cores},
                                   "gpu_output_equals_oracle": same}
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
