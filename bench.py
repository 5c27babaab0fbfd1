#!/usr/bin/env python3
"""bench.py -- compress MB/s (uncompressed input) at -9 on N MI355X GPUs.

One step = one pass of the whole hot path (block split + RLE1 + CRC, BWT, MTF/RLE2, Huffman,
bit packing, framing) over synthetic data resident in HBM, ending with the complete .bz2 stream
resident in HBM.

N = 1: the `mixed256` corpus (256 x 1 MiB segments) through
       bz2hip_compress_device.
N > 1: ONE stream of N x 256 MiB of the `stream1g` generator (config 4; N = 4 is exactly the 1 GiB
       stream) sharded over the ranks: every rank holds its byte range (+ lookahead) in HBM and
       scans it, the block chain hops from rank to rank (one integer), every rank compresses its
       blocks, rank 0 gathers the bit segments and concatenates them in order on its GPU.  Scan,
       chain, compression, gather and stitch are all inside the timed region (`ship` = rank 0's
       gather + stitch time per stream).  After the clock rank 0 also runs the same stream through
       the OTHER front end, one process driving all N GPUs through bz2hip_opts.devices[] from host
       buffers (`in_process`), and checks the timed stream against it, against a single-GPU stream
       and through the GPU decoder.  Per-GPU work is fixed as N grows ("weak").
       `python bench.py --gpus N` without a launcher starts its own N ranks (a child
       torch.distributed.run); under a launcher whose world is not N it exits non-zero.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel by
GPU time) and, at N = 1, `cpu_baseline` (the CPU oracle timed on a bounded sample of the same
workload on this host) and `host_to_host` (the same step through bz2hip_compress from and to host
buffers, PCIe included: SURVEY §8(d)'s wall-clock definition; never `value`).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
KERNEL = {"radix_scatter": "k2_radix_scatter", "radix_hist": "k2_radix_hist",
          "bwt_keys": "k2_keys", "bwt_ssort": "k2_ssort", "mtf_replay": "k3_mtf_replay"}
PMC_NAME = {"k2_ssort": ["void k2_ssort<0>"],
            "k2_radix_scatter": ["void k2_radix_scatter<false, 8, true>", "void k2_radix_scatter<false>"],
            "k2_radix_hist": ["void k2_radix_hist<false, 8>", "void k2_radix_hist<false>"]}


def load_pmc():
    for name in ("r3_pmc_traffic.json", "r2c_pmc_traffic.json", "r2b_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return name, json.load(f)
        except (OSError, ValueError):
            continue
    return None, {}


def launch_decision(gpus, env):
    """What `bench.py --gpus N` does before anything touches torch or the GPU.
    -> ("run", None): this process is the single rank, or one rank of the right world;
       ("spawn", None): N > 1 and no launcher -> bench.py starts torch.distributed.run itself;
       ("error", message): launched with a world that is not N -> exit non-zero (never a line
       that says n_gpus: 1 for a --gpus 8 request)."""
    ws = env.get("WORLD_SIZE")
    if ws is None and env.get("RANK") is None:
        return ("run", None) if gpus <= 1 else ("spawn", None)
    try:
        world = int(ws if ws is not None else "1")
    except ValueError:
        return ("error", f"WORLD_SIZE={ws!r} is not a number")
    if world != max(gpus, 1):
        return ("error", f"--gpus {gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {gpus}` "
                         f"(it launches its own ranks) or with --nproc-per-node {gpus}")
    return ("run", None)


def spawn_ranks(gpus, argv):
    """One process per GPU as a CHILD process (never exec: this pool forbids replacing a process, and
    nothing here has touched the GPU yet anyway); its stdout -- the one JSON line -- passes through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] no launcher in the environment: starting {gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--segments", type=int, default=256, help="1 MiB segments per GPU")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--batch", type=int, default=0, help="blocks per batch (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host_to_host leg")
    ap.add_argument("--cpu-sample-mib", type=int, default=48)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP events around every kernel class (adds ~2 ms per step of event overhead)")
    args = ap.parse_args()
    what, msg = launch_decision(args.gpus, os.environ)
    if what == "error":
        print(f"[bench] {msg}", file=sys.stderr)
        sys.exit(2)
    if what == "spawn":
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries the ONE JSON line and nothing else: libraries write there too (gloo prints its
    # connection notices on stdout), so file descriptor 1 is pointed at stderr and the JSON line is
    # written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    seg_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The default group is gloo (host transport): the chain hop, sizes, CRCs, barriers and the
        # timing reduction are a few bytes each and must work whatever the state of RCCL.  RCCL
        # carries only the bit segments (GPU to GPU); its group is created lazily and probed before
        # the warm-up, and if creating or probing it raises on ANY rank every rank falls back to host
        # transport for the segments too (config.segment_transport says which one ran).
        dist.init_process_group(backend="gloo")
        if backend == "nccl":
            try:
                seg_group = dist.new_group(backend="nccl")
            except Exception as ex:
                print(f"[bench] rank {rank}: RCCL group could not be created ({ex!r})", file=sys.stderr)
                seg_group = None

    from bzip2_amd import Context, PinnedBuffer, corpus, parallel

    prof_level = 0 if args.no_profile else (2 if args.profile_all else 1)
    ctx = Context(device=local_rank, max_batch_blocks=args.batch, profile=prof_level)
    L = ctx.L
    seg = args.segments
    level = args.level

    cpu_comm = False
    if world == 1:
        data = corpus.mixed(seg, corpus.MIXED256_SEED)
        workload = f"mixed{seg} ({seg} x 1 MiB segments, 8 kinds cycling), one stream on one GPU"
        n_in = int(data.size)
        n_total = n_in
        d_in = torch.from_numpy(data).to(dev)
        cap = L.bz2hip_compress_bound(n_in)
        d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
        out_len = 0

        def step():
            return ctx.compress_device(d_in.data_ptr(), n_in, level, d_out.data_ptr(), cap)
    else:
        # one stream of world x seg MiB; rank r holds [lo, end) = its nominal range + lookahead
        n_total = world * seg * corpus.MIB
        lo, hi = parallel.nominal_ranges(n_total, world)[rank]
        # a block that starts in this rank's last segment ends within the next two; six more are
        # resident (not scanned) in case the chain asks for more lookahead (plan_rank grows it x4)
        look_segs, held_segs = 2, 8
        s0, s1 = lo // corpus.MIB, min(world * seg, hi // corpus.MIB + held_segs)
        parts = [corpus.segment(corpus._KINDS[i % 8], corpus.MIB, corpus.STREAM1G_SEED + i) for i in range(s0, s1)]
        mine = np.concatenate(parts)
        end = s1 * corpus.MIB
        d_slice = torch.from_numpy(mine).to(dev)
        n_in = hi - lo
        workload = (f"stream1g generator, ONE stream of {world * seg} MiB sharded over {world} GPUs "
                    f"({seg} MiB per GPU), ordered concatenation on rank 0's GPU")
        # Streams back to back through bzip2_amd.parallel.RankPipeline: two engines (contexts)
        # alternate between consecutive streams; while one compresses stream i, a plan thread runs
        # the scan and the chain hop of stream i + 1 on the other (the serial hop from rank to rank,
        # about 1 ms per rank, is hidden behind compression instead of idling every GPU after the
        # first), and this thread ships stream i - 1's segment to rank 0, which stitches it.  Every
        # stream is scanned, chained, compressed, gathered and stitched in full inside the timed
        # region.
        ctxs = [ctx, Context(device=local_rank, max_batch_blocks=args.batch, profile=0)]
        engs = [parallel.GpuShardEngine(c, local_rank) for c in ctxs]
        for e in engs:
            e.bind_device(d_slice, lo, end)
            e.compress = e.compress_device
            e.empty = lambda: (d_none, 0, np.zeros(0, dtype=np.uint32))
        # gloo groups for the host-side messages, one per thread that uses it
        hop_group = dist.new_group(backend="gloo")   # the block chain: one int64 per hop (plan thread)
        meta_group = dist.new_group(backend="gloo")  # segment sizes and block CRCs (this thread)
        cap_seg = L.bz2hip_compress_bound(n_in + held_segs * corpus.MIB)
        seg_words = (cap_seg + 3) // 4 * 4
        d_none = torch.zeros(8, dtype=torch.uint8, device=dev)
        # rank 0: receive buffers for every other rank's segment, the stream, and a context of its
        # own for the stitch (the two above are busy compressing)
        if rank == 0:
            d_recv = [None] + [torch.empty(seg_words + 64, dtype=torch.uint8, device=dev) for _ in range(1, world)]
            cap = L.bz2hip_compress_bound(n_total)
            d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
            ctx_stitch = Context(device=local_rank, profile=0)
        cpu_comm = seg_group is None
        hop_t = torch.zeros(1, dtype=torch.int64)
        MAXCRC = n_in // (level * 50000) + 16

        def recv_first():
            dist.recv(hop_t, src=rank - 1, group=hop_group)
            return int(hop_t.item())

        def send_next(v):
            dist.send(torch.tensor([v], dtype=torch.int64), dst=rank + 1, group=hop_group)

        def plan(eng):
            return parallel.plan_rank(eng, level, rank, world, n_total, recv_first, send_next,
                                      lookahead=min(end - hi, look_segs * corpus.MIB) if rank + 1 < world else 0)

        def wait_current_stream():
            # the transfers are ordered on torch's current stream; wait for that stream only (a
            # device-wide synchronize would also wait for the next stream's compression)
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()

        ship_times = []  # rank 0: (gather seconds, stitch seconds) per stream

        def ship(eng, seg):
            t_ship = time.perf_counter()
            d_seg, nbits, crcs = seg
            # gather: sizes + CRCs (host, tiny), then the bit segments (device to device)
            meta = torch.zeros(2 + MAXCRC, dtype=torch.int64)
            meta[0] = nbits
            meta[1] = len(crcs)
            meta[2:2 + len(crcs)] = torch.from_numpy(crcs.astype(np.int64))
            metas = [torch.zeros_like(meta) for _ in range(world)] if rank == 0 else None
            dist.gather(meta, metas, dst=0, group=meta_group)
            nbytes = (nbits + 31) // 32 * 4
            if rank == 0:
                ops, segs = [], []
                for r in range(world):
                    nb_r, nc_r = int(metas[r][0]), int(metas[r][1])
                    cr = metas[r][2:2 + nc_r].numpy().astype(np.uint32)
                    if r == 0:
                        segs.append((d_seg.data_ptr(), nb_r, cr))
                        continue
                    k = (nb_r + 31) // 32 * 4
                    if k:
                        if cpu_comm:
                            buf = torch.empty(k, dtype=torch.uint8)
                            dist.recv(buf, src=r, group=meta_group)
                            d_recv[r][:k].copy_(buf)
                        else:
                            ops.append(dist.P2POp(dist.irecv, d_recv[r][:k], r, group=seg_group))
                    segs.append((d_recv[r].data_ptr(), nb_r, cr))
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                wait_current_stream()
                t_got = time.perf_counter()
                n_out = ctx_stitch.stitch_device(level, segs, d_out.data_ptr(), cap)  # returns with the stream complete
                ship_times.append((t_got - t_ship, time.perf_counter() - t_got))
                return n_out
            if nbytes:
                if cpu_comm:
                    dist.send(d_seg[:nbytes].cpu(), dst=0, group=meta_group)
                else:
                    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, d_seg[:nbytes], 0, group=seg_group)]):
                        w.wait()
                    wait_current_stream()  # the segment has left before its buffer is reused
            return 0

        pipe = parallel.RankPipeline(engs, plan, ship, thread_init=lambda: torch.cuda.set_device(local_rank))

        def run_streams(k):
            """k streams back to back; returns the last one's length (rank 0).  The plan of the first
            is requested here, so nothing of a timed stream happens before the clock starts."""
            res = pipe.run(k)
            return res[-1] if res else 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=meta_group)
        torch.cuda.synchronize()

    transport = None

    if world == 1:
        def run_streams(k):
            out = 0
            for _ in range(k):
                out = step()
            return out
    else:
        # connections are made on first use (RCCL point-to-point pairs, gloo pairs): open every one
        # this run uses with a 8-byte message, so that --warmup 0 does not time connection set-up
        z = torch.zeros(1, dtype=torch.int64)
        dist.gather(z, [torch.zeros_like(z) for _ in range(world)] if rank == 0 else None, dst=0, group=meta_group)
        if rank > 0:
            dist.recv(z, src=rank - 1, group=hop_group)
        if rank + 1 < world:
            dist.send(z, dst=rank + 1, group=hop_group)
        # the host path for segments is opened in any case (it is the fall-back)
        if rank == 0:
            for r in range(1, world):
                dist.recv(z, src=r, group=meta_group)
        else:
            dist.send(z, dst=0, group=meta_group)
        ok = 0 if seg_group is None else 1
        if seg_group is not None:
            try:
                dz = torch.zeros(8, dtype=torch.uint8, device=dev)
                if rank == 0:
                    ops = [dist.P2POp(dist.irecv, d_recv[r][:8], r, group=seg_group) for r in range(1, world)]
                else:
                    ops = [dist.P2POp(dist.isend, dz, 0, group=seg_group)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                torch.cuda.synchronize()
            except Exception as ex:  # the run still measures the path; the segments then travel through the host
                ok = 0
                print(f"[bench] rank {rank}: RCCL point-to-point set-up failed ({ex!r}); "
                      f"segments will be gathered through host memory (gloo)", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=meta_group)
        if int(flag.item()) == 0:
            cpu_comm = True
        transport = ("host memory (gloo)" if cpu_comm else "RCCL point-to-point, GPU to GPU")
    out_len = run_streams(args.warmup)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    out_len = run_streams(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=meta_group)
        dt = float(t.item())

    stats = ctx.last_stats()
    prof = ctx.profile_get() if not args.no_profile else {}
    # streams the profiled context handled inside the timed region (N > 1: every second one)
    prof_steps = args.steps if world == 1 else max(1, sum(1 for j in range(args.warmup, args.warmup + args.steps) if j % 2 == 0))

    # ---- after the clock (N > 1): the whole input goes to rank 0's host memory (every rank sends the
    # bytes of its nominal range over gloo: nobody regenerates another rank's data), and rank 0
    #   * times the OTHER front end of the same path: ONE process, Context(devices=[0..N-1]),
    #     bz2hip_compress from and to page-locked host buffers -- a worker thread and a block queue per
    #     GPU, ordered concatenation on the calling thread (S/BZip2OutputStream.swift:29,94-102,131-141);
    #     PCIe both ways is inside that clock, so it is reported beside `value`, never as `value`;
    #   * checks the stream of the timed region: SHA-256 against the in-process stream, the GPU decoder's
    #     round trip, and against a single-GPU stream of the same bytes (whole up to 512 MiB, else the
    #     first 256 MiB cut at a block boundary: the streams agree up to the shorter one's trailer).
    check, in_process = {}, None
    if world > 1:
        ship_stats = None
        if rank == 0 and ship_times:
            tail = ship_times[-args.steps:] if args.steps else ship_times
            ship_stats = {"gather_ms": round(sum(t[0] for t in tail) / len(tail) * 1e3, 3),
                          "stitch_ms": round(sum(t[1] for t in tail) / len(tail) * 1e3, 3)}
        pipe.close()
        pipe = None
        for e in engs:
            e.d_out = None
        ctxs[1].close()
        if rank == 0:
            pin_whole = PinnedBuffer(n_total)
            pin_whole.a[lo:hi] = mine[: hi - lo]
            for r in range(1, world):
                rlo, rhi = parallel.nominal_ranges(n_total, world)[r]
                dist.recv(torch.from_numpy(pin_whole.a[rlo:rhi]), src=r, group=meta_group)
        else:
            dist.send(torch.from_numpy(mine[: hi - lo]), dst=0, group=meta_group)
    if rank == 0 and world > 1:
        comp = d_out[:out_len].cpu().numpy().tobytes()
        check["sha256"] = hashlib.sha256(comp).hexdigest()
        one_gpu = bool(os.environ.get("BZ2_BENCH_ONE_GPU"))
        devices = [0] * world if one_gpu else list(range(world))
        try:
            cap_w = L.bz2hip_compress_bound(n_total)
            pin_out = PinnedBuffer(cap_w)
            ctx_ip = Context(devices=devices, max_batch_blocks=args.batch, profile=0)
            got = ctx_ip.compress_into(pin_whole.a, level, pin_out.a)  # allocates the workspaces
            k_ip = max(1, min(args.steps, 5))
            th = time.perf_counter()
            for _ in range(k_ip):
                got = ctx_ip.compress_into(pin_whole.a, level, pin_out.a)
            th = (time.perf_counter() - th) / k_ip
            in_process = {"value": round(n_total / th / 1e6, 1), "unit": "MB/s", "ms_per_step": round(th * 1e3, 2),
                          "steps": k_ip, "devices": devices,
                          "what": "one process, bz2hip_opts.devices[]: host buffer in, host buffer out (PCIe inside the clock)",
                          "same_bytes": bool(got == out_len and
                                             hashlib.sha256(pin_out.a[:got].tobytes()).hexdigest() == check["sha256"])}
            ctx_ip.close()
            pin_out.free()
        except Exception as ex:  # reported, never silently dropped
            in_process = {"error": repr(ex)}
        # single-GPU stream of the same bytes
        c1 = Context(device=local_rank)
        if n_total <= (512 << 20):
            k_cut, how = n_total, "whole stream"
        else:
            d_head = torch.from_numpy(pin_whole.a[: (320 << 20)]).to(dev)
            c1.shard_scan(d_head.data_ptr(), int(d_head.numel()), level)
            _, k_cut = c1.shard_chain(0, 256 << 20, False)
            how = f"first {k_cut} bytes (a block boundary): equal up to the shorter stream's trailer"
            del d_head
        d_head = torch.from_numpy(pin_whole.a[:k_cut]).to(dev)
        cap1 = L.bz2hip_compress_bound(k_cut)
        d_one = torch.empty(cap1 + 64, dtype=torch.uint8, device=dev)
        n1 = c1.compress_device(d_head.data_ptr(), k_cut, level, d_one.data_ptr(), cap1)
        one = d_one[:n1].cpu().numpy().tobytes()
        check["equals_single_gpu_stream"] = (one == comp) if k_cut == n_total else (one[: n1 - 11] == comp[: n1 - 11])
        check["single_gpu_check"] = how
        del d_head, d_one
        # the GPU decoder returns the input
        d_whole = torch.from_numpy(pin_whole.a).to(dev)
        d_back = torch.empty(n_total + 64, dtype=torch.uint8, device=dev)
        drc, dn = c1.decompress_device(d_out.data_ptr(), out_len, d_back.data_ptr(), n_total)
        check["roundtrip_ok"] = bool(drc == 0 and dn == n_total and torch.equal(d_back[:dn], d_whole))
        c1.close()
        del d_whole, d_back
        pin_whole.free()
    elif rank == 0:
        import bz2
        comp = d_out[:out_len].cpu().numpy().tobytes()
        try:
            check["roundtrip_ok"] = bz2.decompress(comp) == data.tobytes()
        except (OSError, ValueError):
            check["roundtrip_ok"] = False

    res = None
    if rank == 0:
        total_bytes = n_total * args.steps
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
            "config": {"workload": workload, "level": level, "input_bytes_per_gpu": n_in,
                       "stream_bytes": n_total, "blocks_rank0": stats["blocks"],
                       "compressed_bytes": out_len, "parallelism": f"blocks-sharded x{world}",
                       **({"segment_transport": transport} if transport else {}), **check},
        }
        if world > 1:
            res["ship"] = ship_stats       # rank 0's share per stream inside the clock: receive + stitch
            res["in_process"] = in_process  # the same streams through ONE process driving all N GPUs
        if prof:
            # dominant kernel class = the one with the most GPU time among those with a defined
            # algorithmic byte count
            dom = max(KERNEL, key=lambda k: prof[k]["ms"])
            sc = prof[dom]
            launches = max(sc["launches"], 1)
            avg_ms = sc["ms"] / launches
            alg = sc["alg_bytes"] / launches
            achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            # HBM bytes per launch from the committed PMC passes of this same command at N = 1
            # (profiles/r*_pmc_traffic.json, made by tools/profile_round.sh)
            pmc_file, pmc = load_pmc()
            if not (world == 1 and seg == 256 and level == 9):
                pmc_file, pmc = None, {}  # the PMC passes were made on the N = 1 headline workload
            traffic = None
            for nm in PMC_NAME.get(KERNEL[dom], [KERNEL[dom]]):
                if nm in pmc:
                    traffic = pmc[nm]["hbm_bytes_per_call"]
                    break
            res["roofline"] = {"bound": "hbm", "kernel": KERNEL[dom],
                               "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": pmc_file,
                               "avg_launch_ms": round(avg_ms, 4),
                               "launches": launches, "alg_bytes_per_launch": round(alg)}
            res["kernel_rooflines"] = {
                KERNEL[k]: {"ms_per_step": round(prof[k]["ms"] / prof_steps, 3),
                            "alg_GBs": round(prof[k]["alg_bytes"] / max(prof[k]["ms"], 1e-9) / 1e6, 1)}
                for k in KERNEL}
            # elapsed time of the BWT stage in the last step (host clock around its round loop, which
            # ends on a device round trip); the per-class event times overlap (two streams)
            t_bwt = stats["bwt_seconds"]
            n_rle = stats["rle_bytes"]
            # B_bwt = 15 n + 140 R n; our first sort reaches depth d0 = 5 symbols (byte
            # alphabets), which plain doubling from depth 1 reaches in log2(5) = 2.32 rounds.  This is
            # the byte count of a FULL-WIDTH sorter, divided by the time of one that skips most of that
            # work: an equivalent rate, not achieved bandwidth.  physical_frac = bytes the k2 kernels
            # really moved (PMC, at N = 1) / that time / peak.
            b_bwt = 15.0 * n_rle + 140.0 * (stats["n_times_rounds"] + 2.32 * n_rle)
            phys = sum(v["hbm_bytes_per_call"] * v["calls"] for k, v in pmc.items()
                       if "k2_" in k and "hbm_bytes_per_call" in v) if pmc else None
            res["bwt_stage"] = {"seconds_per_step": round(t_bwt, 5),
                                "full_width_equivalent_GBs": round(b_bwt / t_bwt / 1e9, 1) if t_bwt > 0 else None,
                                "frac_full_width_equivalent": round(b_bwt / t_bwt / 1e9 / HBM_PEAK_GBS, 4) if t_bwt > 0 else None,
                                "physical_GBs": round(phys / t_bwt / 1e9, 1) if (phys and t_bwt > 0) else None,
                                "physical_frac": round(phys / t_bwt / 1e9 / HBM_PEAK_GBS, 4) if (phys and t_bwt > 0) else None,
                                "mean_doubling_rounds": round(stats["rounds_sum"] / max(stats["blocks"], 1), 2),
                                "periodic_blocks": stats["periodic_blocks"]}
            res["stage_ms_per_step"] = {k: round(v["ms"] / prof_steps, 3) for k, v in prof.items() if v["launches"]}
        if world == 1:
            # the decoder on the stream just written: informational
            d_back = torch.empty(n_in + 64, dtype=torch.uint8, device=dev)
            for _ in range(2):  # the first pass allocates the decoder's workspace; the second is reported
                torch.cuda.synchronize()
                td = time.perf_counter()
                drc, dn = ctx.decompress_device(d_out.data_ptr(), out_len, d_back.data_ptr(), n_in)
                torch.cuda.synchronize()
                td = time.perf_counter() - td
            res["decode"] = {"value": round(n_in / td / 1e6, 1), "unit": "MB/s", "ms": round(td * 1e3, 1),
                             "ok": bool(drc == 0 and dn == n_in and torch.equal(d_back[:dn], d_in))}
            del d_back
            # several streams per call (bz2hip_decompress_streams): the stream just written, four times
            # over, so that the per-block Huffman chains of > 1000 blocks run side by side
            reps = 4
            d_cat = d_out[:out_len].repeat(reps)
            d_back = torch.empty(n_in * reps, dtype=torch.uint8, device=dev)
            for _ in range(2):
                torch.cuda.synchronize()
                td = time.perf_counter()
                drc, dn, dns = ctx.decompress_streams_device(d_cat.data_ptr(), out_len * reps, d_back.data_ptr(), n_in * reps)
                torch.cuda.synchronize()
                td = time.perf_counter() - td
            res["decode_streams"] = {"value": round(n_in * reps / td / 1e6, 1), "unit": "MB/s", "ms": round(td * 1e3, 1),
                                     "streams": int(dns), "blocks": int(ctx.last_stats()["blocks"]),
                                     "ok": bool(drc == 0 and dn == n_in * reps and dns == reps and
                                                torch.equal(d_back[n_in * (reps - 1):], d_in) and torch.equal(d_back[:n_in], d_in))}
            del d_back, d_cat
        if world == 1 and not args.no_host:
            # SURVEY §8(d)'s definition: first byte submitted to last output byte in host memory,
            # through bz2hip_compress (PCIe both ways inside the clock).  Never `value`.
            h2h = {}
            ctx_h = Context(device=local_rank, max_batch_blocks=args.batch, profile=0)  # no event pairs around kernels
            out_host = np.empty(cap, dtype=np.uint8)
            pin_in, pin_out = PinnedBuffer(n_in), PinnedBuffer(cap)
            pin_in.a[:] = data
            for name, src, dst in (("pageable", data, out_host), ("pinned", pin_in.a, pin_out.a)):
                ctx_h.compress_into(src, level, dst)
                th = time.perf_counter()
                for _ in range(max(args.steps, 1)):
                    got = ctx_h.compress_into(src, level, dst)
                th = (time.perf_counter() - th) / max(args.steps, 1)
                h2h[name] = {"value": round(n_in / th / 1e6, 1), "unit": "MB/s", "ms_per_step": round(th * 1e3, 2),
                             "same_bytes": bool(got == out_len and dst[:got].tobytes() == comp)}
            h2h["ratio_pinned_to_device_resident"] = round(h2h["pinned"]["value"] / value, 3)
            # the incremental API (BZip2OutputStream.write into a sink): 1 MiB writes, output read
            # after every write; the context's worker thread compresses while the caller writes
            # (the output is read into the caller's own buffer, as a C caller's sink would: bz2hip_stream_read
            # copies from the library's landing buffer to wherever it is told to)
            def stream_pass():
                pos = 0
                ctx_h.stream_begin(level)
                for o in range(0, n_in, 1 << 20):
                    ctx_h.stream_write(data[o:o + (1 << 20)])
                    while ctx_h.stream_pending():
                        pos += ctx_h.stream_read_into(out_host, pos)
                ctx_h.stream_finish()
                while ctx_h.stream_pending():
                    pos += ctx_h.stream_read_into(out_host, pos)
                return pos
            for _ in range(3):  # the stream's buffers (page-locked chunks, output queue) reach their size
                stream_pass()
            th = time.perf_counter()
            for _ in range(max(args.steps, 1)):
                got = stream_pass()
            th = (time.perf_counter() - th) / max(args.steps, 1)
            h2h["stream_api_1MiB_writes"] = {"value": round(n_in / th / 1e6, 1), "unit": "MB/s",
                                             "ms_per_step": round(th * 1e3, 2),
                                             "same_bytes": bool(got == out_len and out_host[:got].tobytes() == comp)}
            res["host_to_host"] = h2h
            ctx_h.close()
            pin_in.free()
            pin_out.free()
        if not args.no_cpu and world == 1:  # the CPU baseline is timed at N = 1 only
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orc  # CPU oracle: test infrastructure, used here only as the timed baseline
            sample = data[: min(n_in, args.cpu_sample_mib << 20)]
            tc = time.perf_counter()
            o1, _ = orc.compress(sample, level, threads=1)
            t1 = time.perf_counter() - tc
            cores = min(os.cpu_count() or 1, 16)
            tc = time.perf_counter()
            o2, _ = orc.compress(sample, level, threads=cores)
            tm = time.perf_counter() - tc
            # the same sample through the GPU path must give the oracle's bytes (checker, not timed)
            ln = ctx.compress_device(d_in.data_ptr(), int(sample.size), level, d_out.data_ptr(), cap)
            same = d_out[:ln].cpu().numpy().tobytes() == bytes(o1) and bytes(o1) == bytes(o2)
            res["cpu_baseline"] = {"value": round(sample.size / t1 / 1e6, 2), "unit": "MB/s",
                                   "cores": 1, "kind": "port",
                                   "sample": f"first {sample.size >> 20} MiB of the same corpus, "
                                             f"C oracle -O2, 1 thread",
                                   "all_cores": {"value": round(sample.size / tm / 1e6, 2),
                                                 "cores": cores},
                                   "gpu_output_equals_oracle": same}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    ctx.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
