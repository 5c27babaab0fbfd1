#!/bin/bash
# bench.py at N = 1, then a 2-rank rehearsal of --gpus 2 on one GPU (gloo, both ranks on cuda:0)
set -e
mkdir -p gpurun_out/r2
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r2/bench1.json 2> gpurun_out/r2/bench1.err
tail -c 3000 gpurun_out/r2/bench1.json
BZ2_BENCH_BACKEND=gloo BZ2_BENCH_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --segments 64 > gpurun_out/r2/bench2.json 2> gpurun_out/r2/bench2.err
tail -c 1500 gpurun_out/r2/bench2.json
