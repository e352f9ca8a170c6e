#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 || exit 1
SAGE_BENCH_REHEARSE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-lines -1 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err || { tail -5 gpurun_out/bench_2rank.err; exit 1; }
tail -c 600 gpurun_out/bench_2rank.json
