#!/bin/bash
# One-GPU rehearsal of what the driver runs at round end: smoke(), then bench.py's N>1 code paths with two ranks sharing
# cuda:0 (SAGE_BENCH_REHEARSE=1: gloo, collectives staged through the host; the numbers mean nothing) -- weak scaling
# (every rank its own shard) and strong scaling (ONE corpus cut over the ranks by the product's sharding).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 || exit 1
for mode in weak strong; do
  SAGE_BENCH_REHEARSE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-lines -1 --lines 300000 --scaling $mode > gpurun_out/bench_2rank_$mode.json 2> gpurun_out/bench_2rank_$mode.err || { tail -5 gpurun_out/bench_2rank_$mode.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_2rank_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["scaling"], d["n_gpus"], d["config"]["total_tokens"], d["config"]["total_triples"], round(d["ms_per_step"], 1), d["config"]["lines_per_gpu"])
PY
done
