#!/bin/bash
# GPU check used during development: parity tests, then the 1M-line bench without the CPU baseline leg.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -8 || exit 1
timeout -k 10 600 python -u bench.py --cpu-lines -1 "$@" > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err || { tail -5 gpurun_out/bench_1m.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_1m.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["total_triples"])
print({k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
print(d["roofline"])
PY
