#!/bin/bash
# SQ pass for the LDS-heavy kernels: LDS instruction counts, bank conflicts, waits (run through gpurun from the repo root)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_sq3
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-lines -1 --no-api-timing > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
ls $O
