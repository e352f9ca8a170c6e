#!/bin/bash
# second SQ pass: where the issue cycles go (run through gpurun from the repo root)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_sq2
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -o "SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_INST_CYCLES_[A-Z_]*\|SQ_THREAD_CYCLES_VALU\|SQ_VALU_MFMA_BUSY_CYCLES" $O/counters.txt | sort -u | tr '\n' ' ' > $O/names.txt
cat $O/names.txt
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-lines -1 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
ls $O
