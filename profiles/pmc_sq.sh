#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_sq
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-lines -1 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
ls $O
