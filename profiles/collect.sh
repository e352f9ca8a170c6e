#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py workload on the GPU box (run through gpurun from the repo root):
#     profiles/collect.sh <out-name> [bench.py workload arguments ...]
#   kernel-trace stats of the bench, then FETCH_SIZE and WRITE_SIZE in passes of their own (never together with a trace),
#   then the full bench line (with cpu_baseline unless --cpu-lines -1 is among the arguments).
# Outputs land in gpurun_out/prof_<out-name>/ and are summarised by profiles/summarize.py into profiles/.
set -o pipefail
export TMPDIR=/tmp
NAME=$1; shift
O=gpurun_out/prof_$NAME
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-lines -1 --no-api-timing "$@" > $O/bench_kt.json 2> $O/bench_kt.err || exit 1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-lines -1 --no-api-timing "$@" > $O/bench_fetch.json 2> $O/bench_fetch.err || exit 1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-lines -1 --no-api-timing "$@" > $O/bench_write.json 2> $O/bench_write.err || exit 1
timeout -k 10 900 python3 -u bench.py "$@" > $O/bench_full.json 2> $O/bench_full.err || exit 1
tail -c 300 $O/bench_full.json
