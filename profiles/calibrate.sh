#!/bin/bash
# PMC calibration on the GPU box (through gpurun, from the repo root): tools/calib_pmc's known-byte launches under
# FETCH_SIZE and WRITE_SIZE passes of their own.  profiles/summarize_calibration.py turns the result into
# profiles/<tag>_pmc_calibration.json.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/calib
rm -rf $O && mkdir -p $O
timeout -k 10 120 tools/calib_pmc > $O/plain.jsonl 2> $O/plain.err || exit 1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- tools/calib_pmc > $O/fetch.jsonl 2> $O/fetch.err || exit 1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- tools/calib_pmc > $O/write.jsonl 2> $O/write.err || exit 1
cat $O/plain.jsonl
