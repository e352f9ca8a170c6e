#!/bin/bash
# SQ pass over the matcher variants (tools/matchbench.py): instruction counts and issue/wait cycles
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_match
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O -o run --output-format csv -- python3 tools/matchbench.py "$@" > $O/bench.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import csv, collections, glob
for f in glob.glob('gpurun_out/pmc_match/*counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k] += 1
    for k, v in agg.items():
        if 'match' not in k: continue
        L = n[k] / len(v)
        print(k, 'launches', L, {c: f'{x / L:.4g}' for c, x in v.items()})
PY
