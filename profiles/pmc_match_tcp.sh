#!/bin/bash
# vL1D / L2 pass over the matcher variants (tools/matchbench.py)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_match_tcp
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O -o run --output-format csv -- python3 tools/matchbench.py "$@" > $O/bench.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import csv, collections, glob
for f in glob.glob('gpurun_out/pmc_match_tcp/*counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k] += 1
    for k, v in agg.items():
        if 'match' not in k: continue
        L = n[k] / len(v)
        print(k, 'launches', L, {c: f'{x / L:.4g}' for c, x in v.items()})
PY
