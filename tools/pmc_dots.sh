#!/bin/bash
# SQ counters of the step's kernels on the 300k-line workload (run through gpurun from the repo root; PMC="..." picks the
# counters, at most 8 SQ_ ones; KERNELS=regex filters the kernels printed)
set -o pipefail
export TMPDIR=/tmp SB_WARM_MATCHES=1
O=gpurun_out/pmc_dots
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS} -d $O -o run --output-format csv -- python3 tools/stepbench.py > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_dots/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:40]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    import os, re
    if re.search(os.environ.get('KERNELS', 'dots|logsig'), k):
        print(k, {c: f"{x:.3g}" for c, x in v.items()})
PY
