#!/bin/bash
# vL1D / TA / L2 pass over every kernel of the step (tools/stepbench.py, 300k lines)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_step_tcp
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O -o run --output-format csv -- python3 tools/stepbench.py > $O/bench.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import csv, collections, glob
for f in glob.glob('gpurun_out/pmc_step_tcp/*counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k] += 1
    print(f"{'kernel':40s} {'launches':>8s} {'cyc/launch':>10s} {'TA_busy':>8s} {'TA_stallTC':>10s} {'TCP_pend':>9s} {'acc/cyc/CU':>10s} {'L2req/acc':>9s} {'L2miss%':>8s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE']):
        L = n[k] / len(v)
        cyc = v['GRBM_GUI_ACTIVE'] / 8           # per XCD
        if cyc / L < 2e4: continue
        tcc = v['TCC_HIT_sum'] + v['TCC_MISS_sum']
        print(f"{k:40s} {L:8.0f} {cyc / L:10.3g} {v['TA_BUSY_avr'] / cyc:8.2f} {v['TA_ADDR_STALLED_BY_TC_CYCLES_sum'] / 256 / cyc:10.2f} {v['TCP_PENDING_STALL_CYCLES_sum'] / 256 / cyc:9.2f} {v['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256 / cyc:10.3f} {v['TCP_TCC_READ_REQ_sum'] / max(1, v['TCP_TOTAL_CACHE_ACCESSES_sum']):9.2f} {100 * v['TCC_MISS_sum'] / max(1, tcc):8.1f}")
PY
