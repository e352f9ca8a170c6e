#!/usr/bin/env python3
"""Summarise gpurun_out/prof (written by profiles/collect.sh) into profiles/<tag>_*: the rocprofv3 kernel stats
CSV as is, the bench JSON lines, and a per-kernel table of FETCH_SIZE / WRITE_SIZE (KiB as rocprofv3 reports them,
summed over XCDs, per launch).  FETCH_SIZE on gfx950 counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section);
profiles/r02_pmc_calibration.json (tools/calib_pmc) measured the same factor 2.00 for this library's read shapes --
16-B-per-lane streams AND the 8-lane x 16-B row gathers of k_dots / k_pair_cache -- so `fetch_x2` is the read figure
for every kernel here.  WRITE_SIZE is exact for 16-B-per-lane streams, 64-byte chunks and runs of 8-byte records, and
4x the bytes for isolated 8-byte stores (32-byte granules); `calibrated` = fetch_x2 + write, the figure bench.py
quotes as `traffic`."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

# usage: summarize.py <tag> [<collect.sh out-name>]   e.g.  summarize.py r03_en16_1m_d50_v256k en16  (bench.py looks the
# PMC file up as profiles/r*_<workload_tag>_pmc_traffic.json; the tag of the default workload is en16_1m_d50_v256k)
tag = sys.argv[1]
src = Path("gpurun_out/prof" + (f"_{sys.argv[2]}" if len(sys.argv) > 2 else ""))
dst = Path("profiles")
shutil.copy(src / "kt/run_kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")
shutil.copy(src / "bench_full.json", dst / f"{tag}_bench_full.json")
shutil.copy(src / "bench_kt.json", dst / f"{tag}_bench_under_rocprof.json")
agg = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    seen = collections.Counter()
    for r in csv.DictReader(open(src / name / "run_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][counter] += float(r["Counter_Value"])
        seen[k] += 1
    for k, n in seen.items():
        agg[k]["launches"] = n
rows = []
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["FETCH_SIZE"]):
    n = max(1, a["launches"])
    rows.append({"kernel": k, "launches": a["launches"], "fetch_GB_per_launch": a["FETCH_SIZE"] * 1024 / n / 1e9,
                 "fetch_x2_GB_per_launch": 2 * a["FETCH_SIZE"] * 1024 / n / 1e9,
                 "write_GB_per_launch": a["WRITE_SIZE"] * 1024 / n / 1e9,
                 "calibrated_GB_per_launch": (2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024 / n / 1e9})
# Whole-step HBM-side bytes: per-launch bytes x launches per timed step.  The kernel-trace run holds 4 steps (1 warm-up
# + 3 timed) plus the set-up tokenization, the PMC runs hold 1 step plus the same set-up, so the difference / 3 is the
# number of launches one step makes.
kt_calls = {}
for r in csv.DictReader(open(src / "kt/run_kernel_stats.csv")):
    kt_calls[r["Name"].split("(")[0].replace("void ", "")] = int(r["Calls"])
step_fetch = step_write = 0.0
for r in rows:
    per_step = max(0.0, (kt_calls.get(r["kernel"], 0) - r["launches"]) / 3.0)
    r["launches_per_step"] = per_step
    step_fetch += r["fetch_x2_GB_per_launch"] * per_step
    step_write += r["write_GB_per_launch"] * per_step
rows.append({"kernel": "__step_total__", "launches": 0, "fetch_GB_per_launch": step_fetch / 2, "fetch_x2_GB_per_launch": step_fetch,
             "write_GB_per_launch": step_write, "calibrated_GB_per_launch": step_fetch + step_write, "launches_per_step": 1.0})
json.dump(rows, open(dst / f"{tag}_pmc_traffic.json", "w"), indent=1)
print(f"step total: fetch x2 {step_fetch:.1f} GB + write {step_write:.1f} GB")
for r in rows[:12]:
    print(f"{r['kernel']:28s} n={r['launches']:3d} fetch={r['fetch_GB_per_launch']:8.2f} GB (x2 {r['fetch_x2_GB_per_launch']:8.2f})"
          f" write={r['write_GB_per_launch']:7.2f} GB")
