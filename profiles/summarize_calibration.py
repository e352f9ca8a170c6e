#!/usr/bin/env python3
"""gpurun_out/calib (profiles/calibrate.sh) -> profiles/<tag>_pmc_calibration.json: for every known-byte launch of
tools/calib_pmc, the FETCH_SIZE / WRITE_SIZE counter values (KiB as rocprofv3 reports them, summed over XCDs) and the
factor known bytes / counter bytes for that access shape."""
import collections
import csv
import json
import sys
from pathlib import Path

tag = sys.argv[1]
src = Path("gpurun_out/calib")
known = {}
for ln in (src / "plain.jsonl").read_text().splitlines():
    r = json.loads(ln)
    known[r["kernel"]] = r
counters = collections.defaultdict(dict)
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for r in csv.DictReader(open(src / name / "run_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        counters[k][counter] = counters[k].get(counter, 0.0) + float(r["Counter_Value"]) * 1024
rows = []
for k, r in known.items():
    c = counters.get(k, {})
    row = dict(r)
    row["FETCH_SIZE_bytes"] = c.get("FETCH_SIZE")
    row["WRITE_SIZE_bytes"] = c.get("WRITE_SIZE")
    row["fetch_factor"] = r["known_read_bytes"] / c["FETCH_SIZE"] if c.get("FETCH_SIZE") and r["known_read_bytes"] else None
    row["write_factor"] = r["known_write_bytes"] / c["WRITE_SIZE"] if c.get("WRITE_SIZE") and r["known_write_bytes"] else None
    rows.append(row)
    print(f"{k:26s} read {r['known_read_bytes'] / 1e9:7.3f} GB  FETCH_SIZE {(c.get('FETCH_SIZE') or 0) / 1e9:7.3f} GB  factor {row['fetch_factor']}"
          f" | write {r['known_write_bytes'] / 1e9:7.3f} GB  WRITE_SIZE {(c.get('WRITE_SIZE') or 0) / 1e9:7.3f} GB  factor {row['write_factor']}  | {r['ms']:.2f} ms")
json.dump(rows, open(Path("profiles") / f"{tag}_pmc_calibration.json", "w"), indent=1)
