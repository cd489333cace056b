#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (separate runs, counter_collection.csv each) per kernel name:
   python tools/pmc_kernels.py out.json <substring>[;<substring>...] <counter_collection.csv> [...]
Average of every collected counter per dispatch of the kernels whose name contains one of the substrings, plus the
derived MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), MI355X_MICROARCH.md)."""
import collections, csv, json, sys

subs = sys.argv[2].split(";")          # kernel names contain commas: the separator is ";"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        hit = [s for s in subs if s in k]
        if not hit:
            continue
        a = acc[hit[0]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {}
for k, cs in acc.items():
    d = {c: v[1] / v[0] for c, v in cs.items()}
    d["dispatches"] = max(v[0] for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        d["mfma_util_pct"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        d["lds_conflict_pct"] = 100.0 * d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
        d["wait_pct"] = 100.0 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
    out[k] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
