#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection csv rows per kernel whose name contains a substring (kernel names contain commas, so the
filter is ONE substring, not a list) and print the ratios that say what a kernel waits on.
usage: pmc_any.py SUBSTR out.json a_counter_collection.csv [b_counter_collection.csv ...]"""
import csv, json, sys
from collections import defaultdict
sub, out, files = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if sub not in name:
            continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[name].add((f, row["Dispatch_Id"]))
res = {}
for name, v in acc.items():
    d = dict(v)
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    pct = lambda k, base: round(100.0 * d.get(k, 0.0) / base, 2) if base else None
    d["summary"] = {"wait_any_pct": pct("SQ_WAIT_ANY", wc), "wait_inst_any_pct": pct("SQ_WAIT_INST_ANY", wc),
                    "active_inst_any_pct": pct("SQ_ACTIVE_INST_ANY", wc), "valu_active_pct": pct("SQ_ACTIVE_INST_VALU", wc),
                    "mfma_busy_pct": pct("SQ_VALU_MFMA_BUSY_CYCLES", d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0),
                    "lds_conflict_pct_of_lds_active": pct("SQ_LDS_BANK_CONFLICT", d.get("SQ_LDS_IDX_ACTIVE", 0.0)),
                    "lds_idx_active_pct_of_busy": pct("SQ_LDS_IDX_ACTIVE", d.get("SQ_BUSY_CYCLES", 0.0)),
                    "valu_per_mfma": round(d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"], 2) if d.get("SQ_INSTS_MFMA") else None}
    res[name[:120]] = d
    print(name[:100])
    print("  ", json.dumps(d["summary"]))
    print("  ", {k: "%.3g" % x for k, x in d.items() if k != "summary"})
json.dump(res, open(out, "w"), indent=1)
