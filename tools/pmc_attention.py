#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over tools/attn_bench.py into profiles/*/pmc_attention.json:
   python tools/pmc_attention.py out.json <counter_collection.csv> [...]
Per attention kernel: average of every collected counter per dispatch, plus the derived MFMA utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs), MI355X_MICROARCH.md) when both are present."""
import collections, csv, json, sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "attn_fwd" not in k:
            continue
        name = "attn_fwd_hd64p4" if "hd64p4" in k else "attn_fwd_hd64p8" if "hd64p8" in k else "attn_fwd_hd64r" if "hd64r" in k else ("attn_fwd_hd64x2" if "x2" in k else "attn_fwd_hd64")
        a = acc[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {}
for k, cs in acc.items():
    d = {c: v[1] / v[0] for c, v in cs.items()}
    d["dispatches"] = max(v[0] for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        d["mfma_util_pct"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)   # 8 XCDs summed; 256 CU x 4 SIMD
    out[k] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
