#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide
prescribes) into per-kernel-family HBM bytes per launch.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream, i.e. reports half of the bytes -> doubled here (`fetch_corrected`); WRITE_SIZE is
exact for 16-B-per-lane streaming stores.  Both counters are in KiB."""
import collections, csv, json, re, sys

FAMILY = [("conv16p_kernel", "conv16p"), ("gemm16_tn", "gemm16_tn"), ("attn_bwd", "attention_bwd"), ("layernorm_bwd", "layernorm_bwd"), ("ew_train", "ew_train"),
          ("framed_gemm_split16", "framed_gemm_split16"), ("gemm16v2_kernel", "gemm16"), ("gemm16w_kernel", "gemm16"), ("gemm16p_kernel", "gemm16"), ("gemm16_kernel", "gemm16"), ("attn_fwd", "attention_fwd"),
          ("framed_gemm", "framed_gemm_f32"), ("sinc_fir16_kernel", "sinc_fir16"), ("ffn_fused", "ffn_fused"), ("gn_apply", "gn_apply"),
          ("layernorm", "layernorm"), ("bilstm", "bilstm_layer"), ("dwconv", "dwconv_bn_swish"),
          ("pool_time", "pool_time"), ("polar_mask", "polar_mask")]


def fam(name):
    for key, f in FAMILY:
        if key in name:
            return f
    return None


def agg(path, cname):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        f = fam(r["Kernel_Name"])
        if f:
            a[f][0] += 1
            a[f][1] += float(r["Counter_Value"])
    return a


def main():
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        n = f[k][0]
        fr = f[k][1] / n * 1024.0
        wr = w[k][1] / max(w[k][0], 1) * 1024.0
        out[k] = {"launches": n, "fetch_raw_bytes": fr, "fetch_corrected_bytes": 2.0 * fr, "write_bytes": wr,
                  "hbm_bytes_per_launch": 2.0 * fr + wr}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print("%-18s n=%4d  %8.1f MB/launch" % (k, v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
