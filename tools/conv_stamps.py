#!/usr/bin/env python3
"""Diagnostic (library built with -DSFM_CONVP_STAMPS: tools/convp_variants.sh stamps -DSFM_CONVP_STAMPS): where a conv16p
workgroup spends its cycles - staging the patch, the weight-tile loop, the epilogue(s)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sincformer_metacog_speech_enhancement_amd import ops, lib
ops.set_compute_dtype("f16")
dt = torch.float16
B = 64
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
L_ = lib.load()
rd = ctypes.CDLL(lib.LIB_PATH).sfm_conv16p_read_stamps
for name, cin, cout, k, s, p, two, skip, L in [("b0.c1+cs", 64, 128, 7, 2, 3, False, True, 64000), ("b0.c2", 128, 128, 3, 1, 1, False, False, 32000),
                                               ("b1.c1+cs", 128, 128, 7, 2, 3, True, True, 32000), ("b2.c2", 256, 256, 3, 1, 1, False, False, 8000)]:
    x1, x2 = R(B, L, cin).to(dt), (R(B, L, cin).to(dt) if two else None)
    sc1, sh1 = R(B, cin) * 0.1 + 1, R(B, cin) * 0.1
    sc2, sh2 = (R(B, cin) * 0.1 + 1, R(B, cin) * 0.1) if two else (None, None)
    pw = ops.pack_linear(R(cout, cin, k) / (cin * k) ** 0.5, R(cout))
    spw = ops.pack_linear(R(cout, cin, 1) / cin ** 0.5, R(cout)) if skip else None
    Lout = (L + 2 * p - k) // s + 1
    P = 2 * ((Lout + 127) // 128)
    out = torch.empty(B, Lout, cout, device="cuda", dtype=dt)
    part = torch.zeros(B, P, 16, 2, device="cuda")
    outs = torch.empty(B, Lout, cout, device="cuda", dtype=dt) if skip else None
    parts = torch.zeros(B, P, 16, 2, device="cuda") if skip else None
    for _ in range(3):
        ops.conv16p(x1, sc1, sh1, pw, out, B=B, Lin=L, stride=s, pad=p, x2=x2, sc2=sc2, sh2=sh2, gn_partial=part, gn_group=cout // 16,
                    skip_pw=spw, out_s=outs, gn_partial_s=parts)
    torch.cuda.synchronize()
    n = min(32768, B * ((Lout + 127) // 128))
    buf = np.zeros((n, 8), dtype=np.uint64)
    assert rd(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
    st = buf.astype(np.float64)
    d = np.diff(st[:, :4], axis=1)
    tot = st[:, 3] - st[:, 0]
    # co-residency: workgroups by CU (XCC_ID, SE/SH/CU of HW_ID); for every workgroup, what its CU neighbour was doing while it
    # was in its k-loop (stamps 1..2): fraction of that interval the OTHER workgroup on the CU spent in ITS k-loop
    hw = buf[:, 6]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)           # xcc | se, sh, cu bits of HW_ID
    tg = (hw >> 16) & 0xf
    order = np.argsort(cu, kind="stable")
    ov = []
    by = {}
    for i in order:
        by.setdefault(int(cu[i]), []).append(i)
    for k, ids in by.items():
        ids = sorted(ids, key=lambda i: st[i, 0])
        for a_ in range(len(ids)):
            i = ids[a_]
            lo, hi = st[i, 1], st[i, 2]
            o = 0.0
            for b_ in range(max(0, a_ - 3), min(len(ids), a_ + 4)):
                if b_ == a_:
                    continue
                j = ids[b_]
                o += max(0.0, min(hi, st[j, 2]) - max(lo, st[j, 1]))
            ov.append(o / max(hi - lo, 1.0))
    # turnaround of a thread-group slot: end of one workgroup -> start of the next one the dispatcher puts there
    slot = cu * 16 + tg
    gaps, busy = [], []
    for k in set(int(x) for x in slot):
        ids = [i for i in np.nonzero(slot == k)[0]]
        ids.sort(key=lambda i: st[i, 0])
        for a_, b_ in zip(ids[:-1], ids[1:]):
            gaps.append(st[b_, 0] - st[a_, 3])
        if len(ids) > 1:
            busy.append(sum(st[i, 3] - st[i, 0] for i in ids) / max(st[ids[-1], 3] - st[ids[0], 0], 1.0))
    coloc = {"slot_turnaround_cycles_p50": float(np.median(gaps)) if gaps else None, "slot_turnaround_cycles_p90": float(np.percentile(gaps, 90)) if gaps else None,
             "slot_busy_fraction_p50": float(np.median(busy)) if busy else None,"cus": len(by), "workgroups_per_cu_p50": float(np.median([len(v) for v in by.values()])), "tg_ids": sorted(set(int(x) for x in tg))[:8],
             "kloop_overlap_with_neighbour_kloop_p50": float(np.median(ov)), "mean": float(np.mean(ov))}
    fine = {"dma_issue+scale_loads": float(np.median(st[:, 4] - st[:, 0])), "raw_rows_wait": float(np.median(st[:, 5] - st[:, 4])),
            "transform": float(np.median(st[:, 1] - st[:, 5]))}
    print(json.dumps({"layer": name, "cycles_p50": {"stage_first_slab": float(np.median(d[:, 0])), "rest_slabs+k_loop": float(np.median(d[:, 1])),
                                                   "epilogue": float(np.median(d[:, 2])), "total": float(np.median(tot))}, "stage_first_slab": fine, "co_residency": coloc}))
