#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: see the round prompt / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3p|c1|c5|c3se|c2t] [--dtype bf16|f16] [--streams S]

A step = one forward pass of the north-star path (PerceptionAgent -> pool -> CPEA -> STFT ->
MaskSynthesisAgent -> apply_mask -> iSTFT) over one batch of synthetic 16 kHz utterances that is
already resident in HBM.  Default workload = BASELINE.json configs[1]: batch 64 x 4 s (L = 64 000,
T = 801 STFT frames/utterance), forward only, bf16 MFMA operands with fp32 accumulation.
value = STFT frames/s over all ranks (utterances shard over ranks: no data-path collective).
Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (batch per GPU, samples, description)
    "c1": (1, 16000, "B1 x 1 s (L16000, T201) forward"),
    "c2": (64, 64000, "B64 x 4 s (L64000, T801) SincNet+Conformer forward (BASELINE configs[1])"),
    "c3p": (256, 40880, "B256 x 512-frame utterances (L40880, T512) forward"),
    "c3se": (256, 64000, "B256 x 4 s (L64000, T801) SpeechEnhancer training step: STFT, forward, SI-SNR + L1 + multi-res "
                         "STFT objective, backward, gradient all-reduce, clip, AdamW (BASELINE configs[2]/[3])"),
    "c2t": (64, 64000, "B64 x 4 s (L64000, T801) SincNet+Conformer path (PerceptionAgent, CPEA, MaskSynthesisAgent) training "
                       "step: forward, SI-SNR + L1 + multi-res STFT objective, backward through every module, all-reduce, "
                       "clip, AdamW"),
    "c5": (32, 480000, "B32 x 30 s (L480000, T6001) forward with episodic memory (BASELINE configs[4], fwd)"),
}
PEAKS = {"mfma16": 2500.0, "mfma32": 157.3, "hbm": 8000.0}      # TFLOP/s, TFLOP/s, GB/s (MI355X_MICROARCH.md)
FAMILY_BOUND = {"gemm16": "mfma16", "attention_fwd": "mfma16", "framed_gemm_f32": "mfma32", "gemm16_tn": "mfma16",
                "attention_bwd": "mfma16"}


def build_path(dtype, seed=1234, use_memory=False):
    import torch
    from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    ops.set_compute_dtype(dtype)
    path = EnhancementPath(sample_rate=16000, use_memory=use_memory)
    sd = path.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    keep = {k: v.numpy() for k, v in sd.items() if k.split(".")[-1] in ("low_hz_", "band_hz_", "window", "n_")}
    new = syn.synth_state_dict(shapes, seed, keep=keep, sinc_scale=2000.0)
    path.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
    return path, new


def host_cores():
    """threads this process may really use: affinity mask, cgroup cpu quota, capped at the
    16-core share a 1-GPU box gets (oversubscribing a quota makes the CPU leg crawl)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(weights, L, batch=None, iters=4):
    """Oracle (CPU restatement, kind 'port') timed on the host cores on a bounded sample (about 10 s of CPU work)."""
    batch = batch or max(1, min(8, 512000 // L))
    import torch
    from oracle import sfm_oracle as orc
    from sincformer_metacog_speech_enhancement_amd import synthetic as syn
    cores = host_cores()
    torch.set_num_threads(cores)

    def subd(prefix):
        p = prefix + "."
        return {k[len(p):]: torch.from_numpy(v) for k, v in weights.items() if k.startswith(p)}
    sds = {"pa": subd("perception"), "cpea": subd("cpea"), "msa": subd("msa")}
    noisy, _ = syn.synth_wave(batch, L, 1234)
    orc.enhance_path(sds, noisy, 16000)           # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        orc.enhance_path(sds, noisy, 16000)
    dt = (time.perf_counter() - t0) / iters
    T = 1 + L // 80
    return {"value": batch * T / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "oracle enhance_path, batch %d x L%d, %d iterations after 1 warm-up, fp32, torch %d threads"
                      % (batch, L, iters, cores)}


def cpu_baseline_train(sd, L, iters=8, batch=8):
    """oracle forward + backward (torch autograd on the host cores) of the same training step, small batch."""
    import torch
    from oracle import sfm_oracle as orc
    from sincformer_metacog_speech_enhancement_amd import synthetic as syn
    cores = host_cores()
    torch.set_num_threads(cores)
    noisy, clean = syn.synth_wave(batch, L, 77)
    noisy, clean = torch.from_numpy(noisy), torch.from_numpy(clean)
    ref = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
           for k, v in sd.items()}
    t0 = time.perf_counter()
    for i in range(iters):
        print("[bench] cpu baseline (training) iter %d/%d" % (i + 1, iters), file=sys.stderr, flush=True)
        total, _, _ = orc.enhancer_loss(ref, noisy, clean, 4, bn_train=True)
        total.backward()
    dt = (time.perf_counter() - t0) / iters
    T = 1 + L // 80
    return {"value": batch * T / dt, "unit": "STFT frames/s", "cores": cores, "kind": "port",
            "sample": "oracle (torch fp32 autograd restatement) forward+backward of the same step, batch %d x %d samples, %d "
                      "iterations, %.1f s each; no optimiser step" % (batch, L, iters, dt)}


def cpu_baseline_path_train(weights, L, iters=3, batch=4):
    """oracle forward + backward of the whole-path training step (torch autograd on the host cores), small batch."""
    import torch
    from oracle import sfm_oracle as orc
    from sincformer_metacog_speech_enhancement_amd import synthetic as syn
    cores = host_cores()
    torch.set_num_threads(cores)
    noisy, clean = syn.synth_wave(batch, L, 77)
    noisy, clean = torch.from_numpy(noisy), torch.from_numpy(clean)

    def subd(prefix):
        p = prefix + "."
        out = {}
        for k, v in weights.items():
            if k.startswith(p):
                v = torch.from_numpy(v) if not torch.is_tensor(v) else v
                leaf = v.dtype.is_floating_point and "running" not in k and k.split(".")[-1] not in ("window", "n_")
                out[k[len(p):]] = v.clone().requires_grad_(True) if leaf else v.clone()
        return out
    sds = {"pa": subd("perception"), "cpea": subd("cpea"), "msa": subd("msa")}
    t0 = time.perf_counter()
    for i in range(iters):
        print("[bench] cpu baseline (path training) iter %d/%d" % (i + 1, iters), file=sys.stderr, flush=True)
        total, _, _ = orc.path_loss(sds, noisy, clean, 16000, bn_train=True)
        total.backward()
    dt = (time.perf_counter() - t0) / iters
    T = 1 + L // 80
    return {"value": batch * T / dt, "unit": "STFT frames/s", "cores": cores, "kind": "port",
            "sample": "oracle (torch fp32 autograd restatement) forward+backward of the same step, batch %d x %d samples, %d "
                      "iterations, %.1f s each; no optimiser step" % (batch, L, iters, dt)}


def main_train(args):
    """--workload c3se / c2t: one training step of training/conformer_pipeline.py per bench step (c3se: the reference's
    SpeechEnhancer; c2t: the north-star SincNet + Conformer composition)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0 if os.environ.get("SFM_SINGLE_DEVICE") else local_rank)   # rehearsal of N ranks on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SFM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
    from sincformer_metacog_speech_enhancement_amd.optim import FlatAdamW
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import SpeechEnhancer, batch_stft, compute_loss
    ops.set_compute_dtype(args.dtype)
    B, L, desc = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    T = 1 + L // 80
    whole_path = args.workload == "c2t"
    if whole_path:
        from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import compute_path_loss
        model, sd = build_path(args.dtype, seed=4321)
    else:
        model = SpeechEnhancer(n_freq=129, d_model=256, num_blocks=4, num_heads=4, d_ff=1024, kernel_size=31, dropout=0.15)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, 4321).items()}
        model.load_state_dict(sd)                                   # same weights on every rank
    model.cuda().train()
    opt = FlatAdamW(model.parameters(), lr=5e-4, betas=(0.9, 0.98), weight_decay=0.01, max_norm=5.0)
    noisy, clean = syn.synth_wave(B, L, 1234 + rank)                # each rank trains on its own utterance shard
    noisy, clean = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()
    torch.manual_seed(1000 + rank)                                  # dropout seeds

    def step():
        opt.zero_grad()
        if whole_path:
            total, _ = compute_path_loss(model, noisy, clean)
            total.backward()
            opt.step(loss=total)
            return total
        nr, ni = batch_stft(noisy, 256, 80, 160)
        cr, ci = batch_stft(clean, 256, 80, 160)
        total, _ = compute_loss(model, nr, ni, clean, cr, ci)
        total.backward()
        opt.step(loss=total)
        return total

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        print("[bench] %s, dtype %s, world %d, batch/GPU %d" % (desc, args.dtype, world, B), file=sys.stderr, flush=True)
    for i in range(max(args.warmup, 1)):
        if i == max(args.warmup, 1) - 1:
            ops.profiler.enable(None)
        loss = step()
    breakdown = ops.profiler.summary()
    ops.profiler.disable()
    dominant = max(breakdown, key=lambda k: breakdown[k]["ms_total"])
    ops.profiler.enable({dominant})
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    dom = ops.profiler.summary()[dominant]
    ops.profiler.disable()
    tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    frames = world * B * T * args.steps
    if rank == 0:
        kind = FAMILY_BOUND.get(dominant, "hbm")
        secs = dom["ms_avg"] * 1e-3
        if kind == "hbm":
            ach, peak, unit, bound = dom["bytes"] / dom["n"] / secs / 1e9, PEAKS["hbm"], "GB/s", "hbm"
        else:
            ach, peak, unit, bound = dom["flops"] / dom["n"] / secs / 1e12, PEAKS[kind], "TFLOP/s", "mfma"
        st = opt.stats()
        traffic = None                                  # HBM bytes per launch of the dominant family from the committed PMC passes
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.workload)
        if os.path.exists(pmc_path):
            traffic = json.load(open(pmc_path)).get(dominant, {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "audio frames/sec/GPU (16 kHz, 512-frame utts) + mask RMSE vs CPU ref",
            "value": frames / elapsed, "unit": "STFT frames/s trained (whole job)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": desc, "batch_per_gpu": B, "samples": L, "frames_per_utt": T,
                       "sharding": "utterances over ranks; one bucketed all-reduce (RCCL) of the flat fp32 gradient per step, "
                                   "overlapped with backward", "optimizer": "AdamW lr 5e-4 betas (0.9, 0.98) wd 0.01, clip 5.0",
                       "dropout": "module defaults (0.1 / 0.15)" if whole_path else 0.15},
            "roofline": {"bound": bound, "kernel": dominant, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                         "traffic": traffic, "traffic_unit": "HBM bytes/launch (PMC)",
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["n"],
                         "algorithmic_flops_per_launch": dom["flops"] / dom["n"], "launches": dom["n"], "avg_ms": dom["ms_avg"]},
            "frames_per_s_per_gpu": frames / elapsed / world,
            "final_loss": float(loss.detach()), "optimizer_state": st,
            "breakdown_ms_per_step": {k: round(v["ms_total"], 4) for k, v in
                                      sorted(breakdown.items(), key=lambda kv: -kv[1]["ms_total"])},
        }
        print("[bench] gpu leg done: %.1f ms/step, %.3e frames/s; dominant kernel %s" %
              (line["ms_per_step"], line["value"], dominant), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_path_train(sd, L) if whole_path else cpu_baseline_train(sd, L)
        if args.breakdown:
            with open(args.breakdown, "w") as fh:
                json.dump(breakdown, fh, indent=1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", default=None, help="write the per-kernel-family breakdown JSON here")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch of the workload")
    ap.add_argument("--graph", action="store_true", help="replay the forward as one hipGraph (launch-bound small batches)")
    ap.add_argument("--streams", type=int, default=0, help="forward passes in flight: consecutive steps alternate over this many "
                    "HIP streams (each step is still one whole pass over its own buffers; 1 = strictly one pass at a time; "
                    "0 = auto: a short calibration during warm-up picks the fastest of 1, 2 and 3 on this machine)")
    args = ap.parse_args()
    if args.workload in ("c3se", "c2t"):
        return main_train(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0 if os.environ.get("SFM_SINGLE_DEVICE") else local_rank)   # rehearsal of N ranks on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SFM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
    B, L, desc = WORKLOADS[args.workload]
    T = 1 + L // 80
    path, weights = build_path(args.dtype, use_memory=(args.workload == "c5"))
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(B, L, 1234 + rank)          # each rank enhances its own utterance shard
    wave = torch.from_numpy(noisy).cuda()

    if args.graph:
        from sincformer_metacog_speech_enhancement_amd.graph import GraphedForward
        # one captured graph (with its own static buffers) per pass in flight: a graph is never replayed concurrently with itself
        graphs = [GraphedForward(lambda w: path(w)) for _ in range(max(args.streams, 1))]
        for g_ in graphs:
            g_(wave)                                      # capture outside the timed region
        graphed = graphs[0]

    auto_streams = args.streams == 0
    if auto_streams:
        args.streams = 2
    # the current (default) stream + S-1 new ones: as few HIP streams as possible, so that they never have to share one of
    # the process's hardware queues (5 streams on 4 queues ran SLOWER than a single stream: 12.9 vs 11.0 ms)
    streams = ([torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(args.streams - 1)]) if args.streams > 1 else None
    counter = [0]

    def step():
        if streams is not None:
            st = streams[counter[0] % len(streams)]
            counter[0] += 1
            with torch.cuda.stream(st):
                return graphs[(counter[0] - 1) % len(streams)](wave) if args.graph else path(wave)
        return graphed(wave) if args.graph else path(wave)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        print("[bench] %s, dtype %s, world %d" % (desc, args.dtype, world), file=sys.stderr, flush=True)
    with torch.no_grad():
        if auto_streams and not args.graph:
            # calibration (untimed, part of the warm-up): 6 passes each with 1, 2 and 3 passes in flight; the fastest wins
            # (more streams than the process has hardware queues run SLOWER than one: never assume, measure)
            def timed(n):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    step()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / n
            pool = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
            res = {}
            for cand in (3, 2, 1):
                streams = pool[:cand] if cand > 1 else None
                for _ in range(cand):
                    step()
                res[cand] = timed(6)
            best = min(res, key=lambda c: res[c] * (1.0 + 0.01 * c))          # prefer fewer passes in flight on a tie
            args.streams = best
            streams = pool[:best] if best > 1 else None
            if rank == 0:
                print("[bench] calibration (ms/pass): %s -> --streams %d" %
                      (", ".join("%d in flight %.2f" % (c, res[c] * 1e3) for c in (1, 2, 3)), best), file=sys.stderr, flush=True)
        # warm-up; the last warm-up step is instrumented per kernel family to find the dominant one
        for i in range(max(args.warmup, 1)):
            if i == max(args.warmup, 1) - 1:
                ops.profiler.enable(None, tags=bool(args.breakdown))
                path(wave)                                # eager, so the per-launch events exist even with --graph
            else:
                step()
        breakdown = ops.profiler.summary()
        ops.profiler.disable()
        dominant = max((k for k in breakdown if "[" not in k), key=lambda k: breakdown[k]["ms_total"])
        if not args.graph:
            ops.profiler.enable({dominant})
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        dom = breakdown[dominant] if args.graph else ops.profiler.summary()[dominant]
        ops.profiler.disable()
        # with several passes in flight the live per-launch durations include the time a kernel shares the chip with the
        # other stream's kernels; a few strictly sequential passes AFTER the timed region give the exclusive durations
        dom_excl = None
        if streams is not None and not args.graph:
            torch.cuda.synchronize()
            ops.profiler.enable({dominant})
            for _ in range(3):
                path(wave)
            dom_excl = ops.profiler.summary()[dominant]
            ops.profiler.disable()

    tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    frames = world * B * T * args.steps
    if rank == 0:
        kind = FAMILY_BOUND.get(dominant, "hbm")
        secs = dom["ms_avg"] * 1e-3
        if kind == "hbm":
            ach, peak, unit, bound = dom["bytes"] / dom["n"] / secs / 1e9, PEAKS["hbm"], "GB/s", "hbm"
        else:
            ach, peak, unit, bound = dom["flops"] / dom["n"] / secs / 1e12, PEAKS[kind], "TFLOP/s", "mfma"
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # tools/pmc_summary.py) for this workload, when present; None otherwise
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.workload)
        if os.path.exists(pmc_path):
            traffic = json.load(open(pmc_path)).get(dominant, {}).get("hbm_bytes_per_launch")
        att = breakdown.get("attention_fwd")
        line = {
            "metric": "audio frames/sec/GPU (16 kHz, 512-frame utts) + mask RMSE vs CPU ref",
            "value": frames / elapsed, "unit": "STFT frames/s (whole job)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": desc, "batch_per_gpu": B, "samples": L, "frames_per_utt": T,
                       "sharding": "utterances over ranks, no data-path collective",
                       "launch": "one hipGraph replay per step" if args.graph else "eager launches",
                       "streams": args.streams},
            "roofline": {"bound": bound, "kernel": dominant, "achieved": ach, "peak": peak, "unit": unit,
                         "frac": ach / peak, "traffic": traffic, "traffic_unit": "HBM bytes/launch (PMC)",
                         "algorithmic_bytes_per_launch": dom["bytes"] / dom["n"],
                         "algorithmic_flops_per_launch": dom["flops"] / dom["n"], "launches": dom["n"],
                         "avg_ms": dom["ms_avg"]},
            "frames_per_s_per_gpu": frames / elapsed / world,
        }
        if dom_excl is not None:
            ex = (dom_excl["bytes"] if kind == "hbm" else dom_excl["flops"]) / dom_excl["n"] / (dom_excl["ms_avg"] * 1e-3) / \
                (1e9 if kind == "hbm" else 1e12)
            line["roofline"].update({"passes_in_flight": args.streams, "exclusive_achieved": ex, "exclusive_frac": ex / peak,
                                     "exclusive_avg_ms": dom_excl["ms_avg"],
                                     "note": "achieved / avg_ms are live in the timed region, where %d passes share the chip; "
                                             "exclusive_* are the same launches in 3 strictly sequential passes right after "
                                             "it (= what --streams 1 measures)" % args.streams})
        if att:
            tf = att["flops"] / att["n"] / (att["ms_avg"] * 1e-3) / 1e12
            line["attention"] = {"tflops": tf, "frac_bf16_mfma_peak": tf / PEAKS["mfma16"], "avg_ms": att["ms_avg"]}
        line["breakdown_ms_per_step"] = {k: round(v["ms_total"], 4) for k, v in
                                         sorted(breakdown.items(), key=lambda kv: -kv[1]["ms_total"]) if "[" not in k}
        print("[bench] gpu leg done: %.1f ms/step, %.3e frames/s; dominant kernel %s" %
              (line["ms_per_step"], line["value"], dominant), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights, L)
        if args.breakdown:
            with open(args.breakdown, "w") as fh:
                json.dump(breakdown, fh, indent=1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
