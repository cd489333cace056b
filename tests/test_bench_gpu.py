"""bench.py end to end on the GPU box: the one-line JSON contract at N = 1, and the N = 2 launch exactly as the driver issues
it (python -m torch.distributed.run, one rank per 'GPU') rehearsed on the single card: both ranks map to device 0 and the
collectives run over gloo (SFM_SINGLE_DEVICE / SFM_DIST_BACKEND), which exercises rank handling, the utterance sharding, the
max-over-ranks timing, the gradient all-reduce of the training step and rank 0's aggregate line."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_bench_single_gpu_line():
    d = _json_line(_run([sys.executable, "bench.py", "--workload", "c1", "--steps", "6", "--warmup", "2"]))
    assert d["sustained"]["steps"] >= 250 and d["sustained"]["seconds"] >= 2.0 and 0.5 < d["sustained"]["vs_timed_region"] < 2.0
    assert all(k in d for k in REQUIRED + ["cpu_baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 201 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1 and r["achieved"] > 0 and r["peak"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    # the line carries the parity check its metric string promises, at the north-star bound, for the dtype it names
    assert d["mask_rmse"]["bound"] == 1e-3 and 0 < d["mask_rmse"]["value"] <= 1e-3
    assert "fp16" in d["dtype"] and "bf16 attention" in d["dtype"] and d["config"]["precision_policy"] == "mixed"
    # the per-family breakdown comes from ONE pass alone on the device: it cannot exceed the single-pass step
    assert sum(d["breakdown_ms_per_step"].values()) <= 1.1 * d["single_pass_ms_per_step"]


def test_bench_default_line_is_on_the_metrics_shape_and_carries_configs1_and_train():
    """the driver's one command: primary workload = the shape BASELINE's metric names (B 256 x 512-frame utterances), with
    BASELINE configs[1] (`configs1`), configs[2] on the north-star composition (`train`: fp16 operands under the device-side
    dynamic loss scale, no skipped step inside its timed steps) and the attention kernel alone (`headline`) as sub-records;
    the roof of the dominant family is chosen per launch"""
    d = _json_line(_run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--train-batch", "16",
                         "--train-steps", "2", "--train-warmup", "1"]))
    assert "B256 x 512-frame" in d["config"]["workload"] and d["config"]["frames_per_utt"] == 512
    assert abs(d["value"] - 256 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert d["roofline"]["traffic"] is None or "NOT measured" in d["roofline"]["traffic_source"]
    assert sum(d["breakdown_ms_per_step"].values()) <= 1.1 * d["single_pass_ms_per_step"]
    r = d["roofline"]
    assert "per launch" in r["roof_choice"] and 0 < r["roofline_time_frac"] <= 1 and r["family_launches"] >= r["launches"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    a = d["headline"]["attention"]
    assert a["operands"] == "bf16" and a["tflops"] > 300 and abs(a["frac_bf16_mfma_peak"] - a["tflops"] / 2500.0) < 1e-9
    assert len(a["window_avg_ms"]) == 5 and a["sustained_launches"] >= 2000 and 0 < a["frac_windows_median"] < 1
    assert abs(a["tflops"] - 4.0 * 256 * 4 * 512 * 512 * 64 / a["avg_ms"] / 1e9) < 1e-6 * a["tflops"]
    c = d["configs1"]
    assert "B64 x 4 s" in c["workload"] and c["frames_per_s"] > 1e6 and c["roofline"]["frac"] > 0
    assert "T801" in c["attention"]["shape"] and c["attention"]["tflops"] > 100
    t = d["train"]
    assert "B256 x 4 s" in t["workload"] and t["batch_per_gpu"] == 16 and t["steps"] == 2
    assert t["applied_steps_in_timed_region"] == 2 and t["loss_scale"]["skipped_in_timed_region"] == 0
    assert 0 < t["loss_scale"]["scale"] <= 65536.0 and "dynamic loss scale" in t["dtype"]
    assert abs(t["frames_per_s"] - 16 * 801 / (t["ms_per_step"] * 1e-3)) < 1e-3 * t["frames_per_s"]
    assert t["roofline"]["bound"] in ("hbm", "mfma") and 0 < t["roofline"]["frac"] <= 1 and math.isfinite(t["final_loss"])
    assert not t["optimizer_state"]["skipped"]


def test_bench_refuses_to_time_with_a_variant_override():
    e = dict(os.environ)
    e["SFM_GEMM_VARIANT"] = "9"
    r = subprocess.run([sys.executable, "bench.py", "--workload", "c1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-sustained"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "variant override" in (r.stderr + r.stdout)


def test_bench_gpus_flag_starts_the_ranks_itself():
    """plain `python bench.py --gpus 2` (no launcher): the parent spawns both ranks before touching the GPU; rehearsed on the
    single card (both ranks on device 0, collectives over gloo)"""
    env = {"SFM_SINGLE_DEVICE": "1", "SFM_DIST_BACKEND": "gloo"}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    d = _json_line(_run([sys.executable, "bench.py", "--gpus", "2", "--workload", "c1", "--steps", "3", "--warmup", "1"], env))
    assert d["n_gpus"] == 2 and abs(d["value"] - 2 * 201 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert "cpu_baseline" not in d


def test_bench_rejects_a_gpus_flag_that_disagrees_with_the_launcher():
    e = dict(os.environ)
    e.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "c1", "--steps", "1", "--warmup", "1"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.parametrize("workload,extra", [("c1", []), ("c3se", ["--batch", "4"])])
def test_bench_two_ranks_rehearsal(workload, extra):
    env = {"SFM_SINGLE_DEVICE": "1", "SFM_DIST_BACKEND": "gloo"}
    port = "29531" if workload == "c1" else "29532"
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", port, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload] + extra, env)
    d = _json_line(out)
    assert all(k in d for k in REQUIRED)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    per_rank = (1 if workload == "c1" else 4) * (201 if workload == "c1" else 801)
    assert abs(d["value"] - 2 * per_rank / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]      # whole job = both ranks
    assert "cpu_baseline" not in d                                                         # N = 1 only


def test_bench_default_two_ranks_carry_the_dp_train_record():
    """the driver's N > 1 command (no --workload): the line must carry BOTH the forward shard (no data-path collective) and the
    `dp_train` sub-record - the training step of training/conformer_pipeline.py:496-532 with the bucketed gradient all-reduce
    (north_star: "RCCL all-reduce of gradients"; here over gloo, two ranks on the one card) - and the sustained figure"""
    env = {"SFM_SINGLE_DEVICE": "1", "SFM_DIST_BACKEND": "gloo"}
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-headline",
                "--configs1-batch", "2", "--dp-batch", "4", "--dp-steps", "2", "--dp-warmup", "1"], env)
    d = _json_line(out)
    assert all(k in d for k in REQUIRED) and d["n_gpus"] == 2 and "B256 x 512-frame" in d["config"]["workload"]
    assert "train" not in d and "B64 x 4 s" in d["configs1"]["workload"]
    assert "no data-path collective" in d["config"]["sharding"]
    s = d["sustained"]
    # (two timed steps of two ranks sharing one card: the ratio to the timed region is only checked for sanity)
    assert s["steps"] >= 250 and s["seconds"] >= 2.0 and s["ms_per_step"] > 0 and 0.1 < s["vs_timed_region"] < 10.0
    t = d["dp_train"]
    assert t["rccl_ranks"] == 2 and t["backend"] == "gloo" and t["batch_per_gpu"] == 4 and t["steps"] == 2 and t["overlap"] is True
    assert t["allreduce_bytes_per_step"] > 20e6 and t["allreduce_buckets"] >= 2          # SpeechEnhancer: 24.9 MB of fp32 gradients
    assert t["ms_per_step"] > 0 and abs(t["frames_per_s"] - 2 * 4 * 801 / (t["ms_per_step"] * 1e-3)) < 1e-3 * t["frames_per_s"]
    assert t["exposed_allreduce_ms_per_step"] is not None and 0 <= t["exposed_allreduce_ms_per_step"] <= t["ms_per_step"]
    assert t["applied_steps_in_timed_region"] == 2 and not t["optimizer_state"]["skipped"]
    assert t["loss_scale"]["skipped_in_timed_region"] == 0
