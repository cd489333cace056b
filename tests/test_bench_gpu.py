"""bench.py end to end on the GPU box: the one-line JSON contract at N = 1, and the N = 2 launch exactly as the driver issues
it (python -m torch.distributed.run, one rank per 'GPU') rehearsed on the single card: both ranks map to device 0 and the
collectives run over gloo (SFM_SINGLE_DEVICE / SFM_DIST_BACKEND), which exercises rank handling, the utterance sharding, the
max-over-ranks timing, the gradient all-reduce of the training step and rank 0's aggregate line."""
import json
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
    assert all(k in d for k in REQUIRED + ["cpu_baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 201 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1 and r["achieved"] > 0 and r["peak"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1


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
