#!/bin/bash
# Round-4 evidence set, part A (un-profiled): the full GPU suite, smoke(), the driver's command, every other bench workload, the
# per-shape profiles, the training precision probe.  -> gpurun_out/r04/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/ -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.txt
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; echo "bench default rc $?"
for w in c2 c3t c3se c2t c5; do
  timeout -k 10 400 python3 bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.log; echo "bench $w rc $?"
done
timeout -k 10 200 python3 bench.py --streams 1 --no-cpu-baseline --no-train --no-configs1 > $O/bench_c3p_streams1.json 2>/dev/null; echo "bench c3p streams1 rc $?"
timeout -k 10 300 python3 tools/train_profile.py --workload c3t > $O/train_profile_c3t.txt 2>&1; echo "train_profile rc $?"
timeout -k 10 300 python3 tools/infer_profile.py > $O/infer_profile_c2.txt 2>&1; echo "infer_profile rc $?"
timeout -k 10 300 python3 tests/probe_train_precision.py > $O/train_precision_probe.json 2> $O/train_precision_probe.log; echo "probe rc $?"; tail -3 $O/train_precision_probe.log
python3 - <<PY
import json
for w in ("default", "c2", "c3t", "c3se", "c2t", "c5"):
    try:
        d = json.loads(open("$O/bench_%s.json" % w).read().strip().splitlines()[-1])
        print(w, "%.3f ms/step  %.4g %s" % (d["ms_per_step"], d["value"], d["unit"]))
    except Exception as e:
        print(w, "no line", e)
PY
