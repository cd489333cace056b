# end-of-round measurement set: full GPU suite, training / long-utterance bench lines, kernel trace of the training step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${SFM_ROUND:-r03}
mkdir -p $O/prof
cd $R
timeout -k 10 400 python3 -m pytest tests/ -q -m gpu -x > $O/pytest_gpu.txt 2>&1 || { tail -20 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
for w in c3t c3se c5 c2t; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.log || { echo "bench $w failed"; tail -5 $O/bench_$w.log; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'], d['unit'])"
done
timeout -k 10 300 python3 tools/train_profile.py --workload c3t > $O/train_profile_c3t.txt 2>&1 && head -3 $O/train_profile_c3t.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof/c3t -o c3t --output-format csv -- python3 $R/bench.py --workload c3t --no-cpu-baseline --steps 4 --warmup 2 > $O/prof/bench_c3t_profiled.json 2> $O/prof/c3t.log
for f in $(find $O/prof/c3t -name "*kernel_stats.csv"); do cp $f $O/rocprofv3_kernel_stats_bench_c3t.csv; head -8 $f | cut -c1-160; done
