cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${SFM_ROUND:-r03}
mkdir -p $O/prof
# 1. kernel trace of the default bench command and of --streams 1   (SKIP_TRACE=1: already taken in this round)
if [ -z "$SKIP_TRACE" ]; then
rocprofv3 --kernel-trace --stats -d $O/prof/c2 -o c2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-headline > $O/prof/bench_c2_profiled.json 2> $O/prof/c2.log
rocprofv3 --kernel-trace --stats -d $O/prof/c2s1 -o c2s1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-headline --streams 1 > $O/prof/bench_c2s1_profiled.json 2> $O/prof/c2s1.log
fi
echo "traces done" > $O/prof/progress.txt
# 2. HBM traffic: separate PMC passes (streams 1 so that kernels do not overlap; no sustained loop: counter collection
#    serialises every launch, 250+ steps of it ran into the 7-minute silence limit of the box)
rocprofv3 --pmc FETCH_SIZE -d $O/prof/fetch -o fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-headline --streams 1 --steps 3 --warmup 1 --no-sustained > /dev/null 2> $O/prof/fetch.log
echo "fetch done" >> $O/prof/progress.txt
rocprofv3 --pmc WRITE_SIZE -d $O/prof/write -o write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-headline --streams 1 --steps 3 --warmup 1 --no-sustained > /dev/null 2> $O/prof/write.log
echo "write done" >> $O/prof/progress.txt
python3 $R/tools/pmc_summary.py $(find $O/prof/fetch -name "*counter_collection.csv") $(find $O/prof/write -name "*counter_collection.csv") $O/pmc_traffic_c2.json > $O/prof/pmc_traffic.txt
# 3. attention headline: kernel trace
rocprofv3 --kernel-trace --stats -d $O/prof/attn -o attn --output-format csv -- python3 $R/tools/attn_bench.py --iters 50 > $O/attn_bench_b256_t512.json 2> $O/prof/attn.log
# 4. un-profiled reference runs
python3 $R/bench.py > $O/bench_c2.json 2> $O/bench_c2.log
python3 $R/bench.py --streams 1 --no-cpu-baseline > $O/bench_c2_streams1.json 2>/dev/null
ls $O/prof/*/ | head -30
for f in $(find $O/prof -name "*kernel_stats.csv"); do echo $f; head -12 $f; done
cat $O/prof/pmc_traffic.txt
