#!/bin/bash
# Round-4 evidence set, part B (rocprofv3): kernel-trace statistics of the driver's command (forward legs only) and of one pass at a
# time, HBM traffic by separate PMC passes (FETCH_SIZE / WRITE_SIZE; FETCH doubled per the gfx950 note: tools/pmc_summary.py) for the
# primary workload (c3p) and configs[1] (c2), the attention kernel alone (trace + PMC), conv16p on its layer shapes (PMC), the
# training step (trace).  Everything lands under gpurun_out/r04/; the summaries are copied to profiles/r04/ by hand afterwards.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export SFM_ROUND=r04
O=$R/gpurun_out/r04
mkdir -p $O/prof
FW="--no-cpu-baseline --no-headline --no-train --no-configs1"
step() { echo "[profile] $1" | tee -a $O/prof/progress.txt; }
step "trace c3p (streams auto)"
rocprofv3 --kernel-trace --stats -d $O/prof/c3p -o c3p --output-format csv -- python3 $R/bench.py $FW --steps 20 --warmup 5 > $O/prof/bench_c3p_profiled.json 2> $O/prof/c3p.log
step "trace c3p streams 1"
rocprofv3 --kernel-trace --stats -d $O/prof/c3ps1 -o c3ps1 --output-format csv -- python3 $R/bench.py $FW --streams 1 --steps 20 --warmup 5 > $O/prof/bench_c3ps1_profiled.json 2> $O/prof/c3ps1.log
step "trace c2 streams 1"
rocprofv3 --kernel-trace --stats -d $O/prof/c2s1 -o c2s1 --output-format csv -- python3 $R/bench.py --workload c2 --no-cpu-baseline --streams 1 --steps 20 --warmup 5 > $O/prof/bench_c2s1_profiled.json 2> $O/prof/c2s1.log
for w in c3p c2; do
  step "pmc fetch $w"
  rocprofv3 --pmc FETCH_SIZE -d $O/prof/fetch_$w -o fetch --output-format csv -- python3 $R/bench.py --workload $w $FW --streams 1 --steps 3 --warmup 1 --no-sustained > /dev/null 2> $O/prof/fetch_$w.log
  step "pmc write $w"
  rocprofv3 --pmc WRITE_SIZE -d $O/prof/write_$w -o write --output-format csv -- python3 $R/bench.py --workload $w $FW --streams 1 --steps 3 --warmup 1 --no-sustained > /dev/null 2> $O/prof/write_$w.log
  python3 $R/tools/pmc_summary.py $(find $O/prof/fetch_$w -name "*counter_collection.csv") $(find $O/prof/write_$w -name "*counter_collection.csv") $O/pmc_traffic_$w.json > $O/prof/pmc_traffic_$w.txt
done
step "attention trace (50 launches from idle)"
rocprofv3 --kernel-trace --stats -d $O/prof/attn -o attn --output-format csv -- python3 $R/tools/attn_bench.py --iters 50 > $O/attn_bench_b256_t512.json 2> $O/prof/attn.log
step "attention trace (20 000 launches: steady state)"
rocprofv3 --kernel-trace --stats -d $O/prof/attns -o attns --output-format csv -- python3 $R/tools/attn_bench.py --iters 20000 > $O/attn_bench_b256_t512_sustained.json 2> $O/prof/attns.log
step "attention pmc"
bash $R/tools/pmc_attn_run.sh attention_pipe_t512 --iters 20 > $O/prof/pmc_attn.txt 2>&1
cp $O/pmc_attention_pipe_t512/summary.json $O/pmc_attention_pipe_t512.json 2>/dev/null
step "conv16p pmc"
bash $R/tools/pmc_conv_run.sh > $O/prof/pmc_conv16p.txt 2>&1
cp $O/pmc_conv16p/summary.json $O/pmc_conv16p.json 2>/dev/null
step "gemm16v2 pmc"
bash $R/tools/pmc_run.sh gemm16v2 gemm16v2_kernel tools/gemm_k256_bench.py > $O/prof/pmc_gemm16v2.txt 2>&1
cp $O/pmc_gemm16v2/summary.json $O/pmc_gemm16v2.json 2>/dev/null
step "trace c3t"
rocprofv3 --kernel-trace --stats -d $O/prof/c3t -o c3t --output-format csv -- python3 $R/bench.py --workload c3t --no-cpu-baseline --steps 4 --warmup 2 > $O/prof/bench_c3t_profiled.json 2> $O/prof/c3t.log
for t in c3p c3ps1 c2s1 attn attns c3t; do
  f=$(find $O/prof/$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_$t.csv && echo "== $t" && head -8 $f | cut -c1-170
done
cat $O/prof/pmc_traffic_c3p.txt | head -12
step "done"
