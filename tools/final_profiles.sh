#!/bin/bash
# Every committed line of a round from ONE box and ONE build (run through gpurun from the repo root):
#   bash tools/final_profiles.sh r05        -> gpurun_out/final/<tag>_*  (copy what is to be judged into profiles/)
# bench lines (driver's run, default run, per-rank batch, c2, c4 capped, uniform indices), the CLI at c3, then the rocprofv3 passes
# of tools/profile_round.sh (kernel stats + trace, FETCH_SIZE / WRITE_SIZE of the gather, MFMA busy) with their summaries.
# The c5 whole-window line (8000 steps, ~2 min) runs with C5=1 only.
set -e
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
line() { tail -n 1 "$1" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ms/step %.4f  value %.4g  gather %.2f us  frac %.3f  loss %.6f' % (d['ms_per_step'], d['value'], r['avg_launch_us'] or 0, r['frac'] or 0, d['config']['final_loss']))"; }
run() { name=$1; shift; echo "== $name: bench.py $*"; python3 bench.py "$@" > $OUT/${TAG}_bench_$name.json 2> $OUT/${TAG}_bench_$name.err; line $OUT/${TAG}_bench_$name.json; }
# PART=lines: the bench lines only; PART=prof: the rocprofv3 passes only (a gpurun call is at most 20 minutes: the two halves
# of a round's profiles go in two calls, on two boxes -- each half is self-consistent); default: both
if [ "${PART:-all}" != prof ]; then
run steps20_n1 --steps 20 --warmup 5
run steps20_two_launches_n1 --steps 20 --warmup 5 --no-fuse-gather --no-cpu-baseline --whole-window off
run steps20_chained_n1 --steps 20 --warmup 5 --no-cpu-baseline --whole-window on --engine-attr gather_alone_min=8192
run default_n1 --no-cpu-baseline
run c3_batch1024_n1 --batch 1024 --steps 1000 --warmup 100 --no-cpu-baseline
run c2_n1 --config c2 --steps 600 --warmup 50 --no-cpu-baseline
run c4_capped_n1 --config c4 --max-ind-range 2000000 --steps 300 --warmup 50 --no-cpu-baseline
run c3_a0_whole_window_n1 --alpha 0 --steps 3000 --no-cpu-baseline
run c3_batch4096_n1 --batch 4096 --steps 1000 --warmup 100 --no-cpu-baseline
run c3_batch2048_n1 --batch 2048 --steps 1000 --warmup 100 --no-cpu-baseline
if [ -n "$C5" ]; then run c5_whole_window_n1 --config c5 --steps 8000 --no-cpu-baseline; fi
# the drop-in CLI at the README configuration (its own progress lines: ms per iteration without the caching overhead)
echo "== cli_c3_n1: tools/run_cli_c3.sh 4000"; bash tools/run_cli_c3.sh 4000 > $OUT/${TAG}_cli_c3_n1.log 2>&1; grep Finished $OUT/${TAG}_cli_c3_n1.log | tail -2 | cut -c1-70
fi
if [ "${PART:-all}" = lines ]; then ls -la $OUT; exit 0; fi
echo "== rocprofv3 passes (c3)"
bash tools/profile_round.sh c3
P=$ROOT/gpurun_out/prof_c3
python3 tools/pmc_summary.py $P/fetch $P/write $OUT/${TAG}_gather_pmc_c3_a1p05.json c3 1.05
python3 tools/mfma_summary.py $P/mfma $OUT/${TAG}_mfma_pmc.json
python3 tools/gather_launches.py $P/stats $OUT/${TAG}_c3_gather_launches.json
cp $(ls -t $P/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_c3_n1_kernel_stats.csv
python3 tools/trace_timeline.py $(ls -t $P/stats/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_c3_step_timeline.txt
python3 tools/a6_summary.py $P/stats $OUT/${TAG}_a6_whole_c3.json
echo "== rocprofv3 passes (c3, uniform indices)"
PASSES="stats fetch write" bash tools/profile_round.sh a0 --alpha 0
PA=$ROOT/gpurun_out/prof_a0
python3 tools/pmc_summary.py $PA/fetch $PA/write $OUT/${TAG}_gather_pmc_c3_a0.json c3 0
python3 tools/gather_launches.py $PA/stats $OUT/${TAG}_c3_a0_gather_launches.json
cp $(ls -t $PA/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_c3_a0_n1_kernel_stats.csv
if [ -n "$C5" ]; then
  echo "== rocprofv3 counter passes (c5: the gather at 65536 lookups per table)"
  STEPS=24 WARMUP=6 PASSES="stats fetch write" bash tools/profile_round.sh c5 --config c5
  P5=$ROOT/gpurun_out/prof_c5
  python3 tools/pmc_summary.py $P5/fetch $P5/write $OUT/${TAG}_gather_pmc_c5_a1p05.json c5 1.05
  python3 tools/gather_launches.py $P5/stats $OUT/${TAG}_c5_gather_launches.json 65536
fi
echo "== rocprofv3 kernel trace (per-rank batch 1024)"
PASSES=stats bash tools/profile_round.sh b1024 --batch 1024
python3 tools/trace_timeline.py $(ls -t $ROOT/gpurun_out/prof_b1024/stats/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_c3_batch1024_step_timeline.txt
PASSES=stats bash tools/profile_round.sh b2048 --batch 2048
python3 tools/trace_timeline.py $(ls -t $ROOT/gpurun_out/prof_b2048/stats/*/*kernel_trace.csv | head -1) > $OUT/${TAG}_c3_batch2048_step_timeline.txt
ls -la $OUT
