#!/bin/bash
# One round's rocprofv3 passes of bench.py on the MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag> [extra bench args]   (STEPS=50 WARMUP=10: 60 steps stay in front of the iteration
#   at which bench.py launches the next window's plan, so every profiled launch runs without a plan beside it)      e.g.  bash tools/profile_round.sh c3 ; ... a0 --alpha 0
# Passes (separate runs: --pmc is never combined with other trace domains):
#   stats   : --kernel-trace --stats                                   -> per-kernel averages + the raw kernel trace
#   fetch   : --kernel-trace --pmc FETCH_SIZE                          -> HBM read bytes of the gather kernel
#   write   : --kernel-trace --pmc WRITE_SIZE                          -> HBM write bytes
#   mfma    : --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
# (--prewarm-ms 0: the clock pre-warm's scratch GEMMs would otherwise sit in the kernel statistics beside the step's own;
#  --whole-window off / --no-standalone-legs: the 3000-step leg and the stand-alone operator timings behind the timed region
#  are not what these passes profile: the kernel statistics hold the step's launches only)
# Output under gpurun_out/prof_<tag>/{stats,fetch,write,mfma}; summaries are made afterwards by tools/pmc_summary.py,
# tools/mfma_summary.py and tools/gather_launches.py and committed under profiles/.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=${PASSES:-"stats fetch write mfma"}
for p in $PASSES; do
  case $p in
    stats) ARGS="--kernel-trace --stats" ;;
    fetch) ARGS="--kernel-trace --pmc FETCH_SIZE" ;;
    write) ARGS="--kernel-trace --pmc WRITE_SIZE" ;;
    mfma)  ARGS="--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" ;;
  esac
  echo "== pass $p"
  rocprofv3 $ARGS --output-format csv -d $OUT/$p -- python3 $ROOT/bench.py --no-cpu-baseline --whole-window off --no-standalone-legs --prewarm-ms 0 --steps ${STEPS:-50} --warmup ${WARMUP:-10} "$@" > $OUT/$p.log 2>&1
  tail -c 400 $OUT/$p.log | tr '\n' ' ' | cut -c1-300; echo
done
