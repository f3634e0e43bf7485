#!/bin/bash
# Host-side profile of the drop-in CLI at the README configuration (where does an iteration's host time go?):
#   bash tools/cli_profile.sh [num_batches]   -> gpurun_out/cli_profile.txt
NB=${1:-1500}
EMB=39884406-39043-17289-7420-20263-3-7120-1543-63-38532951-2953546-403346-10-2208-11938-155-4-976-14-39979771-25641295-39664984-585935-12972-108-36
mkdir -p gpurun_out
python -m cProfile -o /tmp/cli.prof -m cdlrm_amd.main_no_ddp --arch-sparse-feature-size=128 --arch-mlp-bot=13-512-256-128 --arch-mlp-top=512-512-256-1 \
  --arch-embedding-size=$EMB --data-generation=criteo-synthetic --mini-batch-size=8192 --num-batches=$NB --lookahead=3000 \
  --cache-size=150000 --num-ways=16 --table-agg-freq=100 --learning-rate=0.8 --lr-embeds=0.8 --loss-function=bce \
  --round-targets=True --print-freq=500 --world-size=1 --cache-workers=4 --batch-fifo-size=8 --device-rng > gpurun_out/cli_profile.log 2>&1
python - <<'PY' > gpurun_out/cli_profile.txt
import pstats
p = pstats.Stats('/tmp/cli.prof')
p.sort_stats('cumulative').print_stats(45)
p.sort_stats('tottime').print_stats(25)
PY
tail -5 gpurun_out/cli_profile.log
