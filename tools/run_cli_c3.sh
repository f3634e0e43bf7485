#!/bin/bash
# The reference's README configuration (README.md:7) through the drop-in CLI on Criteo-Terabyte-shaped synthetic data, one
# GPU: prints the reference's own progress line ("Finished j/N in X ms/it. Caching overhead = ...") every 500 iterations.
#   bash tools/run_cli_c3.sh [num_batches]
NB=${1:-4000}
EMB=39884406-39043-17289-7420-20263-3-7120-1543-63-38532951-2953546-403346-10-2208-11938-155-4-976-14-39979771-25641295-39664984-585935-12972-108-36
exec python -m cdlrm_amd.main_no_ddp --arch-sparse-feature-size=128 --arch-mlp-bot=13-512-256-128 --arch-mlp-top=512-512-256-1 \
  --arch-embedding-size=$EMB --data-generation=criteo-synthetic --mini-batch-size=8192 --num-batches=$NB --lookahead=3000 \
  --cache-size=150000 --num-ways=16 --table-agg-freq=100 --learning-rate=0.8 --lr-embeds=0.8 --loss-function=bce \
  --round-targets=True --print-freq=${PF:-500} --world-size=1 --cache-workers=4 --batch-fifo-size=8 --device-rng
