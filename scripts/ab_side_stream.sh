#!/bin/bash
# A/B on one box: weight gradients on a side stream (default) vs everything on one stream, alternating, 100 steps each.
# usage: ab_side_stream.sh [train|pix2pix]
MODE=${1:-train}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 1 0; do
    S2S_WGRAD_STREAM=$v python bench.py --mode $MODE --steps 100 --warmup 10 --no-cpu-baseline --no-pix2pix 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE side=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
