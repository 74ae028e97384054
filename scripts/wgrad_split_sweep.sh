#!/bin/bash
# Workgroup-count sweep of the weight-gradient launches over the FULL optimisation step (the kernel alone prefers two
# workgroups per CU; the step prefers one).  Run on the GPU box from the repo root:
#   scripts/wgrad_split_sweep.sh            CFM step, S2S_WGRAD_BLOCKS
#   scripts/wgrad_split_sweep.sh pix2pix    pix2pix G + D step, S2S_P2P_WGRAD_BLOCKS
MODE=${1:-cfm}
if [ "$MODE" = "pix2pix" ]; then VAR=S2S_P2P_WGRAD_BLOCKS; A="--steps 60 --warmup 10 --no-extras --mode pix2pix"
else VAR=S2S_WGRAD_BLOCKS; A="--steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-pix2pix"; fi
for rep in 1 2; do
  for v in 512 384 256 224 192 128; do
    env $VAR=$v python bench.py $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], 'ms/step', {k: (x.get('tflops'), x['ms_per_step']) for k, x in d['kernels'].items()})"
  done
done
# the kernel alone, per layer shape:  S2S_WGRAD_BLOCKS=512 python scripts/conv_bench.py wgrad
