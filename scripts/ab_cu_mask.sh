#!/bin/bash
# A/B of the weight-gradient side stream's CU mask (S2S_WGRAD_CUS=K:M enables CU i iff i % M < K) on the full CFM step
# and the pix2pix G + D step; run on the GPU box from the repo root.
for rep in 1 2; do
  for spec in 0 3:4 7:8 192:256; do
    S2S_BENCH_OWN_STREAM=1 S2S_WGRAD_CUS=$spec python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cus=$spec', 'cfm', d['value'], d['ms_per_step'], 'pix2pix', d['pix2pix']['value'], d['pix2pix']['ms_per_step'])"
  done
done
