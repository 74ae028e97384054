#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats, and the two PMC passes the
# roofline's `traffic` comes from.  Outputs land in gpurun_out/prof_$1/; scripts/summarise_profiles.py turns them
# into the files committed under profiles/.
set -e
TAG=${1:-r01_x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python bench.py --steps 8 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python bench.py --steps 3 --warmup 1 > $OUT/pmc_write.out 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
