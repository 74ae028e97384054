#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats, and the two PMC passes the
# roofline's `traffic` comes from, for the CFM step and for the pix2pix G + D step.  Outputs land in
# gpurun_out/prof_$1/; scripts/summarise_profiles.py turns them into the files committed under profiles/.
# The kernel-stats and PMC passes run with S2S_WGRAD_STREAM=0 (everything on one stream): a kernel's average duration
# is then its own, as in bench.py's event-bracketed steps; <tag>/stats_overlap is the default two-stream schedule.
set -e
TAG=${1:-r02_x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
export S2S_WGRAD_STREAM=0
CFM="--no-pix2pix --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python bench.py --steps 8 --warmup 2 $CFM > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 $CFM > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python bench.py --steps 3 --warmup 1 $CFM > $OUT/pmc_write.out 2> $OUT/pmc_write.err
# SQ / GRBM passes: wave-cycle split, LDS conflicts and the MFMA instruction counts behind a non-saturated matrix-pipe
# utilisation (scripts/sq_counters.py)
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o q -- python bench.py --steps 3 --warmup 1 $CFM > $OUT/pmc_sq1.out 2> $OUT/pmc_sq1.err
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o q -- python bench.py --steps 3 --warmup 1 $CFM > $OUT/pmc_sq2.out 2> $OUT/pmc_sq2.err
echo "cfm passes done"
P2P="--mode pix2pix --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p2p_stats -o k -- python bench.py --steps 8 --warmup 2 $P2P > $OUT/p2p_bench_under_rocprof.json 2> $OUT/p2p_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p2p_pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 $P2P > $OUT/p2p_pmc_fetch.out 2> $OUT/p2p_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p2p_pmc_write -o w -- python bench.py --steps 3 --warmup 1 $P2P > $OUT/p2p_pmc_write.out 2> $OUT/p2p_pmc_write.err
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/p2p_pmc_sq1 -o q -- python bench.py --steps 3 --warmup 1 $P2P > $OUT/p2p_pmc_sq1.out 2> $OUT/p2p_pmc_sq1.err
echo "pix2pix passes done"
unset S2S_WGRAD_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_overlap -o k -- python bench.py --steps 8 --warmup 2 $CFM > $OUT/bench_under_rocprof_overlap.json 2> $OUT/stats_overlap.err
# the per-dispatch traces are large: keep the counter tables and the statistics, drop the kernel traces of the PMC passes
find $OUT -path "*pmc*" -name "*kernel_trace.csv" -delete
find $OUT -name "*.csv" | head -40
