set -e
cd /root/repo; mkdir -p gpurun_out/r03j
python -m pytest tests/test_instnorm_gpu.py tests/test_pix2pix_engine_gpu.py tests/test_pix2pix_gpu.py -x -q -m gpu > gpurun_out/r03j/tests.log 2>&1 || { tail -n 30 gpurun_out/r03j/tests.log; exit 1; }
tail -n 3 gpurun_out/r03j/tests.log
python bench.py --mode pix2pix --steps 30 --warmup 5 --no-extras > gpurun_out/r03j/bench_p2p.json 2> gpurun_out/r03j/bench_p2p.err
cat gpurun_out/r03j/bench_p2p.json
cd /tmp && export TMPDIR=/tmp
S2S_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03j/p2p_trace -o k --output-format csv -- python3 /root/repo/bench.py --mode pix2pix --steps 3 --warmup 1 --no-extras > /root/repo/gpurun_out/r03j/p2p_trace.json 2> /root/repo/gpurun_out/r03j/p2p_trace.err
cd /root/repo; mkdir -p gpurun_out/r03j
D=$(dirname $(find gpurun_out/r03j/p2p_trace -name k_kernel_stats.csv | head -n 1))
python scripts/step_timeline.py $D 4 --timeline 4 60 > gpurun_out/r03j/p2p_timeline.txt
rm -rf gpurun_out/r03j/p2p_trace
head -n 45 gpurun_out/r03j/p2p_timeline.txt
