set -e
cd /root/repo; mkdir -p gpurun_out/r03m
python -m pytest tests/test_pix2pix_gpu.py tests/test_pix2pix_engine_gpu.py tests/test_graph_step_gpu.py tests/test_instnorm_gpu.py -x -q -m gpu > gpurun_out/r03m/tests.log 2>&1 || { tail -n 40 gpurun_out/r03m/tests.log; exit 1; }
tail -n 3 gpurun_out/r03m/tests.log
python __graft_entry__.py smoke 2>&1 | tail -n 3
python bench.py --mode pix2pix --steps 30 --warmup 5 --no-extras > gpurun_out/r03m/bench_p2p.json 2> gpurun_out/r03m/bench_p2p.err
python scripts/bench_summary.py gpurun_out/r03m/bench_p2p.json
