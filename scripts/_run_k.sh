set -e
cd /root/repo; mkdir -p gpurun_out/r03k
python -m pytest tests/test_graph_step_gpu.py tests/test_pix2pix_engine_gpu.py -x -q -m gpu > gpurun_out/r03k/tests.log 2>&1 || { tail -n 40 gpurun_out/r03k/tests.log; exit 1; }
tail -n 3 gpurun_out/r03k/tests.log
python bench.py --mode pix2pix --steps 30 --warmup 5 --no-extras > gpurun_out/r03k/bench_p2p.json 2> gpurun_out/r03k/bench_p2p.err
python scripts/bench_summary.py gpurun_out/r03k/bench_p2p.json || cat gpurun_out/r03k/bench_p2p.json
S2S_P2P_OPT_IN_BWD=0 python bench.py --mode pix2pix --steps 30 --warmup 5 --no-extras > gpurun_out/r03k/bench_p2p_eos.json 2> gpurun_out/r03k/bench_p2p_eos.err
python scripts/bench_summary.py gpurun_out/r03k/bench_p2p_eos.json || cat gpurun_out/r03k/bench_p2p_eos.json
