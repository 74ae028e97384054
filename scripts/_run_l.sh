set -e
cd /root/repo; mkdir -p gpurun_out/r03l
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
S2S_P2P_OPT_IN_BWD=$v rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r03l/tr$v -o k --output-format csv -- python3 /root/repo/bench.py --mode pix2pix --steps 4 --warmup 2 --no-extras > /root/repo/gpurun_out/r03l/tr$v.json 2> /root/repo/gpurun_out/r03l/tr$v.err
D=$(dirname $(find /root/repo/gpurun_out/r03l/tr$v -name k_kernel_trace.csv | head -n 1))
python3 /root/repo/scripts/step_overlap.py $D 6 > /root/repo/gpurun_out/r03l/overlap$v.txt 2>&1 || true
cat /root/repo/gpurun_out/r03l/overlap$v.txt
head -n 3 $D/k_kernel_trace.csv > /root/repo/gpurun_out/r03l/trace_head$v.csv
rm -rf /root/repo/gpurun_out/r03l/tr$v
done
