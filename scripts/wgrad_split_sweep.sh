# Split-K sweep of the weight-gradient kernel (S2S_WGRAD_BLOCKS / S2S_WGRAD_CAP) on the headline layer shapes.
for cfg in "512 0" "512 512" "640 512" "768 512" "512 0"; do
set -- $cfg
echo "BLOCKS=$1 CAP=$2"
S2S_WGRAD_BLOCKS=$1 S2S_WGRAD_CAP=$2 python scripts/conv_bench.py wgrad 2>&1 | grep -v amdgpu | awk '{print $0}' | cut -c1-90
done
