# Cost of the roofline HIP-event brackets: same box, alternating runs with brackets on every step, every 4th
# step and the first step only (instrumentation note).
for r in 1 2; do
for e in 1 4 1000; do
echo "every=$e"; python bench.py --steps 120 --warmup 5 --no-cpu-baseline --event-every $e | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['launches_timed'], d['kernels']['conv3x3_wgrad_mfma'])"
done; done
