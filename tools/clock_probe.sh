#!/bin/bash
# dev tool (GPU box): the GPU's clocks and power while the bench's batch runs (rocm-smi sampled every ~0.2 s next to a long
# batch), to put a number on "the chip clocks to its power budget under k_accumulate" (DESIGN.md section 4).
#   tools/clock_probe.sh > gpurun_out/r03_clocks_under_load.txt
cd "$(dirname "$0")/.."
echo "# idle:"; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | sed 's/^/  /'
python bench.py --steps 400 --warmup 4 --no-cpu-baseline --no-secondary > /tmp/probe_bench.json 2>/dev/null &
BP=$!
sleep 6   # import + input generation + the initialisation passes
echo "# under load (python bench.py --steps 400: ~0.6 s of back-to-back batches), one sample per line:"
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' | sed 's/  */ /g'; echo
  sleep 0.15
done
wait $BP
python -c "
import json; d=json.load(open('/tmp/probe_bench.json')); print('# bench: %.4f ms per MSM at %d steps' % (d['ms_per_step'], d['steps']))"
