#!/bin/bash
# dev tool (GPU box): the GPU's clocks and power while the bench's workload runs (rocm-smi sampled next to back-to-back
# batches of 20 MSMs of 2^20 points), to put a number on "the chip clocks to its power budget under k_accumulate"
# (DESIGN.md section 4).      tools/clock_probe.sh > gpurun_out/r03_clocks_under_load.txt
cd "$(dirname "$0")/.."
echo "# idle:"; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | sed 's/^/  /'
rm -f /tmp/load_loop.ready
python tools/load_loop.py --seconds 8 > /tmp/load_loop.out 2>&1 &
BP=$!
for i in $(seq 1 600); do [ -f /tmp/load_loop.ready ] && break; sleep 0.1; done
echo "# under load (back-to-back batches of 20 x 2^20-point MSMs), one rocm-smi sample per line:"
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | sed 's/GPU\[0\]\t*: //' | tr '\n' '|' | sed 's/  */ /g'; echo
  sleep 0.1
done
wait $BP
cat /tmp/load_loop.out | tail -1
