#!/bin/bash
# shader clock / power while the bench loop runs: is the step clock-limited (DVFS) rather than kernel-limited?
# usage (GPU box, repo root): bash tools/clock_watch.sh > gpurun_out/clocks.txt
python bench.py --no-cpu-baseline --no-roofline --no-ocr-excluded --steps 400 --warmup 3 > /tmp/cw_bench.json 2>/dev/null &
BP=$!
sleep 45
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
head -c 300 /tmp/cw_bench.json; echo
echo "--- idle"
sleep 3
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' '
