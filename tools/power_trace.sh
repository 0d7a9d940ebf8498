#!/bin/bash
# sclk / power samples (rocm-smi, every 0.2 s) while the unit-tensor kernels run back to back (tools/bench_units.py) and while the
# bench's graph-replayed steps run: the evidence asked for the "power-limited" reading of MFMA-busy 0.57-0.69.
# usage (GPU box, repo root): tools/power_trace.sh TAG
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/${TAG}_power_trace.txt
SMI=/opt/rocm/bin/rocm-smi
sample() {  # one line per sample: t, sclk, mclk, power
  while true; do
    L=$($SMI --showclocks --showpower 2>/dev/null | tr '\n' ' ')
    S=$(echo "$L" | grep -o 'sclk clock level: [0-9]*: ([0-9]*Mhz)' | head -1 | grep -o '([0-9]*Mhz)')
    M=$(echo "$L" | grep -o 'mclk clock level: [0-9]*: ([0-9]*Mhz)' | head -1 | grep -o '([0-9]*Mhz)')
    P=$(echo "$L" | grep -o 'Power (W): [0-9.]*' | head -1)
    echo "$(date +%s.%N | cut -c1-14) sclk $S mclk $M $P"
    sleep 0.2
  done
}
echo "# idle" > $OUT; (sample >> $OUT) & SP=$!; sleep 2
echo "# tools/loop_units.py 4: conv_units_fprop_kernel<3,2> for 4 s, then conv_wgrad_units_kernel<3> for 4 s (64x256 128->128, B=16)" >> $OUT
(cd $R && python tools/loop_units.py 4 > $R/gpurun_out/${TAG}_loop_units.txt 2>&1)
echo "# graph-replayed plain steps (tools/trace_graph_step.py 1 200)" >> $OUT
(cd $R && python tools/trace_graph_step.py 1 200 > /dev/null 2>&1)
echo "# idle again" >> $OUT; sleep 2
kill $SP
$SMI --showclocks --showpower > $R/gpurun_out/${TAG}_rocm_smi_raw.txt 2>&1
