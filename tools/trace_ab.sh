#!/bin/bash
# per-kernel time under graph replay for two settings of ops.TUNING, side by side (rocprofv3 --kernel-trace --stats, 24 plain steps each)
# usage (GPU box, repo root): tools/trace_ab.sh TAG "attrA=valA" "attrB=valB" [dtype] [batch]
TAG=$1; A=$2; B=$3; export TBG_DTYPE=${4:-f32x3}; export TBG_BATCH=${5:-16}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for V in A B; do
  if [ $V = A ]; then export TBG_TUNING="$A"; else export TBG_TUNING="$B"; fi
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_${TAG}_$V -o t -- python tools/trace_graph_step.py 1 20 > /tmp/tr_${TAG}_$V.log 2>&1)
  (cd $R && python tools/prof_summary.py /tmp/tr_${TAG}_$V 70 > $R/gpurun_out/${TAG}_${V}_kernels.txt; python tools/prof_groups.py /tmp/tr_${TAG}_$V 24 > $R/gpurun_out/${TAG}_${V}_groups.txt)
done
