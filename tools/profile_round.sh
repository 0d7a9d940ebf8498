#!/bin/bash
# Round profile: (1) the default bench line, (2) rocprofv3 kernel stats of the same bench command,
# (3) rocprofv3 kernel stats of the roofline pass alone (same launches as roofline.avg_launch_us).
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
(cd $R && python bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err); tail -c 600 $R/gpurun_out/${TAG}_bench.json
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_bench -o b -- python bench.py --no-cpu-baseline --no-roofline --no-sub-records > /tmp/prof_${TAG}_bench.log 2>&1)
(cd $R && python tools/prof_summary.py /tmp/prof_${TAG}_bench 45 > $R/gpurun_out/${TAG}_bench_kernel_stats.txt)
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_roof -o r -- python bench.py --roofline-only --steps 4 --warmup 2 > $R/gpurun_out/${TAG}_roofline.json 2> /tmp/prof_${TAG}_roof.log)
(cd $R && python tools/prof_summary.py /tmp/prof_${TAG}_roof 30 > $R/gpurun_out/${TAG}_roofline_kernel_stats.txt)
# the same pass by kernel family, per step (6 eager non-regularised steps: 2 warm-up + 4)
(cd $R && python tools/prof_groups.py /tmp/prof_${TAG}_roof 6 > $R/gpurun_out/${TAG}_roofline_groups.txt)
head -8 $R/gpurun_out/${TAG}_roofline_kernel_stats.txt | cut -c1-170
