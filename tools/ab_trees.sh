#!/bin/bash
# A/B of two source TREES on the same box: graph-replayed plain step, alternating child processes (each tree imports its own package)
# usage (GPU box, repo root): tools/ab_trees.sh <treeA dir> <treeB dir> [rounds] [dtype] [batch]
A=$1; B=$2; R=${3:-3}; DT=${4:-f32x3}; BS=${5:-16}
cat > /tmp/ab_child.py <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dt, bs = sys.argv[1], int(sys.argv[2])
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=bs)
b = synthetic_batch(cfg, dev, 1234)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=True, compute_dtype=dt); bench_init_(st)
ts = st["training_step"]
args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
for _ in range(4): ts.dist_train_step(*args)
res = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(32): ts.dist_train_step(*args)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 32 * 1e3)
print("MS %.3f" % min(res))
PY
for r in $(seq $R); do for T in $A $B; do echo -n "$T $DT B=$BS: "; (cd $T && python /tmp/ab_child.py $DT $BS 2>/dev/null | grep MS); done; done
