"""N graph-replayed plain steps (f32x3, B=16) for `rocprofv3 --kernel-trace --stats`: python tools/trace_graph_step.py <USE_UNITS 0|1> [steps] [plain|pl|r1]
(per-kernel time UNDER GRAPH REPLAY -- the eager roofline pass does not see cache / clock effects between kernels)
environment: TBG_TUNING="attr=value,..." (ops.TUNING attributes), TBG_DTYPE (f32x3 | bf16 | f32), TBG_BATCH"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
ops.TUNING.use_units = bool(int(sys.argv[1]))
import os
for kv in filter(None, os.environ.get("TBG_TUNING", "").split(",")):  # e.g. TBG_TUNING="unit_sinks=False,fuse_skip_grad=False"
    k, v = kv.split("=")
    assert hasattr(ops.TUNING, k), k
    setattr(ops.TUNING, k, eval(v))
DTYPE = os.environ.get("TBG_DTYPE", "f32x3")
BATCH = int(os.environ.get("TBG_BATCH", "16"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=BATCH)
b = synthetic_batch(cfg, dev, 1234)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=True, compute_dtype=DTYPE); bench_init_(st)
ts = st["training_step"]
REG = sys.argv[3] if len(sys.argv) > 3 else "plain"
args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], REG == "r1", REG in ("pl", "r1"), 1e-4)
for _ in range(4 + n): ts.dist_train_step(*args)
torch.cuda.synchronize()
