"""Per-shape time / TFLOP/s of every conv launch in one non-regularised training step (B=16)."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
st = build_trainer_state(cfg, dev, seed=0); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]
args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
for _ in range(2): ts.dist_train_step(*args)
ops.PROFILE.by_shape = True; ops.PROFILE.enable()
N = 3
for _ in range(N): ts.dist_train_step(*args)
recs = ops.PROFILE.collect()
tot = sum(r["ms"] for r in recs.values())
print(f"conv kernels: {tot/N:.2f} ms/step over {sum(r['n'] for r in recs.values())//N} launches/step")
for k, r in sorted(recs.items(), key=lambda kv: -kv[1]["ms"])[:45]:
    print(f"{r['ms']/N:7.3f} ms/step n={r['n']//N:3d} avg={1e3*r['ms']/r['n']:7.1f}us {r['flops']/(r['ms']*1e-3)/1e12:6.1f} TF  {k[:150]}")
