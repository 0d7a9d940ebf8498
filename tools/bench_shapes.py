"""Per-shape time / TFLOP/s of every conv launch in one non-regularised training step (B=16)."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
DTYPE = sys.argv[1] if len(sys.argv) > 1 else "f32"
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if DTYPE == "bf16" else 16)
cfg = Config(batch_size_per_gpu=BATCH)
st = build_trainer_state(cfg, dev, seed=0, compute_dtype=DTYPE); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]
args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
for _ in range(2): ts.dist_train_step(*args)
ops.PROFILE.by_shape = True; ops.PROFILE.enable()
N = 3
for _ in range(N): ts.dist_train_step(*args)
recs = ops.PROFILE.collect()
tot = sum(r["ms"] for r in recs.values())
print(f"conv kernels: {tot/N:.2f} ms/step over {sum(r['n'] for r in recs.values())//N} launches/step")
for k, r in sorted(recs.items(), key=lambda kv: -kv[1]["ms"])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{r['ms']/N:7.3f} ms/step n={r['n']//N:3d} avg={1e3*r['ms']/r['n']:7.1f}us {r['flops']/(r['ms']*1e-3)/1e12:6.1f} TF  {k[:150]}")

import re, collections
cat = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k, r in recs.items():
    m = re.search(r"in=(\d+)x(\d+)", k) or re.search(r"S=(\d+)x(\d+)", k)
    if m is None:  # HBM-bound records (FIR, bias_act, adam): not convolutions
        continue
    h, w_ = int(m.group(1)), int(m.group(2))
    ocr = (w_ in (25, 50, 100) or (h, w_) in ((32, 64), (16, 32), (8, 16), (4, 8), (2, 4), (1, 2))) and f"B={BATCH}" in k and not ("C=512 M=512 in=4x8" in k or "C=256 M=256 in=8x16" in k)
    if "wgrad" in k:
        c = "wgrad 1x1" if "k=1 " in k else ("wgrad strided" if "s=(1, 1)" not in k else "wgrad 3x3 s1")
    elif ocr: c = "OCR convs"
    elif "k=1x1" in k: c = "1x1 (RGB/skip)"
    elif "T=1" in k: c = "transposed (classes)"
    elif "s=(1, 1)" not in k: c = "strided fprop"
    else: c = "3x3 s1 fprop/bwd-data"
    cat[c][0] += r["ms"] / N; cat[c][1] += r["flops"] / N; cat[c][2] += r["n"] // N
print("--- categories (ms/step, GFLOP/step, TF, launches)")
for c, (ms, fl, n) in sorted(cat.items(), key=lambda kv: -kv[1][0]):
    print(f"{c:28s} {ms:7.2f} ms {fl/1e9:8.1f} GF {fl/(ms*1e-3)/1e12:6.1f} TF  n={n}")
