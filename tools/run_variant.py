"""Run N eager steps of one variant (for rocprofv3): usage run_variant.py {plain|pl|r1} [N]"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
v = sys.argv[1] if len(sys.argv) > 1 else "pl"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = Config(batch_size_per_gpu=16)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=False); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]
a = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], v == "r1", v in ("pl", "r1"), 1e-4)
for _ in range(n): ts.dist_train_step(*a)
torch.cuda.synchronize()
