"""Replay the captured plain step N times (for rocprofv3 --kernel-trace): tools/trace_busy.py then gives busy vs span."""
import sys; sys.path.insert(0, '.')
import torch, time
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=True); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]
args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
for _ in range(3): ts.dist_train_step(*args)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ts.dist_train_step(*args)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 100)
