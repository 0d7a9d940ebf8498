"""ms per step of the three captured step variants (plain / +PL / +PL+R1), graph replay."""
import sys; sys.path.insert(0, '.')
import torch, time
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=True); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]
ts.prepare_graphs(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"])
for name, r1, pl in (("plain", False, False), ("+PL", False, True), ("+PL+R1", True, True)):
    a = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], r1, pl, 1e-4)
    for _ in range(2): ts.dist_train_step(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): ts.dist_train_step(*a)
    torch.cuda.synchronize(); print(f"{name:8s} {(time.perf_counter() - t0) / 8 * 1e3:7.2f} ms/step")
