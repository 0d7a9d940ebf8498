"""Does the frozen-OCR branch (forked onto a second HIP stream) actually overlap the main stream?  Times the bench loop
(16-step lazy-reg cycle) with overlap_ocr on / off, HIP graphs and eager.  GPU box: python tools/exp_ocr_overlap.py"""
import sys, time; sys.path.insert(0, '.')
import torch
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_, run_steps
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
for graphs in (True, False):
    for overlap in (True, False):
        st = build_trainer_state(cfg, dev, seed=0, use_graphs=graphs); bench_init_(st)
        ts = st["training_step"]; ts.overlap_ocr = overlap
        b = synthetic_batch(cfg, dev, 1234)
        if graphs:
            ts.prepare_graphs(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"])
        run_steps(st, b, 3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run_steps(st, b, 16)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16 * 1e3
        print(f"graphs={graphs} overlap_ocr={overlap}: {dt:.2f} ms/step")
        del st, ts
        torch.cuda.empty_cache()
