"""BASELINE configs[4]: projector/projector.py's LPIPS + OCR latent optimisation at its stated size -- 1000 steps, one
[1, 512] latent, full channel widths, 1 MI355X (synthetic VGG / LPIPS / OCR weights: the real ones are absent external
downloads; the arithmetic and the schedule are the reference's).  Prints one JSON line.
usage (GPU box): python tools/bench_projector.py [f32x3|f32] [steps]"""
import json, math, sys, time; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
from textboxgan_amd.aster import AsterInferer, AsterLikeOCRHip
from textboxgan_amd.config import Config
from textboxgan_amd.models import Generator
from textboxgan_amd.projector import Projector
arith = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
cfg = Config(batch_size_per_gpu=4)
torch.manual_seed(0)
with ops.compute_dtype(arith):
    gen = Generator(cfg).to(dev)
    for p in gen.parameters():
        p.requires_grad_(False)
    proj = Projector("Hello", gen, AsterInferer(model=AsterLikeOCRHip()).to(dev), cfg, device=dev)
    target = torch.randint(0, 256, (1, 64, 32 * 5, 3)).float()
    proj.main(target, num_steps=5)  # warm-up (filter packs, allocator)
    proj._m = proj._v = None; proj._t = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w, saved, losses = proj.main(target, num_steps=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
ls = [float(l) for l in losses]
print(json.dumps({"workload": "BASELINE configs[4]: projector LPIPS + 0.1 * OCR-CE latent optimisation, text 'Hello' (64x160 target), "
                  f"{steps} steps incl. the 10000-latent mean-style pass, 1 MI355X, {arith} arithmetic, synthetic VGG/LPIPS/OCR weights",
                  "steps": steps, "seconds": round(dt, 3), "steps_per_s": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3),
                  "loss_first": round(ls[0], 4), "loss_last": round(ls[-1], 4), "loss_min": round(min(ls), 4),
                  "all_finite": all(math.isfinite(v) for v in ls), "saved_latents": len(saved)}))
