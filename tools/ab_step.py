"""A/B of a host-side knob on the SAME box: graph-replayed plain step (f32x3, B=16), alternating settings.
usage: python tools/ab_step.py <attribute of ops.TUNING | module.attribute> <value A> <value B> [rounds] [dtype] [batch] [plain|pl|r1]
(pl = a path-length step, r1 = a path-length + R1 step)"""
import sys; sys.path.insert(0, '.')
import torch, time
from textboxgan_amd import ops
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
dev = torch.device('cuda:0')
attr, va, vb = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3])
mod = ops.TUNING
if "." in attr:
    import importlib
    mname, attr = attr.split(".")
    mod = importlib.import_module("textboxgan_amd." + mname)
REG = sys.argv[7] if len(sys.argv) > 7 else "plain"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32x3"
cfg = Config(batch_size_per_gpu=int(sys.argv[6]) if len(sys.argv) > 6 else 16)
b = synthetic_batch(cfg, dev, 1234)
res = {repr(va): [], repr(vb): []}
for r in range(rounds):
    for v in (va, vb):
        setattr(mod, attr, v)
        st = build_trainer_state(cfg, dev, seed=0, use_graphs=True, compute_dtype=dtype); bench_init_(st)
        ts = st["training_step"]
        args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], REG == "r1", REG in ("pl", "r1"), 1e-4)
        for _ in range(4): ts.dist_train_step(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(32): ts.dist_train_step(*args)
        torch.cuda.synchronize()
        res[repr(v)].append((time.perf_counter() - t0) / 32 * 1e3)
        del st, ts
        torch.cuda.empty_cache()
for k, v in res.items():
    print(f"{attr}={k}: " + " ".join(f"{x:.3f}" for x in v) + f"  mean {sum(v) / len(v):.3f} ms/step ({REG} step, {dtype}, graph replay)", flush=True)
