"""repeat the gradient computation of ONE plain step from identical state and seeds; report which parameter tensors differ between runs"""
import sys, torch
sys.path.insert(0, '.')
from textboxgan_amd.config import Config
from textboxgan_amd.training_step import build_trainer_state
from bench import synthetic_batch, bench_init_
if len(sys.argv) > 2:
    import os
    from textboxgan_amd import native
    native.LIB_PATH = os.path.abspath(sys.argv[2])
dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
batch = synthetic_batch(cfg, dev, 7)
torch.manual_seed(11)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=False); bench_init_(st)
ts = st["training_step"]
gf, df = st["generator"]._flat, st["discriminator"]._flat
def grads():
    torch.manual_seed(100)
    outs = ts._compute_grads(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], False, False, 1e-4, {}, [])
    torch.cuda.synchronize()
    return ts.g_grad.clone(), ts.o_grad.clone(), ts.d_grad.clone(), [float(v) for grp in outs[:2] for v in grp] + [float(outs[2])]
ref = grads()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(n):
    cur = grads()
    msg = []
    for nm, a, b in (("g", ref[0], cur[0]), ("o", ref[1], cur[1]), ("d", ref[2], cur[2])):
        rel = float((a - b).double().norm() / (a.double().norm() + 1e-30))
        msg.append(f"{nm} {rel:.2e}")
    line = f"run {it}: " + "  ".join(msg)
    big = max(float((ref[k] - cur[k]).double().norm() / (ref[k].double().norm() + 1e-30)) for k in range(3))
    if big > 1e-4:  # locate the tensors
        for nm, flat, views, rng, a, b in (("g", gf, ts.g_views, ts.g_range, ref[0], cur[0]), ("o", gf, ts.o_views, ts.o_range, ref[1], cur[1]),
                                            ("d", df, ts.d_views, (0, df.total), ref[2], cur[2])):
            names = [n_ for n_ in flat.names]
            sel = flat.select(("latent_encoder.", "synthesis.")) if nm == "g" else (flat.select(("synthesis.", "word_encoder.")) if nm == "o" else list(flat.params))
            # walk the views: same order as the parameter list of that set
            off = 0
            base = views[0].storage_offset()
            for v in views:
                o0 = v.storage_offset() - base; o1 = o0 + v.numel()
                da = (a[o0:o1] - b[o0:o1]).double().norm(); na = a[o0:o1].double().norm()
                if float(da) > 1e-3 * float(na) + 1e-12:
                    line += f"\n      {nm}[{o0}:{o1}] shape {tuple(v.shape)} rel {float(da / (na + 1e-30)):.2e}"
    print(line, flush=True)
print("losses ref", ref[3])
