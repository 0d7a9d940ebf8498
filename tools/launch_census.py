"""Kernel-launch census of one non-regularised training step, phase by phase (torch.profiler, eager mode).

Prints, per phase, the launch count, the summed device time, and the top kernels -- used to find where the
small-launch tail of the step comes from."""
import sys; sys.path.insert(0, '.')
import collections
import torch
from torch.profiler import profile, ProfilerActivity
from textboxgan_amd import ops
from textboxgan_amd.config import Config
from textboxgan_amd.models import mask_text_box
from textboxgan_amd.optim import write_grads
from textboxgan_amd.training_step import build_trainer_state, generator_loss, discriminator_loss
from bench import synthetic_batch, bench_init_

dev = torch.device('cuda:0')
cfg = Config(batch_size_per_gpu=16)
st = build_trainer_state(cfg, dev, seed=0, use_graphs=False); bench_init_(st)
b = synthetic_batch(cfg, dev, 1234); ts = st["training_step"]

args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
for _ in range(2): ts.dist_train_step(*args)
torch.cuda.synchronize()

G, D = ts.generator, ts.discriminator
TOP = int(sys.argv[1]) if len(sys.argv) > 1 else 12
state = {}


def phase(name, fn):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn(); torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            agg[e.name][0] += 1; agg[e.name][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    n = sum(v[0] for v in agg.values()); t = sum(v[1] for v in agg.values())
    small = sum(v[0] for v in agg.values() if v[1] / v[0] < 15.0)
    tsmall = sum(v[1] for v in agg.values() if v[1] / v[0] < 15.0)
    print(f"=== {name}: {n} launches, {t/1e3:.2f} ms device; kernels averaging <15us: {small} launches, {tsmall/1e3:.2f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:TOP]:
        print(f"   {v[0]:5d} x {v[1]/v[0]:7.1f} us = {v[1]/1e3:6.2f} ms  {k[:110]}")


def g_fwd():
    z = torch.randn(16, cfg.z_dim, device=dev)
    fake = G((b["input_words"], z), training=True, rand={})
    state["fake"] = mask_text_box(fake, b["input_words"], cfg.char_width)


def d_fwd_fake():
    state["fs"] = D(state["fake"])
    state["g_loss"] = generator_loss(state["fs"], ts.batch_size)


def d_fwd_real():
    state["rs"] = D(b["real_images"])
    state["d_loss"] = discriminator_loss(state["fs"], state["rs"], ts.batch_size)


def ocr_fwd():
    state["ocr"] = 1e-4 * ts._get_ocr_loss(state["fake"], b["ocr_labels"], b["ocr_images"])


def g_bwd():
    ops.FLAGS.skip_d_wgrad = True
    grads = torch.autograd.grad(state["g_loss"], ts.g_params, retain_graph=True, allow_unused=True)
    ops.FLAGS.skip_d_wgrad = False
    write_grads(ts.g_views, grads)


def o_bwd():
    grads = torch.autograd.grad(state["ocr"], ts.o_params, retain_graph=True, allow_unused=True)
    write_grads(ts.o_views, grads)


def d_bwd():
    ops.FLAGS.skip_image_grad = True
    grads = torch.autograd.grad(state["d_loss"], ts.d_params, allow_unused=True)
    ops.FLAGS.skip_image_grad = False
    write_grads(ts.d_views, grads)


def updates():
    ts.exchange.reduce_now((ts.g_grad, ts.o_grad, ts.d_grad))
    ts._apply_updates()
    ts.g_clone.set_as_moving_average_of(G) if hasattr(ts, "g_clone") else None


for name, fn in (("G forward + mask", g_fwd), ("D forward (fake) + g_loss", d_fwd_fake), ("D forward (real) + d_loss", d_fwd_real),
                 ("OCR forward + loss", ocr_fwd), ("g-pass backward", g_bwd), ("ocr-pass backward", o_bwd),
                 ("d-pass backward", d_bwd), ("all-reduce + Adam", updates)):
    phase(name, fn)
