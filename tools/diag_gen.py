import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from oracle import ref_model as M
from textboxgan_amd.config import small_config
from textboxgan_amd.models import Generator
from test_layers_gpu import to64, rnd, rel_err, l2_err, _load
dev = torch.device('cuda:0')
cfg = small_config(4)
for mode in ("fused", "composable"):
  for training in (False,):
    P = to64(M.init_generator(cfg, seed=5, bench_init=True))
    names = [k for k in P if k not in M.NON_TRAINABLE]
    for n in names: P[n].requires_grad_(True)
    batch = M.make_batch(cfg); rand = M.make_rand(cfg, seed=7)
    rand64 = {k: ([t.double() for t in v] if isinstance(v, list) else (v.double() if torch.is_tensor(v) else v)) for k, v in rand.items()}
    G = _load(Generator(cfg), P, dev)
    img = M.generator(P, cfg, batch["input_words"], rand64["z"], rand64, training=training)
    gi = rnd(*img.shape, seed=23)
    grads = torch.autograd.grad(img, [P[n] for n in names], gi, allow_unused=True)
    randd = {k: ([t.to(dev) for t in v] if isinstance(v, list) else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in rand.items()}
    imgd = G((batch["input_words"].to(dev), randd["z"]), training=training, rand=randd, mode=mode)
    pd = dict(G.named_parameters())
    gd = torch.autograd.grad(imgd, [pd[n] for n in names], gi.float().to(dev), allow_unused=True)
    print(mode, training, 'img', rel_err(imgd, img))
    for n, a, b in zip(names, gd, grads):
        if b is None: continue
        l2, mx = l2_err(a, b), rel_err(a, b)
        if l2 > 5e-4 or mx > 5e-3: print('   %-60s l2=%.2e max=%.2e' % (n, l2, mx))
