"""Generate the golden fixtures under tests/golden/ (small .npz files).

The reference (TensorFlow 2.8) cannot be imported in the build container and ships no tests or
vectors, so these fixtures are produced by the float64 definition-level oracle
(oracle/ref_ops.py, oracle/ref_model.py) -- they pin the ORACLE against regressions and give the
GPU tests data that does not depend on the oracle code being importable/unchanged.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_model as M, ref_ops as R  # noqa: E402
from textboxgan_amd.config import small_config  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.random.default_rng(2024)
    # --- upfirdn2d: the five (pad, factor) combinations the model uses + one asymmetric-filter case
    cases = {}
    k = R.setup_kernel([1, 3, 3, 1]).astype(np.float64)
    specs = dict(blur_up=(1, 1, 1, 1, (1, 1, 1, 1), 4.0), blur_down3=(1, 1, 1, 1, (2, 3, 2, 3), 1.0),
                 blur_skip=(1, 1, 1, 1, (1, 2, 1, 2), 1.0), rgb_up=(2, 2, 1, 1, (2, 1, 2, 1), 4.0),
                 skip_dec=(1, 1, 2, 2, (1, 2, 1, 2), 1.0), skip_dec_w=(1, 1, 2, 1, (1, 2, 1, 2), 1.0))
    for name, (ux, uy, dx, dy, pad, gain) in specs.items():
        x = g.standard_normal((3, 9, 14, 1))
        kk = k * gain
        y = R.np_upfirdn2d_cu(x, kk, ux, uy, dx, dy, pad[0], pad[1], pad[2], pad[3])
        cases[f"{name}_x"] = x.astype(np.float32)
        cases[f"{name}_k"] = kk.astype(np.float32)
        cases[f"{name}_y"] = y.astype(np.float32)
        cases[f"{name}_p"] = np.array([ux, uy, dx, dy, *pad], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "upfirdn2d.npz"), **cases)

    # --- modulated conv (3x3, demod) and its up-sampling form, float64 definition
    B, I, O, H, W, sd = 2, 6, 8, 5, 7, 4
    x = g.standard_normal((B, I, H, W)); style = g.standard_normal((B, sd))
    w = g.standard_normal((3, 3, I, O)); mw = g.standard_normal((sd, I)); mb = g.standard_normal(I) * 0.1
    t = lambda a: torch.from_numpy(a)
    y = R.t_modulated_conv2d(t(x), t(style), t(w), t(mw), t(mb), up=False, demodulate=True, fused=True)
    yu = R.t_modulated_conv2d(t(x), t(style), t(w), t(mw), t(mb), up=True, demodulate=True, fused=True)
    np.savez_compressed(os.path.join(HERE, "modconv.npz"), x=x.astype(np.float32), style=style.astype(np.float32),
                        w=w.astype(np.float32), mod_w=mw.astype(np.float32), mod_b=mb.astype(np.float32),
                        y=y.numpy().astype(np.float32), y_up=yu.numpy().astype(np.float32))

    # --- whole networks at the reduced-channel config: generator image + discriminator scores
    cfg = small_config(2)
    G = {k_: v.double() for k_, v in M.init_generator(cfg, seed=11, bench_init=True).items()}
    D = {k_: v.double() for k_, v in M.init_discriminator(cfg, seed=12, bench_init=True).items()}
    from textboxgan_amd.char_tokens import string_to_main_int_sequence
    words = torch.from_numpy(string_to_main_int_sequence(["Hello", "GAN-2024"]))
    rand = M.make_rand(cfg, seed=5, with_pl=False)
    rand = {k_: ([t_.double() for t_ in v] if isinstance(v, list) else (v.double() if torch.is_tensor(v) else v))
            for k_, v in rand.items()}
    img = M.generator(G, cfg, words, rand["z"], rand, training=False)
    img_m = R.t_mask_text_box(img, words, cfg.char_width)
    scores = M.discriminator(D, cfg, img_m)
    np.savez_compressed(os.path.join(HERE, "networks_small.npz"), words=words.numpy(), z=rand["z"].numpy().astype(np.float32),
                        **{f"noise{i}": n.numpy().astype(np.float32) for i, n in enumerate(rand["noises"])},
                        image_checksum=np.array([float(img.sum()), float(img.abs().sum()), float(img.square().sum())]),
                        image_row=img[:, :, 31, :].numpy().astype(np.float32), scores=scores.numpy().astype(np.float32))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
