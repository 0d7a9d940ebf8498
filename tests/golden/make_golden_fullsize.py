"""Full-size "Hello" fixture (one full G forward at the real channel widths, B = 1):
BASELINE configs[0] -- infer.py --infer_type chosen_words 'Hello' -- evaluated by the FLOAT64 oracle.

Inputs kept in the fixture: the tokens of "Hello", z, the ten noise maps.  Weights are NOT stored (10 M floats):
they are the oracle's seeded initialisation M.init_generator(Config(1), seed=11, bench_init=True), reproduced by the
test.  Outputs: three image rows, three checksums of the unmasked image, the uint8 crop checksum of the masked image
(utils/utils.py:48-63).  Run from the repo root:  python tests/golden/make_golden_fullsize.py   (~1 min on 8 cores)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_model as M, ref_ops as R  # noqa: E402
from textboxgan_amd.char_tokens import string_to_main_int_sequence  # noqa: E402
from textboxgan_amd.config import Config  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cfg = Config(batch_size_per_gpu=1)
    G = {k: v.double() for k, v in M.init_generator(cfg, seed=11, bench_init=True).items()}
    words = torch.from_numpy(string_to_main_int_sequence(["Hello"]))
    rand = M.make_rand(cfg, seed=7, with_pl=False)
    z, noises = rand["z"], rand["noises"]
    r64 = dict(z=z.double(), noises=[n.double() for n in noises])
    img = M.generator(G, cfg, words, r64["z"], r64, training=False)
    masked = R.t_mask_text_box(img, words, cfg.char_width)
    u8 = ((masked.clamp(-1, 1) + 1.0) * 127.5).permute(0, 2, 3, 1).to(torch.uint8)[0, :, : 32 * 5]
    np.savez_compressed(
        os.path.join(HERE, "hello_fullsize.npz"), words=words.numpy(), z=z.numpy(),
        **{f"noise{i}": n.numpy() for i, n in enumerate(noises)},
        image_rows=img[0, :, (5, 31, 60), :].numpy().astype(np.float32),
        image_checksum=np.array([float(img.sum()), float(img.abs().sum()), float(img.square().sum())]),
        image_absmax=np.array(float(img.abs().max())),
        u8_sum=np.array(int(u8.to(torch.int64).sum())), u8_shape=np.array(u8.shape))
    print("hello_fullsize.npz written; |img|max =", float(img.abs().max()), "u8 crop", tuple(u8.shape))


if __name__ == "__main__":
    main()
