"""Hyper-parameters of the TextBoxGAN training step, restated as a plain object.

Mirrors the values of the reference's global EasyDict (reference
``config/config.py:40-149``) without its import-time side effects (no
MirroredStrategy, no Keras tokenizers).  ``cfg`` is the module-level default
instance; ``Config(...)`` builds variants (tests use reduced channel counts).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


@dataclass
class OptParams:
    learning_rate: float = 0.002
    beta1: float = 0.0
    beta2: float = 0.99
    epsilon: float = 1e-08
    reg_interval: int = 8

    def lazy_reg_rescaled(self) -> "OptParams":
        """reference ``train.py:110-129`` (``update_optimizer_params``)."""
        mb_ratio = self.reg_interval / (self.reg_interval + 1)
        return OptParams(
            learning_rate=self.learning_rate * mb_ratio,
            beta1=self.beta1 ** mb_ratio,
            beta2=self.beta2 ** mb_ratio,
            epsilon=self.epsilon,
            reg_interval=self.reg_interval,
        )


@dataclass
class Config:
    # Text boxes specs (config.py:40-42)
    char_height: int = 64
    char_width: int = 32
    max_char_number: int = 8
    # Model (config.py:45-77)
    embedding_out_dim: int = 32
    word_encoder_dense_dim: int = 256
    generator_resolutions: List[Tuple[int, int]] = field(
        default_factory=lambda: [(2, 8), (4, 16), (8, 32), (16, 64), (32, 128), (64, 256)]
    )
    generator_feat_maps: List[Optional[int]] = field(
        default_factory=lambda: [None, 512, 256, 256, 128, 128]
    )
    discrim_resolutions: List[Tuple[int, int]] = field(
        default_factory=lambda: [(64, 256), (32, 128), (16, 64), (8, 32), (8, 16), (4, 8), (4, 4)]
    )
    discrim_feat_maps: List[int] = field(
        default_factory=lambda: [64, 128, 128, 256, 256, 512, 512]
    )
    z_dim: int = 512
    style_dim: int = 512
    n_mapping: int = 5
    # Optimizers (config.py:80-94)
    g_opt: OptParams = field(default_factory=lambda: OptParams(reg_interval=8))
    d_opt: OptParams = field(default_factory=lambda: OptParams(reg_interval=16))
    # batch (config.py:104,141) -- batch_size = per_gpu * replicas
    batch_size_per_gpu: int = 4
    num_replicas: int = 1
    # OCR (config.py:108-110,124)
    ocr_loss_weight: float = 0.0001
    ocr_loss_type: str = "softmax_crossentropy"
    aster_image_dims: Tuple[int, int] = (64, 256)
    max_steps: int = 130000
    # vocabulary sizes (config/char_tokens.py:4-17): 69 main chars (+ pad 0), 94 aster chars
    main_vocab: int = 69

    def __post_init__(self):
        self.generator_feat_maps = list(self.generator_feat_maps)
        # config.py:128-136: word encoder output [B, C0, h0, w0] must hold dense_dim*max_chars values
        h0, w0 = self.generator_resolutions[0]
        self.generator_feat_maps[0] = int(
            self.word_encoder_dense_dim * self.max_char_number / (h0 * w0)
        )
        assert self.ocr_loss_type in ("softmax_crossentropy", "mse")
        # config.py:145-149
        assert (
            tuple(self.generator_resolutions[-1])
            == tuple(self.discrim_resolutions[0])
            == (self.char_height, self.image_width)
        ), "G/D resolutions must match (char_height, image_width)"

    @property
    def image_width(self) -> int:
        return self.char_width * self.max_char_number

    @property
    def batch_size(self) -> int:
        return self.batch_size_per_gpu * self.num_replicas

    def replace(self, **kw) -> "Config":
        c = copy.deepcopy(self)
        for k, v in kw.items():
            setattr(c, k, v)
        c.__post_init__()
        return c


cfg = Config()


def small_config(batch_size_per_gpu: int = 4, num_replicas: int = 1) -> Config:
    """Reduced-channel variant with the same topology/geometry; used by parity tests so
    the CPU oracle finishes in seconds."""
    return Config(
        word_encoder_dense_dim=32,  # -> C0 = 32*8/16 = 16
        generator_feat_maps=[None, 32, 16, 16, 8, 8],
        discrim_feat_maps=[8, 8, 16, 16, 32, 32, 32],
        z_dim=32,
        style_dim=32,
        n_mapping=2,
        batch_size_per_gpu=batch_size_per_gpu,
        num_replicas=num_replicas,
    )
