"""Forward-only neighbours of the training step.

* ``ValidationStep``  = reference ``validation_step.py:24-90``: g_clone forward (training=False) ->
  mask -> OCR -> softmax-CE, SUM-reduced over ranks.
* ``generate_chosen_words`` = the tensor part of ``Infer.genererate_chosen_words``
  (``infer.py:37-104``): tokenise, ONE z shared by all words, generator(training=False, psi=1.0),
  uint8 conversion (``utils/utils.py:48-63``) and crop to ``char_width * len(word)``.
  Writing PNG files (cv2.imwrite) is the CLI's job and stays out of scope.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .char_tokens import string_to_main_int_sequence
from .config import Config, cfg as default_cfg
from .dist_utils import GradExchange
from .models import Generator, generator_output_to_uint8, mask_text_box
from .training_step import softmax_cross_entropy_loss


class ValidationStep:
    def __init__(self, generator: Generator, aster_ocr, cfg: Config = default_cfg, process_group=None):
        self.generator, self.aster_ocr, self.cfg = generator, aster_ocr, cfg
        self.exchange = GradExchange(process_group)

    @torch.no_grad()
    def dist_validation_step(self, input_words: torch.Tensor, ocr_labels: torch.Tensor,
                             z: Optional[torch.Tensor] = None, rand: Optional[dict] = None) -> torch.Tensor:
        cfg = self.cfg
        B = input_words.shape[0]
        if z is None:
            z = torch.randn(B, cfg.z_dim, device=input_words.device)
        fake = self.generator((input_words, z), batch_size=B, training=False, rand=rand)
        fake = mask_text_box(fake, input_words, cfg.char_width)
        logits = self.aster_ocr(self.aster_ocr.convert_inputs(fake, ocr_labels, blank_label=1))
        loss = softmax_cross_entropy_loss(logits, ocr_labels, cfg.batch_size)
        return self.exchange.reduce_scalars([loss])[0]


@torch.no_grad()
def generate_chosen_words(generator: Generator, words: List[str], cfg: Config = default_cfg,
                          z: Optional[torch.Tensor] = None, truncation_psi: float = 1.0,
                          w_latents: Optional[torch.Tensor] = None) -> List[np.ndarray]:
    """-> one uint8 HxWx3 array per word, width = char_width * len(word) (infer.py:37-100).
    w_latents [1, style_dim]: a style vector (e.g. from the Projector) instead of a random z (infer.py:60-71)."""
    device = next(generator.parameters()).device
    tokens = torch.from_numpy(string_to_main_int_sequence(words, cfg.max_char_number)).to(device)
    if w_latents is not None:
        with torch.no_grad():
            we = generator.word_encoder(tokens, batch_size=len(words))
            styles = w_latents.to(device).reshape(1, 1, -1).expand(len(words), generator.n_style, -1).contiguous()
            img = generator.synthesis(we, styles)
        u8 = generator_output_to_uint8(img).cpu().numpy()
        return [u8[i, :, : cfg.char_width * min(len(w), cfg.max_char_number)] for i, w in enumerate(words)]
    if z is None:
        z = torch.randn(1, cfg.z_dim, device=device)
    z = z.expand(len(words), -1).contiguous()  # the same style for every word (infer.py:72-76)
    img = generator((tokens, z), batch_size=len(words), training=False, truncation_psi=truncation_psi)
    u8 = generator_output_to_uint8(img).cpu().numpy()
    return [u8[i, :, : cfg.char_width * min(len(w), cfg.max_char_number)] for i, w in enumerate(words)]
