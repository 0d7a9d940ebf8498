"""Character tokenisers for the generator (main) and the OCR (ASTER) vocabularies.

Restates the behaviour of the Keras ``Tokenizer(char_level=True, lower=False,
oov_token="<OOV>")`` objects the reference builds in ``config/char_tokens.py:12-17``
and of ``utils/utils.py:66-105`` (``string_to_main_int_sequence`` /
``string_to_aster_int_sequence``) as plain dict look-ups:

* Keras gives ``<OOV>`` index 1 and the fitted characters 2.. in frequency order,
  ties broken by first appearance -- every char appears once, so index = position + 2.
* main sequence: pad with 1 ("post"), then subtract 1  -> pad/OOV = 0, chars 1..69.
* aster sequence: pad with 1 ("post")                  -> pad/OOV(EOS) = 1, chars 2..95.
* ``pad_sequences`` default ``truncating="pre"`` keeps the LAST ``maxlen`` tokens.
"""
from __future__ import annotations

from typing import List

import numpy as np

MAIN_CHAR_VECTOR = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ-'.!?,\""
ASTER_CHAR_VECTOR = (
    "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
)


class _CharTokenizer:
    def __init__(self, chars: str):
        self.word_index = {"<OOV>": 1}
        for pos, ch in enumerate(chars):
            self.word_index[ch] = pos + 2

    def texts_to_sequences(self, words: List[str]) -> List[List[int]]:
        return [[self.word_index.get(ch, 1) for ch in w] for w in words]


class CharTokenizer:
    def __init__(self):
        self.main = _CharTokenizer(MAIN_CHAR_VECTOR)
        self.aster = _CharTokenizer(ASTER_CHAR_VECTOR)


char_tokenizer = CharTokenizer()


def _pad_sequences(seqs: List[List[int]], maxlen: int, value: int) -> np.ndarray:
    out = np.full((len(seqs), maxlen), value, dtype=np.int32)
    for r, s in enumerate(seqs):
        s = s[-maxlen:]  # Keras default truncating="pre"
        out[r, : len(s)] = s
    return out


def string_to_main_int_sequence(words: List[str], max_char_number: int = 8) -> np.ndarray:
    seq = char_tokenizer.main.texts_to_sequences(words)
    return _pad_sequences(seq, max_char_number, 1) - 1


def string_to_aster_int_sequence(words: List[str], max_char_number: int = 8) -> np.ndarray:
    seq = char_tokenizer.aster.texts_to_sequences(words)
    return _pad_sequences(seq, max_char_number, 1)


def main_to_aster_labels(input_words: np.ndarray) -> np.ndarray:
    """Re-encode main-vocabulary ids (0 pad, 1..69) as ASTER labels (1 pad/EOS, 2..95)."""
    table = np.ones(len(MAIN_CHAR_VECTOR) + 1, dtype=np.int32)
    for pos, ch in enumerate(MAIN_CHAR_VECTOR):
        table[pos + 1] = char_tokenizer.aster.word_index.get(ch, 1)
    return table[np.asarray(input_words)]
