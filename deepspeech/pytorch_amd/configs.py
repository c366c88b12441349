"""Constructor-argument dataclasses mirroring the reference's (``deepspeech_pytorch/configs/train_config.py:16-73``,
``enums.py:17-21``) so that the drop-in class can be built without Hydra/OmegaConf.  The class accepts the
reference's own config objects as well: only the attribute names below are read (duck typing), and
bidirectionality is decided by the config type NAME exactly like reference model.py:152.
"""
from dataclasses import dataclass
from enum import Enum

import torch.nn as nn


class RNNType(Enum):          # enums.py:17-21 -- the enum VALUE is the torch class, as in the reference
    lstm = nn.LSTM
    rnn = nn.RNN
    gru = nn.GRU


def rnn_kind(rnn_type) -> str:
    """Normalises reference RNNType members / torch classes / strings to 'gru' | 'lstm' | 'rnn'."""
    v = getattr(rnn_type, "value", rnn_type)
    if isinstance(v, str):
        name = v.lower()
    else:
        name = getattr(v, "__name__", str(v)).lower()
    if name in ("gru", "lstm", "rnn"):
        return name
    raise ValueError("unsupported rnn_type %r (expected GRU, LSTM or RNN)" % (rnn_type,))


@dataclass
class SpectConfig:            # train_config.py:16-21
    sample_rate: int = 16000
    window_size: float = .02
    window_stride: float = .01
    window: str = "hamming"


@dataclass
class BiDirectionalConfig:    # train_config.py:46-50
    rnn_type: RNNType = RNNType.lstm
    hidden_size: int = 1024
    hidden_layers: int = 5


@dataclass
class UniDirectionalConfig(BiDirectionalConfig):   # train_config.py:53-55
    lookahead_context: int = 20


@dataclass
class OptimConfig:            # train_config.py:58-62
    learning_rate: float = 1.5e-4
    learning_anneal: float = 0.99
    weight_decay: float = 1e-5


@dataclass
class SGDConfig(OptimConfig):  # train_config.py:65-67
    momentum: float = 0.9


@dataclass
class AdamConfig(OptimConfig):  # train_config.py:70-73
    eps: float = 1e-8
    betas: tuple = (0.9, 0.999)


LABELS = ["_", "'"] + [chr(ord("A") + i) for i in range(26)] + [" "]   # reference labels.json (29 symbols, blank first)
