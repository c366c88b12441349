"""Reference harness: import /root/reference/deepspeech_pytorch/model.py UNMODIFIED.

Test infrastructure only (never imported by the product path, never used on the
GPU box: /root/reference does not exist there).  The image lacks
pytorch_lightning / omegaconf / torchmetrics / Levenshtein, so four minimal stub
modules are injected into ``sys.modules`` before the reference is imported
(SURVEY.md Appendix C lists exactly what each stub must provide and which
reference lines use it).
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("DS2_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "deepspeech_pytorch", "model.py"))


def _install_stubs():
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):  # model.py:138
            def save_hyperparameters(self, *a, **k):  # model.py:147: Lightning records the constructor arguments of the caller
                import inspect
                frame = inspect.currentframe().f_back
                try:
                    names = [n for n in inspect.signature(type(self).__init__).parameters if n != "self"]
                    self._hparams = {n: frame.f_locals[n] for n in names if n in frame.f_locals}
                finally:
                    del frame

            @property
            def hparams(self):  # written into a checkpoint as "hyper_parameters"
                return getattr(self, "_hparams", {})

            @classmethod
            def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):  # utils.py:31
                ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
                hp = dict(ckpt.get("hyper_parameters", {}))
                hp.update(overrides)
                model = cls(**hp)
                model.load_state_dict(ckpt["state_dict"], strict=strict)
                return model

            def log(self, *a, **k):  # model.py:270-271
                pass

            @property
            def device(self):  # model.py:254
                return next(self.parameters()).device

        class LightningDataModule:  # loader/data_module.py:9
            pass

        def seed_everything(seed):  # training.py:14
            torch.manual_seed(seed)

        pl.LightningModule = LightningModule
        pl.LightningDataModule = LightningDataModule
        pl.seed_everything = seed_everything
        sys.modules["pytorch_lightning"] = pl

    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")

        class OmegaConf:  # model.py:152,274,282
            @staticmethod
            def get_type(cfg):
                return type(cfg)

        oc.OmegaConf = OmegaConf
        oc.MISSING = "???"  # configs/train_config.py:4
        sys.modules["omegaconf"] = oc

    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")

        class Metric(nn.Module):  # validation.py:13-18,63-64
            def __init__(self, dist_sync_on_step=False):
                super().__init__()

            def add_state(self, name, default, dist_reduce_fx=None):
                setattr(self, name, default)

            def forward(self, *a, **k):  # model.py:257-268: `self.wer(preds=...)` accumulates through update()
                self.update(*a, **k)

        tm.Metric = Metric
        sys.modules["torchmetrics"] = tm

    if "Levenshtein" not in sys.modules:
        lv = types.ModuleType("Levenshtein")

        def distance(a, b):  # validation.py:84,132
            prev = list(range(len(b) + 1))
            for i, ca in enumerate(a, 1):
                cur = [i]
                for j, cb in enumerate(b, 1):
                    cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
                prev = cur
            return prev[-1]

        lv.distance = distance
        sys.modules["Levenshtein"] = lv


def load_reference():
    """Returns a namespace with the reference's DeepSpeech class, config dataclasses, RNNType and labels."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import json
    from deepspeech_pytorch.model import DeepSpeech  # noqa
    from deepspeech_pytorch.configs.train_config import (  # noqa
        BiDirectionalConfig, UniDirectionalConfig, AdamConfig, SGDConfig, SpectConfig)
    from deepspeech_pytorch.enums import RNNType  # noqa
    from deepspeech_pytorch.decoder import GreedyDecoder  # noqa
    with open(os.path.join(REFERENCE_ROOT, "labels.json")) as f:
        labels = json.load(f)
    ns = types.SimpleNamespace(
        DeepSpeech=DeepSpeech, BiDirectionalConfig=BiDirectionalConfig,
        UniDirectionalConfig=UniDirectionalConfig, AdamConfig=AdamConfig, SGDConfig=SGDConfig,
        SpectConfig=SpectConfig, RNNType=RNNType, GreedyDecoder=GreedyDecoder, labels=labels)
    return ns


def extended_labels(n):
    """A label set of n symbols in the reference's labels.json convention (blank '_' first): its 29 symbols, then digits and
    punctuation -- for models with more output classes than the English set (the reference takes any labels file, model.py:154)."""
    base = ["_", "'"] + [chr(ord("A") + i) for i in range(26)] + [" "]
    extra = list("0123456789.,?!-:;()[]{}<>/\\@#$%^&*+=~|abcdefghijklmnopqrstuvwxyz")
    extra += [chr(c) for c in range(0x4E00, 0x4E00 + max(0, n - len(base) - len(extra)))]     # then CJK ideographs (a Mandarin-sized set)
    assert n <= len(base) + len(extra)
    return (base + extra)[:n] if n > len(base) else base[:n]


def build_reference_model(ns, rnn_type="gru", hidden_size=32, hidden_layers=2, bidirectional=True,
                          lookahead_context=20, seed=0, labels=None, sample_rate=16000):
    torch.manual_seed(seed)
    rt = getattr(ns.RNNType, rnn_type)
    if bidirectional:
        mcfg = ns.BiDirectionalConfig(rnn_type=rt, hidden_size=hidden_size, hidden_layers=hidden_layers)
    else:
        mcfg = ns.UniDirectionalConfig(rnn_type=rt, hidden_size=hidden_size, hidden_layers=hidden_layers,
                                       lookahead_context=lookahead_context)
    model = ns.DeepSpeech(labels=labels if labels is not None else ns.labels, model_cfg=mcfg, precision=32, optim_cfg=ns.AdamConfig(),
                          spect_cfg=ns.SpectConfig(sample_rate=sample_rate))
    return model
