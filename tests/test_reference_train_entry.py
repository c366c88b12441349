"""The reference's OWN training entry, ``deepspeech_pytorch.training.train`` (training.py:13-47), executed unmodified with the
INTEGRATION.md switch applied (``deepspeech_pytorch.model.DeepSpeech = <drop-in class>``): it must construct the drop-in class
from the reference's own config dataclasses and hand it to ``trainer.fit`` -- and what Lightning then asks of a LightningModule
before the first batch (``configure_optimizers``, ``state_dict`` / ``hparams`` for the checkpoint, a strict ``load_state_dict`` of a
checkpoint written by the REFERENCE class) must work.  Build container only (needs /root/reference; no GPU: the first training
step itself is covered on the device by tests/test_gpu_loop.py with the same MiniTrainer).  Lightning, Hydra, the audio loader and
the checkpoint callback are out of scope (SURVEY.md section 8b "callers: unchanged") and absent from the image: they are stubbed at the
module level (golden/ref_harness.py + the stubs below)."""
import os
import sys
import types

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_harness  # noqa: E402
from lightning_stub import MiniTrainer  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="needs the reference tree (build container)")


def _install_loop_stubs(trainer_box):
    ref_harness._install_stubs()
    pl = sys.modules["pytorch_lightning"]
    if not hasattr(pl, "callbacks"):
        cb = types.ModuleType("pytorch_lightning.callbacks")

        class ModelCheckpoint:                      # checkpoint.py:4,9
            def __init__(self, *a, **k):
                self.kwargs = k

        cb.ModelCheckpoint = ModelCheckpoint
        pl.callbacks = cb
        sys.modules["pytorch_lightning.callbacks"] = cb
    if "hydra" not in sys.modules:
        hy = types.ModuleType("hydra")
        hu = types.ModuleType("hydra.utils")
        hu.to_absolute_path = lambda p: p if os.path.isabs(p) else os.path.join(ref_harness.REFERENCE_ROOT, p)   # training.py:16

        def instantiate(config, **kw):              # training.py:42-46: hydra.utils.instantiate(config=cfg.trainer, replace_sampler_ddp=False, callbacks=...)
            t = MiniTrainer(max_epochs=config.max_epochs, precision=config.precision, gradient_clip_val=config.gradient_clip_val,
                            dry_run=True, **kw)
            trainer_box.append(t)
            return t

        hu.instantiate = instantiate
        hy.utils = hu
        sys.modules["hydra"], sys.modules["hydra.utils"] = hy, hu
    else:
        sys.modules["hydra.utils"].instantiate.__globals__  # noqa: B018 (already installed by an earlier test in this process)
    if "deepspeech_pytorch.loader.data_module" not in sys.modules:
        dm = types.ModuleType("deepspeech_pytorch.loader.data_module")

        class DeepSpeechDataModule:                 # loader/data_module.py:9-24 (librosa / sox / torchaudio are not in the image)
            def __init__(self, labels, data_cfg, normalize):
                self.labels, self.data_cfg, self.normalize = labels, data_cfg, normalize

        dm.DeepSpeechDataModule = DeepSpeechDataModule
        sys.modules["deepspeech_pytorch.loader.data_module"] = dm


def test_reference_train_builds_and_fits_the_dropin_class(tmp_path):
    box = []
    _install_loop_stubs(box)
    ns = ref_harness.load_reference()
    import deepspeech_pytorch.model as ref_model
    from deepspeech.pytorch_amd.model import DeepSpeech as Ds2HipDeepSpeech
    RefDeepSpeech = ref_model.DeepSpeech
    ref_model.DeepSpeech = Ds2HipDeepSpeech                      # the INTEGRATION.md switch, before training.py binds the name
    sys.modules.pop("deepspeech_pytorch.training", None)
    try:
        from deepspeech_pytorch import training
        from deepspeech_pytorch.configs.train_config import DeepSpeechConfig, AdamConfig, BiDirectionalConfig
        from deepspeech_pytorch.configs.lightning_config import ModelCheckpointConf
        cfg = DeepSpeechConfig(optim=AdamConfig(), model=BiDirectionalConfig(rnn_type=ns.RNNType.gru, hidden_size=32, hidden_layers=2),
                               checkpoint=ModelCheckpointConf())
        cfg.trainer.enable_checkpointing = False                  # FileCheckpointHandler is Lightning's ModelCheckpoint: out of scope
        cfg.trainer.precision = 16                                # configs/an4.yaml:11
        cfg.trainer.gradient_clip_val = 400                       # configs/an4.yaml:12
        cfg.trainer.max_epochs = 2
        training.train(cfg)                                       # training.py:13-47, unmodified
    finally:
        ref_model.DeepSpeech = RefDeepSpeech
        sys.modules.pop("deepspeech_pytorch.training", None)
    assert len(box) == 1
    tr = box[0]
    assert tr.ignored_kwargs.get("replace_sampler_ddp") is False  # training.py:44
    (model, data), = tr.fitted
    assert type(model) is Ds2HipDeepSpeech
    assert data.normalize is True and data.labels == ns.labels    # training.py:27-31
    assert model.bidirectional and model.precision == 16 and len(model.rnns) == 2
    assert tr.amp and tr.gradient_clip_val == 400
    # what Lightning does with the module before the first batch / at a checkpoint
    assert type(tr.optimizer).__name__ in ("AdamW", "FusedAdamW") and isinstance(tr.optimizer, torch.optim.AdamW)
    assert isinstance(tr.scheduler, torch.optim.lr_scheduler.ExponentialLR)
    assert set(model.hparams) == {"labels", "model_cfg", "precision", "optim_cfg", "spect_cfg"}
    path = os.path.join(tmp_path, "last.ckpt")
    tr.save_checkpoint(path)
    restored = Ds2HipDeepSpeech.load_from_checkpoint(path)         # utils.py:31
    assert list(restored.state_dict()) == list(model.state_dict())
    # a checkpoint written by the REFERENCE class loads into the drop-in class and back, strict
    torch.manual_seed(1)
    ref = RefDeepSpeech(labels=ns.labels, model_cfg=cfg.model, precision=32, optim_cfg=cfg.optim, spect_cfg=cfg.data.spect)
    model.load_state_dict(ref.state_dict(), strict=True)
    ref.load_state_dict(model.state_dict(), strict=True)
    # the hot path itself has no CPU fallback: the first training step on CPU tensors must raise, loudly
    from deepspeech.pytorch_amd._lib import Ds2HipError
    x = torch.zeros(2, 1, 161, 40)
    with pytest.raises(Ds2HipError):
        model.training_step((x, torch.ones(4, dtype=torch.int64), torch.ones(2), torch.tensor([2, 2], dtype=torch.int32)), 0)
