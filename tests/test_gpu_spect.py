"""GPU parity of the spectrogram front-end (csrc/ds2_spect.hip) against the oracle's restatement of
SpectrogramParser.compute_spectrogram + _collate_fn (reference loader/data_loader.py:73-94, 247-270)."""
import numpy as np
import pytest
import torch

from oracle import ds2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pad_mode,normalize", [("constant", True), ("reflect", True), ("constant", False)])
def test_spectrogram_batch_matches_oracle(pad_mode, normalize):
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.spectrogram import SpectrogramFrontEnd
    rs = np.random.RandomState(7)
    lens = [24000, 23999, 16161, 8000, 801, 480]          # sorted descending like _collate_fn; odd lengths and a 3-frame clip
    wavs = [(rs.standard_normal(n) * rs.uniform(0.05, 0.6) + 0.01).astype(np.float32) for n in lens]
    fe = SpectrogramFrontEnd(configs.SpectConfig(), normalize=normalize, pad_mode=pad_mode)
    buf = torch.zeros((len(lens), max(lens)))
    for i, w in enumerate(wavs):
        buf[i, :len(w)] = torch.from_numpy(w)
    inputs, pct, frames = fe(buf.cuda(), lens)
    Tmax = 1 + max(lens) // 160
    assert tuple(inputs.shape) == (len(lens), 1, 161, Tmax) and inputs.dtype == torch.float32
    got = inputs.cpu().numpy().astype(np.float64)
    for i, w in enumerate(wavs):
        ref = O.log_spectrogram(w.astype(np.float64), normalize=normalize, pad_mode=pad_mode)
        T = ref.shape[1]
        assert int(frames[i]) == T
        assert np.abs(got[i, 0, :, :T] - ref).max() < 2e-4, (i, np.abs(got[i, 0, :, :T] - ref).max())
        assert np.all(got[i, 0, :, T:] == 0)              # zero padding of the batch layout
        assert abs(float(pct[i]) - np.float32(T / float(Tmax))) < 1e-7
    # the percentages reproduce the frame counts through training_step's float round trip (model.py:243)
    assert (pct.cpu().mul(Tmax).int().numpy() == np.array([1 + n // 160 for n in lens])).all()


def test_spectrogram_feeds_the_model():
    """waveforms -> front-end -> training_step on the device: finite loss and gradients (the path of SURVEY 8(f)-3)."""
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.model import DeepSpeech
    from deepspeech.pytorch_amd.spectrogram import SpectrogramFrontEnd
    rs = np.random.RandomState(1)
    wavs = [rs.standard_normal(n).astype(np.float32) * 0.1 for n in (12000, 9000, 16000)]
    inputs, pct, order = SpectrogramFrontEnd(configs.SpectConfig()).collate(wavs)
    assert order == [2, 0, 1]
    torch.manual_seed(0)
    mc = configs.BiDirectionalConfig(rnn_type=configs.RNNType.gru, hidden_size=32, hidden_layers=2)
    m = DeepSpeech(configs.LABELS, mc, 32, configs.AdamConfig(), configs.SpectConfig()).cuda().train()
    targets = torch.from_numpy(rs.randint(1, 29, size=12).astype(np.int64))
    loss = m.training_step((inputs, targets, pct.clone(), torch.tensor([5, 4, 3], dtype=torch.int32)), 0)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters())
