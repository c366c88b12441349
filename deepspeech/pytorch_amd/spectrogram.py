"""Device-side replacement for the CPU spectrogram workers of the reference loader: ``SpectrogramParser.compute_spectrogram``
(loader/data_loader.py:73-94) for every utterance of a batch + the zero-padded batch layout of ``_collate_fn`` (:247-270), as one
call on the HIP kernels of csrc/ds2_spect.hip.  Output = exactly the ``inputs`` / ``input_percentages`` the model's
``training_step`` takes.  Geometry: 16 kHz, 20 ms window, 10 ms stride (n_fft 320, hop 160 -> 161 bins), the only geometry the
conv kernels support; the window type follows ``SpectConfig.window`` (enums.py:8-14).

The reference's STFT lives in a third-party dependency that is not vendored and not pinned (``librosa``, requirements.txt:4):
``center=True`` padding is zeros in librosa >= 0.10 (``pad_mode="constant"``, the default here) and reflection before."""
import math

import numpy as np
import torch

from . import ops
from ._lib import Ds2HipError, call, query

N_FFT, HOP, N_BIN = 320, 160, 161


def window_values(name, n=N_FFT):
    """Periodic (fftbins=True) windows as scipy.signal.get_window / librosa.filters.get_window produce them."""
    k = np.arange(n, dtype=np.float64)
    name = getattr(name, "value", name)
    if name == "hamming":
        return 0.54 - 0.46 * np.cos(2 * np.pi * k / n)
    if name == "hann":
        return 0.5 - 0.5 * np.cos(2 * np.pi * k / n)
    if name == "blackman":
        return 0.42 - 0.5 * np.cos(2 * np.pi * k / n) + 0.08 * np.cos(4 * np.pi * k / n)
    if name == "bartlett":
        return 1.0 - np.abs(2.0 * k / n - 1.0)
    raise ValueError("unsupported spectrogram window %r" % (name,))


def dft_basis(window):
    """[2*161][320] float32: rows 0..160 = w[k] cos(2 pi f k / 320), rows 161..321 = -w[k] sin(2 pi f k / 320)."""
    w = window_values(window)
    f = np.arange(N_BIN, dtype=np.float64)[:, None]
    k = np.arange(N_FFT, dtype=np.float64)[None, :]
    ang = 2 * np.pi * f * k / N_FFT
    return np.concatenate([np.cos(ang) * w, -np.sin(ang) * w], 0).astype(np.float32)


class SpectrogramFrontEnd:
    def __init__(self, spect_cfg=None, normalize=True, pad_mode="constant"):
        sr = getattr(spect_cfg, "sample_rate", 16000)
        n_fft = int(sr * getattr(spect_cfg, "window_size", 0.02))
        hop = int(sr * getattr(spect_cfg, "window_stride", 0.01))
        if (n_fft, hop) != (N_FFT, HOP):
            raise ValueError("the gfx950 front-end is specialised for n_fft 320 / hop 160 (16 kHz, 20 ms, 10 ms); got %d / %d" % (n_fft, hop))
        if pad_mode not in ("constant", "reflect"):
            raise ValueError("pad_mode must be 'constant' (librosa >= 0.10) or 'reflect'")
        self.window = getattr(spect_cfg, "window", "hamming")
        self.normalize, self.reflect = bool(normalize), pad_mode == "reflect"
        self._basis = {}

    def _basis_on(self, dev):
        b = self._basis.get(dev)
        if b is None:
            b = self._basis[dev] = torch.from_numpy(dft_basis(self.window)).to(dev)
        return b

    def __call__(self, wav, nsamples):
        """wav: [N][Lmax] float32 on a HIP device (row n = utterance n, zero beyond nsamples[n]); nsamples: [N] ints.
        Returns (inputs (N,1,161,Tmax) float32, input_percentages [N] float32 (CPU), frames [N] int64 (CPU))."""
        if not wav.is_cuda:
            raise Ds2HipError("SpectrogramFrontEnd needs the waveforms on a HIP device; there is no CPU path")
        wav = wav.float().contiguous()
        N, Lmax = wav.shape
        ns = torch.as_tensor(nsamples, dtype=torch.int32).cpu()
        if ns.numel() != N or int(ns.min()) < 1:
            raise ValueError("nsamples must hold one positive sample count per waveform row")
        Lm = int(ns.max())
        if Lm > Lmax:
            raise ValueError("nsamples exceeds the waveform buffer")
        Tmax = 1 + Lm // HOP
        out = torch.empty((N, 1, N_BIN, Tmax), dtype=torch.float32, device=wav.device)
        ws = torch.empty(query("ds2_spect_ws_bytes", N, Lm), dtype=torch.uint8, device=wav.device)
        basis = self._basis_on(wav.device)
        ns_dev = ns.to(wav.device)       # held in a local until the launch is enqueued (a temporary would be freed -- and its block
        #                                  possibly re-used by the basis upload -- before the kernels read it)
        call("ds2_spectrogram", ops.P(wav), wav.stride(0), ops.P(ns_dev), N, Lm, ops.P(basis),
             1 if self.reflect else 0, 1 if self.normalize else 0, ops.P(out), ops.P(ws), ops.S())
        frames = 1 + ns.to(torch.int64) // HOP
        pct = (frames.to(torch.float64) / float(Tmax)).to(torch.float32)      # _collate_fn: seq_length / float(max_seqlength)
        return out, pct, frames

    def collate(self, waveforms, transcripts=None, int16_scale=False):
        """waveforms: list of 1-D float tensors.  Sorts by length descending (as _collate_fn sorts by frame count,
        data_loader.py:251), pads, uploads and runs the front-end.  Returns (inputs, input_percentages, order), or -- with
        `transcripts` (one sequence of label indices per waveform) -- the reference's whole batch tuple
        (inputs, targets, input_percentages, target_sizes) in the sorted order, as _collate_fn builds it (data_loader.py:259-270).

        AMPLITUDE: log1p(|STFT|) is not scale-free.  The reference's load_audio (data_loader.py:23-30) hands
        compute_spectrogram samples in [-1, 1] (int16 / 32767); pass waveforms on that scale, or raw int16-range samples with
        int16_scale=True (they are divided by 32767 here, as load_audio does)."""
        order = sorted(range(len(waveforms)), key=lambda i: -len(waveforms[i]))
        Lmax = len(waveforms[order[0]])
        buf = torch.zeros((len(waveforms), Lmax), dtype=torch.float32)
        for r, i in enumerate(order):
            buf[r, :len(waveforms[i])] = torch.as_tensor(waveforms[i], dtype=torch.float32)
        if int16_scale:
            buf /= 32767.0
        inputs, pct, _ = self(buf.cuda(), [len(waveforms[i]) for i in order])
        if transcripts is None:
            return inputs, pct, order
        tg = [torch.as_tensor(transcripts[i], dtype=torch.int64).reshape(-1) for i in order]
        target_sizes = torch.tensor([len(t) for t in tg], dtype=torch.int32)
        targets = torch.cat(tg) if tg else torch.zeros(0, dtype=torch.int64)
        return inputs, targets, pct, target_sizes
