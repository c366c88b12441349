"""Synthetic batches in the layout of the reference collate function.

Layout contract (reference ``loader/data_loader.py:247-270`` ``_collate_fn``): samples sorted by length
descending, zero-padded to ``(N, 1, 161, Tmax)`` float32, targets flattened to one int64 vector,
``input_percentages[i] = T_i / float(Tmax)`` float32, ``target_sizes`` int32.  Value distribution per
SURVEY.md section 8(d): spectrograms are per-utterance normalised log-magnitudes (``data_loader.py:86-92``) so
N(0,1) is the faithful synthetic fill; targets uniform in [1,28] (0 is the CTC blank, never a target,
``data_loader.py:240``); ~12 characters per second of audio.

``numpy.random.RandomState`` is used because its stream is frozen across numpy versions: the golden fixtures
store only a seed, not the input tensors.
"""
import numpy as np

N_FREQ = 161
FRAME_SECONDS = 0.01  # SpectConfig.window_stride (reference configs/train_config.py:20)


def synth_lengths(n, t_min, t_max, seed, linear=False):
    """Frame counts sorted descending; the longest clip always equals t_max (so Tmax is deterministic)."""
    if n == 1:
        return np.array([t_max], dtype=np.int64)
    if linear:
        ln = np.round(np.linspace(t_max, t_min, n)).astype(np.int64)
    else:
        rs = np.random.RandomState(seed)
        ln = rs.randint(t_min, t_max + 1, size=n).astype(np.int64)
        ln[0] = t_max
    return np.sort(ln)[::-1].copy()


def synth_batch(lengths, seed, chars_per_second=12.0, n_labels=29, n_freq=N_FREQ):
    """Returns numpy arrays (inputs f32 (N,1,n_freq,Tmax), targets i64 [sum S], input_percentages f32 [N],
    target_sizes i32 [N]) -- the reference 4-tuple.  n_freq = sample_rate * window_size / 2 + 1 (161 at 16 kHz / 20 ms)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n, t_max = len(lengths), int(lengths.max())
    rs = np.random.RandomState(seed)
    inputs = np.zeros((n, 1, n_freq, t_max), dtype=np.float32)
    for i, t in enumerate(lengths):
        inputs[i, 0, :, :t] = rs.standard_normal((n_freq, int(t))).astype(np.float32)
    target_sizes = np.maximum(1, np.floor(chars_per_second * lengths * FRAME_SECONDS)).astype(np.int32)
    targets = rs.randint(1, n_labels, size=int(target_sizes.sum())).astype(np.int64)
    input_percentages = (lengths / float(t_max)).astype(np.float32)
    return inputs, targets, input_percentages, target_sizes


def audio_seconds(lengths):
    """True (unpadded) audio seconds in the batch: the numerator of the headline metric."""
    return float(np.asarray(lengths).sum() * FRAME_SECONDS)


def synth_params(shapes, seed, scale=None):
    """Deterministic parameter fill for fixtures: dict name -> float32 array.  ``shapes`` maps the reference
    state_dict names to shapes.  Weights ~ U(-a, a) with a = 1/sqrt(fan_in) unless `scale` given; BN weight in
    [0.5,1.5], BN bias/running_mean small, running_var in [0.5,1.5]."""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shp, dtype=np.int64)
            continue
        is_bn = ("batch_norm" in name) or name.startswith("fc.0.module.0.") or \
                name.startswith("conv.seq_module.1.") or name.startswith("conv.seq_module.4.")
        if is_bn:
            if name.endswith("running_var") or name.endswith(".weight"):
                v = rs.uniform(0.5, 1.5, size=shp)
            else:
                v = rs.uniform(-0.2, 0.2, size=shp)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else int(shp[0])
            a = scale if scale is not None else 1.0 / np.sqrt(max(fan_in, 1))
            if name.startswith("lookahead"):
                a = 0.3
            v = rs.uniform(-a, a, size=shp)
        out[name] = v.astype(np.float32)
    return out
