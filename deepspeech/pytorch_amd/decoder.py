"""Greedy decoder + WER / CER with the reference's interfaces (``deepspeech_pytorch.decoder.GreedyDecoder``,
decoder.py:117-181; ``deepspeech_pytorch.validation.WordErrorRate / CharErrorRate``, validation.py:13-132), so that
``DeepSpeech.validation_step`` (model.py:251-271) works exactly as in the reference.

The arg-max + repeat collapse + blank removal of ``decode`` runs on the device (ds2_greedy_decode); only the surviving
labels travel to the host.  String building and the edit distance are host-side bookkeeping, as in the reference.
When the reference's metric classes are importable the model uses THEM (with this decoder inside); the classes below are the
stand-ins for images without torchmetrics / Levenshtein."""
import torch

from . import ops


class GreedyDecoder:
    def __init__(self, labels, blank_index=0):
        self.labels = labels
        self.int_to_char = dict((i, c) for (i, c) in enumerate(labels))
        self.blank_index = blank_index
        space_index = len(labels)          # decoder.py:36-38: out of range unless ' ' is a label
        if ' ' in labels:
            space_index = labels.index(' ')
        self.space_index = space_index

    # ---- reference decoder.py:121-162 (host-side string building; used for the TARGET side by the metrics) ----------
    def convert_to_strings(self, sequences, sizes=None, remove_repetitions=False, return_offsets=False):
        strings, offsets = [], ([] if return_offsets else None)
        for x in range(len(sequences)):
            seq_len = sizes[x] if sizes is not None else len(sequences[x])
            string, string_offsets = self.process_string(sequences[x], seq_len, remove_repetitions)
            strings.append([string])
            if return_offsets:
                offsets.append([string_offsets])
        return (strings, offsets) if return_offsets else strings

    def process_string(self, sequence, size, remove_repetitions=False):
        seq = [int(v) for v in (sequence.tolist() if hasattr(sequence, "tolist") else sequence)][:int(size)]
        chars, offsets = [], []
        for i, v in enumerate(seq):
            if v != self.blank_index:
                if remove_repetitions and i != 0 and v == seq[i - 1]:
                    continue
                chars.append(' ' if v == self.space_index else self.int_to_char[v])
                offsets.append(i)
        return ''.join(chars), torch.tensor(offsets, dtype=torch.int)

    # ---- reference decoder.py:164-181, on the device -------------------------------------------------------------------
    def decode(self, probs, sizes=None):
        """probs: (N, T', C) scores on a HIP device.  Returns (strings, offsets) exactly as the reference: strings[n] = [str],
        offsets[n] = [int tensor of the frame of every emitted character]."""
        toks, offs = ops.greedy_decode(probs, sizes, self.blank_index)
        strings = [[''.join(' ' if v == self.space_index else self.int_to_char[v] for v in t)] for t in toks]
        return strings, [[o.to(torch.int)] for o in offs]


def _edit_distance(a, b):
    try:
        import Levenshtein as Lev
        return Lev.distance(a, b)
    except Exception:
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]


class _ErrorRate:
    """validation.py:13-45 without torchmetrics (single process; Lightning's DDP metric sync needs the reference class)."""

    def __init__(self, decoder, target_decoder):
        self.decoder, self.target_decoder = decoder, target_decoder
        self.errors, self.total = 0, 0

    def __call__(self, preds, preds_sizes, targets, target_sizes):
        return self.update(preds, preds_sizes, targets, target_sizes)

    def update(self, preds, preds_sizes, targets, target_sizes):
        split_targets, offset = [], 0
        for size in target_sizes:
            split_targets.append(targets[offset:offset + int(size)])
            offset += int(size)
        decoded_output, _ = self.decoder.decode(preds, preds_sizes)
        target_strings = self.target_decoder.convert_to_strings(split_targets)
        for x in range(len(target_strings)):
            self.calculate_metric(decoded_output[x][0], target_strings[x][0])

    def compute(self):
        return float(self.errors) / max(self.total, 1) * 100

    def reset(self):
        self.errors, self.total = 0, 0


class CharErrorRate(_ErrorRate):   # validation.py:48-87
    def calculate_metric(self, transcript, reference):
        self.errors += _edit_distance(transcript.replace(' ', ''), reference.replace(' ', ''))
        self.total += len(reference.replace(' ', ''))


class WordErrorRate(_ErrorRate):   # validation.py:90-132
    def calculate_metric(self, transcript, reference):
        b = set(transcript.split() + reference.split())
        word2char = dict(zip(b, range(len(b))))
        w1 = ''.join(chr(word2char[w]) for w in transcript.split())
        w2 = ''.join(chr(word2char[w]) for w in reference.split())
        self.errors += _edit_distance(w1, w2)
        self.total += len(reference.split())
