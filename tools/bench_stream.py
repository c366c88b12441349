#!/usr/bin/env python
"""Chunked batch-1 inference with hidden-state carry -- reference inference.py:79-99 (run_transcribe: eval mode, one utterance,
`hs` fed back chunk after chunk, outputs concatenated on the time axis) -- on the drop-in class, and the same loop on stock
PyTorch-ROCm (oracle/ds2_torch_port.py).  SURVEY.md section 8(f)-4: the serving path reuses the training kernels (the persistent
forward sweep takes h0/c0 and returns hn/cn).  Prints one JSON line per model: per-chunk latency and real-time factor.

    python tools/bench_stream.py [--seconds 30 --chunk 2.0 --stock]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def run(model_fn, chunks, reps):
    lat = []
    for r in range(reps + 1):
        hs, outs = None, []
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for c in chunks:
            t0 = time.perf_counter()
            out, hs = model_fn(c, hs)
            outs.append(out.cpu())              # as run_transcribe does: every chunk's output leaves the device
            if r > 0:
                lat.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        total = time.perf_counter() - t_all
    return np.array(lat), total, torch.cat(outs, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--chunk", type=float, default=2.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--stock", action="store_true")
    a = ap.parse_args()
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.model import DeepSpeech
    from oracle import ds2_torch_port as TP
    dev = "cuda"
    T, tc = int(a.seconds * 100), int(a.chunk * 100)
    rs = np.random.RandomState(0)
    spect = torch.from_numpy(rs.standard_normal((1, 1, 161, T)).astype(np.float32)).to(dev)
    chunks = [spect[:, :, :, i:i + tc].contiguous() for i in range(0, T, tc)]
    for name, kind, H, L, bi in (("uni-LSTM-1024x5+lookahead", "lstm", 1024, 5, False), ("BiGRU-1024x5", "gru", 1024, 5, True)):
        cfg = dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=20)
        state = TP.random_state(cfg, 0)
        rt = getattr(configs.RNNType, kind)
        mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
            configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=20)
        for impl in (["ds2hip"] + (["stock-pytorch-rocm"] if a.stock else [])):
            if impl == "ds2hip":
                m = DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig())
                m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
                m = m.to(dev).eval()

                def fn(c, hs, m=m):
                    with torch.no_grad():
                        out, _, hs2 = m(c, torch.tensor([c.shape[3]], dtype=torch.int), hs)
                    return out, hs2
            else:
                port = TP.Port(cfg, state, dev)

                def fn(c, hs, port=port):
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                        out, _, hs2 = port.forward(c, torch.tensor([c.shape[3]], dtype=torch.int), train=False, hs=hs, return_hs=True)
                    return out.float(), hs2
            try:
                lat, total, out = run(fn, chunks, a.reps)
            except TypeError as e:            # the stock port has no hs plumbing: report and move on
                print(json.dumps({"model": name, "impl": impl, "error": str(e)[:200]}))
                continue
            print(json.dumps({"metric": "chunked batch-1 inference (reference inference.py:79-99)", "model": name, "impl": impl,
                              "audio_seconds": a.seconds, "chunk_seconds": a.chunk, "chunks": len(chunks),
                              "ms_per_chunk_median": round(float(np.median(lat)) * 1e3, 3),
                              "ms_per_chunk_p95": round(float(np.percentile(lat, 95)) * 1e3, 3),
                              "real_time_factor": round(a.seconds / total, 1), "out_frames": int(out.shape[1]), "dtype": "bf16"}))


if __name__ == "__main__":
    main()
