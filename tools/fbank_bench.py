"""Timing of the GPU log-mel front-end at the training batch: 512 clips x 10.05 s at 16 kHz -> [512, 1, 1000, 128].
Usage: python tools/fbank_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vipant_amd.frontend import KaldiFbank  # noqa: E402

dev = "cuda:0"
for sr, b in ((16000, 512), (44100, 512)):
    n = int(10.05 * sr)
    fb = KaldiFbank(sr, 128, 1000, norms=(-4.94, 5.76), freq_mask_param=32, time_mask_param=200, device=dev)
    w = 0.1 * torch.randn(b, n, device=dev)
    masks = fb.draw_masks(b).to(dev)
    fb(w, None, masks); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = fb(w, None, masks)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = w.numel() * 4 + out.numel() * 4
    print(f"sr={sr} b={b}: {ms:7.3f} ms per batch  ({b / ms * 1e3:9.0f} clips/s, {byt / ms / 1e6:6.1f} GB/s of waveform-in + fbank-out)")
