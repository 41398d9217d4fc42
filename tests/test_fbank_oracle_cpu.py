"""Known-answer and invariance checks of the CPU restatement of the Kaldi filter bank (oracle/fbank_cpu.py).
torchaudio is absent from the image, so these analytic properties -- not torchaudio outputs -- are what pins it
(the oracle's header says PARITY UNPINNED for that reason)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fbank_cpu as FB  # noqa: E402


def test_framing_and_window_sizes():
    assert FB.window_properties(16000) == (160, 400, 512)
    assert FB.window_properties(44100) == (441, 1102, 2048)
    assert FB.window_properties(32000) == (320, 800, 1024)
    assert FB.num_frames(399, 400, 160) == 0 and FB.num_frames(400, 400, 160) == 1 and FB.num_frames(16000, 400, 160) == 98
    # 10.05 s at 16 kHz -> 1003 frames, cropped to max_len 1000 by the dataset (transform.py:24-34)
    assert FB.num_frames(int(10.05 * 16000), 400, 160) == 1003


def test_mel_bank_geometry():
    b = FB.mel_banks(128, 512, 16000.0)
    assert b.shape == (128, 257) and float(b.min()) == 0.0 and float(b.max()) <= 1.0 and float(b[:, -1].abs().max()) == 0.0
    centers = (b * torch.arange(257)).sum(1) / b.sum(1).clamp_min(1e-9)
    live = b.sum(1) > 0
    assert bool((centers[live][1:] >= centers[live][:-1]).all())         # filters ordered by frequency (narrow low ones share a bin)
    # HTK mel spacing: centre of filter m at mel_lo + (m + 1) * delta
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    delta = (mel(8000.0) - mel(20.0)) / 129
    k = 100
    f_center = 700.0 * (math.exp((mel(20.0) + (k + 1) * delta) / 1127.0) - 1.0)
    assert abs(float(b[k].argmax()) * 31.25 - f_center) <= 31.25
    # a wide bank: between the first and the last centre the overlapping triangles form a partition of unity
    w = FB.mel_banks(23, 512, 16000.0)
    lo, hi = int(w[0].argmax()) + 1, int(w[-1].argmax())
    assert torch.allclose(w.sum(0)[lo:hi], torch.ones(hi - lo), atol=1e-5) and float(w.sum(0).max()) <= 1.0 + 1e-5


def test_tone_lands_in_the_right_filter_and_scales():
    sr, f0 = 16000, 2000.0
    t = torch.arange(sr) / sr
    x = 0.25 * torch.sin(2 * math.pi * f0 * t)
    y = FB.kaldi_fbank(x, sr, 128)
    assert y.shape == (98, 128)
    b = FB.mel_banks(128, 512, float(sr))
    want = int(b[:, int(round(f0 / 31.25))].argmax())
    assert int(y[10].argmax()) == want
    # amplitude x a  ->  log power + 2 ln a (away from the epsilon floor); DC offset removed per frame
    y2 = FB.kaldi_fbank(3.0 * x + 0.7, sr, 128)
    hot = y[10] > -5.0
    assert torch.allclose((y2[10] - y[10])[hot], torch.full((int(hot.sum()),), 2 * math.log(3.0)), atol=2e-3)
    # silence -> log(eps) everywhere
    z = FB.kaldi_fbank(torch.zeros(sr), sr, 128)
    assert torch.allclose(z, torch.full_like(z, math.log(FB.EPS)))


def test_dataset_item_glue():
    sr = 16000
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sr, generator=g) * 0.1
    item = FB.spectrogram_item(x, sr, 120, 64, norms=(-4.0, 4.0), freq_mask=(3, 9), time_mask=(100, 110))
    assert item.shape == (120, 64)
    assert float(item[:, 3:9].abs().max()) == 0.0 and float(item[100:110].abs().max()) == 0.0
    assert torch.allclose(item[98:100, 10:], torch.full((2, 54), 1.0))   # zero padding then (0 - mean) / std
    raw = FB.spectrogram_item(x, sr, 120, 64)
    assert torch.allclose(raw[:98], FB.kaldi_fbank(x - x.mean(), sr, 64))
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    s, e = FB.draw_mask(128, 32, g1)
    v = torch.rand(1, generator=g2) * 32; m = torch.rand(1, generator=g2) * (128 - v)
    assert (s, e) == (int(m.long()), int(m.long()) + int(v.long())) and 0 <= e - s < 32


def test_oracle_agrees_with_an_independent_kaldi_compatible_implementation():
    """tests/golden/fbank_crosscheck.npz: log-mel features of four waveforms computed by transformers.audio_utils.spectrogram -- the
    Kaldi-compatible numpy fallback HuggingFace ships for its AST feature extractor when torchaudio is absent -- with the reference's
    parameters (tests/golden/make_fbank_crosscheck.py).  Not torchaudio itself (the oracle stays 'parity unpinned' in that strict
    sense), but an independent restatement of the same published algorithm: on log energies spanning [-15.9, 7] the median
    difference is below 2e-5, the 99th percentile below 1e-3, the maximum below 1e-2 (observed: 5e-6 / 3e-4 / 3.1e-3 at 16 kHz, 1e-4
    maximum at the other rates; float32 here, float64 there)."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fbank_crosscheck.npz"))
    for tag in ("16k", "44k", "16k_64", "22k"):
        wav, sr, bins = torch.from_numpy(g[f"{tag}_wav"]), int(g[f"{tag}_sr"]), int(g[f"{tag}_bins"])
        ref = torch.from_numpy(g[f"{tag}_fbank"])
        got = FB.kaldi_fbank(wav, sr, bins).double()
        assert got.shape == ref.shape, (tag, got.shape, ref.shape)
        diff = (got - ref).abs()
        # mel bins whose energy is ~1e-6 of their frame's loudest (next to the chirp, or in the frames leaving the silent stretch)
        # carry the float32 round-off of this restatement's FFT: up to 3e-3 there, 1e-4 everywhere else
        assert float(diff.median()) < 2e-5, (tag, float(diff.median()))
        assert float(torch.quantile(diff.flatten(), 0.99)) < 1e-3, (tag, float(torch.quantile(diff.flatten(), 0.99)))
        assert float(diff.max()) < 1e-2, (tag, float(diff.max()))
        assert float(diff.mean()) < 3e-5, (tag, float(diff.mean()))
        floor = ref < -15.0                      # the silent stretch sits on the epsilon floor in both
        assert bool(floor.any()) == bool((got < -15.0).any())
