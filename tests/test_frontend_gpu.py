"""GPU log-mel front-end against the CPU restatement of the Kaldi filter bank (oracle/fbank_cpu.py).
PARITY UNPINNED with respect to torchaudio itself (absent from the image; see the oracle's header)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fbank_cpu as FB  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _clips(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    w = 0.05 * torch.randn(b, n, generator=g)
    for i in range(b):
        f0 = 200.0 + 700.0 * i
        w[i] += 0.3 * torch.sin(2 * torch.pi * f0 * t) + 0.1 * torch.sin(2 * torch.pi * 3.1 * f0 * t) + 0.02 * (i - 1)
    return w


@pytest.mark.parametrize("sr,secs,T,F", [(16000, 3.05, 300, 128), (16000, 1.0, 128, 64), (44100, 1.3, 128, 128), (32000, 0.9, 96, 80)])
def test_fbank_matches_oracle(sr, secs, T, F):
    from vipant_amd.frontend import KaldiFbank
    n = int(sr * secs)
    w = _clips(3, n, seed=sr)
    lens = torch.tensor([n, n - 1234, max(n // 3, 900)])
    fb = KaldiFbank(sr, F, T, norms=(), zero_mean_wf=True, device=DEV)
    got = fb(w.to(DEV), lens.to(DEV)).cpu()
    assert got.shape == (3, 1, T, F)
    for i in range(3):
        ref = FB.spectrogram_item(w[i, : int(lens[i])], sr, T, F, norms=(), zero_mean_wf=True)
        err = (got[i, 0] - ref).abs()
        # log-mel of band-limited tones + noise: empty low filters sit at log(eps); tolerance covers fp32 FFT ordering
        assert err.max() < 2e-3, (i, float(err.max()))
        assert err.mean() < 5e-5


def test_fbank_normalisation_padding_and_masks():
    from vipant_amd.frontend import KaldiFbank
    sr, T, F = 16000, 200, 128
    w = _clips(4, 16000, seed=5)                       # 98 frames < T: rows 98.. are padding
    fb = KaldiFbank(sr, F, T, norms=(-4.94, 5.76), zero_mean_wf=True, freq_mask_param=32, time_mask_param=60, device=DEV)
    g = torch.Generator().manual_seed(11)
    masks = fb.draw_masks(4, generator=g)
    g2 = torch.Generator().manual_seed(11)
    got = fb(w.to(DEV), None, masks).cpu()
    for i in range(4):
        fm = FB.draw_mask(F, 32, g2); tm = FB.draw_mask(T, 60, g2)
        assert masks[i].tolist() == [fm[0], fm[1], tm[0], tm[1]]
        ref = FB.spectrogram_item(w[i], sr, T, F, norms=(-4.94, 5.76), zero_mean_wf=True, freq_mask=fm, time_mask=tm)
        assert (got[i, 0] - ref).abs().max() < 5e-4
    nfr = fb.num_frames(16000)
    assert nfr == 98
    keep = torch.ones(T, F, dtype=torch.bool)
    keep[:, masks[0, 0]:masks[0, 1]] = False; keep[masks[0, 2]:masks[0, 3], :] = False
    pad_val = (0.0 - (-4.94)) / 5.76
    assert torch.allclose(got[0, 0][nfr:][keep[nfr:]], torch.tensor(pad_val), atol=1e-6)
    assert float(got[0, 0][~keep].abs().max()) == 0.0


def test_fbank_feeds_the_audio_tower():
    """The front-end's output is exactly the [b, 1, T, F] layout ViTPreEncoder takes (cvap/module/val.py:228-259)."""
    from vipant_amd.frontend import KaldiFbank
    from vipant_amd import ops
    fb = KaldiFbank(16000, 64, 256, norms=(-4.94, 5.76), device=DEV)
    x = fb(_clips(2, 16000 * 3, seed=2).to(DEV))
    assert x.shape == (2, 1, 256, 64) and x.is_contiguous() and bool(torch.isfinite(x).all())
    from vipant_amd._ffi import VipantError
    with pytest.raises(VipantError):
        fb(torch.zeros(2, 16000))                      # CPU tensor: no fallback
