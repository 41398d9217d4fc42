"""CPU restatement of the log-mel front-end that feeds the hot path (TEST INFRASTRUCTURE ONLY -- imported by tests/
and nothing else).

PARITY UNPINNED.  The arithmetic lives in a third-party dependency that is absent from /root/reference and from this
image: `torchaudio.compliance.kaldi.fbank` and `torchaudio.functional.mask_along_axis`, pinned `torchaudio==0.8.1`
(/root/reference/requirements.txt:12).  What follows restates their published algorithm (Kaldi's `compute-fbank-feats`
as transcribed by torchaudio: snip_edges framing, per-frame DC removal, pre-emphasis with a replicated first sample,
symmetric Hann window, zero padding to the next power of two, power spectrum, HTK-mel triangular bank from 20 Hz to
Nyquist, log with an epsilon floor), anchored on the reference's own call sites:

  * cvap/data/audio/transform.py:12-35   `_extract_kaldi_spectrogram`: zero-mean waveform, kaldi.fbank(**params), crop
  * cvap/data/image_audio.py:119-126     params = {htk_compat: True, use_energy: False, window_type: 'hanning',
                                         num_mel_bins, dither: 0.0, frame_shift: 10}
  * cvap/data/image_audio.py:183-207     zero padding to max_audio_len frames, (x - mean) / std, SpecAugment masks
  * configs/running/audio/default.yaml   FrequencyMasking(32), TimeMasking(200), zero_mean_wf: True

No golden vector from torchaudio itself could be generated here, so tests pin the HIP kernels to THIS file only.

Cross-check (round 2; it does not lift the "unpinned" status above): an independent Kaldi-compatible implementation that IS
importable in the build container -- `transformers.audio_utils.spectrogram` (transformers 5.15.0), the numpy fallback HuggingFace's
ASTFeatureExtractor uses in place of `torchaudio.compliance.kaldi.fbank` when torchaudio is missing -- run with the reference's
parameters agrees with `kaldi_fbank` below on four waveforms at 16 / 22.05 / 44.1 kHz, 64 and 128 bins: median difference 5e-6,
maximum 3e-3 (on bins at ~1e-6 of their frame's loudest energy: float32 FFT round-off here, float64 there), log energies in
[-15.9, 7].  Vectors: tests/golden/fbank_crosscheck.npz (made by tests/golden/make_fbank_crosscheck.py); test:
tests/test_fbank_oracle_cpu.py::test_oracle_agrees_with_an_independent_kaldi_compatible_implementation.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

EPS = torch.finfo(torch.float32).eps


def window_properties(sample_rate: float, frame_shift_ms: float = 10.0, frame_length_ms: float = 25.0) -> Tuple[int, int, int]:
    """(window_shift, window_size, padded_window_size) in samples; padded = next power of two."""
    shift = int(sample_rate * frame_shift_ms * 0.001)
    size = int(sample_rate * frame_length_ms * 0.001)
    return shift, size, 1 << (size - 1).bit_length()


def num_frames(num_samples: int, size: int, shift: int) -> int:
    """snip_edges=True framing: only whole windows."""
    return 0 if num_samples < size else 1 + (num_samples - size) // shift


def mel_scale(f):
    return 1127.0 * torch.log(1.0 + f / 700.0)


def mel_banks(num_bins: int, padded: int, sample_rate: float, low_freq: float = 20.0, high_freq: float = 0.0) -> Tensor:
    """[num_bins, padded // 2 + 1] triangular filters, equally spaced on the mel axis, unit peak; last column zero."""
    nfft_bins = padded // 2
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    bin_width = sample_rate / padded
    mel_lo, mel_hi = 1127.0 * math.log(1.0 + low_freq / 700.0), 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1.0) * delta, mel_lo + (b + 2.0) * delta
    mel = mel_scale(bin_width * torch.arange(nfft_bins, dtype=torch.float32)).unsqueeze(0)
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    banks = torch.max(torch.zeros(1), torch.min(up, down))
    return torch.nn.functional.pad(banks, (0, 1))


def kaldi_fbank(waveform: Tensor, sample_rate: float, num_mel_bins: int = 128, frame_shift_ms: float = 10.0,
                frame_length_ms: float = 25.0, preemphasis: float = 0.97) -> Tensor:
    """One clip: waveform [n] (or [c, n]: channel 0) -> [frames, num_mel_bins] log-mel energies."""
    if waveform.dim() == 2:
        waveform = waveform[0]
    waveform = waveform.float()
    shift, size, padded = window_properties(sample_rate, frame_shift_ms, frame_length_ms)
    m = num_frames(waveform.shape[0], size, shift)
    if m == 0:
        return torch.empty(0, num_mel_bins)
    frames = waveform.as_strided((m, size), (shift, 1)).clone()
    frames = frames - frames.mean(dim=1, keepdim=True)                                   # remove_dc_offset
    prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)                             # replicate-padded shift
    frames = frames - preemphasis * prev
    frames = frames * torch.hann_window(size, periodic=False).unsqueeze(0)
    frames = torch.nn.functional.pad(frames, (0, padded - size))
    power = torch.fft.rfft(frames, dim=1).abs().pow(2.0)
    mel = power @ mel_banks(num_mel_bins, padded, sample_rate).t()
    return torch.max(mel, torch.tensor(EPS)).log()


def draw_mask(size: int, mask_param: int, generator: Optional[torch.Generator] = None) -> Tuple[int, int]:
    """torchaudio.functional.mask_along_axis: [start, end) of one mask along an axis of length `size`."""
    value = torch.rand(1, generator=generator) * mask_param
    min_value = torch.rand(1, generator=generator) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


def spectrogram_item(waveform: Tensor, sample_rate: float, max_len: int, num_mel_bins: int = 128,
                     norms: Sequence[float] = (), zero_mean_wf: bool = True,
                     freq_mask: Optional[Tuple[int, int]] = None, time_mask: Optional[Tuple[int, int]] = None) -> Tensor:
    """What the dataset hands the model for one clip (image_audio.py:183-207): [max_len, num_mel_bins]."""
    waveform = waveform.float()
    if zero_mean_wf:
        waveform = waveform - waveform.mean()
    x = kaldi_fbank(waveform, sample_rate, num_mel_bins)[:max_len]
    if x.shape[0] < max_len:
        x = torch.nn.functional.pad(x, (0, 0, 0, max_len - x.shape[0]))
    if len(norms) == 2:
        x = (x - norms[0]) / norms[1]
    if freq_mask is not None:
        x[:, freq_mask[0]:freq_mask[1]] = 0.0
    if time_mask is not None:
        x[time_mask[0]:time_mask[1], :] = 0.0
    return x
