"""GPU log-mel front-end: the step in front of the hot path.

The reference computes one Kaldi filter bank per clip on CPU dataloader workers
(cvap/data/audio/transform.py:12-35, parameters cvap/data/image_audio.py:119-126), pads / normalises it and applies
SpecAugment masks (image_audio.py:183-207, configs/running/audio/default.yaml).  `KaldiFbank` does the same for a whole
batch of device-resident waveforms in two launches (`vipant_fbank`) and returns the [b, 1, T, F] tensor
`ViTPreEncoder` takes.  File decoding, resampling and random cropping stay with the loader.

The filter tables (Hann window, mel weights) are built once per (sample rate, bins) with torch ops on the device.
SpecAugment start / width draws follow torchaudio.functional.mask_along_axis (two uniform draws per mask, CPU generator),
one (frequency, time) pair per clip as the reference's per-item transform does.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

from . import ops
from ._ffi import VipantError, call, query

F32 = torch.float32


def _tables(sample_rate: float, num_mel_bins: int, frame_shift_ms: float, frame_length_ms: float, device):
    shift = int(sample_rate * frame_shift_ms * 0.001)
    size = int(sample_rate * frame_length_ms * 0.001)
    padded = 1 << (size - 1).bit_length()
    window = torch.hann_window(size, periodic=False, dtype=F32, device=device)
    nyquist = 0.5 * sample_rate
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    mel_lo, mel_hi = mel(20.0), mel(nyquist)
    delta = (mel_hi - mel_lo) / (num_mel_bins + 1)
    b = torch.arange(num_mel_bins, dtype=F32, device=device).unsqueeze(1)
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1.0) * delta, mel_lo + (b + 2.0) * delta
    freqs = (sample_rate / padded) * torch.arange(padded // 2, dtype=F32, device=device)
    melf = (1127.0 * torch.log(1.0 + freqs / 700.0)).unsqueeze(0)
    banks = torch.clamp_min(torch.minimum((melf - left) / (center - left), (right - melf) / (right - center)), 0.0)
    banks = torch.nn.functional.pad(banks, (0, 1)).contiguous()
    nz = banks > 0
    start = nz.int().argmax(dim=1).to(torch.int32)
    length = nz.sum(dim=1).to(torch.int32)
    return shift, size, padded, window, banks, start.contiguous(), length.contiguous()


class KaldiFbank:
    """kaldi.fbank(htk_compat=True, use_energy=False, window_type='hanning', dither=0, frame_shift=10) + dataset glue."""

    def __init__(self, sample_rate: float = 16000.0, num_mel_bins: int = 128, max_len: int = 1000,
                 norms: Sequence[float] = (), zero_mean_wf: bool = True, frame_shift: float = 10.0, frame_length: float = 25.0,
                 preemphasis: float = 0.97, freq_mask_param: int = 0, time_mask_param: int = 0, device="cuda:0"):
        self.device = torch.device(device)
        self.sample_rate, self.num_mel_bins, self.max_len = float(sample_rate), int(num_mel_bins), int(max_len)
        self.norms = tuple(float(v) for v in norms) if len(norms) == 2 else ()
        self.zero_mean_wf, self.preemphasis = bool(zero_mean_wf), float(preemphasis)
        self.freq_mask_param, self.time_mask_param = int(freq_mask_param), int(time_mask_param)
        (self.shift, self.size, self.padded, self.window, self.banks, self.bank_start, self.bank_len) = _tables(
            self.sample_rate, self.num_mel_bins, frame_shift, frame_length, self.device)

    @classmethod
    def from_config(cls, acfg, sample_rate: float = 16000.0, train: bool = True, device="cuda:0"):
        """`acfg` = the reference's `running.audio` node (configs/running/audio/default.yaml)."""
        fm = tm = 0
        if train and getattr(acfg, "transform_fbank", False) and not getattr(acfg, "eval_norms", False):
            for name, params in getattr(acfg, "fbank_transforms", []):
                if name == "FrequencyMasking":
                    fm = int(params[0])
                elif name == "TimeMasking":
                    tm = int(params[0])
        norms = () if getattr(acfg, "eval_norms", False) else tuple(getattr(acfg, "norms", ()) or ())
        return cls(sample_rate, acfg.num_mel_bins, acfg.max_len, norms, getattr(acfg, "zero_mean_wf", True),
                   getattr(acfg, "frame_shift", 10), freq_mask_param=fm, time_mask_param=tm, device=device)

    def num_frames(self, num_samples: int) -> int:
        return 0 if num_samples < self.size else 1 + (num_samples - self.size) // self.shift

    def draw_masks(self, b: int, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
        """[b, 4] int32 (f0, f1, t0, t1); mask_along_axis's draws: value = U*param, min = U*(size - value)."""
        if self.freq_mask_param <= 0 and self.time_mask_param <= 0:
            return None
        rows = []
        for _ in range(b):
            row = []
            for param, size in ((self.freq_mask_param, self.num_mel_bins), (self.time_mask_param, self.max_len)):
                if param <= 0:
                    row += [0, 0]
                    continue
                value = torch.rand(1, generator=generator) * param
                min_value = torch.rand(1, generator=generator) * (size - value)
                start = int(min_value.long())
                row += [start, start + int(value.long())]
            rows.append(row)
        return torch.tensor(rows, dtype=torch.int32)

    def __call__(self, wave: torch.Tensor, nsamples: Optional[torch.Tensor] = None, masks: Optional[torch.Tensor] = None):
        """wave fp32 [b, n] on the device (rows zero-padded to a common n), nsamples int64 [b] valid lengths
        (default: n), masks int32 [b, 4] or None  ->  [b, 1, max_len, num_mel_bins] fp32."""
        if not wave.is_cuda or wave.dtype != F32 or wave.dim() != 2:
            raise VipantError("fbank: wave must be a 2-d float32 device tensor (there is no CPU path)")
        wave = wave.contiguous()
        b, n = wave.shape
        dev = wave.device
        if nsamples is None:
            nsamples = torch.full((b,), n, dtype=torch.int64, device=dev)
        nsamples = nsamples.to(device=dev, dtype=torch.int64).contiguous()
        if masks is not None:
            masks = masks.to(device=dev, dtype=torch.int32).contiguous()
        out = torch.empty((b, 1, self.max_len, self.num_mel_bins), dtype=F32, device=dev)
        ws = ops.scratch("fbank", query("vipant_fbank_workspace_bytes", b), dev)
        mean, std = self.norms if self.norms else (0.0, 0.0)
        call("vipant_fbank", wave.data_ptr(), n, nsamples.data_ptr(), out.data_ptr(), self.window.data_ptr(),
             self.banks.data_ptr(), self.bank_start.data_ptr(), self.bank_len.data_ptr(),
             masks.data_ptr() if masks is not None else None, b, self.max_len, self.num_mel_bins, self.size, self.shift,
             self.padded, self.preemphasis, int(self.zero_mean_wf), float(mean), float(std), ws.data_ptr(), ws.numel(),
             ops._stream())
        return out
