"""Cross-check vectors for oracle/fbank_cpu.py from an INDEPENDENT Kaldi-compatible implementation that is importable in the build
container: `transformers.audio_utils.spectrogram` -- the numpy fallback HuggingFace's ASTFeatureExtractor uses in place of
`torchaudio.compliance.kaldi.fbank` when torchaudio is missing (transformers/models/audio_spectrogram_transformer/
feature_extraction_audio_spectrogram_transformer.py, `_extract_fbank_features`), called with the reference's parameters
(cvap/data/image_audio.py:119-126: hanning window, 25 ms / 10 ms frames, pre-emphasis 0.97, DC removal, power spectrum, HTK-mel bank
from 20 Hz to Nyquist, natural log with the float32-epsilon floor, no dither).

This does NOT pin the oracle to torchaudio 0.8.1 itself (absent from the image and from /root/reference): the oracle's header keeps
saying so.  It shows that two independent restatements of Kaldi's compute-fbank-feats agree.

    python tests/golden/make_fbank_crosscheck.py        # writes tests/golden/fbank_crosscheck.npz
"""
import os
import warnings

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

HERE = os.path.dirname(os.path.abspath(__file__))


def hf_fbank(wav: np.ndarray, sr: int, num_mel_bins: int) -> np.ndarray:
    shift, size = int(sr * 0.010), int(sr * 0.025)
    padded = 1 << (size - 1).bit_length()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mel = mel_filter_bank(num_frequency_bins=padded // 2 + 1, num_mel_filters=num_mel_bins, min_frequency=20, max_frequency=sr // 2,
                              sampling_rate=sr, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    win = window_function(size, "hann", periodic=False)
    return spectrogram(wav, win, frame_length=size, hop_length=shift, fft_length=padded, power=2.0, center=False, preemphasis=0.97,
                       mel_filters=mel, log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T.astype(np.float64)


def main():
    rng = np.random.default_rng(1213)
    out = {"transformers_version": np.array(transformers.__version__)}
    for tag, sr, dur, bins in (("16k", 16000, 1.27, 128), ("44k", 44100, 0.53, 128), ("16k_64", 16000, 0.61, 64), ("22k", 22050, 0.40, 128)):
        t = np.arange(int(sr * dur)) / sr
        chirp = np.sin(2 * np.pi * (200.0 + 3000.0 * t) * t)
        wav = (0.4 * chirp + 0.2 * rng.standard_normal(t.shape) * np.exp(-3.0 * t) + 0.03).astype(np.float32)
        wav[: sr // 50] = 0.0                      # a silent stretch: frames on the epsilon floor
        out[f"{tag}_wav"] = wav
        out[f"{tag}_sr"] = np.array(sr)
        out[f"{tag}_bins"] = np.array(bins)
        out[f"{tag}_fbank"] = hf_fbank(wav, sr, bins)
    np.savez_compressed(os.path.join(HERE, "fbank_crosscheck.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
