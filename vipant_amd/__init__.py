"""vipant_amd -- MI355X (gfx950) native hot path of VIP-ANT's bimodal contrastive training step.

Host side mirrors the reference's operator API (`cvap.module`, `cvap.model`, `train.py`); all numerics run in
hand-written HIP kernels behind the C ABI in include/vipant_hip.h (vipant_amd/lib/libvipant_hip.so).
"""
__version__ = "0.1.0"
