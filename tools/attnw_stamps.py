"""Cycle stamps of one wave of mha_fwd_wide_kernel (debug build with -DVIPANT_ATTN_STAMPS, loaded through VIPANT_HIP_LIB;
tools/build_stamps.sh makes it)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops, _ffi
b, S, H = 512, 316, 12
qkv = (torch.randn(b * S, 3 * H * 64, device="cuda:0") * 0.5).to(torch.bfloat16)
for _ in range(3):
    out, lse = ops.mha_fwd(qkv, b, S, H, False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = _ffi.lib()
lib.vipant_debug_attnw_stamps.argtypes = [ctypes.c_void_p]
rc = lib.vipant_debug_attnw_stamps(buf)
v = list(buf)
print("rc", rc)
print("dma issue", v[1] - v[0], "image wait", v[2] - v[1])
for it in range(5):
    print("half-unit", it, "QK", v[4 + 3 * it] - v[3 + 3 * it], "softmax+PV", v[5 + 3 * it] - v[4 + 3 * it], "store/loop", (v[6 + 3 * it] if it < 4 else v[18]) - v[5 + 3 * it])
print("exchange + last store", v[19] - v[18], "total", v[19] - v[0])
