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
print("top wait + request", v[2] - v[0])
names = ["QK(0)", "PV(0)+QK(1)", "PV(1)+QK(2)", "store+", "PV(2)+QK(3)", "PV(3)+QK(4)", "store+", "PV(4)", "exchange+store"]
for i, nm in enumerate(names):
    print("%-16s %6d" % (nm, v[4 + i] - v[3 + i]))
print("total", v[12] - v[0])
