"""Cycle stamps of waves 1 and 3 of mha_bwd_wide_kernel (debug build, tools/build_stamps.sh; stamps are global stores, the counted
vmcnt waits are slightly off in this build: read the shape)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd import ops, _ffi
b, S, H = 512, 316, 12
qkv = (torch.randn(b * S, 3 * H * 64, device="cuda:0") * 0.5).to(torch.bfloat16)
out, lse = ops.mha_fwd(qkv, b, S, H, False)
dout = torch.randn_like(out)
for _ in range(3):
    ops.mha_bwd(qkv, out, dout, lse, b, S, H, False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib = _ffi.lib()
lib.vipant_debug_attnw_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.vipant_debug_attnw_stamps(buf))
for name, o in (("wave 1 (three key blocks)", 64), ("wave 3 (two key blocks + dQ)", 96)):
    v = list(buf)[o:o + 32]
    print(name, ": switch", v[1] - v[0], " steps", [v[2 + u] - v[1 + u] for u in range(10)])
    print("   step 5: requests + operand fragments", v[20] - v[6], " regions", v[21] - v[20], " dQ store", v[22] - v[21], " wait + barrier + stats", v[7] - v[22])
    print("   tail (dQ of the last step)", v[12] - v[11], " dK / dV stores", v[13] - v[12], " end wait + barrier", v[14] - v[13], " total", v[14] - v[0])
