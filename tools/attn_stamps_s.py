"""Cycle stamps of one wave of mha_bwd1s_kernel (debug build, tools/build_stamps.sh; stamps are global stores, so the counted
vmcnt waits of the real build are slightly off in this one: read the shape, not the last per cent)."""
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
buf = (ctypes.c_ulonglong * 64)()
lib = _ffi.lib()
lib.vipant_debug_attn_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.vipant_debug_attn_stamps(buf))
v = list(buf)
print("switch", v[1] - v[0])
for u in range(10):
    print("step", u, v[2 + u] - (v[1 + u] if u else v[1]))
print("step 5: body", v[20] - v[6], "wait+barrier+stats", v[7] - v[20])
print("tail (A stage, dQ of the last step)", v[12] - v[11], "barrier", v[13] - v[12], "dK/dV staging+stores", v[14] - v[13], "end wait+barrier", v[15] - v[14])
print("total", v[15] - v[0])
