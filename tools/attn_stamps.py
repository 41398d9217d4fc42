"""Cycle stamps of one wave of the round-3 resident-image backward (tools/probes/mha_bwd1_resident.hip.txt; debug build with -DVIPANT_ATTN_STAMPS, loaded through VIPANT_HIP_LIB).  attn_stamps_s.py is the shipped streamed kernel's."""
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
rc = lib.vipant_debug_attn_stamps(buf)
v = list(buf)
print("rc", rc)
print("prologue issue", v[1] - v[0], "image wait", v[27] - v[1], "delta+sync", v[2] - v[27], "setup", v[3] - v[2])
for u in range(10):
    print("step", u, "body", v[5 + 2 * u] - v[4 + 2 * u], "barrier+loop", (v[6 + 2 * u] if u < 9 else v[24]) - v[5 + 2 * u])
print("tail dq", v[25] - v[24], "dk/dv stores", v[26] - v[25], "total", v[26] - v[0])
if v[30]:
    print("step 5: head", v[30] - v[14], "regions", [v[31 + i] - v[30 + i] for i in range(5)], "store+copy", v[15] - v[35])
