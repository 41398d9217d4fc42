"""How far is the e4m3 tower's loss from the bf16 tower's on the same weights and inputs -- as a statistic over random batches, so that
two quantisation schemes can be told apart from one draw's luck (the golden fixtures are one batch each).  The e2e_L12 fixture's
shape and weights (12 blocks, 32 clips of 256 x 64), K random batches; prints mean / median / max of |loss_e4m3 - loss_bf16| and of
the features' max relative error.  Run once per library (VIPANT_HIP_LIB) / switch (VIPANT_FP8_TN) to compare.
    python tools/fp8_loss_noise.py [K] [tag]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from types import SimpleNamespace as NS
import gen
import vipant_amd.module as M
from vipant_amd import ops
from test_model_gpu import audio_cfg
DEV = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = sys.argv[2] if len(sys.argv) > 2 else "L12"
L, b, T, Fq = (12, 32, 256, 64) if tag == "L12" else (12, 64, 1024, 128)
head = M.build_audio_head(audio_cfg(T, Fq, L))
S = head.misc.positional_embedding.shape[0]
head.load_state_dict(gen.det_weights(f"e2e/{'L12' if tag == 'L12' else 'cfg2'}", gen.vit_head_shapes(768, L, 512, S)), strict=True)
lhead = M.build_loss_head(NS(name="CELossHead", layers=[], scaling=True, scale_max=None))
head, lhead = head.to(DEV).train(), lhead.to(DEV).train()
rows = os.environ.get("ROWS", "1") == "1"
head.encoder.last_block_rows = rows
dl, df, dg = [], [], []
for k in range(K):
    g = torch.Generator(device=DEV); g.manual_seed(1000 + k)
    aud = torch.randn(b, 1, T, Fq, generator=g, device=DEV)
    img = ops.l2_normalize(torch.randn(b, 512, generator=g, device=DEV))
    res = []
    for fp8 in (False, True):
        head.encoder.fp8 = fp8
        for p in head.parameters(): p.grad = None
        feat = head(aud, normalized=True)
        loss = lhead(img, feat, None, normalized=True)
        loss.backward()
        res.append((float(loss), feat.detach().clone(), head.encoder.resblocks[0].mlp.c_fc.weight.grad.clone()))
    dl.append(abs(res[1][0] - res[0][0]))
    df.append(float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max()))
    dg.append(float((res[1][2] - res[0][2]).norm() / res[0][2].norm()))
t = torch.tensor(dl); f = torch.tensor(df); gr = torch.tensor(dg)
print(f"{tag} rows={rows} lib={os.environ.get('VIPANT_HIP_LIB', 'tree')} FP8_TN={os.environ.get('VIPANT_FP8_TN', '1')}: |loss_e4m3 - loss_bf16| over {K} batches: "
      f"mean {float(t.mean()):.2e} median {float(t.median()):.2e} max {float(t.max()):.2e} rms {float((t * t).mean().sqrt()):.2e}; feature max-rel mean {float(f.mean()):.3e}; "
      f"d c_fc.weight(block 0) rel-L2 mean {float(gr.mean()):.3e}")
