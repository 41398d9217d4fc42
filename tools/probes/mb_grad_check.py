import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd.config import compose
from vipant_amd.monitor import VAMonitor
B = 64
def grads(mb):
    ov = ("+running=bimodal worker=CVALP mode=dp eval=False +model/image=vit_val +model/audio=vit_val +model/text=dummy "
          "+model/loss=ce +optimizer=standard +running/audio=default model.audio.pre_encoder.in_channels=3 "
          "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 model.audio.encoder.layers=2 running.audio.max_len=1024 "
          f"running.audio.num_mel_bins=128 running.batch_size={B} running.epochs=1000 running.save_epoch=False "
          f"running.save_rate=1e9 running.peep_rate=1000000 running.synthetic_steps=4 num_gpus=1 running.micro_batch={mb}").split()
    cfg = compose(ov); cfg.rank = 0
    torch.manual_seed(cfg.seed)
    mon = VAMonitor(cfg, lambda *_: None, torch.device("cuda:0"))
    g = torch.Generator().manual_seed(1213)
    images = torch.randn(B, 3, 224, 224, generator=g).cuda(); audios = torch.randn(B, 1, 1024, 128, generator=g).cuda()
    mon.optimizer.zero_grad(set_to_none=True)
    if mb:
        loss = mon._forward_backward_micro(images, audios, None, mb)
    else:
        loss = mon.model(images, audios, None); loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().clone() for k, p in mon.model.named_parameters() if p.requires_grad and p.grad is not None}
l0, g0 = grads(0)
l1, g1 = grads(16)
print(l0, l1, len(g0), len(g1))
rows = []
for k in g0:
    a, b = g0[k].double(), g1[k].double()
    rows.append((float((a - b).norm() / (a.norm() + 1e-30)), float(b.norm() / (a.norm() + 1e-30)), k))
rows.sort(reverse=True)
for r in rows[:12]: print("%.3e  ratio %.4f  %s" % r)
