import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vipant_amd.config import compose
from vipant_amd.module import adjust_learning_rate
from vipant_amd.monitor import VAMonitor
B = int(sys.argv[1]); mb = int(sys.argv[2])
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
out = []
for i in range(3):
    adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, i + 10)
    out.append(float(mon.step(images, audios, None).detach()))
print(B, mb, ["%.6f" % v for v in out])
