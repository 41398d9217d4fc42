import os, sys, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vipant_amd.config import compose
from vipant_amd.monitor import VAMonitor
from vipant_amd.module import adjust_learning_rate
ov = ("+running=bimodal worker=CVALP mode=dp eval=False +model/image=vit_val +model/audio=vit_val +model/text=dummy "
      "+model/loss=ce +optimizer=standard +running/audio=default model.audio.pre_encoder.in_channels=3 "
      "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=2 running.audio.max_len=1024 "
      "running.audio.num_mel_bins=128 running.batch_size=64 running.epochs=1000 running.save_epoch=False "
      "running.save_rate=1e9 running.peep_rate=1000000 running.synthetic_steps=4 num_gpus=1").split()
cfg = compose(ov); cfg.rank = 0
torch.manual_seed(1)
mon = VAMonitor(cfg, lambda *_: None, torch.device("cuda:0"))
images = torch.randn(64, 3, 224, 224, device="cuda:0"); audios = torch.randn(64, 1, 1024, 128, device="cuda:0")
for i in range(2):
    adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, i + 10); mon.step(images, audios, None)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, 12); mon.step(images, audios, None)
torch.cuda.set_sync_debug_mode("default")
print("done")
