"""Which ATen fills / copies does one training step of the benchmark configuration issue, and under which autograd node?  (The
per-kernel rocprofv3 summaries of a short run also count the fills of model construction -- 153 momentum buffers, parameter
initialisation -- which is where most of their `FillFunctor` / `copyBuffer` launches come from: a step itself has 6 fills and 5
copies, round 4.)  Usage: python tools/aten_sites.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vipant_amd.config import compose  # noqa: E402
from vipant_amd.module import adjust_learning_rate  # noqa: E402
from vipant_amd.monitor import VAMonitor  # noqa: E402

dev = torch.device("cuda", 0)
b, T, Fq = 512, 1024, 128
ov = ("+running=bimodal worker=CVALP mode=ddp eval=False +model/image=vit_val +model/audio=vit_val +model/text=dummy "
      "+model/loss=ce +optimizer=standard +running/audio=default model.audio.pre_encoder.in_channels=3 "
      "model.audio.pre_encoder.stride=[16,24] "
      f"running.audio.max_len={T} running.audio.num_mel_bins={Fq} running.batch_size={b} running.epochs=1000 running.save_epoch=False "
      "running.save_rate=1e9 running.peep_rate=1000000 running.synthetic_steps=8 num_gpus=1").split()
cfg = compose(ov)
cfg.rank = 0
torch.manual_seed(cfg.seed)
mon = VAMonitor(cfg, (lambda *_: None), dev)
mon.total_loss = mon.total_step = mon.total_inst = 0
import time
mon.start_time = time.time()
g = torch.Generator().manual_seed(1213)
images = torch.randn(b, 3, 224, 224, generator=g).to(dev)
audios = torch.randn(b, 1, T, Fq, generator=g).to(dev)


def one_step(i):
    adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, i + 10)
    return mon.step(images, audios, None)


for i in range(3):
    one_step(i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    one_step(3)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::add_", "aten::add",
                   "aten::mul", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::empty_strided"):
        chain, q = [], ev.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name[:48])
            q = q.cpu_parent
        sites[(ev.name, " < ".join(chain), str(ev.input_shapes)[:60])] += 1
names = collections.Counter(ev.name for ev in prof.events())
print({k: v for k, v in names.items() if v >= 3 and k.startswith("aten::")})
for (name, chain, shp), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n:4d}  {name:18s} {shp:60s} {chain}")
