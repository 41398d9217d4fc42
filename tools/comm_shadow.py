"""What does sharing the chip with the gradient all-reduce cost the step?  (profiles/r5_comm_shadow.md)

One process, one GPU, the headline VA step (bench.py's workload).  A stand-in kernel (`probe_comm_shadow`, tools/probes/comm_shadow.hip: N workgroups x 256 threads
copying a block's 28 MB bucket, each holding its CU for at least T us) is launched on the side stream wherever
GradSync.reduce_async would start the RCCL all-reduce.  Settings are interleaved in one process (boxes differ by +-2 %):

    python tools/comm_shadow.py [--steps 8] [--rounds 2] [--quick]

walk:    ticket (default build) | static (VIPANT_GEMM_VARIANT bit 22: the round-4 static-stride tile walk of the NT kernels)
overlap: block (bucket per block, overlapping the backward) | step (one hand-over after the backward)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--cfg5", action="store_true", help="BASELINE configs[4]'s tower instead of the headline step: AT script, audio ViT-L "
                    "(width 1024, 24 blocks), e4m3 contractions; pass --batch 1024.  A block's bucket is 50 MB there: the stand-in holds its "
                    "CUs for min_us x 50 / 28.4 (ShadowGradSync scales with the bucket)")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import probe_lib
    from vipant_amd import _ffi
    from vipant_amd.config import compose
    from vipant_amd.monitor import VAMonitor
    from vipant_amd.module import adjust_learning_rate
    _ffi.call("vipant_device_check")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    b, T, Fq = args.batch, 1024, 128
    ov = ("+running=bimodal worker=CVALP mode=ddp eval=False +model/image=vit_val +model/audio=vit_val +model/text=dummy "
          "+model/loss=ce +optimizer=standard +running/audio=default model.audio.pre_encoder.in_channels=3 "
          "model.audio.pre_encoder.stride=[16,24] model.image.encoder.layers=12 model.audio.width=768 model.audio.encoder.layers=12 "
          f"running.audio.max_len={T} running.audio.num_mel_bins={Fq} running.batch_size={b} running.epochs=1000 "
          "running.save_epoch=False running.save_rate=1e9 running.peep_rate=1000000 running.synthetic_steps=100000 num_gpus=1").split()
    if args.cfg5:
        from vipant_amd.monitor import VALMonitor as VAMonitor          # noqa: F811  (the AT trainer)
        ov = ("+running=trimodal monitor=VALMonitor worker=CVALP mode=ddp eval=False +model/image=vit_val +model/audio=vit_val "
              "+model/text=transformer_val +model/loss=ce_val +optimizer=standard +running/audio=default "
              "model.audio.pre_encoder.in_channels=3 model.audio.pre_encoder.stride=[16,24] running.siamese.alive=True "
              "running.imagine=False model.loss.va=False model.image.encoder.layers=12 model.audio.width=1024 "
              "model.audio.encoder.layers=24 +running.negatives=local running.fp8_gemm=True "
              f"running.audio.max_len={T} running.audio.num_mel_bins={Fq} running.batch_size={b} running.epochs=1000 "
              "running.save_epoch=False running.save_rate=1e9 running.peep_rate=1000000 running.synthetic_steps=100000 num_gpus=1").split()
    cfg = compose(ov)
    cfg.rank = 0
    torch.manual_seed(cfg.seed)
    mon = VAMonitor(cfg, (lambda *_: None), dev)
    mon.total_loss = mon.total_step = mon.total_inst = 0
    mon.start_time = time.time()
    sync = probe_lib.install(mon, probe_lib.ShadowGradSync())      # the stand-in takes GradSync's place; switched per segment below
    g = torch.Generator().manual_seed(1213)
    text = None
    if args.cfg5:
        images, audios, text, _, _ = mon.make_batch(next(iter(mon.dataloader)))
        text = torch.cat([text, text.new_zeros(text.shape[0], 77 - text.shape[1])], dim=1) if text.shape[1] < 77 else text
    else:
        images = torch.randn(b, 3, 224, 224, generator=g).to(dev)
        audios = torch.randn(b, 1, T, Fq, generator=g).to(dev)
    it = [0]

    def run(n):
        for _ in range(n):
            adjust_learning_rate(cfg.optimizer, mon.optimizer, mon.dataloader, it[0] + 10)
            mon.step(images, audios, text)
            it[0] += 1

    def timed(walk, overlap, shadow):
        os.environ["VIPANT_GEMM_VARIANT"] = "4194304" if walk == "static" else "0"
        nwg, _, us = shadow.partition(":")
        sync.nwg, sync.min_us, sync.overlap = int(nwg), float(us or 0.0), overlap
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    run(3)
    shadows = ["0", "8:300", "16:300", "32:300", "64:300", "32:0", "32:900", "8:900", "16:900"]
    settings = [(w, "block", s) for w in ("ticket", "static") for s in shadows]
    settings += [("ticket", "step", s) for s in ("32:300", "32:900")]
    if args.quick:
        settings = [(w, "block", s) for w in ("ticket", "static") for s in ("0", "32:300")]
    if args.cfg5:        # the e4m3 NT kernels and the weight-gradient kernels walk statically: one walk, the shadow sizes that matter
        settings = [("ticket", "block", s) for s in ("0", "16:300", "32:300", "64:300", "32:900")] + [("ticket", "step", "32:300")]
    res = {k: [] for k in settings}
    for r in range(args.rounds):
        for k in settings:
            res[k].append(timed(*k))
            print(k, "%.3f" % res[k][-1], flush=True)
    base = {w: min(res[(w, "block", "0")]) for w in ("ticket", "static") if (w, "block", "0") in res}
    rows = []
    for k in settings:
        best = min(res[k])
        rows.append({"walk": k[0], "overlap": k[1], "shadow": k[2], "ms": [round(x, 3) for x in res[k]], "best_ms": round(best, 3),
                     "vs_no_shadow_pct": round((best / base[k[0]] - 1) * 100, 2)})
    out = {"steps": args.steps, "rounds": args.rounds, "batch": b, "rows": rows}
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
