#!/bin/bash
# final numbers of the e4m3 towers: loss-noise statistics (round-5 library against this tree, same box), the configs[4] A/B (3 rounds),
# its kernel profile, and the RCCL small-collective latency a strip-only InfoNCE would add
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r6}
{
for cfg in "VIPANT_HIP_LIB=vipant_amd/lib/libvipant_hip_r5.so VIPANT_FP8_TN=0" "VIPANT_FP8_TN=1"; do env $cfg python tools/fp8_loss_noise.py 160 L12 2>&1 | tail -1; done
for cfg in "VIPANT_HIP_LIB=vipant_amd/lib/libvipant_hip_r5.so VIPANT_FP8_TN=0" "VIPANT_FP8_TN=1"; do env $cfg python tools/fp8_loss_noise.py 48 cfg2 2>&1 | tail -1; done
} > gpurun_out/${tag}_fp8_loss_noise.txt 2>&1
cat gpurun_out/${tag}_fp8_loss_noise.txt | cut -c1-330
for r in 1 2 3; do
  for cfg in "VIPANT_FP8_TN=1" "VIPANT_FP8_TN=0" "VIPANT_HIP_LIB=vipant_amd/lib/libvipant_hip_r5.so VIPANT_FP8_TN=0"; do
    env $cfg python bench.py --script at --width 1024 --layers 24 --batch 1024 --fp8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg round $r:', d['ms_per_step'], 'ms/step', d['peak_mem_gb'], 'GB, loss', d['loss'], 'step TFLOP/s', d['step_tflops'])"
  done
done | tee gpurun_out/${tag}_cfg5_ab.txt
python bench.py --script at --width 1024 --layers 24 --batch 1024 --fp8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_cfg5_fp8.json
python bench.py --script at --width 1024 --layers 24 --batch 1024 --fp8 --recompute-mlp --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_cfg5_fp8_recompute.json
python bench.py --script at --width 1024 --layers 24 --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_cfg5_bf16.json
for f in fp8 fp8_recompute bf16; do python -c "import json; d=json.load(open('gpurun_out/${tag}_cfg5_$f.json')); print('$f', d['ms_per_step'], d['peak_mem_gb'])"; done
bash tools/r6_cfg5_prof.sh ${tag} | head -24
python - <<'PY'
import os, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.randn(4096, device="cuda"); out = torch.empty(4096, device="cuda")
for name, fn in (("all_gather_into_tensor 16 KB", lambda: dist.all_gather_into_tensor(out, x)), ("all_reduce 16 KB", lambda: dist.all_reduce(x))):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"RCCL, ONE rank (launch + kernel floor, no peer): {name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call")
dist.destroy_process_group()
PY
