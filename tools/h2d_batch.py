"""Host-to-device copy of one headline batch (what Monitor.make_batch hands over per step): the PCIe-inclusive figure DESIGN.md section 8 quotes."""
import torch, time
b=512
for pin in (False, True):
    img = torch.randn(b,3,224,224); aud = torch.randn(b,1024,128)
    if pin: img, aud = img.pin_memory(), aud.pin_memory()
    for _ in range(2): img.cuda(non_blocking=True); aud.cuda(non_blocking=True); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): img.cuda(non_blocking=True); aud.cuda(non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(f"H2D of one headline batch (512 x 3x224x224 + 512 x 1024x128 fp32 = {(img.numel()+aud.numel())*4/1e6:.0f} MB), {'pinned' if pin else 'pageable'} host memory: {dt*1e3:.1f} ms = {(img.numel()+aud.numel())*4/dt/1e9:.1f} GB/s")
