"""ctypes binding of tools/probes/libvipant_probes.so -- measurement probes that are NOT part of the product library
(`vipant_amd.build.build_probes()` compiles it; `__graft_entry__.build()` calls that so the file travels to the GPU box)."""
import ctypes as C
import os

import torch  # noqa: F401  (the HIP runtime torch ships must be the one the probe binds to, as for libvipant_hip.so)

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libvipant_probes.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            raise RuntimeError(f"{PATH} not found: run `python -c 'from vipant_amd import build; build.build_probes()'`")
        _lib = C.CDLL(PATH)
        _lib.probe_comm_shadow.restype = C.c_int32
        _lib.probe_comm_shadow.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_float, C.c_void_p]
    return _lib


def comm_shadow(src_ptr, dst_ptr, nbytes, nwg, min_us, stream):
    """`nwg` workgroups copy nbytes (16-byte aligned) src -> dst on `stream`, each holding its CU for at least min_us."""
    rc = lib().probe_comm_shadow(src_ptr, dst_ptr, nbytes, nwg, min_us, stream)
    if rc != 0:
        raise RuntimeError(f"probe_comm_shadow failed with code {rc}")


class ShadowGradSync:
    """Drop-in for vipant_amd.parallel.GradSync on ONE GPU: wherever the trainer would start a bucket's all-reduce, the stand-in
    kernel runs on a side stream instead (nwg workgroups x min_us per 28.4 MB bucket).  Install with `install(monitor, ...)`."""

    def __init__(self, nwg=0, min_us=0.0, overlap="block"):
        self.nwg, self.min_us, self.overlap = int(nwg), float(min_us), overlap
        self.stream = None
        self.dst = None
        self.held = []

    def _go(self, flat):
        if self.nwg <= 0 or not flat.is_cuda:
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=flat.device)
        self.stream.wait_stream(torch.cuda.current_stream(flat.device))
        if self.dst is None or self.dst.numel() < flat.numel():
            self.dst = torch.empty_like(flat)
        nbytes = flat.numel() * flat.element_size() // 16 * 16
        with torch.cuda.stream(self.stream):
            # min_us is quoted for one block's bucket of the ViT-B tower (28.4 MB); other sizes hold in proportion
            comm_shadow(flat.data_ptr(), self.dst.data_ptr(), nbytes, self.nwg, self.min_us * max(nbytes / 28.4e6, 0.05),
                        self.stream.cuda_stream)
        flat.record_stream(self.stream)

    def reduce_async(self, flat, views=None, params=None):
        if self.overlap == "step":
            self.held.append(flat)
        else:
            self._go(flat)

    def reduce_params(self, params):
        pass

    def wait(self):
        if self.held:
            held, self.held = self.held, []
            self._go(torch.cat(held) if len(held) > 1 else held[0])
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


def install(mon, sync):
    """Make the trainer `mon` hand its gradient buckets to `sync` (a ShadowGradSync)."""
    mon.grad_sync = sync
    for head in (mon.model.audio_head, mon.model.image_head, mon.model.text_head):
        if head is not None and hasattr(head, "encoder"):
            head.encoder.grad_sync = sync
    return sync
