"""flash_mla_prefill timing (bench.py's shape: 16 sequences x 512 new tokens over 4096 cached keys, H = 128, causal, and a
short-chunk shape). Under LD_PRELOAD of build/libsglk_probes.so the MLA tile-loop variants listed in MLA_VARIANTS
(216 = QK^T on 16x16x32, 232 = on 32x32x16) are timed interleaved."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel

dev = "cuda"
try:
    setv = ctypes.CDLL(None).sglk_debug_set_mla_variant
except AttributeError:
    setv = None
variants = [int(v) for v in os.environ.get("MLA_VARIANTS", "0").split(",")] if setv else [0]
page = 64


def case(pb, psq, psk, H=128):
    pcache = torch.randn(pb * psk // page, page, 576, device=dev, dtype=torch.bfloat16)
    ptable = torch.arange(pb * psk // page, device=dev, dtype=torch.int32).view(pb, psk // page)
    pq = torch.randn(pb * psq, H, 576, device=dev, dtype=torch.bfloat16)
    pqn, pqp = pq[..., :512], pq[..., 512:].contiguous()
    pcu = torch.arange(pb + 1, device=dev, dtype=torch.int32) * psq
    psl = torch.full((pb,), psk, device=dev, dtype=torch.int32)
    pws = torch.empty(1, device=dev, dtype=torch.uint8)
    f = lambda: sgl_kernel.flash_mla_prefill(pqn, pqp, pcache, pcu, psl, psq, ptable, pws, 576 ** -0.5, True)
    flop = 2.0 * pb * H * (576 + 512) * (psq * (psk - psq) + psq * (psq + 1) / 2)
    for rnd in range(3):
        for v in variants:
            if setv:
                setv(v)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ms = sorted(ts)[2]
            print(f"mla_prefill[{v}] {pb}x{psq} over {psk} H={H}: {ms:.4f} ms  {flop / ms / 1e9:.1f} TFLOP/s", flush=True)


case(16, 512, 4096)
case(8, 128, 8192)
case(4, 64, 2048)
case(2, 48, 2048)
case(4, 256, 2048, H=16)
