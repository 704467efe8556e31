#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_y
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests/test_moe_gpu.py -m gpu -q -x -n 4 2>&1 | tail -8
timeout 600 python3 - <<'PY' 2>&1 | grep -v amdgpu
import torch, sgl_kernel, sys
sys.path.insert(0, "tests")
dev = "cuda:0"
E, topk, Hd, I, gs = 8, 2, 4096, 14336, 128
g = torch.Generator().manual_seed(0)
w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), dtype=torch.uint8, generator=g).to(dev)
w2 = torch.randint(0, 256, (E, Hd, I // 2), dtype=torch.uint8, generator=g).to(dev)
s1 = (torch.rand(E, 2 * I, Hd // gs, generator=g) * 0.01).to(torch.bfloat16).to(dev)
s2 = (torch.rand(E, Hd, I // gs, generator=g) * 0.01).to(torch.bfloat16).to(dev)
for T in (1, 4, 16, 32, 64, 128, 256):
    x = (torch.randn(T, Hd, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    tw = torch.rand(T, topk, generator=g).to(dev)
    ti = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(T)]).to(torch.int32).to(dev)
    f = lambda: sgl_kernel.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2)
    for _ in range(3): f()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): f()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    print(f"fused_experts int4 mixtral T={T}: {best*1e3:.1f} us")
PY
