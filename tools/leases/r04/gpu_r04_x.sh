#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_x
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 600 python3 -m pytest tests/test_quant_extra_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python3 - <<'PY' 2>&1 | grep -v amdgpu
import torch, sgl_kernel
dev = "cuda:0"
x = torch.randn(4096, 4096, dtype=torch.bfloat16, device=dev)
q = torch.empty(4096, 4096, dtype=torch.float8_e4m3fn, device=dev)
s1 = torch.zeros(1, dtype=torch.float32, device=dev)
for _ in range(5): sgl_kernel.sgl_per_tensor_quant_fp8(x, q, s1, False)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): sgl_kernel.sgl_per_tensor_quant_fp8(x, q, s1, False)
g.replay(); torch.cuda.synchronize()
for rep in range(3):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"per_tensor_quant dynamic 4096x4096: {ms*1e3:.1f} us  {x.numel()*5/ms/1e6:.0f} GB/s")
PY
