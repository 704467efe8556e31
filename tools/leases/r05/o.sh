#!/bin/bash
# r05 lease o: fp8_scaled_mm / int8_scaled_mm / QServe W4A8 above 128 rows as ONE launch (whole tiles, then the half tiles of the
# last partial round): parity, timing next to the round-4 library on one box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_o
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_gemm_gpu.py tests/test_qserve_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
cat > /tmp/scaled_mm_bench.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
dev = "cuda"
M, N, K = 4096, 14336, 4096
FP8 = torch.float8_e4m3fn
a8 = torch.randn(M, K, device=dev).clamp(-3, 3).to(FP8); b8 = torch.randn(N, K, device=dev).clamp(-3, 3).to(FP8).t()
sa = torch.rand(M, 1, device=dev) * 0.01 + 0.001; sb = torch.rand(1, N, device=dev) * 0.01 + 0.001
ai = torch.randint(-127, 128, (M, K), device=dev, dtype=torch.int8); bi = torch.randint(-127, 128, (N, K), device=dev, dtype=torch.int8).t()
def timeit(f, it=20):
    for _ in range(30): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(it): f()
        en.record(); torch.cuda.synchronize(); ts.append(st.elapsed_time(en) / it)
    return sorted(ts)[2]
t1 = timeit(lambda: sgl_kernel.fp8_scaled_mm(a8, b8, sa, sb, torch.bfloat16))
t2 = timeit(lambda: sgl_kernel.int8_scaled_mm(ai, bi, sa, sb, torch.bfloat16))
print(f"fp8_scaled_mm {t1:.4f} ms {2.0*M*N*K/t1/1e9:.1f} TFLOP/s | int8_scaled_mm {t2:.4f} ms {2.0*M*N*K/t2/1e9:.1f} TOP/s")
PY
for rep in 1 2 3; do
  echo "== r05"; timeout 200 python3 /tmp/scaled_mm_bench.py 2>&1 | grep -v amdgpu
  echo "== r04"; LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes_r04.so timeout 200 python3 /tmp/scaled_mm_bench.py 2>&1 | grep -v "amdgpu\|ld.so"
done | tee $OUT/scaled_mm.log
for rep in 1 2; do
  echo "== r05"; timeout 300 python3 tools/qserve_bench.py 4096 2>&1 | grep -v amdgpu
  echo "== r04"; LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes_r04.so timeout 300 python3 tools/qserve_bench.py 4096 2>&1 | grep -v "amdgpu\|ld.so"
done | tee $OUT/qserve4096.log
