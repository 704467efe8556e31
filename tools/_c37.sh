O=gpurun_out/c37; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
for R in 256 512 1024; do timeout 120 $KB w4a16 28672 4096 $R 0:4 0:8; timeout 120 $KB w4a16 4096 14336 $R 0:4 0:8; done > $O/w4.log 2>&1
timeout 120 $KB w4a16 28672 4096 300,700,512,100,900,512,400,672 0:4 0:8 >> $O/w4.log 2>&1
cat $O/w4.log
timeout 1500 python -m pytest tests/test_moe_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
