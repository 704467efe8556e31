O=gpurun_out/c38; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
for R in 512; do timeout 120 $KB w4a16 28672 4096 $R 0:4 0:8 16:4 16:8; timeout 120 $KB w4a16 4096 14336 $R 0:4 0:8; done > $O/w4.log 2>&1
cat $O/w4.log
