O=gpurun_out/c43; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
timeout 120 $KB w4a16 28672 4096 16 0:1 0:9 0:1 0:9 > $O/w4.log 2>&1
timeout 120 $KB w4a16 4096 14336 16 0:1 0:9 >> $O/w4.log 2>&1
timeout 120 $KB w4a16 28672 4096 4 0:1 0:9 >> $O/w4.log 2>&1
cat $O/w4.log
