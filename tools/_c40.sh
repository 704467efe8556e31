O=gpurun_out/c40; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
timeout 120 $KB w4a16 28672 4096 16 0:1 0:2 > $O/w4.log 2>&1
timeout 120 $KB w4a16 4096 14336 16 0:1 0:2 >> $O/w4.log 2>&1
timeout 120 $KB w4a16 28672 4096 1 0:1 >> $O/w4.log 2>&1
timeout 120 $KB w4a16 28672 4096 32 0:2 >> $O/w4.log 2>&1
timeout 120 $KB w4a16 4096 14336 32 0:2 >> $O/w4.log 2>&1
cat $O/w4.log
