O=gpurun_out/c39; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
timeout 120 $KB w4a16 28672 4096 16 0:1 32:1 64:1 96:1 16:1 112:1 0:2 96:2 > $O/w4.log 2>&1
timeout 120 $KB w4a16 4096 14336 16 0:1 32:1 64:1 96:1 16:1 112:1 0:2 96:2 >> $O/w4.log 2>&1
timeout 120 $KB stream 28672 2048 >> $O/w4.log 2>&1
cat $O/w4.log
