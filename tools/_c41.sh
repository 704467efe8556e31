O=gpurun_out/c41; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
timeout 120 $KB w4a16 28672 4096 16 0:1 128:1 256:1 384:1 16:1 > $O/w4.log 2>&1
cat $O/w4.log
