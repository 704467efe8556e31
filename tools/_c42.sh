O=gpurun_out/c42; mkdir -p $O
KB=sgl-kernel-xpu_amd/build/kbench
timeout 300 $KB stream 229376 2048 > $O/stream.log 2>&1
cat $O/stream.log
