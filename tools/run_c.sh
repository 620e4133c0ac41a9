export BENCH_LAYERS="vgg 128->128,vgg 256->256,conv2_1,conv3_1,iconv4"
BENCH_VARIANTS=dma,dma128 python tools/bench_conv.py > gpurun_out/r03d_conv.txt 2>&1
export BENCH_LAYERS="vgg 128->128,vgg 256->256"
for t in probe4 probe8; do
  echo "== $t" >> gpurun_out/r03d_conv.txt
  FALNET_LIB=$PWD/fal_net_amd/libfalnet_hip_$t.so BENCH_NOCHECK=1 BENCH_VARIANTS=dma python tools/bench_conv.py >> gpurun_out/r03d_conv.txt 2>&1
done
echo "== base again" >> gpurun_out/r03d_conv.txt
BENCH_VARIANTS=dma,dma128 python tools/bench_conv.py >> gpurun_out/r03d_conv.txt 2>&1
cat gpurun_out/r03d_conv.txt
