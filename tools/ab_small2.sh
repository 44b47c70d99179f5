for env in "A=1" "YNET_KSPLIT_ITEMS=1024 YNET_KSPLIT_TARGET=1024" "YNET_CONV_DMA_R1=1" "YNET_CONV_DMA_R1=1 YNET_KSPLIT_ITEMS=1024 YNET_KSPLIT_TARGET=1024" "YNET_CONV_DMA_R1=1 YNET_KSPLIT_ITEMS=2048 YNET_KSPLIT_TARGET=2048"; do
echo "== $env"
for sh in 32,32,32,64,64,3 32,32,32,32,64,3 32,32,32,96,64,3 32,32,32,64,32,3; do
  env $env python tools/conv_bench.py --shape $sh --iters 50 2>&1 | tail -1
done
env $env python tools/conv_bench.py --shape 32,32,32,64,64,3 --mask 1 --iters 50 2>&1 | tail -1
env $env python tools/conv_bench.py --shape 32,32,32,64,96,3 --mask 1 --iters 50 2>&1 | tail -1
done
