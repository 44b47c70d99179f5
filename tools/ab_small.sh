for env in "YNET_CONV_FOLD=0" "YNET_CONV_FOLD=1" "YNET_CONV_FOLD=1 YNET_CONV_DMA_R1=1"; do
echo "== $env"
for sh in 32,8,8,128,128,3 32,8,8,130,130,3 32,8,8,64,128,3 32,16,16,64,64,3 32,16,16,128,64,3 32,16,16,65,130,3 32,32,32,64,64,3 32,32,32,32,64,3 32,32,32,96,64,3; do
  env $env python tools/conv_bench.py --shape $sh --iters 50 2>&1 | tail -1
done
env $env python tools/conv_bench.py --shape 32,16,16,64,64,3 --mask 1 --iters 50 2>&1 | tail -1
env $env python tools/conv_bench.py --shape 32,8,8,128,128,3 --mask 1 --iters 50 2>&1 | tail -1
done
