for d in 0 1; do
echo "== YNET_WGRAD_DMA=$d"
for sh in 32,256,256,14,32,3 32,128,128,32,32,3 32,64,64,32,64,3 32,64,64,64,64,3 32,32,32,64,64,3 32,16,16,64,64,3; do
  YNET_WGRAD_DMA=$d python tools/conv_bench.py --wgrad --mask 1 --shape $sh --iters 30 2>&1 | tail -1
done
done
