for sh in 32,256,256,14,32,3 32,128,128,32,32,3 32,64,64,32,64,3 32,64,64,64,64,3 32,32,32,64,64,3 32,16,16,64,64,3; do
  python tools/conv_bench.py --wgrad --mask 1 --shape $sh --iters 30 2>&1 | tail -1
done
