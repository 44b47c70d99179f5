for x4 in 0 1 0 1; do
  echo "== X4=$x4"
  for sh in 32,256,256,32,32,3 32,256,256,48,32,3 32,128,128,64,32,3 32,64,64,96,64,3 32,256,256,32,16,3; do
    YNET_CONV_X4=$x4 python tools/conv_bench.py --shape $sh --iters 40 2>&1 | tail -1
  done
  YNET_CONV_X4=$x4 python tools/conv_bench.py --shape 32,256,256,32,32,3 --mask 1 --iters 40 2>&1 | tail -1
  YNET_CONV_X4=$x4 python tools/conv_bench.py --shape 32,256,256,32,48,3 --mask 1 --iters 40 2>&1 | tail -1
done
