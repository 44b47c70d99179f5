# A/B of library builds within one box: tools/ab/lib_prev.so (previous commit) against the in-tree build
for rep in 1 2; do
for lib in tools/ab/lib_prev.so ""; do
  echo "== lib=${lib:-current}"
  for sh in 32,256,256,32,32,3 32,256,256,32,48,3 32,128,128,32,64,3 32,128,128,32,32,3 32,64,64,64,96,3 32,64,64,64,64,3; do
  YNET_HIP_LIB=$lib python tools/conv_bench.py --shape $sh --mask 1 --iters 40 2>&1 | tail -1
  done
done
done
