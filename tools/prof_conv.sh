for sh in 32,256,256,32,32,3 32,256,256,48,32,3 32,128,128,64,32,3 32,64,64,96,64,3; do
  YNET_HIP_LIB=tools/ab/lib_prof.so python tools/conv_bench.py --shape $sh --iters 3 2>&1 | tail -2
done
YNET_HIP_LIB=tools/ab/lib_prof.so python tools/conv_bench.py --shape 32,256,256,32,32,3 --mask 1 --iters 3 2>&1 | tail -2
YNET_HIP_LIB=tools/ab/lib_prof.so python tools/conv_bench.py --shape 32,256,256,32,32,3 --srcbs0 --iters 3 2>&1 | tail -2
YNET_HIP_LIB=tools/ab/lib_prof.so python tools/conv_bench.py --shape 32,256,256,32,32,3 --nostore --iters 3 2>&1 | tail -2
