# Isolated launches of the small-map layers of C2 (B = 32 and the reference scripts' B = 10) per chunk depth of the small tiles
# (YNET_CONV_SMALL_CC = 8: round 3's setting; 16 / 32: deeper chunks, all with the flat DMA items of round 4):
#   gpurun -- 'bash tools/ab_small.sh'
for B in 32 10; do
for cc in 8 16 32; do
echo "== B=$B YNET_CONV_SMALL_CC=$cc"
for sh in $B,8,8,64,128,3 $B,8,8,128,128,3 $B,8,8,130,130,3 $B,16,16,64,64,3 $B,16,16,128,64,3 $B,16,16,130,65,3 $B,32,32,64,64,3 $B,32,32,64,32,3 $B,32,32,96,64,3 $B,64,64,32,64,3; do
  YNET_CONV_SMALL_CC=$cc python tools/conv_bench.py --shape $sh --iters 50 2>&1 | tail -1
done
YNET_CONV_SMALL_CC=$cc python tools/conv_bench.py --shape $B,16,16,64,64,3 --mask 1 --iters 50 2>&1 | tail -1
YNET_CONV_SMALL_CC=$cc python tools/conv_bench.py --shape $B,8,8,128,128,3 --mask 1 --iters 50 2>&1 | tail -1
done
done
