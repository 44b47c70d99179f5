mkdir -p gpurun_out/b10r
run() { tag=$1; shift; env "$@" python bench.py --batch 10 --steps 100 > gpurun_out/b10r/$tag.json 2>gpurun_out/b10r/$tag.err; }
run base A=1
run r1dma YNET_CONV_DMA_R1=1
run r2min128 YNET_CONV_R2_MIN=128
run r2min256 YNET_CONV_R2_MIN=256
run both YNET_CONV_DMA_R1=1 YNET_CONV_R2_MIN=256
run r4min512 YNET_CONV_R4_MIN=512
