# in-step A/B of dispatch knobs: sum of the instrumented step's conv launches + the step time (two repetitions each)
run() {
  for rep in 1 2; do
    env "$@" python bench.py --no-cpu-baseline --steps 12 --warmup 3 --layers 2>/tmp/layers.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],3))"
    awk '/LAYER/ {s+=$(NF-3)} END {print "   conv us/step", s}' /tmp/layers.txt
  done
}
for k in "X=0" "YNET_CONV_DMA_R1=1" "YNET_KSPLIT_TARGET=256" "YNET_KSPLIT_TARGET=1024" "YNET_KSPLIT_ITEMS=128" "YNET_KSPLIT_ITEMS=512" "YNET_CONV_NO_KSPLIT=1"; do
  echo "== $k"; run $k
done
