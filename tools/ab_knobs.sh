# in-step A/B of dispatch knobs: conv launches of the instrumented step (ConvTimer, serialized streams) and the step time;
# two repetitions each -- compare the smaller values (some runs land in a slower clock state)
#   gpurun -- 'bash tools/ab_knobs.sh "X=0" "YNET_CONV_X4=0" ...'
run() {
  for rep in 1 2; do
    env "$@" python bench.py --no-cpu-baseline --steps 12 --warmup 3 --layers 2>/tmp/layers.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],3), end='')"
    awk '/LAYER/ {s+=$(NF-3)} END {print "   conv us/step", s}' /tmp/layers.txt
  done
}
for k in "$@"; do
  echo "== $k"; run $k
done
