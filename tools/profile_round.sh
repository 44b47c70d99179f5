#!/bin/bash
# Round profile (run on the GPU box through gpurun; results land in gpurun_out/prof_<tag>/, tools/pmc_aggregate.py and a
# copy step turn them into the files committed under profiles/):
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r02'
#  1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (captured step, parallel branches; without the CPU baseline, the C5
#     leg and the repeat regions of the default run, so that the table holds the C2 step's kernels only)
#  2. the same with YNET_STEP_GRAPH=0 YNET_SERIAL_DECODERS=1: eager launches on one stream = isolated per-kernel durations,
#     the condition under which bench.py times the dominant kernel for its `roofline` object
#  2b. the eager serial trace of C1 (every conv trainable: the wgrad kernels)
#  3. PMC passes FETCH_SIZE / WRITE_SIZE (separate passes, never combined with a trace domain) of the eager serial run
#  4. bench lines: C2 (default, with cpu_baseline + parity_check), C2 at the reference scripts' batch 10, C1, C3, C4, C5
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
rm -rf /tmp/tr_*
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_graph -o t -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/trace_graph.log" 2>&1
echo "trace graph rc=$?"; python3 "$R/tools/trace_summary.py" /tmp/tr_graph "$OUT/${TAG}_bench_C2" --tail-frac 0.6 > "$OUT/timeline_graph.txt"
export YNET_STEP_GRAPH=0 YNET_SERIAL_DECODERS=1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_serial -o t -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/trace_serial.log" 2>&1
echo "trace serial rc=$?"; python3 "$R/tools/trace_summary.py" /tmp/tr_serial "$OUT/${TAG}_bench_C2_serial" --tail-frac 0.6 > "$OUT/timeline_serial.txt"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c1 -o t -- $B --config C1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-sustained --no-repeats > "$OUT/trace_c1_serial.log" 2>&1
echo "trace C1 serial rc=$?"; python3 "$R/tools/trace_summary.py" /tmp/tr_c1 "$OUT/${TAG}_bench_C1_serial" --tail-frac 0.6 > "$OUT/timeline_c1_serial.txt"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o c2 -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o c2 -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/write.log" 2>&1
echo "write rc=$?"
python3 "$R/tools/pmc_aggregate.py" "$TAG" --to "$OUT"      # per-kernel HBM bytes -> $OUT/pmc_traffic.json; the counter databases stay on the box
# 3b. the per-LAUNCH-SHAPE table (round 5): bench.py's launch order of one eager serial step joined with the serial trace's kernel-only
#     durations and the two counter passes -> ${TAG}_conv_shapes_C2.json (every per-shape roofline fraction reproducible from profiles/)
$B --steps 3 --warmup 1 --no-cpu-baseline --no-c5 --no-legs --no-sustained --no-repeats --conv-layers "$OUT/${TAG}_conv_layers_C2.json" > /dev/null 2> "$OUT/conv_layers.err"
python3 "$R/tools/conv_shapes.py" "$OUT/${TAG}_conv_layers_C2.json" /tmp/tr_serial "$OUT/${TAG}_conv_shapes_C2.json" --fetch "$OUT/fetch" --write "$OUT/write" > "$OUT/conv_shapes.txt" 2>&1
echo "conv shapes rc=$?"; head -14 "$OUT/conv_shapes.txt"
# (the bench line below reads its `roofline.traffic` from profiles/<tag>_conv_shapes_C2.json: this build's table, on the box's scratch copy of the repository too)
[ -s "$OUT/${TAG}_conv_shapes_C2.json" ] && cp "$OUT/${TAG}_conv_shapes_C2.json" "$R/profiles/${TAG}_conv_shapes_C2.json"
rm -rf "$OUT/fetch" "$OUT/write"
unset YNET_STEP_GRAPH YNET_SERIAL_DECODERS
cd "$R"
$B --steps 30 --warmup 5 > "$OUT/${TAG}_bench_C2_line.json" 2> "$OUT/bench_C2.err"; echo "C2 rc=$?"
$B --steps 30 --warmup 20 --batch 10 --no-cpu-baseline > "$OUT/${TAG}_bench_C2_batch10_line.json" 2> "$OUT/bench_C2_b10.err"; echo "C2 b10 rc=$?"
YNET_STEP_GRAPH=0 $B --steps 30 --warmup 5 --batch 10 --no-cpu-baseline --no-roofline > "$OUT/${TAG}_bench_C2_batch10_eager_line.json" 2>/dev/null
for c in C1 C3 C4 C5; do
  $B --steps 20 --warmup 5 --config $c --no-cpu-baseline > "$OUT/${TAG}_bench_${c}_line.json" 2> "$OUT/bench_$c.err"; echo "$c rc=$?"
done
ls -la "$OUT"
