#!/bin/bash
# C5 (K = 20 evaluation sweep) measured in isolation (run on the GPU box through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/profile_c5.sh r04'
#  1. rocprofv3 --kernel-trace --stats of `bench.py --config C5` with YNET_SERIAL_DECODERS=1: the K-sample groups run back to back on ONE
#     stream = isolated per-kernel durations (the default sweep alternates two streams, whose kernels share the chip)
#  2. PMC passes FETCH_SIZE / WRITE_SIZE (separate passes, never combined with a trace domain) of the same serial run ->
#     pmc_traffic_c5.json (per-kernel HBM bytes per launch)
#  3. TCC_EA0_RDREQ / TCC_EA0_WRREQ per channel (one pass): channel balance of the plane-strided HBM-bound kernels
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_c5_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config C5 --no-cpu-baseline --no-roofline"
export YNET_SERIAL_DECODERS=1
rm -rf /tmp/tr_c5
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c5 -o t -- $B --steps 3 --warmup 2 > "$OUT/trace_c5_serial.log" 2>&1
echo "trace C5 serial rc=$?"; python3 "$R/tools/trace_summary.py" /tmp/tr_c5 "$OUT/${TAG}_bench_C5_serial" --tail-frac 0.5 > "$OUT/timeline_c5_serial.txt"
mkdir -p "$R/gpurun_out/prof_c5tmp_$TAG"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 400 rocprofv3 --pmc $c -d "$R/gpurun_out/prof_c5tmp_$TAG/$d" -o c5 -- $B --steps 1 --warmup 1 > "$OUT/$d.log" 2>&1
  echo "$c rc=$?"
done
python3 "$R/tools/pmc_aggregate.py" "c5tmp_$TAG" --to "$OUT" && mv "$OUT/pmc_traffic.json" "$OUT/pmc_traffic_c5.json"
rm -rf "$R/gpurun_out/prof_c5tmp_$TAG"
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ -d "$OUT/tcc" -o c5 -- $B --steps 1 --warmup 1 > "$OUT/tcc.log" 2>&1
echo "tcc rc=$?"
python3 - "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
dbs = glob.glob(os.path.join(out, "tcc", "**", "*.db"), recursive=True)
res = {"tables": {}, "kernels": {}}
for db in dbs:
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    for n in names:
        try:
            cols = [c[1] for c in con.execute(f'pragma table_info("{n}")')]
            res["tables"][n] = cols
        except Exception as e:
            res["tables"][n] = str(e)
    # per-dispatch rows of the kernels of interest with every column the view offers (dimension columns, if any, show per-channel values)
    try:
        cols = [c[1] for c in con.execute('pragma table_info("pmc_events")')]
        for row in con.execute("select * from pmc_events"):
            d = dict(zip(cols, row))
            nm = str(d.get("name", ""))
            for key in ("pred_softargmax_kernel", "conv_dma_add_kernel", "upsample2x_fwd_rows_kernel<4>", "softargmax_kernel", "conv_dma_kernel<2, 4, 4"):
                if key in nm:
                    lst = res["kernels"].setdefault(key, [])
                    if len(lst) < 40:
                        lst.append({k: (v if not isinstance(v, bytes) else v.hex()) for k, v in d.items() if k != "name"})
    except Exception as e:
        res["error"] = str(e)
    con.close()
    os.remove(db)
json.dump(res, open(os.path.join(out, "tcc_channels_raw.json"), "w"), indent=1, default=str)
PY
rm -rf "$OUT/tcc"
ls -la "$OUT"
