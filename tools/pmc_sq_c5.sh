#!/bin/bash
# SQ counter passes over the C5 sweep (development aid): the fused read-out and the additive-term convs.
#   gpurun -- 'bash tools/pmc_sq_c5.sh r02'  ->  gpurun_out/pmc_sq_c5_<tag>/summary.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sq_c5_${1:-r02}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp -d "$OUT/p$i" -o c5 -- python3 "$R/bench.py" --config C5 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
res = {}
for db in glob.glob(os.path.join(out, "p*", "*.db")):
    con = sqlite3.connect(db)
    for name, counter, value in con.execute("select name, counter_name, counter_value from pmc_events"):
        if "pred_softargmax_kernel" in name or "conv_dma_add_kernel" in name or "conv_dma_kernel<2, 4, 4" in name or "upsample2x_fwd_rows_kernel<4>" in name:
            d = res.setdefault(name[:64], {}).setdefault(counter, [0.0, 0])
            d[0] += float(value); d[1] += 1
    con.close()
    os.remove(db)
summ = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in res.items()}
for k, d in summ.items():
    if d.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * d["SQ_BUSY_CU_CYCLES"])
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: round(v.get("mfma_busy_frac", -1), 3) for k, v in summ.items()}, indent=1))
PY
