#!/bin/bash
# SQ counters of the Winograd prototype (development aid):  gpurun --timeout 900 -- 'bash tools/pmc_wino.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_wino
mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -o /tmp/conv_wino "$R/tools/conv_wino_proto.hip" 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/p$i" -o w -- /tmp/conv_wino > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
res = {}
for db in glob.glob(os.path.join(out, "p*", "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select name, counter_name, counter_value from pmc_events"))
    except Exception as e:
        print("db", db, e); rows = []
    for name, counter, value in rows:
        if "wino" in name:
            d = res.setdefault(name[:40], {}).setdefault(counter, [0.0, 0])
            d[0] += float(value); d[1] += 1
    con.close()
    os.remove(db)
summary = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in res.items()}
for k, e in summary.items():
    busy, mfma = e.get("SQ_BUSY_CU_CYCLES"), e.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if busy and mfma is not None: e["mfma_busy_frac"] = mfma / (4 * busy)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
