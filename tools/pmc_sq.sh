#!/bin/bash
# SQ counter passes over the C2 bench (development aid): where do the conv / wgrad kernels wait?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sq_${1:-r02}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export YNET_SERIAL_DECODERS=1 YNET_STEP_GRAPH=0      # eager launches on one stream: one counter sample per isolated kernel
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/p$i" -o c2 -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats ${PMC_BENCH_ARGS:-} > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
res = {}
for db in glob.glob(os.path.join(out, "p*", "*.db")):
    con = sqlite3.connect(db)
    for name, counter, value in con.execute("select name, counter_name, counter_value from pmc_events"):
        # (round 4: + the small-map kernels -- one / two rows per wave of one 16-channel tile, folded 8^2 / 16^2 maps, the register-staged
        # one-row kernel -- which had no counters: VERDICT r3 item 1)
        if any(k in name for k in ("kernel<2, 4, 4", "kernel<4, 2, 4", "kernel<3, 2, 4", "kernel<2, 2, 4", "wgrad_dma", "wgrad_roll", "lora_wgrad_kernel",
                                   "conv_wino", "kernel<1, 1, ", "kernel<1, 2, ", "kernel<1, 4, ", "kernel<2, 1, ", "conv_mfma_kernel<3, 1, 1, 8", "conv_chain", "pred_bce", "conv_split_reduce")):
            d = res.setdefault(name[:60], {}).setdefault(counter, [0.0, 0])
            d[0] += float(value); d[1] += 1
    con.close()
    os.remove(db)
summary = {}
for k, d in res.items():
    e = {c: v[0] / v[1] for c, v in d.items()}
    busy, mfma, act, conf = e.get("SQ_BUSY_CU_CYCLES"), e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("SQ_LDS_IDX_ACTIVE"), e.get("SQ_LDS_BANK_CONFLICT")
    e["derived"] = {"mfma_busy_frac": mfma / (4 * busy) if busy and mfma is not None else None,          # 4 SIMDs per CU
                    "lds_bank_conflict_per_lds_active": conf / act if act else None,
                    "lds_active_per_busy_cu_cycle": act / busy if busy and act is not None else None}
    summary[k] = e
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
