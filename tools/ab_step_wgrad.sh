# in-step wgrad time per library build: kernel trace of 8 bench steps each
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  rm -rf /tmp/abt; 
  YNET_HIP_LIB=$([ "$lib" = current ] && echo "" || echo $R/$lib) YNET_SERIAL_DECODERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abt -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/abt.log 2>&1
  python3 - "$lib" <<'PY'
import glob, sqlite3, sys
db = glob.glob("/tmp/abt/**/*results.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), avg(end-start), max(end-start), sum(end-start) from kernels where name like '%wgrad%' or name like '%conv_dma_kernel<2, 4, 4, false%' group by name").fetchall()
for r in rows: print(sys.argv[1], r[0][:60], r[1], round(r[2]/1e3,1), round(r[3]/1e3,1))
PY
done
