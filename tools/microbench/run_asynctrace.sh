cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/at && mkdir -p gpurun_out/at
SPFE_DEFER_SIDE_LAYER=${1:--1} rocprofv3 --kernel-trace -d gpurun_out/at/kt -o trace -- python bench.py --precision bf16 --height 720 --width 1280 --no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --no-aten --steps 8 > gpurun_out/at/log.txt 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/at/kt/*.db")[0])
cur = db.cursor()
ks = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
tl = [k for k in ks if 'tail_kernel' in k[0]]
t0 = tl[-4][2]
for k in ks:
    if t0 - 20000 <= k[1] <= t0 + 1300000:
        nm = k[0].replace('spfe::','').replace('void ','')[:46]
        print("%8.3f -> %8.3f (%6.1f us) q%s %s" % ((k[1]-t0)/1e6, (k[2]-t0)/1e6, (k[2]-k[1])/1e3, k[3], nm))
PY
rm -rf gpurun_out/at/kt
