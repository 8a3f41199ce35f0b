cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ht && mkdir -p gpurun_out/ht
SPFE_PIPE_COPY_KERNEL=${1:-2} rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/ht/kt -o trace -- python tools/microbench/hostprof.py bf16 > gpurun_out/ht/log.txt 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/ht/kt/*.db")[0])
cur = db.cursor()
ks = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
mc = cur.execute("select name, start, end, size from memory_copies order by start").fetchall()
tl = [k for k in ks if 'tail_kernel' in k[0]]
t0 = tl[-6][1]
ev = []
for k in ks:
    if t0 - 100000 <= k[1] <= t0 + 2300000 and any(x in k[0] for x in ('tail_kernel','true, 2','copyBuffer','fillBuffer','cov_replay','select_kernel','zero_ints','copy16','head1x1_bf16_kernel<256')):
        ev.append((k[1], "K q%s %s" % (k[3], k[0][:44]), k[2]))
for m in mc:
    if t0 - 100000 <= m[1] <= t0 + 2300000:
        ev.append((m[1], "COPY %s %d B" % (m[0][12:], m[3]), m[2]))
for e in sorted(ev):
    print("%8.3f -> %8.3f  %s" % ((e[0]-t0)/1e6, (e[2]-t0)/1e6, e[1]))
PY
rm -rf gpurun_out/ht/kt
