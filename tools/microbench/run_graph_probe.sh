cd "$GRAFT_REPO_ROOT"
run() { echo -n "[$*] "; env "$@" timeout 200 python tools/microbench/graph_capture_probe.py $CFG 2>&1 | grep -E "^f32|^bf16" | cut -c1-150; echo; }
CFG="f32 1"; run SPFE_INLINE_CHAIN=0; run SPFE_INLINE_CHAIN=0 SPFE_SPARSE_DB=0; run SPFE_INLINE_CHAIN=0 SPFE_SPARSE_DB=0 SPFE_DEFER_DB=0
CFG="bf16 1"; run SPFE_DEFER_DB=0; run SPFE_SPARSE_DB=1 SPFE_SPARSE_DA=1; run SPFE_DESC_IN_REPLAY=0
