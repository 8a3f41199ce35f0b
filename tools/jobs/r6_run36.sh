#!/bin/bash
# round 6, job 36: the single-frame trace / timeline and the default bench line of the tree with conv1a fused in single-frame calls
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round_r06; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt_b1 -o trace -- python tools/latency_stages.py --calls 200 > $out/kt_b1.log 2>&1
python tools/rocpd_summary.py $out/kt_b1/*.db > $out/kernel_stats_f32_batch1.txt 2>&1
python tools/rocpd_timeline.py $out/kt_b1/*.db "conv_f32_kernel<2," 60 > $out/timeline_f32_batch1.txt 2>&1
grep '^{' $out/kt_b1.log | tail -1 > $out/latency_stages_f32_batch1.json
rm -rf $out/kt_b1
cat $out/timeline_f32_batch1.txt
