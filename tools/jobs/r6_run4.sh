#!/bin/bash
# round 6, job 4: replay with chains laid out contiguously + windows COV_AHEAD members ahead (bitwise tests, single-frame latency A/B, a trace of
# one single-frame call), and the conv1a store forms once more (old 8 x b64 against 4 x b128 with the scalar epilogue)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4; mkdir -p $out
B=tools/microbench/bin
( time timeout 2400 python -m pytest tests/test_gpu_selection.py tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_stress.py tests/test_gpu_sparse_db.py -x -q ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for rep in 1 2; do for which in prev this; do
  [ $which = prev ] && cp $B/libspfe_prev.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  echo -n "$which f32 752x480: " >> $out/latency.txt; python tools/latency_stages.py --calls 400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'], d['stage_ms'].get('post_side'), d['stage_ms'].get('total'))" >> $out/latency.txt
  echo -n "$which bf16 1280x720: " >> $out/latency.txt; python tools/latency_stages.py --calls 400 --precision bf16 --height 720 --width 1280 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'])" >> $out/latency.txt
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
# one single-frame call, kernel by kernel
rocprofv3 --kernel-trace --stats -d $out/kt_b1 -o trace -- python tools/latency_stages.py --calls 200 > $out/kt_b1.log 2>&1
python tools/rocpd_summary.py $out/kt_b1/*.db > $out/kernel_stats_f32_batch1.txt 2>&1
python tools/rocpd_timeline.py $out/kt_b1/*.db conv1a 60 > $out/timeline_f32_batch1.txt 2>&1
rm -rf $out/kt_b1
# replay phase counters (chains of >= 8 members), frames 201 (21 members) and 200
cp $B/libspfe_chainprobe.so sp_orb_slam_amd/libspfe.so
python tools/cov_chain_stats.py 480 752 1000 dense 200,201 > $out/chainprobe.txt 2>&1
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
# pipelined A/B of the two libraries (three workloads)
bash tools/microbench/ab_lib.sh $B/libspfe_prev.so 2 > $out/ab_lib.txt 2>&1
for rep in 1 2 3; do for v in w_old w_b128s; do
  echo -n "$v 720p fuse: " >> $out/probe.txt
  PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v 720 1280 8 64 1 200 fuse 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt
done; done
tail -3 $out/pytest.log; cat $out/latency.txt $out/probe.txt $out/ab_lib.txt; cat $out/timeline_f32_batch1.txt | tail -25
