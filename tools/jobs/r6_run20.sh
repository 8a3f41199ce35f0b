#!/bin/bash
# round 6, job 20: deferred moments in synchronous calls only + no fence per replay member (two-list patch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r20; mkdir -p $out
B=tools/microbench/bin
( time timeout 2400 python -m pytest tests/test_gpu_selection.py tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_stress.py tests/test_gpu_sparse_db.py tests/test_gpu_bf16.py -x -q ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
cp $B/libspfe_chainprobe.so sp_orb_slam_amd/libspfe.so
python tools/cov_chain_stats.py 480 752 1000 dense 200,201 > $out/chainprobe.txt 2>&1
for rep in 1 2; do for which in head this; do
  [ $which = head ] && cp $B/libspfe_head.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  for cfg in "" "--precision bf16 --height 720 --width 1280" "--precision bf16"; do
    echo -n "$which [$cfg]: " >> $out/latency.txt; python tools/latency_stages.py --calls 400 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'], d['stage_ms'].get('post_side'))" >> $out/latency.txt
  done
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
bash tools/microbench/ab_lib.sh $B/libspfe_head.so 3 > $out/ab_lib.txt 2>&1
tail -3 $out/pytest.log; cat $out/chainprobe.txt $out/latency.txt $out/ab_lib.txt
