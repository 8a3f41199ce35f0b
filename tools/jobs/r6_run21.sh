#!/bin/bash
# round 6, job 21: the committed tree (deferred moments in synchronous calls, scalar head / tail, fence kept): whole GPU suite,
# single-frame latency against the library before the change, the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r21; mkdir -p $out
B=tools/microbench/bin
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for rep in 1 2; do for which in head this; do
  [ $which = head ] && cp $B/libspfe_head.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  for cfg in "" "--precision bf16 --height 720 --width 1280" "--precision bf16"; do
    echo -n "$which [$cfg]: " >> $out/latency.txt; python tools/latency_stages.py --calls 400 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'], d['stage_ms'].get('post_side'))" >> $out/latency.txt
  done
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
bash tools/microbench/ab_lib.sh $B/libspfe_head.so 2 > $out/ab_lib.txt 2>&1
python bench.py > $out/bench_default.json 2> $out/bench_default.err
cp gpurun_out/bench_full.json $out/bench_default_full.json 2>/dev/null
tail -4 $out/pytest.log; cat $out/latency.txt $out/ab_lib.txt; wc -c $out/bench_default.json; head -c 600 $out/bench_default.json
