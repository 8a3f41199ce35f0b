#!/bin/bash
# round 6, job 16: 8-byte stores in the no-pool epilogue (conv2a / conv3a shapes), and the single-frame bf16 1280x720 call (its p50 read 0.35 ms in
# job 15 against 0.32 before): probe at 1 / 2 frames, latency A/B of the two libraries
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r16; mkdir -p $out
B=tools/microbench/bin
for args in "120 160 1 64 0 3" "16 32 1 64 0 3" "240 376 8 64 0 3" "360 640 2 64 0 3" "120 188 2 128 0 3" "24 40 3 64 0 3"; do
  echo "== w_h $args" >> $out/probe.txt; timeout 120 $B/w_h $args 2>&1 | grep -v "sampled" | tail -2 >> $out/probe.txt
done
for rep in 1 2 3; do for v in w_old w_h0 w_h; do
  for a in "360 640 8 64 0 200" "180 320 8 64 0 200" "360 640 8 128 0 200"; do echo -n "$v $a: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
done; done
for rep in 1 2; do for v in w_old w_h; do
  for a in "720 1280 1 64 1 300 fuse" "720 1280 2 64 1 300 fuse" "480 752 1 64 1 300 fuse"; do echo -n "$v $a: " >> $out/probe.txt; PROBE_FAST=1 PROBE_ONLY=new timeout 120 $B/$v $a 2>&1 | grep "^conv" | sed 's/.*ws /ws /' >> $out/probe.txt; done
done; done
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for rep in 1 2; do for which in prev this; do
  [ $which = prev ] && cp $B/libspfe_prev.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  for cfg in "--precision bf16 --height 720 --width 1280" "--precision bf16" ""; do
    echo -n "$which [$cfg]: " >> $out/latency.txt; python tools/latency_stages.py --calls 400 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['p50_ms'], d['p99_ms'], {k: d['stage_ms'][k] for k in ('conv1a','conv1b','conv2a','tail','post_side','total') if k in d['stage_ms']})" >> $out/latency.txt
  done
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
grep -v "^old vs\|PROBE OK" $out/probe.txt; grep -c "PROBE OK" $out/probe.txt; cat $out/latency.txt
