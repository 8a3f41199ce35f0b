#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r18; mkdir -p $out
B=tools/microbench/bin
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 60 --warmup 5 --height 2160 --width 3840 --frames-per-gpu 1"
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for rep in 1 2 3; do for which in prev this; do
  [ $which = prev ] && cp $B/libspfe_prev.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  for p in f32 bf16; do r=$(python bench.py $C --precision $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"); echo "$which $p $r" >> $out/ab.txt; done
done; done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
cat $out/ab.txt
