#!/bin/bash
# same-box A/B of two builds of libspfe.so over the three pipelined workloads: usage tools/microbench/ab_lib.sh <other.so> [reps]
# (the other build — e.g. the previous commit's, copied to tools/microbench/bin/ before the change — takes the library's place for its runs)
other=$1; reps=${2:-2}
cd "$(dirname "$0")/../.."
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --steps 200 --warmup 20"
cp sp_orb_slam_amd/libspfe.so /tmp/libspfe_this.so
trap 'cp /tmp/libspfe_this.so sp_orb_slam_amd/libspfe.so' EXIT
for rep in $(seq $reps); do
  for which in other this; do
    [ $which = other ] && cp "$other" sp_orb_slam_amd/libspfe.so || cp /tmp/libspfe_this.so sp_orb_slam_amd/libspfe.so
    for cfg in "f32:" "bf16_720p:--precision bf16 --height 720 --width 1280" "bf16_752:--precision bf16"; do
      n=${cfg%%:*}; a=${cfg#*:}
      [ -n "$ONLY" ] && [ "$ONLY" != "$n" ] && continue
      r=$(python bench.py $C $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])")
      echo "$which $n $r"
    done
  done
done
