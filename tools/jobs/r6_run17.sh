#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r17; mkdir -p $out
B=tools/microbench/bin
cp sp_orb_slam_amd/libspfe.so /tmp/this.so
for which in prev this; do
  [ $which = prev ] && cp $B/libspfe_prev.so sp_orb_slam_amd/libspfe.so || cp /tmp/this.so sp_orb_slam_amd/libspfe.so
  echo "== $which" >> $out/chains.txt
  python tools/cov_chain_stats.py 720 1280 1000 dense 200,201,202,203 bf16 2>&1 | grep seed >> $out/chains.txt
done
cp /tmp/this.so sp_orb_slam_amd/libspfe.so
cat $out/chains.txt
