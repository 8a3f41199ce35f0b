#!/bin/bash
# round 6, job 44: the twin by the refined workload rule: whole GPU suite; configs[3] default against SPFE_TWO_CHAINS=1 / 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r44; mkdir -p $out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
C="--no-cpu-baseline --no-bf16-leg --no-host-path --no-match --no-latency --no-stage-table --warmup 20 --steps 200 --precision bf16 --height 720 --width 1280"
run() { python bench.py $C "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])"; }
for rep in 1 2; do for tc in default 1 0; do
  [ $tc = default ] && unset SPFE_TWO_CHAINS || export SPFE_TWO_CHAINS=$tc
  echo -n "720p b8 two_chains $tc: " >> $out/ab.txt; run >> $out/ab.txt
  echo -n "720p b2 two_chains $tc: " >> $out/ab.txt; run --frames-per-gpu 2 >> $out/ab.txt
done; done
unset SPFE_TWO_CHAINS
tail -4 $out/pytest.log; cat $out/ab.txt
