#!/bin/bash
B=tools/microbench/bin
out=gpurun_out/probe_ws3.txt
: > $out
$B/clock_probe 20000 >> $out 2>&1
(for i in $(seq 1 30); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done) > gpurun_out/smi.txt &
for v in probe_t probe_a1; do
  for args in "720 1280 8 64 1 400"; do
    echo "== $v $args" >> $out
    PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep -v "PROBE\|sampled" >> $out
    PROBE_ONLY=old timeout 120 $B/$v $args 2>&1 | grep -v "PROBE\|sampled" >> $out
  done
done
wait
cat $out; cat gpurun_out/smi.txt | sort | uniq -c | sort -rn | head -20
