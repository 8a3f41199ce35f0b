#!/bin/bash
# ablation builds of the probe (see conv_ws_probe.hip): timing, no-DMA, no-MFMA, consumer priority 0
B=tools/microbench/bin
out=gpurun_out/probe_ws2.txt
mkdir -p gpurun_out
: > $out
for v in probe_t probe_a1 probe_a2 probe_p0; do
  for args in "480 752 8 64 1 20" "720 1280 8 64 1 20" "360 640 8 64 0 20"; do
    echo "== $v $args" >> $out
    PROBE_ONLY=new timeout 120 $B/$v $args 2>&1 | grep -v "PROBE\|sampled" >> $out
  done
done
cat $out
