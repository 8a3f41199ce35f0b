#!/bin/bash
# runs the conv_ws probe over the Cin = 64 layer shapes of the bf16 mode (build it first, see conv_ws_probe.hip)
P=tools/microbench/bin/conv_ws_probe
out=gpurun_out/probe_ws.txt
mkdir -p gpurun_out
: > $out
for args in "64 96 2 64 1 5" "480 752 8 64 1 20" "720 1280 8 64 1 20" "240 376 8 64 0 20" "240 376 8 64 1 20" "120 188 8 128 0 20" "360 640 8 64 0 20" "360 640 8 64 1 20" "180 320 8 128 0 20"; do
  timeout 120 $P $args >> $out 2>&1 || echo "rc=$? for $args" >> $out
done
cat $out
