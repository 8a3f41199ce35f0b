#!/bin/bash
# Build container: the stand-alone probes tools/profile_round.sh runs on the GPU box (tools/microbench/bin/ travels with gpurun).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/microbench/bin
F="--offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -Iinclude -Isp_orb_slam_amd/csrc"
/opt/rocm/bin/hipcc $F tools/microbench/clock_probe.hip -o tools/microbench/bin/clock_probe
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form tools/microbench/conv_rw_probe.hip -o tools/microbench/bin/conv_rw_plain
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -DRW_PROBE tools/microbench/conv_rw_probe.hip -o tools/microbench/bin/conv_rw_probe_a0
/opt/rocm/bin/hipcc $F tools/microbench/mfma_chain_probe.hip -o tools/microbench/bin/mfma_chain_probe
/opt/rocm/bin/hipcc $F tools/microbench/barrier_probe.hip -o tools/microbench/bin/barrier_probe
# libspfe.so with dust.hip's phase counters (-DSPFE_DUST_PROBE): on the GPU box copy it over sp_orb_slam_amd/libspfe.so, run
# tools/microbench/dust_time.py, read the "dust probe:" lines
( cd sp_orb_slam_amd/csrc && make -j8 >/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fPIC -fvisibility=hidden -I../../include -DSPFE_DUST_PROBE \
      -c dust.hip -o /tmp/dust_probe.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/microbench/bin/libspfe_dustprobe.so \
      $(ls *.o | grep -v '^dust.o$') /tmp/dust_probe.o )
