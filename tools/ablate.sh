#!/bin/bash
# Build ablation variants of libmpe_hip.so: add temporary `#if defined(MPE_ABLATE) && MPE_ABLATE == n` blocks to a kernel,
# run this (one .so per n under tools/ubench/ablate/), time them with MPE_HIP_LIB=... bench.py, remove the blocks.
cd /root/repo
mkdir -p tools/ubench/ablate
for n in "$@"; do
  ( cd multiagent_particle_envs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DMPE_ABLATE=$n -shared -o ../../tools/ubench/ablate/libmpe_ab$n.so mpe_abi.hip mpe_narrow.hip mpe_split.hip mpe_wide.hip mpe_rng.hip 2>&1 | grep -v warning | head -5 ) &
done
wait
ls -la tools/ubench/ablate
