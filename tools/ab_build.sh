#!/bin/bash
# A/B build: recompile ONE kernel file with extra -D flags and link it with the other objects of the normal build.
#   usage: tools/ab_build.sh <tag> <stem: split|wide|narrow|rng|abi> [-DNAME=VALUE ...]   -> lib/libmpe_hip_ab_<tag>.so
# Select it at run time with MPE_HIP_LIB=<path> (multiagent_particle_envs_amd/_abi.py).
set -e
cd "$(dirname "$0")/.."
P=multiagent_particle_envs_amd
TAG=$1; STEM=$2; shift 2
python -m $P._build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=14 "$@" \
    -c $P/csrc/mpe_$STEM.hip -o $P/build/mpe_${STEM}_ab_$TAG.o
OBJS=""
for s in abi narrow split wide rng rows; do
  if [ $s == $STEM ]; then OBJS="$OBJS $P/build/mpe_${STEM}_ab_$TAG.o"; else OBJS="$OBJS $P/build/mpe_$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/libmpe_hip_ab_$TAG.so $OBJS
echo $P/lib/libmpe_hip_ab_$TAG.so
