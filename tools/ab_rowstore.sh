#!/bin/bash
# Same-box A/B of the row-store policy (mpe_device.h): `plain` = a library built with -DMPE_ROW_STORE=0 (ordinary stores everywhere,
# what the library did before session 40), `base` = the library's own choice (nontemporal / agent scope / plain by shape and size).
# Build `plain` first:  for s in split wide narrow; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
#   -DMPE_ROW_STORE=0 -c multiagent_particle_envs_amd/csrc/mpe_$s.hip -o multiagent_particle_envs_amd/build/mpe_${s}_ab_plain.o; done; then link the three with
#   build/mpe_abi.o and build/mpe_rng.o into multiagent_particle_envs_amd/lib/libmpe_hip_ab_plain.so
cd $GRAFT_REPO_ROOT
REPS=3 tools/ab_run.sh rs_h "--steps 200 --warmup 20" plain base | sed "s/^/spread3 65536 /"
REPS=2 tools/ab_run.sh rs_1m "--batch 1048576 --steps 25 --warmup 5" plain base | sed "s/^/spread3 1M /"
REPS=2 tools/ab_run.sh rs_tag "--scenario simple_tag --batch 16384 --steps 200 --warmup 20" plain base | sed "s/^/tag 16384 /"
REPS=2 tools/ab_run.sh rs_c2 "--batch 4096 --steps 200 --warmup 20" plain base | sed "s/^/spread3 4096 /"
REPS=3 tools/ab_run.sh rs_64 "--agents 64 --batch 4096 --steps 200 --warmup 20" plain base | sed "s/^/N64 step /"
REPS=2 tools/ab_run.sh rs_64f "--agents 64 --batch 4096 --steps 200 --warmup 20 --mode fused" plain base | sed "s/^/N64 fused /"
REPS=1 tools/ab_run.sh rs_100 "--agents 100 --batch 2048 --steps 100 --warmup 10" plain base | sed "s/^/N100 step /"
REPS=1 tools/ab_run.sh rs_40 "--agents 40 --batch 4096 --steps 100 --warmup 10" plain base | sed "s/^/N40 step /"
REPS=1 tools/ab_run.sh rs_8 "--agents 8 --batch 65536 --steps 100 --warmup 10" plain base | sed "s/^/N8 step /"
REPS=1 tools/ab_run.sh rs_16 "--agents 16 --batch 16384 --steps 100 --warmup 10" plain base | sed "s/^/N16 step /"
REPS=1 tools/ab_run.sh rs_16f "--agents 16 --batch 16384 --steps 100 --warmup 10 --mode fused" plain base | sed "s/^/N16 fused /"
