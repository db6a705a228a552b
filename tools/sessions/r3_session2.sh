#!/bin/bash
# round 3, GPU visit 2: full parity suite on the kernarg-preload / pinned-constants / own-reward k_split, A/B matrix, phase clocks
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3s2; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
tools/ab_matrix.sh r3s2 "tag:3:16384 spread:3:4096 spread:3:65536 simple_adversary:3:65536" old nopre base rw own3
for v in clk clkrw; do
  export MPE_HIP_LIB=$R/multiagent_particle_envs_amd/lib/libmpe_hip_ab_$v.so
  echo "== $v" | tee -a $O/phase.txt
  timeout 120 python tools/phase_clock.py simple_tag 16384 roll 2>>$O/err.log | tee -a $O/phase.txt
  timeout 120 python tools/phase_clock.py simple_tag 16384 step 2>>$O/err.log | tee -a $O/phase.txt
done
timeout 120 python tools/phase_clock.py simple_spread 4096 roll 2>>$O/err.log | tee -a $O/phase.txt
timeout 120 python tools/phase_clock.py simple_spread 4096 step 2>>$O/err.log | tee -a $O/phase.txt
timeout 120 python tools/phase_clock.py simple_spread 65536 step 2>>$O/err.log | tee -a $O/phase.txt
tail -5 $O/err.log
