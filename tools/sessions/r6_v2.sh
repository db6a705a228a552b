cd $GRAFT_REPO_ROOT; O=gpurun_out/r6v2; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$PWD
tools/sessions/_gpu_ok.sh || exit 0
timeout 600 python tools/c4_parity_ab.py > $O/c4_parity_product.txt 2> $O/c4_parity_product.err; echo rc=$?; tail -8 $O/c4_parity_product.txt
MPE_HIP_LIB=$PWD/multiagent_particle_envs_amd/lib/libmpe_hip_ab_exact.so timeout 600 python tools/c4_parity_ab.py > $O/c4_parity_exact.txt 2> $O/c4_parity_exact.err; echo rc=$?; tail -8 $O/c4_parity_exact.txt
timeout 600 python tools/c4_parity_ab.py > $O/c4_parity_product2.txt 2>/dev/null; tail -1 $O/c4_parity_product2.txt
MPE_HIP_LIB=$PWD/multiagent_particle_envs_amd/lib/libmpe_hip_ab_exact.so timeout 600 python tools/c4_parity_ab.py > $O/c4_parity_exact2.txt 2>/dev/null; tail -1 $O/c4_parity_exact2.txt
timeout 900 python tools/device_span.py C2 C3 C5 --out $O > $O/span.log 2> $O/span.err; echo "span rc=$?"; grep -E "^#|period|span |gap|slope|cross|device period" $O/span.log | head -40
