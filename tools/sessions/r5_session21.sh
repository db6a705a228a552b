#!/bin/bash
# round 5, GPU visit 21: which of convoy's thresholds do 2.5 % of the worlds sit on after a few free-running steps?
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
tools/sessions/_gpu_ok.sh || { echo 'BAD BOX: leaving'; exit 0; }
python - <<'PY'
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import symtrace
env = mpe.make_env("tests/refstyle/convoy.py", batch_size=65536, seed=6)
tr = env.scenario.t
rs = np.random.RandomState(1)
env.reset()
B = 65536
for t in range(4):
    moves = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(env.n, B))]).cuda()
    env.step(moves)
    P, V = env.world.get_state(all_entities=True)
    st = dict(P=P.astype(np.float64), V=V.astype(np.float64), Cw=np.zeros((B, tr.A, tr.dim_c)), K=env.world.choice_i32.cpu().numpy().T)
    roots = [x for row in tr.obs for x in row] + list(tr.rew)
    print("t", t, "masked", (symtrace.decision_margin(roots, B, **st) <= 2e-6).mean(), "nan worlds", np.isnan(P).any(axis=(1, 2)).mean())
    for n in symtrace.topo(roots):
        if n.op in ("lt", "le"):
            a = symtrace.evaluate([n.args[0], n.args[1]], B, **st)
            d = np.abs(a[0] - a[1])
            f = (d <= 2e-6).mean()
            if f > 1e-3:
                r = repr(n)
                print("   ", f, r[:60], "...", r[-40:], " example lhs/rhs", a[0][d <= 2e-6][:3], a[1][d <= 2e-6][:3])
PY
exit 0
