"""Rehearsal of bench.py's N>1 path on ONE GPU: two ranks launched exactly as the driver launches
them (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on cuda:0, bookkeeping
collectives over gloo instead of RCCL.  Checks the line rank 0 prints (whole-job value, global batch) and that rank
r's worlds are worlds [r*B, (r+1)*B) of one big batch bit for bit (reset draws and moves are keyed by the global
world number: no collective on the step path, nothing shared between ranks)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd.rollout import RandomRollout

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line_and_full_record(line, tmp_path):
    """Rank 0 prints the COMPACT line (what the driver parses: < 4 KB) and writes the full record to --full-json; the line
    must be the record's own numbers.  Returns the full record (per-rank blocks, diagnostics) for the detailed checks."""
    assert len(line) < 4096, len(line)
    out = json.loads(line)
    full = json.load(open(str(tmp_path / "full.json")))
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data", "higher_is_better", "vs_baseline"):
        assert out[k] == full[k], k
    for k in ("value", "ms_per_step"):
        assert abs(out[k] / full[k] - 1) < 1e-5, k
    assert abs(out["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-5 and out["roofline"]["bound"] == "hbm"
    assert out["config"]["batch_per_gpu"] == full["config"]["batch_per_gpu"] and out["config"]["global_batch"] == full["config"]["global_batch"]
    assert out["full_record"] == str(tmp_path / "full.json")
    if full["n_gpus"] > 1:
        assert len(out["roofline"]["kernel_us_per_launch_by_rank"]) == full["n_gpus"] and out["per_gpu_value"]["ranks"] == full["n_gpus"]
    if "cpu_baseline" in full:
        assert out["cpu_baseline"]["kind"] == full["cpu_baseline"]["kind"] and out["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    return full


def _check_two_rank_job(out, tmp_path, B, n=2):
    assert out["n_gpus"] == n and out["steps"] == 20 and out["warmup"] == 5
    assert out["config"]["batch_per_gpu"] == B and out["config"]["global_batch"] == n * B
    assert out["scaling"] == "weak" and out["unit"] == "env-steps/s"
    # whole-job value = all ranks' worlds over the slowest rank's time
    assert abs(out["value"] - n * B / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["per_gpu"] and len(out["roofline"]["kernel_us_per_launch_by_rank"]) == n
    assert out["roofline"]["frac_timed_region"] > 0 and out["roofline"]["frac_timed_region"] <= out["roofline"]["frac"] * 1.05
    assert out["config"]["ranks_seen"] == n and [r["rank"] for r in out["config"]["ranks"]] == list(range(n))
    assert [r["world_offset"] for r in out["config"]["ranks"]] == [r * B for r in range(n)] and all(r["name"] for r in out["config"]["ranks"])
    assert out["config"]["distinct_gpus"] == 1 and all(r["uuid"] or r["pci"] for r in out["config"]["ranks"])   # the rehearsal: one GPU under all ranks
    pg = out["per_gpu_value"]
    assert pg["ranks"] == n and pg["min"] <= pg["median"] <= pg["max"] and pg["min"] > 0
    sd = out["config"]["scaling_diagnostic"]
    assert sd["rank0_solo_env_steps_per_s"] > 0 and 0 < sd["value_over_n_times_rank0_solo"] < 1.5
    assert out["timed_region_s"] > 0 and out["timed_steps"] == out["config"]["timed_steps"]

    # one process, one batch of n*B worlds: the same episode-0 reset + step 0
    big = mpe.make_env("simple_spread", batch_size=n * B, seed=0)
    rr = RandomRollout(big, episode_len=25, pool=25, regenerate=True)
    o = rr.enqueue(1)
    torch.cuda.synchronize()
    ref = {"pos": big.world.pos.cpu().numpy(), "vel": big.world.vel.cpu().numpy(), "rew": o.rew.cpu().numpy()}
    obs = [x.cpu().numpy() for x in o.obs_n]
    for rank in range(n):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(d["world_offset"]) == rank * B
        sl = slice(rank * B, (rank + 1) * B)
        assert np.array_equal(d["pos"], ref["pos"][:, :, sl])
        assert np.array_equal(d["vel"], ref["vel"][:, :, sl])
        assert np.array_equal(d["rew"], ref["rew"][:, sl])
        off = 0
        for i in range(3):   # agent i's [B, 18] block of the rank's obs buffer
            blk = d["obs"][off * B:(off + 18) * B].reshape(B, 18)
            assert np.array_equal(blk, obs[i][sl])
            off += 18
    assert not np.array_equal(ref["pos"][:, :, :B], ref["pos"][:, :, B:])   # the two shards are different worlds


def _bench_env():
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_two_ranks_one_gpu(tmp_path):
    B = 4096
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--batch", str(B),
           "--backend", "gloo", "--all-ranks-on-gpu0", "--no-cpu-baseline", "--region-ms", "50", "--dump-state", str(tmp_path), "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_bench_env(), timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0) expected, got %d" % len(lines)
    out = _line_and_full_record(lines[0], tmp_path)
    assert "cpu_baseline" not in out and out["config"]["barrier_backend"] == "gloo"
    assert out["config"]["launcher"].startswith("torch.distributed.run")
    _check_two_rank_job(out, tmp_path, B)


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the shape of the driver's N=1 command): bench.py starts the two
    ranks itself.  Both on cuda:0 here, where RCCL cannot come up (one GPU, two ranks): `--backend auto` must notice on
    every rank, carry the barrier over gloo and still produce the line -- with the CPU baseline and the per-GPU roofline."""
    B = 4096
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--batch", str(B),
           "--all-ranks-on-gpu0", "--cpu-seconds", "1", "--region-ms", "50", "--dump-state", str(tmp_path), "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_bench_env(), timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = _line_and_full_record(lines[0], tmp_path)
    assert out["config"]["launcher"].startswith("self-spawned")
    assert out["config"]["barrier_backend"] in ("gloo", "nccl")
    if out["config"]["barrier_backend"] == "gloo":
        assert "RCCL not adopted" in out["config"]["barrier_note"]
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
    assert [r["gpu_env"] for r in out["config"]["ranks"]] == [out["config"]["ranks"][0]["gpu_env"]] * 2      # both were handed the first GPU
    _check_two_rank_job(out, tmp_path, B)


def test_bench_eight_ranks_rehearsal_on_one_gpu(tmp_path):
    """The driver's largest job, rehearsed: `python bench.py --gpus 8 --all-ranks-on-gpu0` -- an 8-way rendezvous, 8 shards of
    4096 worlds each bit-identical to one batch of 32 768, one line with n_gpus = 8, inside a bounded wall time."""
    import time
    B = 4096
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--batch", str(B),
           "--all-ranks-on-gpu0", "--backend", "gloo", "--no-cpu-baseline", "--region-ms", "50", "--dump-state", str(tmp_path), "--full-json", str(tmp_path / "full.json")]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=_bench_env(), timeout=900, cwd=ROOT)
    wall = time.time() - t0
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = _line_and_full_record(lines[0], tmp_path)
    _check_two_rank_job(out, tmp_path, B, n=8)
    assert wall < 240, wall        # (8 Python hosts importing torch on one box included; the timed work is ~1 s)


def test_bench_refuses_two_ranks_mapped_to_one_gpu():
    """Two ranks on the same GPU WITHOUT the rehearsal flag (a mis-mapped job: here --device-map 0,0): refused before anything
    is measured, non-zero exit, no line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--batch", "1024",
           "--device-map", "0,0", "--backend", "gloo", "--no-cpu-baseline", "--region-ms", "20"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_bench_env(), timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "share a GPU" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_more_gpus_than_the_box_has_is_an_error():
    n = torch.cuda.device_count() + 7
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, env=_bench_env(), timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_single_gpu_line_carries_the_contract():
    """`python bench.py --steps K --warmup W` (N = 1, as the driver runs it): ONE JSON line with the contract's keys, the
    HBM roofline object and the CPU baseline object; value == worlds x steps / time."""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-extra",
                        "--cpu-seconds", "1", "--repeats", "3", "--region-ms", "100"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 10 and out["warmup"] == 3 and out["higher_is_better"] is True
    assert out["unit"] == "env-steps/s" and out["dtype"] == "f32" and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"] and out["config"]["batch_per_gpu"] == 65536
    assert abs(out["value"] - 65536 / (out["ms_per_step"] * 1e-3)) <= 1e-4 * out["value"]
    roof = out["roofline"]
    assert len(lines[0]) < 4096
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and roof["regime_label"] == "l3+launch"
    launched = roof["launched"] if out["config"]["mode"] == "step-server" else roof      # (the launched steps' figures, either way)
    assert launched["kernel_us_rocprof"]["mean"] > 0 and launched["kernel_us_rocprof"]["source"].startswith("profiles/")
    if out["config"]["mode"] == "step-server":
        assert roof["launched"]["value"] < out["value"] and roof["kernel_us_per_step"] * 1e-3 <= out["ms_per_step"] * 1.05
        assert roof["kernel_us_rocprof"]["mean_per_step"] > 0 and roof["traffic"] > 0.5 * roof["algorithmic_bytes_per_launch"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5 and 0.2 < roof["frac"] < 1.0
    assert abs(out["timed_steps"] * out["ms_per_step"] * 1e-3 / out["timed_region_s"] - 1) < 1e-3
    assert roof["algorithmic_bytes_per_env_step"] == 411 and (roof["traffic"] is None or roof["traffic"] > 2e7)
    cpu = out["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == "env-steps/s" and cpu["sample"]
