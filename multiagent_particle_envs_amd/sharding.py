"""Multi-GPU: worlds shard by batch index, one process per GPU, no collective on the step path.

World b of a global batch lives on rank b // (B_global / world_size) (contiguous slices).  The only
communication is bookkeeping around the timed region: a barrier on each side, a MAX-reduce of the
elapsed time and a gather of per-rank records.  RNG streams are indexed by global world number
(`world.world_offset`), so G shards reproduce one big batch bit-for-bit.

Two things live here besides the shard arithmetic:

`Rendezvous`         the bookkeeping group.  The default process group is ALWAYS gloo (host values are what
                     is reduced: Python floats and small records); when the caller asks for RCCL ("nccl" /
                     "auto") an RCCL subgroup is tried on top of it -- created, exercised with one
                     all-reduce under a deadline, and adopted only if EVERY rank reports success (agreed
                     over gloo).  A failing or hanging RCCL init therefore costs a warning, not the run:
                     the step path has no collective, so nothing measured depends on which backend carries
                     the barrier.
`spawn_local_ranks`  `python bench.py --gpus N` without a launcher: the parent starts N workers (RANK /
                     LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rank r -> GPU r), relays rank
                     0's stdout, and fails if any worker fails.
"""
import datetime
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """(offset, count) of rank's contiguous slice; counts differ by at most one."""
    base, rem = divmod(int(global_batch), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def _sync(device):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def barrier(device=None):
    _sync(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    _sync(device)


def _reduce(value, op, device="cpu", group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op, group=group)
    return float(t.item())


def reduce_max(value, device="cpu"):
    """MAX over ranks of a Python float (elapsed seconds)."""
    return _reduce(value, dist.ReduceOp.MAX, device)


def reduce_sum(value, device="cpu"):
    return _reduce(value, dist.ReduceOp.SUM, device)


def whole_job_throughput(local_units, local_seconds, device="cpu"):
    """Units all ranks processed / slowest rank's time (the bench contract's `value`)."""
    return reduce_sum(local_units, device) / reduce_max(local_seconds, device)


class Rendezvous(object):
    """The bookkeeping group of one job: barrier, MAX / SUM of host floats, gather of small records.

    backend: "gloo"  gloo only;  "nccl"  RCCL or fail;  "auto"  RCCL when every rank can bring it up, else gloo.
    `self.backend` says what carries the barrier ("none" for a single process), `self.note` why a fallback
    happened.  Reductions and gathers of host values always travel over gloo (they are host values)."""

    def __init__(self, rank=None, world=None, device=None, backend="auto", nccl_deadline_s=40.0, timeout_s=1800.0):
        self.rank = int(os.environ.get("RANK", "0") if rank is None else rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1") if world is None else world)
        self.device = torch.device(device) if device is not None else None
        self.backend, self.note, self._nccl, self._abandoned, self._stuck = "none", None, None, False, False
        self._nccl_candidate = None
        self.barrier_fallbacks = 0     # RCCL barriers that failed at run time and were redone over gloo
        if self.world == 1:
            return
        if backend not in ("auto", "nccl", "gloo"):
            raise ValueError("backend must be auto / nccl / gloo, not %r" % (backend,))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=timeout_s))
        self.backend = "gloo"
        if backend in ("auto", "nccl"):
            if self.device is None or self.device.type != "cuda":
                ok, why = False, "no GPU device for RCCL"
            else:
                ok, why = self._try_nccl(nccl_deadline_s)
            everyone = _reduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) == 1.0
            cand = getattr(self, "_nccl_candidate", None)
            if everyone:
                self._nccl = cand        # adopted by the main thread, and only on a unanimous yes
                self.backend = "nccl"
            else:
                reasons = self.gather(why)
                self.note = "RCCL not adopted (%s); barrier over gloo" % "; ".join(
                    "rank %d: %s" % (r, w) for r, w in enumerate(reasons) if w)
                self._abandoned = cand is not None or self._stuck
                self._nccl = None
                if backend == "nccl":
                    raise RuntimeError(self.note)

    def _try_nccl(self, deadline_s):
        """One RCCL subgroup + one all-reduce under a deadline.  -> (ok, reason).
        The bring-up runs in a helper thread: communicator creation happens inside the first collective CALL (not behind
        its async handle), so a bring-up that hangs would otherwise hang this process before any deadline could be looked
        at.  A thread that does not come back is left behind (daemon), the vote is "no", and close() leaves the process
        without tearing the half-made communicator down."""
        import threading
        # a hung RCCL op must not take the process down when the deadline gives up on it
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        result = {}

        def bring_up():
            try:
                torch.cuda.set_device(self.device)
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(deadline_s * 4, 120.0)))
                result["group"] = g      # handed to the main thread; adopted there only if the bring-up came back in time
                t = torch.ones(1, device=self.device)
                w = dist.all_reduce(t, group=g, async_op=True)
                t_end = time.time() + deadline_s
                while not w.is_completed():
                    if time.time() > t_end:
                        result["v"] = (False, "RCCL all-reduce did not complete in %.0f s" % deadline_s)
                        return
                    time.sleep(0.02)
                torch.cuda.synchronize(self.device)
                if int(t.item()) != self.world:
                    result["v"] = (False, "RCCL all-reduce returned %r" % (t.item(),))
                    return
                result["v"] = (True, None)
            except Exception as e:   # duplicate GPU, no P2P, missing IPC mode, ...: the reason is reported, not raised
                result["v"] = (False, "%s: %s" % (type(e).__name__,
                                                  str(e).strip().splitlines()[0][:200] if str(e).strip() else ""))
        th = threading.Thread(target=bring_up, name="rccl-bring-up", daemon=True)
        th.start()
        th.join(deadline_s + 10.0)
        if th.is_alive() or "v" not in result:
            self._stuck = True       # (a thread that returns later finds nobody reading `result`: self._nccl is the main thread's)
            return False, "RCCL bring-up did not return in %.0f s" % (deadline_s + 10.0)
        self._nccl_candidate = result.get("group")
        return result["v"]

    def barrier(self):
        """All ranks meet here, with this rank's GPU idle on both sides.  Over RCCL when the bring-up adopted it; an RCCL
        barrier that raises at run time (the adopted group's first real use on an N-GPU box) is not fatal: every rank then
        agrees over gloo -- one MIN-reduce of "mine worked" -- to drop RCCL for the rest of the job (`backend` becomes
        "gloo", `note` says why, one warning on stderr), and the barrier is completed over gloo.  Nothing measured
        depends on which backend carries it: the step path has no collective."""
        _sync(self.device)
        if self.world > 1:
            if self._nccl is not None:
                ok, why = True, None
                try:
                    dist.barrier(group=self._nccl, device_ids=[self.device.index])
                    _sync(self.device)
                except Exception as e:
                    ok, why = False, "%s: %s" % (type(e).__name__, (str(e).strip().splitlines() or [""])[0][:200])
                # the vote is what completes the barrier when RCCL failed somewhere; it costs one small gloo all-reduce
                # (outside the timed region: barrier() brackets it, and the clock starts after the barrier returns)
                if _reduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) != 1.0:
                    reasons = self.gather(why)
                    self.note = "RCCL barrier failed at run time (%s); barrier over gloo from here on" % "; ".join(
                        "rank %d: %s" % (r, w) for r, w in enumerate(reasons) if w)
                    if self.rank == 0:
                        sys.stderr.write("sharding.Rendezvous: %s\n" % self.note)
                    self._nccl, self.backend, self._abandoned = None, "gloo", True
                    self.barrier_fallbacks += 1
                    dist.barrier()
            else:
                dist.barrier()
        _sync(self.device)

    def reduce_max(self, value):
        return _reduce(value, dist.ReduceOp.MAX) if self.world > 1 else float(value)

    def reduce_sum(self, value):
        return _reduce(value, dist.ReduceOp.SUM) if self.world > 1 else float(value)

    def gather(self, obj):
        """Every rank's `obj`, in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def close(self):
        """Leave the job.  Returns True when the process group was torn down normally, False when an RCCL communicator
        was abandoned (its bring-up failed, hung, or a barrier failed at run time): tearing such a communicator down may
        block, so the group is left as it is and THE CALLER decides how to end the process (bench.py: flush its output,
        then `os._exit(status)` with the status it would have returned).  A library call never exits the process."""
        if self.world > 1 and dist.is_initialized():
            try:
                dist.barrier()
            except Exception:
                pass
            if self._abandoned:
                return False
            dist.destroy_process_group()
        return True


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _visible_gpu_ids(env=None):
    """The GPU ids a process with environment `env` may use, as the strings HIP_VISIBLE_DEVICES takes (None: no restriction)."""
    env = os.environ if env is None else env
    vis = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
    if not vis:
        return None
    return [v.strip() for v in vis.split(",") if v.strip()]


def rank_cpu_slice(local_rank, local_world, allowed=None):
    """The CPUs rank `local_rank` of `local_world` ranks on this node is pinned to: an even, contiguous slice of the CPUs
    the process may run on (all of them when there are fewer CPUs than ranks).  The step path leaves the host idle inside
    the timed region (graph replay); the slice keeps N Python hosts from migrating over each other around it."""
    cpus = sorted(allowed if allowed is not None else
                  (os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(os.cpu_count() or 1)))
    n, g = len(cpus), max(1, int(local_world))
    if n < g:
        return cpus
    per = n // g
    return cpus[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank, local_world):
    """sched_setaffinity to rank_cpu_slice; -> the CPU list in force afterwards (a description for the bench record)."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    try:
        sl = rank_cpu_slice(local_rank, local_world)
        if sl:
            os.sched_setaffinity(0, sl)
        return sorted(os.sched_getaffinity(0))
    except OSError:
        return None


def spawn_local_ranks(argv, n, env=None, one_device=False, timeout_s=3600.0, python=None, gpu_slots=None):
    """Start `n` copies of `python argv...` as ranks 0..n-1 of one job on this node (what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would do), relay rank 0's stdout to ours and
    every rank's stderr to ours, and return the job's exit code: 0 only if every rank exited 0.
    Rank r is handed exactly ONE GPU -- HIP_VISIBLE_DEVICES = the r-th GPU this process may use, so the worker's
    `cuda:0` is that GPU and a mis-mapped rank cannot land on a neighbour's device -- and LOCAL_RANK 0 with
    MPE_LOCAL_RANK = r for the CPU slice (`one_device`: every rank gets the first GPU, the one-GPU rehearsal;
    `gpu_slots`: rank r gets visible GPU number gpu_slots[r] instead of number r)."""
    base = dict(os.environ if env is None else env)
    base["MASTER_ADDR"] = "127.0.0.1"
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["MPE_SELF_SPAWNED"] = "1"
    ids = _visible_gpu_ids(base)
    procs = []
    for r in range(n):
        e = dict(base)
        e["RANK"] = str(r)
        e["LOCAL_RANK"] = "0"              # the one GPU the rank sees
        e["MPE_LOCAL_RANK"] = str(r)       # its place among the node's ranks (CPU slice, records)
        k = 0 if one_device else (int(gpu_slots[r]) if gpu_slots is not None else r)
        gpu = ids[k] if ids is not None and k < len(ids) else str(k)
        e["HIP_VISIBLE_DEVICES"] = gpu
        e.pop("CUDA_VISIBLE_DEVICES", None)
        e["MPE_GPU_ID"] = gpu
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_end = time.time() + timeout_s
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:                       # one rank failed: the others would wait for it in a barrier
                rc = bad[0] if bad[0] > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > t_end:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:            # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.kill()
                p.wait()
                rc = rc or 1
    reader.join(5.0)
    return rc, "".join(c for c in chunks if c)
