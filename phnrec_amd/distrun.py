"""One-process-per-GPU harness: rendezvous, barriers, max-over-ranks timing, sharding.

The LCRC posterior path has NO exchange step: utterances (file-list lines) are
independent, every GPU holds a full replica of the 2-8 MB of weights, and results
are gathered by concatenating per-utterance outputs in list order on the host
(SURVEY.md 8e).  So torch.distributed is used only for what a launcher needs:
agree on the start/stop of the timed region and reduce the elapsed time (MAX).
No data-path collective exists and none is invented.
"""
import os
import socket
import subprocess
import sys
import time


def device_map(world):
    """Physical GPU of every local rank.  Default: rank r -> GPU r.  PHNREC_DEVICE_MAP="0,0" maps several
    ranks onto one GPU (a 1-GPU box can then exercise the N-rank path; RCCL refuses two ranks on one device,
    so such a run makes its rendezvous over gloo and says so in its output)."""
    spec = os.environ.get("PHNREC_DEVICE_MAP", "").strip()
    if not spec:
        return list(range(world))
    try:
        m = [int(x) for x in spec.split(",")]
    except ValueError:
        raise SystemExit("PHNREC_DEVICE_MAP must be a comma-separated list of GPU indices, got %r" % spec)
    if len(m) < world or min(m) < 0:
        raise SystemExit("PHNREC_DEVICE_MAP=%s names %d devices, %d ranks were asked for" % (spec, len(m), world))
    return m[:world]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(script, argv, world, poll_s=0.05):
    """`python bench.py --gpus N` without a launcher: start N fresh single-GPU ranks of `script` (the
    environment torch.distributed.run would give them, rendezvous on 127.0.0.1) and return the worst exit
    code.  The calling process must not have touched the GPU and never does: the ranks are CHILD
    processes, nothing is re-executed.  Rank 0 inherits stdout (its one JSON line is the run's output);
    the other ranks' stdout goes to stderr.  If a rank fails, the others are terminated (by PID) instead
    of being left waiting at a barrier."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PHNREC_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in live:          # a rank died: the rest would hang at the next barrier
                    q.terminate()
        time.sleep(poll_s)
    return worst


class Ranks:
    """World description read from the torch.distributed.run environment."""

    def __init__(self, gpus=1):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.gpus = gpus
        self.launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
        self.launcher = ("self" if os.environ.get("PHNREC_SELF_LAUNCHED") == "1" else
                         "torch.distributed.run" if self.launched else "none")
        self.pg = False
        self.backend = None

    def init(self, backend):
        # also under a launcher with a single rank, so that the rendezvous / RCCL path is the one
        # exercised whenever the driver starts us through torch.distributed.run
        if self.world > 1 or (self.launched and backend == "nccl"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.pg = True
            self.backend = backend
            # what the process group itself says, not what the environment promised
            self.world = dist.get_world_size()
        return self

    def barrier(self):
        if self.pg:
            import torch.distributed as dist
            dist.barrier()

    def max_float(self, v, device=None):
        if not self.pg:
            return float(v)
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v)], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, v, device=None):
        if not self.pg:
            return int(v)
        import torch
        import torch.distributed as dist
        t = torch.tensor([int(v)], dtype=torch.int64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def finish(self):
        if self.pg:
            import torch.distributed as dist
            dist.destroy_process_group()
            self.pg = False


def shard_by_frames(lengths, world):
    """Greedy longest-first partition of utterances over `world` replicas by frame count.

    Returns a list (len = number of utterances) of owning ranks.  Deterministic, so every
    rank computes the same assignment from the same file list without communicating.
    """
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    owner = [0] * len(lengths)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(lengths[i])
    return owner


def timed_steps(ranks, step, sync, steps, warmup, device=None):
    """The bench contract: W untimed steps, then EXACTLY K steps bracketed by a barrier
    and a device synchronise on both sides; the elapsed time is the MAX over ranks."""
    import time
    for _ in range(warmup):
        step()
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    ranks.barrier()
    sync()
    dt = time.perf_counter() - t0
    return ranks.max_float(dt, device=device)
