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


def _stop_group(p, sig):
    """signal the child's whole process group (it leads its own session), falling back to the child alone"""
    try:
        os.killpg(p.pid, sig)
    except (ProcessLookupError, PermissionError, OSError):
        try:
            p.send_signal(sig)
        except (ProcessLookupError, OSError):
            pass


def self_launch(script, argv, world, poll_s=0.05):
    """`python bench.py --gpus N` without a launcher: start N fresh single-GPU ranks of `script` (the
    environment torch.distributed.run would give them, rendezvous on 127.0.0.1) and return the worst exit
    code.  The calling process must not have touched the GPU and never does: the ranks are CHILD
    processes, nothing is re-executed.  Rank 0 inherits stdout (its one JSON line is the run's output);
    the other ranks' stdout goes to stderr.  No rank outlives this call: if one fails, or this process is
    interrupted or told to terminate (harness timeout, SIGTERM, Ctrl-C), every live rank's process group gets
    SIGTERM, then SIGKILL -- never a rank left holding its GPU at a barrier."""
    import signal
    port = free_port()
    procs = []

    def interrupted(signum, frame):
        raise KeyboardInterrupt("signal %d" % signum)

    handlers = {}
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            handlers[sig] = signal.signal(sig, interrupted)
        except ValueError:              # not the main thread: the finally clause still runs on exceptions
            pass
    worst = 0
    try:
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                       LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       PHNREC_SELF_LAUNCHED="1")
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                          stdout=None if rank == 0 else sys.stderr, start_new_session=True))
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
                        _stop_group(q, signal.SIGTERM)
            time.sleep(poll_s)
    except KeyboardInterrupt:
        worst = worst or 130
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            _stop_group(p, signal.SIGTERM)
        deadline = time.time() + 5.0
        for p in live:
            try:
                p.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                _stop_group(p, signal.SIGKILL)
                p.wait()
        for sig, h in handlers.items():
            signal.signal(sig, h)
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
        self.host_group = None

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
            # a host-side group for waits that must not occupy the GPUs (an RCCL barrier is a kernel that spins on
            # every waiting rank's device)
            self.host_group = dist.new_group(backend="gloo") if backend != "gloo" else None
        return self

    def host_barrier(self):
        """all ranks meet on the CPU (gloo); the GPUs stay free for whoever is still working"""
        if self.pg:
            import torch.distributed as dist
            dist.barrier(group=self.host_group) if self.host_group is not None else dist.barrier()

    def barrier(self):
        if self.pg:
            import torch.distributed as dist
            if self.backend == "nccl":
                # name the device: an RCCL barrier otherwise guesses it from the rank number and warns
                import torch
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
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


def timed_steps(ranks, step, sync, steps, warmup, device=None, detail=None):
    """The bench contract: W untimed steps, then EXACTLY K steps bracketed by a barrier
    and a device synchronise on both sides; the elapsed time is the MAX over ranks.

    Returned (= `ms_per_step`, `value`): the contract's bracket -- the clock stops behind the closing barrier and the
    synchronise that follows it, on every rank, and the MAX over ranks is taken.  `detail` (a dict), when given, also
    receives "before_closing_barrier": the MAX over ranks of each rank's clock at its OWN synchronise behind step K, i.e.
    without the closing collective (30-100 us of RCCL at N > 1 that an N = 1 run, which has no process group, never pays:
    1-3 % of a 20 x 0.2 ms region).  Disclosed beside the contract's figure, never in place of it."""
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
    dt_own = time.perf_counter() - t0
    ranks.barrier()
    sync()
    dt = time.perf_counter() - t0
    if detail is not None:
        detail["before_closing_barrier"] = ranks.max_float(dt_own, device=device)
    return ranks.max_float(dt, device=device)
