"""One-process-per-GPU harness: rendezvous, barriers, max-over-ranks timing, sharding.

The LCRC posterior path has NO exchange step: utterances (file-list lines) are
independent, every GPU holds a full replica of the 2-8 MB of weights, and results
are gathered by concatenating per-utterance outputs in list order on the host
(SURVEY.md 8e).  So torch.distributed is used only for what a launcher needs:
agree on the start/stop of the timed region and reduce the elapsed time (MAX).
No data-path collective exists and none is invented.
"""
import os


class Ranks:
    """World description read from the torch.distributed.run environment."""

    def __init__(self, gpus=1):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.gpus = gpus
        self.launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
        self.pg = False

    def init(self, backend):
        # also under a launcher with a single rank, so that the rendezvous / RCCL path is the one
        # exercised whenever the driver starts us through torch.distributed.run
        if self.world > 1 or (self.launched and backend == "nccl"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.pg = True
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
