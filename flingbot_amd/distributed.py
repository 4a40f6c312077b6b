"""Episode sharding across GPUs: one process per GPU, episodes never communicate.

The reference parallelises with one Ray actor process per environment and fractional GPU shares (utils.py:144-157); here
each rank owns one MI355X and a batch of episodes (global episode g lives on rank g // episodes_per_rank).  The only
exchange step of the path is the episode-batch gather of the per-episode coverage rewards (RCCL all_gather over xGMI
with the "nccl" backend; "gloo" in the CPU tests): 4 bytes per episode, latency bound, once per action -- never per
sim step.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns
    (rank, local_rank, world).  World size 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def episode_range(rank, episodes_per_rank):
    """Global episode ids owned by `rank` (weak scaling: the per-rank batch is fixed)."""
    return range(rank * episodes_per_rank, (rank + 1) * episodes_per_rank)


def gather_rewards(local_rewards, device=None):
    """all_gather of the per-episode rewards of every rank -> 1-D float32 tensor ordered by global episode id."""
    t = torch.as_tensor(local_rewards, dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return torch.cat(parts)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the bench's timed-region length)."""
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class SharedTaskCounter:
    """The reference's task queue is GLOBAL: one TaskLoader actor hands the next task to whichever environment asks first
    (utils.py setup_envs: `get_task_fn = lambda: ray.get(task_loader.get_next_task.remote())`), so a worker that draws long
    episodes simply takes fewer of them.  Across ranks the same thing is one atomic counter on the process group's own
    rendezvous store (c10d TCPStore `add`: no extra socket, no collective): claim(k) returns up to k task indices nobody else
    has, [] when the set is used up.  World size 1 (no process group): a local counter.  `key` must be the same on every rank
    and fresh per task set."""

    def __init__(self, n_tasks, key="tasks"):
        self.n, self.local = int(n_tasks), 0
        self.store = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.store = dist.PrefixStore("flingsim/" + str(key), dist.distributed_c10d._get_default_store())
        self.claimed = []

    def claim(self, k=1):
        k = int(k)
        if k <= 0:
            return []
        if self.store is not None:
            end = int(self.store.add("next", k))
        else:
            self.local += k
            end = self.local
        got = [i for i in range(end - k, end) if i < self.n]
        self.claimed += got
        return got


def sum_over_ranks(array, device=None):
    """SUM all-reduce of a float64 array (every rank contributes its own episodes' entries, zeros elsewhere)."""
    t = torch.as_tensor(array, dtype=torch.float64).clone()
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def _hash63(text):
    import hashlib
    return int.from_bytes(hashlib.sha1(str(text).encode()).digest()[:8], "big") >> 1


def rank_census(device_identity, arch=""):
    """What the process group ITSELF saw, as opposed to what the environment variables claim: every rank contributes
    (rank, LOCAL_RANK, hash of its device's identity -- the PCI bus id, FlingSim.device_key() --, hash of its architecture
    name) to one all_gather (RCCL when the backend is "nccl", on the rank's own device) and every rank gets the same answer:
        ranks_seen        distinct rank numbers in the gathered table (== world size when the collective spans the job)
        distinct_devices  distinct physical devices behind those ranks (one process per GPU <=> == world size)
        distinct_archs    1 on a homogeneous node
        backend, collective_library   what carried the gather ("nccl" + RCCL's version on ROCm, "gloo" in the CPU tests)
    bench.py and evaluate.py refuse to report a multi-GPU figure when distinct_devices != world size (two ranks on one GPU)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    mine = torch.tensor([rank, local_rank, _hash63(device_identity), _hash63(arch)], dtype=torch.int64)
    out = {"backend": "none (world size 1: no process group)", "collective_library": None}
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        backend = str(dist.get_backend())
        out["backend"] = backend
        if backend == "nccl":
            mine = mine.to(torch.device("cuda", torch.cuda.current_device()))
            try:
                out["collective_library"] = "RCCL/NCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                out["collective_library"] = "RCCL (version unavailable)"
        rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
        table = torch.stack(rows).cpu()
        out["world_size"] = int(dist.get_world_size())
    else:
        table = mine.reshape(1, 4)
        out["world_size"] = 1
    out["ranks_seen"] = int(torch.unique(table[:, 0]).numel())
    out["distinct_devices"] = int(torch.unique(table[:, 2]).numel())
    out["distinct_archs"] = int(torch.unique(table[:, 3]).numel())
    out["local_ranks"] = [int(v) for v in table[:, 1]]
    return out
