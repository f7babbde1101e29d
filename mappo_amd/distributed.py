"""Data parallelism over rollout threads: one process per GPU, each owning
n_rollout_threads / world of the threads and the matching [T+1][N/world * M][.] slice of every buffer
array in its own HBM.  Collect, insert and the GAE scan need no communication; per minibatch the ranks
exchange, with `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests):

  * 4 doubles  : {sum ret, sum ret^2, sum active, B}  -> global ValueNorm statistics and loss denominators
  * 1 flat fp32: the joint actor|critic gradient (89 KB at BASELINE config 2)  -> identical clip + Adam on every rank
and once per train(): 3 doubles (advantage moments) and 4 doubles (logged statistics).

Gradients are computed against GLOBAL denominators, so the sum over ranks equals the single-process
gradient (up to fp32 summation order) and replicas stay identical without a broadcast."""
import os

import torch


def shard_threads(n_rollout_threads, rank, world):
    """Contiguous block of rollout threads owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_rollout_threads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sampling_seed(seed, rank=None):
    """Seed of the action-sampling Philox stream of one rank.  Every rank is built with the same `args.seed` so that the
    replicas' PARAMETERS start identical (torch.manual_seed), but the sampling stream is indexed by the LOCAL row, so an
    un-keyed seed would give row r of every shard the same uniform at every step — exploration noise perfectly correlated
    across the shards the gradient all-reduce treats as independent.  The rank is folded into the sampling seed only."""
    if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return (int(seed) + int(rank) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


class DataParallel:
    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs torch.distributed.init_process_group (nccl on GPUs, gloo in CPU tests)")
        self._dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.world_is_gpu = dist.get_backend(group) == "nccl"      # hipGraph segments only make sense on the GPU backend

    def all_reduce_sum_(self, tensor):
        """In-place SUM on the tensor's own device / current stream (no host sync for nccl)."""
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_max_(self, tensor):
        self._dist.all_reduce(tensor, op=self._dist.ReduceOp.MAX, group=self.group)
        return tensor

    def broadcast_(self, tensor, src=0):
        self._dist.broadcast(tensor, src=src, group=self.group)
        return tensor


def init_from_env(backend=None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return DataParallel()
