"""ShardedChargingHub -- the multi-GPU form of the path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), env shards are contiguous ranges of the global env index, and the only collective is ONE gather
of the packed per-step outputs (obs[D], reward, done) to rank 0.  Nothing else crosses GPUs: environments are
independent, and the Philox streams are keyed by the GLOBAL env id, so results do not depend on the sharding.

torch is used here for what the C ABI does not do: device buffers, streams and the process group.  The step itself
is libchub's (``engine='hip'``).  ``engine`` can also be any object with the small interface below, which is how
the CPU test drives the same sharding / gather logic over gloo without a GPU.

Engine interface:  reset() -> None;  step(actions_local) -> None;  packed -> torch.Tensor [n_local, D+2] f32
                   (filled by step: obs, reward, done);  reset_obs -> torch.Tensor [n_local, D] f32
"""
import os


def shard_range(total_envs, world_size, rank):
    """Contiguous shard of the global env index: returns (env_id0, n_local)."""
    if total_envs % world_size != 0:
        raise ValueError("total_envs (%d) must be divisible by the number of ranks (%d)" % (total_envs, world_size))
    per = total_envs // world_size
    return rank * per, per


class HipEngine(object):
    """The real thing: libchub on this rank's GPU, outputs written straight into torch device buffers."""

    def __init__(self, n_local, env_id0, device_index, seed, hub_kwargs):
        import torch  # before libchub: both must share one HIP runtime

        from .vec_env import VecChargingHub

        self.torch = torch
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.vec = VecChargingHub(n_local, seed=seed, rng="philox", device=device_index, env_id0=env_id0, **hub_kwargs)
        D, A = self.vec.obs_dim, self.vec.act_dim
        self.obs_dim, self.act_dim = D, A
        self.packed = torch.empty((n_local, D + 2), dtype=torch.float32, device=self.dev)
        self.reset_obs = torch.empty((n_local, D), dtype=torch.float32, device=self.dev)

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def reset(self):
        self.vec.reset_device(self.reset_obs.data_ptr(), stream=self._stream())

    def step(self, actions_local):
        a = actions_local
        assert a.is_cuda and a.dtype == self.torch.float32 and a.is_contiguous() and tuple(a.shape) == (self.vec.n_envs, self.act_dim)
        self.vec.step_device_packed(a.data_ptr(), self.packed.data_ptr(), stream=self._stream())


class ShardedChargingHub(object):
    def __init__(self, total_envs, hub_kwargs, seed=0, engine="hip", group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_envs = int(total_envs)
        self.env_id0, self.n_local = shard_range(self.total_envs, self.world, self.rank)
        if engine == "hip":
            local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
            engine = HipEngine(self.n_local, self.env_id0, local_rank, seed, hub_kwargs)
        elif callable(engine):
            engine = engine(self.n_local, self.env_id0)
        self.engine = engine
        self.obs_dim, self.act_dim = engine.obs_dim, engine.act_dim
        self._gather_list = None
        if self.rank == 0 and self.world > 1:
            self._gather_list = [torch.empty_like(engine.packed) for _ in range(self.world)]
            self._obs_list = [torch.empty_like(engine.reset_obs) for _ in range(self.world)]

    def _gather(self, local, lst):
        if self.world == 1:
            return local
        self.dist.gather(local, gather_list=lst if self.rank == 0 else None, dst=0, group=self.group)
        return self.torch.cat(lst, dim=0) if self.rank == 0 else None

    def reset(self):
        """-> obs [total_envs, D] on rank 0 (None elsewhere)"""
        self.engine.reset()
        return self._gather(self.engine.reset_obs, getattr(self, "_obs_list", None))

    def step(self, actions_local):
        """actions_local: this rank's [n_local, A] slice of the policy output.
        -> (obs [total, D], reward [total], done [total]) on rank 0, None elsewhere: ONE gather per step."""
        self.engine.step(actions_local)
        full = self._gather(self.engine.packed, self._gather_list)
        if full is None:
            return None
        D = self.obs_dim
        return full[:, :D], full[:, D], full[:, D + 1] > 0.5
