"""The N > 1 path on CPU: two processes over gloo run the product's sharding + one-gather-per-step logic
(charginghub-env_amd/sharded.py) with the CPU oracle standing in for the GPU engine, and rank 0 checks the gathered
result against a single-process run over all envs."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine(object):
    """Checker-side engine with the HipEngine interface (tests only)."""

    def __init__(self, n_local, env_id0, seed, piles):
        import torch
        import orclib

        self.orclib = orclib
        self.cfg = orclib.make_config(piles=piles, types=("fast", "slow"))
        self.h = orclib.orc.orc_vec_create(C.byref(self.cfg), orclib.tables(), n_local, env_id0, orclib.PHILOX, seed)
        self.n = n_local
        self.obs_dim, self.act_dim = 13, piles[0] + piles[1] + 2
        self.packed = torch.zeros((n_local, self.obs_dim + 2), dtype=torch.float32)
        self.reset_obs = torch.zeros((n_local, self.obs_dim), dtype=torch.float32)
        self._obs = np.zeros((n_local, self.obs_dim))
        self._rew = np.zeros(n_local)
        self._done = np.zeros(n_local, dtype=np.uint8)

    def reset(self):
        import torch
        o = self.orclib
        o.orc.orc_vec_reset(self.h, None, None, o.ptr(self._obs))
        self.reset_obs.copy_(torch.from_numpy(self._obs.astype(np.float32)))

    def step(self, actions_local):
        import torch
        o = self.orclib
        a = np.ascontiguousarray(actions_local.numpy(), dtype=np.float32)
        o.orc.orc_vec_step(self.h, o.ptr(a), None, o.ptr(self._obs), o.ptr(self._rew), o.ptr(self._done), 1)
        D = self.obs_dim
        self.packed[:, :D] = torch.from_numpy(self._obs.astype(np.float32))
        self.packed[:, D] = torch.from_numpy(self._rew.astype(np.float32))
        self.packed[:, D + 1] = torch.from_numpy(self._done.astype(np.float32))


def _worker(rank, world, port, total, piles, seed, steps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from charginghub_env_amd.sharded import ShardedChargingHub

        hub = ShardedChargingHub(total, {}, seed=seed,
                                 engine=lambda n, id0: OracleEngine(n, id0, seed, piles))
        rs = np.random.RandomState(11)  # every rank draws the same global action tape and takes its slice
        out = []
        o = hub.reset()
        if rank == 0:
            out.append(o.numpy().copy())
        for t in range(steps):
            act = rs.uniform(-1, 1, size=(total, hub.act_dim)).astype(np.float32)
            local = torch.from_numpy(np.ascontiguousarray(act[hub.env_id0:hub.env_id0 + hub.n_local]))
            res = hub.step(local)
            if rank == 0:
                obs, rew, done = res
                out.append(np.concatenate([obs.numpy(), rew.numpy()[:, None], done.numpy()[:, None].astype(np.float32)], axis=1))
            else:
                assert res is None
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


class _GlooComm(object):
    """host-side gather (gloo) behind the interface of multi_gpu.Comm -- what the no-PyTorch multi-GPU host is given in
    place of libchub's RCCL leg when there is no GPU"""

    def __init__(self, rank, world):
        self.rank, self.world, self.device = rank, world, 0

    def gather(self, send, recv, nbytes, stream=None):
        import torch
        import torch.distributed as dist
        assert send.nbytes == nbytes
        t = torch.from_numpy(send)
        if self.rank == 0:
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.gather(t, gather_list=parts, dst=0)
            recv[...] = torch.cat(parts, dim=0).numpy()
        else:
            dist.gather(t, dst=0)


class _OracleShard(object):
    """the CPU oracle as this rank's shard (tests only): numpy buffers instead of device buffers"""

    def __init__(self, n_local, env_id0, total, is_root, seed, piles):
        self.eng = OracleEngine(n_local, env_id0, seed, piles)
        self.obs_dim, self.act_dim = self.eng.obs_dim, self.eng.act_dim
        self.packed = [np.zeros((n_local, self.obs_dim + 2), dtype=np.float32) for _ in range(2)]
        self.gathered = [np.zeros((total, self.obs_dim + 2), dtype=np.float32) if is_root else None for _ in range(2)]

    def reset(self):
        self.eng.reset()

    def step(self, actions, b):
        import torch
        self.eng.step(torch.from_numpy(actions))
        self.packed[b][...] = self.eng.packed.numpy()

    def fetch(self, b):
        return self.gathered[b].copy()

    def close(self):
        pass


def _native_worker(rank, world, port, total, piles, seed, steps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from charginghub_env_amd.multi_gpu import NativeShardedHub

        hub = NativeShardedHub(total, {}, seed=seed, comm=_GlooComm(rank, world),
                               shard=lambda n, id0, tot, root: _OracleShard(n, id0, tot, root, seed, piles))
        rs = np.random.RandomState(11)
        out = []
        hub.reset()
        for t in range(steps):
            act = rs.uniform(-1, 1, size=(total, hub.act_dim)).astype(np.float32)
            b = hub.step(np.ascontiguousarray(act[hub.env_id0:hub.env_id0 + hub.n_local]))
            res = hub.fetch(b)
            if rank == 0:
                obs, rew, done = res
                out.append(np.concatenate([obs, rew[:, None], done[:, None].astype(np.float32)], axis=1))
            else:
                assert res is None
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_native_sharded_hub_two_ranks_host_gather():
    """multi_gpu.NativeShardedHub (the no-PyTorch multi-GPU host: shards, double-buffered packed rows, one gather per step)
    with a host-side gloo gather injected in place of libchub's RCCL leg and the CPU oracle as the shard"""
    import torch.multiprocessing as mp

    total, piles, seed, steps, world = 12, (20, 25), 77, 12, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, total, piles, seed, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import torch
    eng = OracleEngine(total, 0, seed, piles)
    rs = np.random.RandomState(11)
    eng.reset()
    for t in range(steps):
        act = rs.uniform(-1, 1, size=(total, eng.act_dim)).astype(np.float32)
        eng.step(torch.from_numpy(act))
        assert np.array_equal(got[t], eng.packed.numpy()), t


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp

    total, piles, seed, steps, world = 12, (20, 25), 4242, 20, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, piles, seed, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over all envs
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    eng = OracleEngine(total, 0, seed, piles)
    rs = np.random.RandomState(11)
    eng.reset()
    assert np.array_equal(got[0], eng.reset_obs.numpy())
    for t in range(steps):
        act = rs.uniform(-1, 1, size=(total, eng.act_dim)).astype(np.float32)
        import torch
        eng.step(torch.from_numpy(act))
        assert np.array_equal(got[t + 1], eng.packed.numpy()), t
