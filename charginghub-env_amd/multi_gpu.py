"""The multi-GPU form of the path without PyTorch: ctypes + numpy over libchub's own RCCL leg (include/chub.h, chub_comm_*).

One process per GPU (launched by any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK, e.g. ``python -m
torch.distributed.run`` -- only its environment variables are used).  Env shards are contiguous ranges of the global env
index; per step each shard's packed output [n_local, D+2] f32 (obs, reward, done) goes to rank 0 in ONE grouped
ncclSend / ncclRecv enqueued on the same HIP stream as the step kernels (``chub_step_gather``): no host wait per step.

The 128-byte RCCL id travels from rank 0 to the other ranks of the node through a file in a private directory (single node: the
north star's 8 GPUs of one node) -- the launcher's (CHUB_RENDEZVOUS_DIR: `bench.py --gpus N` makes one per launch) or one under
/tmp keyed on user, port and launcher; nothing else is exchanged on the host side.
"""
import contextlib
import ctypes as C
import os
import sys
import time

import numpy as np

# the pool's host driver only supports dmabuf IPC: RCCL's peer-to-peer setup between the ranks of a node needs this before the
# HIP runtime initialises (it is exported in the images this runs in; a bare environment gets it here)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import _lib  # noqa: E402
from ._lib import check, load_library  # noqa: E402


class DeviceBuffer(object):
    """a plain device allocation (chub_malloc_device) with numpy round trips"""

    def __init__(self, nbytes, device=0):
        self._lib = load_library()
        self.device = int(device)
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(self._lib.chub_malloc_device(self.device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def to_host(self, dtype, shape, stream=0):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self._lib.chub_copy_to_host(self.device, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, stream or None))
        return out

    def from_host(self, a, stream=0):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(self._lib.chub_copy_to_device(self.device, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, stream or None))

    def free(self):
        if getattr(self, "ptr", None):
            self._lib.chub_free_device(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream(object):
    def __init__(self, device=0):
        self._lib = load_library()
        self.device = int(device)
        p = C.c_void_p()
        check(self._lib.chub_stream_create(self.device, C.byref(p)))
        self.ptr = p.value

    def sync(self):
        check(self._lib.chub_stream_sync(self.device, self.ptr))

    def destroy(self):
        if getattr(self, "ptr", None):
            self._lib.chub_stream_destroy(self.device, self.ptr)
            self.ptr = None


RENDEZVOUS_ENV = "CHUB_RENDEZVOUS_DIR"  # a private directory made by the launcher (bench.py --gpus N makes one per launch)
_ID_MAGIC = b"CHUBID01"


def _launcher_start_time():
    """wall-clock time at which the parent process (the launcher all ranks share) started, or 0.0 if /proc does not say"""
    try:
        with open("/proc/%d/stat" % os.getppid(), "rb") as f:
            fields = f.read().rsplit(b")", 1)[1].split()
        ticks = int(fields[19])  # starttime: field 22 of proc(5), in clock ticks since boot
        with open("/proc/uptime") as f:
            uptime = float(f.read().split()[0])
        return time.time() - uptime + ticks / float(os.sysconf("SC_CLK_TCK"))
    except Exception:
        return 0.0


def _rendezvous_dir():
    """Where rank 0 leaves the RCCL id.  A launcher that passes CHUB_RENDEZVOUS_DIR (a fresh private directory per launch) is
    taken at its word.  Otherwise (torch.distributed.run and the like: only RANK / WORLD_SIZE / MASTER_PORT) a directory
    /tmp/chub_rdv_<uid>_<port>_<run id>_<launcher pid>, mode 0700, which must be a real directory owned by this user."""
    d = os.environ.get(RENDEZVOUS_ENV)
    if d:
        return d, False
    d = "/tmp/chub_rdv_%d_%s_%s_%d" % (os.getuid(), os.environ.get("MASTER_PORT", "0"),
                                       os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat

    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise _lib.ChubError("rendezvous directory %s is not a private directory of this user" % d)
    return d, True


def _id_file_name():
    """one id file per incarnation of the worker group: a launcher that restarts its workers (torchrun --max-restarts: same pid,
    port and run id, TORCHELASTIC_RESTART_COUNT counts up) must not hand the new ranks the id of the group that crashed"""
    n = os.environ.get("TORCHELASTIC_RESTART_COUNT", "")
    return "rccl_id" + ("_r" + n if n.isdigit() and int(n) > 0 else "")


def exchange_unique_id(rank, world, timeout=None, make_id=None):
    """rank 0 makes the RCCL id (chub_comm_unique_id) and publishes it; the others wait for it.

    The file is created with O_EXCL | O_NOFOLLOW, mode 0600, under a temporary name and renamed into place, and carries a
    header (magic, time written).  A reader only accepts an id written after the launcher of THIS launch started, so the
    leftover of a crashed earlier launch (same port, same recycled launcher pid) is never picked up; rank 0 removes such a
    leftover before it writes.  `timeout` seconds (default 120, CHUB_RENDEZVOUS_TIMEOUT) without an id is an error naming
    the path.  make_id: tests inject the id source (the real one needs RCCL)."""
    import struct

    if timeout is None:
        timeout = float(os.environ.get("CHUB_RENDEZVOUS_TIMEOUT", "120"))
    d, shared_tmp = _rendezvous_dir()
    path = os.path.join(d, _id_file_name())
    not_before = _launcher_start_time() - 1.0 if shared_tmp else 0.0
    if rank == 0:
        if make_id is None:
            lib = load_library()
            buf = (C.c_char * 128)()
            check(lib.chub_comm_unique_id(buf))
            uid = bytes(buf)
        else:
            uid = make_id()
        assert len(uid) == 128
        for leftover in (path, path + ".tmp"):
            try:
                os.unlink(leftover)
            except FileNotFoundError:
                pass
        fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(_ID_MAGIC + struct.pack("<d", time.time()) + uid)
        os.rename(path + ".tmp", path)
        return uid
    t0 = time.time()
    why = "no file"
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            with os.fdopen(fd, "rb") as f:
                b = f.read()
            if len(b) == 144 and b[:8] == _ID_MAGIC:
                if struct.unpack("<d", b[8:16])[0] >= not_before:
                    return b[16:]
                why = "only a stale id of an earlier launch"
            else:
                why = "a malformed id file"
        except FileNotFoundError:
            why = "no file"
        if time.time() - t0 > timeout:
            raise _lib.ChubError("rank %d of %d: no RCCL id from rank 0 within %.0f s at %s (%s): is rank 0 running, and do all "
                                 "ranks share %s / MASTER_PORT and the launcher?" % (rank, world, timeout, path, why, RENDEZVOUS_ENV))
        time.sleep(0.01)


def _rendezvous_cleanup():
    d = os.environ.get(RENDEZVOUS_ENV)
    try:
        if d:
            os.unlink(os.path.join(d, _id_file_name()))  # the directory is the launcher's
        else:
            d, _ = _rendezvous_dir()
            os.unlink(os.path.join(d, _id_file_name()))
            os.rmdir(d)
    except OSError:
        pass


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """RCCL prints a banner ("Hostname", "Librccl path") to the C stdout when it initialises; programs that print results on
    stdout (bench.py: one JSON line) want it on stderr"""
    libc = C.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class Comm(object):
    """libchub's communicator: ncclCommInitRank behind the C ABI"""

    def __init__(self, rank=None, world=None, device=None):
        self._lib = load_library()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self.device %= max(1, self._lib.chub_device_count())  # a rank that is shown only its own GPU sees it as ordinal 0
        with _c_stdout_to_stderr():
            uid = exchange_unique_id(self.rank, self.world)
            h = C.c_void_p()
            check(self._lib.chub_comm_create(uid, self.world, self.rank, self.device, C.byref(h)))
            self._h = h
            self.barrier()
        if self.rank == 0:  # every rank holds the id by now
            _rendezvous_cleanup()

    def world_seen(self):
        """ncclCommCount: the communicator's size as RCCL reports it"""
        n = self._lib.chub_comm_world(self._h)
        if n < 0:
            check(n)
        return n

    def ranks_seen(self, stream=0):
        """all-reduce sum of one 1 per rank: how many processes RCCL actually moved data between"""
        n = C.c_int(0)
        check(self._lib.chub_comm_ranks_seen(self._h, C.byref(n), stream or None))
        return n.value

    def gather(self, d_send, d_recv, nbytes, stream=0):
        check(self._lib.chub_comm_gather(self._h, d_send, d_recv or None, int(nbytes), stream or None))

    def set_overlap(self, on=True):
        """gathers on the communicator's own stream, tied to the caller's by events (graph edges inside a capture): chub_comm_set_overlap"""
        check(self._lib.chub_comm_set_overlap(self._h, int(bool(on))))

    def join(self, stream=0):
        """`stream` waits for every overlapped gather still out (chub_comm_join)"""
        check(self._lib.chub_comm_join(self._h, stream or None))

    def gather_us(self, d_send, d_recv, nbytes, stream=0, reps=100):
        """microseconds per gather, `reps` of them back to back between two HIP events (chub_comm_gather_timed); every rank calls it"""
        us = C.c_double(0.0)
        check(self._lib.chub_comm_gather_timed(self._h, d_send, d_recv or None, int(nbytes), stream or None, int(reps), C.byref(us)))
        return us.value

    def max(self, value, stream=0):
        v = C.c_double(float(value))
        check(self._lib.chub_comm_max_f64(self._h, C.byref(v), stream or None))
        return v.value

    def barrier(self, stream=0):
        check(self._lib.chub_comm_barrier(self._h, stream or None))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.chub_comm_destroy(self._h)
            self._h = None


class HipShard(object):
    """this rank's env shard on its GPU: libchub handle + the device buffers of the packed step output"""

    def __init__(self, n_local, env_id0, device, seed, hub_kwargs, total_envs, is_root):
        from .vec_env import VecChargingHub

        self.vec = VecChargingHub(n_local, seed=seed, rng="philox", device=device, env_id0=env_id0, **hub_kwargs)
        self.obs_dim, self.act_dim = self.vec.obs_dim, self.vec.act_dim
        self.device = device
        self.stream = Stream(device)
        row = (self.obs_dim + 2) * 4
        self.packed = [DeviceBuffer(n_local * row, device) for _ in range(2)]
        self.gathered = [DeviceBuffer(total_envs * row, device) if is_root else None for _ in range(2)]
        self.reset_obs = DeviceBuffer(n_local * self.obs_dim * 4, device)
        self.total_envs = total_envs

    def reset(self):
        self.vec.reset_device(self.reset_obs.ptr, stream=self.stream.ptr)

    def step_gather(self, d_actions, b, comm):
        """the multi-GPU step in ONE ABI call: step kernels + the grouped ncclSend / ncclRecv on the same stream"""
        check(self.vec._lib.chub_step_gather(self.vec._h, comm._h, d_actions, self.packed[b].ptr,
                                             self.gathered[b].ptr if self.gathered[b] is not None else None, self.stream.ptr))

    def fetch(self, b):
        return self.gathered[b].to_host(np.float32, (self.total_envs, self.obs_dim + 2), self.stream.ptr)

    def close(self):
        self.stream.sync()
        self.vec.close()
        for buf in self.packed + [g for g in self.gathered if g is not None] + [self.reset_obs]:
            buf.free()
        self.stream.destroy()


class NativeShardedHub(object):
    """Env shard of this rank + the per-step gather, no PyTorch: contiguous shards of the global env index, ONE gather of
    the packed (obs, reward, done) rows to rank 0 per step, double-buffered.  step() returns at once (stream order is the
    only synchronisation); rank 0 reads a gathered block with fetch() when it wants it on the host.

    `comm` / `shard` can be injected (tests drive this same class on CPU, two processes, with a host-side gather and the
    CPU oracle as the shard): comm has .rank, .world and .gather(send, recv, nbytes, stream); a shard without
    step_gather() has .step(actions, b), .packed[b], .gathered[b] and .fetch(b)."""

    def __init__(self, total_envs, hub_kwargs, seed=0, comm=None, shard=None):
        from .sharded import shard_range

        self.comm = comm if comm is not None else Comm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.total_envs = int(total_envs)
        self.env_id0, self.n_local = shard_range(self.total_envs, self.world, self.rank)
        if shard is None:
            shard = HipShard(self.n_local, self.env_id0, self.comm.device, seed, hub_kwargs, self.total_envs, self.rank == 0)
        elif callable(shard):
            shard = shard(self.n_local, self.env_id0, self.total_envs, self.rank == 0)
        self.shard = shard
        self.obs_dim, self.act_dim = shard.obs_dim, shard.act_dim
        self._i = 0

    def reset(self):
        self.shard.reset()

    def step(self, actions):
        """actions: this rank's [n_local, A] f32 actions (a device address for the HIP shard).  Returns the buffer index."""
        b = self._i & 1
        self._i += 1
        if hasattr(self.shard, "step_gather"):
            self.shard.step_gather(actions, b, self.comm)
        else:
            self.shard.step(actions, b)
            row = (self.obs_dim + 2) * 4
            self.comm.gather(self.shard.packed[b], self.shard.gathered[b], self.n_local * row, getattr(self.shard, "stream", None))
        return b

    def fetch(self, b):
        """rank 0: (obs [total, D], reward [total], done [total]) of buffer b on the host"""
        if self.rank != 0:
            return None
        D = self.obs_dim
        full = self.shard.fetch(b)
        return full[:, :D], full[:, D], full[:, D + 1] > 0.5

    def close(self):
        self.shard.close()
