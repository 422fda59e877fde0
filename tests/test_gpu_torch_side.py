"""The multi-GPU orchestration of bench.py / sharded.py on the one GPU a test box has (plus the device-resident torch adapter): a world of ONE rank over the
`nccl` (= RCCL) backend, in a child process.  It cannot show scaling, but it runs the very calls the N > 1 bench makes
(process group with device_id, async gather into a gather list, wait, barrier, all_reduce of the timing) against the real
RCCL of the image, with the HIP engine writing the packed step output that is gathered."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["CHUB_ROOT"])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev)
from charginghub_env_amd.sharded import ShardedChargingHub
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
          init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
hub = ShardedChargingHub(256, kw, seed=5, engine="hip")
obs0 = hub.reset()
eng = hub.engine
gathered = [torch.empty_like(eng.packed)]
a = torch.rand((256, hub.act_dim), device=dev) * 2 - 1
for i in range(4):
    eng.step(a)
    work = dist.gather(eng.packed, gather_list=gathered, dst=0, async_op=True)   # bench.py's per-step collective
    work.wait()
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
assert torch.equal(gathered[0], eng.packed) and bool(torch.isfinite(gathered[0]).all())
assert obs0.shape == (256, hub.obs_dim)
dist.destroy_process_group()
print("RCCL_SINGLE_OK")
'''


def test_rccl_world_of_one():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CHUB_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


NATIVE_CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["CHUB_ROOT"])
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
assert "torch" not in sys.modules
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
          init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
n = 512
hub = multi_gpu.NativeShardedHub(n, kw, seed=9)          # RANK / WORLD_SIZE / LOCAL_RANK from the environment: a world of one
ref = chub.VecChargingHub(n, seed=9, **kw)
acts = [multi_gpu.DeviceBuffer(n * hub.act_dim * 4) for _ in range(2)]
host_acts = []
for b, a in enumerate(acts):
    hub.shard.vec.random_actions_device(a.ptr, 77, b, hub.shard.stream.ptr)
    host_acts.append(a.to_host(np.float32, (n, hub.act_dim), hub.shard.stream.ptr))
hub.reset(); ref.reset()
for t in range(100):
    if t == 96:
        hub.reset(); ref.reset()
    b = hub.step(acts[t & 1].ptr)                         # chub_step_gather: step kernels + grouped ncclSend/ncclRecv, one stream
    obs, rew, done = hub.fetch(b)
    ro, rr, rd, _ = ref.step(host_acts[t & 1])
    assert np.array_equal(obs, ro) and np.array_equal(rew, rr) and np.array_equal(done, rd), t
assert abs(hub.comm.max(2.5, hub.shard.stream.ptr) - 2.5) < 1e-12
hub.comm.barrier(hub.shard.stream.ptr)
# the same steps issued from C with the communicator (chub_run_steps: what bench.py calls at N > 1): steps 100 .. 199 of the run,
# a reset at the day boundary inside the span, the last two gathered blocks against the single handle
import ctypes as C
sh = hub.shard
c_acts = (C.c_void_p * 2)(acts[0].ptr, acts[1].ptr)
c_packed = (C.c_void_p * 2)(sh.packed[0].ptr, sh.packed[1].ptr)
c_gath = (C.c_void_p * 2)(sh.gathered[0].ptr, sh.gathered[1].ptr)
multi_gpu.check(sh.vec._lib.chub_run_steps(sh.vec._h, hub.comm._h, c_acts, 2, c_packed, c_gath, sh.reset_obs.ptr, 100, 100, sh.stream.ptr))
want = {}
for t in range(100, 200):
    if t == 192:
        ref.reset()
    want[t & 1] = ref.step(host_acts[t & 1])
D = hub.obs_dim
for b in (0, 1):
    full = sh.gathered[b].to_host(np.float32, (n, D + 2), sh.stream.ptr)
    assert np.array_equal(full[:, :D], want[b][0]) and np.array_equal(full[:, D], want[b][1]) and np.array_equal(full[:, D + 1] > 0.5, want[b][2]), b
hub.close(); hub.comm.close(); ref.close()
assert "torch" not in sys.modules
print("NATIVE_RCCL_OK")
'''


def test_native_rccl_gather_world_of_one():
    """libchub's own RCCL leg (chub_comm_*, chub_step_gather) from a host without PyTorch: the gathered block equals the
    single-handle run step for step; max / barrier run through RCCL"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CHUB_ROOT=root, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29731",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", NATIVE_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NATIVE_RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


OVERLAP_CHILD = r'''
import os, sys, ctypes as C
sys.path.insert(0, os.environ["CHUB_ROOT"])
import numpy as np
import charginghub_env_amd as chub
from charginghub_env_amd import multi_gpu
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
          init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.02, renew_fluctuate=0.1)
n, STEPS = 3000, 192
comm = multi_gpu.Comm(0, 1, 0)
lib = chub.load_library()
def run(form):
    v = chub.VecChargingHub(n, seed=11, **kw)
    D, A = v.obs_dim, v.act_dim
    st = multi_gpu.Stream(0)
    acts = [multi_gpu.DeviceBuffer(n * A * 4) for _ in range(4)]
    for b, a in enumerate(acts):
        v.random_actions_device(a.ptr, 77, b, st.ptr)
    packed = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(2)]        # the send blocks: double-buffered, as everywhere
    gathered = [multi_gpu.DeviceBuffer(n * (D + 2) * 4) for _ in range(STEPS)]  # what arrives on "rank 0": one block per step, kept
    obs0 = multi_gpu.DeviceBuffer(n * D * 4)
    comm.set_overlap(form != "serial")
    def two_days():
        for i in range(STEPS):
            if i % 96 == 0:
                v.reset_device(obs0.ptr, stream=st.ptr)
            chub._lib.check(lib.chub_step_gather(v._h, comm._h, acts[i % 4].ptr, packed[i & 1].ptr, gathered[i].ptr, st.ptr))
    if form == "graph":  # (a lock-step graph replays from the clock it was captured at: whole episodes)
        st.sync()
        v.graph_begin(st.ptr)
        two_days()
        g = v.graph_end(st.ptr)
        v.graph_launch(g, st.ptr)
    else:
        two_days()
    comm.join(st.ptr)
    out = [b.to_host(np.float32, (n, D + 2), st.ptr).copy() for b in gathered]
    if form == "graph":
        v.graph_launch(g, st.ptr)  # a second replay: other random numbers (the tick base moved on), same structure -- it runs through
        comm.join(st.ptr); st.sync()
        assert not np.array_equal(gathered[5].to_host(np.float32, (n, D + 2), st.ptr), out[5])
        v.graph_destroy(g)
    comm.set_overlap(False)
    v.close()
    for b in acts + packed + gathered + [obs0]:
        b.free()
    st.destroy()
    return out
ref = run("serial")
assert len(ref) == STEPS and all(np.isfinite(x).all() and np.abs(x).sum() > 0 for x in ref) and (ref[95][:, -1] > 0.5).all() and not (ref[94][:, -1] > 0.5).any()
assert not np.array_equal(ref[0], ref[96])
for form in ("eager", "graph"):
    got = run(form)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), (form, i)
comm.close()
print("OVERLAP_OK")
'''


def test_overlapped_gather_equals_the_serial_form_world_of_one():
    """chub_comm_set_overlap: the gather of step k on the communicator's own stream beside the kernels of step k + 1 (event edges; graph
    edges inside a capture) -- call by call and as replays of a captured graph, every gathered block equals the serial form's bit for bit"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CHUB_ROOT=root, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29733",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", OVERLAP_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OVERLAP_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


TORCH_CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["CHUB_ROOT"])
import numpy as np, torch
import charginghub_env_amd as chub
kw = dict(station_list=[20, 25], station_type_list=["fast", "slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0,
          init_soc=0.2, fc_max_power=100.0, fcev_permeate=0.01)
n = 128
tv = chub.TorchHubVecEnv(n, seed=3, **kw)
ref = chub.VecChargingHub(n, seed=3, **kw)
o = tv.reset()
assert o.is_cuda and np.array_equal(o.cpu().numpy(), ref.reset())
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(100):
    a = torch.rand((n, 47), device="cuda", generator=g) * 2 - 1
    obs, rew, done, info = tv.step(a)
    ro, rr, rd, _ = ref.step(a.cpu().numpy())
    assert np.array_equal(rew.cpu().numpy(), rr) and np.array_equal(done.cpu().numpy(), rd), t
    if rd.all():
        assert t == 95 and np.array_equal(tv.last_obs.cpu().numpy(), ro)
        ro = ref.reset()
    assert np.array_equal(obs.cpu().numpy(), ro), t
# the packed-action form: bits built on the device from the same rows, the same results as step()
tb = chub.TorchHubVecEnv(n, seed=3, **kw)
ref.close()
ref = chub.VecChargingHub(n, seed=3, **kw)      # a fresh handle: the Philox ticks count a handle's launches
tb.reset(); ref.reset()
for t in range(100):
    a = torch.rand((n, 47), device="cuda", generator=g) * 2 - 1
    if t % 7 == 0:
        a[:, :45] = -2.0 ** -25 if t % 2 else float(np.nextafter(np.float32(-2.0 ** -25), np.float32(-1)))
    bits, tail = tb.pack_bits(a)
    hb, _ = ref.pack_actions(a.cpu().numpy())
    assert np.array_equal(bits.cpu().numpy().view(np.uint64), hb), t
    obs, rew, done, info = tb.step_bits(bits, tail)
    ro, rr, rd, _ = ref.step(a.cpu().numpy())
    assert np.array_equal(rew.cpu().numpy(), rr) and np.array_equal(done.cpu().numpy(), rd), t
    if rd.all():
        ro = ref.reset()
    assert np.array_equal(obs.cpu().numpy(), ro), t
tb.close()
try:
    tv.step(torch.zeros((n, 46), device="cuda"))
    raise SystemExit("shape check missing")
except AssertionError:
    pass
print("TORCH_VEC_OK")
'''


def test_torch_device_resident_adapter():
    """TorchHubVecEnv: CUDA tensors in / out through the device-pointer entry points == the host-pointer path"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", TORCH_CHILD], env=dict(os.environ, CHUB_ROOT=root), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "TORCH_VEC_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
