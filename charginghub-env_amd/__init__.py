"""chub -- MI355X-native vectorised charging-hub environment.

Host side of the drop-in for the reference's ``reset()/step()`` path
(``evcssp_env_cpp.envs.EvcsspManagerEnv_v6``, evcssp_manager.py:19): a thin ctypes + numpy layer over
``libchub.so`` (HIP kernels for gfx950 behind the C ABI of ``include/chub.h``).  No PyTorch here: the multi-GPU host
(``multi_gpu.py``) drives libchub's own RCCL leg; ``sharded.py`` / ``TorchHubVecEnv`` are adapters for trainers that live in torch.
"""
from ._lib import ChubError, lib_path, load_library  # noqa: F401
from .vec_env import VecChargingHub, make_config  # noqa: F401
from .env import Box, EvcsspManagerEnv_v6  # noqa: F401
from .wrappers import HubVecEnv, HubVectorEnv, StaggeredHub, TimeLimit, TorchHubVecEnv, make, register  # noqa: F401
from .data_io import write_data_dir  # noqa: F401

__all__ = ["VecChargingHub", "EvcsspManagerEnv_v6", "Box", "make_config", "ChubError", "load_library", "lib_path",
           "HubVecEnv", "HubVectorEnv", "StaggeredHub", "TimeLimit", "TorchHubVecEnv", "make", "register", "write_data_dir"]

try:  # same gym id as the reference (evcssp_env_cpp/__init__.py:3-8) when gym / gymnasium is importable
    register()
except Exception:  # id already taken
    pass
