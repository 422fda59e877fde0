"""TEST INFRASTRUCTURE (fixture generation only).  Inert placeholder for the five `gym` names the
reference host imports (evcssp_manager.py:3,6,7; evcssp_env_cpp/__init__.py:1).  gym 0.18.3 is not
installed in the build image; none of these names takes part in any arithmetic of step()/reset()."""
from . import error, spaces, utils  # noqa: F401


class Env(object):
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None
