"""gym_minigrid_amd -- MI355X-native batched MiniGrid (step -> gen_obs -> encode hot path).

Host-side mirror of the reference's env interface over libmgx.so (HIP, gfx950).  See DESIGN.md."""
from ._lib import MgxError, InvalidAction, OutOfBounds, env_config, env_ids, Config  # noqa: F401
from .vec_env import VecMiniGrid, generate_levels, generate_level_stream  # noqa: F401
from .actions import action_stream  # noqa: F401
from .compat import SingleEnv, Actions, make  # noqa: F401

__all__ = ["VecMiniGrid", "generate_levels", "generate_level_stream", "action_stream", "env_config", "env_ids", "Config",
           "MgxError", "InvalidAction", "OutOfBounds", "SingleEnv", "Actions", "make"]
