"""Stand-in for the `gym` package (gym is not installed in this image, and there
is no network).  TEST TOOL ONLY: it exists so that oracle/gen_golden.py can import
the read-only reference at /root/reference in THIS container and record golden
vectors.  It is our own code, carries nothing of the reference, and is never
imported by the product path.

Surface = exactly what gym_minigrid touches: gym.Env, gym.core.{Wrapper,
ObservationWrapper,GoalEnv}, gym.spaces.{Box,Discrete,Dict}, gym.error,
gym.utils.seeding.np_random (legacy, RandomState-based), registration.
"""
from . import error, spaces, utils, core          # noqa: F401
from .core import Env, Wrapper, ObservationWrapper, GoalEnv  # noqa: F401
from .envs.registration import register, make     # noqa: F401
from . import envs                                # noqa: F401
