"""guardx_amd -- MI355X-native batched GUARD environment step (gfx950 HIP kernels)
behind the guardX `safe_rl_envs` Engine interface."""
from .engine import Engine, ResamplingError  # noqa: F401
from .env_config import configuration, create_env  # noqa: F401

__version__ = "0.1.0"
