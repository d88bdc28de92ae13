"""safemotionsrisk_amd - MI355X-native batched hot path of translearn/safeMotionsRisk.

Host mirror of the reference's env surface over a C-ABI library of HIP kernels (gfx950):
``VecSafeMotionsEnv`` (env.py), scene construction from ``env_config`` (scene.py), the ctypes binding
(lib.py), rollout sharding over the GPUs of a node (dist.py), episode metrics and the RLlib-style vector
interface (metrics.py).
"""
from .scene import SPACE_ENV_CONFIG, BALL_ENV_CONFIG, REACH_ENV_CONFIG, build_scene  # noqa: F401
from .env import VecSafeMotionsEnv, SafeMotionsEnvCollisionAvoidance, SafeMotionsEnv  # noqa: F401
from .metrics import EpisodeMetrics, RllibVectorEnv  # noqa: F401
from .policy import Mlp, backup_policy, risk_network  # noqa: F401
