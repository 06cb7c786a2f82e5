"""balatro_gym_amd -- MI355X-native vectorised Balatro environment (hand-written HIP for gfx950).

Drop-in for ONE hot path of cassiusfive/balatro-gym: the reset()/step() loop of
`balatro_gym/balatro_env_2.py::BalatroEnv`.  `BalatroEnv` keeps the single-env Gymnasium surface; `BalatroVecEnv`
steps tens of thousands of games in lockstep on one GPU; `ShardedBalatroVecEnv` spreads them over a node's GPUs.
"""
from .constants import Action, Phase  # noqa: F401

__all__ = ["Action", "Phase", "BalatroEnv", "BalatroVecEnv", "BalatroSB3VecEnv", "ShardedBalatroVecEnv", "make_balatro_env",
           "shard_range", "classify_batch", "score_hand_batch", "sim_evaluate_batch", "sim_score_batch"]


def __getattr__(name):  # lazy: importing the package must not need torch / the GPU
    if name in ("BalatroVecEnv", "ObsBuffers", "RowBuffers", "classify_batch", "score_hand_batch", "sim_evaluate_batch", "sim_score_batch"):
        from . import vec_env
        return getattr(vec_env, name)
    if name in ("BalatroEnv", "make_balatro_env"):
        from . import env
        return getattr(env, name)
    if name in ("BalatroSB3VecEnv", "FIXED_SPEC", "fix_observation"):
        from . import sb3_adapter
        return getattr(sb3_adapter, name)
    if name in ("ShardedBalatroVecEnv", "shard_range"):
        from . import sharded
        return getattr(sharded, name)
    raise AttributeError(name)
