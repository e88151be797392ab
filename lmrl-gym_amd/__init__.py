"""lmrl_gym_amd — MI355X-native rollout-and-train engine behind LMRL-Gym's LLM_RL API surface.

Layout
  environment.py   the LLM_RL.environment protocol (Text, TextEnv, interact_environment, Token* ...)
  envs/            device-backed Wordle / Maze environments (host text <-> packed device state)
  algorithms/      PPO / ILQL / MC / BC data shaping + losses on the HIP kernels
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/lmrl_amd.h)
  _lib.py          ctypes binding of liblmrl_amd.so (fails loudly when the library is missing)
"""
__version__ = "0.1.0"
