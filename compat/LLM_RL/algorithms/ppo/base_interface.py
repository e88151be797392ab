"""`LLM_RL.algorithms.ppo.base_interface` names (reference: ppo/base_interface.py:38-293, 295-343)."""
from lmrl_gym_amd.algorithms.ppo import (AdaptiveKLController, FixedKLController, get_action_state_next_state_idxs,  # noqa: F401
                                         get_advantages_and_returns, ppo_loss_fn, whiten)
from lmrl_gym_amd.algorithms.ppo_inference import CombinedTokenTrajectoryChain, PPOForwardOutput  # noqa: F401
