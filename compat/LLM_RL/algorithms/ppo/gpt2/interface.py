"""`LLM_RL.algorithms.ppo.gpt2.interface` (reference: ppo/gpt2/interface.py): trainer, inference and policy on the HIP engine.
Constructors differ from the flax versions (they take engine objects instead of TrainStates); call signatures of
`step` / `forward` / `act` / `get_ppo_data_from_*` are the reference's."""
from lmrl_gym_amd.algorithms.ppo import GPT2PPOTrain  # noqa: F401
from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference  # noqa: F401
from lmrl_gym_amd.policies import GPT2PPOPolicy  # noqa: F401
