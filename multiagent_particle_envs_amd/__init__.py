"""MI355X-native batched multi-agent particle environment: the `MultiAgentEnv.step` hot path of
openai/multiagent-particle-envs as hand-written HIP kernels (gfx950) behind a C ABI, driven from
Python with PyTorch-ROCm tensors holding the SoA world state.  See DESIGN.md."""
from .make_env import make_env  # noqa: F401
from .environment import MultiAgentEnv, BatchMultiAgentEnv, GraphedStep  # noqa: F401
from .scenario import BaseScenario  # noqa: F401
from . import core, scenarios  # noqa: F401

__all__ = ["make_env", "MultiAgentEnv", "BatchMultiAgentEnv", "GraphedStep", "BaseScenario", "core", "scenarios"]
