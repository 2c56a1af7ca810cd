"""open_spiel_amd — MI355X-native batched game-step and search engine.

A from-scratch HIP (gfx950) implementation of OpenSpiel's data-parallel hot
path: batched LegalActions / ApplyAction / IsTerminal / Returns /
ObservationTensor for tic_tac_toe, connect_four, hex, kuhn_poker and
leduc_poker, random-rollout evaluation and MCTS over batches of roots, and
tabular CFR / external-sampling MCCFR.  The C-ABI is include/osg_abi.h.
"""
from ._abi import OsgError, describe, lib  # noqa: F401


def _load_torch_runtime_first():
    """PyTorch-ROCm ships its own HIP/HSA runtime under torch/lib.  Two HSA runtimes in one
    process do not work (the second one finds no device), so whenever torch is installed its
    runtime is loaded BEFORE libosg_hip.so / the pybind module, which then bind to the same
    libamdhip64.so.7 by SONAME."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def __getattr__(name):
    if name == "pyspiel_hip":
        import importlib
        _load_torch_runtime_first()
        return importlib.import_module(".pyspiel_hip", __name__)
    # torch-backed classes are imported lazily so that `import open_spiel_amd`
    # (and the ABI symbol check) works without touching torch.
    if name in ("Context", "Game", "StateBatch", "TabularSolver"):
        from . import engine
        return getattr(engine, name)
    if name in ("BatchedEnvironment", "TimeStep", "StepType", "ObservationType"):
        from . import vector_env
        return getattr(vector_env, name)
    raise AttributeError(name)
