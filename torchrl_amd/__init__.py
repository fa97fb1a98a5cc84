"""torchrl_amd -- MI355X-native engine for torchrl's collector -> replay_buffer ->
algo.update hot path, behind the reference package's own class API.

Sub-packages mirror the reference layout (torchrl/{env,networks,policies,
replay_buffers,collector,algo,utils}); the arithmetic lives in hand-written
gfx950 HIP kernels reached through the C ABI in ``include/trl_hip.h``
(``torchrl_amd._C``).  There is no CPU or eager-PyTorch implementation of the
hot path: kernels fail loudly when the library or a GPU is missing.

``import torchrl`` (the thin alias package at the repo root) resolves to this
package, so the reference's ``examples/*_vec.py`` import lines work unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"


def _ensure_gym():
    """The reference examples `import gym` and the algos test
    isinstance(space, gym.spaces.Box) (torchrl/algo/rl_algo.py:35).  When no real
    gym is installed a minimal stand-in (spaces + wrapper bases) is registered."""
    try:
        import gym  # noqa: F401
    except Exception:
        shim = importlib.import_module("torchrl_amd._shims.gym")
        sys.modules.setdefault("gym", shim)
        sys.modules.setdefault("gym.spaces", shim.spaces)


_ensure_gym()
