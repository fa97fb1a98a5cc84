"""`torchrl` import alias for `torchrl_amd`.

The reference's scripts say `from torchrl.algo import PPO`,
`from torchrl.collector.on_policy import VecOnPolicyCollector`, ... -- this thin
package makes every `torchrl[.x.y]` import resolve to the SAME module object as
`torchrl_amd[.x.y]` (one class identity, no duplicate module state)."""
import importlib
import importlib.abc
import importlib.util
import sys

import torchrl_amd

_PREFIX = __name__ + "."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return importlib.import_module(self.real)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = "torchrl_amd." + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = torchrl_amd
