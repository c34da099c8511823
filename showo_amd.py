"""Importable alias: the package directory is `show-o_amd/` (hyphenated, per the repo contract), which
`import` cannot spell.  `import showo_amd` resolves to it, and `showo_amd.<submodule>` resolves to the SAME module object as
`show-o_amd.<submodule>` (one instance of every class: isinstance checks across the two spellings must hold)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL, _ALIAS = "show-o_amd", __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`showo_amd.x.y` -> the module object of `show-o_amd.x.y` (imported on demand), never a second copy"""

    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or not fullname.startswith(_ALIAS + "."):
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
for _k in [k for k in sys.modules if k.startswith(_REAL + ".")]:
    sys.modules[_ALIAS + _k[len(_REAL):]] = sys.modules[_k]
sys.modules[_ALIAS] = _pkg
