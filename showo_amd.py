"""Importable alias: the package directory is `show-o_amd/` (hyphenated, per the repo contract), which
`import` cannot spell.  `import showo_amd` resolves to it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("show-o_amd")
sys.modules[__name__] = _pkg
