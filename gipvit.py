"""Importable alias of the package directory ``gipmed-project-self-supervised-vit_amd/``
(hyphens are not legal in a module name).  ``import gipvit.ops`` resolves to
``gipmed-project-self-supervised-vit_amd/ops.py``."""
import os as _os

_PKG = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gipmed-project-self-supervised-vit_amd")
__path__ = [_PKG]
with open(_os.path.join(_PKG, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG, "__init__.py"), "exec"))
