"""Import alias: ``mrag_amd`` is the importable name of the package that lives in
``a-modular-rag-framework_amd/`` (a hyphen cannot appear in an ``import`` statement,
and plug-in strings such as ``"mrag_amd.provider:HipEmbeddingProvider"`` go through
``importlib.import_module`` in the reference's app/di/factory.py:12-16).

The package body is executed once, under this name; submodules resolve through the
real directory via ``__path__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "a-modular-rag-framework_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
