"""Import alias: ``import distributed_vae_amd`` -> the package in ``distributed-vae_amd/``.

The package directory carries the project's hyphenated name; this one-file shim loads it and
registers it (and its submodules) under an importable name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "distributed-vae_amd")
_spec = importlib.util.spec_from_file_location("distributed_vae_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["distributed_vae_amd"] = _mod
_spec.loader.exec_module(_mod)
