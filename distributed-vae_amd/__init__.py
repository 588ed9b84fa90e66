"""MI355X-native engine for the cpl-mixVAE (MMIDAS) train step.

Package directory name is ``distributed-vae_amd`` (not a valid identifier); import it as
``distributed_vae_amd`` (the shim module at the repo root registers that alias) or with
``importlib.import_module("distributed-vae_amd")``.

Public surface = the reference's API for the hot path:

    from distributed_vae_amd.nn_model import mixVAE_model, mk_vae, VAEConfig
    from distributed_vae_amd.cpl_mixvae import cpl_mixVAE
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: ~2 us per launch on MI355X, ~45 launches
# per step.  Read by the HIP runtime at initialisation, so it only helps when this package is imported before the
# first CUDA/HIP call of the process.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _native  # noqa: F401,E402
from .nn_model import VAEConfig, mixVAE_model, mk_vae  # noqa: F401,E402

__all__ = ["mixVAE_model", "mk_vae", "VAEConfig", "_native"]
