"""vppstereo_amd -- MI355X-native virtual pattern projection + rSGM hot path.

Drop-in modules (same names / signatures as the reference's):
    vppstereo_amd.vpp_standalone.vpp            <- vpp_standalone.py:396
    vppstereo_amd.vpp_core_opt.*                <- vpp_core/vpp_core_opt.pyx
    vppstereo_amd.pyrSGM.*                      <- pyrSGM natives (rsgm.py:6)
    vppstereo_amd.rsgm.compute_rsgm             <- models/rsgm/rsgm.py:250
    vppstereo_amd.filter.occlusion_heuristic    <- filter.py:246
    vppstereo_amd.losses.{sample_hints,guided_metrics} <- losses.py
Batched, device-resident front-end (torch tensors as plumbing): vppstereo_amd.engine.
All compute runs in libvppx.so (HIP, gfx950); there is no CPU fallback.
"""
from ._lib import Context, VppxError, default_context, load  # noqa: F401
from .vpp_standalone import vpp  # noqa: F401
from .rsgm import compute_rsgm  # noqa: F401
from .filter import occlusion_heuristic  # noqa: F401

__all__ = ["Context", "VppxError", "default_context", "load", "vpp", "compute_rsgm", "occlusion_heuristic"]
