"""cerberusnet_amd -- MI355X-native (gfx950) correlation + flow-warp hot path for
CerberusNet, behind the reference's own ``correlation_package`` /
``flow_warp`` surface.  Importing the package registers ``torch.ops.cerberus.*``.
"""
__version__ = "0.1.0"

from . import ops  # noqa: F401  (registers torch.ops.cerberus.*)
from .correlation_package.correlation import (Correlation, CorrelationFunction,
                                              CorrelationTorch)
from .loss_functions.UnFlowLoss import area_pyramid, area_resize, flow_warp, mesh_grid, norm_grid

__all__ = ["Correlation", "CorrelationFunction", "CorrelationTorch", "flow_warp",
           "mesh_grid", "norm_grid", "area_resize", "area_pyramid", "ops"]
