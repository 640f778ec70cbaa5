from .UnFlowLoss import flow_warp, mesh_grid, norm_grid

__all__ = ["flow_warp", "mesh_grid", "norm_grid"]
