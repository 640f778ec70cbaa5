from .UnFlowLoss import flow_warp, mesh_grid, norm_grid, unFlowLoss

__all__ = ["flow_warp", "mesh_grid", "norm_grid", "unFlowLoss"]
