from .pwcnet_sfd import PWCNetHead
from .pwcnet_modules import FlowEstimatorDense, FlowEstimatorLite, ContextNetwork

__all__ = ["PWCNetHead", "FlowEstimatorDense", "FlowEstimatorLite", "ContextNetwork"]
