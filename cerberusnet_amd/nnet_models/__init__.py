from .pwcnet_sfd import PWCNetHead
from .pwcnet_modules import FlowEstimatorDense, FlowEstimatorLite, ContextNetwork
from .hrnetv2 import HighResolutionNet, hrnet_config
from .cerberus import CerberusBase, cerberus_flow_config

__all__ = ["PWCNetHead", "FlowEstimatorDense", "FlowEstimatorLite", "ContextNetwork",
           "HighResolutionNet", "hrnet_config", "CerberusBase", "cerberus_flow_config"]
