"""cerberusnet_amd -- MI355X-native correlation + flow-warp hot path for CerberusNet."""
__version__ = "0.1.0"
