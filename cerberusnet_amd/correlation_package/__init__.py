from .correlation import Correlation, CorrelationFunction, CorrelationTorch

__all__ = ["Correlation", "CorrelationFunction", "CorrelationTorch"]
