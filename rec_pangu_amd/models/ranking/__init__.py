from .deepfm import DeepFM
from .xdeepfm import xDeepFM
from .dcn import DCN
from .autoint import AutoInt
from .fm import FM

__all__ = ["DeepFM", "xDeepFM", "DCN", "AutoInt", "FM"]
