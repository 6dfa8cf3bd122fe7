from .deepfm import DeepFM
from .xdeepfm import xDeepFM
from .dcn import DCN
from .autoint import AutoInt
from .fm import FM
from .wdl import WDL
from .nfm import NFM
from .lr import LR

__all__ = ["DeepFM", "xDeepFM", "DCN", "AutoInt", "FM", "WDL", "NFM", "LR"]
