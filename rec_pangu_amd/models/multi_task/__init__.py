from .mmoe import MMOE
from .omoe import OMOE
from .mlmmoe import MLMMOE
from .sharebottom import ShareBottom

__all__ = ["MMOE", "OMOE", "MLMMOE", "ShareBottom"]
