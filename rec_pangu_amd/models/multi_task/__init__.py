from .mmoe import MMOE

__all__ = ["MMOE"]
