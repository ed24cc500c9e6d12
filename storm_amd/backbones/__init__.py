from .shared import BackboneRegistry
from .ncsnpp import NCSNpp, NCSNppLarge, NCSNpp12M, NCSNpp6M

__all__ = ["BackboneRegistry", "NCSNpp", "NCSNppLarge", "NCSNpp12M", "NCSNpp6M"]
