from .ranchor_generator import PseudoAnchorGenerator, RAnchorGenerator
from .rutils import ranchor_inside_flags

__all__ = ['RAnchorGenerator', 'ranchor_inside_flags', 'PseudoAnchorGenerator']
