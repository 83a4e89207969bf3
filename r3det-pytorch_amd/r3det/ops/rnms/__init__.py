from .rnms_wrapper import batched_rnms, rnms

__all__ = ['batched_rnms', 'rnms']
