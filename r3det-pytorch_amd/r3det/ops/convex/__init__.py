from .convex_wrapper import convex_sort  # noqa: F401, F403

__all__ = ['convex_sort']
