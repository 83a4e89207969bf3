"""r3det/ops/fr/feature_refine_module.py:11-127 under its module name."""
from ..feature_refine import FR, FeatureRefineFunction, FeatureRefineModule, feature_refine  # noqa: F401
