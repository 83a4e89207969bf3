from .feature_refine_module import FeatureRefineModule

__all__ = ['FeatureRefineModule']
