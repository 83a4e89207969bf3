from .delta_xywha_rbbox_coder import DeltaXYWHAOBBoxCoder, bbox2delta_v1, delta2bbox_v1

__all__ = ['DeltaXYWHAOBBoxCoder', 'bbox2delta_v1', 'delta2bbox_v1']
