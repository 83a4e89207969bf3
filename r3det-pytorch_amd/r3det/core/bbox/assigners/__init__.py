from .max_iou_assigner import AssignResult, MaxIoUAssigner

__all__ = ['AssignResult', 'MaxIoUAssigner']
