"""Registry shim.  With mmdet installed the classes are registered into mmdet's own
registries (as the reference does, rotate_iou2d_calculator.py:2,6); otherwise a minimal local
registry offers the same ``register_module`` / ``build(cfg)`` surface so that config dicts
such as ``dict(type='RBboxOverlaps2D_v1')`` resolve unchanged."""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, **default_args):
        args = dict(cfg)
        args.update(default_args)
        typ = args.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        return cls(**args)

    def __contains__(self, key):
        return key in self._modules


try:  # pragma: no cover - mmdet is not installed in the build image
    from mmdet.core.bbox.iou_calculators.builder import IOU_CALCULATORS
except Exception:  # noqa: BLE001
    IOU_CALCULATORS = Registry('IoU calculator')


# The model-side registries of the reference stack (mmdet.models.builder / mmdet.core builders), local: the names
# of configs/r3det/*.py and configs/rretinanet/*.py resolve to this package's classes (models/__init__.py registers
# them), so ``build_detector(cfg.model)`` builds the same module tree from the reference's ``model = dict(...)``.
DETECTORS = Registry('detector')
BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
LOSSES = Registry('loss')
PRIOR_GENERATORS = Registry('prior generator')
BBOX_CODERS = Registry('bbox coder')
BBOX_ASSIGNERS = Registry('bbox assigner')


def _plain(cfg):
    """mmcv.Config / ConfigDict / dict -> a plain (shallow-copied) dict."""
    return {k: cfg[k] for k in cfg.keys()}


def build_from(registry, cfg, **default_args):
    return registry.build(_plain(cfg), **default_args)


def build_iou_calculator(cfg, default_args=None):
    if hasattr(IOU_CALCULATORS, 'build') and isinstance(IOU_CALCULATORS, Registry):
        return IOU_CALCULATORS.build(cfg, **(default_args or {}))
    from mmcv.utils import build_from_cfg  # pragma: no cover
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)  # pragma: no cover
