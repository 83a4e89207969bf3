"""r3det (MI355X-native): drop-in operator package for the r3det custom-op hot path.

Import as ``r3det`` after putting ``<repo>/r3det-pytorch_amd`` on ``sys.path``; the public
names of ``r3det.ops``, ``r3det.core.bbox.iou_calculators`` and
``r3det.core.post_processing`` are those of the reference package.
"""
__version__ = '0.1.0'
