"""instageo_amd -- MI355X-native hot path of InstaGeo's Prithvi segmentation model (host side).

Mirrors the reference's ``instageo.model`` surface for that path (``PrithviSeg``, the segmentation task
module, ``chip_inference``, the ``run.py`` key=value CLI); all arithmetic runs in ``libinstageo_hip.so``.
"""
__version__ = "0.1.0"
