"""Alias so that the reference's config string works unchanged except for the package prefix:
    network.callable: "babe_amd.networks.cqtdiff+.Unet_CQT_oct_with_attention"
(importlib.import_module handles the '+', exactly as dnnlib.call_func_by_name does for the reference,
utils/dnnlib/util.py:250)."""
from .cqtdiff_plus import *  # noqa: F401,F403
from .cqtdiff_plus import Unet_CQT_oct_with_attention  # noqa: F401
