"""Drop-in import name of the reference package: `rectified_spaattn.<module>` IS `rectified_spaattn_amd.<module>`.

The reference's scripts import `rectified_spaattn.rectified_hunyuan_attn`, `rectified_spaattn.attn_processor`, ...
(scripts/main_hunyuan.py:7, main_wan21t2v.py:8-9, main_wan22ti2v.py:8-9, main_cogvideox.py, main_upflux.py).  With this
repository on sys.path ahead of the reference checkout those imports resolve to the MI355X implementation with no edit
to the scripts (option A).  Every sub-module below is the very same module object as its
rectified_spaattn_amd counterpart (registered in sys.modules), not a copy.
"""
import importlib
import sys

_MODULES = ("attn", "attn_processor", "gapr_mask", "rectified_hunyuan_attn", "rectified_flux_attn",
            "rectified_wan21_attn", "rectified_wan22_attn", "rectified_cogvideo_attn", "teacache")

for _name in _MODULES:
    _mod = importlib.import_module("rectified_spaattn_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    setattr(sys.modules[__name__], _name, _mod)

from rectified_spaattn_amd import __version__, set_dense_fp8, set_qkv_fp8  # noqa: E402,F401
