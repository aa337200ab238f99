"""`rectified_spaattn` (the import name the reference's scripts use) is an alias of rectified_spaattn_amd."""


def test_alias_package_exposes_the_same_module_objects():
    import rectified_spaattn
    import rectified_spaattn.attn_processor as ap
    import rectified_spaattn_amd.attn_processor as ap2
    from rectified_spaattn.attn import fullattn, get_cu_seqlens  # noqa: F401
    from rectified_spaattn.gapr_mask import estimate_pr_gain  # noqa: F401
    from rectified_spaattn.rectified_cogvideo_attn import RectifiedCogVideoXVideoSpaAttnProcessor2_0  # noqa: F401
    from rectified_spaattn.rectified_flux_attn import RectifiedFluxSpaAttnProcessor2_0  # noqa: F401
    from rectified_spaattn.rectified_hunyuan_attn import RectifiedHunyuanVideoSpaAttnProcessor2_0 as A
    from rectified_spaattn.rectified_wan21_attn import RectifiedWanI2VSpaAttnProcessor2_0  # noqa: F401
    from rectified_spaattn.rectified_wan22_attn import (RectifiedWanI2VSpaAttnProcessor2_0 as W22I,  # noqa: F401
                                                        RectifiedWanT2VSpaAttnProcessor2_0, RectifiedWanTI2VSpaAttnProcessor2_0)
    from rectified_spaattn_amd.rectified_hunyuan_attn import RectifiedHunyuanVideoSpaAttnProcessor2_0 as B
    assert A is B and ap is ap2 and rectified_spaattn.__version__
