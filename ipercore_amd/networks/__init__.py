"""Generator factory of the per-frame path (reference iPERCore/models/networks/__init__.py:3-67)."""
from .params import concat_generator_param_shapes, generator_param_shapes  # noqa: F401


class NetworksFactory(object):
    """``NetworksFactory.get_by_name("AttLWB-SPADE", cfg=..., temporal=...)`` as the reference's runners call it
    (models/imitator.py:158-175).  Only the generators of the hot path are built natively; other names raise."""

    @staticmethod
    def get_by_name(network_name, *args, **kwargs):
        from . import generator as g
        table = {"AttLWB-SPADE": g.AttentionLWBGenerator, "AttLWB-Front-SPADE": g.AttentionLWBFrontGenerator,
                 "AddLWB": g.AddLWBGenerator, "AvgLWB": g.AvgLWBGenerator,
                 "SoftGateAddLWB": g.SoftGateAddLWBGenerator, "SoftGateAvgLWB": g.SoftGateAvgLWBGenerator,
                 "InputConcat": g.InputConcatGenerator, "TextureWarping": g.TextureWarpingGenerator}
        if network_name in table:
            return table[network_name](*args, **kwargs)
        if network_name in ("patch_global", "patch_global_local", "patch_global_body_head"):
            from ..trainers import create_discriminator
            return create_discriminator(network_name, *args, **kwargs)
        if network_name == "multi_scale":
            from ..trainers import MultiScaleDiscriminator
            return MultiScaleDiscriminator(*args, **kwargs)
        raise ValueError(f"Network {network_name} not recognized: built are {sorted(table)} and the discriminators patch_global, "
                         "patch_global_local, patch_global_body_head, multi_scale; AttLWB-AdaIN is not (its reference constructor does not "
                         "match the factory's call - it cannot be built through the reference's factory either)")
