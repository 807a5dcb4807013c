"""Generator factory of the per-frame path (reference iPERCore/models/networks/__init__.py:3-67)."""
from .params import generator_param_shapes  # noqa: F401


class NetworksFactory(object):
    """``NetworksFactory.get_by_name("AttLWB-SPADE", cfg=..., temporal=...)`` as the reference's runners call it
    (models/imitator.py:158-175).  Only the generators of the hot path are built natively; other names raise."""

    @staticmethod
    def get_by_name(network_name, *args, **kwargs):
        from . import generator as g
        table = {"AttLWB-SPADE": g.AttentionLWBGenerator, "AttLWB-Front-SPADE": g.AttentionLWBFrontGenerator,
                 "AddLWB": g.AddLWBGenerator, "AvgLWB": g.AvgLWBGenerator,
                 "SoftGateAddLWB": g.SoftGateAddLWBGenerator, "SoftGateAvgLWB": g.SoftGateAvgLWBGenerator}
        if network_name in table:
            return table[network_name](*args, **kwargs)
        if network_name in ("patch_global", "patch_global_local", "patch_global_body_head"):
            from ..trainers import create_discriminator
            return create_discriminator(network_name, *args, **kwargs)
        raise ValueError(f"Network {network_name} is outside the MI355X hot path (SURVEY.md section 8): built are "
                         f"{sorted(table)}; AttLWB-AdaIN, InputConcat, TextureWarping and the discriminators' factory names are not")
