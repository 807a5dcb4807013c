"""Generator factory of the per-frame path (reference iPERCore/models/networks/__init__.py:3-67)."""
from .params import generator_param_shapes  # noqa: F401


class NetworksFactory(object):
    """``NetworksFactory.get_by_name("AttLWB-SPADE", cfg=..., temporal=...)`` as the reference's runners call it
    (models/imitator.py:158-175).  Only the generators of the hot path are built natively; other names raise."""

    @staticmethod
    def get_by_name(network_name, *args, **kwargs):
        from .generator import AttentionLWBFrontGenerator, AttentionLWBGenerator
        if network_name == "AttLWB-SPADE":
            return AttentionLWBGenerator(*args, **kwargs)
        if network_name == "AttLWB-Front-SPADE":
            return AttentionLWBFrontGenerator(*args, **kwargs)
        raise ValueError(f"Network {network_name} is outside the MI355X hot path (SURVEY.md section 8): "
                         "only AttLWB-SPADE / AttLWB-Front-SPADE are built")
