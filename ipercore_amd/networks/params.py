"""Parameter inventory of the AttLWB-SPADE generator (names and shapes only).

The drop-in boundary requires checkpoint compatibility with ``AttLWB-SPADE_id_G_2020-05-18.pth``: the
``state_dict`` keys produced by the module structure of the reference's
``iPERCore/models/networks/generators/attlwb_spade_resunet.py:567-613`` (``AttentionLWBGenerator.__init__``),
``:255-412`` (Encoder / Decoder / SkipDecoder / ResAutoEncoder) and ``bg_inpaintor.py:24-60``.
The MI355X build does not need the reference's layer objects - kernels consume packed weight panels - so the
module keeps a *parameter tree* with exactly those dotted names.  ``tests/test_oracle_golden.py`` pins the
(name, shape) set against the reference module's own ``state_dict()`` (hash in tests/golden).
"""
from collections import OrderedDict


def _conv(spec, name, cout, cin, k, bias=True):
    spec[name + ".weight"] = (cout, cin, k, k)
    if bias:
        spec[name + ".bias"] = (cout,)


def _convT(spec, name, cin, cout, bias=True):
    spec[name + ".weight"] = (cin, cout, 4, 4)          # ConvTranspose2d stores (Cin, Cout, kh, kw)
    if bias:
        spec[name + ".bias"] = (cout,)


def _attlwb(spec, p, cq, cs, c):
    """SelfAttentionLWB (:194-206): fq/fk/fv 1x1 + SPADE(cond = attended feature, hidden 128)."""
    _conv(spec, p + ".fq", c, cq, 1)
    _conv(spec, p + ".fk", c, cs, 1)
    _conv(spec, p + ".fv", c, cs, 1)
    _conv(spec, p + ".spade.mlp_shared.0", 128, c, 3)
    _conv(spec, p + ".spade.mlp_gamma", cq, 128, 3)
    _conv(spec, p + ".spade.mlp_beta", cq, 128, 3)


def bg_net_param_shapes(spec, cond_nc, filters, n_res):
    """ResNetInpaintor Sequential indices (bg_inpaintor.py:31-57): conv/IN/ReLU triplets count 3 each."""
    i = 0
    _conv(spec, f"bg_net.main.{i}", filters[0], cond_nc, 7)
    i += 3
    for d in range(1, len(filters)):
        _conv(spec, f"bg_net.main.{i}", filters[d], filters[d - 1], 3)
        i += 3
    for _ in range(n_res):
        _conv(spec, f"bg_net.main.{i}.main.0", filters[-1], filters[-1], 3)
        _conv(spec, f"bg_net.main.{i}.main.3", filters[-1], filters[-1], 3)
        i += 1
    for d in range(len(filters) - 1, 0, -1):
        _convT(spec, f"bg_net.main.{i}", filters[d], filters[d - 1], bias=False)
        i += 3
    _conv(spec, f"bg_net.main.{i}", 3, filters[0], 7, bias=False)


def _softgate(spec, p, c_src, c_tsf):
    """SoftGateLWB.gate_conv (lwb_softgate_resunet.py:83-88): Conv3(in = src filters, out = tsf filters) + ReLU + Conv3 + Sigmoid."""
    _conv(spec, p + ".gate_conv.0", c_tsf, c_src, 3)
    _conv(spec, p + ".gate_conv.2", c_tsf, c_tsf, 3)


def generator_param_shapes(num_filters=(64, 128, 256), n_res_block=6, bg_filters=(64, 128, 128, 256),
                           cond_nc=6, bg_cond_nc=4, with_bg=True, lwb="att"):
    """OrderedDict name -> shape for AttLWB-SPADE (``with_bg=False`` gives AttLWB-Front-SPADE, :702-834).
    lwb = "plain": AddLWB / AvgLWB (lwb_resunet.py:315-363, no block parameters); "softgate": SoftGateAdd/AvgLWB
    (lwb_softgate_resunet.py:286-372)."""
    nf = list(num_filters)
    n_down = len(nf)
    spec = OrderedDict()
    if with_bg:
        bg_net_param_shapes(spec, bg_cond_nc, list(bg_filters), n_res_block)
    # SIDNet = ResAutoEncoder(:360-412): encoders (bias), res blocks, decoders (reversed filters), regressors
    for i in range(n_down):
        _conv(spec, f"src_net.encoders.layers.{i}.0", nf[i], cond_nc if i == 0 else nf[i - 1], 3)
    for i in range(n_res_block):
        _conv(spec, f"src_net.res_blocks.{i}.main.0", nf[-1], nf[-1], 3)
        _conv(spec, f"src_net.res_blocks.{i}.main.2", nf[-1], nf[-1], 3)
    rev = list(reversed(nf))
    for i in range(n_down):
        _convT(spec, f"src_net.decoders.layers.{i}.0", nf[-1] if i == 0 else rev[i - 1], rev[i])
    _conv(spec, "src_net.img_reg.0", 3, nf[0], 5, bias=False)
    _conv(spec, "src_net.att_reg.0", 1, nf[0], 5, bias=False)
    # TSFNet encoder: no bias (:588-592)
    for i in range(n_down):
        _conv(spec, f"tsf_net_enc.layers.{i}.0", nf[i], cond_nc if i == 0 else nf[i - 1], 3, bias=False)
    # SkipDecoder (:316-357)
    for i in range(n_down):
        d_in = nf[-1] if i == 0 else rev[i - 1]
        if i != n_down - 1:
            _conv(spec, f"tsf_net_dec.skippers.{i}.0", rev[i], nf[n_down - 2 - i] + rev[i], 3)
    for i in range(n_down):
        d_in = nf[-1] if i == 0 else rev[i - 1]
        _convT(spec, f"tsf_net_dec.upconvs.{i}.0", d_in, rev[i])
    if lwb == "att":
        for i in range(n_down):
            _attlwb(spec, f"enc_attlwbs.{i}", nf[i], nf[i], nf[i])
        for i in range(n_res_block):
            _attlwb(spec, f"res_attlwbs.{i}", nf[-1], nf[-1], nf[-1])
    elif lwb == "softgate":
        for i in range(n_down):
            _softgate(spec, f"enc_attlwbs.{i}", nf[i], nf[i])
        for i in range(n_res_block):
            _softgate(spec, f"res_attlwbs.{i}", nf[-1], nf[-1])
    else:
        assert lwb == "plain", lwb
    for i in range(n_res_block):
        _conv(spec, f"res_blocks.{i}.main.0", nf[-1], nf[-1], 3)
        _conv(spec, f"res_blocks.{i}.main.2", nf[-1], nf[-1], 3)
    _conv(spec, "tsf_img_reg.0", 3, nf[0], 5, bias=False)
    _conv(spec, "tsf_att_reg.0", 1, nf[0], 5, bias=False)
    return spec


def resautoencoder_param_shapes(spec, prefix, in_channel, num_filters, n_res_block):
    """ResAutoEncoder (input_concat_resunet.py:126-179 / attlwb_spade_resunet.py:360-412): Encoder with bias, residual blocks, the plain
    Decoder (reversed filters, no skips), the 5x5 image / mask regressors - under ``prefix`` ("src_net", "tsf_net")."""
    nf = list(num_filters)
    n_down = len(nf)
    for i in range(n_down):
        _conv(spec, f"{prefix}.encoders.layers.{i}.0", nf[i], in_channel if i == 0 else nf[i - 1], 3)
    for i in range(n_res_block):
        _conv(spec, f"{prefix}.res_blocks.{i}.main.0", nf[-1], nf[-1], 3)
        _conv(spec, f"{prefix}.res_blocks.{i}.main.2", nf[-1], nf[-1], 3)
    rev = list(reversed(nf))
    for i in range(n_down):
        _convT(spec, f"{prefix}.decoders.layers.{i}.0", nf[-1] if i == 0 else rev[i - 1], rev[i])
    _conv(spec, f"{prefix}.img_reg.0", 3, nf[0], 5, bias=False)
    _conv(spec, f"{prefix}.att_reg.0", 1, nf[0], 5, bias=False)


def concat_generator_param_shapes(num_filters=(64, 128, 256), n_res_block=6, bg_filters=(64, 128, 128, 256), cond_nc=27, bg_cond_nc=4,
                                  bg_n_res_block=None):
    """InputConcatGenerator (input_concat_resunet.py:182-307; cond_nc = 27) / TextureWarpingGenerator (texture_warping_resunet.py:8-112;
    cond_nc = 6): ``bg_net`` (ResNetInpaintor) + ``tsf_net`` (ResAutoEncoder)."""
    spec = OrderedDict()
    bg_net_param_shapes(spec, bg_cond_nc, list(bg_filters), n_res_block if bg_n_res_block is None else bg_n_res_block)
    resautoencoder_param_shapes(spec, "tsf_net", cond_nc, num_filters, n_res_block)
    return spec
