"""Weight containers for the mel-inversion engine.

Raw variables follow the reference's weight-normalised convolution
(reference MBExWN_NVoc/vocoder/model/tf2_components/layers/conv_layers.py:79-103):
``<layer>.v`` (ks, cin, cout), ``<layer>.g`` (cout,), ``<layer>.bias`` (cout,) and the
PReLU slopes ``<act>.alpha`` (channels,).  ``fold_weights`` turns them into the plain
``<layer>.w`` / ``<layer>.b`` tensors the HIP engine consumes.

The pretrained checkpoints are not in the reference tree; ``synthetic_weights``
generates seeded weights of the exact architecture.
A checkpoint saved with ``save_weights`` is a plain ``.npz`` of the raw variables.
"""
import re

import numpy as np

from .config import ModelDims
from .subnet import build_subnet


def block_prefix(index):
    """Tensor-name prefix of WaveNet block ``index``: "wn." for the first (and usually only) one, "wn1." ..."""
    return "wn." if index == 0 else f"wn{index}."


def is_wavenet_layer(name):
    """Layers built with pp_mod_subnet's use_weight_norm / use_equalized_lr switches: everything inside a WaveNetAE
    (reference custom_AE_layers.py:177-260), of any block; not the up-sampling convolutions behind the blocks."""
    import re
    return re.match(r"wn\d*\.", name) is not None


def layer_table(config):
    """[(name, ks, cin, cout)] of every weight-normed conv and [(name, channels)] of every PReLU."""
    dims = ModelDims(config)
    mb = config["mbexwn_config"]
    convs, prelus = [], []
    use_prelu = mb.get("use_prelu", True)
    f0_ops, _, _ = build_subnet(mb["pp_subnet"], "PulsPar", dims.mel_channels, 1, 1,
                                mb.get("pp_activation", "soft_sigmoid"), target_ups=dims.pulse_per_frame,
                                pad_to_valid=mb.get("pp_subnet_use_valid_padding", False), use_prelu=use_prelu,
                                alpha=dims.alpha)
    vtf_ops = []
    if not dims.ps_off:                                       # ps_off builds no VTF-net (reference custom_pulsed_generator.py:413)
        vtf_ops, _, _ = build_subnet(mb["ps_subnet"], "PS", dims.mel_channels, dims.n_ceps, 1, None,
                                     pad_to_valid=mb.get("ps_subnet_use_valid_padding", False), use_prelu=use_prelu,
                                     alpha=dims.alpha)
    for op in f0_ops + vtf_ops:
        if op["kind"] == "conv":
            convs.append((op["name"], op["ks"], op["cin"], op["cout"]))
        elif op["kind"] == "prelu":
            prelus.append((op["name"], op["channels"]))
    L = dims.wn_layers
    G = dims.wn_groups
    rows_per_frame = dims.wn_in_rows_per_frame
    for bb, (C, ups) in enumerate(zip(dims.wn_block_channels, dims.wn_block_ups)):
        # block b: "wn." (b = 0) / "wn<b>." ; its input: the folded excitation (b = 0) or the previous block's n_out channels
        pre = block_prefix(bb)
        convs.append((pre + "start", 1, dims.wn_in_channels if bb == 0 else dims.wn_out_channels, C))
        if not dims.wn_disable_conditioning:
            cond_in = dims.mel_channels
            for ii, chans in enumerate(dims.wn_pre_cond_channels):      # reference custom_AE_layers.py:192-201
                convs.append((f"{pre}precond_{ii}", dims.cond_kernel_size, cond_in, chans))
                cond_in = chans
            convs.append((pre + "cond", dims.cond_kernel_size, cond_in, 2 * C * (rows_per_frame // dims.cond_lin_upsampling)))
        # n_ch_groups independent channel groups, each with its own layers "conv1D_<l>", "conv1D_<l>g1", ...
        # (reference custom_AE_layers.py:235-260)
        Cg = C // G
        for ll in range(L):
            for gg in range(G):
                sfx = f"g{gg}" if gg else ""
                convs.append((f"{pre}conv1D_{ll}{sfx}", dims.wn_kernel_size, Cg, 2 * Cg))
                convs.append((f"{pre}res_skip_{ll}{sfx}", 1, Cg, 2 * Cg if ll < L - 1 else Cg))
        convs.append((pre + "end", 1, C, dims.wn_out_channels))
        if ups > 1:      # Conv1DUpDownSample behind the block: k = 3, n_out -> n_out * ups, depth -> time (conv_layers.py:177-261)
            convs.append((f"up{bb}", 3, dims.wn_out_channels, dims.wn_out_channels * ups))
        rows_per_frame *= ups
    convs.append(("post", 1, dims.wn_out_channels, dims.subbands))
    return convs, prelus


# per-layer gain of the synthetic weights (weight-norm g relative to ||v||=1 columns); chosen so
# that activations stay O(1), F0 moves inside [fmin,fmax] and the envelope filter is well inside
# its +-range.  Arbitrary but fixed: part of the definition of the synthetic benchmark model.
_SYNTH_GAIN = {"PS_Layer_final": 0.02, "wn.end": 0.04, "post": 1.0, "wn.cond": 0.7, "PulsPar_Layer_final": 2.0}


def synthetic_weights(config, seed=1234, bias_std=0.0, alpha_jitter=0.0):
    """Seeded raw variables: v ~ N(0, 0.02^2), g = gain, bias ~ N(0, bias_std^2), alpha = 0.2 (+jitter)."""
    rng = np.random.default_rng(seed)
    convs, prelus = layer_table(config)
    alpha0 = float(config["mbexwn_config"].get("alpha", 0.2))
    wn_cfg = config["mbexwn_config"]["pp_mod_subnet"]
    eq_norm = bool(wn_cfg.get("use_equalized_lr", False)) and bool(wn_cfg.get("use_weight_norm", False))
    raw = {}
    for name, ks, cin, cout in convs:
        raw[name + ".v"] = rng.normal(0.0, 0.02, size=(ks, cin, cout)).astype(np.float32)
        gain = _SYNTH_GAIN.get(re.sub(r"^wn\d+\.", "wn.", name), 1.0)     # the blocks behind the first one like the first
        if eq_norm and is_wavenet_layer(name):
            # use_equalized_lr normalises the kernel to unit variance instead of unit norm (W = g v / sqrt(mean v^2)): the
            # gain takes the 1 / sqrt(fan-in) that keeps the activations at the scale of the weight-normed model
            gain = gain / np.sqrt(ks * cin)
        raw[name + ".g"] = np.full((cout,), gain, dtype=np.float32)
        raw[name + ".bias"] = (rng.normal(0.0, 1.0, size=(cout,)) * bias_std).astype(np.float32)
    for name, channels in prelus:
        raw[name + ".alpha"] = (alpha0 + alpha_jitter * rng.uniform(-1, 1, size=(channels,))).astype(np.float32)
    return raw


def fold_weight_norm(v, g):
    """float32 fold W = g * v * rsqrt(max(sum_{k,ci} v^2, 1e-12)), the arithmetic of
    tf.nn.l2_normalize at reference conv_layers.py:153."""
    v = np.asarray(v, dtype=np.float32)
    sq = np.sum(np.square(v), axis=(0, 1), keepdims=True, dtype=np.float32)
    inv = (np.float32(1) / np.sqrt(np.maximum(sq, np.float32(1e-12)))).astype(np.float32)
    return (np.asarray(g, dtype=np.float32) * (v * inv)).astype(np.float32)


def fold_weight_norm_f64(v, g):
    """The same fold W = g v / sqrt(max(sum_{k,ci} v^2, 1e-12)) (reference conv_layers.py:149-153) evaluated in float64 on the
    float32 variables: the weights of the F0-net under mbx_config.f0_accumulate = MBX_F0_ACC_F64 ("<layer>.w64",
    csrc/conv_mfma.hip::conv1d_f64_tile) -- the contour feeds the phase integrator, so its net runs on the exact fold."""
    v = np.asarray(v, dtype=np.float64)
    sq = np.sum(v * v, axis=(0, 1), keepdims=True)
    return np.asarray(g, dtype=np.float64) * (v / np.sqrt(np.maximum(sq, 1e-12)))


# layers the reference may build without weight normalisation: only the WaveNet's own (pp_mod_subnet.use_weight_norm,
# reference custom_AE_layers.py:124,177-260); the F0 / VTF sub-nets and the post-net are always weight-normed
# (reference custom_pulsed_generator.py:84-136, 491)


def fold_weights(raw, wavenet_weight_norm=None, wavenet_equalized_lr=False):
    """raw variables -> {'<layer>.w': (ks,cin,cout) f32, '<layer>.b': (cout,) f32, '<act>.alpha': ...}.

    A layer without a gain ``<layer>.g`` is taken as built with use_weight_norm=False (the kernel is the weight,
    reference conv_layers.py:157-165) only where the reference can build it that way: the ``wn.*`` layers, and only when
    ``wavenet_weight_norm`` is False or unknown (None: raw dicts of tests / converted checkpoints).  A missing gain on any
    other layer -- or on a WaveNet layer of a model configured with weight normalisation -- raises KeyError instead of
    silently using the un-normalised direction as the weight.

    ``wavenet_equalized_lr`` (pp_mod_subnet.use_equalized_lr: every convolution of the WaveNet, reference
    custom_AE_layers.py:177-260 -> conv_layers.py:133-153): with weight norm W = g v / sqrt(mean_{k,ci} v^2); without it
    the layer multiplies its whole output by g, i.e. W = g K and b = g bias."""
    out = {}
    for key, val in raw.items():
        if key.endswith(".v"):
            name = key[:-2]
            bias = np.asarray(raw[name + ".bias"], dtype=np.float32)
            eq = wavenet_equalized_lr and is_wavenet_layer(name)
            if eq:
                if name + ".g" not in raw:
                    raise KeyError(f"{name}.g is missing: layers built with use_equalized_lr always hold a gain")
                v = np.asarray(val, dtype=np.float32)
                g = np.asarray(raw[name + ".g"], dtype=np.float32)
                if wavenet_weight_norm:      # conv_layers.py:151: g * v / sqrt(reduce_mean(square(v), axes [0, 1]))
                    ms = np.mean(np.square(v), axis=(0, 1), keepdims=True, dtype=np.float32)
                    out[name + ".w"] = ((g * v) / np.sqrt(ms)).astype(np.float32)
                else:                        # conv_layers.py:135-136: act = g * conv(x) (bias included)
                    out[name + ".w"] = (g * v).astype(np.float32)
                    bias = (g * bias).astype(np.float32)
            elif name + ".g" in raw:
                out[name + ".w"] = fold_weight_norm(val, raw[name + ".g"])
            else:
                plain_ok = is_wavenet_layer(name) and wavenet_weight_norm is not True
                if not plain_ok:
                    raise KeyError(f"{name}.g is missing: the reference builds this layer with weight normalisation, "
                                   f"so its checkpoint holds a gain (refusing to use {name}.v as the weight)")
                out[name + ".w"] = np.asarray(val, dtype=np.float32)
            out[name + ".b"] = bias
        elif key.endswith(".alpha"):
            out[key] = np.asarray(val, dtype=np.float32)
    return out


def merge_channel_groups(folded, dims):
    """n_ch_groups > 1 (reference custom_AE_layers.py:303-340): the WaveNet is G independent stacks of C/G channels
    between a shared start and end convolution.  The HIP kernels run ONE stack of C channels, so the per-group layers
    become block-diagonal dense layers (a group's channels only meet that group's weights; the zero blocks cost matrix
    work but no accuracy), with the channel order the dense layout implies:

      conv1D_l    (ks, C, 2C): group g rows [g Cg, (g+1) Cg) -> tanh columns [g Cg, ..) and sigmoid columns C + [g Cg, ..)
      res_skip_l  (1, C, 2C):  res columns [g Cg, ..), skip columns C + [g Cg, ..)   (last layer: (1, C, C), skip only)
      cond        the reference splits its 2C channels into G chunks [tanh_g | sigmoid_g] (:289): its output columns are
                  permuted to [tanh_0 .. tanh_G-1 | sigmoid_0 .. sigmoid_G-1] inside every sub-pixel phase
    Returns a new dict; per-group entries ("...g1.w") are replaced by the dense ones under the group-0 names."""
    G = dims.wn_groups
    if G == 1:
        return folded
    C, L, ks = dims.wn_channels, dims.wn_layers, dims.wn_kernel_size
    Cg = C // G
    import re
    out = {kk: vv for kk, vv in folded.items() if not re.fullmatch(r"wn\.(conv1D|res_skip)_\d+g\d+\.[wb]", kk)}
    for ll in range(L):
        last = ll == L - 1
        wc = np.zeros((ks, C, 2 * C), dtype=np.float32)
        bc = np.zeros((2 * C,), dtype=np.float32)
        wr = np.zeros((1, C, C if last else 2 * C), dtype=np.float32)
        br = np.zeros((C if last else 2 * C,), dtype=np.float32)
        for gg in range(G):
            sfx = f"g{gg}" if gg else ""
            rows = slice(gg * Cg, (gg + 1) * Cg)
            w, b = folded[f"wn.conv1D_{ll}{sfx}.w"], folded[f"wn.conv1D_{ll}{sfx}.b"]
            wc[:, rows, gg * Cg:(gg + 1) * Cg] = w[:, :, :Cg]
            wc[:, rows, C + gg * Cg:C + (gg + 1) * Cg] = w[:, :, Cg:]
            bc[gg * Cg:(gg + 1) * Cg] = b[:Cg]
            bc[C + gg * Cg:C + (gg + 1) * Cg] = b[Cg:]
            w, b = folded[f"wn.res_skip_{ll}{sfx}.w"], folded[f"wn.res_skip_{ll}{sfx}.b"]
            if last:
                wr[:, rows, gg * Cg:(gg + 1) * Cg] = w
                br[gg * Cg:(gg + 1) * Cg] = b
            else:
                wr[:, rows, gg * Cg:(gg + 1) * Cg] = w[:, :, :Cg]
                wr[:, rows, C + gg * Cg:C + (gg + 1) * Cg] = w[:, :, Cg:]
                br[gg * Cg:(gg + 1) * Cg] = b[:Cg]
                br[C + gg * Cg:C + (gg + 1) * Cg] = b[Cg:]
        out[f"wn.conv1D_{ll}.w"], out[f"wn.conv1D_{ll}.b"] = wc, bc
        out[f"wn.res_skip_{ll}.w"], out[f"wn.res_skip_{ll}.b"] = wr, br
    # conditioning: new column (u, [tanh | sigmoid], g, i) <- old column (u, g, [tanh | sigmoid], i)
    if "wn.cond.w" in folded:                                 # absent with disable_conditioning
        up = dims.cond_conv_upsampling
        perm = np.arange(up * 2 * C).reshape(up, G, 2, Cg).transpose(0, 2, 1, 3).reshape(-1)
        out["wn.cond.w"] = np.ascontiguousarray(folded["wn.cond.w"][:, :, perm])
        out["wn.cond.b"] = np.ascontiguousarray(folded["wn.cond.b"][perm])
    return out


def save_weights(path, raw):
    np.savez(path, **raw)


def load_weights(path):
    with np.load(path) as data:
        return {kk: data[kk] for kk in data.files}
