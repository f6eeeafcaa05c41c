"""Sub-net grammar of the F0-net (``pp_subnet``) and the VTF-net (``ps_subnet``).

Restates the layer list construction of reference
MBExWN_NVoc/vocoder/model/custom_pulsed_generator.py:38-148 (generate_subnet_from_specs)
as a flat list of plain-dict ops that the HIP engine and the oracle both execute.

Op kinds
  conv : {"kind":"conv","name",ks,cin,cout,pad_l,pad_r,pad_mode,"up"}  (up>1 = sub-pixel depth->time)
  lin  : {"kind":"lin","up"}                       linear interpolation, num_pad_end=1, drop_last=True
  prelu: {"kind":"prelu","name","channels"}        per-channel PReLU shared over time
  leaky: {"kind":"leaky","alpha"}
  act  : {"kind":"act","fn"}                       final activation (soft_sigmoid, tanh, ...)
"""

PAD_ZERO, PAD_SYMMETRIC, PAD_EDGE = 0, 1, 2


def _same_pad(ks):
    # Keras "SAME" for stride 1, dilation 1: total ks-1, extra sample on the right for even ks
    total = ks - 1
    return total // 2, total - total // 2


def build_subnet(specs, base_name, in_channels, final_n_channels, final_nks, final_activation,
                 target_ups=None, pad_to_valid=False, remove_inactive_pad_layers=False,
                 use_prelu=True, alpha=0.2, force_causal=False):
    """Return (ops, total_ups, out_channels). See module docstring."""
    if force_causal:
        # reference custom_pulsed_generator.py:213-217: "would require a dedicated implementation"
        raise NotImplementedError("force_causal is not supported")
    ops = []
    total_ups = 1
    cin = in_channels
    if not specs:
        return ops, total_ups, cin

    def act_op(ii):
        if use_prelu:
            return {"kind": "prelu", "name": f"{base_name}_ActLayer_{ii}", "channels": cin}
        return {"kind": "leaky", "alpha": float(alpha)}

    for ii, spec in enumerate(specs):
        if spec[0] == "L":
            # reference :57-60 -- bare interpolation, no conv, no activation, total_ups NOT updated
            ops.append({"kind": "lin", "up": int(spec[1])})
            continue
        ks, nf = int(spec[0]), int(spec[1])
        linear_up = False
        up = 1
        if len(spec) > 2:
            if isinstance(spec[2], str):
                if spec[2][0] == "L":
                    linear_up = True
                up = int(spec[2][1:])
            else:
                up = int(spec[2])
        pad_l = (ks - 1) // 2 + ((ks - 1) % 2)
        pad_r = (ks - 1) // 2
        explicit_mode = PAD_EDGE if pad_to_valid else PAD_SYMMETRIC
        name = f"{base_name}_Layer_{ii}"
        if linear_up:
            # reference :74-89  pad -> VALID conv -> LinInterp
            ops.append({"kind": "conv", "name": name, "ks": ks, "cin": cin, "cout": nf,
                        "pad_l": pad_l, "pad_r": pad_r, "pad_mode": explicit_mode, "up": 1})
            cin = nf
            ops.append({"kind": "lin", "up": up})
        elif up > 1:
            # reference :91-108  sub-pixel conv; Keras SAME zero padding unless pad_to_valid
            if pad_to_valid:
                pl, pr, mode = pad_l, pad_r, PAD_EDGE
            else:
                pl, pr = _same_pad(ks)
                mode = PAD_ZERO
            ops.append({"kind": "conv", "name": name, "ks": ks, "cin": cin, "cout": nf * up,
                        "pad_l": pl, "pad_r": pr, "pad_mode": mode, "up": up})
            cin = nf
        else:
            # reference :109-122  pad -> VALID conv
            ops.append({"kind": "conv", "name": name, "ks": ks, "cin": cin, "cout": nf,
                        "pad_l": pad_l, "pad_r": pad_r, "pad_mode": explicit_mode, "up": 1})
            cin = nf
        ops.append(act_op(ii))
        total_ups *= up

    if final_nks is not None:
        # reference :126-138
        fks = int(final_nks)
        if pad_to_valid:
            pl = (fks - 1) // 2 + ((fks - 1) % 2)
            pr = (fks - 1) // 2
            mode = PAD_EDGE
        else:
            pl, pr = _same_pad(fks)
            mode = PAD_ZERO
        ops.append({"kind": "conv", "name": f"{base_name}_Layer_final", "ks": fks, "cin": cin,
                    "cout": int(final_n_channels), "pad_l": pl, "pad_r": pr, "pad_mode": mode, "up": 1})
        cin = int(final_n_channels)
        if (target_ups is not None) and total_ups != target_ups:
            # reference :29-35,140-144
            up = target_ups // total_ups
            if total_ups * up != target_ups:
                raise RuntimeError(f"get_missing_upsamling_factor::error:: Upsampling to target upsampling factor "
                                   f"{target_ups} from {total_ups} is not possible for subnet {base_name}")
            ops.append({"kind": "lin", "up": int(up)})
            total_ups *= up
        if final_activation is not None:
            ops.append({"kind": "act", "fn": str(final_activation).lower()})
    return ops, total_ups, cin


def subnet_time_factor(ops):
    """Actual time-axis expansion of the op list (differs from total_ups when a bare
    ["L", up] entry is present -- reference quirk, custom_pulsed_generator.py:57-60)."""
    fac = 1
    for op in ops:
        if op["kind"] == "lin":
            fac *= op["up"]
        elif op["kind"] == "conv":
            fac *= op["up"]
    return fac
