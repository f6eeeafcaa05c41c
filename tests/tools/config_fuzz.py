#!/usr/bin/env python3
"""One-off fuzz (GPU box): random model configurations x random ragged batches through the engine against the oracle.

Known non-bug outliers (fuzz_case.py shows them stage by stage): with add_subharm_chans the channels sin(2 pi phase / ii) of
the WRAPPED phase jump wherever the phase wraps (the reference's own design); an F0 contour that differs by 3e-4 Hz can wrap
one sample earlier, which moves that jump by a sample and shows as ~1e-2 in the audio around it (seed 30000 + 363).  A
second class of that kind (once in 1 100 draws, seed 52000 + 271): the lifter row of a frame is the nearest of 30 rows to its
smoothed log F0, and a contour within rounding of a midpoint selects the neighbouring row -- accepted (ROW) only if the frame
sits at a midpoint in the oracle and the audio meets the tolerance against the oracle run with the engine's rows."""
import os, sys, traceback
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mbexwn_vocoder_amd.config import ModelDims, canonical_config
from mbexwn_vocoder_amd.engine import MBExWNEngine
from mbexwn_vocoder_amd.tables import WaveTables
from mbexwn_vocoder_amd.weights import synthetic_weights
from oracle.mbexwn_oracle import OracleModel
M, W = "mbexwn_config:", "mbexwn_config:pp_mod_subnet:"
n_cases, seed0 = int(sys.argv[1]), int(sys.argv[2])
fails = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    over = {W + "n_channels": int(rng.choice([24, 32, 36, 48, 64, 68, 96])), W + "n_layers": int(rng.integers(1, 7)),
            W + "activation": str(rng.choice(["gtu", "gfu", "gsu", "glu"])),
            W + "cond_lin_upsampling": int(rng.choice([2, 4, 5, 10, 20])), W + "cond_kernel_size": int(rng.choice([1, 3, 5])),
            W + "dilation_rate_step": int(rng.choice([1, 2]))}
    if rng.random() < 0.4:
        over[W + "max_log2_dilation_rate"] = int(rng.integers(1, 5))
    if rng.random() < 0.3:
        over[W + "use_weight_norm"] = bool(rng.random() < 0.5)
        over[W + "use_equalized_lr"] = bool(rng.random() < 0.5)
    if rng.random() < 0.25:
        over[W + "pre_cond_layer_channels"] = [int(rng.choice([16, 24, 40]))]
    if rng.random() < 0.15:
        over[W + "disable_conditioning"] = True
    if rng.random() < 0.2:
        over[M + "pp_mod_subnet_noise_channel_sigma"] = 0
    if rng.random() < 0.2:
        over[M + "spect_filters_preserve_energy"] = True
    if rng.random() < 0.2:
        over[M + "filter_max_db_range"] = None
    if rng.random() < 0.2:
        over[M + "ps_env_order_scale"] = None
    if rng.random() < 0.15:
        over[M + "wavetable_config:add_subharm_chans"] = int(rng.integers(1, 3))
    if rng.random() < 0.15:
        over[W + "n_ch_groups"] = 2
    if rng.random() < 0.15:
        over[M + "use_prelu"] = False
    deep = len(sys.argv) > 3 and sys.argv[3] == "deep"
    if deep:
        # round 6: the reference's default depth and its neighbourhood -- 6 .. 12 layers without (mostly) a dilation cycle, so that
        # dilations up to 2048 run (F(4,3) over interleaved sub-sequences, pairs of sub-sequences per block, the direct-form
        # fall-back for short items), items of up to 260 frames with ragged batches
        over[W + "n_layers"] = int(rng.integers(6, 13))
        over[W + "dilation_rate_step"] = 1
        over.pop(W + "max_log2_dilation_rate", None)
        if rng.random() < 0.25:
            over[W + "max_log2_dilation_rate"] = int(rng.integers(6, 10))
        over[W + "cond_lin_upsampling"] = int(rng.choice([10, 10, 20, 5]))
    if len(sys.argv) > 3 and sys.argv[3] == "structure":     # structural variants: blocks, padding, excitation / filter paths
        pick = rng.random()
        if pick < 0.3:
            ups, chf = [([2, 1], [1, 0.5]), ([1, 1], [1, 1]), ([2], [1]), ([2, 1], [0.5, 1])][int(rng.integers(0, 4))]
            over[M + "pp_mod_subnet_upsampling_factors"], over[M + "pp_mod_subnet_channel_factors"] = ups, chf
            if int(np.prod(ups)) == 2:
                over[M + "pulse_channels"] = 10
                over[W + "cond_lin_upsampling"] = int(rng.choice([2, 5, 10]))
            over[W + "n_channels"] = int(rng.choice([32, 48, 64]))
            over.pop(W + "n_ch_groups", None)
            over.pop(M + "wavetable_config:add_subharm_chans", None)
        if rng.random() < 0.25:
            over[W + "padding"] = "CAUSAL"
        pick = rng.random()
        if pick < 0.15:
            over[M + "ps_off"] = True
        elif pick < 0.3:
            over[M + "ps_use_stft"] = False
        if rng.random() < 0.2:
            over[M + "pp_mod_subnet_use_pqmf"] = False
        if rng.random() < 0.2 and M + "wavetable_config:add_subharm_chans" not in over:
            pc = over.get(M + "pulse_channels", 5)
            over[M + "pulse_channels_use_pqmf"] = True
            over[M + "pulse_channels_multi_band_config"] = {"subbands": pc, "taps": int(rng.choice([8, 6]) * pc),
                                                            "cutoff_ratio": 0.6 / pc, "beta": 9.0}
        if rng.random() < 0.2:
            over[M + "wavetable_config:use_sinusoid_as_fun"] = True
    if rng.random() < 0.2:
        over[M + "pp_subnet"] = [[int(rng.choice([3, 5])), int(rng.choice([32, 48]))], [3, 32, "L2"]] if rng.random() < 0.5 else [[3, 40], [5, 24]]
    try:
        cfg = canonical_config(str(rng.choice(["SPEECH", "VOICE"])), **over)
        dims = ModelDims(cfg)
    except Exception as ee:                                  # an invalid combination: the reference's own config errors
        print(case, "config refused:", type(ee).__name__, str(ee)[:80], flush=True)
        continue
    try:
        raw = synthetic_weights(cfg, seed=int(rng.integers(1, 10 ** 6)), bias_std=0.05, alpha_jitter=0.05)
        wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
        # the default handle (auto: the form is calibrated on the model's own weights) in half of the draws, a pinned form
        # (direct, F(2,3), F(4,3), F(4,3) with batch-invariant kernels) in the others
        form = str(rng.choice(["auto", "auto", "auto", "auto", "f43", "f23", "direct", "f43i"]))
        eng = MBExWNEngine(cfg, raw, wt, conv_form=form.rstrip("i"), batch_invariant=form.endswith("i"))
        om, om32 = OracleModel(cfg, raw, wt), OracleModel(cfg, raw, wt, dtype=np.float32)
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 45))
        if deep:
            B, T = int(rng.integers(1, 4)), int(rng.choice([int(rng.integers(1, 60)), int(rng.integers(60, 261))]))
        lengths = [T] + [int(rng.integers(1, T + 1)) for _ in range(B - 1)]
        mel = np.clip(np.log(np.exp(rng.normal(-5.0, 2.0, size=(B, T, 80))) + 1e-5), -11.5, 2.0).astype(np.float32)
        rpf = dims.wn_in_rows_per_frame
        noise = rng.normal(size=(B, T * rpf)).astype(np.float32)
        got = eng.forward(torch.as_tensor(mel).cuda(), n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda"),
                          noise=torch.as_tensor(noise).cuda() if dims.noise_sigma else None).cpu().numpy()
        # yardstick: what float32 arithmetic itself does to this random model (the numpy float32 port of the same graph
        # against the float64 one): glu gates are unbounded and sub-harmonic sinusoids jump where the phase wraps, so some
        # draws are ill-conditioned; a kernel bug would show as a difference well above that
        worst, yard = 0.0, 0.0
        for ii, ll in enumerate(lengths):
            nz = noise[ii:ii + 1, :ll * rpf] if dims.noise_sigma else None
            ref = om.forward(mel[ii:ii + 1, :ll], nz)[0]
            amp = max(1.0, float(np.abs(ref).max()))
            worst = max(worst, float(np.abs(got[ii, :ll * 300] - ref).max()) / amp)
            yard = max(yard, float(np.abs(om32.forward(mel[ii:ii + 1, :ll], nz)[0] - ref).max()) / amp)
            assert np.all(got[ii, ll * 300:] == 0.0), "tail not zero"
        # The bar: the plain tolerance 1e-4 * max(1, |ref|) for EVERY handle on the bounded gates (gtu / gfu / gsu) -- the
        # default one and the pinned forms alike (round 5: with the F0-net in float64 the contour is exact, and the draws
        # between 4 and 12 yardsticks of round 4, all of which had a contour 1.2-1.4e-3 Hz off, are gone).  The 16-yardstick
        # escape is left to the draws that are ill-conditioned by construction: glu (unbounded linear half) and sub-harmonic
        # sinusoids (jumps at phase wraps).
        ill = dims.wn_activation == "glu" or bool(dims.wt_subharm)
        plain = not ill
        ok = worst <= (1e-4 if plain else max(1e-4, 16 * yard))
        note = ""
        if not ok and plain:
            # is it the form's rounding, or this random model's conditioning in float32?  The same inputs through a handle
            # pinned to the direct form: the default handle may not be materially worse (1.25x) than that one
            eng_d = MBExWNEngine(cfg, raw, wt, conv_form="direct")
            got_d = eng_d.forward(torch.as_tensor(mel).cuda(), n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda"),
                                  noise=torch.as_tensor(noise).cuda() if dims.noise_sigma else None).cpu().numpy()
            worst_d = 0.0
            for ii, ll in enumerate(lengths):
                nz = noise[ii:ii + 1, :ll * rpf] if dims.noise_sigma else None
                ref = om.forward(mel[ii:ii + 1, :ll], nz)[0]
                worst_d = max(worst_d, float(np.abs(got_d[ii, :ll * 300] - ref).max()) / max(1.0, float(np.abs(ref).max())))
            info = eng.conv_form_info()
            cal = f"{info['err_f43']:.1e} <= {info['threshold']:.1e}" if info["err_f43"] is not None else "not run"
            note = f" [direct form {worst_d:.1e}, calibration {cal}]"
            ok = worst <= 1.25 * worst_d and worst <= 16 * yard
            del eng_d
        if not ok and getattr(om, "ceps_windows", None) is not None and not (cfg["mbexwn_config"].get("ps_off") or not cfg["mbexwn_config"].get("ps_use_stft", True)):
            # a second known class: the lifter row of a frame is the NEAREST of n rows to the frame's smoothed log F0 (reference
            # custom_pulsed_generator.py:507-525) -- discontinuous at the midpoints.  Accepted only if every frame whose row differs
            # sits within 2e-3 rows of a midpoint in the oracle AND the audio meets the plain tolerance against the oracle run
            # with the engine's rows
            ix_hip = eng.stage("ceps_index").cpu().numpy().view(np.int32)
            worst_r, flips, borderline = 0.0, 0, True
            for ii, ll in enumerate(lengths):
                nz = noise[ii:ii + 1, :ll * rpf] if dims.noise_sigma else None
                m64 = mel[ii:ii + 1, :ll]
                f0_ref = om.generate_f0(np.asarray(m64).astype(om.dtype))
                ix_ref, pos = om.cepstral_window_index(f0_ref, return_position=True)
                diff = np.nonzero(ix_hip[ii, :ll] != ix_ref[0])[0]
                flips += len(diff)
                borderline &= bool(np.all(np.abs(np.abs(pos[0, diff] - np.floor(pos[0, diff])) - 0.5) < 2e-3))
                exc_ref = om.generate_excitation(np.asarray(m64).astype(om.dtype), f0_ref, nz)
                env = om.generate_specenv(np.asarray(m64).astype(om.dtype), f0_ref, window_index=ix_hip[ii:ii + 1, :ll].astype(np.int64))
                ref_r = om.istft(om.stft(exc_ref, ll) * env, f0_ref.shape[1] * int(om.sample_rate // om.pulse_rate))[0, :ll * 300]
                worst_r = max(worst_r, float(np.abs(got[ii, :ll * 300] - ref_r).max()) / max(1.0, float(np.abs(ref_r).max())))
            if flips and borderline and worst_r <= 1e-4:
                ok = "row"
                note += f" [{flips} lifter row(s) at a midpoint; with the engine's rows {worst_r:.1e}]"
        if not ok and dims.wt_subharm:
            # the known class (see the docstring): the wrapped phases of the two F0 contours differ by a whole turn somewhere
            ph_hip = eng.wavetable(eng.stage("f0"))[1].cpu().numpy()
            for ii, ll in enumerate(lengths):
                f0_ref = om.generate_f0(mel[ii:ii + 1, :ll].astype(np.float64))
                ph_ref = om.phase_from_f0(np.asarray(f0_ref, dtype=np.float32))[0]
                if np.any(np.abs(ph_hip[ii, :ph_ref.shape[0]] - ph_ref) > 0.5):
                    ok = "wrap"
        fails += not ok
        if deep:
            note += " kernels " + ",".join(kk.replace("f43_strided", "S").replace("folded_start", "0") for kk in eng.conv_form_info()["gate_kernels"])
        print(case, "OK  " if ok is True else ({"wrap": "WRAP", "row": "ROW "}.get(ok, "FAIL")), f"{worst:.1e}", f"(f32 port {yard:.1e}){note}", "form", form + "->" + eng.conv_form_info()["form"], "B", B, "T", T, {kk.split(':')[-1]: vv for kk, vv in over.items()}, flush=True)
        del eng
    except Exception:                                        # noqa: BLE001
        fails += 1
        print(case, "EXC", over, flush=True)
        traceback.print_exc()
print("failures:", fails)
sys.exit(1 if fails else 0)
