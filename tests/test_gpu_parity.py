"""HIP engine (through the C ABI) against the CPU oracle and the reference-generated golden vectors.

Tolerances (float32 HIP path vs float64 oracle), stated per SURVEY.md section 8(c):
  isolated stage : max-abs-delta <= 2e-5 * max(1, max|ref|)
  end to end     : max-abs-delta <= 1e-4 * max(1, max|ref|)   (audio of amplitude ~1..4)
  integer work   : bit exact (phase accumulator, lifter indices away from rounding ties)
"""
import os

import numpy as np
import pytest

from oracle import mbexwn_oracle as orc
from helpers import form_kwargs, GOLDEN_CASES, build_case, synthetic_inputs

pytestmark = pytest.mark.gpu

STAGE_TOL = 2e-5
E2E_TOL = 1e-4
E2E_TIGHT = 2e-5         # what the default handle is held to where its contour is exact (F0-net in float64, round 5)


def _tol(ref, rel):
    return rel * max(1.0, float(np.max(np.abs(ref))))


def _maxdiff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    wide = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    return float(np.max(np.abs(a.astype(wide) - b.astype(wide))))


@pytest.fixture(scope="module")
def torch():
    import torch as _torch
    assert _torch.cuda.is_available(), "GPU tests need an MI355X"
    return _torch


_ENGINES = {}


def get_engine(case_key, voice, overrides):
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    if case_key not in _ENGINES:
        cfg, raw, wt = build_case(voice, overrides)
        _ENGINES[case_key] = (MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt), cfg, raw, wt)
    return _ENGINES[case_key]


SMALL = ("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3})
CANON = ("SPEECH", {})
VOICE = ("VOICE", {})


def dev(torch, arr, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype or torch.float32).cuda()


# ------------------------------------------------------------------------------------------------
# stage parity
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,ks,dil,mode,rows,prelu", [
    (80, 128, 3, 1, "SYMMETRIC", 37, True),
    (128, 64, 3, 1, "EDGE", 5, True),
    (64, 1, 1, 1, "CONSTANT", 130, False),
    (6, 32, 1, 1, "CONSTANT", 200, False),
    (30, 15, 1, 1, "CONSTANT", 260, False),
    (32, 64, 3, 4, "CONSTANT", 300, False),
    (320, 640, 3, 16, "CONSTANT", 150, False),
    (340, 680, 3, 2, "CONSTANT", 129, False),
    (256, 240, 1, 1, "CONSTANT", 3, False),
    (80, 1280, 3, 1, "CONSTANT", 1, False),
    # the reference's default depth reaches d = 2048 (custom_AE_layers.py:229-233): rows on both sides of the taps, items
    # shorter than the dilation (only the centre tap sees data), a large launch (>= 12 288 rows)
    (32, 64, 3, 64, "CONSTANT", 300, False),
    (320, 640, 3, 512, "CONSTANT", 1300, False),
    (64, 128, 3, 2048, "CONSTANT", 4500, False),
    (64, 128, 3, 2048, "CONSTANT", 1200, False),
    (96, 192, 3, 1024, "CONSTANT", 7000, False),
])
def test_conv1d(torch, cin, cout, ks, dil, mode, rows, prelu):
    eng = get_engine("small", *SMALL)[0]
    rng = np.random.default_rng(cin * 1000 + cout)
    B = 2
    x = rng.normal(size=(B, rows, cin)).astype(np.float32)
    w = (rng.normal(size=(ks, cin, cout)) / np.sqrt(ks * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    alpha = rng.uniform(0.05, 0.4, size=(cout,)).astype(np.float32) if prelu else None
    total = (ks - 1) * dil
    if mode == "CONSTANT":
        pl, pr = total // 2, total - total // 2
    else:
        pl, pr = (ks - 1) // 2 + ((ks - 1) % 2), (ks - 1) // 2
    ref = orc.conv1d_valid(orc.pad_time(x.astype(np.float64), pl, pr, mode), w.astype(np.float64),
                           b.astype(np.float64), dilation=dil)
    if prelu:
        ref = orc.prelu(ref, alpha.astype(np.float64))
    got = eng.conv1d(dev(torch, x), dev(torch, w), dev(torch, b), dev(torch, alpha) if prelu else None,
                     dilation=dil, pad_l=pl, pad_mode={"CONSTANT": 0, "SYMMETRIC": 1, "EDGE": 2}[mode])
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert _maxdiff(got.cpu().numpy(), ref) <= _tol(ref, STAGE_TOL)


@pytest.mark.parametrize("cin,cout,ks,mode,rows,prelu", [
    (80, 128, 3, "SYMMETRIC", 37, True),       # the F0-net's layers
    (128, 128, 3, "SYMMETRIC", 240, True),
    (128, 64, 3, "EDGE", 5, True),
    (64, 1, 1, "CONSTANT", 130, False),
    (84, 20, 5, "CONSTANT", 33, False),        # a short last group of 16 channels, columns that do not fill a tile
    (12, 7, 3, "SYMMETRIC", 100, True),
    (4, 16, 1, "CONSTANT", 1, False),
])
def test_conv1d_f64_accumulation(torch, cin, cout, ks, mode, rows, prelu):
    """mbx_conv1d_f64acc (csrc/conv_mfma.hip::conv1d_f64_tile, the F0-net's arithmetic): float32 operands, float64 sums on
    the f64 matrix cores, one rounding -- every output is the float32 nearest to the exact result (half an ulp)."""
    eng = get_engine("small", *SMALL)[0]
    rng = np.random.default_rng(cin * 1000 + cout + 7)
    B = 2
    x = rng.normal(size=(B, rows, cin)).astype(np.float32)
    w = (rng.normal(size=(ks, cin, cout)) / np.sqrt(ks * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    alpha = rng.uniform(0.05, 0.4, size=(cout,)).astype(np.float32) if prelu else None
    if mode == "CONSTANT":
        pl, pr = (ks - 1) // 2, (ks - 1) - (ks - 1) // 2
    else:
        pl, pr = (ks - 1) // 2 + ((ks - 1) % 2), (ks - 1) // 2
    ref = orc.conv1d_valid(orc.pad_time(x.astype(np.float64), pl, pr, mode), w.astype(np.float64), b.astype(np.float64))
    if prelu:
        ref = orc.prelu(ref, alpha.astype(np.float64))
    got = eng.conv1d(dev(torch, x), dev(torch, w), dev(torch, b), dev(torch, alpha) if prelu else None, pad_l=pl,
                     pad_mode={"CONSTANT": 0, "SYMMETRIC": 1, "EDGE": 2}[mode], f64_accumulate=True).cpu().numpy()
    assert got.shape == ref.shape
    half_ulp = 0.5 * np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got.astype(np.float64) - ref) <= half_ulp * 1.001 + 1e-300)
    plain = eng.conv1d(dev(torch, x), dev(torch, w), dev(torch, b), dev(torch, alpha) if prelu else None, pad_l=pl,
                       pad_mode={"CONSTANT": 0, "SYMMETRIC": 1, "EDGE": 2}[mode]).cpu().numpy()
    assert _maxdiff(plain, ref) <= _tol(ref, STAGE_TOL)          # (the float32 kernels on the same case, for the record)


def test_conv1d_f64_tile_shapes_give_the_same_bits(torch):
    """Large launches run the float64 tile as 32 x 32 blocks inside the large mel-rate launch kernel, small ones as 16 x 16
    blocks: every output is summed in the same order, so a row's bits do not depend on the launch it ran in."""
    eng = get_engine("small", *SMALL)[0]
    rng = np.random.default_rng(99)
    B, rows, cin, cout, ks = 2, 6200, 80, 48, 3          # 12 400 rows: the large-launch tile
    x = rng.normal(size=(B, rows, cin)).astype(np.float32)
    w = (rng.normal(size=(ks, cin, cout)) / np.sqrt(ks * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    alpha = rng.uniform(0.05, 0.4, size=(cout,)).astype(np.float32)
    big = eng.conv1d(dev(torch, x), dev(torch, w), dev(torch, b), dev(torch, alpha), pad_l=1, pad_mode=1,
                     f64_accumulate=True).cpu().numpy()
    small = eng.conv1d(dev(torch, x[:, :300]), dev(torch, w), dev(torch, b), dev(torch, alpha), pad_l=1, pad_mode=1,
                       f64_accumulate=True).cpu().numpy()
    assert np.array_equal(big[:, :298], small[:, :298])            # (row 299 sees the short item's own edge)
    ref = orc.prelu(orc.conv1d_valid(orc.pad_time(x[:, 5900:].astype(np.float64), 1, 1, "SYMMETRIC"), w.astype(np.float64),
                                     b.astype(np.float64)), alpha.astype(np.float64))
    tail = big[:, 5901:].astype(np.float64)
    assert np.all(np.abs(tail - ref[:, 1:]) <= 0.5 * np.spacing(np.abs(ref[:, 1:]).astype(np.float32)) * 1.001 + 1e-300)


@pytest.mark.parametrize("ks,cin,cout,mode,rows", [
    (3, 80, 256, "CONSTANT", 800),       # the mel-rate convolutions of the front end (K = 240: quarters of 7 / 8 groups)
    (3, 256, 256, "CONSTANT", 800),
    (1, 256, 240, "CONSTANT", 800),      # columns that do not fill the second 128-column tile
    (3, 80, 1280, "CONSTANT", 800),
    (3, 80, 128, "SYMMETRIC", 777),      # padding per row in the first / last tile of an item; a short last tile
    (3, 88, 132, "EDGE", 801),
    (5, 24, 36, "CONSTANT", 790),        # taps that change inside a slice
    (3, 16, 4, "CONSTANT", 800),         # one chunk of columns, quarters of 1 / 2 groups
    (1, 16, 64, "CONSTANT", 800),        # two groups only (empty K quarters): stays on the small-launch tile at every size
    (1, 8, 36, "CONSTANT", 800),
])
def test_mel_tile_large_launch_same_bits(torch, ks, cin, cout, mode, rows):
    """Large launches (>= 12 288 rows) run the mel-rate convolutions as LDS-DMA tiles of 64 x 128 (csrc/conv_mfma.hip::
    conv1d_mel_tile_dma), small ones as 32 x 32 tiles with K split over the waves: the same MFMA steps in the same order and
    the same fold of the K quarters, so a row's bits do not depend on the launch it ran in -- and both match the oracle."""
    eng = get_engine("small", *SMALL)[0]
    rng = np.random.default_rng(ks * 100000 + cin * 100 + cout)
    B = 16
    x = rng.normal(size=(B, rows, cin)).astype(np.float32)
    w = (rng.normal(size=(ks, cin, cout)) / np.sqrt(ks * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    alpha = rng.uniform(0.05, 0.4, size=(cout,)).astype(np.float32)
    if mode == "CONSTANT":
        pl, pr = (ks - 1) // 2, (ks - 1) - (ks - 1) // 2
    else:
        pl, pr = (ks - 1) // 2 + ((ks - 1) % 2), (ks - 1) // 2
    pm = {"CONSTANT": 0, "SYMMETRIC": 1, "EDGE": 2}[mode]
    xd, wd, bd, ad = dev(torch, x), dev(torch, w), dev(torch, b), dev(torch, alpha)
    big = eng.conv1d(xd, wd, bd, ad, pad_l=pl, pad_mode=pm).cpu().numpy()
    small = np.concatenate([eng.conv1d(xd[i:i + 1], wd, bd, ad, pad_l=pl, pad_mode=pm).cpu().numpy() for i in range(B)])
    assert np.array_equal(big, small)
    ref = orc.prelu(orc.conv1d_valid(orc.pad_time(x[:2].astype(np.float64), pl, pr, mode), w.astype(np.float64), b.astype(np.float64)),
                    alpha.astype(np.float64))
    assert _maxdiff(big[:2], ref) <= _tol(ref, STAGE_TOL)


@pytest.mark.parametrize("rows,channels,up", [(7, 1, 100), (5, 640, 10), (1, 3, 10)])
def test_lin_interp(torch, rows, channels, up):
    eng = get_engine("small", *SMALL)[0]
    rng = np.random.default_rng(rows)
    x = rng.normal(size=(2, rows, channels)).astype(np.float32)
    ref = orc.lin_interp(x.astype(np.float64), up)
    got = eng.lin_interp(dev(torch, x), up).cpu().numpy()
    assert _maxdiff(got, ref) <= 1e-6


@pytest.mark.parametrize("n", [100, 1000, 2300, 12345])
def test_wavetable_phase_bit_exact(torch, n):
    eng, om = get_engine("small", *SMALL)[:2]
    rng = np.random.default_rng(n)
    # slowly varying contour inside and outside the table grid
    f0 = (30 + 650 * np.abs(np.sin(np.cumsum(rng.normal(0, 0.002, size=(3, n)), axis=1)))).astype(np.float32)
    pulse, phase = eng.wavetable(dev(torch, f0))
    torch.cuda.synchronize()
    ref_phase = om.phase_from_f0(f0)
    assert np.array_equal(phase.cpu().numpy(), ref_phase), "phase accumulator must be bit exact"
    ref_pulse = om.wavetable(f0)
    assert _maxdiff(pulse.cpu().numpy(), ref_pulse) <= 2e-6


def test_wavetable_phase_beyond_the_chunk_offset_table(torch):
    """More than 1 024 phase chunks (128 s of audio at the 8 kHz pulse rate): the kernel's per-block table of chunk
    offsets does not hold them and every sample walks the chain itself -- still the reference's order of additions."""
    eng, om = get_engine("small", *SMALL)[:2]
    n = 1_030_500
    rng = np.random.default_rng(3)
    f0 = (60 + 500 * np.abs(np.sin(np.cumsum(rng.normal(0, 0.0005, size=(1, n)), axis=1)))).astype(np.float32)
    pulse, phase = eng.wavetable(dev(torch, f0))
    torch.cuda.synchronize()
    assert np.array_equal(phase.cpu().numpy(), om.phase_from_f0(f0)), "phase accumulator must be bit exact"
    assert float(pulse[:, n - 4000:].abs().max()) > 0.1            # (the lookup itself is covered at the small sizes)


@pytest.mark.parametrize("steps", [1, 9, 64, 65, 333])
def test_pqmf_synthesis(torch, steps):
    eng, om = get_engine("small", *SMALL)[:2]
    rng = np.random.default_rng(steps)
    x = rng.normal(size=(2, steps, 15)).astype(np.float32)
    ref = om.pqmf_synthesis(x.astype(np.float64))
    got = eng.pqmf_synthesis(dev(torch, x)).cpu().numpy()
    assert _maxdiff(got, ref) <= _tol(ref, STAGE_TOL)


def test_pqmf_impulse_response_is_filter_bank(torch):
    """An impulse in band k at step m reproduces 15*g_k shifted to sample 15*m (edge case: known answer)."""
    eng, om = get_engine("small", *SMALL)[:2]
    x = np.zeros((1, 40, 15), dtype=np.float32)
    x[0, 20, 3] = 1.0
    got = eng.pqmf_synthesis(dev(torch, x)).cpu().numpy()[0]
    g = om.pqmf_syn[3]
    expect = np.zeros(600)
    # y[n] = 15 * g[j], j = 15*20 + 60 - n
    for n in range(600):
        j = 300 + 60 - n
        if 0 <= j <= 120:
            expect[n] = 15 * g[j]
    assert _maxdiff(got, expect) <= 1e-6


@pytest.mark.parametrize("frames", [1, 2, 3, 4, 11])
def test_stft_filter(torch, frames):
    eng, om = get_engine("small", *SMALL)[:2]
    rng = np.random.default_rng(frames)
    B = 2
    exc = rng.normal(size=(B, frames * 300)).astype(np.float32)
    ceps = (0.05 * rng.normal(size=(B, frames, 240))).astype(np.float32)
    idx = rng.integers(0, 30, size=(B, frames)).astype(np.int32)
    x = ceps.astype(np.float64) * om.ceps_windows[idx]
    full = np.zeros((B, frames, om.fft_size))
    full[:, :, 1:240] = x[:, :, 1:]
    spec = np.fft.rfft(full, axis=-1)
    env = np.exp(om.max_log_range * np.tanh(spec.real) + 1j * spec.imag)
    ref = om.istft(om.stft(exc.astype(np.float64), frames) * env, frames * 300)
    got = eng.stft_filter(dev(torch, exc), dev(torch, ceps), dev(torch, idx, torch.int32)).cpu().numpy()
    assert got.shape == ref.shape
    assert _maxdiff(got, ref) <= _tol(ref, STAGE_TOL)


@pytest.mark.parametrize("scale", [0.5, 5.0])
def test_stft_filter_large_cepstra(torch, scale):
    """The wave-per-frame kernel evaluates exp(R tanh(Re S) + j Im S) with short forms of its own (hardware exp2, a three-term
    Cody-Waite reduction for sin / cos: csrc/stft_filter.hip): cepstra 10 and 100 times the usual size -- phases of +-8 and
    +-80 radians and a fully saturated tanh -- must still match the float64 oracle at
    the stage tolerance (the envelope's magnitude is bounded by exp(R), its phase error is an absolute error in radians)."""
    eng, om = get_engine("small", *SMALL)[:2]
    rng = np.random.default_rng(17)
    B, frames = 2, 9
    exc = rng.normal(size=(B, frames * 300)).astype(np.float32)
    ceps = (scale * rng.normal(size=(B, frames, 240))).astype(np.float32)
    idx = rng.integers(0, 30, size=(B, frames)).astype(np.int32)
    x = ceps.astype(np.float64) * om.ceps_windows[idx]
    full = np.zeros((B, frames, om.fft_size))
    full[:, :, 1:240] = x[:, :, 1:]
    spec = np.fft.rfft(full, axis=-1)
    assert np.abs(spec.imag).max() > (5.0 if scale < 1 else 50.0)
    env = np.exp(om.max_log_range * np.tanh(spec.real) + 1j * spec.imag)
    ref = om.istft(om.stft(exc.astype(np.float64), frames) * env, frames * 300)
    got = eng.stft_filter(dev(torch, exc), dev(torch, ceps), dev(torch, idx, torch.int32)).cpu().numpy()
    tol = _tol(ref, STAGE_TOL)      # (measured: 1.4e-4 on amplitude 374 and 6.5e-4 on amplitude 494 with 86 rad phases)
    print(f"\nstft filter, cepstra x {scale}: max |Im S| {np.abs(spec.imag).max():.0f} rad, max|ref| {np.abs(ref).max():.1f}, "
          f"error {_maxdiff(got, ref):.2e} (tolerance {tol:.1e})")
    assert _maxdiff(got, ref) <= tol


def test_stft_identity_envelope_reproduces_head_tail_taper(torch):
    """Zero cepstrum => H == 1: output = input x overlap-normalisation; the reference keeps frames 0..T-1
    only, so the first hop and the last two hops are attenuated (3/4, 3/4, 2/4 of the overlaps)."""
    eng, om = get_engine("small", *SMALL)[:2]
    T = 8
    exc = np.ones((1, T * 300), dtype=np.float32)
    ceps = np.zeros((1, T, 240), dtype=np.float32)
    idx = np.zeros((1, T), dtype=np.int32)
    got = eng.stft_filter(dev(torch, exc), dev(torch, ceps), dev(torch, idx, torch.int32)).cpu().numpy()[0]
    ref = om.istft(om.stft(exc.astype(np.float64), T), T * 300)[0]
    assert _maxdiff(got, ref) <= 1e-5
    assert np.all(np.abs(got[300:T * 300 - 600] - 1.0) < 1e-5)
    assert got[T * 300 - 1] < 0.6 and got[0] < 1.0


# ------------------------------------------------------------------------------------------------
# end to end
# ------------------------------------------------------------------------------------------------
def _forward_both(torch, key, spec, batch, frames, seed=42):
    eng, om = get_engine(key, *spec)[:2]
    mel, noise = synthetic_inputs(seed, batch, frames)
    audio = eng.forward(dev(torch, mel), noise=dev(torch, noise))
    torch.cuda.synchronize()
    ref, stages = om.forward(mel, noise, return_stages=True)
    return eng, om, mel, noise, audio.cpu().numpy(), ref, stages


@pytest.mark.parametrize("key,spec,batch,frames", [
    ("small", SMALL, 2, 23), ("small", SMALL, 1, 1), ("small", SMALL, 3, 2),
    ("canon", CANON, 1, 12), ("canon", CANON, 2, 40), ("voice", VOICE, 1, 17),
])
def test_forward_matches_oracle(torch, key, spec, batch, frames):
    eng, om, mel, noise, got, ref, stages = _forward_both(torch, key, spec, batch, frames)
    assert got.shape == (batch, frames * 300)
    assert np.all(np.isfinite(got))
    f0 = eng.stage("f0").cpu().numpy()
    # the F0-net runs in float64 (mbx_config.f0_accumulate): the contour is the float32 nearest to the oracle's, in Hz
    assert np.all(np.abs(f0.astype(np.float64) - stages["f0"]) <= 0.5 * np.spacing(stages["f0"].astype(np.float32)) * 1.001)
    exc = eng.stage("excitation").cpu().numpy()
    assert _maxdiff(exc, stages["excitation"]) <= _tol(stages["excitation"], E2E_TOL)
    assert _maxdiff(got, ref) <= _tol(ref, E2E_TOL)
    assert _maxdiff(got, ref) <= _tol(ref, E2E_TIGHT), "with the exact contour the end-to-end error is the WaveNet's and the filters'"


def test_mel_pointer_that_is_not_16_byte_aligned(torch):
    """The C ABI takes any float pointer.  The float64 F0 chain reads the mel rows as 16-byte pieces; a caller's mel that
    starts in the middle of one (a view into a larger buffer) is copied into the workspace first: the same handle and the same
    mel give the same bits however the call was made (round 5's advisor finding: the misaligned call used to drop to float32
    hidden layers)."""
    eng, om = get_engine("canon", *CANON)[:2]
    mel, noise = synthetic_inputs(11, 1, 24)
    base = torch.zeros(1 + mel.size, dtype=torch.float32, device="cuda")
    base[1:] = dev(torch, mel).reshape(-1)
    view = base[1:].view(1, 24, 80)
    assert view.is_contiguous() and view.data_ptr() % 16 == 4
    ref = om.forward(mel, noise)
    got_u = eng.forward(view, noise=dev(torch, noise)).cpu().numpy()
    f0_u = eng.stage("f0").cpu().numpy()
    got_a = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    f0_a = eng.stage("f0").cpu().numpy()
    assert eng.conv_form_info()["f0_float64_chain"]
    assert np.array_equal(f0_u, f0_a) and np.array_equal(got_u, got_a)
    assert _maxdiff(got_a, ref) <= _tol(ref, E2E_TIGHT)


def test_f0_contour_in_float64_keeps_the_error_from_growing_with_the_length(torch):
    """The contour feeds the float32 phase integrator (reference tf_wavetable.py:429-492): a contour that is off in the last
    bit of a few samples sends the phase chain down another rounding path for the rest of the utterance.  With the F0-net in
    float64 (the default) the contour is the float32 nearest to the float64 oracle's everywhere, the phase chains coincide
    and a 10 s utterance is as exact as a short one; the float32 F0-net (mbx_config.f0_accumulate = MBX_F0_ACC_F32, the
    behaviour up to ABI 8) stays within the stated tolerance of short utterances only."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    eng, om, cfg, raw, wt = get_engine("canon", *CANON)
    eng32 = MBExWNEngine(cfg, raw, wt, f0_accumulate="f32")
    mel, noise = synthetic_inputs(77, 1, 800)
    ref, stages = om.forward(mel, noise, return_stages=True)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    f0 = eng.stage("f0").cpu().numpy()
    got32 = eng32.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    f032 = eng32.stage("f0").cpu().numpy()
    err_f0, err_f032 = _maxdiff(f0, stages["f0"]), _maxdiff(f032, stages["f0"])
    err, err32 = _maxdiff(got, ref), _maxdiff(got32, ref)
    print(f"f0 error {err_f0:.3e} Hz (float32 net {err_f032:.3e}), audio error {err:.3e} (float32 net {err32:.3e}) on |audio| <= {np.abs(ref).max():.2f}")
    assert np.all(np.abs(f0.astype(np.float64) - stages["f0"]) <= 0.5 * np.spacing(stages["f0"].astype(np.float32)) * 1.001)
    assert err_f0 <= 2e-4 and err_f0 < err_f032                   # <= 2e-4 Hz on the canonical model
    assert _maxdiff(eng.stage("excitation").cpu().numpy(), stages["excitation"]) <= _tol(stages["excitation"], E2E_TIGHT)
    assert err <= _tol(ref, E2E_TIGHT) and err <= err32


@pytest.mark.parametrize("case", sorted(GOLDEN_CASES))
def test_forward_matches_reference_goldens(torch, golden_dir, case):
    """HIP path vs the output of the reference's own model code (float32 emulation run)."""
    gold = np.load(os.path.join(golden_dir, "reference_forward_f32.npz"))
    voice, overrides, batch, frames = GOLDEN_CASES[case]
    eng = get_engine("gold_" + case, voice, overrides)[0]
    mel, noise = gold[f"{case}/mell"], gold[f"{case}/noise"]
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    ref = gold[f"{case}/audio"]
    assert _maxdiff(got, ref) <= _tol(ref, E2E_TOL)
    assert _maxdiff(eng.stage("excitation").cpu().numpy(), gold[f"{case}/excitation"]) <= _tol(gold[f"{case}/excitation"], E2E_TOL)
    # (with sub-harmonic channels the pulse tensor is (B, N, 1 + n): the stage is its flat image)
    assert _maxdiff(eng.stage("pulse").cpu().numpy(), gold[f"{case}/pulse"].reshape(batch, -1)) <= 5e-4     # F0 rounding moves the phase
    cond = eng.stage("cond").cpu().numpy().reshape(batch, -1)
    # the engine keeps the conditioning at the sub-pixel rate (2T, 2C) and interpolates on the fly
    pulse_from_ref_f0, phase = eng.wavetable(dev(torch, gold[f"{case}/f0"]))
    assert np.array_equal(phase.cpu().numpy(), gold[f"{case}/phase"])
    assert _maxdiff(pulse_from_ref_f0.cpu().numpy(), gold[f"{case}/pulse"]) <= 2e-6
    assert cond.shape[1] == frames * eng.dims.cond_conv_upsampling * 2 * eng.dims.wn_channels


@pytest.mark.parametrize("form", ["0", "2", "44"])
@pytest.mark.parametrize("key,spec,batch,frames", [("canon", CANON, 2, 40), ("voice", VOICE, 1, 17)])
def test_gate_layer_forms_match_oracle(torch, monkeypatch, form, key, spec, batch, frames):
    """The dilated convolution has three float32 implementations (direct, Winograd F(2,3), Winograd F(4,3); the engine
    calibrates and picks by launch size): each one is pinned here (mbx_config.wn_conv_form) and held to the same tolerance,
    including a ragged batch."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*spec)
    eng = MBExWNEngine(cfg, raw, wt, **form_kwargs(form))
    om = get_engine(key, *spec)[1]
    mel, noise = synthetic_inputs(5, batch, frames)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    ref = om.forward(mel, noise)
    assert _maxdiff(got, ref) <= _tol(ref, E2E_TOL)
    if batch > 1:
        lengths = [frames, frames // 3]
        nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
        rag = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
        ll = lengths[1]
        single = eng.forward(dev(torch, mel[1:2, :ll]), noise=dev(torch, noise[1:2, :ll * 20])).cpu().numpy()
        assert np.array_equal(rag[1, :ll * 300], single[0])
        assert np.all(rag[1, ll * 300:] == 0.0)


@pytest.mark.parametrize("key,spec", [("canon", CANON), ("voice", VOICE)])
def test_start_convolution_folded_into_layer_0_matches_the_unfolded_graph(torch, monkeypatch, key, spec):
    """Layer 0 with the start convolution folded in (csrc/wn_gate0.hip, the default) against the un-folded graph
    (mbx_config.wn_keep_start: start kernel writes h0, layer 0 runs the full dilated convolution): same audio to float32
    rounding, both inside the tolerance against the oracle; ragged batch, item shorter than one block."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*spec)
    om = get_engine(key, *spec)[1]
    mel, noise = synthetic_inputs(7, 3, 45)
    lengths = [45, 13, 1]
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    outs = {}
    for flag in ("1", "0"):
        eng = MBExWNEngine(cfg, raw, wt, keep_start=flag == "0")
        assert eng.folds_start == (flag == "1")
        outs[flag] = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    for ii, ll in enumerate(lengths):
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * 20])[0]
        for flag in ("1", "0"):
            assert _maxdiff(outs[flag][ii, :ll * 300], ref) <= _tol(ref, E2E_TOL)
            assert np.all(outs[flag][ii, ll * 300:] == 0.0)
        assert _maxdiff(outs["1"][ii], outs["0"][ii]) <= _tol(ref, 2e-5)


@pytest.mark.parametrize("overrides", [
    {"mbexwn_config:pp_mod_subnet:n_channels": 36, "mbexwn_config:pp_mod_subnet:n_layers": 3},      # C not a multiple of 8
    {"mbexwn_config:pp_mod_subnet:n_channels": 24, "mbexwn_config:pp_mod_subnet:n_layers": 7,
     "mbexwn_config:pp_mod_subnet:max_log2_dilation_rate": 3},                                         # dilation cycle 1..8,1..4
    {"mbexwn_config:pp_mod_subnet:n_channels": 64, "mbexwn_config:pp_mod_subnet:n_layers": 1},       # one layer: tail only
    {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 4,
     "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5},                                            # 4 x 5 conditioning: un-folded first layer, generic gate
    {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 4,
     "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 20, "mbexwn_config:pp_mod_subnet:cond_kernel_size": 5},
    {"mbexwn_config:pp_mod_subnet:n_channels": 36, "mbexwn_config:pp_mod_subnet:n_layers": 4,
     "mbexwn_config:pp_mod_subnet:padding": "CAUSAL", "mbexwn_config:pp_mod_subnet:cond_kernel_size": 5},   # all padding in front
], ids=["C36_L3", "C24_L7_cycle", "C64_L1", "cond_4x5", "cond_1x20_k5", "causal_C36_k5"])
def test_other_wavenet_geometries(torch, overrides):
    """Every size is configuration driven: partial channel tiles and slices, repeating dilation cycles, a single layer
    (the folded skip path then consists of the tail kernel alone), other splits of the conditioning up-sampling (x5
    does not fit the folded first layer's kernel nor the Winograd kernels' conditioning stage: the handle keeps the
    un-folded first layer and the generic gate kernel)."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case("SPEECH", overrides)
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    for batch, frames in ((2, 21), (1, 3)):
        mel, noise = synthetic_inputs(batch + frames, batch, frames)
        got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
        ref = om.forward(mel, noise)
        assert _maxdiff(got, ref) <= _tol(ref, E2E_TOL)
    with pytest.raises(ValueError, match="multiple of 4"):
        MBExWNEngine(*build_case("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 34}))


def test_subharmonic_channels_through_the_folded_first_layer(torch):
    """add_subharm_chans with 3 folded samples per row (pulse rate 4 800 Hz): 6 excitation channels + noise + the constant
    channel = the 8 channels of wn_gate0_kernel, so the sub-harmonic sinusoids go through the folded first layer; the
    canonical 5 x 2 channels (golden case "subharm") take the un-folded start convolution instead."""
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import synthetic_weights
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pulse_rate_factor": 5, "mbexwn_config:pulse_channels": 3,
                                        "mbexwn_config:wavetable_config:add_subharm_chans": 1,
                                        "mbexwn_config:pp_mod_subnet:n_channels": 64, "mbexwn_config:pp_mod_subnet:n_layers": 3})
    dims = ModelDims(cfg)
    assert dims.pulse_channels_eff == 6 and dims.wn_in_channels == 7 and dims.pulse_per_frame == 60
    raw = synthetic_weights(cfg, seed=21, bias_std=0.05, alpha_jitter=0.05)
    wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    assert "wn.conv1D_0.start_fold" in eng._tensors
    mel, noise = synthetic_inputs(9, 2, 25)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise),
                      n_frames=torch.tensor([25, 11], dtype=torch.int32, device="cuda")).cpu().numpy()
    for ii, ll in enumerate((25, 11)):
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * 20])[0]
        assert _maxdiff(got[ii, :ll * 300], ref) <= _tol(ref, E2E_TOL)
    pulse, phase = eng.wavetable(eng.stage("f0"))
    assert pulse.shape == (2, 25 * 60, 2)
    sub = np.sin((phase.cpu().numpy().astype(np.float32) * np.float32(2)) * np.float32(np.pi) / np.float32(2))
    assert _maxdiff(pulse[..., 1].cpu().numpy(), sub) <= 2e-6


@pytest.mark.parametrize("overrides", [
    {"mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 1], "mbexwn_config:pp_mod_subnet_channel_factors": [1, 0.5],
     "mbexwn_config:pulse_channels": 10, "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5,
     "mbexwn_config:pp_mod_subnet:n_channels": 64, "mbexwn_config:pp_mod_subnet:n_layers": 3},
    {"mbexwn_config:pp_mod_subnet_upsampling_factors": [2], "mbexwn_config:pp_mod_subnet_channel_factors": [1],
     "mbexwn_config:pulse_channels": 10, "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 10,
     "mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 2},
    {"mbexwn_config:pp_mod_subnet_upsampling_factors": [2, 2, 1], "mbexwn_config:pp_mod_subnet_channel_factors": [1, 0.75, 0.5],
     "mbexwn_config:pulse_channels": 20, "mbexwn_config:pp_mod_subnet:cond_lin_upsampling": 5,
     "mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 2,
     "mbexwn_config:pp_mod_subnet:pre_cond_layer_channels": [24], "mbexwn_config:pp_mod_subnet:activation": "gfu"},
    {"mbexwn_config:pp_mod_subnet_upsampling_factors": [1, 1], "mbexwn_config:pp_mod_subnet_channel_factors": [1, 2],
     "mbexwn_config:pp_mod_subnet:n_channels": 16, "mbexwn_config:pp_mod_subnet:n_layers": 2,
     "mbexwn_config:pp_mod_subnet:disable_conditioning": True},
], ids=["2blocks_C64", "1block_up2", "3blocks_precond_gfu", "2blocks_noup_nocond"])
def test_several_wavenet_blocks(torch, overrides):
    """pp_mod_subnet_upsampling_factors / _channel_factors (reference custom_pulsed_generator.py:456-488): stacked WaveNet
    blocks with sub-pixel up-sampling convolutions in between, each block with its own conditioning chain; ragged batch
    against the oracle (which the golden case "blocks" pins to the reference).  Such a handle does not stream."""
    from mbexwn_vocoder_amd.config import ModelDims
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    from mbexwn_vocoder_amd.streaming import StreamingSynthesizer
    cfg, raw, wt = build_case("SPEECH", overrides)
    dims = ModelDims(cfg)
    assert dims.wn_multi and dims.wn_in_rows_per_frame * int(np.prod(dims.wn_block_ups)) == 20
    eng, om = MBExWNEngine(cfg, raw, wt), orc.OracleModel(cfg, raw, wt)
    rpf = dims.wn_in_rows_per_frame
    mel, noise = synthetic_inputs(17, 2, 23, steps_per_frame=rpf)
    lengths = (23, 7)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise),
                      n_frames=torch.tensor(lengths, dtype=torch.int32, device="cuda")).cpu().numpy()
    for ii, ll in enumerate(lengths):
        ref = om.forward(mel[ii:ii + 1, :ll], noise[ii:ii + 1, :ll * rpf])[0]
        assert _maxdiff(got[ii, :ll * 300], ref) <= _tol(ref, E2E_TOL)
        assert np.all(got[ii, ll * 300:] == 0.0)
    assert eng.layer_state_info()[0] == 0
    syn = StreamingSynthesizer(eng, chunk_frames=8)
    syn.open(0)
    with pytest.raises((NotImplementedError, ValueError)):
        syn.push(0, mel[0], noise[0], last=True)
        for _ in range(4):
            syn.tick()


def test_item_longer_than_4_gib_of_activation_rows(torch, monkeypatch):
    """One utterance of 41 minutes: its (rows, C) activation tensors are 5.1 GiB, so byte offsets from the item's first
    row do not fit 32 bits.  The LDS-DMA kernels address rows relative to their own block; the result must agree with the
    direct-form path (which addresses with 64-bit pointers) over the whole length, the last seconds included, and its
    first second with the oracle."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*CANON)
    T = 200_000
    rng = np.random.default_rng(8)
    base_mel, base_noise = synthetic_inputs(8, 1, 500)
    mel = torch.as_tensor(np.tile(base_mel, (1, T // 500, 1))).cuda()
    noise = torch.as_tensor(rng.normal(size=(1, T * 20)).astype(np.float32)).cuda()
    assert T * 20 * 320 * 4 > 2 ** 32
    outs = {}
    for form in ("4", "0"):
        eng = MBExWNEngine(cfg, raw, wt, **form_kwargs(form))
        outs[form] = eng.forward(mel, noise=noise)
        torch.cuda.synchronize()
        del eng
    amp = float(outs["0"].abs().max())
    diff = (outs["4"] - outs["0"]).abs()
    assert float(diff.max()) <= 5e-5 * max(1.0, amp), float(diff.max())
    assert float(diff[:, -24000 * 30:].max()) <= 5e-5 * max(1.0, amp)          # the last 30 s live above the 4 GiB mark
    assert float(outs["4"][:, -24000:].abs().max()) > 0.05
    ref = orc.OracleModel(cfg, raw, wt).forward(mel[:, :80].cpu().numpy(), noise[:, :1600].cpu().numpy())
    head = outs["4"][:, :68 * 300].cpu().numpy()
    assert _maxdiff(head, ref[:, :68 * 300]) <= _tol(ref, E2E_TOL)
    # the documented limit: 2^24 sub-band rows per item (2.9 h), refused before anything is allocated
    eng = MBExWNEngine(cfg, raw, wt)
    with pytest.raises(NotImplementedError, match="2\\^24"):
        eng.forward(torch.zeros((1, (1 << 24) // 20 + 1, 80), device="cuda"), noise=torch.zeros((1, 8), device="cuda"))


@pytest.mark.parametrize("form", ["4", "2"])
def test_empty_items_inside_a_batch(torch, monkeypatch, form):
    """Items of zero frames (and of one frame) between ordinary ones: they produce zeros and leave their neighbours alone."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*CANON)
    eng = MBExWNEngine(cfg, raw, wt, **form_kwargs(form))
    mel, noise = synthetic_inputs(3, 4, 30)
    nf = torch.tensor([30, 0, 11, 1], dtype=torch.int32, device="cuda")
    got = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    assert np.all(np.isfinite(got)) and np.all(got[1] == 0.0)
    for ii, ll in ((0, 30), (2, 11), (3, 1)):
        one = eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()[0]
        assert _maxdiff(got[ii, :ll * 300], one) <= 2e-5 * max(1.0, np.abs(one).max())
        assert np.all(got[ii, ll * 300:] == 0.0)


def test_engine_without_weight_images_runs_the_generic_kernels(torch):
    """A handle created from the folded weights and tables alone (no operand-order images) must give the same audio
    through the generic convolution kernels (direct gate, C->2C res/skip with the skip tensor, separate end/post)."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*CANON)
    eng = MBExWNEngine(cfg, raw, wt, weight_images=False)
    assert not any(kk.endswith((".wino2w", ".wino4w", ".packed", ".fold")) for kk in eng._tensors)
    om = get_engine("canon", *CANON)[1]
    mel, noise = synthetic_inputs(9, 2, 25)
    got = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    ref = om.forward(mel, noise)
    assert _maxdiff(got, ref) <= _tol(ref, E2E_TOL)
    assert eng.stage("wn_skip").shape[-1] == 25 * 20 * eng.dims.wn_channels      # the skip tensor exists on this path


@pytest.mark.parametrize("batch", [16, 1])
def test_full_size_gate_forms_agree(torch, monkeypatch, batch):
    """BASELINE config 3 / config 2 sizes (16 x 10 s, 1 x 10 s): the default picks the F(4,3) kernel with 256-row
    blocks / with product-split 128-row blocks; its result must agree with the F(2,3) form of the same engine to
    float32 rounding (size-independent property: two algebraically identical evaluations)."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(*CANON)
    mel, noise = synthetic_inputs(21, batch, 800)
    outs = {}
    for form in ("4", "2"):
        eng = MBExWNEngine(cfg, raw, wt, **form_kwargs(form))
        outs[form] = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
        del eng
    assert np.all(np.isfinite(outs["4"]))
    assert _maxdiff(outs["4"], outs["2"]) <= 2e-5 * max(1.0, float(np.abs(outs["2"]).max()))
    assert not np.array_equal(outs["4"], outs["2"])       # the default really took the other kernel


@pytest.mark.parametrize("voice", ["SPEECH", "VOICE"])
def test_f43_block_shapes_give_the_same_bits(torch, monkeypatch, voice):
    """The two block shapes of the F(4,3) gate kernel (mbx_config.tune_gate_shape pins the one small launches take): 256-row blocks and
    128-row blocks whose waves split the six PRODUCTS form every sum in the same order -> identical audio.  C = 320 and
    C = 340 (partial column tile, partial last slice), ragged batch of two with an item that ends inside a block; the
    short item is held to the oracle."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case(voice, {})
    lengths = [240, 133]
    mel, noise = synthetic_inputs(77, 2, 240)
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    outs = {}
    for shape in ("0", "1", "2"):
        eng = MBExWNEngine(cfg, raw, wt, conv_form="f43", tune={"gate_shape": 1 + int(shape)})
        assert eng.gate_form(2, 240) == {"0": "winograd_f43", "1": "winograd_f43_psplit", "2": "winograd_f43_hsplit"}[shape]
        outs[shape] = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
        del eng
    assert np.array_equal(outs["0"], outs["1"]), "the product-split blocks must give the bits of the 256-row blocks"
    assert np.array_equal(outs["0"], outs["2"]), "... and so must the product-split blocks of half a column tile"
    ref = orc.OracleModel(cfg, raw, wt).forward(mel[1:2, :133], noise[1:2, :133 * 20])[0]
    assert _maxdiff(outs["1"][1, :133 * 300], ref) <= _tol(ref, E2E_TOL)
    # the default policy picks one of the two equivalent shapes by how the work divides over the SIMDs
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f43")
    assert eng.gate_form(2, 240) in ("winograd_f43", "winograd_f43_psplit", "winograd_f43_hsplit")
    assert np.array_equal(eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy(), outs["1"])


def test_padded_batch_equals_one_at_a_time(torch):
    """Every boundary op honours the item's own length: a ragged batch gives the per-utterance results
    (the reference processes one utterance at a time, bin/resynth_mel.py:74)."""
    eng = get_engine("small", *SMALL)[0]
    lengths = [23, 7, 0, 1, 16]                      # ragged, including an empty item
    T = max(lengths)
    mel, noise = synthetic_inputs(7, len(lengths), T)
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    batch_out = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
    for ii, ll in enumerate(lengths):
        if ll == 0:
            assert np.all(batch_out[ii] == 0.0)
            continue
        single = eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()
        assert np.array_equal(batch_out[ii, :ll * 300], single[0]), f"item {ii} differs from its single run"
        assert np.all(batch_out[ii, ll * 300:] == 0.0)


def test_large_launch_equals_single_runs_under_batch_invariant(torch):
    """A launch of more than 4 096 mel frames (the mel-rate convolutions take their large-launch tile kernel, the bandwidth
    stages their full grids) under batch_invariant (one set of WaveNet kernels at every size): a ragged batch of 9 x 500 frames
    must give, bit for bit, what each item gives alone; three times, with other work queued on the stream in between."""
    from mbexwn_vocoder_amd.engine import MBExWNEngine
    cfg, raw, wt = build_case("SPEECH", {"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 2})
    eng = MBExWNEngine(cfg, raw, wt, conv_form="f43", batch_invariant=True)
    lengths = [500, 471, 500, 123, 500, 500, 1, 388, 500]
    mel, noise = synthetic_inputs(41, len(lengths), 500)
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    singles = [eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20])).cpu().numpy()[0]
               for ii, ll in enumerate(lengths)]
    for rep in range(3):
        eng.forward(dev(torch, mel[::-1].copy()), n_frames=nf, noise=dev(torch, noise[::-1].copy()))
        out = eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise)).cpu().numpy()
        for ii, ll in enumerate(lengths):
            assert np.array_equal(out[ii, :ll * 300], singles[ii]), f"run {rep}: item {ii} differs from its single run"
            assert np.all(out[ii, ll * 300:] == 0.0)
    ref = orc.OracleModel(cfg, raw, wt).forward(mel[3:4, :123], noise[3:4, :123 * 20])[0]
    assert _maxdiff(out[3, :123 * 300], ref) <= _tol(ref, E2E_TOL)


def test_mel_rate_group_launch_large_ragged_equals_small_launches_stage_by_stage(torch):
    """the shared mel-rate launches at a large launch size (conv1d_mel_group_kernel: 64 x 128 float32 tiles and
    32 x 32 float64 tiles, three members per launch -- F0-net in float64, VTF-net and conditioning convolution in float32 --,
    ragged n_frames, 6 x 900 = 5400 frames >= the 4096-frame switch) against the small-launch group kernel that each item takes
    alone: the F0 contour, the cepstrum and the conditioning rows agree bit for bit (one summation order at every launch size).
    (ISA check of the same commit: conv1d_mel_group_kernel 141 VGPRs, no scratch, 3 blocks per CU at 48 KB of LDS.)"""
    eng = get_engine("canon", *CANON)[0]
    lengths = [900, 611, 900, 37, 768, 899]
    mel, noise = synthetic_inputs(91, len(lengths), 900)
    nf = torch.as_tensor(lengths, dtype=torch.int32).cuda()
    eng.forward(dev(torch, mel), n_frames=nf, noise=dev(torch, noise))
    big = {kk: eng.stage(kk).cpu().numpy() for kk in ("f0", "cepstrum", "cond")}
    per_frame = {"f0": eng.dims.pulse_per_frame, "cepstrum": eng.dims.n_ceps, "cond": eng.dims.cond_conv_upsampling * 2 * eng.dims.wn_channels}
    for ii in (0, 1, 3, 5):
        ll = lengths[ii]
        eng.forward(dev(torch, mel[ii:ii + 1, :ll]), noise=dev(torch, noise[ii:ii + 1, :ll * 20]))
        for kk, pf in per_frame.items():
            single = eng.stage(kk).cpu().numpy()[0]
            assert np.array_equal(big[kk][ii, :ll * pf], single[:ll * pf]), f"stage {kk} of item {ii}"


@pytest.mark.parametrize("batch,frames", [(2, 30), (1, 800), (16, 800)])
def test_deterministic(torch, batch, frames):
    """Same input, same bits -- also at the BASELINE sizes, where the large-launch kernels (F(4,3) with its three-stage
    ring, channel-split blocks meeting through LDS) run: a missing barrier or wait shows up as run-to-run noise."""
    eng = get_engine("canon", *CANON)[0]
    mel, noise = synthetic_inputs(3, batch, frames)
    a = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    b = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()
    assert np.array_equal(a, b)


def test_full_size_prefix_property(torch):
    """BASELINE config 2 size (10 s, T=800): size-independent property instead of a full oracle run --
    the graph has a finite receptive field and a causal phase accumulator, so the first part of the
    audio of a long utterance equals the audio of its prefix (away from the cut)."""
    eng = get_engine("canon", *CANON)[0]
    mel, noise = synthetic_inputs(11, 1, 800)
    full = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()[0]
    assert full.shape == (240000,) and np.all(np.isfinite(full))
    cut = 200
    part = eng.forward(dev(torch, mel[:, :cut]), noise=dev(torch, noise[:, :cut * 20])).cpu().numpy()[0]
    margin = 12 * 300       # WaveNet RF (31 steps) + PQMF + 2 STFT hops + sub-net kernels + F0 smoother (3 frames)
    keep = cut * 300 - margin
    assert _maxdiff(full[:keep], part[:keep]) <= 1e-5 * max(1.0, np.abs(full).max())
    # and the oracle agrees on the prefix
    om = get_engine("canon", *CANON)[1]
    ref = om.forward(mel[:, :40], noise[:, :800])[0]
    assert _maxdiff(full[:40 * 300 - margin], ref[:40 * 300 - margin]) <= _tol(ref, E2E_TOL)


def test_one_minute_utterance(torch):
    """A 60 s utterance (4800 frames, 96 000 WaveNet steps, 1.44 M samples) in one call: finite, and its first seconds
    equal the synthesis of the 5 s prefix up to float32 rounding (the long launch runs the large-launch kernels, the
    prefix the small-launch ones)."""
    eng = get_engine("canon", *CANON)[0]
    mel, noise = synthetic_inputs(77, 1, 4800)
    full = eng.forward(dev(torch, mel), noise=dev(torch, noise)).cpu().numpy()[0]
    assert full.shape == (4800 * 300,) and np.all(np.isfinite(full)) and np.abs(full).max() > 0.1
    cut = 400
    part = eng.forward(dev(torch, mel[:, :cut]), noise=dev(torch, noise[:, :cut * 20])).cpu().numpy()[0]
    keep = (cut - 12) * 300
    assert _maxdiff(full[:keep], part[:keep]) <= 2e-5 * max(1.0, np.abs(full).max())


def test_error_paths(torch):
    eng = get_engine("small", *SMALL)[0]
    mel, noise = synthetic_inputs(1, 1, 4)
    with pytest.raises(ValueError):
        eng.forward(dev(torch, mel))                                   # noise missing
    with pytest.raises(ValueError):
        eng.forward(dev(torch, mel[:, :, :40]), noise=dev(torch, noise))  # wrong channel count
    out = eng.forward(dev(torch, mel[:0]), noise=dev(torch, noise[:0]))
    assert out.shape == (0, 1200)


def test_infer_components_and_transposition(torch):
    """F0 / excitation / envelope outputs, external and transposed F0."""
    eng, om = get_engine("small", *SMALL)[:2]
    mel, noise = synthetic_inputs(31, 1, 18)
    f0, exc, env, rms = eng.infer_components(mel, noise=noise)
    ref_audio, st = om.forward(mel, noise, return_stages=True)
    assert rms is None and f0.shape == (1, 1800) and exc.shape == (1, 5400) and env.shape == (1, 18, 1025)
    assert _maxdiff(f0, st["f0"]) <= 1e-3
    assert _maxdiff(exc, st["excitation"]) <= _tol(st["excitation"], E2E_TOL)
    assert _maxdiff(env, st["envelope"]) <= _tol(np.abs(st["envelope"]), 2e-4)
    assert _maxdiff(eng.last_audio.cpu().numpy(), ref_audio) <= _tol(ref_audio, E2E_TOL)
    # transposed: the reference multiplies the contour, then excitation and envelope follow the new contour
    f0_t, exc_t, env_t, _ = eng.infer_components(mel, noise=noise, transposition_factor=1.5)
    np.testing.assert_allclose(f0_t, 1.5 * f0, rtol=1e-6)
    f0_ref = 1.5 * st["f0"]
    exc_ref = om.generate_excitation(mel.astype(np.float64), f0_ref, noise)
    assert _maxdiff(exc_t, exc_ref) <= _tol(exc_ref, E2E_TOL)
    # external contour
    contour = np.full((1, 1800), 220.0, dtype=np.float32)
    f0_e, exc_e, _, _ = eng.infer_components(mel, noise=noise, F0=contour)
    assert np.array_equal(f0_e, contour)
    exc_ref = om.generate_excitation(mel.astype(np.float64), contour.astype(np.float64), noise)
    assert _maxdiff(exc_e, exc_ref) <= _tol(exc_ref, E2E_TOL)


def test_infer_accepts_the_reference_arguments(torch):
    """PaNWaveNet.infer(spect, sigma, z_in, synth_length, F0, return_F0, return_components, ...) (reference
    wavegen_1d.py:483-526): sigma and z_in are inert; F0 is only used by the training branch of MBExWN.call
    (custom_pulsed_generator.py:640-663), so at inference it changes nothing; synth_length = 0 means the model's
    segment_length (:489); return_F0 / return_components add the parameter list / return the list of signals
    (:512-526, custom_pulsed_generator.py:756-771); the training-side switches raise."""
    eng, om = get_engine("small", *SMALL)[:2]
    mel, noise = synthetic_inputs(31, 1, 9)
    base = eng.infer(mel, synth_length=9 * 300, noise=noise).numpy()
    ref, st = om.forward(mel, noise, return_stages=True)
    assert _maxdiff(base, ref) <= _tol(ref, E2E_TOL)
    assert np.array_equal(eng.infer(mel, sigma=0.1, z_in=None, synth_length=9 * 300, noise=noise).numpy(), base)
    f0_net = eng.stage("f0").cpu().numpy()
    ignored = eng.infer(mel, synth_length=9 * 300, F0=np.full_like(f0_net, 220.0), noise=noise).numpy()
    assert np.array_equal(ignored, base)                       # as in the reference: F0 is a training-only input
    # synth_length: shorter cuts, 0 = segment_length (24000 > 9 frames: the last mel frame is repeated once, :490-491)
    assert np.array_equal(eng.infer(mel, synth_length=1000, noise=noise).numpy(), base[:, :1000])
    noise10 = np.concatenate((noise, noise[:, -20:]), axis=1)
    seg = eng.infer(mel, noise=noise10).numpy()
    mel10 = np.concatenate((mel, mel[:, -1:]), axis=1)
    assert seg.shape == (1, 10 * 300) and _maxdiff(seg, om.forward(mel10, noise10)) <= _tol(ref, E2E_TOL)
    # return_F0 / return_components
    sig, params = eng.infer(mel, synth_length=9 * 300, noise=noise, return_F0=True)
    assert np.array_equal(sig.numpy(), base) and [pp[0] for pp in params] == ["F0", "PSig", "PS"]
    assert _maxdiff(params[0][1].numpy(), st["f0"][:, ::3]) <= 1e-3            # the reference's own slice [:, :len:3]
    assert _maxdiff(params[1][1].numpy(), st["excitation"]) <= _tol(st["excitation"], E2E_TOL)
    assert all(hasattr(pp[1], "numpy") for pp in params)          # every entry is tensor-like, as in the reference
    assert params[2][1].shape == (1, 9, 1025) and _maxdiff(params[2][1].numpy(), np.abs(st["envelope"])) <= _tol(np.abs(st["envelope"]), 2e-4)
    sigs = eng.infer(mel, synth_length=9 * 300, noise=noise, return_components=True)
    assert isinstance(sigs, list) and len(sigs) == 1 and np.array_equal(sigs[0].numpy(), base)
    sigs, params = eng.infer(mel, synth_length=9 * 300, noise=noise, return_components=True, return_F0=True)
    assert isinstance(sigs, list) and len(params) == 3
    for kwargs in ({"training": True}, {"test_grad": 1}):
        with pytest.raises(NotImplementedError):
            eng.infer(mel, **kwargs)
    with pytest.raises(ValueError):                            # region arguments belong to a streaming window
        eng.forward(dev(torch, mel), noise=dev(torch, noise), active=(0, torch.as_tensor([9], dtype=torch.int32).cuda()))


def test_forward_is_graph_capturable(torch):
    """mbx_forward only enqueues kernels (no allocation, no synchronisation): it can be captured into a hipGraph."""
    eng = get_engine("small", *SMALL)[0]
    mel, noise = synthetic_inputs(17, 2, 12)
    mel_d, noise_d = dev(torch, mel), dev(torch, noise)
    out = torch.empty((2, 12 * 300), dtype=torch.float32, device="cuda")
    eager = eng.forward(mel_d, noise=noise_d).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.forward(mel_d, noise=noise_d, out=out)          # warm-up on the capture stream (workspace allocation)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.forward(mel_d, noise=noise_d, out=out)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    mel2, noise2 = synthetic_inputs(18, 2, 12)
    mel_d.copy_(dev(torch, mel2))
    noise_d.copy_(dev(torch, noise2))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eng.forward(mel_d, noise=noise_d))
