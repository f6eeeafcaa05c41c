"""Host-side weight images consumed by the HIP kernels (engine.pack_*, engine.fold_skip_weights): every image is
unpacked here with the index formula of the kernel that reads it and fed through a float64 numpy restatement of that
kernel's arithmetic, which must reproduce the plain convolution."""
import numpy as np
import pytest

from mbexwn_vocoder_amd import engine


def _direct_dilated(x, w, d):
    """y[t] = x[t-d] W0 + x[t] W1 + x[t+d] W2 with zero padding; x (T, C), w (3, C, N)."""
    T = x.shape[0]
    xp = np.concatenate((np.zeros((d, x.shape[1])), x, np.zeros((d, x.shape[1]))))
    return xp[0:T] @ w[0] + xp[d:d + T] @ w[1] + xp[2 * d:2 * d + T] @ w[2]


def _unpack_gate_16x16x4(packed, C, n_prod):
    """Image of engine.pack_winograd2w_weights / pack_winograd4w_weights -> U (n_prod, C, 2C) with the lane/step map of the
    gate kernels on v_mfma_f32_16x16x4_f32: [product j][parity e][lane = 16 kq + n][tanh step 0, 1, sigmoid step 0, 1]."""
    nt, nk, _ = packed.shape
    img = packed.reshape(nt, nk, n_prod, 2, 64, 2, 2)               # tile, slice, j, e, lane, tanh|sigmoid, step
    U = np.zeros((n_prod, C, 2 * C))
    for tile in range(nt):
        for kt in range(nk):
            for e in range(2):
                for lane in range(64):
                    kq, n = lane >> 4, lane & 15
                    ch = 32 * tile + 2 * n + e
                    for ts in range(2):
                        for m in range(2):
                            k = 8 * kt + 2 * kq + m
                            if k < C and ch < C:
                                U[:, k, ts * C + ch] = img[tile, kt, :, e, lane, ts, m]
                            else:
                                assert np.all(img[tile, kt, :, e, lane, ts, m] == 0.0)
    return U


@pytest.mark.parametrize("C", [32, 40])
def test_winograd_f23_image_reproduces_the_convolution(C):
    """wn_gate_winograd2w_kernel: pairs (t, t + d) inside blocks of 2 d rows counted from the first computed row."""
    rng = np.random.default_rng(C)
    w = rng.normal(size=(3, C, 2 * C))
    packed = engine.pack_winograd2w_weights(w)
    assert packed.shape == ((C + 31) // 32, (C + 7) // 8, 2048)
    U = _unpack_gate_16x16x4(packed.astype(np.float64), C, 4)
    x = rng.normal(size=(64, C))
    for d in (1, 2, 8, 16):
        ref = _direct_dilated(x, w, d)
        xp = np.concatenate((np.zeros((d, C)), x, np.zeros((2 * d, C))))
        got = np.zeros_like(ref)
        for t0 in range(0, 64, 2 * d):
            for r in range(d):
                t = t0 + r
                x0, x1, x2, x3 = (xp[t + i * d] for i in range(4))           # h[t-d], h[t], h[t+d], h[t+2d]
                m1, m2, m3, m4 = (x0 - x2) @ U[0], (x1 + x2) @ U[1], (x2 - x1) @ U[2], (x1 - x3) @ U[3]
                got[t], got[t + d] = m1 + m2 + m3, m2 - m3 - m4
        assert np.max(np.abs(got - ref)) < 2e-6 * np.max(np.abs(ref))     # float32 storage of the combinations


def test_winograd2w_staging_map_covers_every_row_once():
    """The read-order staging of wn_winograd2w.hip: cell p = (m & 1) * 32 + (m >> 1) * d + b holds row m0 - d + d m + b;
    pair P reads cells (i & 1) * 32 + P + (i >> 1) * d, which must be the rows t - d, t, t + d, t + 2 d of the pair, and
    the DMA's inverse map (cell -> row) must agree."""
    for log2d in range(5):
        d = 1 << log2d
        for P in range(16):
            t = 2 * d * (P >> log2d) + (P & (d - 1))
            for i in range(4):
                cell = (i & 1) * 32 + P + ((i >> 1) << log2d)
                phase, sidx = cell // 32, cell % 32
                assert sidx < 16 + d
                m = 2 * (sidx >> log2d) + phase
                assert -d + (m << log2d) + (sidx & (d - 1)) == t + (i - 1) * d


@pytest.mark.parametrize("C", [32, 40])
def test_winograd_f43_image_reproduces_the_convolution(C):
    rng = np.random.default_rng(C)
    w = rng.normal(size=(3, C, 2 * C))
    U = _unpack_gate_16x16x4(engine.pack_winograd4w_weights(w).astype(np.float64), C, 6)
    x = rng.normal(size=(128, C))
    for d in (1, 4, 16):
        ref = _direct_dilated(x, w, d)
        xp = np.concatenate((np.zeros((d, C)), x, np.zeros((4 * d, C))))
        got = np.zeros_like(ref)
        for t0 in range(0, 128, 4 * d):
            for r in range(d):
                t = t0 + r
                x0, x1, x2, x3, x4, x5 = (xp[t + i * d] for i in range(6))   # h[t-d] .. h[t+4d]
                v = [4 * x0 - 5 * x2 + x4, -4 * x1 - 4 * x2 + x3 + x4, 4 * x1 - 4 * x2 - x3 + x4,
                     -2 * x1 - x2 + 2 * x3 + x4, 2 * x1 - x2 - 2 * x3 + x4, 4 * x1 - 5 * x3 + x5]
                m = [v[j] @ U[j] for j in range(6)]
                got[t] = m[0] + m[1] + m[2] + m[3] + m[4]
                got[t + d] = m[1] - m[2] + 2 * (m[3] - m[4])
                got[t + 2 * d] = m[1] + m[2] + 4 * (m[3] + m[4])
                got[t + 3 * d] = m[1] - m[2] + 8 * (m[3] - m[4]) + m[5]
        assert np.max(np.abs(got - ref)) < 1e-5 * np.max(np.abs(ref))


def _unpack_resskip(packed, C, cout):
    nct, nk, _ = packed.shape
    img = packed.reshape(nct, nk, 2, 4, 64, 4)                      # tile, slice, cc, jn, lane, st
    W = np.zeros((C, cout))
    for tile in range(nct):
        for kt in range(nk):
            for cc in range(2):
                for jn in range(4):
                    for lane in range(64):
                        lk, n = lane >> 5, lane & 31
                        col = 128 * tile + 32 * jn + n
                        for st in range(4):
                            k = 16 * kt + 8 * cc + 4 * lk + st
                            if k < C and col < cout:
                                W[k, col] = img[tile, kt, cc, jn, lane, st]
                            else:
                                assert img[tile, kt, cc, jn, lane, st] == 0.0
    return W


def test_resskip_and_end_images_are_permutations_of_the_weights():
    rng = np.random.default_rng(3)
    C, cout = 40, 70
    w = rng.normal(size=(1, C, cout)).astype(np.float32)
    assert np.array_equal(_unpack_resskip(engine.pack_resskip_weights(w), C, cout), w[0].astype(np.float64))
    we = rng.normal(size=(1, 44, 30)).astype(np.float32)
    img = engine.pack_end_weights(we)                                 # [c][lk][n][st], channel 8c + 4lk + st
    assert img.shape == (6, 2, 32, 4)
    for c in range(6):
        for lk in range(2):
            for st in range(4):
                k = 8 * c + 4 * lk + st
                row = we[0, k] if k < 44 else np.zeros(30, np.float32)
                assert np.array_equal(img[c, lk, :30, st], row) and np.all(img[c, lk, 30:, st] == 0)


@pytest.mark.parametrize("layers", [1, 3])
def test_folded_skip_path_is_the_same_linear_map(layers):
    """end(sum_l (a_l Ws_l + bs_l)) == sum_l a_l (Ws_l We) + const, with the images the kernels read."""
    rng = np.random.default_rng(layers)
    C, n_out, T = 24, 30, 50
    folded = {"wn.end.w": rng.normal(size=(1, C, n_out)), "wn.end.b": rng.normal(size=n_out)}
    for ll in range(layers):
        cout = C if ll == layers - 1 else 2 * C
        folded[f"wn.res_skip_{ll}.w"] = rng.normal(size=(1, C, cout))
        folded[f"wn.res_skip_{ll}.b"] = rng.normal(size=cout)
    acts = [rng.normal(size=(T, C)) for _ in range(layers)]
    # un-folded graph (reference custom_AE_layers.py:322-341)
    skip = np.zeros((T, C))
    res_ref = []
    for ll in range(layers):
        r = acts[ll] @ folded[f"wn.res_skip_{ll}.w"][0] + folded[f"wn.res_skip_{ll}.b"]
        if ll < layers - 1:
            res_ref.append(r[:, :C])
            skip += r[:, C:]
        else:
            skip += r
    ref = skip @ folded["wn.end.w"][0] + folded["wn.end.b"]
    # folded graph as the kernels run it
    extra = engine.fold_skip_weights(folded, layers, C)
    y = np.zeros((T, n_out))
    for ll in range(layers - 1):
        W = _unpack_resskip(extra[f"wn.res_skip_{ll}.fold"].astype(np.float64), C, C + n_out)
        r = acts[ll] @ W + extra[f"wn.res_skip_{ll}.fold_b"]
        assert np.max(np.abs(r[:, :C] - res_ref[ll])) < 1e-5            # residual half untouched
        y = (0 if ll == 0 else y) + r[:, C:]
    tail = extra["wn.tail.fold"].astype(np.float64)                      # [c][lk][n][st]
    P = np.zeros((C, n_out))
    for k in range(C):
        P[k] = tail[k // 8, (k % 8) // 4, :n_out, k % 4]
    y = y + acts[-1] @ P + extra["wn.tail.fold_b"]
    assert np.max(np.abs(y - ref)) < 1e-5 * np.max(np.abs(ref))


def test_start_convolution_folds_into_layer_0_including_the_item_edges():
    """engine.fold_start_weights: the K = 24 contraction of x' = [x | 1 | 0] with the tap products Ws' W0_tau equals the
    dilated convolution of h0 = start(x) with zero "SAME" padding -- also in the first and last rows, where a tap falls
    outside the item (the constant channel is zero there, exactly where the reference pads h0) -- and [a0 | x'] times
    the K-extended res/skip image equals h0 + a0 Wr (+ the folded skip columns)."""
    from types import SimpleNamespace
    rng = np.random.default_rng(11)
    C, pc, n_out, T = 40, 5, 30, 37
    dims = SimpleNamespace(wn_channels=C, wn_layers=2, wn_in_channels=pc + 1, pulse_channels=pc, pulse_channels_eff=pc, wn_out_channels=n_out)
    folded = {"wn.start.w": rng.normal(size=(1, pc + 1, C)), "wn.start.b": rng.normal(size=C),
              "wn.conv1D_0.w": rng.normal(size=(3, C, 2 * C)),
              "wn.res_skip_0.w": rng.normal(size=(1, C, 2 * C)), "wn.res_skip_0.b": rng.normal(size=2 * C)}
    proj = rng.normal(size=(C, n_out))
    out = engine.fold_start_weights(folded, dims, {"wn.res_skip_0.fold": None, "__proj_0": proj})
    img = out["wn.conv1D_0.start_fold"].astype(np.float64)
    nt = img.shape[0]
    img = img.reshape(nt, 3, 2, 64, 2, 2)                          # tile, tap, e, lane, tanh|sigmoid, step
    P = np.zeros((3, 8, 2 * C))
    for tile in range(nt):
        for e in range(2):
            for lane in range(64):
                kq, n = lane >> 4, lane & 15
                ch = 32 * tile + 2 * n + e
                for ts in range(2):
                    for m in range(2):
                        if ch < C:
                            P[:, 2 * kq + m, ts * C + ch] = img[tile, :, e, lane, ts, m]
                        else:
                            assert np.all(img[tile, :, e, lane, ts, m] == 0.0)
    x = rng.normal(size=(T, pc + 1))
    xp = np.concatenate((x, np.ones((T, 1)), np.zeros((T, 1))), axis=1)          # x' (T, 8)
    h0 = x @ folded["wn.start.w"][0] + folded["wn.start.b"]
    for d in (1, 4):
        ref = _direct_dilated(h0, folded["wn.conv1D_0.w"], d)
        xpp = np.concatenate((np.zeros((d, 8)), xp, np.zeros((d, 8))))
        got = sum(xpp[tau * d:tau * d + T] @ P[tau] for tau in range(3))
        assert np.max(np.abs(got - ref)) < 2e-6 * np.max(np.abs(ref))
    W = _unpack_resskip(out["wn.res_skip_0.fold_start"], C + 16, C + n_out)
    a0 = rng.normal(size=(T, C))
    rows = np.concatenate((a0, xp, np.zeros((T, 8))), axis=1)
    ref = np.concatenate((h0 + a0 @ folded["wn.res_skip_0.w"][0][:, :C], a0 @ proj), axis=1)
    assert np.max(np.abs(rows @ W - ref)) < 2e-6 * np.max(np.abs(ref))


def test_resskip_wide_image_is_a_permutation_of_the_weights():
    """engine.pack_resskip_wide_weights unpacked with the lane/step map of wn_resskip_wide_kernel."""
    rng = np.random.default_rng(5)
    K, cout = 44, 70
    w = rng.normal(size=(1, K, cout)).astype(np.float32)
    img = engine.pack_resskip_wide_weights(w)
    nk, npair = (K + 7) // 8, (cout + 31) // 32
    assert img.shape == (nk, npair, 256)
    img = img.reshape(nk, npair, 64, 2, 2)                          # slice, pair, lane, parity, step
    W = np.zeros((K, cout), np.float32)
    for kt in range(nk):
        for pr in range(npair):
            for lane in range(64):
                kq, n = lane >> 4, lane & 15
                for par in range(2):
                    for m in range(2):
                        k, col = 8 * kt + 2 * kq + m, 32 * pr + 2 * n + par
                        if k < K and col < cout:
                            W[k, col] = img[kt, pr, lane, par, m]
                        else:
                            assert img[kt, pr, lane, par, m] == 0.0
    assert np.array_equal(W, w[0])


def test_channel_groups_become_block_diagonal_dense_layers():
    """weights.merge_channel_groups: the oracle's grouped WaveNet (reference custom_AE_layers.py:303-340, per-group layers)
    against the same oracle run as ONE group on the merged dense weights -- the layout the HIP kernels consume."""
    import copy
    from mbexwn_vocoder_amd.config import ModelDims, canonical_config
    from mbexwn_vocoder_amd.tables import WaveTables
    from mbexwn_vocoder_amd.weights import fold_weights, merge_channel_groups, synthetic_weights
    from oracle.mbexwn_oracle import OracleModel, synthetic_mel
    cfg = canonical_config("SPEECH", **{"mbexwn_config:pp_mod_subnet:n_channels": 32, "mbexwn_config:pp_mod_subnet:n_layers": 3,
                                        "mbexwn_config:pp_mod_subnet:n_ch_groups": 2})
    dims = ModelDims(cfg)
    raw = synthetic_weights(cfg, seed=5, bias_std=0.05, alpha_jitter=0.05)
    assert "wn.conv1D_1g1.v" in raw and raw["wn.conv1D_1g1.v"].shape == (3, 16, 32)
    wt = WaveTables(sample_rate=dims.pulse_rate, **cfg["mbexwn_config"]["wavetable_config"])
    rng = np.random.default_rng(1)
    mel = synthetic_mel(rng, 1, 12)
    noise = rng.normal(size=(1, 12 * 20)).astype(np.float32)
    ref = OracleModel(cfg, raw, wt).forward(mel, noise)

    merged = merge_channel_groups(fold_weights(raw), dims)
    assert merged["wn.conv1D_1.w"].shape == (3, 32, 64) and "wn.conv1D_1g1.w" not in merged
    assert np.all(merged["wn.conv1D_1.w"][:, :16, 16:32] == 0) and np.all(merged["wn.conv1D_1.w"][:, 16:, :16] == 0)

    class Dense(OracleModel):                      # folded tensors taken as they are
        def weight(self, name):
            return (np.asarray(merged[name + ".w"], dtype=self.dtype), np.asarray(merged[name + ".b"], dtype=self.dtype))
    cfg1 = copy.deepcopy(cfg)
    cfg1["mbexwn_config"]["pp_mod_subnet"]["n_ch_groups"] = 1
    got = Dense(cfg1, raw, wt).forward(mel, noise)
    assert np.max(np.abs(got - ref)) < 5e-6 * max(1.0, np.max(np.abs(ref)))      # float32 storage of the folded weights


def test_fold_weights_refuses_a_missing_gain_where_the_reference_always_normalises():
    """only the WaveNet's own layers can be built without weight normalisation (reference
    custom_AE_layers.py:124); a sub-net / post-net layer without its gain, or a WaveNet layer without one in a model
    configured with use_weight_norm, must not load silently."""
    import pytest
    from mbexwn_vocoder_amd.weights import fold_weights
    rng = np.random.default_rng(0)

    def layer(name, with_g=True):
        dd = {name + ".v": rng.normal(size=(1, 4, 6)).astype(np.float32), name + ".bias": np.zeros(6, np.float32)}
        if with_g:
            dd[name + ".g"] = np.ones(6, np.float32)
        return dd

    raw = {**layer("wn.conv1D_0", with_g=False), **layer("post")}
    out = fold_weights(raw)                                    # unknown configuration: plain WaveNet kernels are accepted
    assert np.array_equal(out["wn.conv1D_0.w"], raw["wn.conv1D_0.v"])
    assert np.array_equal(fold_weights(raw, wavenet_weight_norm=False)["wn.conv1D_0.w"], raw["wn.conv1D_0.v"])
    with pytest.raises(KeyError):
        fold_weights(raw, wavenet_weight_norm=True)
    with pytest.raises(KeyError):
        fold_weights({**layer("wn.conv1D_0"), **layer("post", with_g=False)})
    with pytest.raises(KeyError):
        fold_weights({**layer("PS_Layer_final", with_g=False)}, wavenet_weight_norm=False)


def test_resskip_wave_image_is_a_permutation_of_the_weights():
    """wn_resskip_wave_kernel: lane (n = lane & 15, kq = lane >> 4) of pair p reads [even tile steps 0..3 | odd tile steps
    0..3]; step m contracts input channel 16 slice + 4 kq + m with output column 32 p + 2 n + parity."""
    rng = np.random.default_rng(11)
    K, cout = 40, 70
    w = rng.normal(size=(1, K, cout)).astype(np.float32)
    img = engine.pack_resskip_wave_weights(w)
    assert img.shape == ((K + 15) // 16, 12, 512)
    got = np.zeros((K, cout), dtype=np.float32)
    seen = 0
    for kt in range(img.shape[0]):
        for pp in range(12):
            for parity in range(2):
                for lane in range(64):
                    kq, n = lane >> 4, lane & 15
                    for st in range(4):
                        k, col = 16 * kt + 4 * kq + st, 32 * pp + 2 * n + parity
                        val = img[kt, pp, parity * 256 + lane * 4 + st]
                        if k < K and col < cout:
                            got[k, col] = val
                            seen += 1
                        else:
                            assert val == 0.0
    assert seen == K * cout and np.array_equal(got, w[0])


def test_equalized_lr_folds():
    """pp_mod_subnet.use_equalized_lr (reference conv_layers.py:133-153): with weight norm W = g v / sqrt(mean v^2) per
    output channel; without it the layer multiplies its output -- bias included -- by g.  Only WaveNet layers."""
    from mbexwn_vocoder_amd.weights import fold_weights
    rng = np.random.default_rng(2)
    raw = {}
    for name in ("wn.conv1D_0", "post"):
        raw[name + ".v"] = rng.normal(size=(3, 4, 6)).astype(np.float32)
        raw[name + ".g"] = rng.uniform(0.5, 2.0, size=6).astype(np.float32)
        raw[name + ".bias"] = rng.normal(size=6).astype(np.float32)
    v, g, b = (raw["wn.conv1D_0" + sfx].astype(np.float64) for sfx in (".v", ".g", ".bias"))
    normed = fold_weights(raw, wavenet_weight_norm=True, wavenet_equalized_lr=True)
    np.testing.assert_allclose(normed["wn.conv1D_0.w"], g * v / np.sqrt(np.mean(v * v, axis=(0, 1), keepdims=True)), rtol=2e-6)
    np.testing.assert_array_equal(normed["wn.conv1D_0.b"], raw["wn.conv1D_0.bias"])
    plain = fold_weights(raw, wavenet_weight_norm=False, wavenet_equalized_lr=True)
    np.testing.assert_allclose(plain["wn.conv1D_0.w"], g * v, rtol=2e-6)
    np.testing.assert_allclose(plain["wn.conv1D_0.b"], g * b, rtol=2e-6)
    # the post-net is not a WaveNetAE layer: weight norm as always
    ref = fold_weights(raw)["post.w"]
    assert np.array_equal(normed["post.w"], ref) and np.array_equal(plain["post.w"], ref)


def test_split_f16_weight_images():
    """The fp16-split weight images of the opt-in precision mode (csrc/wn_resskip_f16.hip, csrc/wn_gate_f16.hip): every
    image entry sits where the kernel's MFMA operand order expects it, hi + 2^-11 lo' reproduces the float32 weight to
    2^-21 of its magnitude, and out-of-range entries are zero."""
    from mbexwn_vocoder_amd.engine import pack_gate_f16_weights, pack_resskip_f16_weights
    rng = np.random.default_rng(11)
    # res/skip: (1, K, cout), K = 340 (partial last step), cout = 370 (12 pairs, the last one partial)
    K, cout = 340, 370
    w = (rng.normal(size=(1, K, cout)) * 0.07).astype(np.float32)
    img = pack_resskip_f16_weights(w)
    nk = (K + 31) // 32
    assert img.shape == (nk, 12, 1024) and img.dtype == np.float32
    halves = img.view(np.float16).reshape(nk, 12, 4, 64, 8)            # step, pair, [even hi, even lo, odd hi, odd lo], lane, v
    rebuilt = np.zeros((nk * 32, 384))
    for kq in range(4):
        for vv in range(8):
            chan = 4 * kq + (vv if vv < 4 else 12 + vv)               # 4 kq .. + 3 and 16 + 4 kq .. + 3
            for par in range(2):
                hi = halves[:, :, 2 * par, 16 * kq:16 * kq + 16, vv].astype(np.float64)       # (step, pair, n)
                lo = halves[:, :, 2 * par + 1, 16 * kq:16 * kq + 16, vv].astype(np.float64)
                val = hi + lo / 2048.0
                for step in range(nk):
                    rebuilt[32 * step + chan, par::2][:384 // 2] = val[step].reshape(-1)      # columns 32 p + 2 n + par
    assert np.max(np.abs(rebuilt[:K, :cout] - w[0])) <= 2.0 ** -21 * np.max(np.abs(w))
    assert np.all(rebuilt[K:] == 0) and np.all(rebuilt[:, cout:] == 0)
    # gate: (3, C, 2C), C = 68 (three column tile blocks, the last one partial; three steps)
    C = 68
    wg = (rng.normal(size=(3, C, 2 * C)) * 0.05).astype(np.float32)
    gi = pack_gate_f16_weights(wg)
    nt = (C + 31) // 32
    assert gi.shape == (nt, nt, 6144)
    gh = gi.view(np.float16).reshape(nt, nt, 3, 2, 2, 2, 4, 16, 8)     # block, step, tap, e, s, part, kq, n, v
    val = gh[:, :, :, :, :, 0].astype(np.float64) + gh[:, :, :, :, :, 1].astype(np.float64) / 2048.0
    back = np.zeros((3, nt * 32, 2, nt * 32))                          # tap, channel, s, gate channel
    for blk in range(nt):
        for step in range(nt):
            # val[blk, step]: (tap, e, s, kq, n, v) -> channel 32 step + 8 kq + v, gate channel 32 blk + 2 n + e
            vv = val[blk, step].transpose(0, 2, 3, 5, 4, 1)            # tap, s, kq, v, n, e
            back[:, 32 * step:32 * step + 32, :, 32 * blk:32 * blk + 32] = \
                vv.reshape(3, 2, 32, 32).transpose(0, 2, 1, 3)
    want = np.zeros_like(back)
    want[:, :C, 0, :C], want[:, :C, 1, :C] = wg[:, :, :C], wg[:, :, C:]
    assert np.max(np.abs(back - want)) <= 2.0 ** -21 * np.max(np.abs(wg))
    with pytest.raises(ValueError, match="fp16"):
        pack_gate_f16_weights(wg * 1e6)
