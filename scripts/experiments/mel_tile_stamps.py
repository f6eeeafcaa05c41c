"""Cycle account of a wave of conv1d_mel_tile_dma (csrc/conv_mfma.hip) from in-kernel s_memtime stamps (mkexp.py m2_stamp):
    EXP_FILE=conv_mfma.hip python scripts/experiments/mkexp.py m2_stamp:m2_stamp
    gpurun -- 'MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_m2_stamp.so python scripts/experiments/mel_tile_stamps.py [ks cin cout]'
One convolution of 16 x 800 rows as a launch of its own (default: PS_1, 3 x 256 -> 256)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from mbexwn_vocoder_amd import engine
    ks, cin, cout = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (3, 256, 256)
    cfg, raw, wt, dims, eng = bench.build_engine("SING")
    lib = engine.load_library()
    lib.mbx_exp_stamps.restype = ctypes.c_int
    lib.mbx_exp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((16, 800, cin), device="cuda", generator=g)
    w = torch.randn((ks, cin, cout), device="cuda", generator=g) * 0.05
    b = torch.zeros(cout, device="cuda")
    for _ in range(5):
        eng.conv1d(x, w, b, pad_l=ks // 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.conv1d(x, w, b, pad_l=ks // 2)
    e1.record()
    torch.cuda.synchronize()
    n_blocks = (16 * 800 // 64) * ((cout + 127) // 128)
    buf = np.zeros((8192, 4, 8), dtype=np.uint64)
    assert lib.mbx_exp_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    st = buf[:n_blocks].reshape(-1, 8).astype(np.int64)
    st = st[st[:, 3] > 0]
    t0, t1, tf, tg, hw = (st[:, i] for i in range(5))
    base = t0.min()
    groups = ks * cin // 8
    print(f"conv {ks} x {cin} -> {cout}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (events), {n_blocks} blocks, {st.shape[0]} waves stamped; "
          f"matrix-pipe cycles of a wave {groups * 8 * 64}")
    print(f"  first start -> last end {tg.max() - base} cycles; prologue median {np.median(t1 - t0):.0f}, K loop median {np.median(tf - t1):.0f} "
          f"(p10 {np.percentile(tf - t1, 10):.0f}, p90 {np.percentile(tf - t1, 90):.0f}), epilogue median {np.median(tg - tf):.0f}")
    # HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID in the upper word
    cu = ((hw >> 32) & 15) << 16 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 11 | ((hw >> 8) & 15) << 4
    per_cu = {}
    for key in np.unique(cu):
        sel = cu == key
        per_cu.setdefault(int(sel.sum()) // 4, []).append((t0[sel].min() - base, tg[sel].max() - base, np.median((tf - t1)[sel])))
    for nb in sorted(per_cu):
        arr = np.array(per_cu[nb], dtype=np.float64)
        print(f"  CUs with {nb} blocks: {len(arr):3d}; first start median {np.median(arr[:, 0]):8.0f}, last end median {np.median(arr[:, 1]):8.0f} "
              f"max {arr[:, 1].max():8.0f}; K loop of a wave median {np.median(arr[:, 2]):8.0f} cycles")


if __name__ == "__main__":
    main()
