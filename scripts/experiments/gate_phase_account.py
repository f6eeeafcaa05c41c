#!/usr/bin/env python3
"""Round 5: per-phase cycle account of the F(4,3) gate kernel from in-kernel stamps.

Runs config 3 (16 x 10 s) through an instrumented build of csrc/wn_winograd4w.hip (scripts/experiments/mkexp.py stampA /
stampA+stampB: s_memtime at kernel start, first stage landed, K loop done, end; with stampB also the cycles a wave spends
between "products 0..4 of a slice issued" and "barrier of the slice passed", summed over the slices) and reads the stamps of
the last gate launch of a forward back through the variant library's export mbx_exp_stamps.

    python scripts/experiments/mkexp.py stampA:stampA stampB:stampA+stampB
    gpurun -- 'MBX_LIB_PATH=$PWD/scripts/experiments/libs/lib_stampA.so python scripts/experiments/gate_phase_account.py'

Prints: the shader clock during the launch, the phases of a block (median / p10 / p90 over all waves, cycles and us), and per
SIMD the share of the launch in which at least one / none of its resident waves is inside its K loop (= the time the matrix
pipe of that SIMD has no MFMA to execute whatever the issue rate: prologue / epilogue exposure)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from mbexwn_vocoder_amd import engine
    batch, frames = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 800)
    cfg, raw, wt, dims, eng = bench.build_engine("SING")
    lib = engine.load_library()
    lib.mbx_exp_stamps.restype = ctypes.c_int
    lib.mbx_exp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    mel_h, noise_h = bench.synthetic_batch(np.random.default_rng(1), batch, frames, dims.steps_per_frame)
    mel, noise = torch.as_tensor(mel_h).cuda(), torch.as_tensor(noise_h).cuda()
    for _ in range(12):
        eng.forward(mel, noise=noise)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for _ in range(5):
        eng.forward(mel, noise=noise)
    torch.cuda.synchronize()
    stage = os.environ.get("STAMP_STAGE", "gate")                 # "res_skip": stamps of wn_resskip_wide_kernel (mkexp.py rwstamp)
    gms, gcnt = eng.profile_read(stage)
    eng.profile_enable(False)
    buf = np.zeros((16384, 4, 8) if stage == "gate" else (8192, 8, 8), dtype=np.uint64)
    rc = lib.mbx_exp_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0, rc
    ok = buf[:, :, 3] != 0
    st = buf[ok].astype(np.int64)                       # (waves, 8)
    n_blocks = int(ok[:, 0].sum())
    ts0, ts1, ts2, ts3, tr0, tr3, tbar, hw = (st[:, ii] for ii in range(8))
    clock_mhz = float((ts3 - ts0).sum()) / float((tr3 - tr0).sum()) * 100.0
    launch_us = float(tr3.max() - tr0.min()) / 100.0

    def stats(cyc):
        cyc = np.asarray(cyc, dtype=np.float64)
        return {"median_cycles": float(np.median(cyc)), "p10": float(np.percentile(cyc, 10)), "p90": float(np.percentile(cyc, 90)),
                "median_us": float(np.median(cyc)) / clock_mhz}
    out = {"lib": os.environ.get("MBX_LIB_PATH", "product").split("/")[-1], "blocks": n_blocks, "waves": int(st.shape[0]),
           "gate_launch_us_events": gms / gcnt * 1e3, "launch_us_from_stamps": launch_us, "shader_clock_mhz": clock_mhz,
           "prologue": stats(ts1 - ts0), "k_loop": stats(ts2 - ts1), "epilogue": stats(ts3 - ts2), "block": stats(ts3 - ts0)}
    if "stampC" in os.environ.get("MBX_LIB_PATH", ""):
        out["prologue_requests_issued"] = stats(tbar & 0xFFFFF)
        out["prologue_tables_written"] = stats((tbar >> 20) & 0xFFFFF)
        out["prologue_first_wait_passed"] = stats((tbar >> 40) & 0xFFFFF)
    elif tbar.max() > 0:
        out["wait_and_barrier_in_loop"] = stats(tbar)
        out["wait_and_barrier_share_of_loop"] = float(np.median(tbar / np.maximum(ts2 - ts1, 1)))
    # matrix-pipe time of a wave's loop: 40 slices x 48 MFMAs x 32 cycles (product-split blocks: 24 MFMAs per wave and slice)
    nk8 = (dims.wn_channels + 7) // 8
    out["mfma_cycles_per_block_wave"] = nk8 * (24 if "stampP" in os.environ.get("MBX_LIB_PATH", "") else 48) * 32
    if stage == "res_skip":
        out["mfma_cycles_per_block_wave"] = nk8 * 44 * 32
    out["batch_frames"] = [batch, frames]
    out["stage"] = stage
    # per SIMD: resident waves over time.  HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID in the upper word
    simd = ((hw >> 32) & 15) << 16 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 11 | ((hw >> 8) & 15) << 4 | ((hw >> 4) & 3)
    cover, none_in_loop, resident, exposed_pe = [], [], [], []
    for key in np.unique(simd):
        sel = simd == key
        a0, a1, a2, a3 = ts0[sel], ts1[sel], ts2[sel], ts3[sel]
        lo, hi = a0.min(), a3.max()
        # event sweep over [lo, hi): +1 / -1 at residency and loop boundaries
        ev = np.concatenate((np.stack((a0, np.full_like(a0, 1), np.zeros_like(a0)), 1), np.stack((a3, np.full_like(a3, -1), np.zeros_like(a3)), 1),
                             np.stack((a1, np.zeros_like(a1), np.full_like(a1, 1)), 1), np.stack((a2, np.zeros_like(a2), np.full_like(a2, -1)), 1)))
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        res = loop = 0
        t_prev = lo
        t_loop = t_res_noloop = t_weighted = 0
        for tt, dr, dl in ev:
            dt = tt - t_prev
            if loop > 0:
                t_loop += dt
            elif res > 0:
                t_res_noloop += dt
            t_weighted += dt * res
            res += dr
            loop += dl
            t_prev = tt
        span = float(hi - lo)
        cover.append(t_loop / span)
        none_in_loop.append(1.0 - t_loop / span)
        exposed_pe.append(t_res_noloop / span)
        resident.append(t_weighted / span)
    out["per_simd"] = {"simds": int(len(cover)), "share_of_launch_with_a_wave_in_its_k_loop": float(np.mean(cover)),
                       "share_with_resident_waves_but_none_in_its_k_loop": float(np.mean(exposed_pe)),
                       "share_with_no_wave_in_its_k_loop": float(np.mean(none_in_loop)), "mean_resident_waves": float(np.mean(resident))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
