// WaveNet residual/skip layer for SMALL launches (one utterance, the steady ticks of the streams), tiled at WAVE
// granularity: r = a W + b (1x1), h (+)= r[:, :C], accumulator (+)= r[:, C:]  (skip path folded into the end
// convolution, engine.fold_skip_weights: cout = C + n_out).
//
// Same layer and same arithmetic per output as wn_resskip_wide_kernel (wn_resskip_wide.hip; reference
// MBExWN_NVoc/vocoder/model/custom_AE_layers.py:322-336).  What differs is how the work is cut.  The block shapes of
// wn_resskip.hip / wn_resskip_wide.hip own 64 / 128 rows: a stream that contributes 160 new rows per tick fills them to
// 83 / 63 %, a 10 s utterance (16 000 rows) is 125 row tiles for 256 CUs, and a few resident blocks per CU cannot hide
// their own prologue and epilogue.  Here a wave owns 16 rows (one v_mfma_f32_16x16x4_f32 row tile) x NPW pairs of
// 16-column tiles, and a block is four INDEPENDENT wave tiles -- wave w of row group g owns tile 4 g + w of the flat
// list (item, 16-row tile), stages its own rows, and only shares the weight slices with the other three:
//   NPW = 11 / 12  all columns of the rows (C + n_out = 350 / 370): 1 000 wave tiles for a 10 s utterance = one wave per
//                  SIMD, 88 / 96 MFMAs per 16-channel slice
//   NPW = 6, 4     two / three column splits (blockIdx.y): more, shorter waves when the tiles alone do not fill the
//                  1 024 SIMDs evenly (64 streams x 10 tiles: 640 -> 1 920 waves)
// K slices of 16 channels through LDS-DMA, NSTAGE stages (deep: these launches run one or two waves per SIMD, so the
// LDS-DMA latency has to be covered by stages in flight, not by other waves):
//   A: a wave's 16 rows x 16 channels = 1 KB, one request; 16-byte chunk c of row r at 4 r + (c ^ (-(r >> 2) & 3)): the
//      ds_read_b128 of lane (r = lane & 15, kq = lane >> 4) -- channels 4 kq .. 4 kq + 3, contracted by MFMA steps 0..3 --
//      is bank-conflict free
//   B: 16 channels x 32 NPW columns in MFMA operand order [pair p][lane][even tile steps 0..3 | odd tile steps 0..3]
//      (engine.pack_resskip_wave_weights): two ds_read_b128 per lane and pair = the weight operands of eight MFMAs
// The accumulators start from old value + bias, the epilogue is a plain float2 store (as in the wide kernel).
// h_init (layer 0 with the start convolution folded in, wn_gate0.hip): rows are [a | x'] (cin = C + 16), h starts from
// the bias alone.
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RV_BK = 16;
constexpr int RV_A_FLOATS = 16 * RV_BK;           // one wave tile of one slice: 256 floats = 1 KB
constexpr int RV_PAIRS = 12;                      // pairs of the weight image (zero padded)

__device__ __forceinline__ void rv_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ void rv_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

#define RV_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define RV_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int N>
using rv_int = std::integral_constant<int, N>;

template <int N>
__device__ __forceinline__ void rv_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

template <int NPW, int NSTAGE>
struct RvShape {
    static constexpr int B_FLOATS = NPW * 512;                     // packed weights of one slice and column split
    static constexpr int B_CHUNKS = 2 * NPW;                       // 1 KB LDS-DMA requests
    static constexpr int B_INST = (B_CHUNKS + 3) / 4;              // per wave (the last ones doubled when 2 NPW % 4 != 0)
    static constexpr int DPS = 1 + B_INST;                         // LDS-DMA requests per wave and stage
    static constexpr int STAGE = 4 * RV_A_FLOATS + B_FLOATS;
    static constexpr int LDS_FLOATS = NSTAGE * STAGE;
    static constexpr int ACC_REGS = 8 * NPW;
    static constexpr int WAVES_PER_SIMD = ACC_REGS > 64 ? 2 : (LDS_FLOATS * 4 * 4 <= 160 * 1024 ? 4 : (LDS_FLOATS * 4 * 3 <= 160 * 1024 ? 3 : 2));
};

template <int NPW, int NSTAGE>
__global__ __launch_bounds__(256, (RvShape<NPW, NSTAGE>::WAVES_PER_SIMD)) void wn_resskip_wave_kernel(ConvArgs p) {
    using SH = RvShape<NPW, NSTAGE>;
    constexpr int STAGE = SH::STAGE, DPS = SH::DPS;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[SH::LDS_FLOATS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    const int g = blockIdx.x;                        // row group: wave tiles 4 g .. 4 g + 3
    const int pair0 = blockIdx.y * NPW;              // column split
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int tpi = p.m_tiles_per_item;
    int b = 0, m0 = 0, rows = 1;
    bool active = false, any = false;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int wt = 4 * g + w;
        if (wt >= p.m_tiles_total) break;
        const int bb = wt / tpi;
        const int rr = item_rows(p.n_frames, bb, p.rows_per_frame, p.max_rows);
        const int mm = 16 * (wt - bb * tpi);
        any = any || mm < rr;
        if (w == wave) {
            b = bb;
            m0 = mm;
            rows = max(rr, 1);
            active = mm < rr;
        }
    }
    if (!any) return;
    if (!active) m0 = 0;
    const int C = p.channels;
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)m0 * p.ldx;   // the tile's first row: 32-bit offsets stay small
    const int nk = (p.cin + RV_BK - 1) / RV_BK;

    // ---- LDS-DMA requests of a slice: this wave's rows (1 KB) + B_INST of the 2 NPW weight kilobytes
    unsigned a_voff;
    bool a_ok;
    int a_ch;
    {
        const int row = lane >> 2;
        a_ch = 4 * ((lane & 3) ^ ((-(row >> 2)) & 3));
        a_ok = active && m0 + row < rows;
        a_voff = 4u * (unsigned)((min(m0 + row, rows - 1) - m0) * p.ldx + a_ch);
    }
    const bool fast_rows = active && p.fast_dma && m0 + 16 <= rows;
    const int whole_slices = p.cin / RV_BK;
    const unsigned b_voff = 16u * (unsigned)lane;
    auto issue = [&](int kt, int stage) {
        const int ci0 = kt * RV_BK;
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * STAGE);
        const unsigned adst = sdst + 4u * (unsigned)(wave * RV_A_FLOATS);
        const unsigned bdst = sdst + 4u * (unsigned)(4 * RV_A_FLOATS);
        if (fast_rows && kt < whole_slices) {
            rv_lds_dma16_s(xb + ci0, a_voff, adst);
        } else {
            const bool ok = a_ok && (ci0 + a_ch < p.cin);
            const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff);
            rv_lds_dma16(ok ? src : p.zeros, adst);
        }
        const float *bbase = p.w + ((long long)kt * RV_PAIRS + pair0) * 512;
#pragma unroll
        for (int i = 0; i < SH::B_INST; ++i) {
            const int k = min(wave + 4 * i, SH::B_CHUNKS - 1);             // the surplus requests repeat the last chunk
            rv_lds_dma16_s(bbase + k * 256, b_voff, bdst + 1024u * (unsigned)k);
        }
    };
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (s < nk) issue(s, s);

    // ---- accumulators start from old value + bias (h columns accumulate unless h_init, accumulator columns unless skip_init)
    // register v of column tile ct: row m0 + 4 kq + v, column 32 (pair0 + (ct >> 1)) + 2 r16 + (ct & 1)
    f32x4 acc[2 * NPW];
    const int skip_ld = p.skip_ld ? p.skip_ld : C;
    const long long skip_bstride = p.skip_bstride ? p.skip_bstride : (p.skip_ld ? (long long)p.max_rows * p.skip_ld : p.hs_bstride);
    float *hb = p.h + (long long)b * p.hs_bstride;
    float *sb = p.skip + (long long)b * skip_bstride;
    const int row0 = m0 + 4 * kq;
    const int row_last = rows - 1;
    // (all old values are requested before the first one is used: these launches run one or two waves per SIMD, so a
    // pre-load that waits batch by batch would be serial time; the loads land in the accumulator registers themselves)
#pragma unroll
    for (int pr = 0; pr < NPW; ++pr) {
        const int colc = min(32 * (pair0 + pr) + 2 * r16, p.cout - 2);      // even: both columns of the lane on the same side of C
        const bool to_h = colc < C;
        const float *src = to_h ? hb + colc : sb + (colc - C);
        const int ld = to_h ? C : skip_ld;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float2 old = *reinterpret_cast<const float2 *>(src + (long long)min(row0 + v, row_last) * ld);
            acc[2 * pr][v] = old.x;
            acc[2 * pr + 1][v] = old.y;
        }
    }
#pragma unroll
    for (int pr = 0; pr < NPW; ++pr) {
        const int col = 32 * (pair0 + pr) + 2 * r16;
        const bool col_ok = col < p.cout;
        const int colc = min(col, p.cout - 2);
        const bool accumulate = col_ok && (colc < C ? !p.h_init : !p.skip_init);
        float2 bias = make_float2(0.f, 0.f);
        if (p.bias && col_ok) bias = *reinterpret_cast<const float2 *>(p.bias + colc);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc[2 * pr][v] = (accumulate ? acc[2 * pr][v] : 0.f) + bias.x;
            acc[2 * pr + 1][v] = (accumulate ? acc[2 * pr + 1][v] : 0.f) + bias.y;
        }
    }

    // A operand: row r16 of this wave's tile, channels 4 kq .. 4 kq + 3
    const float *aptr = lds + wave * RV_A_FLOATS + 16 * r16 + 4 * (kq ^ ((-(r16 >> 2)) & 3));
    const float *bptr = lds + 4 * RV_A_FLOATS + lane * 4;
    float4 av;
    float4 bw[3][2];          // weights of pair p in bw[p % 3] (even tile, odd tile), requested two pairs ahead

    auto load_a = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        av = *reinterpret_cast<const float4 *>(aptr + S * STAGE);
    };
    auto load_b = [&](auto sc, auto pc) {
        constexpr int S = decltype(sc)::value, P = decltype(pc)::value;
        bw[P % 3][0] = *reinterpret_cast<const float4 *>(bptr + S * STAGE + P * 512);
        bw[P % 3][1] = *reinterpret_cast<const float4 *>(bptr + S * STAGE + P * 512 + 256);
    };
    auto mfma8 = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        const float4 we = bw[P % 3][0], wo = bw[P % 3][1];
        acc[2 * P] = RV_MFMA(av.x, we.x, acc[2 * P]);
        acc[2 * P + 1] = RV_MFMA(av.x, wo.x, acc[2 * P + 1]);
        acc[2 * P] = RV_MFMA(av.y, we.y, acc[2 * P]);
        acc[2 * P + 1] = RV_MFMA(av.y, wo.y, acc[2 * P + 1]);
        acc[2 * P] = RV_MFMA(av.z, we.z, acc[2 * P]);
        acc[2 * P + 1] = RV_MFMA(av.z, wo.z, acc[2 * P + 1]);
        acc[2 * P] = RV_MFMA(av.w, we.w, acc[2 * P]);
        acc[2 * P + 1] = RV_MFMA(av.w, wo.w, acc[2 * P + 1]);
    };
    // One slice = NPW phases of 8 MFMAs (one column tile pair each); the weights of pair p+2 are requested from LDS before
    // the MFMAs of pair p issue.  The barrier that publishes slice kt+1 sits in front of the last two pairs: every wave
    // has requested all LDS operands of slice kt by then, so the stage is free for slice kt+NSTAGE.  Behind that barrier
    // the slices kt+2 .. kt+NSTAGE-1 may still be in flight.
    // In: av, bw[0], bw[1] of this slice.  Out: those of the next one.
    auto slice = [&](auto sc, int kt) {
        constexpr int S = decltype(sc)::value;
        rv_int<(S + 1) % NSTAGE> ns;
        if (active) {
#define RV_PHASE(P)                                  \
    if constexpr (P < NPW - 2) {                     \
        load_b(sc, rv_int<P + 2>());                 \
        RV_FENCE();                                  \
        mfma8(rv_int<P>());                          \
        RV_FENCE();                                  \
    }
            RV_PHASE(0) RV_PHASE(1) RV_PHASE(2) RV_PHASE(3) RV_PHASE(4) RV_PHASE(5)
            RV_PHASE(6) RV_PHASE(7) RV_PHASE(8) RV_PHASE(9)
#undef RV_PHASE
        }
        const int later = min(NSTAGE - 2, nk - kt - 2);            // slices behind kt+1 that have been requested
        if (NSTAGE >= 4 && later >= 2) rv_wait_vm<2 * DPS>();
        else if (NSTAGE >= 3 && later >= 1) rv_wait_vm<DPS>();
        else rv_wait_vm<0>();
        __syncthreads();
        if (kt + NSTAGE < nk) issue(kt + NSTAGE, S);
        if (active) {
            RV_FENCE();
            mfma8(rv_int<NPW - 2>());
            mfma8(rv_int<NPW - 1>());
            RV_FENCE();
            load_a(ns);                                  // av of slice kt is dead: every MFMA that reads it has been issued
            load_b(ns, rv_int<0>());
            load_b(ns, rv_int<1>());
            RV_FENCE();
        }
    };

    // ---- the first slice has landed (the accumulator pre-loads were requested behind all stages: waiting for them
    // waits for every stage; the pipeline refills as the loop goes)
    rv_wait_vm<0>();
    __syncthreads();
    if (active) {
        load_a(rv_int<0>());
        load_b(rv_int<0>(), rv_int<0>());
        load_b(rv_int<0>(), rv_int<1>());
    }
    {
        int kt = 0;
        for (; kt + NSTAGE <= nk; kt += NSTAGE) {
            slice(rv_int<0>(), kt);
            slice(rv_int<1 % NSTAGE>(), kt + 1);
            if constexpr (NSTAGE >= 3) slice(rv_int<2 % NSTAGE>(), kt + 2);
            if constexpr (NSTAGE >= 4) slice(rv_int<3 % NSTAGE>(), kt + 3);
        }
        if (kt < nk) {
            slice(rv_int<0>(), kt);
            if (kt + 1 < nk) slice(rv_int<1 % NSTAGE>(), kt + 1);
            if constexpr (NSTAGE >= 4)
                if (kt + 2 < nk) slice(rv_int<2 % NSTAGE>(), kt + 2);
        }
    }
    if (!active) return;

    // ---- epilogue: the accumulators are the new values
#pragma unroll
    for (int pr = 0; pr < NPW; ++pr) {
        const int col = 32 * (pair0 + pr) + 2 * r16;
        if (col >= p.cout) continue;
        const bool to_h = col < C;
        float *dst = to_h ? hb + col : sb + (col - C);
        const int ld = to_h ? C : skip_ld;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = row0 + v;
            if (row < rows) *reinterpret_cast<float2 *>(dst + (long long)row * ld) = make_float2(acc[2 * pr][v], acc[2 * pr + 1][v]);
        }
    }
}

// a.w must point at the image of engine.pack_resskip_wave_weights (ceil(cin/16), 12, 512); returns false if the layer
// does not fit (the caller then uses launch_wn_resskip).  The cut (pairs per wave) follows the number of wave tiles:
// all columns per wave while the tiles alone give every SIMD about one wave, column splits below that.
bool launch_wn_resskip_wave(const ConvArgs &a, hipStream_t stream) {
    const int np = (a.cout + 31) / 32;
    const int nk = (a.cin + RV_BK - 1) / RV_BK;
    const bool ok = a.ks == 1 && (a.h_init ? a.cin >= a.channels : a.cin == a.channels) && !a.last_layer && a.skip_ld > 0 &&
                    (np == 11 || np == 12) && a.cin % 4 == 0 && nk >= 4 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 &&
                    a.channels % 2 == 0 && a.skip_ld % 2 == 0 && a.cout % 2 == 0 && a.cout <= a.channels + a.skip_ld &&
                    (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && (uintptr_t)a.h % 8 == 0 &&
                    (uintptr_t)a.skip % 8 == 0 && (!a.bias || (uintptr_t)a.bias % 8 == 0) && a.hs_bstride % 2 == 0 && a.zeros &&
                    a.h && a.skip;
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = 1;                 // byte offsets are relative to the tile's first row
    r.m_tiles_per_item = (a.max_rows + 15) / 16;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const unsigned groups = (unsigned)((r.m_tiles_total + 3) / 4);
    // waves per SIMD with s column splits = tiles * s / 1024: take the cut whose last round is fullest
    int split = a.tune_split;
    if (split != 1 && split != 2 && split != 3) {
        double best = -1.0;
        for (int s = 1; s <= 3; ++s) {
            const double per_simd = (double)r.m_tiles_total * s / 1024.0;
            const double fill = per_simd / std::ceil(per_simd);
            const double score = fill - 0.02 * (s - 1);            // prefer fewer splits (less re-reading of the rows)
            if (score > best) {
                best = score;
                split = s;
            }
        }
    }
    // (stages: 2, 3 and 4 measured the same for the all-columns cut: 41.5 / 41.9 / 42.1 us for a 10 s utterance)
    if (split == 1) {
        if (np == 11) hipLaunchKernelGGL((wn_resskip_wave_kernel<11, 3>), dim3(groups, 1), dim3(256), 0, stream, r);
        else hipLaunchKernelGGL((wn_resskip_wave_kernel<12, 3>), dim3(groups, 1), dim3(256), 0, stream, r);
    } else if (split == 2) {
        hipLaunchKernelGGL((wn_resskip_wave_kernel<6, 4>), dim3(groups, 2), dim3(256), 0, stream, r);
    } else {
        hipLaunchKernelGGL((wn_resskip_wave_kernel<4, 4>), dim3(groups, 3), dim3(256), 0, stream, r);
    }
    return true;
}

}  // namespace mbx
