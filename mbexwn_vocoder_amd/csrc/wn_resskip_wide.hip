// WaveNet residual/skip layer for LARGE launches: r = a W + b (1x1), h (+)= r[:, :C], skip (+)= r[:, C:], one block
// owning ALL output columns of its rows.
//
// Same layer as wn_resskip_kernel (wn_resskip.hip; reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:322-336), with
// the skip path folded into the end convolution (engine.fold_skip_weights: cout = C + n_out).  That kernel tiles the
// columns in blocks of 128: C + n_out = 350 columns cost 384 (9.7 % of the matrix work is padding), the rows of `a` are
// fetched by three blocks, and a wave requests 4 LDS-DMA kilobytes per 16 MFMAs.  Here:
//   block = 8 waves, 128 rows x NP pairs of 16-column MFMA tiles (v_mfma_f32_16x16x4_f32).  C + n_out = 350 (C = 320):
//   NP = 11, all 352 columns; wave w owns rows 16 w .. 16 w + 15 x 22 column tiles (88 accumulator registers; 2 blocks
//   per CU = 4 waves per SIMD).  C + n_out = 370 (C = 340) needs 12 pairs = 96 accumulator registers, which leaves one
//   block per CU (162 VGPRs, 0.53 of the MFMA peak measured): two blocks per row tile own 6 pairs each instead (NP = 6,
//   SPLIT = 2; the rows of `a` are then read twice, the second time from L2; config 4: 144.8 -> 139.9 ms);
//   K slices of 8 channels, three LDS stages of (4 + NP) KB: a wave requests 2 LDS-DMA kilobytes per 4 NP MFMAs and
//   meets one barrier per slice;
//   lane n of column tile pair p holds columns 32 p + 2 n and 32 p + 2 n + 1 (pairing done by the host-side packing),
//   so h and the output accumulator are read (accumulator start = old value + bias) and written as float2.
//   A: rows [m0, m0+128) x 8 channels, 32 bytes per row, 16-byte chunk c at 2*row + (c ^ ((row>>3)&1)); lane (r = lane & 15,
//      kq = lane >> 4) reads the 8 bytes of channels 2 kq, 2 kq + 1: bank-conflict free; MFMA step m contracts {2 kq + m}
//   B: 8 channels x 32 NP columns, packed on the host in MFMA operand order [pair p][lane][even tile step 0, even tile
//      step 1, odd tile step 0, odd tile step 1] (engine.pack_resskip_wide_weights): one ds_read_b128 per lane and pair
// h_init (layer 0 with the start convolution folded in, wn_gate0.hip): rows are [a | x'] (cin = C + 16), h starts from
// the bias alone.
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RW_ROWS = 128;
constexpr int RW_BK = 8;
constexpr int RW_A_FLOATS = RW_ROWS * RW_BK;      // 1024 floats = 4 KB

__device__ __forceinline__ void rw_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ void rw_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

#define RW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define RW_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int N>
using rw_int = std::integral_constant<int, N>;

// HFULL: the first HFULL pairs of every block are known (launcher) to lie entirely inside the residual columns (32 (pr + 1) <= C)
template <int NP, int SPLIT, int HFULL>
__global__ __launch_bounds__(512, 4) void wn_resskip_wide_kernel(ConvArgs p) {
    constexpr int B_FLOATS = NP * 256;             // packed weights of one slice
    constexpr int STAGE = RW_A_FLOATS + B_FLOATS;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // SPLIT column splits: block -> (row tile g, pairs [pair0, pair0 + NP) of the SPLIT * NP pairs of the image)
    // (compile-time: with a run-time split the 11-pair kernel measured 3 % slower)
    const int g = SPLIT == 1 ? blockIdx.x : blockIdx.x / SPLIT;
    const int pair0 = SPLIT == 1 ? 0 : (blockIdx.x - g * SPLIT) * NP;
    const int b = g / p.m_tiles_per_item;
    const int mt = g - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * RW_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)m0 * p.ldx;   // the block's first row: 32-bit offsets stay small
    const int nk = (p.cin + RW_BK - 1) / RW_BK;

    // ---- LDS-DMA requests of a slice: 4 x 1 KB of activation rows + NP x 1 KB of weights = 15 / 16 chunks, two per wave
    // (NP = 11: the last weight chunk is requested twice, so that every wave has the same number of requests in flight);
    // waves 0..3: activation chunk w (rows 32 w .. 32 w + 31) and weight chunk w; waves 4..7: weight chunks w and w + 4
    unsigned a_voff = 0;
    bool a_ok = false, a_hi = false;
    if (wave < 4) {
        const int pos = wave * 64 + lane;
        const int row = pos >> 1;
        a_hi = ((pos & 1) ^ ((row >> 3) & 1)) != 0;
        a_ok = m0 + row < rows;
        a_voff = 4u * (unsigned)((min(m0 + row, rows - 1) - m0) * p.ldx + 4 * (int)a_hi);
    }
    const bool fast_rows = p.fast_dma && m0 + RW_ROWS <= rows;     // (cin = 340: only the last, half slice takes the masked path)
    const int whole_slices = p.cin / RW_BK;
    const unsigned b_voff = 16u * (unsigned)lane;
    const int bk0 = min(wave, NP - 1), bk1 = min(wave + 4, NP - 1);
    // FAST (whole-row tile, cin a multiple of the slice): only requests with a wave-uniform base and a 32-bit lane offset.  The
    // body below is instantiated twice and a block takes one copy: with the masked path (a 64-bit address per lane, a select
    // against the zero buffer) in the same loop the 128-register budget spilled an address that was reloaded from scratch
    // in EVERY slice -- and the compiler's s_waitcnt vmcnt(0) for that reload drained the two slices of LDS-DMA in flight
    // (read off the ISA in round 4)
    auto issue = [&](auto fastc, int kt, int stage) {
        constexpr bool FAST = decltype(fastc)::value;
        const int ci0 = kt * RW_BK;
        const unsigned adst = lds_base + 4u * (unsigned)(stage * STAGE);
        const unsigned bdst = adst + 4u * (unsigned)RW_A_FLOATS;
        const float *bbase = p.w + (long long)kt * (SPLIT * B_FLOATS) + pair0 * 256;
        if (wave < 4) {
            if (FAST || (fast_rows && kt < whole_slices)) {
                rw_lds_dma16_s(xb + ci0, a_voff, adst + 1024u * (unsigned)wave);
            } else {
                const bool ok = a_ok && (ci0 + 4 * (int)a_hi < p.cin);
                const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff);
                rw_lds_dma16(ok ? src : p.zeros, adst + 1024u * (unsigned)wave);
            }
            rw_lds_dma16_s(bbase + bk0 * 256, b_voff, bdst + 1024u * (unsigned)bk0);
        } else {
            rw_lds_dma16_s(bbase + bk0 * 256, b_voff, bdst + 1024u * (unsigned)bk0);
            rw_lds_dma16_s(bbase + bk1 * 256, b_voff, bdst + 1024u * (unsigned)bk1);
        }
    };
    const bool blk_fast = fast_rows && whole_slices * RW_BK == p.cin;
    // ---- accumulators start from old value + bias (h columns accumulate unless h_init, skip columns unless skip_init)
    // register v of column tile ct: row m0 + 16 wave + 4 kq + v, column 32 (ct >> 1) + 2 r16 + (ct & 1)
    f32x4 acc[2 * NP];
    const int skip_ld = p.skip_ld ? p.skip_ld : C;
    const long long skip_bstride = p.skip_bstride ? p.skip_bstride : (p.skip_ld ? (long long)p.max_rows * p.skip_ld : p.hs_bstride);
    float *hb = p.h + (long long)b * p.hs_bstride;
    float *sb = p.skip + (long long)b * skip_bstride;
    const int row0 = m0 + 16 * wave + 4 * kq;
    const int row_last = rows - 1;
    auto body = [&](auto fastc) {
    // Round 5 (in-kernel stamps, profiles/r05_phase_resskip.json): the prologue was 61 500 of a block's 240 000 cycles -- slice 0
    // requested, then the accumulator pre-loads in three groups, each waited for before the next was requested, then slices 1 and 2:
    // four serial HBM round trips under load.  Now ONE: the three slices first, then every pre-load -- the first HFULL pairs (all
    // 32 columns residual columns: a compile-time count, because a run-time branch around the stores sent the whole accumulator
    // array to scratch) as single-word loads straight into their accumulator registers (four row pointers, the pair offset an
    // immediate: no address registers to spill, no copies), the remaining pair(s) through the group code -- and only then the
    // first use.  Same sums (old value + bias first, then the products): same bits.
    issue(fastc, 0, 0);
    if (nk > 1) issue(fastc, 1, 1);
    if (nk > 2) issue(fastc, 2, 2);
    const float *bias_src = p.bias ? p.bias : p.zeros;
    const float *hrow[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) hrow[v] = hb + (long long)min(row0 + v, row_last) * C + 2 * r16;
    float2 hbias[HFULL > 0 ? HFULL : 1];
    const bool keep_h = !p.h_init;
#pragma unroll
    for (int pr = 0; pr < HFULL; ++pr) {                           // (compile-time: no branch stands between the loads and their registers)
        const int c0 = 32 * (pair0 + pr);
        hbias[pr] = *reinterpret_cast<const float2 *>(bias_src + (p.bias ? c0 + 2 * r16 : 0));
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc[2 * pr][v] = hrow[v][c0];
            acc[2 * pr + 1][v] = hrow[v][c0 + 1];
        }
    }
    // the other pairs (columns of the output accumulator, the pair that straddles C): requests of a group first, from clamped
    // addresses, then the arithmetic (written pair by pair the compiler formed `col_ok ? bias : 0` right behind each bias load:
    // an s_waitcnt vmcnt(0) in every pair, round 4)
    constexpr int PG = 4;
#pragma unroll
    for (int g0 = 0; g0 < NP; g0 += PG) {
        float2 bias4[PG], old4[PG][4];
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int pr = g0 + j;
            if (pr >= HFULL && pr < NP) {
                const int col = 32 * (pair0 + pr) + 2 * r16;       // even: both columns of the lane on the same side of C
                const int colc = min(col, p.cout - 2);
                const bool to_h = colc < C;
                bias4[j] = *reinterpret_cast<const float2 *>(bias_src + (p.bias ? colc : 0));
                const float *src = to_h ? hb + colc : sb + (colc - C);
                const int ld = to_h ? C : skip_ld;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    old4[j][v] = *reinterpret_cast<const float2 *>(src + (long long)min(row0 + v, row_last) * ld);
            }
        }
        RW_FENCE();
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int pr = g0 + j;
            if (pr >= HFULL && pr < NP) {
                const int col = 32 * (pair0 + pr) + 2 * r16;
                const bool col_ok = col < p.cout;
                const bool to_h = min(col, p.cout - 2) < C;
                const bool accumulate = col_ok && (to_h ? !p.h_init : !p.skip_init);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    acc[2 * pr][v] = (accumulate ? old4[j][v].x : 0.f) + (col_ok ? bias4[j].x : 0.f);
                    acc[2 * pr + 1][v] = (accumulate ? old4[j][v].y : 0.f) + (col_ok ? bias4[j].y : 0.f);
                }
            }
        }
        RW_FENCE();
    }
#pragma unroll
    for (int pr = 0; pr < HFULL; ++pr) {
        const float bx = p.bias ? hbias[pr].x : 0.f, by = p.bias ? hbias[pr].y : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc[2 * pr][v] = (keep_h ? acc[2 * pr][v] : 0.f) + bx;
            acc[2 * pr + 1][v] = (keep_h ? acc[2 * pr + 1][v] : 0.f) + by;
        }
    }

    // A operand: row 16 wave + r16, channels 2 kq, 2 kq + 1
    const int arow = 16 * wave + r16;
    const float *aptr = lds + 8 * arow + 4 * ((kq >> 1) ^ ((arow >> 3) & 1)) + 2 * (kq & 1);
    const float *bptr = lds + RW_A_FLOATS + lane * 4;
    float2 av;
    float4 bw[3];          // weights of pair p in bw[p % 3], requested two pairs ahead

    auto load_a = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        av = *reinterpret_cast<const float2 *>(aptr + S * STAGE);
    };
    auto load_b = [&](auto sc, auto pc) {
        constexpr int S = decltype(sc)::value, P = decltype(pc)::value;
        bw[P % 3] = *reinterpret_cast<const float4 *>(bptr + S * STAGE + P * 256);
    };
    auto mfma4 = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        const float4 w = bw[P % 3];
        acc[2 * P] = RW_MFMA(av.x, w.x, acc[2 * P]);
        acc[2 * P + 1] = RW_MFMA(av.x, w.z, acc[2 * P + 1]);
        acc[2 * P] = RW_MFMA(av.y, w.y, acc[2 * P]);
        acc[2 * P + 1] = RW_MFMA(av.y, w.w, acc[2 * P + 1]);
    };
    // One slice = NP phases of 4 MFMAs (one column tile pair each); the weights of pair p+2 are requested from LDS before
    // the MFMAs of pair p issue.  The barrier that publishes slice kt+1 sits in front of the last two pairs: every wave
    // has requested all LDS operands of slice kt by then, so the stage is free for slice kt+3.
    // In: av, bw[0], bw[1] of this slice.  Out: those of the next one.  2 LDS-DMA instructions per wave and slice.
    auto slice = [&](auto sc, int kt) {
        constexpr int S = decltype(sc)::value;
        rw_int<(S + 1) % 3> ns;
#define RW_PHASE(P)                                  \
    if constexpr (P < NP - 2) {                      \
        load_b(sc, rw_int<P + 2>());                 \
        RW_FENCE();                                  \
        mfma4(rw_int<P>());                          \
        RW_FENCE();                                  \
    }
        RW_PHASE(0) RW_PHASE(1) RW_PHASE(2) RW_PHASE(3) RW_PHASE(4) RW_PHASE(5)
        RW_PHASE(6) RW_PHASE(7) RW_PHASE(8) RW_PHASE(9)
#undef RW_PHASE
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // slice kt+2 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 3 < nk) issue(fastc, kt + 3, S);
        RW_FENCE();
        mfma4(rw_int<NP - 2>());
        mfma4(rw_int<NP - 1>());
        RW_FENCE();
        load_a(ns);                                  // av of slice kt is dead: every MFMA that reads it has been issued
        load_b(ns, rw_int<0>());
        load_b(ns, rw_int<1>());
        RW_FENCE();
    };

    // ---- the first slice has landed
    if (nk > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_a(rw_int<0>());
    load_b(rw_int<0>(), rw_int<0>());
    load_b(rw_int<0>(), rw_int<1>());
    {
        int kt = 0;
        for (; kt + 3 <= nk; kt += 3) {
            slice(rw_int<0>(), kt);
            slice(rw_int<1>(), kt + 1);
            slice(rw_int<2>(), kt + 2);
        }
        if (kt < nk) {
            slice(rw_int<0>(), kt);
            if (kt + 1 < nk) slice(rw_int<1>(), kt + 1);
        }
    }

    };      // body
    if (blk_fast) body(std::true_type{});
    else body(std::false_type{});

    // ---- epilogue: the accumulators are the new values
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
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

// a.w must point at the image of engine.pack_resskip_wide_weights (ceil(cin/8), np, 256) with np = ceil(cout/32) in
// {11, 12}; returns false if the layer does not fit (the caller then uses launch_wn_resskip)
bool launch_wn_resskip_wide(const ConvArgs &a, hipStream_t stream) {
    const int np = (a.cout + 31) / 32;
    const bool ok = a.ks == 1 && (a.h_init ? a.cin >= a.channels : a.cin == a.channels) && !a.last_layer && a.skip_ld > 0 &&
                    (np == 11 || np == 12) && a.cin % 4 == 0 && a.cin >= 3 * RW_BK && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 &&
                    a.channels % 2 == 0 && a.skip_ld % 2 == 0 && a.cout % 2 == 0 && a.cout <= a.channels + a.skip_ld &&
                    (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && (uintptr_t)a.h % 8 == 0 &&
                    (uintptr_t)a.skip % 8 == 0 && (!a.bias || (uintptr_t)a.bias % 8 == 0) && a.hs_bstride % 2 == 0 && a.zeros &&
                    a.h && a.skip;
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = 1;                 // byte offsets are relative to the block's first row
    r.m_tiles_per_item = (a.max_rows + RW_ROWS - 1) / RW_ROWS;
    const long long blocks = (long long)r.m_tiles_per_item * a.batch;
    // 11 pairs (C = 320): one block owns all columns of its rows.  12 pairs (C = 340) would need 96 accumulator registers
    // per wave, which leaves one 8-wave block per CU (162 VGPRs; measured 0.53 of the peak): two blocks per row tile own
    // six pairs each instead and read the rows of `a` twice (the second read is an L2 hit)
    if (np == 11 && a.channels >= 320) hipLaunchKernelGGL((wn_resskip_wide_kernel<11, 1, 10>), dim3((unsigned)blocks), dim3(512), 0, stream, r);
    else if (np == 11) hipLaunchKernelGGL((wn_resskip_wide_kernel<11, 1, 0>), dim3((unsigned)blocks), dim3(512), 0, stream, r);
    else hipLaunchKernelGGL((wn_resskip_wide_kernel<6, 2, 0>), dim3((unsigned)(2 * blocks)), dim3(512), 0, stream, r);
    return true;
}

}  // namespace mbx
