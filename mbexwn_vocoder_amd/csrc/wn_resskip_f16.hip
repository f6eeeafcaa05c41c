// WaveNet residual/skip layer in SPLIT half precision (opt-in: mbx_config.wn_precision = MBX_PRECISION_SPLIT_F16; never
// the default and never the headline measurement -- the float32 kernels are wn_resskip_wide.hip / _wave.hip / wn_resskip.hip).
//
// Same layer (reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:322-336, skip path folded into the end convolution:
// r = a W + b, h += r[:, :C], acc += r[:, C:]) with the contraction on the 16-bit matrix pipe, which runs at 16 x the
// float32 one.  Every float32 operand x is split into hi = fp16(x) and lo' = fp16((x - hi) * 2^11); the product
//     x y  ~  hi_x hi_y + 2^-11 (hi_x lo'_y + lo'_x hi_y)
// drops only lo lo (2^-22 relative), every fp16 x fp16 product is exact in the float32 accumulator, and the 2^11 keeps the
// low parts in fp16's normal range: the result has the error of a plain float32 contraction (emulated on the canonical
// K = 960 contraction: 2.7e-6 against 3.1e-6; NOTEBOOK.md R4 section 9).  One accumulator holds 2^11 times the result: the
// hi x hi product takes the activation's high part times 2^11 (exact in fp16 for |x| < 32: the gate output `a` lies in
// (-1, 1) -- which is why the glu gate, whose linear half is unbounded, does not get this mode -- and the excitation channels
// that layer 0's rows carry behind it, pulse samples and the noise draw, are O(1)), the old value and the bias enter times
// 2^11, and the epilogue multiplies by 2^-11 -- powers of two, so no rounding is added.  The weights are split on the
// host (engine.pack_resskip_f16_weights).
//
// Block = 8 waves, 128 rows x 6 pairs of 16-column tiles (two blocks per row tile: 12 pairs = 384 columns cover C + n_out
// <= 384); wave w owns rows 16 w .. 16 w + 15 x 12 column tiles (48 accumulator registers; 2 blocks per CU).
// K steps of 32 channels = one v_mfma_f32_16x16x32_f16 per (tile, product):
//   A: a lane (row r = lane & 15, kq = lane >> 4) holds the channels k0 + 4 kq .. + 3 and k0 + 16 + 4 kq .. + 3 of its row:
//      two LDS-DMA requests per wave and step land them in a 2 KB zone of the wave's own (the lane that requested a
//      16-byte piece reads it back: 16 rows x 64 contiguous bytes per request), and the lane splits them in registers:
//      ~65 vector instructions per step, which hide in the issue slots the 36 16-bit MFMAs of a step leave free (an MFMA
//      of this kind holds the vector issue for 8 of its 16 cycles; MI355X_MICROARCH.md).  (Register loads do not work
//      here: a load the compiler sees makes it insert a wait that -- blind to the LDS-DMA requests queued behind -- drains
//      the whole prefetch; a load hidden in inline asm lets the compiler copy the destination before the data is there.)
//   B: 12 pairs x [even hi | even lo | odd hi | odd lo] x 64 lanes x 8 halves, packed on the host in MFMA operand order
//      (lane n of pair p holds columns 32 p + 2 n, 32 p + 2 n + 1: float2 access to h and the accumulator as in
//      wn_resskip_wide.hip), copied into LDS by LDS-DMA, 3 requests per wave and step.
// Two stages of 24 KB (weights) + 16 KB (activations): 80 KB, two blocks per CU; a step's requests are issued behind the
// barrier of the step before, one barrier per step.
// Accumulators start from 2^11 (old value + bias); the epilogue stores 2^-11 times them.
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RH_ROWS = 128;
constexpr int RH_BK = 32;
constexpr int RH_NP = 6;                               // pairs per block
constexpr int RH_PAIR_FLOATS = 4 * 64 * 4;             // one pair of one step: 4 operand images x 64 lanes x 16 bytes = 4 KB
constexpr int RH_B_FLOATS = RH_NP * RH_PAIR_FLOATS;    // weights of a step: 24 KB
constexpr int RH_A_FLOATS = 8 * 512;                   // activations of a step: 8 waves x 2 KB
constexpr int RH_STAGE = RH_B_FLOATS + RH_A_FLOATS;    // 40 KB
constexpr int RH_NSTAGE = 2;

__device__ __forceinline__ void rh_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

__device__ __forceinline__ void rh_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

#define RH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define RH_FENCE() __builtin_amdgcn_sched_barrier(0)

// x -> (hi, 2^11 hi, lo'): hi = fp16(x) (round to nearest), hs = 2^11 hi (exact in fp16 for |x| < 32), lo' = fp16(2^11 x - hs):
// the fused multiply-subtract is exact in float32 (2^11 (x - hi) has at most 13 significant bits), so lo' is (x - hi) 2^11
// rounded once.  Written so that the compiler can use the mixed-precision FMA (v_fma_mix*_f16: float32 and fp16 sources in
// one instruction): 2 instructions per element instead of 7.
__device__ __forceinline__ void rh_split(const f32x4 &lo4, const f32x4 &hi4, f16x8 &h, f16x8 &hs, f16x8 &l) {
    const float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 hh = (_Float16)x[i];
        const _Float16 hs_i = hh * (_Float16)2048.0f;
        h[i] = hh;
        hs[i] = hs_i;
        l[i] = (_Float16)__builtin_fmaf(x[i], 2048.0f, -(float)hs_i);
    }
}

__global__ __launch_bounds__(512, 4) void wn_resskip_f16_kernel(ConvArgs p) {
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[RH_NSTAGE * RH_STAGE];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // the two column halves of a row tile run on one XCD (workgroup ids go round the 8 XCDs: ids i and i + 8 share one), so
    // that the rows of `a`, which both of them read, come from HBM once and from that XCD's L2 the second time
    const int g = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    const int pair0 = ((blockIdx.x >> 3) & 1) * RH_NP;
    if (g >= p.m_tiles_total) return;
    const int b = g / p.m_tiles_per_item;
    const int mt = g - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * RH_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = (p.cin + RH_BK - 1) / RH_BK;

    // ---- requests of a step: weights (step kt = 12 pairs x 4 KB; this block's 6 pairs are 24 consecutive 1 KB pieces, three
    // per wave) and this wave's activation rows (row of this lane clamped: rows behind the item's end are computed and not
    // stored; channels behind cin come from the zero buffer)
    const unsigned b_voff = 16u * (unsigned)lane;
    const int arow = min(m0 + 16 * wave + r16, rows - 1);
    const float *ap = p.x + (long long)b * p.x_bstride + (long long)arow * p.ldx + 4 * kq;
    auto issue = [&](int kt, int stage) {
        const unsigned dst = lds_base + 4u * (unsigned)(stage * RH_STAGE);
        const float *src = p.w + ((long long)kt * 12 + pair0) * RH_PAIR_FLOATS;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 8 * i;
            rh_lds_dma16_s(src + piece * 256, b_voff, dst + 1024u * (unsigned)piece);
        }
        const unsigned adst = dst + 4u * (unsigned)(RH_B_FLOATS + wave * 512);
        const int c0 = kt * RH_BK + 4 * kq;
        rh_lds_dma16(c0 < p.cin ? ap + kt * RH_BK : p.zeros, adst);
        rh_lds_dma16(c0 + 16 < p.cin ? ap + kt * RH_BK + 16 : p.zeros, adst + 1024u);
    };
    issue(0, 0);

    // ---- accumulators: 2^11 x (old value + bias) (h columns accumulate, skip columns unless skip_init)
    f32x4 acc[2 * RH_NP];
    const int skip_ld = p.skip_ld ? p.skip_ld : C;
    const long long skip_bstride = p.skip_bstride ? p.skip_bstride : (p.skip_ld ? (long long)p.max_rows * p.skip_ld : p.hs_bstride);
    float *hb = p.h + (long long)b * p.hs_bstride;
    float *sb = p.skip + (long long)b * skip_bstride;
    const int row0 = m0 + 16 * wave + 4 * kq;
    const int row_last = rows - 1;
    // (groups of three pairs: all requests of a group first, then the arithmetic -- pair by pair the compiler waits for every
    // bias load on its own, which also drains the previous pair's row loads: wn_resskip_wide.hip)
    const float *bias_src = p.bias ? p.bias : p.zeros;
    const bool planes_only = p.h_planes_only != 0 && p.h_split != nullptr;
    const _Float16 *planes_in = p.h_split ? reinterpret_cast<const _Float16 *>(p.h_split + (long long)b * p.h_split_bstride) : nullptr;
    const int pld_in = 2 * p.h_split_ld;
#pragma unroll
    for (int g0 = 0; g0 < RH_NP; g0 += 3) {
        float2 bias3[3], old3[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pr = g0 + j;
            const int colc = min(32 * (pair0 + pr) + 2 * r16, p.cout - 2);
            const bool to_h = colc < C;
            bias3[j] = *reinterpret_cast<const float2 *>(bias_src + (p.bias ? colc : 0));
            if (planes_only && to_h) {
                // the hidden state lives in the fp16 planes only: old value = hi + 2^-11 lo' (exact in float32)
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const _Float16 *pr_ = planes_in + (long long)min(row0 + v, row_last) * pld_in + colc;
                    const f16x2 hh = *reinterpret_cast<const f16x2 *>(pr_), ll = *reinterpret_cast<const f16x2 *>(pr_ + p.h_split_ld);
                    old3[j][v] = make_float2(fmaf((float)ll[0], 1.0f / 2048.0f, (float)hh[0]), fmaf((float)ll[1], 1.0f / 2048.0f, (float)hh[1]));
                }
            } else {
                const float *src = to_h ? hb + colc : sb + (colc - C);
                const int ld = to_h ? C : skip_ld;
#pragma unroll
                for (int v = 0; v < 4; ++v) old3[j][v] = *reinterpret_cast<const float2 *>(src + (long long)min(row0 + v, row_last) * ld);
            }
        }
        RH_FENCE();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pr = g0 + j;
            const int col = 32 * (pair0 + pr) + 2 * r16;
            const bool col_ok = col < p.cout;
            const bool to_h = min(col, p.cout - 2) < C;
            const bool accumulate = col_ok && (to_h ? !p.h_init : !p.skip_init);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                acc[2 * pr][v] = 2048.0f * ((accumulate ? old3[j][v].x : 0.f) + (col_ok ? bias3[j].x : 0.f));
                acc[2 * pr + 1][v] = 2048.0f * ((accumulate ? old3[j][v].y : 0.f) + (col_ok ? bias3[j].y : 0.f));
            }
        }
        RH_FENCE();
    }
    const f16x8 *bptr = reinterpret_cast<const f16x8 *>(lds) + lane;      // + stage * (RH_STAGE / 4) + (4 pr + image) * 64
    const f32x4 *aptr = reinterpret_cast<const f32x4 *>(lds) + RH_B_FLOATS / 4 + wave * 128 + lane;
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        // every request this wave has in flight belongs to step kt (the next step's are issued behind the barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // step kt's operands are there for every wave; and every wave is done
        if (kt + 1 < nk) issue(kt + 1, stage ^ 1);     // with step kt - 1, whose stage takes step kt + 1
        const f32x4 a_lo4 = aptr[stage * (RH_STAGE / 4)], a_hi4 = aptr[stage * (RH_STAGE / 4) + 64];
        f16x8 ah, ahs, al;
        rh_split(a_lo4, a_hi4, ah, ahs, al);
        const f16x8 *bs = bptr + stage * (RH_STAGE / 4);
#pragma unroll
        for (int pr = 0; pr < RH_NP; ++pr) {
            const f16x8 beh = bs[(4 * pr + 0) * 64], bel = bs[(4 * pr + 1) * 64];
            const f16x8 boh = bs[(4 * pr + 2) * 64], bol = bs[(4 * pr + 3) * 64];
            acc[2 * pr] = RH_MFMA(ahs, beh, acc[2 * pr]);
            acc[2 * pr + 1] = RH_MFMA(ahs, boh, acc[2 * pr + 1]);
            acc[2 * pr] = RH_MFMA(ah, bel, acc[2 * pr]);
            acc[2 * pr + 1] = RH_MFMA(ah, bol, acc[2 * pr + 1]);
            acc[2 * pr] = RH_MFMA(al, beh, acc[2 * pr]);
            acc[2 * pr + 1] = RH_MFMA(al, boh, acc[2 * pr + 1]);
        }
    }

    // ---- epilogue: new value = 2^-11 x accumulator; with h_split the new hidden state also goes out as fp16 planes (hi,
    // lo' = (x - hi) 2^11) for the split gate kernel of the next layer, the plane padding behind C as zeros
    _Float16 *planes = p.h_split ? reinterpret_cast<_Float16 *>(p.h_split + (long long)b * p.h_split_bstride) : nullptr;
    const int pld = 2 * p.h_split_ld;              // halves per row of the planes
#pragma unroll
    for (int pr = 0; pr < RH_NP; ++pr) {
        const int col = 32 * (pair0 + pr) + 2 * r16;
        if (planes && col < p.h_split_ld) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = row0 + v;
                if (row < rows) {
                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                    f16x2 hh = {(_Float16)0.f, (_Float16)0.f}, ll = hh;
                    if (col < C) {
                        const float x0 = acc[2 * pr][v] * (1.0f / 2048.0f), x1 = acc[2 * pr + 1][v] * (1.0f / 2048.0f);
                        hh[0] = (_Float16)x0;
                        hh[1] = (_Float16)x1;
                        ll[0] = (_Float16)__builtin_fmaf(x0, 2048.0f, -2048.0f * (float)hh[0]);
                        ll[1] = (_Float16)__builtin_fmaf(x1, 2048.0f, -2048.0f * (float)hh[1]);
                    }
                    *reinterpret_cast<f16x2 *>(planes + (long long)row * pld + col) = hh;
                    *reinterpret_cast<f16x2 *>(planes + (long long)row * pld + p.h_split_ld + col) = ll;
                }
            }
        }
        if (col >= p.cout) continue;
        const bool to_h = col < C;
        if (to_h && planes_only) continue;             // the planes above are the hidden state
        float *dst = to_h ? hb + col : sb + (col - C);
        const int ld = to_h ? C : skip_ld;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = row0 + v;
            if (row < rows)
                *reinterpret_cast<float2 *>(dst + (long long)row * ld) =
                    make_float2(acc[2 * pr][v] * (1.0f / 2048.0f), acc[2 * pr + 1][v] * (1.0f / 2048.0f));
        }
    }
}

// a.w must point at the image of engine.pack_resskip_f16_weights (ceil(cin/32), 12, 1024 floats); returns false if the
// layer does not fit (the caller then runs the float32 kernels)
bool launch_wn_resskip_f16(const ConvArgs &a, hipStream_t stream) {
    const bool ok = a.ks == 1 && (a.h_init ? a.cin >= a.channels : a.cin == a.channels) && !a.last_layer && a.skip_ld > 0 && a.cout <= 384 &&
                    (!a.h_split || (a.h_split_ld % 8 == 0 && a.h_split_ld >= a.channels && a.h_split_ld <= a.cout + 32 && a.h_split_bstride % 4 == 0)) &&
                    a.gate_act != 3 &&          // glu: the layer's input is not bounded by 1
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 2 == 0 && a.skip_ld % 2 == 0 &&
                    a.cout % 2 == 0 && a.cout <= a.channels + a.skip_ld && (uintptr_t)a.x % 16 == 0 &&
                    (uintptr_t)a.w % 16 == 0 && (uintptr_t)a.h % 8 == 0 && (uintptr_t)a.skip % 8 == 0 &&
                    (!a.bias || (uintptr_t)a.bias % 8 == 0) && a.hs_bstride % 2 == 0 && a.h && a.skip && a.zeros;
    if (!ok) return false;
    ConvArgs r = a;
    r.m_tiles_per_item = (a.max_rows + RH_ROWS - 1) / RH_ROWS;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 16LL * ((r.m_tiles_total + 7) / 8);
    hipLaunchKernelGGL(wn_resskip_f16_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, r);
    return true;
}

}  // namespace mbx
