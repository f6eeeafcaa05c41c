// WaveNet dilated convolution (k = 3, dilation d) + conditioning + tanh*sigmoid in Winograd F(4,3) form.
//
// Layer: reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:305-321.  Four outputs y[t], y[t+d], y[t+2d], y[t+3d]
// per group from six products: with x0..x5 = h[t-d] .. h[t+4d]
//     v0 = 4 x0 - 5 x2 + x4          v1 = -4 x1 - 4 x2 + x3 + x4      v2 = 4 x1 - 4 x2 - x3 + x4
//     v3 = -2 x1 - x2 + 2 x3 + x4    v4 = 2 x1 - x2 - 2 x3 + x4       v5 = 4 x1 - 5 x3 + x5
//     m_j = v_j U_j,  U = G W  (G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]])
//     y[t] = m0+m1+m2+m3+m4   y[t+d] = m1-m2+2(m3-m4)   y[t+2d] = m1+m2+4(m3+m4)   y[t+3d] = m1-m2+8(m3-m4)+m5
// = 6 channel contractions per four outputs instead of 12.  Float32 on the exact fp32 MFMA; U is formed on the host in
// float64 (engine.pack_winograd4w_weights).
//
// What shapes this kernel: on gfx950 the fp32 MFMA runs at the fp32 vector rate and every vector instruction a wave
// issues takes matrix-pipe time away (measured, profiles/README.md: 3-4 cycles per v_fma, 11 per v_exp beside
// v_mfma_f32_16x16x4_f32).  The input combinations v_j are vector work proportional to (rows x channels) of a wave's
// tile, the MFMA work to (rows x channels x columns): so a wave owns FEW rows and MANY columns -- 16 groups
// (v_mfma_f32_16x16x4_f32) x all 64 weight columns of the block (4 column tiles: [16 tanh | 16 sigmoid] of the even
// gate channels and of the odd ones) x 6 products = 24 accumulator tiles of 4 registers.  Per 8-channel slice a wave
// issues 48 MFMAs (1536 matrix-pipe cycles) beside 24 vector instructions.  Lane n of the column tiles holds tanh and
// sigmoid column of gate channels 2n and 2n+1: the epilogue needs no cross-lane traffic, reads the conditioning as
// float2 and stores float2 (a wave instruction writes whole 128-byte rows).  The bias is the initial value of product
// 1's accumulators (m1 enters all four outputs with coefficient 1).
//
// Two block shapes (template):
//   <256, 1>  large launches: 256 consecutive output rows (64 groups; wave w owns groups 16 w .. 16 w + 15) x 32 gate
//             channels; K slices of 8 channels, two LDS stages (52.5 KB with the conditioning tile: 3 blocks per CU --
//             a third wave per SIMD fills matrix-pipe slots the other two leave: 1.40 -> 1.35 ms at batch 16 x 10 s;
//             a third stage at 2 blocks per CU measured the same as two).
//   <128, 2>  small launches (batch 1: a launch is only a few rounds of resident blocks, so what decides its time is how
//             finely the work divides over the 1024 SIMDs): 128 rows x 32 gate channels, waves = 2 row halves x 2 channel
//             halves: wave (rw, kh) contracts the channels 16 s + 8 kh .. + 7 of every double slice s, i.e. half of K, so
//             a 10 s utterance becomes 1250 blocks of 960 MFMAs per wave.  The two partial sums of a row half meet
//             through LDS before the epilogue; each of the two waves then finishes two of the four rows a lane holds.
//             Stages hold a double slice; two stages.
// Group Q of the block: q = Q / d, r = Q % d, t = m0 + 4 d q + r (d a power of two <= 16).
// Per 8-channel slice the block stages, through LDS-DMA:
//   A: activation rows [m0-16, m0+ROWS+16) x 8 channels in read order: row = m0 - d + d*m + b (b < d) lives in 32-byte
//      cell p = (m & 3)*PHASE + (m >> 2)*d + b (PHASE = ROWS/4 + 16), its 16-byte chunk c at 2*p + (c ^ ((p>>3)&1)).
//      Lane (r = lane & 15, kq = lane >> 4) reads the 8 bytes of channels 2 kq, 2 kq + 1 of cell (q & 3)*PHASE + Q +
//      (q >> 2)*d for each of its six rows q: consecutive lanes read consecutive cells, and with the chunk swizzle the
//      32 lanes of a ds_read_b64 half hit 64 different banks at every dilation.  MFMA step m (0, 1) of a slice
//      contracts channels {2 kq + m}.
//   B: 6 products x 8 channels x 64 columns, packed on the host in MFMA operand order [product j][channel parity e]
//      [lane][tanh step 0, tanh step 1, sigmoid step 0, sigmoid step 1]: one ds_read_b128 = the weight operands of four MFMAs
// LDS: stages (44 KB / 72 KB) + the conditioning rows of the block and the per-row interpolation table (8.5 KB / 5 KB)
// -> 3 / 2 blocks per CU.  The stage loop is unrolled by the number of stages so that every LDS address is a loop-invariant
// register plus an immediate.
// (Measured and rejected, batch 16 x 10 s: a fifth wave that issues all LDS-DMA requests, 1.50 ms against 1.39 ms per
// launch -- the requests then queue on one SIMD whose MFMA waves become the stragglers of every barrier; spreading a
// slice's requests over three phases instead of issuing them behind the barrier, and s_setprio: no change.)
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WW_HALO = 16;
constexpr int WW_BK = 8;
constexpr int WW_B_FLOATS = 6 * WW_BK * 64;            // 3072: packed weights of one 8-channel slice

template <int ROWS, int KSPLIT>
struct WwShape {
    static constexpr int AROWS = ROWS + 2 * WW_HALO;            // rows that can be needed
    static constexpr int PHASE = ROWS / 4 + WW_HALO;            // cells per phase (m & 3)
    static constexpr int CELLS = 4 * PHASE;                     // 32-byte cells of 8 channels
    static constexpr int A_FLOATS = CELLS * WW_BK;              // 2560 | 1536
    static constexpr int A_CHUNKS = A_FLOATS / 256;             // 1 KB LDS-DMA instructions per slice (A): 10 | 6
    static constexpr int SUB = A_FLOATS + WW_B_FLOATS;          // one 8-channel slice: A, B behind it
    static constexpr int STAGE = KSPLIT * SUB;                  // 5632 | 9216 floats
    static constexpr int NSTAGE = 2;
    static constexpr int ROW_WAVES = ROWS / 64;                 // 4 | 2
    static constexpr int B_INST = 3 * KSPLIT;                   // weight requests per wave and stage
    static constexpr int DMA_PER_STAGE = 3 + B_INST;            // 6 | 9
    static constexpr int COND_ROWS = ROWS == 256 ? 28 : 16;     // conditioning rows of 64 floats (cond_up >= 10)
    static constexpr int COND_CHUNKS = COND_ROWS / 4;           // 1 KB LDS-DMA requests: 7 | 4, dealt round-robin
    // behind the stages: conditioning tile; per block row lr: (float offset of its conditioning row) << 8 | phase u of
    // the interpolation; the interpolation weights w0[64], w1[64]
    static constexpr int COND = NSTAGE * STAGE;
    static constexpr int TAB = COND + COND_ROWS * 64;
    static constexpr int LERP = TAB + ROWS;
    static constexpr int LDS_FLOATS = LERP + 128;
    // <256,1>: 45056 + 7168 + 1024 + 512 = 53760 bytes -> 3 blocks per CU; <128,2>: 79 KB -> 2
    static constexpr int BLOCKS_PER_CU = LDS_FLOATS * 4 * 3 <= 160 * 1024 ? 3 : 2;
    static_assert(ROW_WAVES * KSPLIT == 4, "four waves per block");
};

__device__ __forceinline__ void ww_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset (no per-lane 64-bit address arithmetic)
__device__ __forceinline__ void ww_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

// tanh(zt) * sigmoid(zs) = (t - 1) / ((t + 1)(1 + s)), t = e^(2 zt), s = e^(-zs): two exponentials and one reciprocal
// (zt is clamped where tanh is 1 in float32, so t stays finite; s = inf gives 0, as it should)
__device__ __forceinline__ float ww_gate_act(float zt, float zs) {
    const float t = __builtin_amdgcn_exp2f(fminf(zt, 15.f) * 2.885390081777927f);
    const float sg = __builtin_amdgcn_exp2f(zs * -1.4426950408889634f);
    const float tp = t + 1.0f;
    return (t - 1.0f) * __builtin_amdgcn_rcpf(fmaf(sg, tp, tp));
}

__device__ __forceinline__ float2 ww_fma(float s, float2 a, float2 b) { return make_float2(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y)); }
__device__ __forceinline__ float2 ww_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 ww_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

#define WW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define WW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WW_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define WW_SG_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
template <int N>
using ww_int = std::integral_constant<int, N>;

template <int ROWS, int KSPLIT>
__global__ __launch_bounds__(256, (WwShape<ROWS, KSPLIT>::BLOCKS_PER_CU)) void wn_gate_winograd4w_kernel(ConvArgs p, int log2d) {
    using SH = WwShape<ROWS, KSPLIT>;
    constexpr int NSTAGE = SH::NSTAGE, STAGE = SH::STAGE, SUB = SH::SUB, A_FLOATS = SH::A_FLOATS, PHASE = SH::PHASE;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[SH::LDS_FLOATS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode (see decode_tile in conv_mfma.hip)
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    if (g_ >= p.m_tiles_total) return;
    const int b = g_ / p.m_tiles_per_item;
    const int mt = g_ - b * p.m_tiles_per_item;
    const int rows = p.n_frames ? p.n_frames[b] * p.rows_per_frame : p.max_rows;
    const int m0 = mt * ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave / KSPLIT, kh = wave % KSPLIT;       // row part and channel half of this wave
    const int r16 = lane & 15, kq = lane >> 4;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nk8 = (p.cin + WW_BK - 1) / WW_BK;            // 8-channel slices of the weight image
    const int nst = (nk8 + KSPLIT - 1) / KSPLIT;            // stage fills

    // ---- per-lane DMA sources (fixed for the whole kernel except the channel offset): byte offset of (row, chunk) from
    // the item's first element + validity bits (bit i: the row exists, bit 4 + i: the chunk is the upper half of the slice)
    unsigned a_voff[3];
    unsigned a_bits = 0;
    int a_inst[3], a_sub[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // <256,1>: 10 chunks; 0..7 are dealt round-robin, 8 and 9 are each written by two waves (same data), which keeps
        // the number of outstanding LDS-DMA instructions per stage the same for every wave (s_waitcnt vmcnt below)
        // <128,2>: 2 x 6 chunks, dealt round-robin
        const int ii = (ROWS == 256) ? (i < 2 ? wave + 4 * i : 8 + (wave & 1)) : wave + 4 * i;
        a_sub[i] = ii / SH::A_CHUNKS;
        a_inst[i] = ii - a_sub[i] * SH::A_CHUNKS;
        const int pos = a_inst[i] * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / PHASE, sidx = cell - phase * PHASE;
        const int m = 4 * (sidx >> log2d) + phase;
        const int row = (m << log2d) + (sidx & (d - 1)) + WW_HALO - d;       // staged row index, m0 - 16 + row = source
        const int src = m0 - WW_HALO + row;
        const int hi = (pos & 1) ^ ((cell >> 3) & 1);
        if (row < SH::AROWS && src >= 0 && src < rows) a_bits |= 1u << i;
        a_bits |= (unsigned)hi << (4 + i);
        a_voff[i] = 4u * (unsigned)(min(max(src, 0), rows - 1) * p.ldx + 8 * a_sub[i] + 4 * hi);
    }
    // interior blocks (every staged row exists, whole stage fills): uniform base + per-lane byte offset, no selects
    // (C = 340: every stage fill but the last one is whole, so only that one takes the masked path)
    const bool fast_rows = p.fast_dma && m0 >= WW_HALO && m0 + ROWS + WW_HALO <= rows;
    const int whole_fills = p.cin / (WW_BK * KSPLIT);
    const float *wtile = p.w + (long long)nt * nk8 * WW_B_FLOATS;
    const unsigned b_voff = 16u * (unsigned)lane;
    // LDS-DMA of stage fill st (channels 8 KSPLIT st ..) into a stage: 3 A + 3 KSPLIT B instructions per wave
    auto issue = [&](int st, int stage) {
        const int ci0 = st * WW_BK * KSPLIT;
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * STAGE);
        if (fast_rows && st < whole_fills) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                ww_lds_dma16_s(abase, a_voff[i], sdst + 4u * (unsigned)(a_sub[i] * SUB) + 1024u * (unsigned)a_inst[i]);
#pragma unroll
            for (int i = 0; i < SH::B_INST; ++i) {
                const int ii = wave + 4 * i;                              // 0 .. 12 KSPLIT - 1
                const int sub = ii / 12, k = ii - 12 * sub;
                ww_lds_dma16_s(wtile + (long long)(KSPLIT * st + sub) * WW_B_FLOATS + k * 256, b_voff,
                               sdst + 4u * (unsigned)(sub * SUB + A_FLOATS) + 1024u * (unsigned)k);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ci = ci0 + 8 * a_sub[i] + 4 * (int)((a_bits >> (4 + i)) & 1u);
            const bool ok = ((a_bits >> i) & 1u) & (ci < p.cin);
            const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff[i]);
            ww_lds_dma16(ok ? src : p.zeros, sdst + 4u * (unsigned)(a_sub[i] * SUB) + 1024u * (unsigned)a_inst[i]);
        }
#pragma unroll
        for (int i = 0; i < SH::B_INST; ++i) {
            const int ii = wave + 4 * i;
            const int sub = ii / 12, k = ii - 12 * sub;
            const int kt8 = KSPLIT * st + sub;
            const float *src = reinterpret_cast<const float *>(
                reinterpret_cast<const char *>(wtile + (long long)kt8 * WW_B_FLOATS + k * 256) + b_voff);
            ww_lds_dma16(kt8 < nk8 ? src : p.zeros, sdst + 4u * (unsigned)(sub * SUB + A_FLOATS) + 1024u * (unsigned)k);
        }
    };
    // ---- conditioning rows of this block (COND_ROWS x (32 tanh | 32 sigmoid) columns): requested first, so every later
    // wait for a stage covers them
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    {
        const int n2 = rows / cond_up;
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
#pragma unroll
        for (int i = 0; i < (SH::COND_CHUNKS + 3) / 4; ++i) {
            if (wave + 4 * i >= SH::COND_CHUNKS) break;
            const int pos = (wave + 4 * i) * 64 + lane;
            const int crow = pos >> 4, cq = pos & 15;
            const int chn = n0 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            ww_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                         lds_base + 4u * (unsigned)SH::COND + 1024u * (unsigned)(wave + 4 * i));
        }
    }
    // ---- prologue: every stage requested
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (s < nst) issue(s, s);

    // lane n of column tile (e, tanh | sigmoid) holds gate channel n0 + 2 n + e
    const bool ch_ok = n0 + 2 * r16 < C;                 // C is even: both channels of the lane exist or neither
    f32x4 acc[6][4];          // [product][column tile: 2 e + (0 tanh | 1 sigmoid)]
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // m1 enters y[t] .. y[t+3d] with coefficient 1: its accumulators (of channel half 0) start from the bias
            const float bv = (j == 1 && kh == 0 && p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][c][r] = bv;
        }
    {
        // block row lr = tid: conditioning row offset and interpolation weights (read in the epilogue)
        if (tid < ROWS) {
            const int row = m0 + tid;
            const int t2 = row / cond_up;
            const int u = row - t2 * cond_up;
            reinterpret_cast<int *>(lds + SH::TAB)[tid] = (((t2 - t2base) * 64) << 8) | u;
        }
        if (tid < 64) {
            lds[SH::LERP + tid] = tid < cond_up ? p.lerp_w0[tid] : 0.f;
            lds[SH::LERP + 64 + tid] = tid < cond_up ? p.lerp_w1[tid] : 0.f;
        }
    }

    // A operand: group of this lane Q = 16*rw + r16 -> t = m0 + 4 d (Q >> log2d) + (Q & (d-1)); channels 2 kq, 2 kq + 1
    const int grp = 16 * rw + r16;
    const float *xptr[6];     // LDS addresses (stage 0, this wave's channel half) of h[t-d] .. h[t+4d]
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int cell = (q & 3) * PHASE + grp + ((q >> 2) << log2d);
        xptr[q] = lds + kh * SUB + 8 * cell + 4 * ((kq >> 1) ^ ((cell >> 3) & 1)) + 2 * (kq & 1);
    }
    const float *bptr = lds + kh * SUB + A_FLOATS + lane * 4;

    float2 x[6];              // raw activation rows of the slice whose combinations are being formed
    float2 u[2];              // input combination of product j in u[j & 1]
    float4 bw[2][2];          // weights of product j in bw[j & 1][channel parity e]
    float2 ca, cb;            // shared sub-expressions (-4 x2 + x4, -4 x1 + x3), then (x4 - x2, x3 - x1)

    auto load_x = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float2 *>(xptr[q] + S * STAGE);
    };
    auto load_b = [&](auto sc, auto jc) {
        constexpr int S = decltype(sc)::value, J = decltype(jc)::value;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            bw[J & 1][e] = *reinterpret_cast<const float4 *>(bptr + S * STAGE + (J * 2 + e) * 256);
    };
    // product j: 2 steps x 4 column tiles
    auto mfma8 = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f32x4 *ac = acc[J];
        const float2 uu = u[J & 1];
        const float4 b0 = bw[J & 1][0], b1 = bw[J & 1][1];
        ac[0] = WW_MFMA(uu.x, b0.x, ac[0]);
        ac[1] = WW_MFMA(uu.x, b0.z, ac[1]);
        ac[2] = WW_MFMA(uu.x, b1.x, ac[2]);
        ac[3] = WW_MFMA(uu.x, b1.z, ac[3]);
        ac[0] = WW_MFMA(uu.y, b0.y, ac[0]);
        ac[1] = WW_MFMA(uu.y, b0.w, ac[1]);
        ac[2] = WW_MFMA(uu.y, b1.y, ac[2]);
        ac[3] = WW_MFMA(uu.y, b1.w, ac[3]);
    };
    // input combination of product J from the rows in x (24 vector instructions per slice)
    auto comb = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (J == 0) u[0] = ww_fma(4.f, x[0], ww_fma(-5.f, x[2], x[4]));
        if (J == 1) {
            ca = ww_fma(-4.f, x[2], x[4]);
            cb = ww_fma(-4.f, x[1], x[3]);
            u[1] = ww_add(ca, cb);
        }
        if (J == 2) u[0] = ww_sub(ca, cb);
        if (J == 3) {
            ca = ww_sub(x[4], x[2]);
            cb = ww_sub(x[3], x[1]);
            u[1] = ww_fma(2.f, cb, ca);
        }
        if (J == 4) u[0] = ww_fma(-2.f, cb, ca);
        if (J == 5) u[1] = ww_fma(4.f, x[1], ww_fma(-5.f, x[3], x[5]));
    };
    // One stage fill = six phases of 8 MFMAs (one product each).  While product j is multiplied, the weights of product
    // j+1 are requested from LDS and its input combination is formed between the MFMAs.  The barrier that publishes
    // fill st+1 sits in front of the last product: every wave has requested all LDS operands of fill st by then, so the
    // stage is free for fill st+NSTAGE.  In: u[0], bw[0] of product 0 of this fill.  Out: those of the next one.
    auto phase = [&](auto sc, auto jc) {
        constexpr int J = decltype(jc)::value;
        load_b(sc, ww_int<J + 1>());
        WW_FENCE();
        comb(ww_int<J + 1>());
        mfma8(jc);
#pragma unroll
        for (int i = 0; i < 6; ++i) { WW_SG_MFMA(1); WW_SG_VALU(1); }
        WW_SG_MFMA(2);
        WW_FENCE();
    };
    auto fill = [&](auto sc, int st) {
        constexpr int S = decltype(sc)::value;
        ww_int<(S + 1) % NSTAGE> ns;
        phase(sc, ww_int<0>());
        phase(sc, ww_int<1>());
        phase(sc, ww_int<2>());
        phase(sc, ww_int<3>());
        phase(sc, ww_int<4>());
        // ---- product 5 behind the barrier; fill st+1 must have landed: only fill st+2 (three stages) may be in flight
        if (NSTAGE == 3 && st + 2 < nst) {
            if (SH::DMA_PER_STAGE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (st + NSTAGE < nst) issue(st + NSTAGE, S);
        load_x(ns);
        load_b(ns, ww_int<0>());
        WW_FENCE();
        mfma8(ww_int<5>());
        WW_FENCE();
        comb(ww_int<0>());
        WW_FENCE();
    };

    // ---- the first fill has landed (the launchers guarantee nst >= NSTAGE)
    if (NSTAGE == 3) {
        if (SH::DMA_PER_STAGE == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    } else {
        if (SH::DMA_PER_STAGE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    }
    __syncthreads();
    load_x(ww_int<0>());
    load_b(ww_int<0>(), ww_int<0>());
    comb(ww_int<0>());
    {
        int st = 0;
        if constexpr (NSTAGE == 3) {
            for (; st + 3 <= nst; st += 3) {
                fill(ww_int<0>(), st);
                fill(ww_int<1>(), st + 1);
                fill(ww_int<2>(), st + 2);
            }
            if (st < nst) {
                fill(ww_int<0>(), st);
                if (st + 1 < nst) fill(ww_int<1>(), st + 1);
            }
        } else {
            for (; st + 2 <= nst; st += 2) {
                fill(ww_int<0>(), st);
                fill(ww_int<1>(), st + 1);
            }
            if (st < nst) fill(ww_int<0>(), st);
        }
    }

    // ---- <128,2>: the two channel halves of a row part meet: wave kh keeps the rows v in {2 kh, 2 kh + 1} of every
    // accumulator tile and hands the other two to its partner (through the stage memory: 4 waves x 12 KB)
    constexpr int NV = 4 / KSPLIT;                   // rows per accumulator tile this wave finishes
    const int v0 = KSPLIT == 2 ? 2 * kh : 0;
    if (KSPLIT == 2) {
        __syncthreads();                             // all LDS operand reads are done
        float2 *mine = reinterpret_cast<float2 *>(lds) + wave * 1536 + lane;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                mine[(j * 4 + c) * 64] = kh ? make_float2(acc[j][c][0], acc[j][c][1]) : make_float2(acc[j][c][2], acc[j][c][3]);
        __syncthreads();
        const float2 *theirs = reinterpret_cast<const float2 *>(lds) + (wave ^ 1) * 1536 + lane;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float2 t = theirs[(j * 4 + c) * 64];
                // the kept rows move to registers 0, 1 of the tile
                acc[j][c][0] = (kh ? acc[j][c][2] : acc[j][c][0]) + t.x;
                acc[j][c][1] = (kh ? acc[j][c][3] : acc[j][c][1]) + t.y;
            }
    }

    // ---- epilogue: combine the six products, add the conditioning, gate, store the four outputs of the group
    const float *cl = lds + SH::COND;
    float *obase = p.out + (long long)b * p.out_bstride + n0 + 2 * r16;
    const float *clane = cl + 2 * r16;
    // everything up to the store is unconditional (every table and conditioning address is valid), so the LDS reads of
    // all rows can be in flight together; only the store is predicated
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
        const int gi = 16 * rw + 4 * kq + v0 + vi;                               // group held by this register
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));         // its first row, relative to m0
        float y[4][4];                                                           // [column tile][output]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s12 = acc[1][c][vi] + acc[2][c][vi], d12 = acc[1][c][vi] - acc[2][c][vi];
            const float s34 = acc[3][c][vi] + acc[4][c][vi], d34 = acc[3][c][vi] - acc[4][c][vi];
            y[c][0] = (acc[0][c][vi] + s12) + s34;
            y[c][1] = fmaf(2.f, d34, d12);
            y[c][2] = fmaf(4.f, s34, s12);
            y[c][3] = fmaf(8.f, d34, d12) + acc[5][c][vi];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int lr = lr0 + (o << log2d);
            const int row = m0 + lr;
            const int e = reinterpret_cast<const int *>(lds + SH::TAB)[lr];
            const float2 w = make_float2(lds[SH::LERP + (e & 255)], lds[SH::LERP + 64 + (e & 255)]);
            const float *c0 = clane + (e >> 8);
            const float2 ct0 = *reinterpret_cast<const float2 *>(c0), ct1 = *reinterpret_cast<const float2 *>(c0 + 64);
            const float2 cs0 = *reinterpret_cast<const float2 *>(c0 + 32), cs1 = *reinterpret_cast<const float2 *>(c0 + 96);
            float2 res;
            res.x = ww_gate_act(y[0][o] + (ct0.x * w.x + ct1.x * w.y), y[1][o] + (cs0.x * w.x + cs1.x * w.y));
            res.y = ww_gate_act(y[2][o] + (ct0.y * w.x + ct1.y * w.y), y[3][o] + (cs0.y * w.x + cs1.y * w.y));
            if (ch_ok && row < rows) *reinterpret_cast<float2 *>(obase + (long long)row * p.ldo) = res;
        }
    }
}

// a.w must point at the host-packed F(4,3) weights (ceil(C/32), ceil(C/8), 3072) of engine.pack_winograd4w_weights;
// small = the 128-row shape whose waves split the input channels; returns false if the layer does not fit
bool launch_wn_gate_winograd4w(const ConvArgs &a, bool small, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const int rows_blk = small ? 128 : 256;
    const int nk8 = (a.cin + WW_BK - 1) / WW_BK;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= WW_HALO && nk8 >= 4 && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros &&
                    a.cond && (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up >= 1 &&
                    a.cond_up <= 64 && (rows_blk + a.cond_up - 2) / a.cond_up + 2 <= (small ? 16 : 28) && a.lerp_w0 && a.lerp_w1 && a.max_rows < (1 << 24);
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = (long long)a.max_rows * a.ldx * 4 < (1LL << 32);
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + rows_blk - 1) / rows_blk;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    if (small) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<128, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    else hipLaunchKernelGGL((wn_gate_winograd4w_kernel<256, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    return true;
}

}  // namespace mbx
