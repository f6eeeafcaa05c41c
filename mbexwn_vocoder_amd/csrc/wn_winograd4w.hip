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
// Two block shapes:
//   wn_gate_winograd4w_kernel  256 consecutive output rows (64 groups; wave w owns groups 16 w .. 16 w + 15) x 32 gate
//             channels; K slices of 8 channels, two LDS stages (52.5 KB with the conditioning tile: 3 blocks per CU --
//             a third wave per SIMD fills matrix-pipe slots the other two leave: 1.40 -> 1.35 ms at batch 16 x 10 s;
//             a third stage at 2 blocks per CU measured the same as two).
//   wn_gate_winograd4p_kernel  128 rows x 32 gate channels, waves = 2 row halves x 2 PRODUCT halves, for launches of a
//             few blocks per CU (one utterance); same bits; described at the kernel below.
//   (Round 1/2 also had a 128-row shape whose waves split the input CHANNELS and summed the two halves in front of the
//   epilogue: 118 us per launch for a 10 s utterance against 110 us for the 256-row shape and 100 us for the
//   product-split one; removed.)
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
// LDS: stages (44 KB) + the conditioning rows of the block and the per-row interpolation table (8.5 KB) -> 3 blocks per
// CU.  The stage loop is unrolled by the number of stages so that every LDS address is a loop-invariant register plus an
// immediate.
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

// CR = conditioning rows the block's tile holds: 28 for cond_up >= 10 (the canonical models: 3 blocks per CU), 56 for
// cond_up >= 5 (block-runner geometries whose first block runs at half the sub-band rate: 60.9 KB, 2 blocks per CU)
template <int CR>
struct WwShape {
    static constexpr int ROWS = 256;
    static constexpr int AROWS = ROWS + 2 * WW_HALO;            // rows that can be needed
    static constexpr int PHASE = ROWS / 4 + WW_HALO;            // cells per phase (m & 3)
    static constexpr int CELLS = 4 * PHASE;                     // 32-byte cells of 8 channels
    static constexpr int A_FLOATS = CELLS * WW_BK;              // 2560
    static constexpr int A_CHUNKS = A_FLOATS / 256;             // 1 KB LDS-DMA instructions per slice (A): 10
    static constexpr int STAGE = A_FLOATS + WW_B_FLOATS;        // one 8-channel slice: A, B behind it: 5632 floats
    static constexpr int NSTAGE = 2;
    static constexpr int DMA_PER_STAGE = 6;                     // 3 (A) + 3 (B) requests per wave
    static constexpr int COND_ROWS = CR;                        // conditioning rows of 64 floats
    static constexpr int COND_CHUNKS = COND_ROWS / 4;           // 1 KB LDS-DMA requests: 7 | 14, dealt round-robin
    // behind the stages: conditioning tile; per block row lr: (float offset of its conditioning row) << 8 | phase u of
    // the interpolation; the interpolation weights w0[64], w1[64]
    static constexpr int COND = NSTAGE * STAGE;
    static constexpr int TAB = COND + COND_ROWS * 64;
    static constexpr int LERP = TAB + ROWS;
    static constexpr int LDS_FLOATS = LERP + 128;               // 45056 + 7168 + 1024 + 512 = 53760 bytes -> 3 blocks per CU
};

__device__ __forceinline__ void ww_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset (no per-lane 64-bit address arithmetic)
__device__ __forceinline__ void ww_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
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

// GA: the gate activation as a compile-time constant (0 = gtu, the canonical models) or -1 = ConvArgs::gate_act at run time.
// With the run-time kind every wn_gate_act call of the epilogue sits behind wave-uniform branches, which cut the epilogue
// into 32 scheduling regions -- each with its own LDS round trip (s_waitcnt lgkmcnt(0)) in front of a few dozen
// instructions: 32 000 cycles per block (round 5, in-kernel stamps: 14 of a block's 94 us).
//
// VS (round 6): dilations above 16.  WaveNetAE's own default depth is 12 layers without a dilation cycle: d = 1 .. 2048
// (reference custom_AE_layers.py:120-123, 229-233).  A convolution with dilation d = 16 s over the rows of an item IS the
// convolution with dilation 16 over each of its s interleaved sub-sequences (rows r, r + s, r + 2 s ...): the launcher turns
// every item into s virtual items whose row stride is s times the real one (ConvArgs::vstride) and this kernel runs them
// at log2d = 4.  An LDS-DMA lane fetches 16 bytes of one row either way, so the staging costs the same; what differs is the
// conditioning: a block's 256 virtual rows span 256 s real rows (~26 s conditioning rows: no LDS tile holds them), so the
// epilogue reads its two conditioning rows per output from global memory (L2) instead, eight float2 per output pair.
// VS == 2: virtual items of at most 128 rows (d = 2048 at 10 s: 125 rows per sub-sequence) would leave half of a 256-row
// block empty: the block takes TWO neighbouring sub-sequences instead -- waves 0, 1 (rows 0 .. 127 of the block) those of
// sub-sequence `strip`, waves 2, 3 those of `strip + 1`.  The staged cells hold 20 rows-of-sixteen (m = 0 .. 19: 320 cells,
// all the LDS stage has): m < 10 the rows [-16, 144) of the first, m >= 10 of the second sub-sequence.  Same groups, same
// sums: the same bits as the other shapes.
template <int CR, int GA, int VS = 0>
__global__ __launch_bounds__(256, CR <= 28 ? 3 : 2) void wn_gate_winograd4w_kernel(ConvArgs p, int log2d) {
    using SH = WwShape<CR>;
    constexpr int ROWS = SH::ROWS, NSTAGE = SH::NSTAGE, STAGE = SH::STAGE, A_FLOATS = SH::A_FLOATS, PHASE = SH::PHASE;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[SH::LDS_FLOATS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode (see decode_tile in conv_mfma.hip)
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    if (g_ >= p.m_tiles_total) return;
    const int bv = g_ / p.m_tiles_per_item;                 // (virtual) item
    const int mt = g_ - bv * p.m_tiles_per_item;
    const int vs = VS ? p.vstride : 1;
    const int vper = VS == 2 ? vs / 2 : vs;                 // (pairs of) virtual items per item
    const int b = VS ? bv / vper : bv;
    const int strip = VS ? (bv - b * vper) * (VS == 2 ? 2 : 1) : 0;          // first real row of the (first) virtual item
    const int rows_item = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int rows = VS ? (rows_item - strip + vs - 1) / vs : rows_item;      // rows strip, strip + vs, ... < rows_item
    const int rows2 = VS == 2 ? max(rows_item - strip - 1 + vs - 1, 0) / vs : 0;      // ... of the second sub-sequence (VS == 2)
    const int m0 = mt * ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave;                                    // row part of this wave
    const int r16 = lane & 15, kq = lane >> 4;
    // rows are addressed relative to the block's first staged row: 32-bit byte offsets never leave the block's window,
    // however long the item is
    const int xrow0 = max(m0 - WW_HALO, 0);
    const int ldxv = VS ? p.ldx * vs : p.ldx;               // floats between (virtual) rows
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)strip * p.ldx + (long long)xrow0 * ldxv;
    const int nk8 = (p.cin + WW_BK - 1) / WW_BK;            // 8-channel slices of the weight image = stage fills
    const int nst = nk8;

    // ---- per-lane DMA sources (fixed for the whole kernel except the channel offset): byte offset of (row, chunk) from
    // the item's first element + validity bits (bit i: the row exists, bit 4 + i: the chunk is the upper half of the slice)
    unsigned a_voff[3];
    unsigned a_bits = 0;
    int a_inst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // 10 chunks; 0..7 are dealt round-robin, 8 and 9 are each written by two waves (same data), which keeps the
        // number of outstanding LDS-DMA instructions per stage the same for every wave (s_waitcnt vmcnt below)
        a_inst[i] = i < 2 ? wave + 4 * i : 8 + (wave & 1);
        const int pos = a_inst[i] * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / PHASE, sidx = cell - phase * PHASE;
        const int m = 4 * (sidx >> log2d) + phase;
        const int hi = (pos & 1) ^ ((cell >> 3) & 1);
        if (VS == 2) {
            // (d = 16, m0 = 0) m < 10: rows -16 + 16 m + b of the first sub-sequence, m >= 10: of the second one (one real row on)
            const int sub = m >= 10 ? 1 : 0;
            const int src = 16 * (m - 10 * sub) + (sidx & 15) - WW_HALO;
            const int rs = sub ? rows2 : rows;
            if (m < 20 && src >= 0 && src < rs) a_bits |= 1u << i;
            a_bits |= (unsigned)hi << (4 + i);
            a_voff[i] = 4u * (unsigned)(min(max(src, 0), max(rs - 1, 0)) * ldxv + sub * p.ldx + 4 * hi);
        } else {
            const int row = (m << log2d) + (sidx & (d - 1)) + WW_HALO - d;       // staged row index, m0 - 16 + row = source
            const int src = m0 - WW_HALO + row;
            if (row < SH::AROWS && src >= 0 && src < rows) a_bits |= 1u << i;
            a_bits |= (unsigned)hi << (4 + i);
            a_voff[i] = 4u * (unsigned)((min(max(src, 0), rows - 1) - xrow0) * ldxv + 4 * hi);
        }
    }
    // interior blocks (every staged row exists, whole stage fills): uniform base + per-lane byte offset, no selects
    // (C = 340: every stage fill but the last one is whole, so only that one takes the masked path)
    const bool fast_rows = p.fast_dma && m0 >= WW_HALO && m0 + ROWS + WW_HALO <= rows;
    const int whole_fills = p.cin / WW_BK;
    const float *wtile = p.w + (long long)nt * nk8 * WW_B_FLOATS;
    const unsigned b_voff = 16u * (unsigned)lane;
    // LDS-DMA of slice st into a stage: 3 A + 3 B instructions per wave
    auto issue = [&](int st, int stage) {
        const int ci0 = st * WW_BK;
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * STAGE);
        if (fast_rows && st < whole_fills) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < 3; ++i) ww_lds_dma16_s(abase, a_voff[i], sdst + 1024u * (unsigned)a_inst[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int ci = ci0 + 4 * (int)((a_bits >> (4 + i)) & 1u);
                const bool ok = ((a_bits >> i) & 1u) & (ci < p.cin);
                const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff[i]);
                ww_lds_dma16(ok ? src : p.zeros, sdst + 1024u * (unsigned)a_inst[i]);
            }
        }
        const float *bsrc = wtile + (long long)st * WW_B_FLOATS;         // st < nk8: the image has nk8 slices
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = wave + 4 * i;
            ww_lds_dma16_s(bsrc + k * 256, b_voff, sdst + 4u * (unsigned)A_FLOATS + 1024u * (unsigned)k);
        }
    };
    // ---- conditioning rows of this block (COND_ROWS x (32 tanh | 32 sigmoid) columns): requested first, so every later
    // wait for a stage covers them
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    if (!VS) {
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
            // m1 enters y[t] .. y[t+3d] with coefficient 1: its accumulators start from the bias
            const float bv = (j == 1 && p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][c][r] = bv;
        }
    {
        // block row lr = tid: conditioning row offset and interpolation weights (read in the epilogue)
        if (tid < ROWS) {
            const int row = VS == 2 ? strip + (tid >> 7) + vs * (tid & 127)    // real row: the block's second half = second sub-sequence
                                    : VS ? strip + vs * (m0 + tid) : m0 + tid;
            const int t2 = row / cond_up;
            const int u = row - t2 * cond_up;
            // VS: the conditioning row itself (< 2^24 / cond_up), read from global memory in the epilogue
            reinterpret_cast<unsigned *>(lds + SH::TAB)[tid] = VS ? ((unsigned)t2 << 8) | (unsigned)u
                                                                  : (unsigned)((((t2 - t2base) * 64) << 8) | u);
        }
        if (tid < 64) {
            lds[SH::LERP + tid] = tid < cond_up ? p.lerp_w0[tid] : 0.f;
            lds[SH::LERP + 64 + tid] = tid < cond_up ? p.lerp_w1[tid] : 0.f;
        }
    }

    // A operand: group of this lane Q = 16*rw + r16 -> t = m0 + 4 d (Q >> log2d) + (Q & (d-1)); channels 2 kq, 2 kq + 1
    const int grp = 16 * rw + r16;
    const float *xptr[6];     // LDS addresses (stage 0) of h[t-d] .. h[t+4d]
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        // row-of-sixteen m = 4 rw + q of the staged window (VS == 2, second sub-sequence: its rows start at m = 10)
        const int m = 4 * rw + q + ((VS == 2 && rw >= 2) ? 2 : 0);
        const int cell = VS == 2 ? (m & 3) * PHASE + ((m >> 2) << 4) + r16 : (q & 3) * PHASE + grp + ((q >> 2) << log2d);
        xptr[q] = lds + 8 * cell + 4 * ((kq >> 1) ^ ((cell >> 3) & 1)) + 2 * (kq & 1);
    }
    const float *bptr = lds + A_FLOATS + lane * 4;

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
        // ---- product 5 behind the barrier; fill st+1 must have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

    // ---- the first fill has landed (the launcher guarantees nst >= NSTAGE); the second one may still be in flight
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    load_x(ww_int<0>());
    load_b(ww_int<0>(), ww_int<0>());
    comb(ww_int<0>());
    {
        int st = 0;
        for (; st + 2 <= nst; st += 2) {
            fill(ww_int<0>(), st);
            fill(ww_int<1>(), st + 1);
        }
        if (st < nst) fill(ww_int<0>(), st);
    }

    // ---- epilogue: combine the six products, add the conditioning, gate, store the four outputs of the group
    const float *cl = lds + SH::COND;
    float *obase = p.out + (long long)b * p.out_bstride + n0 + 2 * r16;
    const float *clane = cl + 2 * r16;
    const float *gcond = p.cond + (long long)b * p.cond_bstride + n0 + 2 * r16;      // (VS)
    // Round 5 (read off the ISA): with the predicated store behind every output the compiler kept each output in a region
    // of its own -- table entry, wait, six conditioning reads, wait, arithmetic, store: 48 serial LDS round trips per block
    // under the LDS traffic of the co-resident blocks' K loops.  Now: all 16 table entries of the lane are requested first;
    // per register vi the 24 conditioning / weight reads of its four outputs are requested together, the four results are
    // formed outside any branch (the empty asm keeps them there) and only then stored.  Same arithmetic, same bits.
    int etab[4][4];
#pragma unroll
    for (int vi = 0; vi < 4; ++vi) {
        const int gi = 16 * rw + 4 * kq + vi;                                    // group held by this register
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));         // its first row, relative to m0
#pragma unroll
        for (int o = 0; o < 4; ++o) etab[vi][o] = reinterpret_cast<const int *>(lds + SH::TAB)[lr0 + (o << log2d)];
    }
#pragma unroll
    for (int vi = 0; vi < 4; ++vi) {
        const int gi = 16 * rw + 4 * kq + vi;
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));
        float2 w[4], ct0[4], ct1[4], cs0[4], cs1[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int e = etab[vi][o];
            w[o] = make_float2(lds[SH::LERP + (e & 255)], lds[SH::LERP + 64 + (e & 255)]);
            if (VS) {
                // the two conditioning rows of this output, from global memory: [C tanh | C sigmoid] per row, the lane's
                // channel pair; rows clamp at the item's last conditioning row like the LDS tile's (edge replication)
                const int n2 = rows_item / cond_up;
                const int t2 = (int)((unsigned)e >> 8);
                const bool live = ch_ok && (VS == 2 ? ((lr0 + (o << log2d)) & 127) < (rw >= 2 ? rows2 : rows) : m0 + lr0 + (o << log2d) < rows);
                const float *g0 = live ? gcond + (long long)min(t2, n2 - 1) * (2 * C) : p.zeros;
                const float *g1 = live ? gcond + (long long)min(t2 + 1, n2 - 1) * (2 * C) : p.zeros;
                const int so = live ? C : 0;
                ct0[o] = *reinterpret_cast<const float2 *>(g0);
                ct1[o] = *reinterpret_cast<const float2 *>(g1);
                cs0[o] = *reinterpret_cast<const float2 *>(g0 + so);
                cs1[o] = *reinterpret_cast<const float2 *>(g1 + so);
            } else {
                const float *c0 = clane + (e >> 8);
                ct0[o] = *reinterpret_cast<const float2 *>(c0);
                ct1[o] = *reinterpret_cast<const float2 *>(c0 + 64);
                cs0[o] = *reinterpret_cast<const float2 *>(c0 + 32);
                cs1[o] = *reinterpret_cast<const float2 *>(c0 + 96);
            }
        }
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
        float2 res[4];
        const int kind = GA < 0 ? p.gate_act : GA;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            // (the interpolation as an explicit fma: both block shapes must round it the same way)
            res[o].x = wn_gate_act(kind, y[0][o] + fmaf(ct0[o].x, w[o].x, ct1[o].x * w[o].y), y[1][o] + fmaf(cs0[o].x, w[o].x, cs1[o].x * w[o].y));
            res[o].y = wn_gate_act(kind, y[2][o] + fmaf(ct0[o].y, w[o].x, ct1[o].y * w[o].y), y[3][o] + fmaf(cs0[o].y, w[o].x, cs1[o].y * w[o].y));
            asm volatile("" : "+v"(res[o].x), "+v"(res[o].y));
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int row = VS == 2 ? (lr0 + (o << log2d)) & 127 : m0 + lr0 + (o << log2d);      // (virtual) row of the output
            const int sub = (VS == 2 && rw >= 2) ? 1 : 0;
            const long long orow = VS ? (long long)strip + sub + (long long)vs * row : row;        // real row
            if (ch_ok && row < (sub ? rows2 : rows)) *reinterpret_cast<float2 *>(obase + orow * p.ldo) = res[o];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// <128 rows, products split>: the shape for launches of a few blocks per CU (one utterance).  Such a launch is bound by
// how evenly its wave tiles divide over the 1 024 SIMDs: a 10 s utterance is 2 500 wave tiles of 16 groups x 64 columns
// = 2.44 per SIMD, so a third of the SIMDs work 3 tiles while the rest wait (0.82).  Here a 128-row block has waves = 2
// row halves x 2 PRODUCT halves: wave (rw, ph) multiplies the products 3 ph .. 3 ph + 2 of its 16 groups over ALL input
// channels -- half a wave tile of work, 5 000 units = 4.88 per SIMD (0.98) -- and the two product halves of a row half
// meet once, in front of the epilogue: the wave with m0, m1, m2 hands (m1 + m2, m1 - m2) over and finishes y[t], y[t+d],
// the wave with m3, m4, m5 hands (m3 + m4, m3 - m4) over and finishes y[t+2d], y[t+3d].  Every sum is formed in the order
// of the 256-row shape, so both shapes give the SAME bits.  All four waves read the same stage (A: 6 KB, B: 12 KB per 8-channel slice, two stages):
// 41.5 KB of LDS with the conditioning tile -> 3 blocks per CU.
struct WpShape {
    static constexpr int ROWS = 128;
    static constexpr int AROWS = ROWS + 2 * WW_HALO;
    static constexpr int PHASE = ROWS / 4 + WW_HALO;            // 48
    static constexpr int CELLS = 4 * PHASE;                     // 192
    static constexpr int A_FLOATS = CELLS * WW_BK;              // 1536
    static constexpr int A_CHUNKS = A_FLOATS / 256;             // 6
    static constexpr int STAGE = A_FLOATS + WW_B_FLOATS;        // 4608 floats = 18 KB
    static constexpr int NSTAGE = 2;                            // (three stages at two blocks per CU measured 111 against 100 us)
    static constexpr int DMA_PER_STAGE = 5;                     // 2 (A: chunks 4, 5 are requested twice) + 3 (B)
    static constexpr int COND_ROWS = 16;
    static constexpr int COND = NSTAGE * STAGE;
    static constexpr int TAB = COND + COND_ROWS * 64;
    static constexpr int LERP = TAB + ROWS;
    static constexpr int LDS_FLOATS = LERP + 128;               // 9216 + 1024 + 128 + 128 = 10496 floats = 41 984 bytes
};

template <int J>
__device__ __forceinline__ void wp_comb(int PH, const float2 (&x)[6], float2 (&u)[2], float2 &ca, float2 &cb) {
    // product 3 PH + J of this wave (PH is wave-uniform: a scalar branch); the combination goes to u[J & 1]
    constexpr int j = J;
    if (PH == 0) {
        if (j == 0) u[0] = ww_fma(4.f, x[0], ww_fma(-5.f, x[2], x[4]));
        if (j == 1) {
            ca = ww_fma(-4.f, x[2], x[4]);
            cb = ww_fma(-4.f, x[1], x[3]);
            u[1] = ww_add(ca, cb);
        }
        if (j == 2) u[0] = ww_sub(ca, cb);
    } else {
        if (j == 0) {
            ca = ww_sub(x[4], x[2]);
            cb = ww_sub(x[3], x[1]);
            u[0] = ww_fma(2.f, cb, ca);
        }
        if (j == 1) u[1] = ww_fma(-2.f, cb, ca);
        if (j == 2) u[0] = ww_fma(4.f, x[1], ww_fma(-5.f, x[3], x[5]));
    }
}

// VS: virtual items of row stride ConvArgs::vstride (dilations above 16), see wn_gate_winograd4w_kernel
template <int GA, int VS = 0>
__global__ __launch_bounds__(256, 3) void wn_gate_winograd4p_kernel(ConvArgs p, int log2d) {
    using SH = WpShape;
    constexpr int ROWS = SH::ROWS, NSTAGE = SH::NSTAGE, STAGE = SH::STAGE, A_FLOATS = SH::A_FLOATS, PHASE = SH::PHASE;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[SH::LDS_FLOATS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    if (g_ >= p.m_tiles_total) return;
    const int bv = g_ / p.m_tiles_per_item;                 // (virtual) item
    const int mt = g_ - bv * p.m_tiles_per_item;
    const int vs = VS ? p.vstride : 1;
    const int b = VS ? bv / vs : bv;
    const int strip = VS ? bv - b * vs : 0;
    const int rows_item = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int rows = VS ? (rows_item - strip + vs - 1) / vs : rows_item;
    const int m0 = mt * ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave >> 1, ph = wave & 1;                // row half and product half of this wave
    const int r16 = lane & 15, kq = lane >> 4;
    // rows are addressed relative to the block's first staged row: 32-bit byte offsets never leave the block's window,
    // however long the item is
    const int xrow0 = max(m0 - WW_HALO, 0);
    const int ldxv = VS ? p.ldx * vs : p.ldx;
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)strip * p.ldx + (long long)xrow0 * ldxv;
    const int nk8 = (p.cin + WW_BK - 1) / WW_BK;

    // ---- per-lane DMA sources of the activation rows (see wn_gate_winograd4w_kernel)
    unsigned a_voff[2];
    unsigned a_bits = 0;
    int a_inst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_inst[i] = i == 0 ? wave : 4 + (wave & 1);         // chunks 4, 5: written by two waves each (same data)
        const int pos = a_inst[i] * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / PHASE, sidx = cell - phase * PHASE;
        const int m = 4 * (sidx >> log2d) + phase;
        const int row = (m << log2d) + (sidx & (d - 1)) + WW_HALO - d;
        const int src = m0 - WW_HALO + row;
        const int hi = (pos & 1) ^ ((cell >> 3) & 1);
        if (row < SH::AROWS && src >= 0 && src < rows) a_bits |= 1u << i;
        a_bits |= (unsigned)hi << (4 + i);
        a_voff[i] = 4u * (unsigned)((min(max(src, 0), rows - 1) - xrow0) * ldxv + 4 * hi);
    }
    const bool fast_rows = p.fast_dma && m0 >= WW_HALO && m0 + ROWS + WW_HALO <= rows;
    const int whole_fills = p.cin / WW_BK;
    const float *wtile = p.w + (long long)nt * nk8 * WW_B_FLOATS;
    const unsigned b_voff = 16u * (unsigned)lane;
    auto issue = [&](int st, int stage) {
        const int ci0 = st * WW_BK;
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * STAGE);
        if (fast_rows && st < whole_fills) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < 2; ++i) ww_lds_dma16_s(abase, a_voff[i], sdst + 1024u * (unsigned)a_inst[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ci = ci0 + 4 * (int)((a_bits >> (4 + i)) & 1u);
                const bool ok = ((a_bits >> i) & 1u) & (ci < p.cin);
                const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff[i]);
                ww_lds_dma16(ok ? src : p.zeros, sdst + 1024u * (unsigned)a_inst[i]);
            }
        }
        const float *bsrc = wtile + (long long)st * WW_B_FLOATS;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = wave + 4 * i;
            ww_lds_dma16_s(bsrc + k * 256, b_voff, sdst + 4u * (unsigned)A_FLOATS + 1024u * (unsigned)k);
        }
    };
    // ---- conditioning rows of this block (16 x (32 tanh | 32 sigmoid) columns): one request per wave, in front of the stages
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    if (!VS) {
        const int n2 = rows / cond_up;
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
        const int pos = wave * 64 + lane;
        const int crow = pos >> 4, cq = pos & 15;
        const int chn = n0 + 4 * (cq & 7);
        const int t = min(t2base + crow, n2 - 1);
        ww_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                     lds_base + 4u * (unsigned)SH::COND + 1024u * (unsigned)wave);
    }
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE; ++s0)
        if (s0 < nk8) issue(s0, s0);

    const bool ch_ok = n0 + 2 * r16 < C;
    f32x4 acc[3][4];          // [product 3 ph + j][column tile: 2 e + (0 tanh | 1 sigmoid)]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // m1 enters all four outputs with coefficient 1: its accumulators start from the bias
            const float bv = (j == 1 && ph == 0 && p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][c][r] = bv;
        }
    if (tid < ROWS) {
        const int row = VS ? strip + vs * (m0 + tid) : m0 + tid;              // real row
        const int t2 = row / cond_up;
        const int u = row - t2 * cond_up;
        reinterpret_cast<unsigned *>(lds + SH::TAB)[tid] = VS ? ((unsigned)t2 << 8) | (unsigned)u
                                                              : (unsigned)((((t2 - t2base) * 64) << 8) | u);
    }
    if (tid < 64) {
        lds[SH::LERP + tid] = tid < cond_up ? p.lerp_w0[tid] : 0.f;
        lds[SH::LERP + 64 + tid] = tid < cond_up ? p.lerp_w1[tid] : 0.f;
    }

    const int grp = 16 * rw + r16;
    const float *xptr[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int cell = (q & 3) * PHASE + grp + ((q >> 2) << log2d);
        xptr[q] = lds + 8 * cell + 4 * ((kq >> 1) ^ ((cell >> 3) & 1)) + 2 * (kq & 1);
    }
    const float *bptr = lds + A_FLOATS + (ph * 6) * 256 + lane * 4;      // this wave's three products

    float2 x[6];
    float2 u[2];
    float4 bw[3][2];          // weights of this wave's product j in bw[j]: the next slice's product 0 is requested while product 2 waits
    float2 ca, cb;

    auto load_x = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float2 *>(xptr[q] + S * STAGE);
    };
    auto load_b = [&](auto sc, auto jc) {
        constexpr int S = decltype(sc)::value, J = decltype(jc)::value;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            bw[J][e] = *reinterpret_cast<const float4 *>(bptr + S * STAGE + (J * 2 + e) * 256);
    };
    auto mfma8 = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f32x4 *ac = acc[J];
        const float2 uu = u[J & 1];
        const float4 b0 = bw[J][0], b1 = bw[J][1];
        ac[0] = WW_MFMA(uu.x, b0.x, ac[0]);
        ac[1] = WW_MFMA(uu.x, b0.z, ac[1]);
        ac[2] = WW_MFMA(uu.x, b1.x, ac[2]);
        ac[3] = WW_MFMA(uu.x, b1.z, ac[3]);
        ac[0] = WW_MFMA(uu.y, b0.y, ac[0]);
        ac[1] = WW_MFMA(uu.y, b0.w, ac[1]);
        ac[2] = WW_MFMA(uu.y, b1.y, ac[2]);
        ac[3] = WW_MFMA(uu.y, b1.w, ac[3]);
    };
    // One slice = three phases of 8 MFMAs (one product each); the barrier that publishes slice st+1 sits in front of the
    // last one.  In: u[0], bw[0] of this wave's first product.  Out: those of the next slice.
    auto fill = [&](auto sc, int st) {
        constexpr int S = decltype(sc)::value;
        ww_int<(S + 1) % NSTAGE> ns;
        load_b(sc, ww_int<1>());
        WW_FENCE();
        wp_comb<1>(ph, x, u, ca, cb);
        mfma8(ww_int<0>());
        WW_FENCE();
        load_b(sc, ww_int<2>());
        WW_FENCE();
        wp_comb<2>(ph, x, u, ca, cb);
        mfma8(ww_int<1>());
        WW_FENCE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + NSTAGE < nk8) issue(st + NSTAGE, S);
        load_x(ns);
        load_b(ns, ww_int<0>());
        WW_FENCE();
        mfma8(ww_int<2>());
        WW_FENCE();
        wp_comb<0>(ph, x, u, ca, cb);
        WW_FENCE();
    };

    // ---- the first slice and the conditioning tile have landed; the later slices may still be in flight
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __syncthreads();
    load_x(ww_int<0>());
    load_b(ww_int<0>(), ww_int<0>());
    wp_comb<0>(ph, x, u, ca, cb);
    {
        int st = 0;
        for (; st + 2 <= nk8; st += 2) {
            fill(ww_int<0>(), st);
            fill(ww_int<1>(), st + 1);
        }
        if (st < nk8) fill(ww_int<0>(), st);
    }

    // ---- the product halves of a row half meet (through the stage memory: 4 waves x 8 KB): every wave hands over the sum
    // and the difference of its two symmetric products and keeps what it needs of its own
    __syncthreads();                                 // all LDS operand reads are done
    float sv[4][4], dv[4][4];                        // [column tile][row v]: s12 / d12 (ph 0) or s34 / d34 (ph 1)
    {
        float2 *mine = reinterpret_cast<float2 *>(lds) + wave * 1024 + lane;
        const int ja = ph == 0 ? 1 : 0, jb = ph == 0 ? 2 : 1;          // m1, m2 | m3, m4 in this wave's accumulators
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                sv[c][v] = acc[ja][c][v] + acc[jb][c][v];
                dv[c][v] = acc[ja][c][v] - acc[jb][c][v];
                mine[(c * 4 + v) * 64] = make_float2(sv[c][v], dv[c][v]);
            }
    }
    __syncthreads();
    const float2 *theirs = reinterpret_cast<const float2 *>(lds) + (wave ^ 1) * 1024 + lane;
    const float *cl = lds + SH::COND;
    float *obase = p.out + (long long)b * p.out_bstride + n0 + 2 * r16;
    const float *clane = cl + 2 * r16;
    const float *gcond = p.cond + (long long)b * p.cond_bstride + n0 + 2 * r16;      // (VS)
    // (as in the 256-row kernel: table entries first, the conditioning reads of a register's outputs together, results formed
    // outside the store branches)
    int etab[4][2];
#pragma unroll
    for (int vi = 0; vi < 4; ++vi) {
        const int gi = 16 * rw + 4 * kq + vi;                                    // group held by this register
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));         // its first row, relative to m0
#pragma unroll
        for (int o = 0; o < 2; ++o) etab[vi][o] = reinterpret_cast<const int *>(lds + SH::TAB)[lr0 + ((2 * ph + o) << log2d)];
    }
#pragma unroll
    for (int vi = 0; vi < 4; ++vi) {
        const int gi = 16 * rw + 4 * kq + vi;
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));
        float2 w[2], ct0[2], ct1[2], cs0[2], cs1[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int e = etab[vi][o];
            w[o] = make_float2(lds[SH::LERP + (e & 255)], lds[SH::LERP + 64 + (e & 255)]);
            if (VS) {
                const int n2 = rows_item / cond_up;
                const int t2 = (int)((unsigned)e >> 8);
                const bool live = ch_ok && m0 + lr0 + ((2 * ph + o) << log2d) < rows;
                const float *g0 = live ? gcond + (long long)min(t2, n2 - 1) * (2 * C) : p.zeros;
                const float *g1 = live ? gcond + (long long)min(t2 + 1, n2 - 1) * (2 * C) : p.zeros;
                const int so = live ? C : 0;
                ct0[o] = *reinterpret_cast<const float2 *>(g0);
                ct1[o] = *reinterpret_cast<const float2 *>(g1);
                cs0[o] = *reinterpret_cast<const float2 *>(g0 + so);
                cs1[o] = *reinterpret_cast<const float2 *>(g1 + so);
            } else {
                const float *c0 = clane + (e >> 8);
                ct0[o] = *reinterpret_cast<const float2 *>(c0);
                ct1[o] = *reinterpret_cast<const float2 *>(c0 + 64);
                cs0[o] = *reinterpret_cast<const float2 *>(c0 + 32);
                cs1[o] = *reinterpret_cast<const float2 *>(c0 + 96);
            }
        }
        float y[4][2];                                                           // [column tile][this wave's two outputs]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float2 t = theirs[(c * 4 + vi) * 64];                          // (s, d) of the other product half
            if (ph == 0) {
                y[c][0] = (acc[0][c][vi] + sv[c][vi]) + t.x;                     // (m0 + s12) + s34
                y[c][1] = fmaf(2.f, t.y, dv[c][vi]);                             // d12 + 2 d34
            } else {
                y[c][0] = fmaf(4.f, sv[c][vi], t.x);                             // s12 + 4 s34
                y[c][1] = fmaf(8.f, dv[c][vi], t.y) + acc[2][c][vi];             // d12 + 8 d34 + m5
            }
        }
        float2 res[2];
        const int kind = GA < 0 ? p.gate_act : GA;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            res[o].x = wn_gate_act(kind, y[0][o] + fmaf(ct0[o].x, w[o].x, ct1[o].x * w[o].y), y[1][o] + fmaf(cs0[o].x, w[o].x, cs1[o].x * w[o].y));
            res[o].y = wn_gate_act(kind, y[2][o] + fmaf(ct0[o].y, w[o].x, ct1[o].y * w[o].y), y[3][o] + fmaf(cs0[o].y, w[o].x, cs1[o].y * w[o].y));
            asm volatile("" : "+v"(res[o].x), "+v"(res[o].y));
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int row = m0 + lr0 + ((2 * ph + o) << log2d);
            const long long orow = VS ? (long long)strip + (long long)vs * row : row;          // real row
            if (ch_ok && row < rows) *reinterpret_cast<float2 *>(obase + orow * p.ldo) = res[o];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// <128 rows, products split, HALF a column tile> (round 5): the product-split block for launches that do not fill the chip
// evenly.  A 3 s utterance is 380 product-split blocks on 256 CUs: 124 CUs work two blocks while 132 work one, and the launch
// lasts as long as two.  Here a block owns the 16 gate channels of ONE channel parity e of its column tile (the packed weight
// image is [product][parity e][lane][4]: the block stages the e-th kilobyte of every product and nothing else): twice the blocks
// of half the matrix work -- 760 for 3 s, every CU gets three, the launch lasts as long as one and a half.  Same groups, same
// products, same sums in the same order as the other two shapes: the SAME bits.  LDS 29.5 KB, 4+ blocks per CU.
// Measured (profiles/r05_gate_shapes.txt): 54.0 against 48.8 us at 3 s -- a block of half the matrix work lasts almost as long
// (its 40 barrier-separated slices set its time, not its MFMAs) -- and 43.8 against 48.5 us at 2 s, where the product-split
// blocks leave CUs empty: the launcher takes this shape there only (mbx_api.hip).
struct WhShape {
    static constexpr int ROWS = 128;
    static constexpr int PHASE = ROWS / 4 + WW_HALO;            // 48
    static constexpr int A_FLOATS = 4 * PHASE * WW_BK;          // 1536
    static constexpr int B_FLOATS = 6 * 256;                    // one parity of the six products
    static constexpr int STAGE = A_FLOATS + B_FLOATS;           // 3072 floats = 12 KB
    static constexpr int NSTAGE = 2;
    static constexpr int COND_ROWS = 16;
    static constexpr int COND = NSTAGE * STAGE;
    static constexpr int TAB = COND + COND_ROWS * 64;
    static constexpr int LERP = TAB + ROWS;
    static constexpr int LDS_FLOATS = LERP + 128;               // 6144 + 1024 + 128 + 128 = 7424 floats = 29 696 bytes
};

template <int GA>
__global__ __launch_bounds__(256, 4) void wn_gate_winograd4h_kernel(ConvArgs p, int log2d) {
    using SH = WhShape;
    constexpr int ROWS = SH::ROWS, NSTAGE = SH::NSTAGE, STAGE = SH::STAGE, A_FLOATS = SH::A_FLOATS, PHASE = SH::PHASE;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[SH::LDS_FLOATS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // p.n_tiles counts HALF column tiles here: nt2 = 2 * (column tile) + parity
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int nt2 = l % p.n_tiles;
    const int nt = nt2 >> 1, eh = nt2 & 1;
    if (g_ >= p.m_tiles_total) return;
    const int b = g_ / p.m_tiles_per_item;
    const int mt = g_ - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave >> 1, ph = wave & 1;                // row half and product half of this wave
    const int r16 = lane & 15, kq = lane >> 4;
    const int xrow0 = max(m0 - WW_HALO, 0);
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)xrow0 * p.ldx;
    const int nk8 = (p.cin + WW_BK - 1) / WW_BK;

    // ---- per-lane DMA sources of the activation rows (as in wn_gate_winograd4p_kernel)
    unsigned a_voff[2];
    unsigned a_bits = 0;
    int a_inst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_inst[i] = i == 0 ? wave : 4 + (wave & 1);         // chunks 4, 5: written by two waves each (same data)
        const int pos = a_inst[i] * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / PHASE, sidx = cell - phase * PHASE;
        const int m = 4 * (sidx >> log2d) + phase;
        const int row = (m << log2d) + (sidx & (d - 1)) + WW_HALO - d;
        const int src = m0 - WW_HALO + row;
        const int hi = (pos & 1) ^ ((cell >> 3) & 1);
        if (row < ROWS + 2 * WW_HALO && src >= 0 && src < rows) a_bits |= 1u << i;
        a_bits |= (unsigned)hi << (4 + i);
        a_voff[i] = 4u * (unsigned)((min(max(src, 0), rows - 1) - xrow0) * p.ldx + 4 * hi);
    }
    const bool fast_rows = p.fast_dma && m0 >= WW_HALO && m0 + ROWS + WW_HALO <= rows;
    const int whole_fills = p.cin / WW_BK;
    const float *wtile = p.w + (long long)nt * nk8 * WW_B_FLOATS + eh * 256;      // product j of this parity: + 512 j
    const unsigned b_voff = 16u * (unsigned)lane;
    // LDS-DMA of slice st: 2 requests for the rows + 2 for the weights per wave (products wave and 4 + (wave & 1): the last two
    // are requested twice, so that every wave has the same number of requests in flight)
    auto issue = [&](int st, int stage) {
        const int ci0 = st * WW_BK;
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * STAGE);
        if (fast_rows && st < whole_fills) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < 2; ++i) ww_lds_dma16_s(abase, a_voff[i], sdst + 1024u * (unsigned)a_inst[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ci = ci0 + 4 * (int)((a_bits >> (4 + i)) & 1u);
                const bool ok = ((a_bits >> i) & 1u) & (ci < p.cin);
                const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff[i]);
                ww_lds_dma16(ok ? src : p.zeros, sdst + 1024u * (unsigned)a_inst[i]);
            }
        }
        const float *bsrc = wtile + (long long)st * WW_B_FLOATS;
        const int j0 = wave, j1 = 4 + (wave & 1);
        ww_lds_dma16_s(bsrc + j0 * 512, b_voff, sdst + 4u * (unsigned)A_FLOATS + 1024u * (unsigned)j0);
        ww_lds_dma16_s(bsrc + j1 * 512, b_voff, sdst + 4u * (unsigned)A_FLOATS + 1024u * (unsigned)j1);
    };
    // ---- conditioning rows of this block (16 x (32 tanh | 32 sigmoid) columns of the whole column tile): one request per wave
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    {
        const int n2 = rows / cond_up;
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
        const int pos = wave * 64 + lane;
        const int crow = pos >> 4, cq = pos & 15;
        const int chn = n0 + 4 * (cq & 7);
        const int t = min(t2base + crow, n2 - 1);
        ww_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                     lds_base + 4u * (unsigned)SH::COND + 1024u * (unsigned)wave);
    }
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE; ++s0)
        if (s0 < nk8) issue(s0, s0);

    const int chl = n0 + 2 * r16 + eh;                    // gate channel of this lane
    const bool ch_ok = chl < C;
    f32x4 acc[3][2];          // [product 3 ph + j][0 tanh | 1 sigmoid]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float bv = (j == 1 && ph == 0 && p.bias && ch_ok) ? p.bias[c * C + chl] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][c][r] = bv;
        }
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

    const int grp = 16 * rw + r16;
    const float *xptr[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int cell = (q & 3) * PHASE + grp + ((q >> 2) << log2d);
        xptr[q] = lds + 8 * cell + 4 * ((kq >> 1) ^ ((cell >> 3) & 1)) + 2 * (kq & 1);
    }
    const float *bptr = lds + A_FLOATS + (ph * 3) * 256 + lane * 4;      // this wave's three products (one parity each)

    float2 x[6];
    float2 u[2];
    float4 bw[3];
    float2 ca, cb;

    auto load_x = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float2 *>(xptr[q] + S * STAGE);
    };
    auto load_b = [&](auto sc, auto jc) {
        constexpr int S = decltype(sc)::value, J = decltype(jc)::value;
        bw[J] = *reinterpret_cast<const float4 *>(bptr + S * STAGE + J * 256);
    };
    auto mfma4 = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f32x4 *ac = acc[J];
        const float2 uu = u[J & 1];
        const float4 b0 = bw[J];
        ac[0] = WW_MFMA(uu.x, b0.x, ac[0]);
        ac[1] = WW_MFMA(uu.x, b0.z, ac[1]);
        ac[0] = WW_MFMA(uu.y, b0.y, ac[0]);
        ac[1] = WW_MFMA(uu.y, b0.w, ac[1]);
    };
    auto fill = [&](auto sc, int st) {
        constexpr int S = decltype(sc)::value;
        ww_int<(S + 1) % NSTAGE> ns;
        load_b(sc, ww_int<1>());
        WW_FENCE();
        wp_comb<1>(ph, x, u, ca, cb);
        mfma4(ww_int<0>());
        WW_FENCE();
        load_b(sc, ww_int<2>());
        WW_FENCE();
        wp_comb<2>(ph, x, u, ca, cb);
        mfma4(ww_int<1>());
        WW_FENCE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + NSTAGE < nk8) issue(st + NSTAGE, S);
        load_x(ns);
        load_b(ns, ww_int<0>());
        WW_FENCE();
        mfma4(ww_int<2>());
        WW_FENCE();
        wp_comb<0>(ph, x, u, ca, cb);
        WW_FENCE();
    };

    // ---- the first slice and the conditioning tile have landed; the second slice may still be in flight
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    load_x(ww_int<0>());
    load_b(ww_int<0>(), ww_int<0>());
    wp_comb<0>(ph, x, u, ca, cb);
    {
        int st = 0;
        for (; st + 2 <= nk8; st += 2) {
            fill(ww_int<0>(), st);
            fill(ww_int<1>(), st + 1);
        }
        if (st < nk8) fill(ww_int<0>(), st);
    }

    // ---- the product halves of a row half meet through the stage memory (4 waves x 4 KB)
    __syncthreads();
    float sv[2][4], dv[2][4];
    {
        float2 *mine = reinterpret_cast<float2 *>(lds) + wave * 512 + lane;
        const int ja = ph == 0 ? 1 : 0, jb = ph == 0 ? 2 : 1;          // m1, m2 | m3, m4 in this wave's accumulators
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                sv[c][v] = acc[ja][c][v] + acc[jb][c][v];
                dv[c][v] = acc[ja][c][v] - acc[jb][c][v];
                mine[(c * 4 + v) * 64] = make_float2(sv[c][v], dv[c][v]);
            }
    }
    __syncthreads();
    const float2 *theirs = reinterpret_cast<const float2 *>(lds) + (wave ^ 1) * 512 + lane;
    const float *cl = lds + SH::COND;
    float *obase = p.out + (long long)b * p.out_bstride + chl;
    const float *clane = cl + 2 * r16 + eh;
    int etab[4][2];
#pragma unroll
    for (int vi = 0; vi < 4; ++vi) {
        const int gi = 16 * rw + 4 * kq + vi;                                    // group held by this register
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));         // its first row, relative to m0
#pragma unroll
        for (int o = 0; o < 2; ++o) etab[vi][o] = reinterpret_cast<const int *>(lds + SH::TAB)[lr0 + ((2 * ph + o) << log2d)];
    }
#pragma unroll
    for (int vi = 0; vi < 4; ++vi) {
        const int gi = 16 * rw + 4 * kq + vi;
        const int lr0 = ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));
        float2 w[2];
        float ct0[2], ct1[2], cs0[2], cs1[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int e = etab[vi][o];
            w[o] = make_float2(lds[SH::LERP + (e & 255)], lds[SH::LERP + 64 + (e & 255)]);
            const float *c0 = clane + (e >> 8);
            ct0[o] = c0[0];
            ct1[o] = c0[64];
            cs0[o] = c0[32];
            cs1[o] = c0[96];
        }
        float y[2][2];                                                           // [tanh | sigmoid][this wave's two outputs]
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float2 t = theirs[(c * 4 + vi) * 64];                          // (s, d) of the other product half
            if (ph == 0) {
                y[c][0] = (acc[0][c][vi] + sv[c][vi]) + t.x;                     // (m0 + s12) + s34
                y[c][1] = fmaf(2.f, t.y, dv[c][vi]);                             // d12 + 2 d34
            } else {
                y[c][0] = fmaf(4.f, sv[c][vi], t.x);                             // s12 + 4 s34
                y[c][1] = fmaf(8.f, dv[c][vi], t.y) + acc[2][c][vi];             // d12 + 8 d34 + m5
            }
        }
        float res[2];
        const int kind = GA < 0 ? p.gate_act : GA;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            res[o] = wn_gate_act(kind, y[0][o] + fmaf(ct0[o], w[o].x, ct1[o] * w[o].y), y[1][o] + fmaf(cs0[o], w[o].x, cs1[o] * w[o].y));
            asm volatile("" : "+v"(res[o]));
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int row = m0 + lr0 + ((2 * ph + o) << log2d);
            if (ch_ok && row < rows) obase[(long long)row * p.ldo] = res[o];
        }
    }
}

// a.w must point at the host-packed F(4,3) weights (ceil(C/32), ceil(C/8), 3072) of engine.pack_winograd4w_weights;
// split = the 128-row shape whose waves split the six products (same bits as the 256-row shape).  Returns false if the
// layer does not fit.
// shape: 0 = 256-row blocks, 1 = 128-row product-split blocks, 2 = product-split blocks of half a column tile
bool launch_wn_gate_winograd4w(const ConvArgs &a, int shape, hipStream_t stream) {
    const bool split = shape != 0;
    // dilations above the halo: d = 16 s runs as s interleaved virtual items per item at dilation 16 (kernels' VS variants)
    const int vs = a.dil > WW_HALO ? a.dil / WW_HALO : 1;
    const int dil_v = vs > 1 ? WW_HALO : a.dil;
    int log2d = 0;
    while ((1 << log2d) < dil_v) ++log2d;
    const int rows_blk = split ? 128 : 256;
    const int nk8 = (a.cin + WW_BK - 1) / WW_BK;
    const int cond_rows = vs > 1 ? 1 : a.cond_up >= 1 ? (rows_blk + a.cond_up - 2) / a.cond_up + 2 : 1 << 30;   // conditioning rows a block's tile holds
    const bool ok = a.ks == 3 && (1 << log2d) == dil_v && dil_v * vs == a.dil && nk8 >= 4 && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros &&
                    a.cond && (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up >= 1 &&
                    a.cond_up <= 64 && cond_rows <= (split ? 16 : 56) && a.lerp_w0 && a.lerp_w1 && a.max_rows < (1 << 24) &&
                    (vs == 1 || (shape != 2 && a.max_rows >= a.cond_up &&
                                 (long long)(rows_blk + 2 * WW_HALO) * vs * a.ldx * 4 < (1LL << 32)));
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = 1;                 // byte offsets are relative to the block's window (< 2^32 for any item length)
    r.vstride = vs;
    r.n_tiles = (a.channels + 31) / 32;
    const int vrows = (a.max_rows + vs - 1) / vs;         // rows of the longest virtual item
    r.m_tiles_per_item = (vrows + rows_blk - 1) / rows_blk;
    r.m_tiles_total = r.m_tiles_per_item * a.batch * vs;
    // sub-sequences of at most 128 rows: a 256-row block takes two of them (VS == 2)
    const bool two = vs > 1 && !split && vrows <= 128 && vs % 2 == 0 && log2d == 4;
    if (two) r.m_tiles_total = a.batch * (vs / 2);
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    const dim3 blk(256);
    const bool gtu = a.gate_act == 0;
    if (shape == 2) {
        r.n_tiles *= 2;                                   // half column tiles
        const dim3 grid2((unsigned)(8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles));
        if (gtu) hipLaunchKernelGGL(wn_gate_winograd4h_kernel<0>, grid2, blk, 0, stream, r, log2d);
        else hipLaunchKernelGGL(wn_gate_winograd4h_kernel<-1>, grid2, blk, 0, stream, r, log2d);
        return true;
    }
    const dim3 grid((unsigned)blocks);
    if (vs > 1) {
        if (split && gtu) hipLaunchKernelGGL((wn_gate_winograd4p_kernel<0, 1>), grid, blk, 0, stream, r, log2d);
        else if (split) hipLaunchKernelGGL((wn_gate_winograd4p_kernel<-1, 1>), grid, blk, 0, stream, r, log2d);
        else if (two && gtu) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<28, 0, 2>), grid, blk, 0, stream, r, log2d);
        else if (two) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<28, -1, 2>), grid, blk, 0, stream, r, log2d);
        else if (gtu) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<28, 0, 1>), grid, blk, 0, stream, r, log2d);
        else hipLaunchKernelGGL((wn_gate_winograd4w_kernel<28, -1, 1>), grid, blk, 0, stream, r, log2d);
        return true;
    }
    if (split && gtu) hipLaunchKernelGGL(wn_gate_winograd4p_kernel<0>, grid, blk, 0, stream, r, log2d);
    else if (split) hipLaunchKernelGGL(wn_gate_winograd4p_kernel<-1>, grid, blk, 0, stream, r, log2d);
    else if (cond_rows <= 28 && gtu) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<28, 0>), grid, blk, 0, stream, r, log2d);
    else if (cond_rows <= 28) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<28, -1>), grid, blk, 0, stream, r, log2d);
    else if (gtu) hipLaunchKernelGGL((wn_gate_winograd4w_kernel<56, 0>), grid, blk, 0, stream, r, log2d);
    else hipLaunchKernelGGL((wn_gate_winograd4w_kernel<56, -1>), grid, blk, 0, stream, r, log2d);
    return true;
}

}  // namespace mbx
