// WaveNet dilated convolution (k = 3, dilation d) + conditioning + tanh*sigmoid in Winograd F(2,3) form on
// v_mfma_f32_16x16x4_f32, tiled at WAVE granularity: the small-launch / streaming form of the gate layer.
//
// Layer: reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:305-321 (in_layered = conv1D_l(h); z = in_layered + cond;
// a = tanh(z[:C]) * sigmoid(z[C:])).  Two outputs d steps apart share their products: with x0..x3 = h[t-d], h[t],
// h[t+d], h[t+2d] and the taps W0, W1, W2 (each C x 2C)
//     m1 = (x0 - x2) W0      m2 = (x1 + x2) (W0 + W1 + W2)/2      m3 = (x2 - x1) (W0 - W1 + W2)/2      m4 = (x1 - x3) W2
//     y[t] = m1 + m2 + m3    y[t+d] = m2 - m3 - m4
// = 4 channel contractions per output pair instead of 6 (weight combinations formed on the host in float64,
// engine.pack_winograd2w_weights).  F(2,3) has no numerical reach beyond the receptive field of the layer, which is why
// the streaming windows (streaming.py) run it: a window then reproduces the offline synthesis bit for bit.
//
// Wave tile (as in wn_winograd4w.hip: few rows, many columns, because every vector instruction beside the fp32 MFMA
// costs matrix-pipe time): 16 pairs = 32 consecutive output rows x 64 weight columns ([16 tanh | 16 sigmoid] of the
// even and of the odd gate channels of a 32-channel column tile) x 4 products = 16 accumulator tiles of 4 registers;
// per 8-channel slice 32 MFMAs beside 8 vector instructions.  The bias is the initial value of product m2 (it enters
// both outputs with coefficient 1).
//
// What is new against the 256 / 128-row block shapes: a block is four INDEPENDENT wave tiles of one column tile -- wave w
// of block g owns wave tile 4 g + w of the flat list (item, 32-row tile) and stages its own activation rows; only the
// weight slices are shared by the block.  A launch therefore rounds every item up to 32 rows, not to 128 or 256: a
// steady streaming tick (64 streams x 160..191 output rows per layer) runs 6 wave tiles per stream and column tile
// instead of two 128-row blocks (256 rows), and a layer-region launch computes only the rows that are consumed
// (ConvArgs::out_row0 / out_rows: the rows in front of and behind them are the layer's reach, needed as inputs only).
// The arithmetic of an output does not depend on the tiling (slices in order, one MFMA chain per product), so streaming
// windows, per-layer regions and whole utterances agree bit for bit.
//
// Pair P of a wave tile: q = P / d, r = P % d, t = m0 + 2 d q + r (d a power of two <= 16).
// Per 8-channel slice a wave stages, through LDS-DMA, its rows [m0 - d, m0 + 32 + d) x 8 channels in read order: row
// m0 - d + d m + b (b < d) lives in 32-byte cell p = (m & 1) * 32 + (m >> 1) * d + b, its 16-byte chunk c at
// 2 p + (c ^ ((p >> 3) & 1)); lane (r = lane & 15, kq = lane >> 4) reads the 8 bytes of channels 2 kq, 2 kq + 1 of cell
// (i & 1) * 32 + P + (i >> 1) * d for its four rows i -- consecutive lanes, consecutive cells, conflict free with the
// chunk swizzle.  The block stages the weights of the slice: 4 products x 8 channels x 64 columns in MFMA operand order
// [product j][channel parity e][lane][tanh step 0, tanh step 1, sigmoid step 0, sigmoid step 1] (one ds_read_b128 = the
// weight operands of four MFMAs).  LDS: two stages x (4 x 2 KB + 8 KB) + the four conditioning tiles (8 rows x 64 floats
// each): 40 KB -> 4 blocks per CU = 4 waves per SIMD, which is what hides the LDS-DMA latency and the
// prologue / epilogue of these short blocks.
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W2_HALO = 16;                            // largest dilation
constexpr int W2_BK = 8;
constexpr int W2_TROWS = 32;                           // output rows of a wave tile
constexpr int W2_PHASE = W2_TROWS / 2 + W2_HALO;       // cells per phase (m & 1)
constexpr int W2_A_FLOATS = 2 * W2_PHASE * W2_BK;      // 512: one wave's rows of one slice
constexpr int W2_B_FLOATS = 4 * W2_BK * 64;            // 2048: packed weights of one slice
constexpr int W2_STAGE = 4 * W2_A_FLOATS + W2_B_FLOATS;   // 4096 floats = 16 KB
constexpr int W2_COND_ROWS = 8;                        // conditioning rows of a wave tile (64 floats each)

constexpr int W2_NSTAGE = 2;                           // (three stages at 3 blocks per CU: 0.834 against 0.811 ms per 64-stream tick)
constexpr int W2_COND = W2_NSTAGE * W2_STAGE;
constexpr int W2_LDS_FLOATS = W2_COND + 4 * W2_COND_ROWS * 64;            // 40 KB: 4 blocks per CU

__device__ __forceinline__ void w2_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset
__device__ __forceinline__ void w2_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

#define W2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define W2_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W2_SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define W2_SG_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
template <int N>
using w2_int = std::integral_constant<int, N>;

// GA: gate activation as a compile-time constant (0 = gtu) or -1 = ConvArgs::gate_act (see wn_winograd4w.hip)
template <int GA>
__global__ __launch_bounds__(256, 4) void wn_gate_winograd2w_kernel(ConvArgs p, int log2d) {
    constexpr int NSTAGE = W2_NSTAGE;
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[W2_LDS_FLOATS];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode (see decode_tile in conv_mfma.hip): a row group = four wave tiles; its column tiles run back to back
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpi = p.m_tiles_per_item;
    if (4 * g_ >= p.m_tiles_total) return;
    // this wave's tile: item b, output rows [m0, m0 + 32) of the item; the block leaves when none of its four tiles has rows
    int b = 0, m0 = 0, rows = 1, out_hi = 0;
    bool active = false, any = false;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int wt = 4 * g_ + w;
        if (wt >= p.m_tiles_total) break;
        const int bb = wt / tpi;
        const int rr = item_rows(p.n_frames, bb, p.rows_per_frame, p.max_rows);
        const int hi = p.out_rows > 0 ? min(rr, p.out_row0 + p.out_rows) : rr;
        const int mm = p.out_row0 + W2_TROWS * (wt - bb * tpi);
        any = any || mm < hi;
        if (w == wave) {
            b = bb;
            m0 = mm;
            rows = max(rr, 1);
            out_hi = hi;
            active = mm < hi;
        }
    }
    if (!any) return;
    if (!active) m0 = 0;                                 // addresses stay valid; nothing of this wave's tile is used
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int r16 = lane & 15, kq = lane >> 4;
    // rows are addressed relative to the tile's first staged row: 32-bit byte offsets stay small for any item length
    const int xrow0 = max(m0 - d, 0);
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)xrow0 * p.ldx;
    const int nk8 = (p.cin + W2_BK - 1) / W2_BK;         // 8-channel slices of the weight image = stage fills

    // ---- per-lane DMA sources of this wave's rows (fixed except the channel offset): byte offset of (row, chunk) from
    // the item's first element; bit i: the row exists, bit 4 + i: the chunk is the upper half of the slice
    unsigned a_voff[2];
    unsigned a_bits = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = i * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / W2_PHASE, sidx = cell - phase * W2_PHASE;
        const int m = 2 * (sidx >> log2d) + phase;
        const int src = m0 - d + (m << log2d) + (sidx & (d - 1));
        const int hi = (pos & 1) ^ ((cell >> 3) & 1);
        if (active && sidx < W2_TROWS / 2 + d && src >= 0 && src < rows) a_bits |= 1u << i;
        a_bits |= (unsigned)hi << (4 + i);
        a_voff[i] = 4u * (unsigned)((min(max(src, 0), rows - 1) - xrow0) * p.ldx + 4 * hi);
    }
    // interior tiles (every staged row exists, whole slices): uniform base + per-lane byte offset, no selects
    const bool fast_rows = active && p.fast_dma && m0 >= d && m0 + W2_TROWS + d <= rows;
    const int whole_fills = p.cin / W2_BK;
    const float *wtile = p.w + (long long)nt * nk8 * W2_B_FLOATS;
    const unsigned b_voff = 16u * (unsigned)lane;
    // LDS-DMA of slice st into a stage: 2 requests for this wave's rows + 2 of the 8 weight requests
    auto issue = [&](int st, int stage) {
        const int ci0 = st * W2_BK;
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * W2_STAGE);
        const unsigned adst = sdst + 4u * (unsigned)(wave * W2_A_FLOATS);
        const unsigned bdst = sdst + 4u * (unsigned)(4 * W2_A_FLOATS);
        if (fast_rows && st < whole_fills) {
            const float *abase = xb + ci0;
            w2_lds_dma16_s(abase, a_voff[0], adst);
            w2_lds_dma16_s(abase, a_voff[1], adst + 1024u);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ci = ci0 + 4 * (int)((a_bits >> (4 + i)) & 1u);
                const bool ok = ((a_bits >> i) & 1u) & (ci < p.cin);
                const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xb + ci0) + a_voff[i]);
                w2_lds_dma16(ok ? src : p.zeros, adst + 1024u * (unsigned)i);
            }
        }
        const float *bsrc = wtile + (long long)st * W2_B_FLOATS;         // st < nk8 always: the image has nk8 slices
        w2_lds_dma16_s(bsrc + wave * 256, b_voff, bdst + 1024u * (unsigned)wave);
        w2_lds_dma16_s(bsrc + (wave + 4) * 256, b_voff, bdst + 1024u * (unsigned)(wave + 4));
    };
    // ---- conditioning rows of this wave tile (8 rows x (32 tanh | 32 sigmoid) columns), requested first so that every
    // later wait covers them.  cond_phase: conditioning-rate position of item row 0 inside its conditioning row (items
    // that start between two conditioning rows: the per-layer regions of a streaming tick)
    const int cond_up = p.cond_up;
    const int cphase = p.cond_phase;
    const int t2base = (m0 + cphase) / cond_up;
    {
        const int n2 = max((rows + cphase) / cond_up, 1);
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = i * 64 + lane;
            const int crow = pos >> 4, cq = pos & 15;
            const int chn = n0 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            w2_lds_dma16((active && chn < C) ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                         lds_base + 4u * (unsigned)(W2_COND + wave * (W2_COND_ROWS * 64)) + 1024u * (unsigned)i);
        }
    }
    // ---- prologue: every stage requested
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (s < nk8) issue(s, s);

    // lane n of column tile (e, tanh | sigmoid) holds gate channel n0 + 2 n + e
    const bool ch_ok = n0 + 2 * r16 < C;                 // C is even: both channels of the lane exist or neither
    f32x4 acc[4][4];          // [product][column tile: 2 e + (0 tanh | 1 sigmoid)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float bv = (j == 1 && p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][c][r] = bv;
        }

    // A operand: pair P = r16 of this wave tile; channels 2 kq, 2 kq + 1 of rows h[t-d], h[t], h[t+d], h[t+2d]
    const float *xptr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cell = (q & 1) * W2_PHASE + r16 + ((q >> 1) << log2d);
        xptr[q] = lds + wave * W2_A_FLOATS + 8 * cell + 4 * ((kq >> 1) ^ ((cell >> 3) & 1)) + 2 * (kq & 1);
    }
    const float *bptr = lds + 4 * W2_A_FLOATS + lane * 4;

    float2 x[4];              // raw activation rows of the slice whose combinations are being formed
    float2 u[2];              // input combination of product j in u[j & 1]
    float4 bw[2][2];          // weights of product j in bw[j & 1][channel parity e]

    auto load_x = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const float2 *>(xptr[q] + S * W2_STAGE);
    };
    auto load_b = [&](auto sc, auto jc) {
        constexpr int S = decltype(sc)::value, J = decltype(jc)::value;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            bw[J & 1][e] = *reinterpret_cast<const float4 *>(bptr + S * W2_STAGE + (J * 2 + e) * 256);
    };
    auto mfma8 = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f32x4 *ac = acc[J];
        const float2 uu = u[J & 1];
        const float4 b0 = bw[J & 1][0], b1 = bw[J & 1][1];
        ac[0] = W2_MFMA(uu.x, b0.x, ac[0]);
        ac[1] = W2_MFMA(uu.x, b0.z, ac[1]);
        ac[2] = W2_MFMA(uu.x, b1.x, ac[2]);
        ac[3] = W2_MFMA(uu.x, b1.z, ac[3]);
        ac[0] = W2_MFMA(uu.y, b0.y, ac[0]);
        ac[1] = W2_MFMA(uu.y, b0.w, ac[1]);
        ac[2] = W2_MFMA(uu.y, b1.y, ac[2]);
        ac[3] = W2_MFMA(uu.y, b1.w, ac[3]);
    };
    auto comb = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (J == 0) u[0] = make_float2(x[0].x - x[2].x, x[0].y - x[2].y);
        if (J == 1) u[1] = make_float2(x[1].x + x[2].x, x[1].y + x[2].y);
        if (J == 2) u[0] = make_float2(x[2].x - x[1].x, x[2].y - x[1].y);
        if (J == 3) u[1] = make_float2(x[1].x - x[3].x, x[1].y - x[3].y);
    };
    // One slice = four phases of 8 MFMAs (one product each).  While product j is multiplied, the weights of product j+1
    // are requested from LDS and its input combination is formed between the MFMAs.  The barrier that publishes slice
    // st+1 sits in front of the last product: every wave has requested all LDS operands of slice st by then, so the stage
    // is free for slice st+NSTAGE.  In: u[0], bw[0] of product 0 of this slice.  Out: those of the next one.
    auto phase = [&](auto sc, auto jc) {
        constexpr int J = decltype(jc)::value;
        load_b(sc, w2_int<J + 1>());
        W2_FENCE();
        comb(w2_int<J + 1>());
        mfma8(jc);
        W2_SG_MFMA(1); W2_SG_VALU(1); W2_SG_MFMA(1); W2_SG_VALU(1); W2_SG_MFMA(6);
        W2_FENCE();
    };
    auto fill = [&](auto sc, int st) {
        constexpr int S = decltype(sc)::value;
        w2_int<(S + 1) % NSTAGE> ns;
        if (active) {
            phase(sc, w2_int<0>());
            phase(sc, w2_int<1>());
            phase(sc, w2_int<2>());
        }
        // ---- product 3 behind the barrier; slice st+1 must have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + NSTAGE < nk8) issue(st + NSTAGE, S);
        if (active) {
            load_x(ns);
            load_b(ns, w2_int<0>());
            W2_FENCE();
            mfma8(w2_int<3>());
            W2_FENCE();
            comb(w2_int<0>());
            W2_FENCE();
        }
    };

    // ---- the first slice (and the conditioning tile, requested in front of it) has landed, the second one may still be in
    // flight; the launcher guarantees nk8 >= NSTAGE
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    if (active) {
        load_x(w2_int<0>());
        load_b(w2_int<0>(), w2_int<0>());
        comb(w2_int<0>());
    }
    {
        int st = 0;
        for (; st + 2 <= nk8; st += 2) {
            fill(w2_int<0>(), st);
            fill(w2_int<1>(), st + 1);
        }
        if (st < nk8) fill(w2_int<0>(), st);
    }
    if (!active) return;

    // ---- epilogue: combine the four products, add the conditioning, gate, store the two outputs of every pair.
    // Everything up to the store is unconditional (every conditioning address is valid); only the store is predicated.
    const float *cl = lds + W2_COND + wave * (W2_COND_ROWS * 64) + 2 * r16;
    float *obase = p.out + (long long)b * p.out_bstride + n0 + 2 * r16;
    const float inv_up = 1.0f / (float)cond_up;
    // Round 5 (read off the ISA): the interpolation weights of an output were two global loads behind the branches of the
    // run-time gate kind, each output a region of its own with a full wait (vmcnt(0) lgkmcnt(0)) -- eight serial L2 round
    // trips per block, and a streaming tick's blocks all run in one round, where nothing hides them.  Now: positions and
    // weight requests of all eight outputs first, the conditioning reads of a register's two outputs together, results
    // formed outside the store branches.  Same arithmetic, same bits.
    int t2s[4][2];
    float w0s[4][2], w1s[4][2];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int pi = 4 * kq + v;                                               // pair held by this register
        const int lr0 = ((pi >> log2d) << (log2d + 1)) + (pi & (d - 1));         // its first row, relative to m0
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int crow = m0 + lr0 + (o << log2d) + cphase;
            int t2 = (int)((float)crow * inv_up);                                // crow / cond_up (rows < 2^24)
            int uu = crow - t2 * cond_up;
            if (uu < 0) { --t2; uu += cond_up; }
            if (uu >= cond_up) { ++t2; uu -= cond_up; }
            t2s[v][o] = min(max(t2 - t2base, 0), W2_COND_ROWS - 2) * 64;
            w0s[v][o] = p.lerp_w0[uu];
            w1s[v][o] = p.lerp_w1[uu];
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int pi = 4 * kq + v;
        const int lr0 = ((pi >> log2d) << (log2d + 1)) + (pi & (d - 1));
        float2 ct0[2], ct1[2], cs0[2], cs1[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float *c0 = cl + t2s[v][o];
            ct0[o] = *reinterpret_cast<const float2 *>(c0);
            ct1[o] = *reinterpret_cast<const float2 *>(c0 + 64);
            cs0[o] = *reinterpret_cast<const float2 *>(c0 + 32);
            cs1[o] = *reinterpret_cast<const float2 *>(c0 + 96);
        }
        float y[4][2];                                                           // [column tile][output]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            y[c][0] = (acc[0][c][v] + acc[1][c][v]) + acc[2][c][v];
            y[c][1] = (acc[1][c][v] - acc[2][c][v]) - acc[3][c][v];
        }
        float2 res[2];
        const int kind = GA < 0 ? p.gate_act : GA;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float w0 = w0s[v][o], w1 = w1s[v][o];
            res[o].x = wn_gate_act(kind, y[0][o] + fmaf(ct0[o].x, w0, ct1[o].x * w1), y[1][o] + fmaf(cs0[o].x, w0, cs1[o].x * w1));
            res[o].y = wn_gate_act(kind, y[2][o] + fmaf(ct0[o].y, w0, ct1[o].y * w1), y[3][o] + fmaf(cs0[o].y, w0, cs1[o].y * w1));
            asm volatile("" : "+v"(res[o].x), "+v"(res[o].y));
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int row = m0 + lr0 + (o << log2d);
            if (ch_ok && row < out_hi) *reinterpret_cast<float2 *>(obase + (long long)row * p.ldo) = res[o];
        }
    }
}

// a.w must point at the host-packed F(2,3) weights (ceil(C/32), ceil(C/8), 2048) of engine.pack_winograd2w_weights;
// a.out_row0 / a.out_rows: the rows of every item that are computed (out_rows == 0: all of them; out_row0 a multiple of
// 2 * dilation, so that the output pairs are those of a whole-item run).  Returns false if the layer does not fit.
bool launch_wn_gate_winograd2w(const ConvArgs &a, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const int nk8 = (a.cin + W2_BK - 1) / W2_BK;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= W2_HALO && nk8 >= 3 && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros &&
                    a.cond && (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up >= 1 &&
                    (W2_TROWS + a.cond_up - 2) / a.cond_up + 2 <= W2_COND_ROWS && a.lerp_w0 && a.lerp_w1 &&
                    a.max_rows < (1 << 24) && a.cond_phase >= 0 && a.cond_phase < a.cond_up && a.out_row0 >= 0 &&
                    a.out_rows >= 0 && a.out_row0 % (2 * a.dil) == 0 && a.ldo % 2 == 0 && a.out_bstride % 2 == 0 &&
                    (uintptr_t)a.out % 8 == 0;
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = 1;                 // byte offsets are relative to the tile's window
    r.n_tiles = (a.channels + 31) / 32;
    const int span = a.out_rows > 0 ? std::min(a.out_rows, a.max_rows - a.out_row0) : a.max_rows - a.out_row0;
    if (span <= 0) return true;
    r.m_tiles_per_item = (span + W2_TROWS - 1) / W2_TROWS;               // wave tiles per item
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long groups = (r.m_tiles_total + 3) / 4;
    const long long blocks = 8LL * ((groups + 7) / 8) * r.n_tiles;
    if (a.gate_act == 0) hipLaunchKernelGGL(wn_gate_winograd2w_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    else hipLaunchKernelGGL(wn_gate_winograd2w_kernel<-1>, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    return true;
}

}  // namespace mbx
