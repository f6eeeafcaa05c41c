// WaveNet dilated convolution (k = 3, dilation d) + conditioning + tanh*sigmoid in Winograd F(4,3) form.
//
// Same layer as wn_gate_winograd_kernel (wn_winograd.hip; reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:305-321)
// with four outputs y[t], y[t+d], y[t+2d], y[t+3d] per group instead of two: with x0..x5 = h[t-d] .. h[t+4d]
//     v0 = 4 x0 - 5 x2 + x4          v1 = -4 x1 - 4 x2 + x3 + x4      v2 = 4 x1 - 4 x2 - x3 + x4
//     v3 = -2 x1 - x2 + 2 x3 + x4    v4 = 2 x1 - x2 - 2 x3 + x4       v5 = 4 x1 - 5 x3 + x5
//     m_j = v_j U_j,  U = G W  (G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]])
//     y[t] = m0+m1+m2+m3+m4   y[t+d] = m1-m2+2(m3-m4)   y[t+2d] = m1+m2+4(m3+m4)   y[t+3d] = m1-m2+8(m3-m4)+m5
// = 6 channel contractions per four outputs instead of 12 (direct) or 8 (F(2,3)).  Still float32 on the exact fp32
// MFMA; the larger transform constants cost about 1.4x the rounding error of the direct form (measured rms), far
// inside the parity tolerance.  U is formed on the host in float64 (engine.pack_winograd4_weights).
//
// Block = 2 x 2 waves, 256 consecutive output rows (64 groups; wave row wm owns groups 32 wm .. 32 wm + 31) x 32 gate
// channels (wave column wn owns 16 of them as one 32-wide MFMA tile [16 tanh | 16 sigmoid]); 6 accumulator tiles per
// wave.  Group Q of the block: q = Q / d, r = Q % d, t = m0 + 4 d q + r (d a power of two <= 16).
// Per K slice of 8 channels the block stages, through LDS-DMA (see lds_dma16 in conv_mfma.hip):
//   A: activation rows [m0-16, m0+272) x 8 channels in read order: row = m0 - d + d*m + b (b < d) lives in 32-byte cell
//      p = (m & 3)*80 + (m >> 2)*d + b, its 16-byte chunk c at 2*p + (c ^ ((p>>3)&1)).  The six rows a lane reads are then
//      cells (q & 3)*80 + Q + (q >> 2)*d: consecutive lanes read consecutive cells, bank-conflict free for every
//      dilation (the natural row order costs 4-way conflicts at d = 1, 2).  An LDS-DMA lane writes a fixed cell, so
//      the layout is realised by the source row each lane fetches.
//   B: 6 weight combinations x 8 channels x 64 columns, pre-packed on the host in MFMA operand order
//      [product j][wave column wn][lane][4 k steps]: one ds_read_b128 per lane = the weight operands of four MFMAs
// Three LDS stages (66 KB per block, 2 blocks per CU): the slice needed next has landed two slices of compute ago.
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int W4_ROWS = 256;
constexpr int W4_HALO = 16;
constexpr int W4_AROWS = W4_ROWS + 2 * W4_HALO;       // 288 rows can be needed
constexpr int W4_PHASE = W4_ROWS / 4 + W4_HALO;        // 80 cells per phase (m & 3)
constexpr int W4_CELLS = 4 * W4_PHASE;                 // 320 cells of 8 channels
constexpr int W4_BK = 8;
constexpr int W4_A_FLOATS = W4_CELLS * W4_BK;          // 2560
constexpr int W4_B_FLOATS = 6 * W4_BK * 64;            // 3072
constexpr int W4_STAGE = W4_A_FLOATS + W4_B_FLOATS;    // 5632 floats = 22 KB
constexpr int W4_A_CHUNKS = W4_CELLS * 2 / 64;         // 10 x 1 KB LDS-DMA instructions per slice (A)
constexpr int W4_B_INST = W4_B_FLOATS / 4 / 64 / 4;    // 3 per wave (B)

__device__ __forceinline__ void w4_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset (no per-lane 64-bit address arithmetic)
__device__ __forceinline__ void w4_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

__device__ __forceinline__ float w4_gate_act(float zt, float zs) {
    const float e2 = __expf(2.0f * zt);
    const float e1 = __expf(-zs);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
    return th * __builtin_amdgcn_rcpf(1.0f + e1);
}

__device__ __forceinline__ float4 w4_fma(float s, float4 a, float4 b) {     // s * a + b
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 w4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 w4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// input combinations of product pair g (products 2g, 2g+1) from the six activation rows (4 channels = 4 k steps)
__device__ __forceinline__ void w4_input_comb(int g, const float4 (&x)[6], float4 &u0, float4 &u1) {
    if (g == 0) {
        u0 = w4_fma(4.f, x[0], w4_fma(-5.f, x[2], x[4]));
        const float4 a = w4_fma(-4.f, x[2], x[4]), b = w4_fma(-4.f, x[1], x[3]);
        u1 = w4_add(a, b);
    } else if (g == 1) {
        const float4 a = w4_fma(-4.f, x[2], x[4]), b = w4_fma(-4.f, x[1], x[3]);
        u0 = w4_sub(a, b);
        const float4 c = w4_sub(x[4], x[2]), e = w4_sub(x[3], x[1]);
        u1 = w4_fma(2.f, e, c);
    } else {
        const float4 c = w4_sub(x[4], x[2]), e = w4_sub(x[3], x[1]);
        u0 = w4_fma(-2.f, e, c);
        u1 = w4_fma(4.f, x[1], w4_fma(-5.f, x[3], x[5]));
    }
}

__global__ __launch_bounds__(256, 2) void wn_gate_winograd4_kernel(ConvArgs p, int log2d) {
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[3 * W4_STAGE];          // stage s: A at s*STAGE, B behind it
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
    const int m0 = mt * W4_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nk = (p.cin + W4_BK - 1) / W4_BK;

    // ---- per-lane DMA sources (fixed for the whole kernel except the channel offset)
    int a_off[3], a_ch[3];
    unsigned a_voff[3];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // chunks 0..7 are dealt round-robin, chunks 8 and 9 are each written by two waves (same data), which keeps the
        // number of outstanding LDS-DMA instructions per slice the same for every wave (s_waitcnt vmcnt below)
        const int pos = (i < 2 ? wave + 4 * i : 8 + (wave & 1)) * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / W4_PHASE, sidx = cell - phase * W4_PHASE;
        const int m = 4 * (sidx >> log2d) + phase;
        const int row = (m << log2d) + (sidx & (d - 1)) + W4_HALO - d;       // staged row index, m0 - 16 + row = source
        const int src = m0 - W4_HALO + row;
        a_ch[i] = 4 * ((pos & 1) ^ ((cell >> 3) & 1));
        a_off[i] = min(max(src, 0), rows - 1) * p.ldx;
        if (row < W4_AROWS && src >= 0 && src < rows) a_ok |= 1u << i;
        a_voff[i] = 4u * (unsigned)(a_off[i] + a_ch[i]);
    }
    const float *wsrc = p.w + (long long)nt * nk * W4_B_FLOATS + (wave * 64 + lane) * 4;
    // interior blocks (every staged row exists, whole slices): uniform base + per-lane byte offset, no selects
    const bool fast = p.fast_dma && m0 >= W4_HALO && m0 + W4_ROWS + W4_HALO <= rows && p.cin % W4_BK == 0;
    const float *wtile = p.w + (long long)nt * nk * W4_B_FLOATS;
    const unsigned b_voff = 16u * (unsigned)lane;
    auto issue = [&](int kt, int stage) {
        const int ci0 = kt * W4_BK;
        const unsigned adst = lds_base + 4u * (unsigned)(stage * W4_STAGE);
        const unsigned bdst = adst + 4u * (unsigned)W4_A_FLOATS;
        if (fast) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                w4_lds_dma16_s(abase, a_voff[i], adst + 1024u * (unsigned)(i < 2 ? wave + 4 * i : 8 + (wave & 1)));
            const float *bbase = wtile + (long long)kt * W4_B_FLOATS + wave * 256;
#pragma unroll
            for (int i = 0; i < W4_B_INST; ++i)
                w4_lds_dma16_s(bbase + i * 1024, b_voff, bdst + 1024u * (unsigned)(wave + 4 * i));
            return;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ci = ci0 + a_ch[i];
            const bool ok = ((a_ok >> i) & 1u) & (ci < p.cin);
            w4_lds_dma16(ok ? xb + a_off[i] + ci : p.zeros,
                         adst + 1024u * (unsigned)(i < 2 ? wave + 4 * i : 8 + (wave & 1)));
        }
#pragma unroll
        for (int i = 0; i < W4_B_INST; ++i)
            w4_lds_dma16(wsrc + (long long)kt * W4_B_FLOATS + i * 1024, bdst + 1024u * (unsigned)(wave + 4 * i));
    };
    // conditioning rows of this block (<= 32 rows x (32 tanh | 32 sigmoid) columns) -> a free stage near the end
    const int cond_up = p.cond_up;
    const int n2 = rows / cond_up;
    const int t2base = m0 / cond_up;
    const float *cbase = p.cond + (long long)b * p.cond_bstride;
    auto issue_cond = [&](int stage) {
        const unsigned cdst = lds_base + 4u * (unsigned)(stage * W4_STAGE);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = (wave + 4 * i) * 64 + lane;
            const int crow = pos >> 4, cq = pos & 15;
            const int chn = n0 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            w4_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                         cdst + 1024u * (unsigned)(wave + 4 * i));
        }
    };

    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // group of this lane (A operand row): Q = 32*wm + lrow -> t = m0 + 4 d (Q >> log2d) + (Q & (d-1))
    const int grp = 32 * wm + lrow;
    int aoff[6];        // LDS float offsets (inside a stage) of h[t-d] .. h[t+4d], this lane's 4 channels
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int cell = (q & 3) * W4_PHASE + grp + ((q >> 2) << log2d);
        aoff[q] = 4 * (2 * cell + (lk ^ ((cell >> 3) & 1)));
    }
    float4 X0[6], X1[6];
    float4 B0[2], B1[2];
    auto load_x = [&](int stage, float4 (&x)[6]) {
        const float *ab = lds + stage * W4_STAGE;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float4 *>(ab + aoff[q]);
    };
    auto load_b = [&](int stage, int g, float4 (&bw)[2]) {
        const float *bb = lds + stage * W4_STAGE + W4_A_FLOATS + lane * 4;
        bw[0] = *reinterpret_cast<const float4 *>(bb + ((2 * g) * 2 + wn) * 256);
        bw[1] = *reinterpret_cast<const float4 *>(bb + ((2 * g + 1) * 2 + wn) * 256);
    };
    auto mfma8 = [&](int g, const float4 &u0, const float4 &u1, const float4 (&bw)[2]) {
        f32x16 &c0 = acc[2 * g];
        f32x16 &c1 = acc[2 * g + 1];
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.x, bw[0].x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.x, bw[1].x, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.y, bw[0].y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.y, bw[1].y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.z, bw[0].z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.z, bw[1].z, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.w, bw[0].w, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.w, bw[1].w, c1, 0, 0, 0);
    };
    // the input combinations of the next group are computed in the shadow of this group's MFMAs
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
    };
    // One slice = three operand groups (product pairs) of 8 MFMAs.  The LDS operands of group n+1 are requested before
    // the MFMAs of group n issue and its input combinations are formed between those MFMAs.  Groups alternate between
    // the two register sets (ba, ua) / (bb, ub), so with three groups per slice the roles swap from slice to slice.
    // In: ua = combinations of (this slice, pair 0), ba = its weights.  Out: ub, bb = those of (next slice, pair 0).
    // The MFMA phases are straight-line code (the last slice requests "next" operands too: they are valid LDS addresses
    // and never used); only the LDS-DMA requests and the wait in front of the barrier depend on the slice index.
    // 6 LDS-DMA instructions per wave and slice, 2 for the conditioning tile.
    auto slice = [&](int kt, int stage, float4 (&xc)[6], float4 (&xn)[6], float4 (&ba)[2], float4 (&bb)[2],
                     float4 (&ua)[2], float4 (&ub)[2]) {
        const int nstage = stage == 2 ? 0 : stage + 1;
        // ---- pair 0
        load_b(stage, 1, bb);
        __builtin_amdgcn_sched_barrier(0);
        w4_input_comb(1, xc, ub[0], ub[1]);
        mfma8(0, ua[0], ua[1], ba);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // ---- pair 1; before it, the barrier that publishes slice kt+1 (every wave has requested all of slice kt by now)
        load_b(stage, 2, ba);
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // slice kt+2 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 3 < nk) issue(kt + 3, stage);
        else if (kt + 3 == nk) issue_cond(stage);
        load_x(nstage, xn);
        __builtin_amdgcn_sched_barrier(0);
        w4_input_comb(2, xc, ua[0], ua[1]);
        mfma8(1, ub[0], ub[1], bb);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        // ---- pair 2
        load_b(nstage, 0, bb);
        __builtin_amdgcn_sched_barrier(0);
        w4_input_comb(0, xn, ub[0], ub[1]);
        mfma8(2, ua[0], ua[1], ba);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: three slices in flight, the first one landed (the launcher guarantees nk >= 4)
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __syncthreads();
    float4 U0[2], U1[2];
    load_x(0, X0);
    load_b(0, 0, B0);
    w4_input_comb(0, X0, U0[0], U0[1]);
    {
        int stage = 0;
        for (int kt = 0; kt < nk; kt += 2) {
            slice(kt, stage, X0, X1, B0, B1, U0, U1);
            stage = stage == 2 ? 0 : stage + 1;
            if (kt + 1 < nk) {
                slice(kt + 1, stage, X1, X0, B1, B0, U1, U0);
                stage = stage == 2 ? 0 : stage + 1;
            }
        }
    }

    // ---- epilogue: combine the six products, add bias + conditioning, gate, store the four outputs of the group
    // the conditioning tile sits in the stage that held slice nk-3 (issued when slice nk-3 was done), or is loaded now
    const int cstage = (nk - 3) % 3;
    float *lerp_lds = lds + cstage * W4_STAGE + 2048;
    if (tid < cond_up) {
        lerp_lds[tid] = p.lerp_w0[tid];
        lerp_lds[64 + tid] = p.lerp_w1[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float inv_up = 1.0f / (float)cond_up;
    const float *cl = lds + cstage * W4_STAGE;
    float *obase = p.out + (long long)b * p.out_bstride;
    auto finish = [&](int row, int tc, float yt, float ys, float bt, float bsg) {
        int t2 = (int)((float)row * inv_up);                       // row / cond_up (rows < 2^24)
        int u = row - t2 * cond_up;
        if (u < 0) { --t2; u += cond_up; }
        if (u >= cond_up) { ++t2; u -= cond_up; }
        const float w0 = lerp_lds[u], w1 = lerp_lds[64 + u];
        const float *c0 = cl + (t2 - t2base) * 64 + tc;
        const float zt = (yt + bt) + (c0[0] * w0 + c0[64] * w1);
        const float zs = (ys + bsg) + (c0[32] * w0 + c0[96] * w1);
        obase[(long long)row * p.ldo + n0 + tc] = w4_gate_act(zt, zs);
    };
    // lanes 0..15 of each half-wave hold the tanh column of tile channel 16 wn + (lrow & 15), lanes 16..31 its sigmoid
    // column: tanh lanes finish y[t], y[t+d], sigmoid lanes y[t+2d], y[t+3d]; each sends the other the two values it
    // does not finish
    const bool tanh_lane = lrow < 16;
    const int tc = 16 * wn + (lrow & 15);
    const bool ch_ok = n0 + tc < C;
    const float bt = (p.bias && ch_ok) ? p.bias[n0 + tc] : 0.f;
    const float bsg = (p.bias && ch_ok) ? p.bias[C + n0 + tc] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gi = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk;                   // group held by this register
        const int t0 = m0 + ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));
        const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
        const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
        const float y0 = (acc[0][r] + s12) + s34;
        const float y1 = fmaf(2.f, d34, d12);
        const float y2 = fmaf(4.f, s34, s12);
        const float y3 = fmaf(8.f, d34, d12) + acc[5][r];
        const float ga = __shfl_xor(tanh_lane ? y2 : y0, 16);
        const float gb = __shfl_xor(tanh_lane ? y3 : y1, 16);
        const int ra = t0 + (tanh_lane ? 0 : 2 * d), rb = ra + d;
        if (ch_ok && ra < rows) finish(ra, tc, tanh_lane ? y0 : ga, tanh_lane ? ga : y2, bt, bsg);
        if (ch_ok && rb < rows) finish(rb, tc, tanh_lane ? y1 : gb, tanh_lane ? gb : y3, bt, bsg);
    }
}

// a.w must point at the host-packed F(4,3) weights (ceil(C/32), ceil(C/8), 3072); returns false if the layer does not fit
bool launch_wn_gate_winograd4(const ConvArgs &a, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= W4_HALO && a.cin >= 4 * W4_BK && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros &&
                    a.cond && (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up <= 64 &&
                    W4_ROWS / a.cond_up + 2 <= 32 && a.max_rows < (1 << 24);
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = (long long)a.max_rows * a.ldx * 4 < (1LL << 32);
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + W4_ROWS - 1) / W4_ROWS;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    hipLaunchKernelGGL(wn_gate_winograd4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    return true;
}

}  // namespace mbx
