// WaveNet dilated convolution (k = 3, dilation d) + conditioning + tanh*sigmoid in Winograd F(2,3) form.
//
// Same layer as conv1d_mfma_dma_kernel<EPI_GATE> (reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:305-321:
// in_layered = conv1D_l(h); z = in_layered + cond; a = tanh(z[:C]) * sigmoid(z[C:])), but two outputs that are d steps
// apart share their products.  With x0..x3 = h[t-d], h[t], h[t+d], h[t+2d] and the taps W0, W1, W2 (each C x 2C):
//     m1 = (x0 - x2) W0            m2 = (x1 + x2) (W0 + W1 + W2)/2
//     m3 = (x2 - x1) (W0 - W1 + W2)/2      m4 = (x1 - x3) W2
//     y[t] = m1 + m2 + m3          y[t+d] = m2 - m3 - m4
// = 4 channel contractions per output pair instead of 6: the matrix-core work of the layer drops by 1/3.  The four
// weight combinations are formed once on the host (engine.tensor_table); the input combinations are formed in
// registers from the raw activation rows staged in LDS.  Results equal the direct form up to float32 rounding.
//
// Two block shapes (template parameter NS = number of wave columns):
//   NS = 1: 4 waves x 1, 256 consecutive output rows (128 pairs; wave w owns pairs 32w..32w+31) x 32 gate channels,
//           8 accumulator tiles per wave (4 products x (tanh, sigmoid)); the shape for large row counts.
//   NS = 2: 2 waves x 2, 128 rows x 32 gate channels, wave column wn owns 16 gate channels as one 32-wide MFMA tile
//           [16 tanh | 16 sigmoid], 4 accumulator tiles per wave; half-size tiles for small problems, where a launch is
//           only a few rounds of blocks and the granularity of the last round decides the run time.
// Pair P of the block: q = P / d, r = P % d, t = m0 + 2 d q + r (d is a power of two <= 16).
// Per K slice of 16 channels the block stages, through LDS-DMA (see lds_dma16 in conv_mfma.hip):
//   A: activation rows [m0-32, m0+ROWS+32) x 16 channels, chunk (row, c) at position 4*row + (c ^ ((row>>2)&3))
//   B: 4 weight combinations x 16 channels x 64 columns, pre-packed on the host in MFMA operand order
//      [product j][channel half cc][column half h][lane][4 k steps] so that one ds_read_b128 per lane yields the weight
//      operands of four consecutive MFMAs (engine.pack_winograd_weights; h = tanh|sigmoid for NS = 1, h = wn for NS = 2)
// Two LDS stages (72 KB per block, 2 blocks per CU / 52 KB, 3 per CU).
#include <cstdlib>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_BK = 16;
constexpr int WG_B_FLOATS = 4 * WG_BK * 64;            // 4096
constexpr int WG_B_INST = WG_B_FLOATS / 4 / 64 / 4;    // 4 LDS-DMA instructions per wave (B)

__device__ __forceinline__ void wg_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset
__device__ __forceinline__ void wg_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

__device__ __forceinline__ float wg_gate_act(float zt, float zs) {
    const float e2 = __expf(2.0f * zt);
    const float e1 = __expf(-zs);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
    return th * __builtin_amdgcn_rcpf(1.0f + e1);
}

// input combination of product j from the four activation rows x0..x3 (4 channels = 4 k steps)
template <int J>
__device__ __forceinline__ float4 wg_input_comb(const float4 (&x)[4]) {
    if (J == 0) return make_float4(x[0].x - x[2].x, x[0].y - x[2].y, x[0].z - x[2].z, x[0].w - x[2].w);
    if (J == 1) return make_float4(x[1].x + x[2].x, x[1].y + x[2].y, x[1].z + x[2].z, x[1].w + x[2].w);
    if (J == 2) return make_float4(x[2].x - x[1].x, x[2].y - x[1].y, x[2].z - x[1].z, x[2].w - x[1].w);
    return make_float4(x[1].x - x[3].x, x[1].y - x[3].y, x[1].z - x[3].z, x[1].w - x[3].w);
}
__device__ __forceinline__ float4 wg_input_comb(int j, const float4 (&x)[4]) {
    return j == 0 ? wg_input_comb<0>(x) : j == 1 ? wg_input_comb<1>(x) : j == 2 ? wg_input_comb<2>(x) : wg_input_comb<3>(x);
}

template <int NS>
__global__ __launch_bounds__(256, NS == 1 ? 2 : 3) void wn_gate_winograd_kernel(ConvArgs p, int log2d) {
    constexpr int MW = 4 / NS;                 // wave rows
    constexpr int ROWS = 64 * MW;              // output rows per block
    constexpr int HALO = NS == 1 ? 32 : 16;    // staged rows in front of / behind the block (>= max dilation)
    constexpr int AROWS = ROWS + 2 * HALO;     // staged rows per slice (NS = 2: 160 rows, 52 KB per block, 3 per CU)
    constexpr int A_FLOATS = AROWS * WG_BK;
    constexpr int A_CHUNKS = AROWS / 16;       // 1 KB LDS-DMA instructions per slice (A), dealt round-robin to the waves
    constexpr int A_INST = (A_CHUNKS + 3) / 4;
    constexpr int NT = 2 / NS;                 // accumulator column tiles per product
    constexpr int NG = 8 / NS;                 // operand groups (8 MFMAs each) per slice
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[2 * (A_FLOATS + WG_B_FLOATS)];   // A0 A1 B0 B1
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode (see decode_tile in conv_mfma.hip): XCD x takes row blocks x, x+8, ... and walks their column tiles
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    if (g >= p.m_tiles_total) return;
    const int b = g / p.m_tiles_per_item;
    const int mt = g - b * p.m_tiles_per_item;
    const int rows = p.n_frames ? p.n_frames[b] * p.rows_per_frame : p.max_rows;
    const int m0 = mt * ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % MW, wn = wave / MW;
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nk = (p.cin + WG_BK - 1) / WG_BK;

    // ---- per-lane DMA sources (fixed for the whole kernel except the channel offset)
    int a_off[A_INST], a_ch[A_INST];
    unsigned a_voff[A_INST];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int pos = (wave + 4 * i) * 64 + lane;
        const int row = pos >> 2;
        const int src = m0 - HALO + row;
        a_ch[i] = 4 * ((pos & 3) ^ ((row >> 2) & 3));
        a_off[i] = max(src, 0) * p.ldx;
        if (src >= 0 && src < rows) a_ok |= 1u << i;
        a_voff[i] = 4u * (unsigned)(min(max(src, 0), rows - 1) * p.ldx + a_ch[i]);
    }
    // interior blocks (every staged row exists, whole slices): uniform base + per-lane byte offset, no selects
    const bool fast = p.fast_dma && m0 >= HALO && m0 + ROWS + HALO <= rows && p.cin % WG_BK == 0;
    const float *wtile = p.w + (long long)nt * nk * WG_B_FLOATS + wave * 256;
    const unsigned b_voff = 16u * (unsigned)lane;
    // weights: the packed image of (column tile nt, slice kt) is copied verbatim, 16 KB = 4 x 1 KB per wave
    const float *wsrc = p.w + (long long)nt * nk * WG_B_FLOATS + (wave * 64 + lane) * 4;
    auto issue = [&](int kt, int buf) {
        const int ci0 = kt * WG_BK;
        const unsigned adst = lds_base + 4u * (unsigned)(buf * A_FLOATS);
        const unsigned bdst = lds_base + 4u * (unsigned)(2 * A_FLOATS + buf * WG_B_FLOATS);
        if (fast) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < A_INST; ++i) {
                if (A_CHUNKS % 4 != 0 && wave + 4 * i >= A_CHUNKS) continue;  // wave-uniform
                wg_lds_dma16_s(abase, a_voff[i], adst + 1024u * (unsigned)(wave + 4 * i));
            }
            const float *bbase = wtile + (long long)kt * WG_B_FLOATS;
#pragma unroll
            for (int i = 0; i < WG_B_INST; ++i)
                wg_lds_dma16_s(bbase + i * 1024, b_voff, bdst + 1024u * (unsigned)(wave + 4 * i));
            return;
        }
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            if (A_CHUNKS % 4 != 0 && wave + 4 * i >= A_CHUNKS) continue;      // wave-uniform
            const int ci = ci0 + a_ch[i];
            const bool ok = ((a_ok >> i) & 1u) & (ci < p.cin);
            wg_lds_dma16(ok ? xb + a_off[i] + ci : p.zeros, adst + 1024u * (unsigned)(wave + 4 * i));
        }
#pragma unroll
        for (int i = 0; i < WG_B_INST; ++i)
            wg_lds_dma16(wsrc + (long long)kt * WG_B_FLOATS + i * 1024, bdst + 1024u * (unsigned)(wave + 4 * i));
    };

    // conditioning rows of this block (<= 32 rows x (32 tanh | 32 sigmoid) columns) go to a free A stage during the last
    // slice, so that the epilogue reads them from LDS: row index = cond row - t2base, rows clamped to the last one
    // (p.cond_phase: conditioning-rate position of item row 0 inside its conditioning row, for items that start
    // between two conditioning rows: the per-layer regions of a streaming tick)
    const int cond_up = p.cond_up;
    const int cphase = p.cond_phase;
    const int n2 = (rows + cphase) / cond_up;
    const int t2base = (m0 + cphase) / cond_up;
    const float *cbase = p.cond + (long long)b * p.cond_bstride;
    auto issue_cond = [&](int buf) {
        const unsigned cdst = lds_base + 4u * (unsigned)(buf * A_FLOATS);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = (wave + 4 * i) * 64 + lane;
            const int crow = pos >> 4, cq = pos & 15;
            const int chn = n0 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            wg_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                         cdst + 1024u * (unsigned)(wave + 4 * i));
        }
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    // pair of this lane (A operand row): P = 32*wm + lrow -> t = m0 + 2 d (P >> log2d) + (P & (d-1))
    const int pair = 32 * wm + lrow;
    const int trel = HALO + ((pair >> log2d) << (log2d + 1)) + (pair & (d - 1));   // LDS row of h[t]
    int aoff[4][2];     // LDS float offsets of h[t-d], h[t], h[t+d], h[t+2d] for the two channel halves of a slice
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = trel + (q - 1) * d;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) aoff[q][cc] = row * WG_BK + 4 * ((2 * cc + lk) ^ ((row >> 2) & 3));
    }

    // Operand groups of 8 MFMAs on two accumulator tiles: NS = 1: (channel half cc, product j) x (tanh, sigmoid);
    // NS = 2: (cc, product pair).  The operands of group n+1 are requested from LDS before the MFMAs of group n issue;
    // the last group of a slice first passes the barrier that publishes the next slice.
    float4 X[2][4];
    float4 Bv[2][2];
    auto load_x = [&](int buf, int cc, float4 (&x)[4]) {
        const float *ab = lds + buf * A_FLOATS;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const float4 *>(ab + aoff[q][cc]);
    };
    auto load_b = [&](int buf, int gi, float4 (&bw)[2]) {
        const int cc = gi / (NG / 2), q = gi % (NG / 2);
        const float *bb = lds + 2 * A_FLOATS + buf * WG_B_FLOATS + lane * 4;
        if (NS == 1) {
            bw[0] = *reinterpret_cast<const float4 *>(bb + ((q * 2 + cc) * 2 + 0) * 256);
            bw[1] = *reinterpret_cast<const float4 *>(bb + ((q * 2 + cc) * 2 + 1) * 256);
        } else {
            bw[0] = *reinterpret_cast<const float4 *>(bb + (((2 * q) * 2 + cc) * 2 + wn) * 256);
            bw[1] = *reinterpret_cast<const float4 *>(bb + (((2 * q + 1) * 2 + cc) * 2 + wn) * 256);
        }
    };

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nk > 1) issue(1, 1);
    load_x(0, 0, X[0]);
    load_b(0, 0, Bv[0]);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int cc = gi / (NG / 2), q = gi % (NG / 2);
            if (gi < NG - 1) {
                load_b(buf, gi + 1, Bv[(gi + 1) & 1]);
                if (gi == NG / 2 - 1) load_x(buf, 1, X[1]);
            } else if (kt + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kt + 2 < nk) {
                    issue(kt + 2, buf);
                } else {
                    issue_cond(buf);
                }
                load_b(buf ^ 1, 0, Bv[0]);
                load_x(buf ^ 1, 0, X[0]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the requests ahead of this group's MFMAs
            const float4(&bw)[2] = Bv[gi & 1];
            const float4 u0 = wg_input_comb(NS == 1 ? q : 2 * q, X[cc]);
            const float4 u1 = NS == 1 ? u0 : wg_input_comb(2 * q + 1, X[cc]);
            f32x16 &c0 = NS == 1 ? acc[q][0] : acc[2 * q][0];
            f32x16 &c1 = NS == 1 ? acc[q][NT - 1] : acc[2 * q + 1][0];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.x, bw[0].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.x, bw[1].x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.y, bw[0].y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.y, bw[1].y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.z, bw[0].z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.z, bw[1].z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u0.w, bw[0].w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1.w, bw[1].w, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: combine the four products, add bias + conditioning, gate, store both outputs of the pair
    const int cbuf = nk & 1;                       // stage that held slice nk-2 (the conditioning tile now)
    float *lerp_lds = lds + cbuf * A_FLOATS + 2048;
    if (nk < 2) issue_cond(cbuf);
    if (tid < cond_up) {
        lerp_lds[tid] = p.lerp_w0[tid];
        lerp_lds[64 + tid] = p.lerp_w1[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float inv_up = 1.0f / (float)cond_up;
    const float *cl = lds + cbuf * A_FLOATS;
    float *obase = p.out + (long long)b * p.out_bstride;
    // conditioning (tanh | sigmoid column of tile channel tc) interpolated at output row `row`
    auto finish = [&](int row, int tc, float yt, float ys, float bt, float bsg) {
        const int crow = row + cphase;
        int t2 = (int)((float)crow * inv_up);                      // crow / cond_up (rows < 2^24)
        int u = crow - t2 * cond_up;
        if (u < 0) { --t2; u += cond_up; }
        if (u >= cond_up) { ++t2; u -= cond_up; }
        const float w0 = lerp_lds[u], w1 = lerp_lds[64 + u];
        const float *c0 = cl + (t2 - t2base) * 64 + tc;
        const float zt = (yt + bt) + (c0[0] * w0 + c0[64] * w1);
        const float zs = (ys + bsg) + (c0[32] * w0 + c0[96] * w1);
        obase[(long long)row * p.ldo + n0 + tc] = wg_gate_act(zt, zs);
    };
    if (NS == 1) {
        const int ch = n0 + lrow;
        if (ch >= C) return;
        const float bt = p.bias ? p.bias[ch] : 0.f;
        const float bsg = p.bias ? p.bias[C + ch] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pi = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk;          // pair held by this register
            const int t0 = m0 + ((pi >> log2d) << (log2d + 1)) + (pi & (d - 1));
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = t0 + half * d;
                if (row < rows) {
                    const float yt = half == 0 ? (acc[0][0][r] + acc[1][0][r]) + acc[2][0][r]
                                               : (acc[1][0][r] - acc[2][0][r]) - acc[3][0][r];
                    const float ys = half == 0 ? (acc[0][NT - 1][r] + acc[1][NT - 1][r]) + acc[2][NT - 1][r]
                                               : (acc[1][NT - 1][r] - acc[2][NT - 1][r]) - acc[3][NT - 1][r];
                    finish(row, lrow, yt, ys, bt, bsg);
                }
            }
        }
    } else {
        // lanes 0..15 of each half-wave hold the tanh column of tile channel 16 wn + (lrow & 15), lanes 16..31 its
        // sigmoid column: they swap the half they do not finish (tanh lanes finish y[t], sigmoid lanes y[t+d])
        const bool tanh_lane = lrow < 16;
        const int tc = 16 * wn + (lrow & 15);
        const bool ch_ok = n0 + tc < C;
        const float bt = (p.bias && ch_ok) ? p.bias[n0 + tc] : 0.f;
        const float bsg = (p.bias && ch_ok) ? p.bias[C + n0 + tc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pi = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int t0 = m0 + ((pi >> log2d) << (log2d + 1)) + (pi & (d - 1));
            const float y0 = (acc[0][0][r] + acc[1][0][r]) + acc[2][0][r];      // y[t] of this lane's column
            const float y1 = (acc[1][0][r] - acc[2][0][r]) - acc[3][0][r];      // y[t+d]
            const float got = __shfl_xor(tanh_lane ? y1 : y0, 16);
            const int row = t0 + (tanh_lane ? 0 : d);
            if (ch_ok && row < rows) finish(row, tc, tanh_lane ? y0 : got, tanh_lane ? got : y1, bt, bsg);
        }
    }
}

// a.w / w_split: host-packed Winograd weights (ceil(C/32), ceil(C/16), 4096) for the NS = 1 / NS = 2 block shape
// (w_split may be null); returns false if the layer does not fit the kernel
bool launch_wn_gate_winograd(const ConvArgs &a, const float *w_split, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= 16 /* = HALO of the small shape */ && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros &&
                    a.cond && (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up <= 64 &&
                    (256 + a.cond_up - 1) / a.cond_up + 2 <= 32 && a.max_rows < (1 << 24) && a.cond_phase >= 0 &&
                    a.cond_phase < a.cond_up;
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = (long long)a.max_rows * a.ldx * 4 < (1LL << 32);
    r.n_tiles = (a.channels + 31) / 32;
    // half-size tiles while the full-size grid is less than four rounds of the 512 resident blocks (2 per CU x 256 CUs)
    const long long full_blocks = (long long)((a.max_rows + 255) / 256) * a.batch * r.n_tiles;
    const bool split = w_split && (uintptr_t)w_split % 16 == 0 && full_blocks < 4 * 512;
    const int tile_rows = split ? 128 : 256;
    r.m_tiles_per_item = (a.max_rows + tile_rows - 1) / tile_rows;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    if (split) {
        r.w = w_split;
        hipLaunchKernelGGL(wn_gate_winograd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    } else {
        hipLaunchKernelGGL(wn_gate_winograd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    }
    return true;
}

}  // namespace mbx
