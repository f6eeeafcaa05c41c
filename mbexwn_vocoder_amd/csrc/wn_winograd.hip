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
// Block = 4 waves, 256 consecutive output rows (128 pairs; wave w owns pairs 32w..32w+31) x 32 gate channels.
// Pair P of the block: q = P / d, r = P % d, t = m0 + 2 d q + r (d is a power of two <= 16).
// Per K slice of 16 channels the block stages, through LDS-DMA (see lds_dma16 in conv_mfma.hip):
//   A: activation rows [m0-32, m0+288) x 16 channels, chunk (row, c) at position 4*row + (c ^ ((row>>2)&3))
//   B: 4 weight combinations x 16 channels x 64 columns (32 tanh | 32 sigmoid), k-major
// Two LDS stages (72 KB per block, 2 blocks per CU); accumulators: 4 products x (tanh, sigmoid) x 16 = 128 VGPRs.
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_ROWS = 256;       // output rows per block
constexpr int WG_HALO = 32;        // staged rows in front of m0 (>= max dilation)
constexpr int WG_AROWS = 320;      // staged rows per slice (WG_HALO + 256 + 32)
constexpr int WG_BK = 16;
constexpr int WG_A_FLOATS = WG_AROWS * WG_BK;          // 5120
constexpr int WG_B_FLOATS = 4 * WG_BK * 64;            // 4096
constexpr int WG_A_INST = WG_AROWS * 4 / 64 / 4;       // 5 LDS-DMA instructions per wave (A)
constexpr int WG_B_INST = 4 * WG_BK * 16 / 64 / 4;     // 4 per wave (B)

__device__ __forceinline__ void wg_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

__device__ __forceinline__ float wg_gate_act(float zt, float zs) {
    const float e2 = __expf(2.0f * zt);
    const float e1 = __expf(-zs);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
    return th * __builtin_amdgcn_rcpf(1.0f + e1);
}

__global__ __launch_bounds__(256, 2) void wn_gate_winograd_kernel(ConvArgs p, int log2d) {
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[2 * (WG_A_FLOATS + WG_B_FLOATS)];   // A0 A1 B0 B1
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
    const int m0 = mt * WG_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nk = (p.cin + WG_BK - 1) / WG_BK;

    // ---- per-lane DMA sources (fixed for the whole kernel except the channel offset)
    int a_off[WG_A_INST], a_ch[WG_A_INST];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < WG_A_INST; ++i) {
        const int pos = (wave + 4 * i) * 64 + lane;
        const int row = pos >> 2;
        const int src = m0 - WG_HALO + row;
        a_ch[i] = 4 * ((pos & 3) ^ ((row >> 2) & 3));
        a_off[i] = max(src, 0) * p.ldx;
        if (src >= 0 && src < rows) a_ok |= 1u << i;
    }
    long long b_off[WG_B_INST];
    int b_k[WG_B_INST];
    unsigned b_ok = 0;
#pragma unroll
    for (int i = 0; i < WG_B_INST; ++i) {
        const int pos = (wave + 4 * i) * 64 + lane;          // 0..1023 = (j, k, column quad)
        const int j = pos >> 8, k = (pos >> 4) & 15, c = (pos & 15) * 4;
        const int ch = n0 + (c & 31);
        const int gcol = (c < 32 ? 0 : C) + min(ch, C - 4);
        b_k[i] = k;
        b_off[i] = ((long long)j * p.cin + k) * p.cout + gcol;
        if (ch < C) b_ok |= 1u << i;
    }
    auto issue = [&](int kt, int buf) {
        const int ci0 = kt * WG_BK;
        const unsigned adst = lds_base + 4u * (unsigned)(buf * WG_A_FLOATS);
        const unsigned bdst = lds_base + 4u * (unsigned)(2 * WG_A_FLOATS + buf * WG_B_FLOATS);
#pragma unroll
        for (int i = 0; i < WG_A_INST; ++i) {
            const int ci = ci0 + a_ch[i];
            const bool ok = ((a_ok >> i) & 1u) & (ci < p.cin);
            wg_lds_dma16(ok ? xb + a_off[i] + ci : p.zeros, adst + 1024u * (unsigned)(wave + 4 * i));
        }
#pragma unroll
        for (int i = 0; i < WG_B_INST; ++i) {
            const bool ok = ((b_ok >> i) & 1u) & (ci0 + b_k[i] < p.cin);
            wg_lds_dma16(ok ? p.w + b_off[i] + (long long)ci0 * p.cout : p.zeros, bdst + 1024u * (unsigned)(wave + 4 * i));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    // pair of this lane (A operand row): P = 32*wave + lrow -> t = m0 + 2 d (P >> log2d) + (P & (d-1))
    const int pair = 32 * wave + lrow;
    const int trel = WG_HALO + ((pair >> log2d) << (log2d + 1)) + (pair & (d - 1));   // LDS row of h[t]
    int arow[4];        // LDS rows of h[t-d], h[t], h[t+d], h[t+2d]
    arow[0] = trel - d;
    arow[1] = trel;
    arow[2] = trel + d;
    arow[3] = trel + 2 * d;

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        const float *ab = lds + buf * WG_A_FLOATS;
        const float *bb = lds + 2 * WG_A_FLOATS + buf * WG_B_FLOATS + lrow;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                x[q] = *reinterpret_cast<const float4 *>(ab + arow[q] * WG_BK + 4 * ((2 * cc + lk) ^ ((arow[q] >> 2) & 3)));
            float u[4][4];     // [product][k step]
            u[0][0] = x[0].x - x[2].x; u[0][1] = x[0].y - x[2].y; u[0][2] = x[0].z - x[2].z; u[0][3] = x[0].w - x[2].w;
            u[1][0] = x[1].x + x[2].x; u[1][1] = x[1].y + x[2].y; u[1][2] = x[1].z + x[2].z; u[1][3] = x[1].w + x[2].w;
            u[2][0] = x[2].x - x[1].x; u[2][1] = x[2].y - x[1].y; u[2][2] = x[2].z - x[1].z; u[2][3] = x[2].w - x[1].w;
            u[3][0] = x[1].x - x[3].x; u[3][1] = x[1].y - x[3].y; u[3][2] = x[1].z - x[3].z; u[3][3] = x[1].w - x[3].w;
            // weight operands of step st+1 are requested before the 8 MFMAs of step st issue
            float bv[2][4][2];
            {
                const int k = 8 * cc + 4 * lk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bv[0][j][0] = bb[(j * WG_BK + k) * 64];
                    bv[0][j][1] = bb[(j * WG_BK + k) * 64 + 32];
                }
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int cur = st & 1, nxt = cur ^ 1;
                if (st + 1 < 4) {
                    const int k = 8 * cc + 4 * lk + st + 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bv[nxt][j][0] = bb[(j * WG_BK + k) * 64];
                        bv[nxt][j][1] = bb[(j * WG_BK + k) * 64 + 32];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[j][st], bv[cur][j][0], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[j][st], bv[cur][j][1], acc[j][1], 0, 0, 0);
                }
                if (st + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
    }

    // ---- epilogue: combine the four products, add bias + conditioning, gate, store both outputs of the pair
    const int ch = n0 + lrow;
    if (ch >= C) return;
    const float bt = p.bias ? p.bias[ch] : 0.f;
    const float bsg = p.bias ? p.bias[C + ch] : 0.f;
    const float *cb = p.cond + (long long)b * p.cond_bstride + ch;
    const int n2 = rows / p.cond_up;
    float *ob = p.out + (long long)b * p.out_bstride + ch;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pi = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lk;          // pair held by this register
        const int t0 = m0 + ((pi >> log2d) << (log2d + 1)) + (pi & (d - 1));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = t0 + half * d;
            if (row < rows) {
                const float yt = half == 0 ? (acc[0][0][r] + acc[1][0][r]) + acc[2][0][r]
                                           : (acc[1][0][r] - acc[2][0][r]) - acc[3][0][r];
                const float ys = half == 0 ? (acc[0][1][r] + acc[1][1][r]) + acc[2][1][r]
                                           : (acc[1][1][r] - acc[2][1][r]) - acc[3][1][r];
                const int t2 = row / p.cond_up, u = row - t2 * p.cond_up;
                const int t3 = min(t2 + 1, n2 - 1);
                const float w0 = p.lerp_w0[u], w1 = p.lerp_w1[u];
                const float *c0 = cb + t2 * (2 * C);
                const float *c1 = cb + t3 * (2 * C);
                const float zt = (yt + bt) + (c0[0] * w0 + c1[0] * w1);
                const float zs = (ys + bsg) + (c0[C] * w0 + c1[C] * w1);
                ob[row * p.ldo] = wg_gate_act(zt, zs);
            }
        }
    }
}

// a.w must point at the host-transformed weights (4, cin, 2C); returns false if the layer does not fit the kernel
bool launch_wn_gate_winograd(const ConvArgs &a, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= 16 && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros;
    if (!ok) return false;
    ConvArgs r = a;
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + WG_ROWS - 1) / WG_ROWS;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    hipLaunchKernelGGL(wn_gate_winograd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    return true;
}

}  // namespace mbx
