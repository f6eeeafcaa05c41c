// First WaveNet layer with the start convolution folded into it.
//
// Reference (MBExWN_NVoc/vocoder/model/custom_AE_layers.py:280, 305-321): h0 = start(x) is a 1x1 convolution of the
// 6-channel excitation x = [pulse folded to 5 channels | sigma * noise]; layer 0 is a dilated k = 3 convolution of h0 (zero
// "SAME" padding) + conditioning + tanh*sigmoid.  Both are linear, so
//     conv0(h0)[t] = sum_tau h0[t + (tau-1) d] W0_tau = sum_tau x'[t + (tau-1) d] (Ws' W0_tau),
// with x' = [x | 1 | 0] (8 channels; the constant channel carries the start bias) and Ws' = [Ws ; bs ; 0]: rows outside
// the item are zero in x' exactly where the reference pads h0 with zeros, so the boundary rows are exact too.  The
// products Ws' W0_tau (8 x 2C per tap) are formed on the host in float64 (engine.fold_start_weights).  The layer then is
// a K = 24 contraction instead of K = 3C: its (rows, C) input tensor h0 is never written or read, and 1/L of the
// WaveNet's dominant matrix work disappears.  The residual path gets h0 the same way: the res/skip layer of layer 0
// contracts [a0 | x'] with [Wr ; Ws'] (wn_resskip.hip, h_init), for which this kernel appends x' (padded to 16 channels)
// to every row of its output.
//
// Block = 4 waves, 256 rows x 32 gate channels; wave w owns rows 64 w .. 64 w + 63 as four 16-row MFMA tiles
// (v_mfma_f32_16x16x4_f32) x 64 weight columns ([16 tanh | 16 sigmoid] of the even and of the odd gate channels, lane n
// <-> gate channels 2n, 2n+1 as in wn_winograd4w.hip).  MFMA step (tau, m) contracts channels {2 kq + m} of tap tau.
// The weights of a column tile are 24 registers per lane (loaded once per wave).  The block first stages in LDS: the
// rows x'[m0 - d, m0 + 256 + d) (32 bytes each, 16-byte chunk c at 2*row + (c ^ ((row>>3)&1)): bank-conflict free 8-byte
// operand reads), the conditioning rows of the block (<= 32 x (32 tanh | 32 sigmoid)) and the per-row interpolation
// tables, as the epilogue of wn_winograd4w.hip has them.  The kernel is bound by the vector work of the gate activation
// and by writing a0, not by the matrix cores.
#include <cstdlib>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G0_ROWS = 256;

constexpr int G0_MAX_DIL = 16;
constexpr int G0_XROWS = G0_ROWS + 2 * G0_MAX_DIL;      // staged rows of x'
constexpr int G0_COND_ROWS = 32;

__global__ __launch_bounds__(256) void wn_gate0_kernel(Gate0Args p) {
    __shared__ __attribute__((aligned(16))) float xs[G0_XROWS * 8];        // x' rows m0 - d .. (staged row i = m0 - d + i)
    __shared__ __attribute__((aligned(16))) float cs[G0_COND_ROWS * 64];   // conditioning rows t2base ..
    __shared__ float2 tabw[G0_ROWS];                                        // (w0, w1) of block row lr
    __shared__ int tabo[G0_ROWS];                                           // float offset of its conditioning row in cs
    const int id = blockIdx.x;
    const int nt = id % p.n_tiles;
    const int g = id / p.n_tiles;
    const int b = g / p.m_tiles_per_item;
    const int mt = g - b * p.m_tiles_per_item;
    const int rows = p.n_frames ? p.n_frames[b] * p.rows_per_frame : p.max_rows;
    const int m0 = mt * G0_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = p.dil;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const float *pb = p.pulse + (long long)b * p.pulse_bstride;
    const float *nb = p.noise ? p.noise + (long long)b * p.noise_bstride : nullptr;
    const int pc = p.pulse_channels;
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;

    // ---- stage x' (channels: pulse channels | sigma * noise | 1 | 0; rows outside the item are zero)
    for (int idx = tid; idx < (G0_ROWS + 2 * d) * 2; idx += 256) {
        const int i = idx >> 1, half = idx & 1;
        const int srow = m0 - d + i;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (srow >= 0 && srow < rows) {
            float *ov = &o.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = 4 * half + k;
                ov[k] = ch < pc ? pb[(long long)srow * pc + ch] : (ch == pc ? (nb ? p.sigma * nb[srow] : 0.f) : (ch == pc + 1 ? 1.f : 0.f));
            }
        }
        *reinterpret_cast<float4 *>(xs + 8 * i + 4 * (half ^ ((i >> 3) & 1))) = o;
    }
    // ---- conditioning rows of the block (clamped to the item) and the per-row tables
    {
        const int n2 = rows / cond_up;
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
        for (int pos = tid; pos < G0_COND_ROWS * 16; pos += 256) {
            const int crow = pos >> 4, cq = pos & 15;
            const int chn = n0 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (chn < C) v = *reinterpret_cast<const float4 *>(cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn);
            *reinterpret_cast<float4 *>(cs + crow * 64 + 4 * cq) = v;
        }
        const int row = m0 + tid;
        const int t2 = row / cond_up;
        const int u = row - t2 * cond_up;
        tabw[tid] = make_float2(p.lerp_w0[u], p.lerp_w1[u]);
        tabo[tid] = (t2 - t2base) * 64;
    }

    // weights of this column tile: [tap][parity e][lane][tanh m0, tanh m1, sigmoid m0, sigmoid m1]
    float4 bw[3][2];
    {
        const float *wt = p.w + (long long)nt * 1536 + lane * 4;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 2; ++e) bw[t][e] = *reinterpret_cast<const float4 *>(wt + (t * 2 + e) * 256);
    }
    const bool ch_ok = n0 + 2 * r16 < C;                 // C is even: both channels of the lane exist or neither
    float bias[4];                                       // [2 e + (0 tanh | 1 sigmoid)]
#pragma unroll
    for (int c = 0; c < 4; ++c) bias[c] = (p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
    float *ob = p.out + (long long)b * p.out_bstride;
    const float *clane = cs + 2 * r16;
    __syncthreads();

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int l0 = 64 * wave + 16 * i;               // first row of the tile, relative to m0
        float2 xv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int si = l0 + r16 + t * d;             // staged row of tap t: m0 + l0 + r16 + (t - 1) d
            xv[t] = *reinterpret_cast<const float2 *>(xs + 8 * si + 4 * ((kq >> 1) ^ ((si >> 3) & 1)) + 2 * (kq & 1));
        }
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = bias[c];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].x, bw[t][0].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].x, bw[t][0].z, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].x, bw[t][1].x, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].x, bw[t][1].z, acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].y, bw[t][0].y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].y, bw[t][0].w, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].y, bw[t][1].y, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t].y, bw[t][1].w, acc[3], 0, 0, 0);
        }
        // register v of a tile = block row l0 + 4 kq + v, gate channels n0 + 2 r16 (+1)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int lr = l0 + 4 * kq + v;
            const int row = m0 + lr;
            const float2 w = tabw[lr];
            const float *c0 = clane + tabo[lr];
            const float2 ct0 = *reinterpret_cast<const float2 *>(c0), ct1 = *reinterpret_cast<const float2 *>(c0 + 64);
            const float2 cs0 = *reinterpret_cast<const float2 *>(c0 + 32), cs1 = *reinterpret_cast<const float2 *>(c0 + 96);
            float2 res;
            res.x = wn_gate_act(p.gate_act, acc[0][v] + (ct0.x * w.x + ct1.x * w.y), acc[1][v] + (cs0.x * w.x + cs1.x * w.y));
            res.y = wn_gate_act(p.gate_act, acc[2][v] + (ct0.y * w.x + ct1.y * w.y), acc[3][v] + (cs0.y * w.x + cs1.y * w.y));
            if (ch_ok && row < rows) *reinterpret_cast<float2 *>(ob + (long long)row * p.ldo + n0 + 2 * r16) = res;
        }
        // x' (8 channels, padded to 16) behind the C gate channels of the rows of this tile: lane -> row lane / 4, 4 floats
        if (p.write_inputs && nt == 0) {
            const int lr = l0 + (lane >> 2), q = lane & 3;
            const int row = m0 + lr, si = lr + d;
            if (row < rows) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < 2) o = *reinterpret_cast<const float4 *>(xs + 8 * si + 4 * (q ^ ((si >> 3) & 1)));
                *reinterpret_cast<float4 *>(ob + (long long)row * p.ldo + C + 4 * q) = o;
            }
        }
    }
}

// geometry the kernel is built for (mbx_create keeps the un-folded first layer otherwise)
bool wn_gate0_fits(int channels, int pulse_channels, int dil, int cond_up) {
    return pulse_channels >= 1 && pulse_channels + 2 <= 8 && channels % 4 == 0 && cond_up >= 1 && dil >= 1 &&
           dil <= G0_MAX_DIL && (G0_ROWS + cond_up - 2) / cond_up + 2 <= G0_COND_ROWS;
}

// a.w: image of engine.fold_start_weights (ceil(C/32), 3, 2, 64, 4); false: the layer does not fit
bool launch_wn_gate0(const Gate0Args &a, hipStream_t stream) {
    const bool ok = a.pulse_channels >= 1 && a.pulse_channels + 2 <= 8 && a.channels % 4 == 0 && a.ldo % 4 == 0 &&
                    a.out_bstride % 4 == 0 && (uintptr_t)a.out % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.cond &&
                    (uintptr_t)a.cond % 8 == 0 && a.cond_bstride % 2 == 0 && a.cond_up >= 1 && a.lerp_w0 && a.lerp_w1 &&
                    a.dil >= 1 && a.dil <= G0_MAX_DIL && (G0_ROWS + a.cond_up - 2) / a.cond_up + 2 <= G0_COND_ROWS &&
                    (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && (!a.write_inputs || a.ldo >= a.channels + 16);
    if (!ok) return false;
    if (a.max_rows <= 0 || a.batch <= 0) return true;
    Gate0Args r = a;
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + G0_ROWS - 1) / G0_ROWS;
    const long long blocks = (long long)r.m_tiles_per_item * a.batch * r.n_tiles;
    hipLaunchKernelGGL(wn_gate0_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, r);
    return true;
}

}  // namespace mbx
