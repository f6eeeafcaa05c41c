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
// Block = 4 waves, 256 rows x 32 gate channels; wave w owns rows 64 w .. 64 w + 63 as four 16-row tiles x 64 weight
// columns ([16 tanh | 16 sigmoid] of the even and of the odd gate channels).  The matrix instruction
// (v_mfma_f32_16x16x4_f32) takes the weights as its row operand and x' as its column operand, so a lane ends up with one
// row of the tile and eight consecutive gate channels (8 kq .. 8 kq + 7, tanh and sigmoid parts of each): the row's
// interpolation weights are per-lane constants, the conditioning values come as 16-byte LDS reads and the result leaves
// as two 16-byte stores (128 contiguous bytes per row over the four lanes of a row).  MFMA step (tau, m) contracts
// channels {2 kq + m} of tap tau.  The weights of a column tile are 24 registers per lane (loaded once per wave).  The
// 24 MFMAs of tile i + 1 are issued between the activation arithmetic of tile i, three per output channel (two
// accumulator sets).  Everything the block needs from global memory is requested up front (one round trip instead of a
// chain of them), then staged in LDS: the rows x'[m0 - d, m0 + 256 + d) (32 bytes each, 16-byte chunk c at 2*row +
// (c ^ ((row>>3)&1)): bank-conflict free 8-byte operand reads), the conditioning rows of the block (<= 32 x (32 tanh |
// 32 sigmoid), row pitch 68 floats) and the per-row interpolation tables.
//
// What bounds it (16 x 10 s, round 3, ablations scripts/experiments/mkexp.py g0n_*): 158 us = prologue 49 + arithmetic of
// the four tiles 81 + stores 29; the arithmetic is the vector pipe (15 vector + 3 transcendental instructions per output
// and lane: ~100 issue cycles against 96 cycles of MFMA), not the matrix cores and not HBM (436 MB of output = 2.8 TB/s).
// Looping a block over several column tiles with the next tile's operands in flight (one prologue per 5 or 10 tiles)
// gave the same 159 us at 212 registers: the prologue already overlaps with other blocks' arithmetic.
#include <cstdlib>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G0_ROWS = 256;

constexpr int G0_MAX_DIL = 16;
constexpr int G0_XROWS = G0_ROWS + 2 * G0_MAX_DIL;      // staged rows of x'
constexpr int G0_COND_ROWS = 32;
constexpr int G0_CS = 68;                               // floats between conditioning rows in LDS (16-byte reads of rows t2, t2 + 1)

template <int KIND>
__global__ __launch_bounds__(256) void wn_gate0_kernel(Gate0Args p) {
    __shared__ __attribute__((aligned(16))) float xs[G0_XROWS * 8];        // x' rows m0 - d .. (staged row i = m0 - d + i)
    __shared__ __attribute__((aligned(16))) float cs[G0_COND_ROWS * G0_CS];   // conditioning rows t2base ..
    __shared__ float2 tabw[G0_ROWS];                                        // (w0, w1) of block row lr
    __shared__ int tabo[G0_ROWS];                                           // float offset of its conditioning row in cs
    const int id = blockIdx.x;
    const int nt = id % p.n_tiles;
    const int g = id / p.n_tiles;
    const int b = g / p.m_tiles_per_item;
    const int mt = g - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * G0_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int d = p.dil;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const float *pb = p.pulse + (long long)b * p.pulse_bstride;
    const float *nb = p.noise ? p.noise + (long long)b * p.noise_bstride : nullptr;
    const int pc = p.pulse_channels;
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    const int n2 = rows / cond_up;
    const float *cbase = p.cond + (long long)b * p.cond_bstride;

    // conditioning rows of the block for column tile nt (clamped to the item): thread -> 16-byte positions tid, tid + 256
    auto cond_request = [&](int nt, float4 (&cv)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pos = tid + 256 * it;
            const int crow = pos >> 4, cq = pos & 15;
            const int chn = nt * 32 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            cv[it] = *reinterpret_cast<const float4 *>(cbase + (long long)t * (2 * C) + (cq >> 3) * C + (chn < C ? chn : 0));
            if (chn >= C) cv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto cond_store = [&](const float4 (&cv)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pos = tid + 256 * it;
            *reinterpret_cast<float4 *>(cs + (pos >> 4) * G0_CS + 4 * (pos & 15)) = cv[it];
        }
    };
    // weights of a column tile: [tap][parity e][lane][tanh m0, tanh m1, sigmoid m0, sigmoid m1]
    auto weight_request = [&](int nt, float4 (&w)[3][2]) {
        const float *wt = p.w + (long long)nt * 1536 + lane * 4;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 2; ++e) w[t][e] = *reinterpret_cast<const float4 *>(wt + (t * 2 + e) * 256);
    };

    // ---- everything the block needs first is requested before any of it is used (one round trip, not a chain of them)
    // x' (channels: pulse channels | sigma * noise | 1 | 0; rows outside the item are zero): thread -> staged rows tid, tid + 256
    float xp[2][6], xn[2];
    bool xok[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + 256 * it;
        const int srow = m0 - d + i;
        xok[it] = i < G0_ROWS + 2 * d && srow >= 0 && srow < rows;
        const long long sr = xok[it] ? srow : 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) xp[it][k] = pb[sr * pc + min(k, pc - 1)];
        xn[it] = nb ? nb[sr] : 0.f;
    }
    float4 cv[2], bw[3][2];
    cond_request(nt, cv);
    weight_request(nt, bw);
    // lane (row r16, kq): gate channels ch0 + j, j = 2 v + e <-> accumulator [2 e + (0 tanh | 1 sigmoid)][v]
    const int ch0 = nt * 32 + 8 * kq;
    f32x4 bias[4];
    {
        float4 bv[2][2];                                  // [tanh | sigmoid][channels ch0 .. + 3 | ch0 + 4 .. + 7]
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = p.bias && ch0 + 4 * h + 3 < C;
                bv[g2][h] = *reinterpret_cast<const float4 *>(ok ? p.bias + g2 * C + ch0 + 4 * h : p.w);
                if (!ok) bv[g2][h] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int j = 2 * v + (c >> 1);
                bias[c][v] = (&bv[c & 1][j >> 2].x)[j & 3];
            }
    }
    {
        const int row = m0 + tid;
        const int t2 = row / cond_up;
        const int u = row - t2 * cond_up;
        tabw[tid] = make_float2(p.lerp_w0[u], p.lerp_w1[u]);
        tabo[tid] = (t2 - t2base) * G0_CS;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + 256 * it;
        if (i < G0_ROWS + 2 * d) {
            // channel k = pulse k | sigma * noise (k == pc) | 1 (k == pc + 1) | 0, put together from bit masks (the
            // conditions are the same for every lane: selects, not branches)
            float xr[8];
            const unsigned keep = xok[it] ? 0xffffffffu : 0u;
            const unsigned nbits = __float_as_uint(p.sigma * xn[it]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned mp = k < pc ? 0xffffffffu : 0u, mn = k == pc ? 0xffffffffu : 0u, one = k == pc + 1 ? 0x3f800000u : 0u;
                xr[k] = __uint_as_float(((__float_as_uint(xp[it][k < 6 ? k : 5]) & mp) | (nbits & mn) | one) & keep);
            }
            const int sw = (i >> 3) & 1;
            *reinterpret_cast<float4 *>(xs + 8 * i + 4 * sw) = make_float4(xr[0], xr[1], xr[2], xr[3]);
            *reinterpret_cast<float4 *>(xs + 8 * i + 4 * (sw ^ 1)) = make_float4(xr[4], xr[5], xr[6], xr[7]);
        }
    }
    cond_store(cv);
    float *ob = p.out + (long long)b * p.out_bstride;
    __syncthreads();

    {
        // matrix work of a tile in 24 steps: operands of the tile, then step k = 8 t + 4 m + c (tap t, channel 2 kq + m,
        // accumulator c)
        auto matrix_begin = [&](int i, f32x4 (&acc)[4], float2 (&xv)[3]) {
            const int l0 = 64 * wave + 16 * i;           // first row of the tile, relative to m0
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int si = l0 + r16 + t * d;         // staged row of tap t: m0 + l0 + r16 + (t - 1) d
                xv[t] = *reinterpret_cast<const float2 *>(xs + 8 * si + 4 * ((kq >> 1) ^ ((si >> 3) & 1)) + 2 * (kq & 1));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = bias[c];
        };
        auto matrix_step = [&](int k, f32x4 (&acc)[4], const float2 (&xv)[3]) {
            const int t = k >> 3, m = (k >> 2) & 1, c = k & 3;
            const float4 wv = bw[t][c >> 1];
            const float wa = (c & 1) ? (m ? wv.w : wv.z) : (m ? wv.y : wv.x);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, m ? xv[t].y : xv[t].x, acc[c], 0, 0, 0);
        };
        // conditioning operands of a tile: register v of accumulator c = weight column 4 kq + v of its 16-column part,
        // tile row r16 -> the lane's channels ch0 + j
        struct CondRegs { float2 w; float4 t0[2], t1[2], s0[2], s1[2]; };
        auto cond_load = [&](int i, CondRegs &cr) {
            const int lr = 64 * wave + 16 * i + r16;
            cr.w = tabw[lr];
            const float *c0 = cs + tabo[lr] + 8 * kq;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cr.t0[h] = *reinterpret_cast<const float4 *>(c0 + 4 * h);
                cr.t1[h] = *reinterpret_cast<const float4 *>(c0 + G0_CS + 4 * h);
                cr.s0[h] = *reinterpret_cast<const float4 *>(c0 + 32 + 4 * h);
                cr.s1[h] = *reinterpret_cast<const float4 *>(c0 + G0_CS + 32 + 4 * h);
            }
        };
        auto activate = [&](int j, const f32x4 (&acc)[4], const CondRegs &cr) {
            const int h = j >> 2, q = j & 3, v = j >> 1, e = j & 1;
            const float t0 = (&cr.t0[h].x)[q], t1 = (&cr.t1[h].x)[q], s0 = (&cr.s0[h].x)[q], s1 = (&cr.s1[h].x)[q];
            float r = wn_gate_act(KIND, acc[2 * e][v] + fmaf(t1, cr.w.y, t0 * cr.w.x), acc[2 * e + 1][v] + fmaf(s1, cr.w.y, s0 * cr.w.x));
            // the value exists here (not inside the bounds checks of the stores, where the compiler would sink the arithmetic)
            asm volatile("" : "+v"(r));
            return r;
        };
        auto store_tile = [&](int i, const float (&o)[8]) {
            const int l0 = 64 * wave + 16 * i;
            const int row = m0 + l0 + r16;
            if (row < rows) {
                float *orow = ob + (long long)row * p.ldo + ch0;
                if (ch0 + 3 < C) *reinterpret_cast<float4 *>(orow) = make_float4(o[0], o[1], o[2], o[3]);
                if (ch0 + 7 < C) *reinterpret_cast<float4 *>(orow + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
            // x' (8 channels, padded to 16) behind the C gate channels of the rows of this tile: lane -> row lane / 4, 4 floats
            if (p.write_inputs && nt == 0) {
                const int lx = l0 + (lane >> 2), q = lane & 3;
                const int xrow = m0 + lx, si = lx + d;
                if (xrow < rows) {
                    float4 ox = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (q < 2) ox = *reinterpret_cast<const float4 *>(xs + 8 * si + 4 * (q ^ ((si >> 3) & 1)));
                    *reinterpret_cast<float4 *>(ob + (long long)xrow * p.ldo + C + 4 * q) = ox;
                }
            }
        };
        // tile i + 1's matrix work is issued between tile i's activation arithmetic, three MFMAs per output channel:
        // the matrix pipe and the vector / transcendental pipes of the SIMD work side by side inside one wave
        f32x4 acc[2][4];
        float2 xv[3];
        CondRegs cr;
        float o[8];
        matrix_begin(0, acc[0], xv);
#pragma unroll
        for (int k = 0; k < 24; ++k) matrix_step(k, acc[0], xv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cond_load(i, cr);
            if (i + 1 < 4) matrix_begin(i + 1, acc[(i + 1) & 1], xv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (i + 1 < 4) {
#pragma unroll
                    for (int k = 3 * j; k < 3 * j + 3; ++k) matrix_step(k, acc[(i + 1) & 1], xv);
                }
                o[j] = activate(j, acc[i & 1], cr);
                __builtin_amdgcn_sched_barrier(0);
            }
            store_tile(i, o);
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
                    (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && (!a.write_inputs || a.ldo >= a.channels + 16) &&
                    (uintptr_t)a.bias % 16 == 0;
    if (!ok) return false;
    if (a.max_rows <= 0 || a.batch <= 0) return true;
    Gate0Args r = a;
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + G0_ROWS - 1) / G0_ROWS;
    const long long blocks = (long long)r.m_tiles_per_item * a.batch * r.n_tiles;
    if (a.gate_act == 0) hipLaunchKernelGGL(wn_gate0_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, r);
    else if (a.gate_act == 1) hipLaunchKernelGGL(wn_gate0_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, r);
    else if (a.gate_act == 2) hipLaunchKernelGGL(wn_gate0_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, r);
    else hipLaunchKernelGGL(wn_gate0_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, r);
    return true;
}

}  // namespace mbx
