// conv1d (channels-last, stride 1, dilation d) as an implicit GEMM on the gfx950 fp32 matrix cores.
//
//   y[b, t, co] = bias[co] + sum_{j < ks} sum_{ci < cin} xp[b, t + j*d - pad_l, ci] * W[j, ci, co]
//
// restates Keras Conv1D as the reference drives it (weight-norm already folded into W):
//   TF2C_Conv1DWeightNorm.call      reference .../tf2_components/layers/conv_layers.py:149-165
//   TFPad1d (SYMMETRIC / EDGE)      reference .../custom_layers.py:47-71
//   WaveNetAE layer loop            reference .../custom_AE_layers.py:305-335  (gate / res-skip epilogues)
//
// The WaveNet's dilated convolutions are dense C -> 2C contractions (K = 3C = 960, N = 640 for C = 320),
// i.e. genuinely GEMM shaped, so they run on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains: the result
// honours the float32 parity budget; bf16/fp8 MFMA would not).  One kernel template covers every
// convolution of the path; what changes is the tile shape and the epilogue:
//
//   EPI_LINEAR   bias (+ PReLU / leaky)                       F0-net, VTF-net, cond conv, end, post-net
//   EPI_GATE     + conditioning (interpolated on the fly from the (2T, 2C) tensor), tanh * sigmoid
//   EPI_RESSKIP  h += r[:, :C], skip (+)= r[:, C:]            (in place; one owner lane per element)
//
// Tiling: 256 threads = 4 waves (one per SIMD), block tile BM x BN, wave tile (TM*32) x (TN*32),
// K streamed in BK = 16 slices through a double-buffered LDS image:
//   As[k][row]  (k-major so that the 32 lanes of an MFMA A-operand read 32 consecutive rows: conflict free)
//   Bs[k][col]
// The A slice is the dilated receptive-field window: rows t + j*d - pad_l of the activation, fetched with
// the padding rule of the layer and the item's own length (padded batches equal one-at-a-time runs).
#include <cstdlib>

#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

// source row of the padded input: -1 = zero sample.  Branch free (selects only) so that the K loop stays one
// scheduling region.  mode: 0 zero, 1 symmetric (edge sample repeated), 2 edge.
__device__ __forceinline__ int map_row(int s, int n, int mode) {
    const bool inside = (s >= 0) & (s < n);
    const int refl = min(max(s < 0 ? -s - 1 : 2 * n - s - 1, 0), n - 1);
    const int edge = min(max(s, 0), n - 1);
    const int outside = mode == 0 ? -1 : (mode == 1 ? refl : edge);
    return inside ? s : outside;
}

// tanh(zt) * sigmoid(zs) with two hardware exponentials and two reciprocals:
//   tanh(x) = 1 - 2 / (1 + e^{2x}),  sigmoid(x) = 1 / (1 + e^{-x})
// absolute error of the product <= ~3e-7 (v_exp_f32 / v_rcp_f32 are 1 ulp), well inside the stage budget of
// 2e-5; saturates correctly (e^{2x} = inf -> 1, 0 -> -1).
__device__ __forceinline__ float gate_act(float zt, float zs) {
    const float e2 = __expf(2.0f * zt);
    const float e1 = __expf(-zs);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
    return th * __builtin_amdgcn_rcpf(1.0f + e1);
}

// VEC: every row/column group of 4 floats is 16-byte aligned and all-or-nothing valid (cin, cout, C, ldx and
// the batch strides are multiples of 4): the slice loads are then unconditional float4 loads from a clamped
// address, zeroed by a select -- no branch in the K loop, so the prefetch of slice k+1 really overlaps the
// MFMAs of slice k.  The scalar path only serves the tiny odd-sized convolutions (cin = 6, 30, cout = 1, 15).
template <int WM, int WN, int TM, int TN, int EPI, bool VEC>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvArgs p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int LDA = BM + 4;
    constexpr int LDB = BN + 4;
    constexpr int A_TOT = BM * BK / 4;         // float4 per A slice
    constexpr int B_TOT = BN * BK / 4;
    constexpr int A_F4 = (A_TOT + 255) / 256;  // float4 per thread per A slice
    constexpr int B_F4 = (B_TOT + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(EPI != EPI_GATE || TN == 2, "gate needs the tanh and the sigmoid tile in one wave");

    __shared__ float lds[2 * BK * LDA + 2 * BK * LDB];
    float *As = lds;
    float *Bs = lds + 2 * BK * LDA;

    const int b = blockIdx.z;
    const int rows = p.n_frames ? p.n_frames[b] * p.rows_per_frame : p.max_rows;
    const int m0 = blockIdx.x * BM;
    if (m0 >= rows) return;
    const int C = p.channels;
    // column origin of this block in the weight matrix
    const int n0 = (EPI == EPI_GATE) ? blockIdx.y * (BN / 2) : blockIdx.y * BN;
    const int n_lim = (EPI == EPI_GATE) ? C : p.cout;   // valid columns per half / in total

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;

    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nkc = (p.cin + BK - 1) / BK;   // K slices per tap
    const int nk = p.ks * nkc;

    float4 ra[A_F4], rb[B_F4];
    unsigned okmask = 0;   // bit i: A float4 i valid, bit 16+i: B float4 i valid (masking is deferred to the LDS store)

    // per-thread addressing that does not change inside a tap / inside the kernel
    int a_off[A_F4];       // element offset of the source row of A float4 i for the current tap (clamped)
    unsigned a_rowok = 0;  // bit i: that source row is a real sample (not zero padding)
    int b_col[B_F4];       // clamped weight column of B float4 i
    unsigned b_colok = 0;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int q = tid + i * 256;
        const int c = (q % (BN / 4)) * 4;
        int n, lim;
        if (EPI == EPI_GATE) {
            const bool second = c >= BN / 2;
            n = n0 + (second ? c - BN / 2 + C : c);
            lim = n_lim + (second ? C : 0);
        } else {
            n = n0 + c;
            lim = n_lim;
        }
        b_col[i] = VEC ? min(n, p.cout - 4) : n;
        if (n < lim) b_colok |= 1u << i;
    }
    auto set_tap = [&](int tap) {
        a_rowok = 0;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = (tid + i * 256) >> 2;
            const int src = map_row(m0 + row - p.pad_l + tap * p.dil, rows, p.pad_mode);
            a_off[i] = max(src, 0) * p.ldx;
            if (src >= 0) a_rowok |= 1u << i;
        }
    };

    auto load_slice = [&](int kt) {
        const int tap = kt / nkc;
        const int ci0 = (kt - tap * nkc) * BK;
        if (ci0 == 0) set_tap(tap);
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int q = tid + i * 256;
            if (A_TOT % 256 != 0 && q >= A_TOT) break;
            const int ci = ci0 + (q & 3) * 4;
            const bool rowok = (a_rowok >> i) & 1u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (VEC) {
                v = *reinterpret_cast<const float4 *>(xb + a_off[i] + min(ci, p.cin - 4));
                const bool ok = rowok & (ci < p.cin);
                okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));
            } else if (rowok) {
                const float *px = xb + a_off[i] + ci;
                if (ci + 0 < p.cin) v.x = px[0];
                if (ci + 1 < p.cin) v.y = px[1];
                if (ci + 2 < p.cin) v.z = px[2];
                if (ci + 3 < p.cin) v.w = px[3];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int q = tid + i * 256;
            if (B_TOT % 256 != 0 && q >= B_TOT) break;
            const int ci = ci0 + q / (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (VEC) {
                v = *reinterpret_cast<const float4 *>(p.w + (long long)(tap * p.cin + min(ci, p.cin - 1)) * p.cout + b_col[i]);
                const bool ok = ((b_colok >> i) & 1u) & (ci < p.cin);
                okmask = ok ? (okmask | (1u << (16 + i))) : (okmask & ~(1u << (16 + i)));
            } else if (ci < p.cin) {
                const int c = (q % (BN / 4)) * 4;
                int lim;
                if (EPI == EPI_GATE) lim = n_lim + (c >= BN / 2 ? C : 0);
                else lim = n_lim;
                const int n = b_col[i];
                const float *pw = p.w + (long long)(tap * p.cin + ci) * p.cout + n;
                if (n + 0 < lim) v.x = pw[0];
                if (n + 1 < lim) v.y = pw[1];
                if (n + 2 < lim) v.z = pw[2];
                if (n + 3 < lim) v.w = pw[3];
            }
            rb[i] = v;
        }
    };

    auto store_slice = [&](int buf) {
        float *a = As + buf * BK * LDA;
        float *bs = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int q = tid + i * 256;
            const int row = q >> 2, kq = q & 3;
            if (A_TOT % 256 != 0 && q >= A_TOT) break;
            const bool ok = !VEC || ((okmask >> i) & 1u);
            a[(kq * 4 + 0) * LDA + row] = ok ? ra[i].x : 0.f;
            a[(kq * 4 + 1) * LDA + row] = ok ? ra[i].y : 0.f;
            a[(kq * 4 + 2) * LDA + row] = ok ? ra[i].z : 0.f;
            a[(kq * 4 + 3) * LDA + row] = ok ? ra[i].w : 0.f;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int q = tid + i * 256;
            const int k = q / (BN / 4), c = (q % (BN / 4)) * 4;
            if (B_TOT % 256 != 0 && q >= B_TOT) break;
            const bool ok = !VEC || ((okmask >> (16 + i)) & 1u);
            float4 v = rb[i];
            v.x = ok ? v.x : 0.f;
            v.y = ok ? v.y : 0.f;
            v.z = ok ? v.z : 0.f;
            v.w = ok ? v.w : 0.f;
            *reinterpret_cast<float4 *>(bs + k * LDB + c) = v;
        }
    };

    // column of the wave's tile tn inside the block tile
    auto col_base = [&](int tn) { return (EPI == EPI_GATE) ? tn * (BN / 2) + wc * 32 : (wc * TN + tn) * 32; };

    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int ecol = lane & 31;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    load_slice(0);
    store_slice(0);
    __syncthreads();

    const int lrow = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slice(kt + 1);
        const float *a = As + buf * BK * LDA + lk * LDA + wr * TM * 32 + lrow;
        const float *bs = Bs + buf * BK * LDB + lk * LDB + lrow;
        // all operands of the slice are requested first (counted lgkmcnt waits then release the MFMAs in order)
        float av[BK / 2][TM], bv[BK / 2][TN];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[kk / 2][i] = a[kk * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[kk / 2][j] = bs[kk * LDB + col_base(j)];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk / 2][i], bv[kk / 2][j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_slice(buf ^ 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    if (EPI == EPI_GATE) {
        const int ch = n0 + wc * 32 + ecol;   // gate channel of this lane
        if (ch < C) {
            const float bt = p.bias ? p.bias[ch] : 0.f;
            const float bsg = p.bias ? p.bias[C + ch] : 0.f;
            const float *cb = p.cond + (long long)b * p.cond_bstride + ch;
            const int n2 = rows / p.cond_up;
            float *ob = p.out + (long long)b * p.out_bstride + ch;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < rows) {
                        // conditioning interpolated on the fly from the (rows/U, 2C) tensor
                        const int t2 = row / p.cond_up, u = row - t2 * p.cond_up;
                        const int t3 = min(t2 + 1, n2 - 1);
                        const float w0 = p.lerp_w0[u], w1 = p.lerp_w1[u];
                        const float *c0 = cb + t2 * (2 * C);
                        const float *c1 = cb + t3 * (2 * C);
                        const float zt = (acc[i][0][r] + bt) + (c0[0] * w0 + c1[0] * w1);
                        const float zs = (acc[i][1][r] + bsg) + (c0[C] * w0 + c1[C] * w1);
                        ob[row * p.ldo] = gate_act(zt, zs);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + col_base(j) + ecol;
            if (col >= p.cout) continue;
            const float bias = p.bias ? p.bias[col] : 0.f;
            if (EPI == EPI_LINEAR) {
                const float slope = p.alpha ? p.alpha[col] : p.leaky;
                const bool act = p.alpha != nullptr || p.use_leaky;
                float *ob = p.out + (long long)b * p.out_bstride;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (row < rows) {
                            float v = acc[i][j][r] + bias;
                            if (act) v = v > 0.f ? v : slope * v;
                            ob[(long long)row * p.ldo + col] = v;
                        }
                    }
            } else {   // EPI_RESSKIP
                const bool to_h = (!p.last_layer) && col < C;
                const int oc = to_h ? col : (p.last_layer ? col : col - C);
                float *dst = (to_h ? p.h : p.skip) + (long long)b * p.hs_bstride;
                const bool accumulate = to_h || !p.skip_init;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (row < rows) {
                            const float v = acc[i][j][r] + bias;
                            float *q = dst + (long long)row * C + oc;
                            *q = accumulate ? (*q + v) : v;
                        }
                    }
            }
        }
    }
}

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int WM, int WN, int TM, int TN, int EPI>
static void launch_cfg(const ConvArgs &a, hipStream_t stream, int extra_lds = 0) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ncols = (EPI == EPI_GATE) ? a.channels : a.cout;
    const int bn_eff = (EPI == EPI_GATE) ? BN / 2 : BN;
    dim3 grid((a.max_rows + BM - 1) / BM, (ncols + bn_eff - 1) / bn_eff, a.batch);
    const bool vec = (a.cin % 4 == 0) && (a.ldx % 4 == 0) && (a.x_bstride % 4 == 0) && (a.cout % 4 == 0) &&
                     (EPI != EPI_GATE || a.channels % 4 == 0) && ((uintptr_t)a.x % 16 == 0) &&
                     ((uintptr_t)a.w % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, TM, TN, EPI, true>), grid, dim3(256), extra_lds, stream, a);
    else
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, TM, TN, EPI, false>), grid, dim3(256), extra_lds, stream, a);
}

void launch_conv1d(const ConvArgs &a, int epilogue, hipStream_t stream) {
    if (a.max_rows <= 0 || a.batch <= 0) return;
    // tuning knobs (experiments only): tile shape of the two WaveNet GEMMs and extra dynamic LDS to cap blocks/CU
    static const int gate_cfg = env_int("MBX_GATE_CFG", 1);
    static const int extra_lds = env_int("MBX_EXTRA_LDS", 0);
    if (epilogue == EPI_GATE) {
        if (gate_cfg == 1) launch_cfg<2, 2, 1, 2, EPI_GATE>(a, stream, extra_lds);       // 64 rows x 64 gate channels
        else if (gate_cfg == 2) launch_cfg<4, 1, 1, 2, EPI_GATE>(a, stream, extra_lds);  // 128 rows x 32 gate channels
        else launch_cfg<2, 2, 2, 2, EPI_GATE>(a, stream, extra_lds);                     // 128 rows x 64 gate channels
    } else if (epilogue == EPI_RESSKIP) {
        if (gate_cfg == 1) launch_cfg<2, 2, 1, 2, EPI_RESSKIP>(a, stream, extra_lds);    // 64 x 128
        else if (gate_cfg == 2) launch_cfg<4, 1, 1, 2, EPI_RESSKIP>(a, stream, extra_lds);  // 128 x 64
        else launch_cfg<2, 2, 2, 2, EPI_RESSKIP>(a, stream, extra_lds);                  // 128 x 128
    } else if (a.cout <= 32) {
        launch_cfg<4, 1, 1, 1, EPI_LINEAR>(a, stream);        // 128 x 32 (F0 head, post-net, end)
    } else if ((long long)a.max_rows * a.batch >= 4096 && a.cout >= 128) {
        launch_cfg<2, 2, 2, 2, EPI_LINEAR>(a, stream);        // 128 x 128
    } else {
        launch_cfg<2, 2, 1, 1, EPI_LINEAR>(a, stream);        // 64 x 64 (mel-rate sub-nets)
    }
}

}  // namespace mbx
