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
#include <type_traits>

#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// raise the wave priority while it issues the MFMA cluster of a slice (+1 % at large batch, neutral at batch 1)
#ifndef MBX_SETPRIO
#define MBX_SETPRIO 1
#endif


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
__device__ __forceinline__ float gate_act(int kind, float zt, float zs) {
    if (kind != 0) return wn_gate_act(kind, zt, zs);          // gfu / gsu (mbx_kernels.h)
    const float e2 = __expf(2.0f * zt);
    const float e1 = __expf(-zs);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
    return th * __builtin_amdgcn_rcpf(1.0f + e1);
}

// Block -> (item, row tile, column tile).  With p.remap the grid is 1-D and XCD aware: workgroups are dealt
// round-robin over the 8 XCDs (ids b and b+8 share an L2), so XCD x takes the row tiles x, x+8, ... and walks
// all column tiles of a row tile back to back -- the blocks that share an activation tile (and its dilation
// halo) run on one L2 at about the same time.  Placement only affects speed, never results.
__device__ __forceinline__ bool decode_tile(const ConvArgs &p, int &b, int &mt, int &nt) {
    if (!p.remap) {
        b = blockIdx.z;
        mt = blockIdx.x;
        nt = blockIdx.y;
        return true;
    }
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g = (l / p.n_tiles) * 8 + (id & 7);
    nt = l % p.n_tiles;
    if (g >= p.m_tiles_total) return false;
    b = g / p.m_tiles_per_item;
    mt = g - b * p.m_tiles_per_item;
    return true;
}

// Epilogue shared by the register-staged and the LDS-DMA kernels.
// C/D layout of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <int WM, int WN, int TM, int TN, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &p, f32x16 (&acc)[TM][TN], int b, int rows, int m0, int n0,
                                              int wr, int wc, int lane) {
    constexpr int BN = WN * TN * 32;
    const int C = p.channels;
    const int ecol = lane & 31;
    auto col_base = [&](int tn) { return (EPI == EPI_GATE) ? tn * (BN / 2) + wc * 32 : (wc * TN + tn) * 32; };
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
                        ob[(long long)row * p.ldo] = gate_act(p.gate_act, zt, zs);
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
                            float *q = dst + (long long)row * C + oc;
                            if (p.acc_preloaded) {          // old value and bias already sit in the accumulator
                                *q = acc[i][j][r];
                            } else {
                                const float v = acc[i][j][r] + bias;
                                *q = accumulate ? (*q + v) : v;
                            }
                        }
                    }
            }
        }
    }
}

// VEC: every row/column group of 4 floats is 16-byte aligned and all-or-nothing valid (cin, cout, C, ldx and
// the batch strides are multiples of 4): the slice loads are then unconditional float4 loads from a clamped
// address, zeroed by a select -- no branch in the K loop, so the prefetch of slice k+1 really overlaps the
// MFMAs of slice k.  The scalar path only serves the tiny odd-sized convolutions (cin = 6, 30, cout = 1, 15).
template <int WM, int WN, int TM, int TN, int EPI, bool VEC, int BK>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvArgs p) {
    constexpr int KQ = BK / 4;                 // float4 per row of an A slice
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

    int b, mt_, nt_;
    if (!decode_tile(p, b, mt_, nt_)) return;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt_ * BM;
    if (m0 >= rows) return;
    const int C = p.channels;
    // column origin of this block in the weight matrix
    const int n0 = (EPI == EPI_GATE) ? nt_ * (BN / 2) : nt_ * BN;
    const int n_lim = (EPI == EPI_GATE) ? C : p.cout;   // valid columns per half / in total

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;

    const float *xb = p.x + (long long)b * p.x_bstride + (long long)m0 * p.ldx;   // the block's first row: int row offsets stay small for any item length
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
            const int row = (tid + i * 256) / KQ;
            const int src = map_row(m0 + row - p.pad_l + tap * p.dil, rows, p.pad_mode);
            a_off[i] = (max(src, 0) - m0) * p.ldx;
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
            const int ci = ci0 + (q % KQ) * 4;
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
            const int row = q / KQ, kq = q % KQ;
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
    if (EPI == EPI_RESSKIP && p.acc_preloaded) {
        // res/skip: the accumulators start from bias + the value they will be added to (h or skip); the reads
        // overlap the first K slices instead of stalling the epilogue, which is then a plain store
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + col_base(j) + ecol;
            if (col < p.cout) {
                const float bias = p.bias ? p.bias[col] : 0.f;
                const bool to_h = (!p.last_layer) && col < C;
                const int oc = to_h ? col : (p.last_layer ? col : col - C);
                const float *src = (to_h ? p.h : p.skip) + (long long)b * p.hs_bstride + oc;
                const bool accumulate = to_h || !p.skip_init;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const float old = (accumulate && row < rows) ? src[(long long)row * C] : 0.f;
                        acc[i][j][r] = old + bias;
                    }
            }
        }
    }
    load_slice(0);
    store_slice(0);
    __syncthreads();

    const int lrow = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slice(kt + 1);
        const float *a = As + buf * BK * LDA + lk * LDA + wr * TM * 32 + lrow;
        const float *bs = Bs + buf * BK * LDB + lk * LDB + lrow;
        // software pipeline over the k-steps of the slice: the LDS reads of k-step kk+2 are issued before the
        // MFMAs of k-step kk (sched_group_barrier pins that order), so a wave never sits in an LDS round trip
        // with an idle matrix pipe
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = a[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = bs[col_base(j)];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[nxt][i] = a[(kk + 2) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[nxt][j] = bs[(kk + 2) * LDB + col_base(j)];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            if (kk + 2 < BK) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // DS reads of k-step kk+2
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                    // MFMAs of k-step kk
        }
        if (kt + 1 < nk) store_slice(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<WM, WN, TM, TN, EPI>(p, acc, b, rows, m0, n0, wr, wc, lane);
}

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes, global (per-lane address) -> LDS (M0 = wave-uniform byte
// address, lane l lands at M0 + 16*l).  Issued through inline asm on purpose: with the builtin hipcc waits
// vmcnt(0) in front of the next ds_read of the same __shared__ array (it cannot tell the two LDS buffers
// apart), which would serialise the prefetch of slice k+1 with the MFMAs of slice k.  The kernel orders DMA and
// reads itself: s_waitcnt vmcnt(0) + barrier before a buffer is read, barrier before it is overwritten.
__device__ __forceinline__ void lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant for the two WaveNet GEMMs (needs the VEC conditions and p.zeros).
// The K slices go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write, the K loop
// spends its registers on accumulators only).  A wave-instruction writes 64 x 16 B contiguously, the SOURCE
// address is per lane, so the LDS image is linear and the layout is chosen through the source addresses:
//   A slice: BM rows x 4 chunks of 4 channels, chunk (row, c) stored at position 4*row + (c ^ ((row >> 2) & 3)).
//            An MFMA lane (row, h) fetches its chunks c = 2*cc + h with ds_read_b128; the XOR spreads the 16
//            rows of a ds_read_b128 lane group over all 16 bank quads (conflict free).  K order inside the slice
//            is thereby permuted (k = 8*cc + 4*h + s for step (cc, s)); the B operand uses the same order.
//   B slice: 16 k-rows x BN columns, k-major (ds_read_b32, consecutive lanes = consecutive columns).
// Zero padding (rows outside the item, channels/columns beyond the tensor) = lanes pointed at a 16-byte zero buffer.
template <int WM, int WN, int TM, int TN, int EPI>
__global__ __launch_bounds__(256) void conv1d_mfma_dma_kernel(ConvArgs p) {
    constexpr int BK = 16;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int A_INST = BM * 4 / 64 / 4;    // LDS-DMA wave-instructions per wave and slice (A)
    constexpr int B_INST = BK * BN / 4 / 64 / 4;
    static_assert(WM * WN == 4 && A_INST >= 1 && B_INST >= 1, "tile shape");
    static_assert(EPI != EPI_GATE || TN == 2, "gate needs the tanh and the sigmoid tile in one wave");
    typedef __attribute__((address_space(3))) float lds_float;

    constexpr int NBUF = 3;                      // slices in flight: one being read, two landing
    __shared__ __attribute__((aligned(16))) float lds[NBUF * (BM * BK + BK * BN)];   // ONE array (A ring, B ring)

    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);
    int b, mt_, nt_;
    if (!decode_tile(p, b, mt_, nt_)) return;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt_ * BM;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = (EPI == EPI_GATE) ? nt_ * (BN / 2) : nt_ * BN;
    const int n_lim = (EPI == EPI_GATE) ? C : p.cout;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)m0 * p.ldx;   // the block's first row: int row offsets stay small for any item length
    const int nkc = (p.cin + BK - 1) / BK;
    const int nk = p.ks * nkc;

    // ---- per-lane DMA sources
    int a_row[A_INST], a_ch[A_INST], a_off[A_INST];
    unsigned a_rowok = 0;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int pos = (wave + 4 * i) * 64 + lane;           // chunk position in the A image
        const int row = pos >> 2;
        a_row[i] = row;
        a_ch[i] = 4 * ((pos & 3) ^ ((row >> 2) & 3));         // first channel (inside the slice) of the chunk stored here
    }
    int b_k[B_INST], b_col[B_INST];
    unsigned b_colok = 0;
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const int pos = (wave + 4 * i) * 64 + lane;
        const int c = (pos % (BN / 4)) * 4;
        b_k[i] = pos / (BN / 4);
        int n, lim;
        if (EPI == EPI_GATE) {
            const bool second = c >= BN / 2;
            n = n0 + (second ? c - BN / 2 + C : c);
            lim = n_lim + (second ? C : 0);
        } else {
            n = n0 + c;
            lim = n_lim;
        }
        b_col[i] = min(n, p.cout - 4);
        if (n < lim) b_colok |= 1u << i;
    }
    auto set_tap = [&](int tap) {
        a_rowok = 0;
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            const int src = map_row(m0 + a_row[i] - p.pad_l + tap * p.dil, rows, p.pad_mode);
            a_off[i] = (max(src, 0) - m0) * p.ldx;
            if (src >= 0) a_rowok |= 1u << i;
        }
    };
    auto issue = [&](int kt, int buf) {
        const int tap = kt / nkc;
        const int ci0 = (kt - tap * nkc) * BK;
        if (ci0 == 0) set_tap(tap);
        // LDS byte addresses of the two destination images (wave uniform)
        const unsigned adst = lds_base + 4u * (unsigned)(buf * (BM * BK));
        const unsigned bdst = lds_base + 4u * (unsigned)(NBUF * (BM * BK) + buf * (BK * BN));
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            const int ci = ci0 + a_ch[i];
            const bool ok = ((a_rowok >> i) & 1u) & (ci < p.cin);
            const float *src = ok ? xb + a_off[i] + ci : p.zeros;
            lds_dma16(src, adst + 1024u * (unsigned)(wave + 4 * i));
        }
#pragma unroll
        for (int i = 0; i < B_INST; ++i) {
            const int ci = ci0 + b_k[i];
            const bool ok = ((b_colok >> i) & 1u) & (ci < p.cin);
            const float *src = ok ? p.w + (long long)(tap * p.cin + ci) * p.cout + b_col[i] : p.zeros;
            lds_dma16(src, bdst + 1024u * (unsigned)(wave + 4 * i));
        }
    };

    auto col_base = [&](int tn) { return (EPI == EPI_GATE) ? tn * (BN / 2) + wc * 32 : (wc * TN + tn) * 32; };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ring of NBUF slices: iteration kt waits for slice kt (all but the newest slice's DMA instructions retired),
    // a barrier makes it visible block-wide and proves slice kt-1 is no longer read, then slice kt+2 is issued
    // into the buffer slice kt-1 occupied.  One barrier per slice; two slices of prefetch distance.
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    const int lrow = lane & 31, lk = lane >> 5;
    const int swz = (lrow >> 2) & 3;
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INST + B_INST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1);
        const float *ab = lds + buf * (BM * BK) + (wr * TM * 32 + lrow) * BK;
        const float *bb = lds + NBUF * (BM * BK) + buf * (BK * BN) + lrow;
        // k order of the slice: step (cc, st) multiplies channel 8*cc + 4*h + st (h = lane half); the operands of
        // step n+1 are requested before the MFMAs of step n issue (sched_group_barrier pins the interleave)
        float4 a4[2][TM];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a4[cc][i] = *reinterpret_cast<const float4 *>(ab + i * 32 * BK + 4 * ((2 * cc + lk) ^ swz));
        float bv[2][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = bb[(4 * lk) * BN + col_base(j)];
        if (MBX_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int cc = n >> 2, st = n & 3, cur = n & 1, nxt = cur ^ 1;
            if (n + 1 < 8) {
                const int c2 = (n + 1) >> 2, s2 = (n + 1) & 3;
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[nxt][j] = bb[(8 * c2 + 4 * lk + s2) * BN + col_base(j)];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = st == 0 ? a4[cc][i].x : st == 1 ? a4[cc][i].y : st == 2 ? a4[cc][i].z : a4[cc][i].w;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[cur][j], acc[i][j], 0, 0, 0);
            }
            if (n + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
        if (MBX_SETPRIO) __builtin_amdgcn_s_setprio(0);
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    conv_epilogue<WM, WN, TM, TN, EPI>(p, acc, b, rows, m0, n0, wr, wc, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Small-M convolution (mel-rate sub-nets at batch 1: a few hundred rows, K = ks*cin up to 768).
// A 32x32 output tile per block keeps >= 200 blocks in flight for 800 rows; what then bounds a block is the serial
// chain of K/2 dependent-rate MFMAs, so K is split over the 4 waves of the block (each wave runs a quarter of
// the chain) and the four partial tiles are summed through LDS.  Operands go straight from global memory (L2) to
// the MFMA registers: the tile is too small to amortise an LDS stage.  Lane (row r, half h) loads A[r][8g+4h..+3]
// as one float4 and feeds it to four MFMA steps; the B lane (col c, half h) loads W[8g+4h+s][c] for the same steps.
// Needs cin % 8 == 0 and 16-byte aligned rows (checked by the launcher).
// RT x CT output tiles of 32 x 32 per block (1 x 1 is what runs; larger register tiles were measured slower than the
// LDS-staged conv1d_mel_tile below, which takes the large launches).  Every output element is
// summed in the same order whatever the tiling: four K quarters, each one sequential MFMA chain, combined as
// ((q0 + q1) + q2) + q3 + bias -- so a padded batch stays bit-identical to one-at-a-time runs (the F0 contour feeds the
// phase accumulator: rounding there is audible in the last bits everywhere downstream).
template <int RT, int CT>
__device__ __forceinline__ void conv1d_small_tile(const ConvArgs &p, int bx, int by, int b, float *red) {
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = bx * 32 * RT;
    if (m0 >= rows) return;
    const int n0 = by * 32 * CT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int groups_per_tap = p.cin >> 3;                 // groups of 8 input channels
    const int n_groups = p.ks * groups_per_tap;
    const int g_begin = (n_groups * wave) / 4, g_end = (n_groups * (wave + 1)) / 4;
    bool col_ok[CT];
    const float *wcol[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = n0 + 32 * ct + lrow;
        col_ok[ct] = col < p.cout;
        wcol[ct] = p.w + min(col, p.cout - 1);
    }

    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;

    // Requests only (clamped addresses); the masks are applied by mask_group when the batch is consumed.  (Round 4, read off
    // the ISA: with `ok ? t : 0` right behind each load the compiler waited for every group's five loads before it issued the
    // next group's -- six serial round trips per batch of six groups "in flight".)
    auto load_group = [&](int g, float4 (&av)[RT], float (&bv)[CT][4]) {
        const int tap = g / groups_per_tap;
        const int ci = (g - tap * groups_per_tap) * 8 + 4 * lk;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int src = map_row(m0 + 32 * rt + lrow - p.pad_l + tap * p.dil, rows, p.pad_mode);
            av[rt] = *reinterpret_cast<const float4 *>(xb + (long long)max(src, 0) * p.ldx + ci);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float *wk = wcol[ct] + (long long)(tap * p.cin + ci) * p.cout;
#pragma unroll
            for (int st = 0; st < 4; ++st) bv[ct][st] = wk[(long long)st * p.cout];
        }
    };
    auto mask_group = [&](int g, float4 (&av)[RT], float (&bv)[CT][4]) {
        const int tap = g / groups_per_tap;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bool ok = map_row(m0 + 32 * rt + lrow - p.pad_l + tap * p.dil, rows, p.pad_mode) >= 0;
            av[rt].x = ok ? av[rt].x : 0.f;
            av[rt].y = ok ? av[rt].y : 0.f;
            av[rt].z = ok ? av[rt].z : 0.f;
            av[rt].w = ok ? av[rt].w : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int st = 0; st < 4; ++st) bv[ct][st] = col_ok[ct] ? bv[ct][st] : 0.f;
    };

    // the loads of a batch of DEPTH groups are all in flight while the previous batch feeds the matrix pipe
    // (one group = 4 MFMAs per tile = 0.1 us, an L2 round trip is several times that)
    constexpr int DEPTH = RT * CT == 1 ? 4 : 3;          // (round 4, with the loads really batched: 2: 47.8, 3: 45.9, 4: 45.7, 6: 47.1, 9: 48.8, 12: 54 us front end of 3 s)
    float4 a_cur[DEPTH][RT], a_nxt[DEPTH][RT];
    float b_cur[DEPTH][CT][4], b_nxt[DEPTH][CT][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (g_begin + d < g_end) load_group(g_begin + d, a_cur[d], b_cur[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (g_begin + d < g_end) mask_group(g_begin + d, a_cur[d], b_cur[d]);
    for (int g = g_begin; g < g_end; g += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (g + DEPTH + d < g_end) load_group(g + DEPTH + d, a_nxt[d], b_nxt[d]);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g + d < g_end) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        f32x16 c = acc[rt][ct];
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[d][rt].x, b_cur[d][ct][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[d][rt].y, b_cur[d][ct][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[d][rt].z, b_cur[d][ct][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[d][rt].w, b_cur[d][ct][3], c, 0, 0, 0);
                        acc[rt][ct] = c;
                    }
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (g + DEPTH + d < g_end) mask_group(g + DEPTH + d, a_nxt[d], b_nxt[d]);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a_cur[d][rt] = a_nxt[d][rt];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int st = 0; st < 4; ++st) b_cur[d][ct][st] = b_nxt[d][ct][st];
        }
    }
    // reduce the four K quarters: every wave parks its partial tiles in LDS, tile (rt, ct) is then finished by wave
    // rt * CT + ct (1 x 1: by wave 0), which adds the quarters in wave order
    auto slot = [&](int w, int tile, int r) { return red + (((w * RT * CT + tile) * 16 + r) * 64 + lane); };
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int tile = rt * CT + ct;
            if (wave != tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) *slot(wave, tile, r) = acc[rt][ct][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int tile = rt * CT + ct;
            if (wave != tile || !col_ok[ct]) continue;
            const int col = n0 + 32 * ct + lrow;
            const float bias = p.bias ? p.bias[col] : 0.f;
            const float slope = p.alpha ? p.alpha[col] : p.leaky;
            const bool act = p.alpha != nullptr || p.use_leaky;
            float *ob = p.out + (long long)b * p.out_bstride;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < rows) {
                    // quarters in wave order 0, 1, 2, 3; the one of this wave is in registers
                    float q[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) q[w] = w == tile ? acc[rt][ct][r] : *slot(w, tile, r);
                    float v = ((q[0] + q[1]) + q[2]) + q[3] + bias;
                    if (act) v = v > 0.f ? v : slope * v;
                    ob[(long long)row * p.ldo + col] = v;
                }
            }
        }
}

// scalar 0 / 1 flags and selects that stay on the scalar unit
__device__ __forceinline__ int s_flag_ge(int a, int b) {
    int r;
    asm("s_cmp_ge_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}
__device__ __forceinline__ int s_select(int flag, int a, int b) {          // flag ? a : b
    int r;
    asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(r) : "s"(flag), "s"(a), "s"(b) : "scc");
    return r;
}
__device__ __forceinline__ const float *s_ptr_add(const float *base, int byte_off) {
    const unsigned long long u = (unsigned long long)(uintptr_t)base;
    unsigned lo, hi;
    asm("s_add_u32 %0, %2, %4\n\ts_addc_u32 %1, %3, 0" : "=&s"(lo), "=&s"(hi) : "s"((unsigned)u), "s"((unsigned)(u >> 32)), "s"(byte_off) : "scc");
    return reinterpret_cast<const float *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}

// a load from global memory at (wave-uniform base + per-lane byte offset): said with a global address-space pointer, so that the
// compiler emits global_load (scalar base + vector offset) and counts it with vmcnt only -- a pointer rebuilt from integers is
// a generic one: flat_load, 64-bit vector address arithmetic per load and waits on lgkmcnt as well
typedef float g_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float g_load_f32(const float *sbase, unsigned voff_bytes) {
    typedef __attribute__((address_space(1))) const char g_char;
    typedef __attribute__((address_space(1))) const float g_float;
    return *reinterpret_cast<g_float *>((g_char *)sbase + voff_bytes);
}
__device__ __forceinline__ float4 g_load_f32x4(const float *sbase, unsigned voff_bytes) {
    typedef __attribute__((address_space(1))) const char g_char;
    typedef __attribute__((address_space(1))) const g_f32x4 g_quad;
    const g_f32x4 v = *reinterpret_cast<g_quad *>((g_char *)sbase + voff_bytes);
    return make_float4(v.x, v.y, v.z, v.w);
}

// Round 5: conv1d_small_tile<1, 1> rewritten for its issue budget (the same sums in the same order, hence the same bits).
// A streaming tick is 64 items of a dozen rows, a 3 s utterance 240 rows: the launch is a few thousand short blocks, eight of
// them resident per CU, and what a wave does between its MFMAs competes with the other waves' MFMAs for the SIMD (DESIGN.md
// section 4).  The round-4 loop spent ~45 vector instructions per group of 4 MFMAs: the tap by division and map_row per
// group, 64-bit address arithmetic for five loads, the row masks, and the copy of the prefetched batch into the current one.
// Now: tap / channel group advance in scalar registers, a load is a scalar base + a per-lane 32-bit offset computed once (per
// tap at the item's edges), the two operand sets ping-pong, the row mask is formed once per tap and skipped in the interior.
template <bool INTERIOR>
__device__ __forceinline__ void conv1d_small_tile32_body(const ConvArgs &p, int bx, int by, int b, float *red, int rows) {
    const int m0 = bx * 32, n0 = by * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int cin = p.cin, cout = p.cout, ldx = p.ldx, dil = p.dil, pad_l = p.pad_l;
    const int gpt = cin >> 3;                               // groups of 8 input channels per tap
    const int n_groups = p.ks * gpt;
    const int g_begin = (n_groups * wave) / 4, g_end = (n_groups * (wave + 1)) / 4;
    const int col = n0 + lrow;
    const bool col_ok = col < cout;
    // weights: W[8 g + 4 lk + st][col] = scalar base of the group + the lane's offset of step st
    unsigned w_voff[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) w_voff[st] = (unsigned)((4 * lk + st) * cout + min(col, cout - 1)) * 4u;
    // activations: x[source row of the lane at the tap][8 cg + 4 lk ..] = scalar base (item + channel group; in the interior +
    // tap) + the lane's offset; at the item's edges the offset holds the mapped row of the tap and `a_ok` says whether it exists
    unsigned a_voff = (unsigned)((m0 + lrow - pad_l) * ldx + 4 * lk) * 4u;       // interior: relative to tap 0 (never negative there)
    bool a_ok = true;
    int c_tap = __builtin_amdgcn_readfirstlane(g_begin / gpt);
    int c_cg = __builtin_amdgcn_readfirstlane(g_begin - (g_begin / gpt) * gpt);
    auto edge_tap = [&]() {                                  // (!INTERIOR) the lane's source row at tap c_tap
        const int src = map_row(m0 + lrow - pad_l + c_tap * dil, rows, p.pad_mode);
        a_ok = src >= 0;
        a_voff = (unsigned)(max(src, 0) * ldx + 4 * lk) * 4u;
    };
    if (!INTERIOR) edge_tap();
    constexpr int DEPTH = 4;                                 // groups in flight beside the batch being multiplied (round 4: 2: 47.8, 3: 45.9, 4: 45.7, 6: 47.1 us front end of 3 s)
    struct Set {
        float4 a[DEPTH];
        float b[DEPTH][4];
        bool ok[DEPTH];
    };
    auto request = [&](Set &s, int g0) {                     // groups g0 .. g0 + DEPTH - 1 (those below g_end) at the cursor
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g0 + d < g_end) {
                const float *xs = s_ptr_add(xb, ((INTERIOR ? c_tap * dil * ldx : 0) + c_cg * 8) * 4);
                s.a[d] = g_load_f32x4(xs, a_voff);
                s.ok[d] = a_ok;
                const float *ws = s_ptr_add(p.w, (c_tap * cin + c_cg * 8) * cout * 4);
#pragma unroll
                for (int st = 0; st < 4; ++st) s.b[d][st] = g_load_f32(ws, w_voff[st]);
                const int wrap = s_flag_ge(c_cg + 1, gpt);
                c_cg = s_select(wrap, 0, c_cg + 1);
                c_tap += wrap;
                if (!INTERIOR && wrap) edge_tap();
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto consume = [&](Set &s, int g0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g0 + d < g_end) {
                float4 av = s.a[d];
                if (!INTERIOR) {
                    av.x = s.ok[d] ? av.x : 0.f;
                    av.y = s.ok[d] ? av.y : 0.f;
                    av.z = s.ok[d] ? av.z : 0.f;
                    av.w = s.ok[d] ? av.w : 0.f;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, s.b[d][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, s.b[d][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, s.b[d][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, s.b[d][3], acc, 0, 0, 0);
            }
        }
    };
    Set s0, s1;
    request(s0, g_begin);
    for (int g = g_begin; g < g_end; g += 2 * DEPTH) {
        request(s1, g + DEPTH);
        consume(s0, g);
        if (g + DEPTH >= g_end) break;
        request(s0, g + 2 * DEPTH);
        consume(s1, g + DEPTH);
    }
    // reduce the four K quarters: the waves 1..3 park their partial tile in LDS, wave 0 adds the quarters in wave order
    auto slot = [&](int w, int r) { return red + ((w * 16 + r) * 64 + lane); };
    if (wave != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) *slot(wave, r) = acc[r];
    }
    __syncthreads();
    if (wave != 0 || !col_ok) return;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const float slope = p.alpha ? p.alpha[col] : p.leaky;
    const bool act = p.alpha != nullptr || p.use_leaky;
    float *ob = p.out + (long long)b * p.out_bstride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < rows) {
            float v = ((acc[r] + *slot(1, r)) + *slot(2, r)) + *slot(3, r) + bias;
            if (act) v = v > 0.f ? v : slope * v;
            ob[(long long)row * p.ldo + col] = v;
        }
    }
}

__device__ __forceinline__ void conv1d_small_tile32(const ConvArgs &p, int bx, int by, int b, float *red) {
    const int rows = __builtin_amdgcn_readfirstlane(item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows));
    const int m0 = bx * 32;
    if (m0 >= rows) return;
    if (m0 - p.pad_l >= 0 && m0 + 31 - p.pad_l + (p.ks - 1) * p.dil < rows) conv1d_small_tile32_body<true>(p, bx, by, b, red, rows);
    else conv1d_small_tile32_body<false>(p, bx, by, b, red, rows);
}

// ---------------------------------------------------------------------------------------------------------
// The F0-net's convolutions in float64 (ConvArgs::precise; mbx_config.f0_accumulate, the default).
// The F0 contour is the one quantity of the graph that is INTEGRATED (phase = running float32 sum of f0 / pulse_rate,
// reference tf_wavetable.py:429-492): a contour that differs from the exact one in the last bit of a few samples sends the
// float32 phase chain down another rounding path, the difference stays for the rest of the utterance and moves every pulse
// behind it -- measured end to end it, not the WaveNet, spent the float32 error budget, and the error grew with the length
// of the utterance (scripts/experiments/f0_error_probe.py).  The net is mel-rate and < 1 % of the
// work, so it runs on v_mfma_f64_16x16x4_f64.  Three operand modes of one tile:
//   x float32, W float32           float64 accumulation only (sub-net shapes outside the full-float64 pattern, handles
//                                  without the *.w64 tensors): one rounding to float32 per output
//   x float32, W float64 (w64)     first layer of the full-float64 chain: reads the mel input, writes float64 (out64)
//   x float64 (x64), W float64     hidden layers of the chain
// The float64 weights are the exact weight-norm fold g v / |v| (host, engine.tensor_table "<layer>.w64").
// Same skeleton as conv1d_small_tile: operands straight from L2, K in groups of 16 channels of a tap split over the block's
// four waves, the quarters summed through LDS in wave order -- every output is summed in the same order whatever RT x CT, so a
// padded batch stays bit-identical to one-at-a-time runs.  Lane (r = lane & 15, kq = lane >> 4) loads A[r][16g + 4kq .. + 3]
// for four MFMA steps (step s contracts the channels 16g + 4kq' + s, kq' = 0..3) and W[16g + 4kq + s][col r].
// C/D layout of the f64 MFMA (MI355X_MICROARCH.md): col = lane & 15, row = (lane >> 4) + 4 * reg.
// Needs cin % 4 == 0 and 16-byte aligned rows (checked by the launcher; other shapes keep the float32 kernels).
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <bool F64> struct F64Operand;
template <> struct F64Operand<false> {
    typedef float scalar;
    typedef float4 quad;
};
template <> struct F64Operand<true> {
    typedef double scalar;
    typedef double4 quad;
};

template <int RT, int CT, bool XF64, bool WF64>
__device__ __forceinline__ void conv1d_f64_tile(const ConvArgs &p, int bx, int by, int b, double *red) {
    typedef typename F64Operand<XF64>::scalar xs_t;
    typedef typename F64Operand<XF64>::quad xq_t;
    typedef typename F64Operand<WF64>::scalar ws_t;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = bx * 16 * RT;
    if (m0 >= rows) return;
    const int n0 = by * 16 * CT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const xs_t *xb = (XF64 ? reinterpret_cast<const xs_t *>(p.x64) : reinterpret_cast<const xs_t *>(p.x)) + (long long)b * p.x_bstride;
    const ws_t *wbase = WF64 ? reinterpret_cast<const ws_t *>(p.w64) : reinterpret_cast<const ws_t *>(p.w);
    const int groups_per_tap = (p.cin + 15) >> 4;          // groups of 16 input channels (the last one of a tap may be short)
    const int n_groups = p.ks * groups_per_tap;
    const int g_begin = (n_groups * wave) / 4, g_end = (n_groups * (wave + 1)) / 4;
    bool col_ok[CT];
    const ws_t *wcol[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = n0 + 16 * ct + r16;
        col_ok[ct] = col < p.cout;
        wcol[ct] = wbase + min(col, p.cout - 1);
    }
    f64x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = 0.0;

    // requests only (clamped addresses); the masks follow when the batch is consumed (see conv1d_small_tile)
    auto load_group = [&](int g, xq_t (&av)[RT], ws_t (&bv)[CT][4]) {
        const int tap = g / groups_per_tap;
        const int ci = min((g - tap * groups_per_tap) * 16 + 4 * kq, p.cin - 4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int src = map_row(m0 + 16 * rt + r16 - p.pad_l + tap * p.dil, rows, p.pad_mode);
            av[rt] = *reinterpret_cast<const xq_t *>(xb + (long long)max(src, 0) * p.ldx + ci);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const ws_t *wk = wcol[ct] + (long long)(tap * p.cin + ci) * p.cout;
#pragma unroll
            for (int st = 0; st < 4; ++st) bv[ct][st] = wk[(long long)st * p.cout];
        }
    };
    auto mask_group = [&](int g, xq_t (&av)[RT]) {
        const int tap = g / groups_per_tap;
        const bool ci_ok = (g - tap * groups_per_tap) * 16 + 4 * kq < p.cin;      // short last group of a tap
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bool ok = ci_ok && map_row(m0 + 16 * rt + r16 - p.pad_l + tap * p.dil, rows, p.pad_mode) >= 0;
            av[rt].x = ok ? av[rt].x : (xs_t)0;
            av[rt].y = ok ? av[rt].y : (xs_t)0;
            av[rt].z = ok ? av[rt].z : (xs_t)0;
            av[rt].w = ok ? av[rt].w : (xs_t)0;
        }
    };
    // groups in flight beside the one being multiplied: a 16 x 16 tile spends 4 MFMAs (256 cycles) on a group, an L2 round
    // trip is several times that; a 32 x 32 tile 16 MFMAs
    constexpr int DEPTH = RT * CT == 1 ? (XF64 ? 3 : 4) : (XF64 ? 1 : 2);
    xq_t a_cur[DEPTH][RT], a_nxt[DEPTH][RT];
    ws_t b_cur[DEPTH][CT][4], b_nxt[DEPTH][CT][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (g_begin + d < g_end) load_group(g_begin + d, a_cur[d], b_cur[d]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (g_begin + d < g_end) mask_group(g_begin + d, a_cur[d]);
    for (int g = g_begin; g < g_end; g += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (g + DEPTH + d < g_end) load_group(g + DEPTH + d, a_nxt[d], b_nxt[d]);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g + d < g_end) {
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const xs_t af = st == 0 ? a_cur[d][rt].x : st == 1 ? a_cur[d][rt].y : st == 2 ? a_cur[d][rt].z : a_cur[d][rt].w;
                        const double a64 = (double)af;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a64, (double)b_cur[d][ct][st], acc[rt][ct], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (g + DEPTH + d < g_end) mask_group(g + DEPTH + d, a_nxt[d]);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a_cur[d][rt] = a_nxt[d][rt];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int st = 0; st < 4; ++st) b_cur[d][ct][st] = b_nxt[d][ct][st];
        }
    }
    // the four K quarters: every wave parks its partial tiles in LDS, tile t is finished by wave t % 4, which adds the
    // quarters in wave order (and, writing float32, rounds ONCE)
    auto slot = [&](int w, int tile, int r) { return red + (((w * RT * CT + tile) * 4 + r) * 64 + lane); };
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int tile = rt * CT + ct;
            if (wave != (tile & 3)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) *slot(wave, tile, r) = acc[rt][ct][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int tile = rt * CT + ct;
            if (wave != (tile & 3) || !col_ok[ct]) continue;
            const int col = n0 + 16 * ct + r16;
            const double bias = p.bias ? (double)p.bias[col] : 0.0;
            const double slope = (double)(p.alpha ? p.alpha[col] : p.leaky);
            const bool act = p.alpha != nullptr || p.use_leaky;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * rt + kq + 4 * r;
                if (row < rows) {
                    double q[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) q[w] = w == (tile & 3) ? acc[rt][ct][r] : *slot(w, tile, r);
                    double v = ((q[0] + q[1]) + q[2]) + q[3] + bias;
                    if (act) v = v > 0.0 ? v : slope * v;
                    const long long at = (long long)b * p.out_bstride + (long long)row * p.ldo + col;
                    if (p.out64) p.out64[at] = v;
                    else p.out[at] = (float)v;
                }
            }
        }
}

// The 32 x 32 float64 tile of the large launches (round 5).  Same sums in the same order as conv1d_f64_tile -- wave w contracts
// K quarter w of all four 16 x 16 sub-tiles, the quarters meet through LDS in wave order --, but the loop is written for the
// issue budget of a CU it shares with fp32 MFMA blocks (every vector instruction of a wave waits for a gap in the co-resident
// waves' MFMA bursts): tap / channel group of a quarter advance in scalar registers (no division per group), the operand
// addresses are a per-lane base + one scalar offset per group, two loop bodies ping-pong the operand registers (no copies),
// row masks only in the first / last tile of an item.
template <bool XF64, bool WF64, bool INTERIOR>
__device__ __forceinline__ void conv1d_f64_tile32_body(const ConvArgs &p, int bx, int by, int b, double *red, int rows) {
    typedef typename F64Operand<XF64>::scalar xs_t;
    typedef typename F64Operand<XF64>::quad xq_t;
    typedef typename F64Operand<WF64>::scalar ws_t;
    const int m0 = bx * 32, n0 = by * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const xs_t *xb = (XF64 ? reinterpret_cast<const xs_t *>(p.x64) : reinterpret_cast<const xs_t *>(p.x)) + (long long)b * p.x_bstride;
    const ws_t *wbase = WF64 ? reinterpret_cast<const ws_t *>(p.w64) : reinterpret_cast<const ws_t *>(p.w);
    const int cin = p.cin, cout = p.cout, ldx = p.ldx;
    const int gpt = (cin + 15) >> 4;                        // groups of 16 input channels per tap (the last one may be short)
    const int n_groups = p.ks * gpt;
    const int g_begin = (n_groups * wave) / 4, g_end = (n_groups * (wave + 1)) / 4;
    const bool short_tail = (cin & 15) != 0;
    bool col_ok[2];
    const ws_t *wlane[2];                                   // W[4 kq][column] of the lane
    const xs_t *xlane[2];                                   // x[tile row of the lane at tap 0][4 kq]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int col = n0 + 16 * ct + r16;
        col_ok[ct] = col < cout;
        wlane[ct] = wbase + min(col, cout - 1) + (long long)(4 * kq) * cout;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) xlane[rt] = xb + (long long)(m0 + 16 * rt + r16 - p.pad_l) * ldx + 4 * kq;
    f64x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = 0.0;

    // scalar cursor of the quarter: tap and channel group of the next group to request
    int c_tap = g_begin / gpt, c_cg = g_begin - c_tap * gpt;
    c_tap = __builtin_amdgcn_readfirstlane(c_tap);
    c_cg = __builtin_amdgcn_readfirstlane(c_cg);
    auto request = [&](xq_t (&av)[2], ws_t (&bv)[2][4]) {   // the group at the cursor; the cursor moves on
        // channels 16 cg + 4 kq .. + 3 of the lane; behind cin (short last group of a tap) the lane reads the tap's last four
        // channels and is masked when the group is consumed
        const int ci_lane = short_tail ? min(16 * c_cg + 4 * kq, cin - 4) - 4 * kq : 16 * c_cg;
        if (INTERIOR) {
            const long long xoff = (long long)c_tap * p.dil * ldx + ci_lane;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) av[rt] = *reinterpret_cast<const xq_t *>(xlane[rt] + xoff);
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int src = map_row(m0 + 16 * rt + r16 - p.pad_l + c_tap * p.dil, rows, p.pad_mode);
                av[rt] = *reinterpret_cast<const xq_t *>(xb + (long long)max(src, 0) * ldx + 4 * kq + ci_lane);
            }
        }
        const long long woff = ((long long)c_tap * cin + ci_lane) * cout;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int st = 0; st < 4; ++st) bv[ct][st] = wlane[ct][woff + (long long)st * cout];
        const int wrap = s_flag_ge(c_cg + 1, gpt);
        c_cg = s_select(wrap, 0, c_cg + 1);
        c_tap += wrap;
    };
    auto consume = [&](int tap, int cg, xq_t (&av)[2], ws_t (&bv)[2][4]) {
        if (!INTERIOR || short_tail) {
            const bool ci_ok = 16 * cg + 4 * kq < cin;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const bool ok = ci_ok && (INTERIOR || map_row(m0 + 16 * rt + r16 - p.pad_l + tap * p.dil, rows, p.pad_mode) >= 0);
                av[rt].x = ok ? av[rt].x : (xs_t)0;
                av[rt].y = ok ? av[rt].y : (xs_t)0;
                av[rt].z = ok ? av[rt].z : (xs_t)0;
                av[rt].w = ok ? av[rt].w : (xs_t)0;
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const xs_t af = st == 0 ? av[rt].x : st == 1 ? av[rt].y : st == 2 ? av[rt].z : av[rt].w;
                const double a64 = (double)af;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a64, (double)bv[ct][st], acc[rt][ct], 0, 0, 0);
            }
    };
    xq_t a0[2], a1[2];
    ws_t b0[2][4], b1[2][4];
    int t0 = c_tap, g0 = c_cg, t1 = 0, g1 = 0;               // tap / channel group of the groups held in set 0 / 1
    if (g_begin < g_end) request(a0, b0);
    for (int g = g_begin; g < g_end; g += 2) {
        if (g + 1 < g_end) {
            t1 = c_tap;
            g1 = c_cg;
            request(a1, b1);
        }
        consume(t0, g0, a0, b0);
        if (g + 1 >= g_end) break;
        if (g + 2 < g_end) {
            t0 = c_tap;
            g0 = c_cg;
            request(a0, b0);
        }
        consume(t1, g1, a1, b1);
    }
    // the four K quarters: every wave parks its partial tiles in LDS, tile t is finished by wave t, which adds the quarters in
    // wave order (and, writing float32, rounds ONCE)
    auto slot = [&](int w, int tile, int r) { return red + (((w * 4 + tile) * 4 + r) * 64 + lane); };
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int tile = rt * 2 + ct;
            if (wave != tile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) *slot(wave, tile, r) = acc[rt][ct][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int tile = rt * 2 + ct;
            if (wave != tile || !col_ok[ct]) continue;
            const int col = n0 + 16 * ct + r16;
            const double bias = p.bias ? (double)p.bias[col] : 0.0;
            const double slope = (double)(p.alpha ? p.alpha[col] : p.leaky);
            const bool act = p.alpha != nullptr || p.use_leaky;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * rt + kq + 4 * r;
                if (row < rows) {
                    double q[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) q[w] = w == tile ? acc[rt][ct][r] : *slot(w, tile, r);
                    double v = ((q[0] + q[1]) + q[2]) + q[3] + bias;
                    if (act) v = v > 0.0 ? v : slope * v;
                    const long long at = (long long)b * p.out_bstride + (long long)row * p.ldo + col;
                    if (p.out64) p.out64[at] = v;
                    else p.out[at] = (float)v;
                }
            }
        }
}

template <bool XF64, bool WF64>
__device__ __forceinline__ void conv1d_f64_tile32(const ConvArgs &p, int bx, int by, int b, double *red) {
    const int rows = __builtin_amdgcn_readfirstlane(item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows));
    const int m0 = bx * 32;
    if (m0 >= rows) return;
    if (m0 - p.pad_l >= 0 && m0 + 31 - p.pad_l + (p.ks - 1) * p.dil < rows) conv1d_f64_tile32_body<XF64, WF64, true>(p, bx, by, b, red, rows);
    else conv1d_f64_tile32_body<XF64, WF64, false>(p, bx, by, b, red, rows);
}

__device__ __forceinline__ void conv1d_f64_dispatch32(const ConvArgs &p, int bx, int by, int b, double *red) {
    if (p.x64) conv1d_f64_tile32<true, true>(p, bx, by, b, red);
    else if (p.w64) conv1d_f64_tile32<false, true>(p, bx, by, b, red);
    else conv1d_f64_tile32<false, false>(p, bx, by, b, red);
}

// operand mode of a float64 member: wave-uniform (kernel arguments)
template <int RT, int CT>
__device__ __forceinline__ void conv1d_f64_dispatch(const ConvArgs &p, int bx, int by, int b, double *red) {
    if (p.x64) conv1d_f64_tile<RT, CT, true, true>(p, bx, by, b, red);
    else if (p.w64) conv1d_f64_tile<RT, CT, false, true>(p, bx, by, b, red);
    else conv1d_f64_tile<RT, CT, false, false>(p, bx, by, b, red);
}

__global__ __launch_bounds__(256) void conv1d_f64_kernel(ConvArgs p) {
    __shared__ double red[4 * 4 * 64];
    conv1d_f64_dispatch<1, 1>(p, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

__global__ __launch_bounds__(256) void conv1d_small_kernel(ConvArgs p) {
    __shared__ float red[4 * 16 * 64];
    conv1d_small_tile32(p, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// Up to three independent small convolutions in one launch (the n-th layers of the F0-net, the VTF-net and the
// conditioning convolution all read the mel input): at a few hundred rows a launch is latency-bound, so sharing it
// costs nothing and removes launches from the critical path.  Blocks [start[k], start[k+1]) belong to convolution k.
struct SmallConvGroup {
    ConvArgs c[3];
    int start[4];
    int gx[3], gy[3];
};

template <int RT, int CT>
__global__ __launch_bounds__(256) void conv1d_small_group_kernel(SmallConvGroup g) {
    __shared__ __attribute__((aligned(16))) float red[4 * RT * CT * 16 * 64];      // (the f64 tile needs 4 * 4 * 64 doubles of it)
    const int id = blockIdx.x;
    const int k = (id >= g.start[1]) + (id >= g.start[2]);
    const int local = id - g.start[k];
    const int bx = local % g.gx[k];
    const int t = local / g.gx[k];
    if (g.c[k].precise) conv1d_f64_dispatch<1, 1>(g.c[k], bx, t % g.gy[k], t / g.gy[k], reinterpret_cast<double *>(red));   // 16 x 16 tiles
    else conv1d_small_tile32(g.c[k], bx, t % g.gy[k], t / g.gy[k], red);
}

// The same convolutions at large launches (batch 16 x 10 s: 12 800 rows): an LDS-staged tile kernel with the summation
// order of conv1d_small_tile.  Block = 4 waves, 64 rows x 128 columns; wave w owns the columns 32 w .. 32 w + 31 of both
// 32-row tiles.  K runs in slices of groups of 8 channels that never cross a quarter boundary.  A wave keeps two accumulator
// sets per tile: the chain of the quarter in progress and the running sum q0, q0 + q1, ... -- the same left fold
// ((q0 + q1) + q2) + q3 + bias, the same MFMA steps in the same order, hence bit-identical results.
// (Rounds 3-4 staged the slices through registers -- conv1d_mel_tile, removed in round 5: see below.)
constexpr int MT_COLS = 128;

// Round 5: the slices are brought in by LDS-DMA and the K loop has no vector bookkeeping.
// What the round-4 kernel (conv1d_mel_tile, register-staged) lost, read off in-kernel stamps and ablations (NOTEBOOK.md, round 5): at 16 x
// 10 s its members ran at 0.39 / 0.29 / 0.41 of the matrix peak although neither the loads (ablated: -6 %) nor the LDS traffic
// bound them.  Two or three blocks share a CU; while one wave is inside its burst of 64-cycle MFMAs, every VECTOR instruction of
// a co-resident wave waits for a gap between them (~one MFMA each).  A wave's non-MFMA phase of ~25 vector instructions per
// slice (address arithmetic for 18 loads and 6 ds_writes, cursor selects the compiler put on the vector ALU, LDS addresses)
// thereby took longer than the other wave's MFMA burst -- the SIMD idled 40 % of the time --, the four quarter folds (96 vector
// instructions each) and the epilogue (300) likewise.  Here, per slice of two groups of 8 channels, a wave issues
//   3 LDS-DMA requests (activations: 32 rows x 8 channels of its group and row tile; weights: four k rows x 128 columns),
//   8 LDS reads, 16 MFMAs, one counted wait + one barrier,
// and NO vector ALU instruction: the slice cursor lives in scalar registers (s_cmp / s_cselect through inline asm: the
// compiler lowers a uniform bool -> int through v_cndmask + v_readfirstlane), the request addresses are scalar bases + per-lane
// offsets computed once, the four stages of the ring are four unrolled loop bodies whose LDS offsets are immediates.  Folds are
// packed adds (16 + 8 moves per fold), the epilogue 3.5 vector instructions per output.
// Same MFMA steps in the same order, the same quarter fold: bit-identical to conv1d_small_tile (tests: the large launch against
// the same rows in small launches).
//   activations of a stage: [group j][row tile t][32 rows][2 halves of 4 channels], half h of row r at slot h ^ ((r >> 3) & 1)
//                           (the 16 rows of a ds_read_b128 phase then hit all 16 bank quads)
//   weights of a stage:     [16 k rows][128 columns], the row's 16-byte chunks rotated by 8 positions when (k >> 2) & 1 (the two
//                           k rows of a ds_read_b32 -- lanes 0-31 and 32-63 -- then sit in different banks)
constexpr int M2_NG = 2, M2_STAGES = 4;
constexpr int M2_A_FLOATS = M2_NG * 64 * 8, M2_W_FLOATS = M2_NG * 8 * MT_COLS, M2_STAGE_FLOATS = M2_A_FLOATS + M2_W_FLOATS;

__device__ __forceinline__ void lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase) : "memory", "m0");
}
// INTERIOR: every source row of the tile (all taps) lies inside the item -- the activation request is then a scalar base + the
// lanes' constant offsets, and a slice is one basic block (the compiler interleaves its scalar code with the MFMAs)
template <bool INTERIOR>
__device__ __forceinline__ void conv1d_mel_tile_dma_body(const ConvArgs &p, int bx, int by, int b, float *lds, int rows) {
    typedef __attribute__((address_space(3))) float lds_float;
    const int m0 = bx * 64;
    const int n0 = by * MT_COLS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;
    const int gpt = p.cin >> 3;                              // groups of 8 input channels per tap
    const int n_groups = p.ks * gpt;
    const int cout = p.cout, ldx = p.ldx;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // ---- DMA roles.  The wave fetches group jw = wave >> 1 of a slice: the activations of row tile tw = wave & 1 and the k
    // rows 4 (wave & 1) .. + 3 of the group's eight.  (A slice of one group: the waves of group 1 fetch group 0 again -- the
    // same three requests per wave and slice whatever the slice, which is what the counted waits count.)
    const int jw = wave >> 1, tw = wave & 1;
    const int d_row = lane >> 1, d_half = (lane & 1) ^ ((lane >> 4) & 1);            // slot lane & 1 of row lane >> 1 holds half d_half
    const unsigned a_voff = (unsigned)(d_row * ldx + 4 * d_half) * 4u;               // interior of an item: row pitch x row + half
    const unsigned a_dst = lds_base + (unsigned)((jw * 2 + tw) * 256) * 4u;          // + stage
    // weights: instruction i covers k rows kin = 4 tw + 2 i + (lane >> 5) of the group; position lane & 31 of the row holds the
    // chunk (position - 8 ((kin >> 2) & 1)) & 31 = columns n0 + 4 chunk .. + 3 (clamped: columns behind cout are not stored)
    unsigned w_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kin = 4 * tw + 2 * i + (lane >> 5);
        const int chunk = ((lane & 31) - 8 * ((kin >> 2) & 1)) & 31;
        w_voff[i] = (unsigned)(kin * cout + min(n0 + 4 * chunk, cout - 4)) * 4u;
    }
    const unsigned w_dst = lds_base + (unsigned)(M2_A_FLOATS + (jw * 8 + 4 * tw) * MT_COLS) * 4u;      // + 1024 i + stage
    // scalar bases: the tile's first source row at tap 0 (may lie outside the item: only used when the whole 32-row piece is
    // inside) and the weight matrix; per slice a 32-bit byte offset is added (the launcher bounds both tensors by 2^31 bytes)
    const int r_tile = m0 + 32 * tw - p.pad_l;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const float *x_tile = xb + (long long)r_tile * ldx;
    const float *wgt = p.w;
    const int tap_rows = p.dil, tap_bytes = p.dil * ldx * 4, group_bytes = 8 * cout * 4;

    // ---- the slice cursor (scalar registers only): quarter end qe, first group gs, its tap and channel group, and the byte
    // offsets that go with them.  A cursor that has reached the last slice stays there: the requests behind the end fetch the
    // last slice again into a free stage, so every iteration issues the same three requests and waits with the same count.
    int c_q = 0, c_gs = 0, c_qe = n_groups >> 2, c_tap = 0, c_cg = 0;
    auto cursor_groups = [&]() { return min(M2_NG, c_qe - c_gs); };
    auto cursor_info = [&]() {                               // groups | quarter complete << 2: what the MFMA side needs of a slice
        const int n = cursor_groups();
        return n | (s_flag_ge(c_gs + n, c_qe) << 2);
    };
    auto advance = [&]() {
        const int n = cursor_groups();
        const int gs2 = c_gs + n;
        int cg2 = c_cg + n, tap2 = c_tap;
#pragma unroll
        for (int i = 0; i < M2_NG; ++i) {                    // n <= M2_NG wraps at most (gpt >= 1)
            const int wrap = s_flag_ge(cg2, gpt);
            cg2 -= wrap * gpt;
            tap2 += wrap;
        }
        const int next_q = s_flag_ge(gs2, c_qe);             // (quarters are never empty: n_groups >= 4)
        const int q2 = c_q + next_q;
        const int qe2 = s_select(next_q, (n_groups * (q2 + 1)) >> 2, c_qe);
        const int last = s_flag_ge(gs2, n_groups);
        c_gs = s_select(last, c_gs, gs2);
        c_cg = s_select(last, c_cg, cg2);
        c_tap = s_select(last, c_tap, tap2);
        c_q = s_select(last, c_q, q2);
        c_qe = s_select(last, c_qe, qe2);
    };
    auto issue = [&](int stage) {
        const int ng = cursor_groups();
        const int second = jw & s_flag_ge(ng, M2_NG);        // this wave fetches the slice's second group
        const int wrap = second & s_flag_ge(c_cg + 1, gpt);
        const int cg = s_select(wrap, 0, c_cg + second);
        const int tap = c_tap + wrap;
        const int g = c_gs + second;
        const unsigned st_off = (unsigned)(stage * M2_STAGE_FLOATS) * 4u;
        if (INTERIOR) {
            lds_dma16_s(s_ptr_add(x_tile, tap * tap_bytes + cg * 32), a_voff, a_dst + st_off);
        } else {                                             // first / last tile of an item: padding per row
            const int src = map_row(r_tile + tap * tap_rows + d_row, rows, p.pad_mode);
            lds_dma16(src >= 0 ? xb + (long long)src * ldx + cg * 8 + 4 * d_half : p.zeros, a_dst + st_off);
        }
        const float *wk = s_ptr_add(wgt, g * group_bytes);
        lds_dma16_s(wk, w_voff[0], w_dst + st_off);
        lds_dma16_s(wk, w_voff[1], w_dst + st_off + 1024u);
    };

    f32x16 cur[2], sum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[t][r] = sum[t][r] = 0.f;

    int n_slices = 0;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) n_slices += ((((n_groups * (qq + 1)) >> 2) - ((n_groups * qq) >> 2)) + M2_NG - 1) / M2_NG;
    // ring of the slices in flight: what the MFMA side needs of them
    int info0, info1, info2;
    info0 = cursor_info();
    issue(0);
    advance();
    info1 = cursor_info();
    issue(1);
    advance();
    info2 = cursor_info();
    issue(2);
    advance();
    // operand addresses of the MFMA lanes (floats, + stage): activations of (group j, tile t): + (2 j + t) 256; weights of
    // (group j, step st): + j 1024 + st 128
    const int bcol = 32 * wave + lrow;
    const float *al = lds + lrow * 8 + 4 * (lk ^ ((lrow >> 3) & 1));
    const float *bl = lds + M2_A_FLOATS + 4 * lk * MT_COLS + 4 * (((bcol >> 2) + 8 * lk) & 31) + (bcol & 3);

    auto body = [&](auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;         // the stage of this slice; the request goes to the one before it
        // the slice has landed: this wave's requests (all but those of the two slices behind it), then everybody's
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();                                     // ... and the stage read last in the slice before is free
        // the operands of both groups are requested first (a slice of one group reads a stale second group and drops it): their
        // LDS latency passes under the request code
        float4 av[M2_NG][2];
        float bv[M2_NG][4];
#pragma unroll
        for (int j = 0; j < M2_NG; ++j) {
#pragma unroll
            for (int t = 0; t < 2; ++t) av[j][t] = *reinterpret_cast<const float4 *>(al + ST * M2_STAGE_FLOATS + (2 * j + t) * 256);
#pragma unroll
            for (int st = 0; st < 4; ++st) bv[j][st] = bl[ST * M2_STAGE_FLOATS + j * 8 * MT_COLS + st * MT_COLS];
        }
        const int info = info0;
        info0 = info1;
        info1 = info2;
        info2 = cursor_info();
        issue((ST + M2_STAGES - 1) % M2_STAGES);
        advance();
        auto group = [&](int j) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 c = cur[t];
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][t].x, bv[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][t].y, bv[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][t].z, bv[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][t].w, bv[j][3], c, 0, 0, 0);
                cur[t] = c;
            }
        };
        group(0);
        if ((info & 3) == M2_NG) group(1);
        if (info >> 2) {                                     // the quarter is complete: ((q0 + q1) + q2) + q3, q0 as 0 + q0 (the
#pragma unroll                                               // chains start from +0, so q0 is never -0 and 0 + q0 has q0's bits)
            for (int t = 0; t < 2; ++t) {
                sum[t] = sum[t] + cur[t];
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[t][r] = 0.f;
            }
        }
    };
    for (int sl = 0; sl < n_slices; sl += M2_STAGES) {
        body(std::integral_constant<int, 0>{});
        if (sl + 1 >= n_slices) break;
        body(std::integral_constant<int, 1>{});
        if (sl + 2 >= n_slices) break;
        body(std::integral_constant<int, 2>{});
        if (sl + 3 >= n_slices) break;
        body(std::integral_constant<int, 3>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the requests behind the end have landed before the block leaves

    const int col = n0 + bcol;
    if (col >= cout) return;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const float slope = p.alpha ? p.alpha[col] : p.leaky;
    const bool act = p.alpha != nullptr || p.use_leaky;
    float *ob = p.out + (long long)b * p.out_bstride;
    if (m0 + 64 <= rows) {
        // a full tile: one address per lane (row 4 lk of the tile, its column), the 32 rows of the lane at scalar multiples of
        // the row pitch
        float *o0 = ob + (long long)(m0 + 4 * lk) * p.ldo + col;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = sum[t][r] + bias;
                if (act) v = v > 0.f ? v : slope * v;
                o0[(long long)(32 * t + (r & 3) + 8 * (r >> 2)) * p.ldo] = v;
            }
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < rows) {
                float v = sum[t][r] + bias;
                if (act) v = v > 0.f ? v : slope * v;
                ob[(long long)row * p.ldo + col] = v;
            }
        }
}

__device__ __forceinline__ void conv1d_mel_tile_dma(const ConvArgs &p, int bx, int by, int b, float *lds) {
    const int rows = __builtin_amdgcn_readfirstlane(item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows));
    const int m0 = bx * 64;
    if (m0 >= rows) return;
    if (m0 - p.pad_l >= 0 && m0 + 63 - p.pad_l + (p.ks - 1) * p.dil < rows) conv1d_mel_tile_dma_body<true>(p, bx, by, b, lds, rows);
    else conv1d_mel_tile_dma_body<false>(p, bx, by, b, lds, rows);
}

__global__ __launch_bounds__(256, 3) void conv1d_mel_group_kernel(SmallConvGroup g) {
    __shared__ __attribute__((aligned(16))) float lds[M2_STAGES * M2_STAGE_FLOATS];
    const int id = blockIdx.x;
    const int k = (id >= g.start[1]) + (id >= g.start[2]);
    const int local = id - g.start[k];
    const int bx = local % g.gx[k];
    const int t = local / g.gx[k];
    if (g.c[k].precise) conv1d_f64_dispatch32(g.c[k], bx, t % g.gy[k], t / g.gy[k], reinterpret_cast<double *>(lds));     // 32 x 32 tiles
    else conv1d_mel_tile_dma(g.c[k], bx, t % g.gy[k], t / g.gy[k], lds);
}

template <int WM, int WN, int TM, int TN, int EPI, int BK = 16>
static void launch_cfg(const ConvArgs &a, hipStream_t stream, int extra_lds = 0) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ncols = (EPI == EPI_GATE) ? a.channels : a.cout;
    const int bn_eff = (EPI == EPI_GATE) ? BN / 2 : BN;
    dim3 grid((a.max_rows + BM - 1) / BM, (ncols + bn_eff - 1) / bn_eff, a.batch);
    const bool vec = (a.cin % 4 == 0) && (a.ldx % 4 == 0) && (a.x_bstride % 4 == 0) && (a.cout % 4 == 0) &&
                     (EPI != EPI_GATE || a.channels % 4 == 0) && ((uintptr_t)a.x % 16 == 0) &&
                     ((uintptr_t)a.w % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, TM, TN, EPI, true, BK>), grid, dim3(256), extra_lds, stream, a);
    else
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, TM, TN, EPI, false, BK>), grid, dim3(256), extra_lds, stream, a);
}

template <int WM, int WN, int TM, int TN, int EPI>
static bool launch_dma(const ConvArgs &a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const bool vec = (a.cin % 4 == 0) && (a.ldx % 4 == 0) && (a.x_bstride % 4 == 0) && (a.cout % 4 == 0) &&
                     (EPI != EPI_GATE || a.channels % 4 == 0) && ((uintptr_t)a.x % 16 == 0) &&
                     ((uintptr_t)a.w % 16 == 0) && a.zeros != nullptr;
    if (!vec) return false;
    const int ncols = (EPI == EPI_GATE) ? a.channels : a.cout;
    const int bn_eff = (EPI == EPI_GATE) ? BN / 2 : BN;
    ConvArgs r = a;
    r.remap = 1;
    r.n_tiles = (ncols + bn_eff - 1) / bn_eff;
    r.m_tiles_per_item = (a.max_rows + BM - 1) / BM;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    hipLaunchKernelGGL((conv1d_mfma_dma_kernel<WM, WN, TM, TN, EPI>), dim3((unsigned)blocks), dim3(256), 0, stream, r);
    return true;
}

static bool f64_conv_eligible(const ConvArgs &a) {
    // (the launch sequence only sets x64 / w64 / out64 when the shape is regular: rows of 4 k channels, 32-byte aligned)
    if (a.x64) return a.precise && a.w64 && a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && (uintptr_t)a.x64 % 32 == 0;
    return a.precise && a.cin % 4 == 0 && a.cin >= 4 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && (uintptr_t)a.x % 16 == 0;
}

static bool small_conv_eligible(const ConvArgs &a) {
    if (f64_conv_eligible(a)) return true;              // a member of the shared mel-rate launches as well
    // no row limit: the mel-rate convolutions then sum K in the same order at every launch size, which keeps a padded
    // batch bit-identical to one-at-a-time runs (the F0 contour feeds the phase accumulator: rounding there is audible
    // in the last bits everywhere downstream)
    // (32-bit byte offsets inside an item and inside the weight tensor: conv1d_small_tile32)
    return a.cin % 8 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && (uintptr_t)a.x % 16 == 0 &&
           (long long)a.ks * a.cin * a.cout * 4 < (1LL << 31) && ((long long)a.max_rows + a.ks * a.dil) * a.ldx * 4 < (1LL << 31);
}

void launch_conv1d_group(const ConvArgs *convs, int n, hipStream_t stream) {
    int small[3], n_small = 0;
    for (int i = 0; i < n && i < 3; ++i)
        if (convs[i].max_rows > 0 && convs[i].batch > 0 && small_conv_eligible(convs[i])) small[n_small++] = i;
    // large launches: LDS-staged 64 x 128 tiles (conv1d_mel_tile_dma: same sums in the same order), also for a single convolution
    // (the conditioning chains of the WaveNet blocks behind the first one come one by one).  The tile's LDS-DMA needs 16-byte
    // weight chunks, the zero page and tensors below 2^31 bytes (32-bit offsets); other shapes keep the small-launch kernel.
    long long work = 0;
    bool dma_ok = true;
    for (int k = 0; k < n_small; ++k) {
        const ConvArgs &c = convs[small[k]];
        work += (long long)c.max_rows * c.batch;
        if (f64_conv_eligible(c)) continue;
        // (at least four groups of 8 channels: the tile's slice cursor takes every K quarter to be non-empty)
        dma_ok = dma_ok && c.zeros && c.cout % 4 == 0 && c.cout >= 4 && (uintptr_t)c.w % 16 == 0 && c.ks * (c.cin >> 3) >= 4 &&
                 (long long)c.ks * c.cin * c.cout * 4 < (1LL << 31) && ((long long)c.max_rows + c.ks * c.dil) * c.ldx * 4 < (1LL << 31);
    }
    const bool big = work >= 3 * 4096 && dma_ok;
    if (n > 3 || n_small < 1 || (n_small < 2 && !big)) {
        for (int i = 0; i < n; ++i) launch_conv1d(convs[i], EPI_LINEAR, stream);
        return;
    }
    // Block order inside the launch: the float32 members first, the longest chain (largest K) in front, so that the MFMA-bound
    // blocks are resident from the start; the float64 members (short, latency-bound blocks) behind them fill the slots the others
    // leave.  Round 5, 16 x 10 s, same box: float64 last 241 us front end, by K alone 257, float64 first 294.
    auto before = [&](int a, int b) {                        // member a goes in front of member b
        const bool fa = f64_conv_eligible(convs[a]), fb = f64_conv_eligible(convs[b]);
        if (fa != fb) return fb;
        return convs[a].ks * convs[a].cin > convs[b].ks * convs[b].cin;
    };
    for (int i = 0; i < n_small; ++i)
        for (int j = i + 1; j < n_small; ++j)
            if (before(small[j], small[i])) {
                const int t = small[i];
                small[i] = small[j];
                small[j] = t;
            }
    const int tile_m = big ? 64 : 32, tile_n = big ? MT_COLS : 32;
    SmallConvGroup g;
    int total = 0;
    for (int k = 0; k < 3; ++k) {
        g.start[k] = total;
        if (k < n_small) {
            g.c[k] = convs[small[k]];
            g.c[k].precise = f64_conv_eligible(g.c[k]) ? 1 : 0;
            // float64-accumulating members (the F0-net): 16 x 16 tiles in the small launches, 32 x 32 in the large ones
            const int tm = g.c[k].precise ? (big ? 32 : 16) : tile_m, tn = g.c[k].precise ? (big ? 32 : 16) : tile_n;
            g.gx[k] = (g.c[k].max_rows + tm - 1) / tm;
            g.gy[k] = (g.c[k].cout + tn - 1) / tn;
            total += g.gx[k] * g.gy[k] * g.c[k].batch;
        } else {
            g.c[k] = convs[small[0]];
            g.gx[k] = g.gy[k] = 1;
        }
    }
    g.start[3] = total;
    for (int k = n_small; k < 3; ++k) g.start[k] = 0x7fffffff;
    if (big) hipLaunchKernelGGL(conv1d_mel_group_kernel, dim3((unsigned)total), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((conv1d_small_group_kernel<1, 1>), dim3((unsigned)total), dim3(256), 0, stream, g);
    for (int i = 0; i < n; ++i) {
        bool in_group = false;
        for (int k = 0; k < n_small; ++k) in_group |= small[k] == i;
        if (!in_group) launch_conv1d(convs[i], EPI_LINEAR, stream);
    }
}

void launch_conv1d(const ConvArgs &a, int epilogue, hipStream_t stream) {
    if (a.max_rows <= 0 || a.batch <= 0) return;
    // Generic forms of the two WaveNet GEMMs (the engine normally runs the specialised kernels of wn_winograd*.hip and
    // wn_resskip.hip; these serve handles created without the packed weight images and shapes those kernels reject).
    // Tile shapes = the fastest measured on MI355X among the variants tried (profiles/README.md):
    //   gate     : LDS-DMA kernel, 64 rows x 64 gate channels; register-staged 64 x 64 when the layout is not 16-byte regular
    //   res/skip : register-staged kernel, 64 x 128, accumulators pre-loaded with the old values
    if (epilogue == EPI_GATE) {
        if (launch_dma<2, 2, 1, 2, EPI_GATE>(a, stream)) return;
        launch_cfg<2, 2, 1, 2, EPI_GATE>(a, stream);
    } else if (epilogue == EPI_RESSKIP) {
        ConvArgs r = a;
        r.acc_preloaded = 1;
        launch_cfg<2, 2, 1, 2, EPI_RESSKIP>(r, stream);
    } else if (f64_conv_eligible(a)) {
        // F0-net convolution on its own (float64 accumulation): split-K 16x16 tiles at every size
        dim3 grid((a.max_rows + 15) / 16, (a.cout + 15) / 16, a.batch);
        hipLaunchKernelGGL(conv1d_f64_kernel, grid, dim3(256), 0, stream, a);
    } else if (small_conv_eligible(a)) {
        // mel-rate sub-nets at small batch: latency bound, split-K 32x32 tiles
        dim3 grid((a.max_rows + 31) / 32, (a.cout + 31) / 32, a.batch);
        hipLaunchKernelGGL(conv1d_small_kernel, grid, dim3(256), 0, stream, a);
    } else if (a.cout <= 32) {
        launch_cfg<4, 1, 1, 1, EPI_LINEAR>(a, stream);        // 128 x 32 (F0 head, post-net, end)
    } else if ((long long)a.max_rows * a.batch >= 4096 && a.cout >= 128) {
        launch_cfg<2, 2, 2, 2, EPI_LINEAR>(a, stream);        // 128 x 128
    } else {
        launch_cfg<2, 2, 1, 1, EPI_LINEAR>(a, stream);        // 64 x 64 (mel-rate sub-nets)
    }
}

}  // namespace mbx
