// WaveNet residual/skip layer: r = a W + b (1x1, C -> 2C; last layer C -> C), h += r[:, :C], skip (+)= r[:, C:].
// With the skip path folded into the end convolution (engine.fold_skip_weights) the skip half of W is the C x n_out
// product W_skip W_end and "skip" is the n_out-wide WaveNet output accumulator (row stride skip_ld).
//
// Same layer as conv1d_mfma_kernel<EPI_RESSKIP> (reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:322-336:
// res_skip = res_skip_l(acts); x = x + res_skip[:C]; skip_out (+)= res_skip[C:]), restructured like the Winograd gate
// kernels (wn_winograd4w.hip) for large row counts:
//   block = 2 x 2 waves, (64 MT) rows x 128 columns, wave tile (32 MT) x 64; MT = 2 (128-row blocks, 3 per CU) for
//   large launches, MT = 1 (64-row blocks, finer granularity) for small ones
//   A: gate output rows [m0, m0+64 MT) x 16 channels per slice through LDS-DMA, chunk (row, c) at 4*row + (c ^ ((row>>2)&3))
//   B: 16 channels x 128 columns per slice, pre-packed on the host in MFMA operand order
//      [channel half cc][column tile jn][lane][4 k steps] (engine.pack_resskip_weights): one ds_read_b128 per lane
//      = the weight operands of four consecutive MFMAs
//   three (large shape) or two LDS stages of 16 / 12 KB; operand groups of 8 MFMAs, the operands of group n+1 are requested from
//   LDS before the MFMAs of group n issue.
// The accumulators start from (old value + bias), so the epilogue is a plain store (same arithmetic as the
// acc_preloaded path of conv1d_mfma_kernel).  h_init (layer 0 with the start convolution folded in, wn_gate0.hip): the
// input rows are [a | x'] (cin = C + 16), the weights [Wr ; Ws'], and h starts from the bias alone.
#include <cstdlib>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RS_BK = 16;
constexpr int RS_B_FLOATS = RS_BK * 128;     // 2048

__device__ __forceinline__ void rs_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset
__device__ __forceinline__ void rs_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

template <int MT, int NST>
__global__ __launch_bounds__(256, MT == 2 ? 3 : 4) void wn_resskip_kernel(ConvArgs p) {
    constexpr int ROWS = 64 * MT;
    constexpr int RS_A_FLOATS = ROWS * RS_BK;
    constexpr int A_INST = ROWS / 64;          // LDS-DMA instructions per wave (A)
    constexpr int NG = 2 * MT;                 // operand groups (8 MFMAs each) per slice
    typedef __attribute__((address_space(3))) float lds_float;
    constexpr int STAGE = RS_A_FLOATS + RS_B_FLOATS;
    constexpr int N_DMA = A_INST + 2;          // LDS-DMA instructions per wave and slice
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];   // stage s: A at s*STAGE, B behind it
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode (see decode_tile in conv_mfma.hip)
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    if (g >= p.m_tiles_total) return;
    const int b = g / p.m_tiles_per_item;
    const int mt = g - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride + (long long)m0 * p.ldx;   // the block's first row: 32-bit offsets stay small
    const int nk = (p.cin + RS_BK - 1) / RS_BK;

    // ---- per-lane DMA sources
    int a_off[A_INST], a_ch[A_INST];
    unsigned a_voff[A_INST];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int pos = (wave + 4 * i) * 64 + lane;
        const int row = pos >> 2;
        a_ch[i] = 4 * ((pos & 3) ^ ((row >> 2) & 3));
        a_off[i] = (min(m0 + row, rows - 1) - m0) * p.ldx;
        if (m0 + row < rows) a_ok |= 1u << i;
        a_voff[i] = 4u * (unsigned)(a_off[i] + a_ch[i]);
    }
    // full blocks and whole slices: uniform base + per-lane byte offset, no selects
    const bool fast = p.fast_dma && m0 + ROWS <= rows && p.cin % RS_BK == 0;
    const float *wtile = p.w + (long long)nt * nk * RS_B_FLOATS + wave * 256;
    const unsigned b_voff = 16u * (unsigned)lane;
    const float *wsrc = p.w + (long long)nt * nk * RS_B_FLOATS + (wave * 64 + lane) * 4;
    auto issue = [&](int kt, int buf) {
        const int ci0 = kt * RS_BK;
        const unsigned adst = lds_base + 4u * (unsigned)(buf * STAGE);
        const unsigned bdst = adst + 4u * (unsigned)RS_A_FLOATS;
        if (fast) {
            const float *abase = xb + ci0;
#pragma unroll
            for (int i = 0; i < A_INST; ++i) rs_lds_dma16_s(abase, a_voff[i], adst + 1024u * (unsigned)(wave + 4 * i));
            const float *bbase = wtile + (long long)kt * RS_B_FLOATS;
#pragma unroll
            for (int i = 0; i < 2; ++i) rs_lds_dma16_s(bbase + i * 1024, b_voff, bdst + 1024u * (unsigned)(wave + 4 * i));
            return;
        }
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            const int ci = ci0 + a_ch[i];
            const bool ok = ((a_ok >> i) & 1u) & (ci < p.cin);
            rs_lds_dma16(ok ? xb + a_off[i] + ci : p.zeros, adst + 1024u * (unsigned)(wave + 4 * i));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            rs_lds_dma16(wsrc + (long long)kt * RS_B_FLOATS + i * 1024, bdst + 1024u * (unsigned)(wave + 4 * i));
    };
    issue(0, 0);

    // ---- accumulators start from old value + bias (h columns always accumulate, skip columns unless skip_init)
    f32x16 acc[MT][2];
    float *dst[2];
    int ld[2];
    bool col_ok[2];
    const int skip_ld = p.skip_ld ? p.skip_ld : C;
    const long long skip_bstride = p.skip_bstride ? p.skip_bstride : (p.skip_ld ? (long long)p.max_rows * p.skip_ld : p.hs_bstride);
    const int lane_row = m0 + 32 * MT * wm + 4 * lk;      // row of register r of row tile i: lane_row + 32 i + (r & 3) + 8 (r >> 2)
    int off_last[2];                                      // element offset of the last valid row (clamp target)
    {
        // unconditional loads from clamped addresses (no branch, all requests in flight), selected afterwards;
        // element offsets in 32 bits with full-rate 24-bit multiplies (the launcher bounds rows * row stride)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + lrow;
            col_ok[j] = col < p.cout;
            const int colc = min(col, p.cout - 1);
            const bool to_h = (!p.last_layer) && colc < C;
            const int oc = to_h ? colc : (p.last_layer ? colc : colc - C);
            dst[j] = to_h ? p.h + (long long)b * p.hs_bstride + oc : p.skip + (long long)b * skip_bstride + oc;
            ld[j] = to_h ? C : skip_ld;
            off_last[j] = (rows - 1) * ld[j];
            const int off0 = lane_row * ld[j];
            const bool accumulate = (to_h ? !p.h_init : !p.skip_init) && col_ok[j];
            const float bias = (p.bias && col_ok[j]) ? p.bias[colc] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * i + (r & 3) + 8 * (r >> 2);
                    const int off = lane_row + k < rows ? off0 + (int)__umul24(k, ld[j]) : off_last[j];
                    const float old = dst[j][off];
                    acc[i][j][r] = (accumulate ? old : 0.f) + bias;
                }
        }
    }

    int aoff[MT][2];    // LDS float offsets of this lane's A row of row tile i for the two channel halves of a slice
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = 32 * MT * wm + 32 * i + lrow;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) aoff[i][cc] = row * RS_BK + 4 * ((2 * cc + lk) ^ ((row >> 2) & 3));
    }
    float4 Av[2];
    float4 Bv[2][2];
    auto load_a = [&](int buf, int cc, int i, float4 &a) {
        a = *reinterpret_cast<const float4 *>(lds + buf * STAGE + aoff[i][cc]);
    };
    auto load_b = [&](int buf, int cc, float4 (&bw)[2]) {
        const float *bb = lds + buf * STAGE + RS_A_FLOATS + lane * 4;
        bw[0] = *reinterpret_cast<const float4 *>(bb + (cc * 4 + 2 * wn + 0) * 256);
        bw[1] = *reinterpret_cast<const float4 *>(bb + (cc * 4 + 2 * wn + 1) * 256);
    };

    // NST stages: slices 1 .. NST-1 are requested behind slice 0 and the accumulator pre-loads; a slice has landed
    // when at most the (NST-2) slices requested after it are still outstanding (N_DMA instructions each)
    if (nk > 1) issue(1, 1);
    if (NST == 3 && nk > 2) issue(2, 2);
    if (NST == 3 && nk > 2) {
        if (N_DMA == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (NST == 2 && nk > 1) {
        if (N_DMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    load_a(0, 0, 0, Av[0]);
    load_b(0, 0, Bv[0]);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nbuf = buf == NST - 1 ? 0 : buf + 1;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int cc = gi / MT, i = gi % MT;
            if (gi < NG - 1) {
                load_a(buf, (gi + 1) / MT, (gi + 1) % MT, Av[(gi + 1) & 1]);
                if (gi == MT - 1) load_b(buf, 1, Bv[1]);
            } else if (kt + 1 < nk) {
                // slice kt+1 must have landed (with three stages slice kt+2 may still be in flight)
                if (NST == 3 && kt + 2 < nk) {
                    if (N_DMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (kt + NST < nk) issue(kt + NST, buf);
                load_a(nbuf, 0, 0, Av[0]);
                load_b(nbuf, 0, Bv[0]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the requests ahead of this group's MFMAs
            const float4 a = Av[gi & 1];
            const float4(&bw)[2] = Bv[cc];
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[0].x, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[1].x, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[0].y, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[1].y, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[0].z, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[1].z, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[0].w, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[1].w, acc[i][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = nbuf;
    }

    // ---- epilogue: the accumulators are the new values (offsets recomputed: keeping 64 of them live through the
    // K loop would cost occupancy, so the compiler is kept from reusing the ones of the prologue)
    int lane_row_e = lane_row;
    asm volatile("" : "+v"(lane_row_e));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!col_ok[j]) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * i + (r & 3) + 8 * (r >> 2);
                if (lane_row_e + k < rows) dst[j][lane_row_e * ld[j] + (int)__umul24(k, ld[j])] = acc[i][j][r];
            }
    }
}

// a.w must point at the host-packed weights (ceil(cout/128), ceil(C/16), 2048); returns false if the layer does not fit
bool launch_wn_resskip(const ConvArgs &a, hipStream_t stream) {
    const bool ok = a.ks == 1 && (a.h_init ? a.cin >= a.channels && !a.last_layer : a.cin == a.channels) && a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 &&
                    (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros && a.h && a.skip && a.cout > 0 &&
                    (long long)a.max_rows * a.channels < (1LL << 31) &&
                    (a.skip_ld ? a.cout <= a.channels + a.skip_ld : a.cout == (a.last_layer ? a.channels : 2 * a.channels));
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = 1;                 // byte offsets are relative to the block's first row
    r.n_tiles = (a.cout + 127) / 128;
    // 64-row blocks while the 128-row grid is less than three rounds of the 768 resident blocks (3 per CU x 256 CUs)
    const long long big_blocks = (long long)((a.max_rows + 127) / 128) * a.batch * r.n_tiles;
    const bool small = big_blocks < 3 * 768;
    const int tile_rows = small ? 64 : 128;
    r.m_tiles_per_item = (a.max_rows + tile_rows - 1) / tile_rows;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    // stages: three for the large shape (measured -4 % at batch 16), two for the small one (no difference at batch 1)
    if (small) hipLaunchKernelGGL((wn_resskip_kernel<1, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, r);
    else hipLaunchKernelGGL((wn_resskip_kernel<2, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, r);
    return true;
}

}  // namespace mbx
