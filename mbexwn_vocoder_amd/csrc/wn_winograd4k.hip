// Winograd F(4,3) gate layer for SMALL launches: 128-row blocks whose four waves split the input channels.
//
// Same arithmetic as wn_gate_winograd4_kernel (wn_winograd4.hip: combinations, weight image, epilogue), other work
// split.  At batch 1 a launch is only a few rounds of resident blocks, so what decides its time is how finely the work
// divides over the 1024 SIMDs, not the MFMA count alone.  Here a block is 128 rows (32 groups) x 32 gate channels and
// its waves are 2 (column halves [16 tanh | 16 sigmoid], wn) x 2 (channel halves, kh): wave (wn, kh) contracts the
// channels 16 s + 8 kh .. + 7 of every double slice s, i.e. half of K, so a wave issues 480 MFMAs instead of the 960 of
// the large shape and a 10 s utterance becomes 1250 blocks of half the duration.  The two partial sums of a column
// half meet through LDS before the epilogue; each of the two waves then finishes half of the groups.
//
// LDS: two stages x two 8-channel sub-tiles (A in read order, 192 cells x 32 B, + packed weights 12 KB, as in
// wn_winograd4.hip) = 72 KB, + 4.5 KB for the conditioning rows (requested once, at the start) -> 2 blocks per CU.
#include <cstdlib>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int K4_ROWS = 128;
constexpr int K4_HALO = 16;
constexpr int K4_AROWS = K4_ROWS + 2 * K4_HALO;        // 160 rows can be needed
constexpr int K4_PHASE = K4_ROWS / 4 + K4_HALO;        // 48 cells per phase
constexpr int K4_CELLS = 4 * K4_PHASE;                 // 192
constexpr int K4_SUB_A = K4_CELLS * 8;                 // 1536 floats
constexpr int K4_SUB_B = 6 * 8 * 64;                   // 3072 floats
constexpr int K4_SUB = K4_SUB_A + K4_SUB_B;            // 4608 floats = 18 KB
constexpr int K4_STAGE = 2 * K4_SUB;                   // one double slice (16 channels)
constexpr int K4_COND = 2 * K4_STAGE;                  // float offset of the conditioning tile (16 rows x 64) + lerp tables

__device__ __forceinline__ void k4_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}

// same with a wave-uniform base address and a per-lane 32-bit byte offset
__device__ __forceinline__ void k4_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

__device__ __forceinline__ float k4_gate_act(float zt, float zs) {
    const float e2 = __expf(2.0f * zt);
    const float e1 = __expf(-zs);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
    return th * __builtin_amdgcn_rcpf(1.0f + e1);
}

__device__ __forceinline__ float4 k4_fma(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 k4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 k4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// input combinations of product pair g (see wn_winograd4.hip)
__device__ __forceinline__ void k4_input_comb(int g, const float4 (&x)[6], float4 &u0, float4 &u1) {
    if (g == 0) {
        u0 = k4_fma(4.f, x[0], k4_fma(-5.f, x[2], x[4]));
        const float4 a = k4_fma(-4.f, x[2], x[4]), b = k4_fma(-4.f, x[1], x[3]);
        u1 = k4_add(a, b);
    } else if (g == 1) {
        const float4 a = k4_fma(-4.f, x[2], x[4]), b = k4_fma(-4.f, x[1], x[3]);
        u0 = k4_sub(a, b);
        const float4 c = k4_sub(x[4], x[2]), e = k4_sub(x[3], x[1]);
        u1 = k4_fma(2.f, e, c);
    } else {
        const float4 c = k4_sub(x[4], x[2]), e = k4_sub(x[3], x[1]);
        u0 = k4_fma(-2.f, e, c);
        u1 = k4_fma(4.f, x[1], k4_fma(-5.f, x[3], x[5]));
    }
}

__global__ __launch_bounds__(256, 2) void wn_gate_winograd4k_kernel(ConvArgs p, int log2d) {
    typedef __attribute__((address_space(3))) float lds_float;
    __shared__ __attribute__((aligned(16))) float lds[K4_COND + 1024 + 128];
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
    const int m0 = mt * K4_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, kh = wave >> 1;
    const int lrow = lane & 31, lk = lane >> 5;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nk8 = (p.cin + 7) / 8;               // 8-channel slices of the weight image
    const int nds = (nk8 + 1) / 2;                 // double slices

    // ---- per-lane LDS-DMA sources: 12 A instructions per stage (2 sub-tiles x 6), 3 per wave
    int a_off[3], a_ch[3], a_sub[3], a_inst[3];
    unsigned a_voff[3];
    unsigned a_ok = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ii = wave + 4 * i;                                  // 0..11
        a_sub[i] = ii / 6;
        a_inst[i] = ii - 6 * a_sub[i];
        const int pos = a_inst[i] * 64 + lane;
        const int cell = pos >> 1;
        const int phase = cell / K4_PHASE, sidx = cell - phase * K4_PHASE;
        const int m = 4 * (sidx >> log2d) + phase;
        const int row = (m << log2d) + (sidx & (d - 1)) + K4_HALO - d;   // staged row index, m0 - 16 + row = source
        const int src = m0 - K4_HALO + row;
        a_ch[i] = 8 * a_sub[i] + 4 * ((pos & 1) ^ ((cell >> 3) & 1));
        a_off[i] = min(max(src, 0), rows - 1) * p.ldx;
        if (row < K4_AROWS && src >= 0 && src < rows) a_ok |= 1u << i;
        a_voff[i] = 4u * (unsigned)(a_off[i] + a_ch[i]);
    }
    // interior blocks (every staged row exists, whole double slices): uniform base + per-lane byte offset, no selects
    const bool fast = p.fast_dma && m0 >= K4_HALO && m0 + K4_ROWS + K4_HALO <= rows && p.cin % 16 == 0;
    const float *wtile = p.w + (long long)nt * nk8 * K4_SUB_B;
    const unsigned b_voff = 16u * (unsigned)lane;
    // 24 B instructions per stage (2 sub-tiles x 12), 6 per wave
    const float *wbase = p.w + (long long)nt * nk8 * K4_SUB_B + lane * 4;
    auto issue = [&](int ds, int stage) {
        const unsigned sdst = lds_base + 4u * (unsigned)(stage * K4_STAGE);
        if (fast) {
            const float *abase = xb + 16 * ds;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                k4_lds_dma16_s(abase, a_voff[i], sdst + 4u * (unsigned)(a_sub[i] * K4_SUB) + 1024u * (unsigned)a_inst[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int ii = wave + 4 * i;                          // 0..23
                const int sub = ii / 12, k = ii - 12 * sub;
                k4_lds_dma16_s(wtile + (long long)(2 * ds + sub) * K4_SUB_B + k * 256, b_voff,
                               sdst + 4u * (unsigned)(sub * K4_SUB + K4_SUB_A) + 1024u * (unsigned)k);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ci = 16 * ds + a_ch[i];
            const bool ok = ((a_ok >> i) & 1u) & (ci < p.cin);
            k4_lds_dma16(ok ? xb + a_off[i] + ci : p.zeros,
                         sdst + 4u * (unsigned)(a_sub[i] * K4_SUB) + 1024u * (unsigned)a_inst[i]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int ii = wave + 4 * i;                              // 0..23
            const int sub = ii / 12, k = ii - 12 * sub;
            const int kt8 = 2 * ds + sub;
            k4_lds_dma16(kt8 < nk8 ? wbase + (long long)kt8 * K4_SUB_B + k * 256 : p.zeros,
                         sdst + 4u * (unsigned)(sub * K4_SUB + K4_SUB_A) + 1024u * (unsigned)k);
        }
    };
    // conditioning rows of this block (16 rows x (32 tanh | 32 sigmoid) columns) and the interpolation weights
    const int cond_up = p.cond_up;
    const int n2 = rows / cond_up;
    const int t2base = m0 / cond_up;
    {
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
        const int pos = wave * 64 + lane;
        const int crow = pos >> 4, cq = pos & 15;
        const int chn = n0 + 4 * (cq & 7);
        const int t = min(t2base + crow, n2 - 1);
        k4_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                     lds_base + 4u * (unsigned)K4_COND + 1024u * (unsigned)wave);
    }
    float *lerp_lds = lds + K4_COND + 1024;
    if (tid < cond_up) {
        lerp_lds[tid] = p.lerp_w0[tid];
        lerp_lds[64 + tid] = p.lerp_w1[tid];
    }

    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    int aoff[6];        // LDS float offsets (inside this wave's sub-tile) of h[t-d] .. h[t+4d], this lane's 4 channels
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int cell = (q & 3) * K4_PHASE + lrow + ((q >> 2) << log2d);
        aoff[q] = 4 * (2 * cell + (lk ^ ((cell >> 3) & 1)));
    }
    float4 X0[6], X1[6];
    float4 B0[2], B1[2];
    auto load_x = [&](int stage, float4 (&x)[6]) {
        const float *ab = lds + stage * K4_STAGE + kh * K4_SUB;
#pragma unroll
        for (int q = 0; q < 6; ++q) x[q] = *reinterpret_cast<const float4 *>(ab + aoff[q]);
    };
    auto load_b = [&](int stage, int g, float4 (&bw)[2]) {
        const float *bb = lds + stage * K4_STAGE + kh * K4_SUB + K4_SUB_A + lane * 4;
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
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
    };
    // One double slice = three operand groups (product pairs) of 8 MFMAs on this wave's 8 channels; structure as in
    // wn_winograd4.hip, with two stages: the barrier of double slice s frees its stage for double slice s+2.
    auto slice = [&](int ds, int stage, float4 (&xc)[6], float4 (&xn)[6], float4 (&ba)[2], float4 (&bb)[2],
                     float4 (&ua)[2], float4 (&ub)[2]) {
        const int nstage = stage ^ 1;
        load_b(stage, 1, bb);
        __builtin_amdgcn_sched_barrier(0);
        k4_input_comb(1, xc, ub[0], ub[1]);
        mfma8(0, ua[0], ua[1], ba);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        load_b(stage, 2, ba);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // double slice s+1 has landed
        __syncthreads();                                           // ... and every wave has requested all of s
        if (ds + 2 < nds) issue(ds + 2, stage);
        load_x(nstage, xn);
        __builtin_amdgcn_sched_barrier(0);
        k4_input_comb(2, xc, ua[0], ua[1]);
        mfma8(1, ub[0], ub[1], bb);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        load_b(nstage, 0, bb);
        __builtin_amdgcn_sched_barrier(0);
        k4_input_comb(0, xn, ub[0], ub[1]);
        mfma8(2, ua[0], ua[1], ba);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    };

    issue(0, 0);
    if (nds > 1) issue(1, 1);
    if (nds > 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // 9 LDS-DMA instructions per wave and stage
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 U0[2], U1[2];
    load_x(0, X0);
    load_b(0, 0, B0);
    k4_input_comb(0, X0, U0[0], U0[1]);
    for (int ds = 0; ds < nds; ds += 2) {
        slice(ds, 0, X0, X1, B0, B1, U0, U1);
        if (ds + 1 < nds) slice(ds + 1, 1, X1, X0, B1, B0, U1, U0);
    }

    // ---- the two channel halves of a column half meet: wave kh keeps the registers r in [8 kh, 8 kh + 8) and hands
    // the other eight of every product to its partner
    __syncthreads();                                               // all LDS operand reads are done
    float *exch = lds;                                             // 4 waves x 6 x 8 x 64 floats = 48 KB
    {
        float *mine = exch + wave * 3072 + lane;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) mine[(j * 8 + rr) * 64] = kh ? acc[j][rr] : acc[j][8 + rr];
    }
    __syncthreads();
    float part[6][8];
    {
        const float *theirs = exch + (wave ^ 2) * 3072 + lane;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                part[j][rr] = (kh ? acc[j][8 + rr] : acc[j][rr]) + theirs[(j * 8 + rr) * 64];
    }

    // ---- epilogue: combine the six products, add bias + conditioning, gate, store (see wn_winograd4.hip)
    const float inv_up = 1.0f / (float)cond_up;
    const float *cl = lds + K4_COND;
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
        obase[(long long)row * p.ldo + n0 + tc] = k4_gate_act(zt, zs);
    };
    const bool tanh_lane = lrow < 16;
    const int tc = 16 * wn + (lrow & 15);
    const bool ch_ok = n0 + tc < C;
    const float bt = (p.bias && ch_ok) ? p.bias[n0 + tc] : 0.f;
    const float bsg = (p.bias && ch_ok) ? p.bias[C + n0 + tc] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * kh + rr;                                                  // accumulator register index
        const int gi = (r & 3) + 8 * (r >> 2) + 4 * lk;                             // group held by this register
        const int t0 = m0 + ((gi >> log2d) << (log2d + 2)) + (gi & (d - 1));
        const float s12 = part[1][rr] + part[2][rr], d12 = part[1][rr] - part[2][rr];
        const float s34 = part[3][rr] + part[4][rr], d34 = part[3][rr] - part[4][rr];
        const float y0 = (part[0][rr] + s12) + s34;
        const float y1 = fmaf(2.f, d34, d12);
        const float y2 = fmaf(4.f, s34, s12);
        const float y3 = fmaf(8.f, d34, d12) + part[5][rr];
        const float ga = __shfl_xor(tanh_lane ? y2 : y0, 16);
        const float gb = __shfl_xor(tanh_lane ? y3 : y1, 16);
        const int ra = t0 + (tanh_lane ? 0 : 2 * d), rb = ra + d;
        if (ch_ok && ra < rows) finish(ra, tc, tanh_lane ? y0 : ga, tanh_lane ? ga : y2, bt, bsg);
        if (ch_ok && rb < rows) finish(rb, tc, tanh_lane ? y1 : gb, tanh_lane ? gb : y3, bt, bsg);
    }
}

// a.w must point at the host-packed F(4,3) weights (ceil(C/32), ceil(C/8), 3072); returns false if the layer does not fit
bool launch_wn_gate_winograd4k(const ConvArgs &a, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= K4_HALO && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros &&
                    a.cond && (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up <= 64 &&
                    K4_ROWS / a.cond_up + 2 <= 16 && a.max_rows < (1 << 24);
    if (!ok) return false;
    ConvArgs r = a;
    r.fast_dma = (long long)a.max_rows * a.ldx * 4 < (1LL << 32);
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + K4_ROWS - 1) / K4_ROWS;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    hipLaunchKernelGGL(wn_gate_winograd4k_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, r, log2d);
    return true;
}

}  // namespace mbx
