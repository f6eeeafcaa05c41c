// WaveNet dilated convolution (k = 3) + conditioning + gate in SPLIT half precision, direct form (opt-in:
// mbx_config.wn_precision = MBX_PRECISION_SPLIT_F16; never the default and never the headline measurement -- the float32
// kernels are wn_winograd4w.hip / wn_winograd2w.hip / conv_mfma.hip).
//
// Same layer as wn_gate_winograd4w_kernel (reference MBExWN_NVoc/vocoder/model/custom_AE_layers.py:305-321), computed as
// the plain K = 3C contraction on the 16-bit matrix pipe (16 x the float32 rate) with both operands split into
// hi = fp16(x) and lo' = fp16((x - hi) 2^11):  x y ~ hi hi + 2^-11 (hi lo' + lo' hi)  -- float32-class error (DESIGN.md
// section 9).  Unlike the res/skip layer's input the hidden state h is not bounded, so nothing is pre-scaled: the hi hi
// products go to one accumulator set, the two cross products to a second one, and the epilogue forms main + 2^-11 cross.
// fp16's range bounds |h| at 65 504 (the engine's calibration forward would show an overflow as non-finite audio).
//
// Block = 8 waves, 256 rows x 64 weight columns ([16 tanh | 16 sigmoid] of the even and of the odd gate channels of a
// 32-channel column tile: the epilogue is the one of wn_gate_winograd4w_kernel); wave w owns rows 32 w .. 32 w + 31 (two
// 16-row tiles) x the four column tiles x 2 accumulator sets = 64 registers.  K steps of 32 channels:
//   A: the rows [m0 - 16, m0 + 272) x 32 channels of h land as float32 in a staging tile X (LDS-DMA, 36 KB); the block
//      converts them ONCE into the operand tile Y: per row 64 bytes of hi and 64 bytes of lo' halves, 16-byte chunk c8
//      (channels 8 (c8 & 3) .. + 7, hi: c8 < 4, lo': c8 >= 4) at chunk position c8 ^ ((row >> 1) & 7), so that the 16 lanes
//      of an operand read (16 consecutive rows, one chunk) hit 64 different banks at every tap offset.  (The first version
//      split in registers, once per tap and row tile -- 6 splits of 8 values per wave and step: 1.09 ms per launch, 0.83
//      with the split removed.)  Lane (i = lane & 15, kq = lane >> 4) reads the channels 8 kq .. + 7 of its row;
//   B: 3 taps x 4 column tiles x [hi | lo'] x 64 lanes x 8 halves, packed on the host in MFMA operand order
//      (engine.pack_gate_f16_weights), 24 KB per step, copied verbatim by LDS-DMA into one of two stages.
// Per step: wait for the step's requests, barrier, convert X -> Y, barrier, request the next step (X is free, the other
// weight stage too), multiply.  X + Y + 2 weight stages + conditioning tile + tables: 128.5 KB, one block per CU.
// PLANES = true: the producer (wn_resskip_f16_kernel) has written the hidden state as fp16 planes already (ConvArgs::h_split:
// per row hi and lo' halves): the rows are requested straight into the operand layout, X becomes a second operand stage, the
// conversion and one barrier per step go, and a step's requests are in flight during the whole step before it.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "mbx_kernels.h"

namespace mbx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int GH_ROWS = 256;
constexpr int GH_HALO = 16;
constexpr int GH_AROWS = GH_ROWS + 2 * GH_HALO;            // 288 staged rows
constexpr int GH_BK = 32;
constexpr int GH_A_FLOATS = GH_AROWS * GH_BK;              // 9216 floats = 36 KB
constexpr int GH_B_FLOATS = 3 * 4 * 2 * 256;               // 3 taps x 4 tiles x 2 images x 1 KB = 24 KB
constexpr int GH_X = 0;                                    // float32 landing tile
constexpr int GH_Y = GH_A_FLOATS;                          // operand tile (hi / lo' halves), same size
constexpr int GH_B = 2 * GH_A_FLOATS;                      // two weight stages
constexpr int GH_COND_ROWS = 28;                           // conditioning rows of 64 floats (cond_up >= 10)
constexpr int GH_COND = GH_B + 2 * GH_B_FLOATS;
constexpr int GH_TAB = GH_COND + GH_COND_ROWS * 64;
constexpr int GH_LERP = GH_TAB + GH_ROWS;
constexpr int GH_LDS_FLOATS = GH_LERP + 128;               // 18432 + 12288 + 1792 + 256 + 128 floats = 128.5 KB

__device__ __forceinline__ void gh_lds_dma16(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ void gh_lds_dma16_s(const float *sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

#define GH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x -> (hi, lo'): hi = fp16(x) (round to nearest), lo' = fp16(2^11 x - 2^11 hi): exact in float32 before the final rounding
__device__ __forceinline__ void gh_split4(const f32x4 &x, f16x4 &h, f16x4 &l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 hh = (_Float16)x[i];
        h[i] = hh;
        l[i] = (_Float16)__builtin_fmaf(x[i], 2048.0f, -2048.0f * (float)hh);
    }
}

template <bool PLANES>
__global__ __launch_bounds__(512, 2) void wn_gate_f16_kernel(ConvArgs p, int log2d) {
    typedef __attribute__((address_space(3))) float lds_float;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode (as in wn_gate_winograd4w_kernel): the column tiles of a row tile run back to back on one XCD
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int nt = l % p.n_tiles;
    if (g_ >= p.m_tiles_total) return;
    const int b = g_ / p.m_tiles_per_item;
    const int mt = g_ - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * GH_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int n0 = nt * 32;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const float *xb = p.x + (long long)b * p.x_bstride;
    const int nk = (p.cin + GH_BK - 1) / GH_BK;

    // ---- LDS-DMA requests of a step: 36 pieces of activations (piece j = staged rows 8 j .. 8 j + 7: lane -> row 8 j +
    // (lane >> 3), chunk position lane & 7), dealt round-robin over the 8 waves, and 24 pieces of weights, 3 per wave
    const float *wtile = p.w + (long long)nt * nk * GH_B_FLOATS;
    const unsigned b_voff = 16u * (unsigned)lane;
    const _Float16 *pb = PLANES ? reinterpret_cast<const _Float16 *>(p.h_split + (long long)b * p.h_split_bstride) : nullptr;
    auto issue_a = [&](int kt, int stage) {
        const unsigned adst = lds_base + 4u * (unsigned)(PLANES ? (stage ? GH_X : GH_Y) : GH_X);
        const int ci0 = kt * GH_BK;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int piece = wave + 8 * i;
            if (piece >= GH_AROWS / 8) break;
            const int r = 8 * piece + (lane >> 3);                 // staged row: source row m0 - 16 + r
            const int src_row = m0 - GH_HALO + r;
            if (PLANES) {
                // chunk position lane & 7 of the operand tile holds logical chunk c8 = pos ^ key: hi (c8 < 4) or lo' (c8 >= 4)
                // halves of the channels 8 (c8 & 3) .. + 7 of the step
                const int c8 = (lane & 7) ^ ((r >> 1) & 7);
                const int ch = ci0 + 8 * (c8 & 3);
                const bool ok = src_row >= 0 && src_row < rows && ch < p.cin;
                const _Float16 *src = pb + (long long)src_row * (2 * p.h_split_ld) + (c8 >> 2) * p.h_split_ld + ch;
                gh_lds_dma16(ok ? reinterpret_cast<const float *>(src) : p.zeros, adst + 1024u * (unsigned)piece);
            } else {
                const int ch = ci0 + 4 * (lane & 7);               // 4-channel chunk lane & 7 of the step
                const bool ok = src_row >= 0 && src_row < rows && ch < p.cin;
                gh_lds_dma16(ok ? xb + (long long)src_row * p.ldx + ch : p.zeros, adst + 1024u * (unsigned)piece);
            }
        }
    };
    auto issue_b = [&](int kt, int stage) {
        const unsigned bdst = lds_base + 4u * (unsigned)(GH_B + stage * GH_B_FLOATS);
        const float *bsrc = wtile + (long long)kt * GH_B_FLOATS;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 8 * i;
            gh_lds_dma16_s(bsrc + piece * 256, b_voff, bdst + 1024u * (unsigned)piece);
        }
    };
    // ---- conditioning rows of this block (28 x (32 tanh | 32 sigmoid) columns), requested first
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    if (wave < GH_COND_ROWS / 4) {
        const int n2 = rows / cond_up;
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
        const int pos = wave * 64 + lane;
        const int crow = pos >> 4, cq = pos & 15;
        const int chn = n0 + 4 * (cq & 7);
        const int t = min(t2base + crow, n2 - 1);
        gh_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + (cq >> 3) * C + chn : p.zeros,
                     lds_base + 4u * (unsigned)GH_COND + 1024u * (unsigned)wave);
    }
    issue_a(0, 0);
    issue_b(0, 0);

    // lane n of column tile (e, tanh | sigmoid) holds gate channel n0 + 2 n + e
    const bool ch_ok = n0 + 2 * r16 < C;
    f32x4 accm[2][4], accx[2][4];          // [row tile][column tile: 2 e + (0 tanh | 1 sigmoid)]: hi hi | cross products
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float bv = (p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                accm[rt][c][v] = bv;
                accx[rt][c][v] = 0.f;
            }
        }
    if (tid < GH_ROWS) {
        // block row lr = tid: conditioning row offset and interpolation phase (read in the epilogue)
        const int row = m0 + tid;
        const int t2 = row / cond_up;
        const int u = row - t2 * cond_up;
        reinterpret_cast<int *>(lds + GH_TAB)[tid] = (((t2 - t2base) * 64) << 8) | u;
    }
    if (tid < 64) {
        lds[GH_LERP + tid] = tid < cond_up ? p.lerp_w0[tid] : 0.f;
        lds[GH_LERP + 64 + tid] = tid < cond_up ? p.lerp_w1[tid] : 0.f;
    }

    const f16x8 *bbase = reinterpret_cast<const f16x8 *>(lds) + GH_B / 4 + lane;   // + stage * (GH_B_FLOATS / 4) + ((tap * 4 + tile) * 2 + image) * 64
    const f32x4 *xs = reinterpret_cast<const f32x4 *>(lds + GH_X);
    char *ys = reinterpret_cast<char *>(lds + GH_Y);
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        // every request this wave has in flight belongs to step kt (and, in step 0, to the conditioning tile)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // this step's operands are complete; nobody reads the other stages any more
        if (PLANES) {
            if (kt + 1 < nk) {
                issue_a(kt + 1, stage ^ 1);
                issue_b(kt + 1, stage ^ 1);
            }
            ys = reinterpret_cast<char *>(lds + (stage ? GH_X : GH_Y));
        } else {
            // the next step's weights can go to the other stage at once (its readers passed the barrier)
            if (kt + 1 < nk) issue_b(kt + 1, stage ^ 1);
            // ---- X -> Y: thread t converts the 16-byte chunks t, t + 512, ... (chunk q: staged row q >> 3, channels 4 (q & 7) ..);
            // all reads first, so that their latencies overlap
            f32x4 xv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int q = tid + 512 * i;
                xv[i] = xs[q < GH_AROWS * 8 ? q : tid];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int q = tid + 512 * i;
                if (q < GH_AROWS * 8) {
                    const int r = q >> 3, c4 = q & 7;
                    f16x4 hh, ll;
                    gh_split4(xv[i], hh, ll);
                    const int key = (r >> 1) & 7;
                    char *yr = ys + 128 * r + 8 * (c4 & 1);
                    *reinterpret_cast<f16x4 *>(yr + 16 * ((c4 >> 1) ^ key)) = hh;
                    *reinterpret_cast<f16x4 *>(yr + 16 * ((4 + (c4 >> 1)) ^ key)) = ll;
                }
            }
            __syncthreads();                   // Y is complete, X is free
            if (kt + 1 < nk) issue_a(kt + 1, 0);
        }
        const f16x8 *bs = bbase + stage * (GH_B_FLOATS / 4);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bh[c] = bs[((tap * 4 + c) * 2 + 0) * 64];
                bl[c] = bs[((tap * 4 + c) * 2 + 1) * 64];
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                // staged row of this lane's operand: output row 32 wave + 16 rt + r16, input row + (tap - 1) d, + halo
                const int r = 32 * wave + 16 * rt + r16 + GH_HALO + (tap - 1) * d;
                const int key = (r >> 1) & 7;
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(ys + 128 * r + 16 * (kq ^ key));
                const f16x8 al = *reinterpret_cast<const f16x8 *>(ys + 128 * r + 16 * ((4 + kq) ^ key));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    accm[rt][c] = GH_MFMA(ah, bh[c], accm[rt][c]);
                    accx[rt][c] = GH_MFMA(ah, bl[c], accx[rt][c]);
                    accx[rt][c] = GH_MFMA(al, bh[c], accx[rt][c]);
                }
            }
        }
    }

    // ---- epilogue: main + 2^-11 cross, conditioning, gate, store (layout of wn_gate_winograd4w_kernel: register v of
    // column tile c holds row 16 rt + 4 kq + v of the wave's rows, lane n = r16 the gate channels n0 + 2 n + e)
    const float *cl = lds + GH_COND;
    float *obase = p.out + (long long)b * p.out_bstride + n0 + 2 * r16;
    const float *clane = cl + 2 * r16;
    // (round 5, as in the float32 gate kernels: table entries first, then the conditioning reads of a row tile together, results
    // formed outside the store branches -- every output used to be a region of its own with its own LDS round trips)
    int etab[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) etab[rt][v] = reinterpret_cast<const int *>(lds + GH_TAB)[32 * wave + 16 * rt + 4 * kq + v];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        float2 w[4], ct0[4], ct1[4], cs0[4], cs1[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int e = etab[rt][v];
            w[v] = make_float2(lds[GH_LERP + (e & 255)], lds[GH_LERP + 64 + (e & 255)]);
            const float *c0 = clane + (e >> 8);
            ct0[v] = *reinterpret_cast<const float2 *>(c0);
            ct1[v] = *reinterpret_cast<const float2 *>(c0 + 64);
            cs0[v] = *reinterpret_cast<const float2 *>(c0 + 32);
            cs1[v] = *reinterpret_cast<const float2 *>(c0 + 96);
        }
        float2 res[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = fmaf(accx[rt][c][v], 1.0f / 2048.0f, accm[rt][c][v]);
            res[v].x = wn_gate_act(p.gate_act, y[0] + fmaf(ct0[v].x, w[v].x, ct1[v].x * w[v].y), y[1] + fmaf(cs0[v].x, w[v].x, cs1[v].x * w[v].y));
            res[v].y = wn_gate_act(p.gate_act, y[2] + fmaf(ct0[v].y, w[v].x, ct1[v].y * w[v].y), y[3] + fmaf(cs0[v].y, w[v].x, cs1[v].y * w[v].y));
            asm volatile("" : "+v"(res[v].x), "+v"(res[v].y));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = m0 + 32 * wave + 16 * rt + 4 * kq + v;
            if (ch_ok && row < rows) *reinterpret_cast<float2 *>(obase + (long long)row * p.ldo) = res[v];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the plane-fed kernel with 256 x 128 tiles.  What the ablations of wn_gate_f16_kernel<true> said at 16 x 10 s (946 us
// per launch, profiles/r06_gate_f16_ablations.txt): no LDS operand reads at all 953 us (NOT the bound, against round 4's
// reading of the first version), a sixth of the MFMAs 740, no LDS-DMA 727, no epilogue 859, no barrier 916; launch time
// against the K steps per block (C = 160 / 224 / 320): 6.9 us fixed per block + 1.80 us per K step, where the matrix pipe
// needs 0.96 us -- a step lasts as long as the round trip of its 60 KB of operands (256 CUs x 60 KB / 1.8 us = 8.5 TB/s out
// of the L2s), one step of look-ahead being all the LDS holds.  (Persistent blocks that request the next tile's first
// operands during the last step of the current one measured the same 929 us: the fixed part is not the tile change.)  So the
// tile has to do more arithmetic per byte it stages: here a block owns TWO column tiles (128 weight columns) of its 256 rows
// -- the activation rows are staged once for both: 84 KB per step for twice the MFMAs (42 KB per 64 columns instead of 60)
// and the fixed part once per two tiles.  Wave w: rows 64 (w & 3) .. + 63 (four 16-row tiles) x column tile w >> 2: 4 x 4
// tiles x 2 accumulator sets = 128 registers; per tap 8 + 8 operand reads for 48 MFMAs (0.33 KB per MFMA instead of 0.5).
// LDS: two activation stages (72 KB) + the weights as a RING of four tap chunks of 16 KB (a step's 48 KB twice would not
// fit): chunk g = (step, tap) lives in slot g & 3 and is requested three taps ahead, behind the barrier in front of tap
// g - 3 (every wave is past tap g - 4, the slot's last reader); the activation rows of step s + 1 are requested at tap 0 of
// step s.  Requests complete in order, so "chunk g has landed" = at most the requests issued behind it outstanding:
// s_waitcnt vmcnt(4) in front of tap 0, vmcnt(9) in front of taps 1 and 2 (5 activation + 2 weight requests per wave and
// slot), vmcnt(2) / vmcnt(0) in the last step, where nothing is requested any more.  Same products, same order of accumulation per output as wn_gate_f16_kernel: same bits.
constexpr int GW_A = 0;                                     // two operand stages of GH_A_FLOATS
constexpr int GW_B = 2 * GH_A_FLOATS;                       // ring of four tap chunks
constexpr int GW_CHUNK = 2 * 2048;                          // one tap, two column tiles: 16 KB
constexpr int GW_COND = GW_B + 4 * GW_CHUNK;                // 28 conditioning rows x 128 floats
constexpr int GW_TAB = GW_COND + GH_COND_ROWS * 128;
constexpr int GW_LERP = GW_TAB + GH_ROWS;
constexpr int GW_LDS_FLOATS = GW_LERP + 128;                // 18432 + 16384 + 3584 + 256 + 128 floats = 151.5 KB

__global__ __launch_bounds__(512, 1) void wn_gate_f16w_kernel(ConvArgs p, int log2d) {
    typedef __attribute__((address_space(3))) float lds_float;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float *)lds);

    // XCD-aware decode; p.n_tiles counts PAIRS of column tiles here
    const int id = blockIdx.x;
    const int l = id >> 3;
    const int g_ = (l / p.n_tiles) * 8 + (id & 7);
    const int np = l % p.n_tiles;
    if (g_ >= p.m_tiles_total) return;
    const int b = g_ / p.m_tiles_per_item;
    const int mt = g_ - b * p.m_tiles_per_item;
    const int rows = item_rows(p.n_frames, b, p.rows_per_frame, p.max_rows);
    const int m0 = mt * GH_ROWS;
    if (m0 >= rows) return;
    const int C = p.channels;
    const int d = 1 << log2d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave & 3, cw = wave >> 2;                  // row quarter and column tile of this wave
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = (p.cin + GH_BK - 1) / GH_BK;
    const int n_ct = (C + 31) / 32;                           // column tiles of the layer
    const int nt = 2 * np + cw;                               // this wave's column tile (may not exist: C / 32 odd)
    const int n0 = nt * 32;

    // ---- per-lane sources of the activation requests (fixed for the block except the channel offset): piece j = staged rows
    // 8 j .. 8 j + 7 (lane -> row 8 j + (lane >> 3), chunk position lane & 7); 36 pieces, five per wave (32 .. 35 twice)
    const _Float16 *pb = reinterpret_cast<const _Float16 *>(p.h_split + (long long)b * p.h_split_bstride);
    const _Float16 *asrc[5];
    int achl[5];
    bool arow_ok[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int piece = wave + 8 * i;
        if (piece >= GH_AROWS / 8) piece -= 8;                // (the duplicate requests keep the counts of all waves equal)
        const int r = 8 * piece + (lane >> 3);
        const int src_row = m0 - GH_HALO + r;
        const int c8 = (lane & 7) ^ ((r >> 1) & 7);
        achl[i] = 8 * (c8 & 3);
        arow_ok[i] = src_row >= 0 && src_row < rows;
        asrc[i] = pb + (long long)min(max(src_row, 0), rows - 1) * (2 * p.h_split_ld) + (c8 >> 2) * p.h_split_ld + achl[i];
    }
    auto issue_a = [&](int kt) {
        const unsigned adst = lds_base + 4u * (unsigned)(GW_A + (kt & 1) * GH_A_FLOATS);
        const int ci0 = kt * GH_BK;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int piece = wave + 8 * i;
            if (piece >= GH_AROWS / 8) piece -= 8;
            const bool ok = arow_ok[i] && ci0 + achl[i] < p.cin;
            gh_lds_dma16(ok ? reinterpret_cast<const float *>(asrc[i] + ci0) : p.zeros, adst + 1024u * (unsigned)piece);
        }
    };
    // tap chunk g = 3 kt + tap of both column tiles -> slot g & 3: piece `wave` of each tile's 8 KB
    const unsigned b_voff = 16u * (unsigned)lane;
    auto issue_b = [&](int g) {
        const int kt = g / 3, tap = g - 3 * kt;
        const unsigned bdst = lds_base + 4u * (unsigned)(GW_B + (g & 3) * GW_CHUNK);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ct = 2 * np + t;
            const float *bsrc = p.w + ((long long)min(ct, n_ct - 1) * nk + kt) * GH_B_FLOATS + tap * 2048 + wave * 256;
            gh_lds_dma16_s(bsrc, b_voff, bdst + 4u * (unsigned)(t * 2048) + 1024u * (unsigned)wave);
        }
    };
    // ---- conditioning rows of this block: 28 x 2 tiles x (32 tanh | 32 sigmoid) columns = 14 pieces
    const int cond_up = p.cond_up;
    const int t2base = m0 / cond_up;
    {
        const int n2 = rows / cond_up;
        const float *cbase = p.cond + (long long)b * p.cond_bstride;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 8 * i;
            if (piece >= GH_COND_ROWS / 2) break;
            const int pos = piece * 64 + lane;
            const int crow = pos >> 5, cq = pos & 31;
            const int chn = (2 * np + (cq >> 4)) * 32 + 4 * (cq & 7);
            const int t = min(t2base + crow, n2 - 1);
            gh_lds_dma16(chn < C ? cbase + (long long)t * (2 * C) + ((cq >> 3) & 1) * C + chn : p.zeros,
                         lds_base + 4u * (unsigned)GW_COND + 1024u * (unsigned)piece);
        }
    }
    issue_a(0);
    issue_b(0);
    issue_b(1);
    issue_b(2);

    const bool ch_ok = n0 + 2 * r16 < C;
    f32x4 accm[4][4], accx[4][4];          // [row tile][column tile: 2 e + (0 tanh | 1 sigmoid)]: hi hi | cross products
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float bv = (p.bias && ch_ok) ? p.bias[(c & 1) * C + n0 + 2 * r16 + (c >> 1)] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                accm[rt][c][v] = bv;
                accx[rt][c][v] = 0.f;
            }
    }
    if (tid < GH_ROWS) {
        const int row = m0 + tid;
        const int t2 = row / cond_up;
        const int u = row - t2 * cond_up;
        reinterpret_cast<int *>(lds + GW_TAB)[tid] = (((t2 - t2base) * 128) << 8) | u;
    }
    if (tid < 64) {
        lds[GW_LERP + tid] = tid < cond_up ? p.lerp_w0[tid] : 0.f;
        lds[GW_LERP + 64 + tid] = tid < cond_up ? p.lerp_w1[tid] : 0.f;
    }

    const f16x8 *bring = reinterpret_cast<const f16x8 *>(lds) + GW_B / 4 + cw * 512 + lane;     // + slot * (GW_CHUNK / 4) + (c * 2 + image) * 64
    const int ntap = 3 * nk;
    for (int kt = 0; kt < nk; ++kt) {
        const char *ys = reinterpret_cast<const char *>(lds + GW_A + (kt & 1) * GH_A_FLOATS);
        const bool last = kt + 1 == nk;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int g = 3 * kt + tap;
            // chunk g (and, at tap 0, the rows of this step) have landed when at most the requests issued behind them are out
            // (last step: nothing is requested any more, the counts run out)
            if (tap == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (!last) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (tap == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                   // ... for every wave; and every wave is past tap g - 1
            if (tap == 0 && !last) issue_a(kt + 1);
            if (g + 3 < ntap) issue_b(g + 3);
            const f16x8 *bs = bring + (g & 3) * (GW_CHUNK / 4);
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bh[c] = bs[(c * 2 + 0) * 64];
                bl[c] = bs[(c * 2 + 1) * 64];
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int r = 64 * rw + 16 * rt + r16 + GH_HALO + (tap - 1) * d;
                const int key = (r >> 1) & 7;
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(ys + 128 * r + 16 * (kq ^ key));
                const f16x8 al = *reinterpret_cast<const f16x8 *>(ys + 128 * r + 16 * ((4 + kq) ^ key));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    accm[rt][c] = GH_MFMA(ah, bh[c], accm[rt][c]);
                    accx[rt][c] = GH_MFMA(ah, bl[c], accx[rt][c]);
                    accx[rt][c] = GH_MFMA(al, bh[c], accx[rt][c]);
                }
            }
        }
    }

    // ---- epilogue (as in wn_gate_f16_kernel): main + 2^-11 cross, conditioning, gate, store
    float *obase = p.out + (long long)b * p.out_bstride + n0 + 2 * r16;
    const float *clane = lds + GW_COND + cw * 64 + 2 * r16;
    int etab[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) etab[rt][v] = reinterpret_cast<const int *>(lds + GW_TAB)[64 * rw + 16 * rt + 4 * kq + v];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        float2 w[4], ct0[4], ct1[4], cs0[4], cs1[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int e = etab[rt][v];
            w[v] = make_float2(lds[GW_LERP + (e & 255)], lds[GW_LERP + 64 + (e & 255)]);
            const float *c0 = clane + (e >> 8);
            ct0[v] = *reinterpret_cast<const float2 *>(c0);
            ct1[v] = *reinterpret_cast<const float2 *>(c0 + 128);
            cs0[v] = *reinterpret_cast<const float2 *>(c0 + 32);
            cs1[v] = *reinterpret_cast<const float2 *>(c0 + 160);
        }
        float2 res[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = fmaf(accx[rt][c][v], 1.0f / 2048.0f, accm[rt][c][v]);
            res[v].x = wn_gate_act(p.gate_act, y[0] + fmaf(ct0[v].x, w[v].x, ct1[v].x * w[v].y), y[1] + fmaf(cs0[v].x, w[v].x, cs1[v].x * w[v].y));
            res[v].y = wn_gate_act(p.gate_act, y[2] + fmaf(ct0[v].y, w[v].x, ct1[v].y * w[v].y), y[3] + fmaf(cs0[v].y, w[v].x, cs1[v].y * w[v].y));
            asm volatile("" : "+v"(res[v].x), "+v"(res[v].y));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = m0 + 64 * rw + 16 * rt + 4 * kq + v;
            if (ch_ok && row < rows) *reinterpret_cast<float2 *>(obase + (long long)row * p.ldo) = res[v];
        }
    }
}

// a.w must point at the image of engine.pack_gate_f16_weights (ceil(C/32) column tiles, ceil(C/32) steps, 6144 floats);
// returns false if the layer does not fit (the caller then runs the float32 kernels)
bool launch_wn_gate_f16(const ConvArgs &a, hipStream_t stream) {
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const bool ok = a.ks == 3 && (1 << log2d) == a.dil && a.dil <= GH_HALO && a.pad_l == a.dil && a.pad_mode == 0 &&
                    a.cin % 4 == 0 && a.ldx % 4 == 0 && a.x_bstride % 4 == 0 && a.channels % 4 == 0 && a.cin == a.channels &&
                    a.cout == 2 * a.channels && (uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && a.zeros && a.cond &&
                    (uintptr_t)a.cond % 16 == 0 && a.cond_bstride % 4 == 0 && a.cond_up >= 1 && a.cond_up <= 64 &&
                    (GH_ROWS + a.cond_up - 2) / a.cond_up + 2 <= GH_COND_ROWS && a.lerp_w0 && a.lerp_w1 &&
                    a.cond_phase == 0 && a.out_rows == 0 && a.max_rows < (1 << 24) &&
                    (!a.h_split || (a.h_split_ld % 8 == 0 && a.h_split_ld >= a.channels && a.h_split_bstride % 4 == 0 &&
                                    (uintptr_t)a.h_split % 16 == 0));
    if (!ok) return false;
    // the attribute belongs to the (function, device) pair: a process may hold handles on several devices
    static unsigned long long attr_devices = 0;       // bit d: set for device d (devices >= 64: set at every launch)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool attr_set = dev >= 0 && dev < 64 && ((attr_devices >> dev) & 1ull);
    static bool gw_ok = false;
    if (!attr_set) {
        gw_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(wn_gate_f16w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    GW_LDS_FLOATS * (int)sizeof(float)) == hipSuccess;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(wn_gate_f16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                GH_LDS_FLOATS * (int)sizeof(float)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(wn_gate_f16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                GH_LDS_FLOATS * (int)sizeof(float)) != hipSuccess)
            return false;
        if (dev >= 0 && dev < 64) attr_devices |= 1ull << dev;
    }
    ConvArgs r = a;
    r.n_tiles = (a.channels + 31) / 32;
    r.m_tiles_per_item = (a.max_rows + GH_ROWS - 1) / GH_ROWS;
    r.m_tiles_total = r.m_tiles_per_item * a.batch;
    const long long blocks = 8LL * ((r.m_tiles_total + 7) / 8) * r.n_tiles;
    // 256 x 128 tiles (pairs of column tiles) for launches of several rounds of blocks: 0.83 against 0.93 ms at 16 x 10 s; one
    // utterance of 10 s is 315 such blocks on 256 CUs and keeps the 256 x 64 tiles (72.5 against 78.7 us).  Same bits.
    const long long blocks2 = 8LL * ((r.m_tiles_total + 7) / 8) * ((r.n_tiles + 1) / 2);
    if (a.h_split && gw_ok && (a.cin + GH_BK - 1) / GH_BK >= 2 && blocks2 >= 4 * 256) {
        r.n_tiles = (r.n_tiles + 1) / 2;
        hipLaunchKernelGGL(wn_gate_f16w_kernel, dim3((unsigned)blocks2), dim3(512), GW_LDS_FLOATS * sizeof(float), stream, r, log2d);
    } else if (a.h_split)
        hipLaunchKernelGGL(wn_gate_f16_kernel<true>, dim3((unsigned)blocks), dim3(512), GH_LDS_FLOATS * sizeof(float), stream, r, log2d);
    else
        hipLaunchKernelGGL(wn_gate_f16_kernel<false>, dim3((unsigned)blocks), dim3(512), GH_LDS_FLOATS * sizeof(float), stream, r, log2d);
    return true;
}

}  // namespace mbx
