// Wavetable oscillator: F0 -> wrapped phase -> band-limited LF pulse (+ optional sub-harmonic sinusoid channels).
//
// restates PulseWaveTable.call                 reference MBExWN_NVoc/vocoder/model/tf_wavetable.py:495-552
//          stable_cumsum_and_wrap               reference tf_wavetable.py:429-492
//          _linear_lookup                       reference tf_wavetable.py:605-638
//
// The phase is a float32 running sum; a parallel scan would round differently and the error is
// amplified by the table slope, so the reference's order of additions is kept bit for bit:
//   (1) inside each chunk of 1000 samples a sequential float32 running sum (one lane per chunk does the
//       adds out of LDS -- chunks are independent, so a 10 s utterance still exposes 80 x batch chains),
//   (2) chunk offsets = running sum over chunks of (last value of the previous chunk mod 1), then mod 1,
//   (3) phase = (chunk sum + offset) mod 1, table lookup and grid mix: one thread per sample, coalesced.
#include "mbx_kernels.h"

namespace mbx {

__device__ __forceinline__ float mod1(float x) { return x - floorf(x); }   // x >= 0: identical to fmod(x, 1)

// (1) one wavefront per (item, chunk).  The running sum of a chunk must be formed in the reference's order, one float32 add
// after the other (tf.cumsum), so the chain of up to 1 000 dependent adds is the floor of this stage.  Round 4: the chain runs
// ACROSS THE LANES of the wave instead of in lane 0: lane l of group g holds sample 64 g + l (coalesced load, phase velocity);
// the group starts with acc = carry + x in every lane (final in lane 0), and 63 times `v_add_f32_dpp acc, acc, x wave_shr:1`
// -- every lane l >= 1 takes the sum of lane l - 1 and adds its own sample; lane 0, whose source lane does not exist, is
// not written -- so that after step t lanes 0 .. t hold their final sums (a lane's later rewrites repeat the same add on
// the same final inputs).  One instruction per sample, nothing else on the chain: no LDS image, no per-sample moves; the
// carry into the next group is a v_readlane of lane 63 and the sums leave as one coalesced store per group.  Same adds in
// the same order on the same values as the lane-0 loop it replaces (17 us per launch for one 10 s utterance, ~20 cycles
// per add with its LDS reads / writes and moves).
constexpr int PHASE_MAX_CHUNK = 1024;   // multiple of 16

// Streaming: an item may start in the middle of a reference chunk.  `st` (optional) gives, per item, the window
// sample `start` where the carried state applies, the position `pos` of that sample inside its 1000-sample chunk
// and the running sum `cum` reached just before it; chunk 0 of the window is then the remainder of that chunk.
__device__ __forceinline__ void chunk_range(const StreamState *st, int b, int chunk, int c, int n, int &begin, int &end,
                                            float &acc0) {
    const int start = st ? st[b].start_sample : 0;
    const int first_len = st ? chunk - st[b].pos_in_chunk : chunk;
    if (c == 0) {
        begin = start;
        end = min(start + first_len, n);
        acc0 = st ? st[b].cum : 0.f;
    } else {
        begin = start + first_len + (c - 1) * chunk;
        end = min(begin + chunk, n);
        acc0 = 0.f;
    }
}

__global__ __launch_bounds__(64) void phase_chunk_kernel(const float *__restrict__ f0, long long bstride,
                                                          const int *__restrict__ n_frames, int samples_per_frame,
                                                          int n_max, int chunk, float pulse_rate,
                                                          float *__restrict__ cum, float *__restrict__ chunk_last,
                                                          int chunks_max, const StreamState *__restrict__ st) {
    const int b = blockIdx.y;
    const int c = blockIdx.x;
    const int n = item_rows(n_frames, b, samples_per_frame, n_max);
    int begin, end;
    float acc0;
    chunk_range(st, b, chunk, c, n, begin, end, acc0);
    if (begin >= end) return;
    const float *fb = f0 + (long long)b * bstride;
    float *cb = cum + (long long)b * bstride;
    const int len = end - begin;
    const int lane = threadIdx.x;
    constexpr int G = PHASE_MAX_CHUNK / 64;
    // phase velocity = frequency / sample_rate (tf_wavetable.py:516); the reference zero-pads the last chunk (adding the
    // padding zeros leaves the sum unchanged)
    float x[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // (clamped address, masked afterwards: as `i < len ? load : 0` every group's load sits in a branch with a wait of its own)
        const int i = 64 * g + lane;
        x[g] = fb[begin + min(i, len - 1)];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) x[g] = 64 * g + lane < len ? x[g] / pulse_rate : 0.f;
    float carry = acc0;                                      // wave-uniform
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (64 * g < len) {                                  // wave-uniform: all 64 lanes are active inside
            float acc = carry + x[g];
            asm volatile(".rept 63\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr\n\ts_nop 4"
                         : "+v"(acc)
                         : "v"(x[g]));
            carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
            const int i = 64 * g + lane;
            if (i < len) cb[begin + i] = acc;
        }
    }
    if (lane == 0) chunk_last[(long long)b * chunks_max + c] = carry;
}

// (2)+(3) one thread per sample.  The offset of chunk c = (sum_{j < c} (last_j mod 1)) mod 1, summed in chunk order
// (tf_wavetable.py:476-483), is the same for every sample of the chunk: each block forms the offsets of the item's chunks
// once (one lane walks the chain out of LDS; <= WT_MAX_CHUNKS chunks, i.e. 128 s of audio at the 8 kHz pulse rate --
// longer items sum per sample) instead of every sample walking up to its chunk.
constexpr int WT_MAX_CHUNKS = 1024;

__global__ void wavetable_kernel(WaveTableConsts k, const float *f0, long long bstride, const int *n_frames,
                                 int samples_per_frame, int n_max, const float *cum, const float *chunk_last,
                                 int chunks_max, float *pulse, float *phase_out, const StreamState *st) {
    __shared__ float offs[WT_MAX_CHUNKS];
    const int b = blockIdx.y;
    const int n = item_rows(n_frames, b, samples_per_frame, n_max);
    const float *fb = f0 + (long long)b * bstride;
    const float *cb = cum + (long long)b * bstride;
    const float *lb = chunk_last + (long long)b * chunks_max;
    const int nch = 1 + k.n_sub;                              // channels per sample: pulse, then the sub-harmonic sinusoids
    float *pb = pulse + (long long)b * bstride * nch;
    const int start = st ? st[b].start_sample : 0;
    const int first_len = st ? k.chunk - st[b].pos_in_chunk : k.chunk;
    const float off0 = st ? st[b].offset_sum : 0.f;
    // chunks this block's samples can fall into: 0 .. c_hi
    const int i_last = min(n, n_max) - 1;
    const int c_hi = i_last < start + first_len ? 0 : 1 + (i_last - start - first_len) / k.chunk;
    const bool table = c_hi < WT_MAX_CHUNKS;
    if (table) {
        for (int j = threadIdx.x; j < c_hi; j += blockDim.x) offs[j + 1] = mod1(lb[j]);
        __syncthreads();
        if (threadIdx.x == 0) {
            float off = off0;
            offs[0] = mod1(off);
            for (int j = 1; j <= c_hi; ++j) {
                off = off + offs[j];
                offs[j] = mod1(off);
            }
        }
        __syncthreads();
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (i < start) {          // streaming: samples in front of the carried state are not reproducible
            for (int ch = 0; ch < nch; ++ch) pb[(long long)i * nch + ch] = 0.f;
            if (phase_out) phase_out[(long long)b * bstride + i] = 0.f;
            continue;
        }
        const int rel = i - start;
        const int c = rel < first_len ? 0 : 1 + (rel - first_len) / k.chunk;
        // both per-sample inputs requested together (behind the branches below the compiler asks for f0 only when the phase
        // has arrived: one more round trip in front of the table look-up)
        const float cum_i = cb[i], f = fb[i];
        float off;
        if (table) {
            off = offs[c];
        } else {
            off = off0;
            for (int j = 0; j < c; ++j) off = off + mod1(lb[j]);
            off = mod1(off);
        }
        const float phase = mod1(cum_i + off);
        if (phase_out) phase_out[(long long)b * bstride + i] = phase;
        // wrapped_phase * 2 * pi in float32, left to right (tf_wavetable.py:521)
        const float w2pi = (phase * 2.f) * 3.14159265358979323846f;
        for (int ii = 2; ii < nch + 1; ++ii) pb[(long long)i * nch + ii - 1] = sinf(w2pi / (float)ii);   // :554-557
        if (k.sin_fun) {          // :522-523
            pb[(long long)i * nch] = (sinf(w2pi) * 0.5f) * (1.f - cosf(w2pi));
            continue;
        }
        // linear table lookup (tf_wavetable.py:619-638)
        const float pos = phase * (float)k.n_period;
        const float base = floorf(pos);
        const float rem = pos - base;
        const int idx = (int)base;
        // grid mix (tf_wavetable.py:539-548): q = ln(clip(f0 / nominalF0)) / ln(grid); weights max(1-|q-r|, 0)
        const float ratio = fmaxf(k.min_tf, fminf(k.max_tf, f / k.nominal_f0));
        const float q = logf(ratio) * k.grid_norm;
        int r0 = (int)floorf(q);
        r0 = max(0, min(r0, k.n_tables - 1));
        const int r1 = min(r0 + 1, k.n_tables - 1);
        const float *t0 = k.tables + (long long)idx * k.n_tables;
        const float *t1 = t0 + k.n_tables;
        const float one_m = 1.0f - rem;
        const float s0 = t0[r0] * one_m + t1[r0] * rem;
        const float w0 = fmaxf(1.0f - fabsf(q - (float)r0), 0.f);
        float out = s0 * w0;
        if (r1 != r0) {
            const float s1 = t0[r1] * one_m + t1[r1] * rem;
            const float w1 = fmaxf(1.0f - fabsf(q - (float)r1), 0.f);
            out = out + s1 * w1;
        }
        pb[(long long)i * nch] = out;
    }
}

// state of the accumulator just in front of window sample `save_sample` (one thread per item)
__global__ void phase_state_kernel(int chunk, const float *cum, long long bstride, const float *chunk_last,
                                   int chunks_max, const StreamState *st, StreamState *out, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    StreamState o = st[b];
    const int save = st[b].save_sample;
    if (save >= st[b].start_sample) {
        const int rel = save - st[b].start_sample;
        const int first_len = chunk - st[b].pos_in_chunk;
        const int c = rel < first_len ? 0 : 1 + (rel - first_len) / chunk;
        const int pos = rel < first_len ? st[b].pos_in_chunk + rel : (rel - first_len) % chunk;
        float off = st[b].offset_sum;
        for (int j = 0; j < c; ++j) off = off + mod1(chunk_last[(long long)b * chunks_max + j]);
        o.cum = pos == 0 ? 0.f : (rel == 0 ? st[b].cum : cum[(long long)b * bstride + save - 1]);
        o.offset_sum = off;
        o.pos_in_chunk = pos;
    }
    out[b] = o;
}

void launch_wavetable(const WaveTableConsts &c, const float *f0, long long bstride, const int *n_frames,
                      int samples_per_frame, int n_max, int batch, float *pulse, float *phase_out, float *cum,
                      float *chunk_last, const StreamState *st_in, StreamState *st_out, hipStream_t stream) {
    if (n_max <= 0 || batch <= 0) return;
    const int chunks_max = (n_max + c.chunk - 1) / c.chunk + 1;
    hipLaunchKernelGGL(phase_chunk_kernel, dim3(chunks_max, batch), dim3(64), 0, stream, f0, bstride,
                       n_frames, samples_per_frame, n_max, c.chunk, c.pulse_rate, cum, chunk_last, chunks_max, st_in);
    // every block forms the chunk offsets of its item first (one lane, a chain of c_hi adds: ~3 us for a 10 s item): about one
    // resident round of blocks over the whole batch, each striding over its samples, instead of one block per 256 samples
    // (16 x 10 s: 15 008 blocks = 7 rounds of that chain -> 2 048 blocks)
    const int blocks = min((n_max + 255) / 256, max(1, 2048 / batch));
    hipLaunchKernelGGL(wavetable_kernel, dim3(blocks, batch), dim3(256), 0, stream, c, f0, bstride, n_frames,
                       samples_per_frame, n_max, cum, chunk_last, chunks_max, pulse, phase_out, st_in);
    if (st_in && st_out)
        hipLaunchKernelGGL(phase_state_kernel, dim3((batch + 63) / 64), dim3(64), 0, stream, c.chunk, cum, bstride,
                           chunk_last, chunks_max, st_in, st_out, batch);
}

}  // namespace mbx
