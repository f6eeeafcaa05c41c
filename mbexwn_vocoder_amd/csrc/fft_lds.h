// Real FFTs of up to 2048 points inside one 256-thread block: complex Stockham autosort in LDS (radix-4 passes, one
// radix-2 pass when log2 is odd) + the split step of a real transform.  Shared by the STFT-domain filter
// (stft_filter.hip) and the audio -> mel analysis (mel_analysis.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mbx {

constexpr int FFT_THREADS = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// exp(-+ 2 pi i idx / (2 nc)) from the half-circle table tw[0..nc), idx < 2 nc
template <bool INVERSE>
__device__ __forceinline__ float2 twiddle(const float2 *tw, int idx, int nc) {
    float2 w = tw[idx & (nc - 1)];
    if (idx >= nc) w = make_float2(-w.x, -w.y);
    if (INVERSE) w.y = -w.y;
    return w;
}

// nc complex points ping-pong between a and b; returns the buffer holding the (un-normalised) result
template <bool INVERSE>
__device__ float2 *fft_lds(float2 *a, float2 *b, const float2 *tw, int nc, int tid) {
    float2 *in = a, *out = b;
    int ns = 1;
    const int quarter = nc >> 2;
    for (; ns * 4 <= nc; ns <<= 2) {
        const int step = nc / (2 * ns);
        for (int j = tid; j < quarter; j += FFT_THREADS) {
            const int k = j & (ns - 1);
            const float2 u0 = in[j];
            float2 u1 = in[j + quarter], u2 = in[j + 2 * quarter], u3 = in[j + 3 * quarter];
            if (ns > 1) {
                u1 = cmul(u1, twiddle<INVERSE>(tw, k * step, nc));
                u2 = cmul(u2, twiddle<INVERSE>(tw, 2 * k * step, nc));
                u3 = cmul(u3, twiddle<INVERSE>(tw, 3 * k * step, nc));
            }
            const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), t = csub(u1, u3);
            const float2 v3 = INVERSE ? make_float2(-t.y, t.x) : make_float2(t.y, -t.x);    // t * (+-i)
            const int j0 = ((j - k) << 2) + k;
            out[j0] = cadd(v0, v2);
            out[j0 + ns] = cadd(v1, v3);
            out[j0 + 2 * ns] = csub(v0, v2);
            out[j0 + 3 * ns] = csub(v1, v3);
        }
        __syncthreads();
        float2 *tmp = in;
        in = out;
        out = tmp;
    }
    if (ns < nc) {      // ns * 2 == nc: one radix-2 pass
        const int half = nc >> 1;
        for (int j = tid; j < half; j += FFT_THREADS) {
            const int k = j & (ns - 1);
            const float2 v0 = in[j];
            const float2 t = cmul(in[j + half], twiddle<INVERSE>(tw, k * (nc / ns), nc));
            const int j0 = ((j - k) << 1) + k;
            out[j0] = cadd(v0, t);
            out[j0 + ns] = csub(v0, t);
        }
        __syncthreads();
        float2 *tmp = in;
        in = out;
        out = tmp;
    }
    return in;
}

// bin k (0 <= k <= nc) of the real transform of x from Z = FFT_nc(x[2m] + i x[2m+1])
__device__ __forceinline__ float2 real_bin(const float2 *z, const float2 *tw, int k, int nc) {
    const float2 zk = z[k & (nc - 1)];
    const float2 zr = z[(nc - k) & (nc - 1)];
    const float2 xe = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));       // (Z[k] + conj Z[nc-k]) / 2
    const float2 xo = make_float2(0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x));      // (Z[k] - conj Z[nc-k]) / 2i
    const float2 w = k < nc ? tw[k] : make_float2(-1.f, 0.f);
    return cadd(xe, cmul(w, xo));
}

}  // namespace mbx
