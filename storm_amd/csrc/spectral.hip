// STFT / iSTFT front and back end with the magnitude-compression transform fused in.
// Reference map in include/storm_hip.h.  n_fft = 510 is tiny (<= 0.3 GFLOP per utterance),
// so each frame is a direct real DFT out of LDS: one workgroup per frame, one thread per
// frequency bin (forward) / per output sample (inverse), twiddles from a [n_fft] table,
// products in fp32, sums in fp64.  Semantics are torch.stft / torch.istft with center=True,
// reflect padding, periodic Hann, onesided output, window-envelope normalisation.
#include "common.h"

namespace storm {

constexpr int MAX_NFFT = 1024;

__global__ void peak_abs_kernel(const float* __restrict__ wav, float* __restrict__ peak, long long L, long long stride,
                                const int* __restrict__ row_len) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    if (row_len) L = row_len[b];
    const float* p = wav + (long long)b * stride;
    float m = 0.f;
    for (long long i = threadIdx.x; i < L; i += blockDim.x) m = fmaxf(m, fabsf(p[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // (an all-zero utterance: the reference divides 0 / 0 and fills the whole sampler with NaN, model.py:281-284; here its
    // row stays zero - in a batch it would otherwise poison nothing but itself, silently)
    if (threadIdx.x == 0) peak[b] = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-20f);
}

__global__ void stft_kernel(const float* __restrict__ wav, const float* __restrict__ peak, float* __restrict__ spec,
                            const float* __restrict__ window, const float* __restrict__ tw, long long L,
                            long long stride, int n_fft, int hop, int n_frames, int Tpad, float factor, float expo, const int* __restrict__ row_len) {
    __shared__ float xs[MAX_NFFT];
    __shared__ float2 tws[MAX_NFFT];
    const int frame = blockIdx.x, b = blockIdx.y, F = n_fft / 2 + 1;
    float2* out = reinterpret_cast<float2*>(spec) + (long long)b * F * Tpad;
    if (row_len) { L = row_len[b]; n_frames = 1 + (int)(L / hop); }      // ragged batch: this row's own length / frame count
    if (frame >= n_frames) {                       // pad_spec: zero frames
        for (int f = threadIdx.x; f < F; f += blockDim.x) out[(long long)f * Tpad + frame] = make_float2(0.f, 0.f);
        return;
    }
    const float* x = wav + (long long)b * stride;
    const int pad = n_fft / 2;
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
        long long idx = (long long)frame * hop + k - pad;
        if (idx < 0) idx = -idx;                   // reflect (no edge repeat)
        if (idx >= L) idx = 2 * (L - 1) - idx;
        // a ragged row of <= n_fft / 2 samples is rejected by the host wrappers (torch.stft raises for it); a device-side
        // length the host never saw must still not read outside the row
        idx = idx < 0 ? 0 : (idx >= L ? L - 1 : idx);
        const float v = peak ? x[idx] / peak[b] : x[idx];    // y / norm_factor (model.py:284)
        xs[k] = v * window[k];
        tws[k] = reinterpret_cast<const float2*>(tw)[k];
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int k = 0; k < n_fft; ++k) {
            const float2 w = tws[idx];
            re += (double)(xs[k] * w.x);
            im -= (double)(xs[k] * w.y);
            idx += f; if (idx >= n_fft) idx -= n_fft;
        }
        float zr = (float)re, zi = (float)im;
        // spec_fwd: |X|^e exp(j angle X) * factor  (data_module.py:182-186)
        if (expo != 1.0f) {
            const float mag = sqrtf(zr * zr + zi * zi);
            const float sc = mag > 0.f ? powf(mag, expo - 1.0f) : 0.f;
            zr *= sc; zi *= sc;
        }
        out[(long long)f * Tpad + frame] = make_float2(zr * factor, zi * factor);
    }
}

__global__ void istft_frames_kernel(const float* __restrict__ spec, float* __restrict__ frames,
                                    const float* __restrict__ window, const float* __restrict__ tw, int T, int n_fft,
                                    float factor, float expo) {
    __shared__ float2 X[MAX_NFFT / 2 + 1];
    __shared__ float2 tws[MAX_NFFT];
    const int frame = blockIdx.x, b = blockIdx.y, F = n_fft / 2 + 1;
    const float2* in = reinterpret_cast<const float2*>(spec) + (long long)b * F * T;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float2 z = in[(long long)f * T + frame];
        z.x /= factor; z.y /= factor;                         // spec_back (data_module.py:188-193)
        if (expo != 1.0f) {
            const float mag = sqrtf(z.x * z.x + z.y * z.y);
            const float sc = mag > 0.f ? powf(mag, 1.0f / expo - 1.0f) : 0.f;
            z.x *= sc; z.y *= sc;
        }
        X[f] = z;
    }
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x) tws[k] = reinterpret_cast<const float2*>(tw)[k];
    __syncthreads();
    const bool even = (n_fft % 2) == 0;
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
        double acc = (double)X[0].x;
        int idx = 0;
        for (int f = 1; f < F; ++f) {
            idx += k; if (idx >= n_fft) idx -= n_fft;
            const float2 w = tws[idx];
            const double term = (double)(X[f].x * w.x) - (double)(X[f].y * w.y);   // Re(X e^{+j 2 pi f k / N})
            acc += (even && f == F - 1) ? term : 2.0 * term;
        }
        frames[((long long)b * T + frame) * n_fft + k] = (float)(acc / n_fft) * window[k];
    }
}

__global__ void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                 const float* __restrict__ peak, float* __restrict__ wav, int T, long long L,
                                 long long stride, int n_fft, int hop, const int* __restrict__ row_len) {
    const int b = blockIdx.y;
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= L) return;
    if (row_len && n >= row_len[b]) { wav[(long long)b * stride + n] = 0.f; return; }   // ragged batch: istft(..., length = this row's)
    const long long m = n + n_fft / 2;
    long long t1 = m / hop; if (t1 > T - 1) t1 = T - 1;
    long long t0 = (m - n_fft + hop) / hop; if (m - n_fft + 1 <= 0) t0 = 0; if (t0 < 0) t0 = 0;
    float y = 0.f, env = 0.f;
    for (long long t = t0; t <= t1; ++t) {
        const long long k = m - t * hop;
        if (k < 0 || k >= n_fft) continue;
        y += frames[((long long)b * T + t) * n_fft + k];
        env += window[k] * window[k];
    }
    float v = env > 1e-11f ? y / env : 0.f;
    if (peak) v *= peak[b];
    wav[(long long)b * stride + n] = v;
}

// standalone spec_fwd / spec_back on a complex tensor (data_module.py:182-193)
__global__ void spec_transform_kernel(const float* __restrict__ in, float* __restrict__ out, long long n,
                                      float factor, float expo, int inverse) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float2 z = reinterpret_cast<const float2*>(in)[i];
        if (inverse) { z.x /= factor; z.y /= factor; }
        if (expo != 1.0f) {
            const float mag = sqrtf(z.x * z.x + z.y * z.y);
            const float sc = mag > 0.f ? powf(mag, (inverse ? 1.0f / expo : expo) - 1.0f) : 0.f;
            z.x *= sc; z.y *= sc;
        }
        if (!inverse) { z.x *= factor; z.y *= factor; }
        reinterpret_cast<float2*>(out)[i] = z;
    }
}

}  // namespace storm

using namespace storm;

extern "C" int storm_spec_transform(const float* in, float* out, long long n_complex, float spec_factor,
                                    float spec_abs_exponent, int inverse, storm_stream_t s) {
    STORM_CHECK(in && out && n_complex > 0 && spec_factor != 0.f && spec_abs_exponent != 0.f, "storm_spec_transform: bad arguments");
    long long nb = (n_complex + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(spec_transform_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, in, out, n_complex,
                       spec_factor, spec_abs_exponent, inverse);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_peak_abs(const float* wav, float* peak, int B, long long L, long long stride, const int* row_len,
                              storm_stream_t s) {
    STORM_CHECK(wav && peak && B > 0 && L > 0, "storm_peak_abs: bad arguments");
    hipLaunchKernelGGL(peak_abs_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, wav, peak, L, stride, row_len);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_stft(const float* wav, const float* peak, float* spec, const float* window, const float* twiddle,
                          int B, long long L, long long stride, int n_fft, int hop, int n_frames, int Tpad,
                          float spec_factor, float spec_abs_exponent, const int* row_len, storm_stream_t s) {
    STORM_CHECK(wav && spec && window && twiddle && B > 0, "storm_stft: null pointer");
    STORM_CHECK(n_fft >= 2 && n_fft <= MAX_NFFT && hop > 0, "storm_stft: n_fft=%d hop=%d", n_fft, hop);
    STORM_CHECK(L > n_fft / 2, "storm_stft: signal too short for reflect padding (L=%lld)", L);
    STORM_CHECK(n_frames == 1 + (int)(L / hop) && Tpad >= n_frames, "storm_stft: n_frames=%d Tpad=%d L=%lld", n_frames, Tpad, L);
    hipLaunchKernelGGL(stft_kernel, dim3(Tpad, B), dim3(256), 0, (hipStream_t)s, wav, peak, spec, window, twiddle, L, stride,
                       n_fft, hop, n_frames, Tpad, spec_factor, spec_abs_exponent, row_len);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_istft(const float* spec, const float* peak, float* wav, float* frames, const float* window,
                           const float* twiddle, int B, int T, long long L, long long stride, int n_fft, int hop,
                           float spec_factor, float spec_abs_exponent, const int* row_len, storm_stream_t s) {
    STORM_CHECK(spec && wav && frames && window && twiddle && B > 0 && T > 0, "storm_istft: null pointer");
    STORM_CHECK(n_fft >= 2 && n_fft <= MAX_NFFT && hop > 0, "storm_istft: n_fft=%d hop=%d", n_fft, hop);
    STORM_CHECK(L > 0 && L <= (long long)n_fft + (long long)hop * (T - 1) - n_fft / 2, "storm_istft: length %lld not covered by %d frames", L, T);
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(istft_frames_kernel, dim3(T, B), dim3(256), 0, st, spec, frames, window, twiddle, T, n_fft, spec_factor, spec_abs_exponent);
    STORM_LAUNCH_CHECK();
    hipLaunchKernelGGL(istft_ola_kernel, dim3(cdiv(L, 256), B), dim3(256), 0, st, frames, window, peak, wav, T, L, stride, n_fft, hop, row_len);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}
