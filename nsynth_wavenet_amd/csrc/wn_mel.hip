// Log-mel featuriser of the generation path on the device.
//
// What the reference computes on the host with librosa (auxilaries/mel_extractor.py:14-35 parameters, :31-44
// melspectrogram, :65-90 stft / mel basis / amp_to_db / normalise), restated for the GPU:
//   frames : centred, reflect-padded, hop 200; a periodic Hann window of 800 samples in the middle of a 2048-point
//            frame -- so only 800 samples of a frame are non-zero and frame f reads wav[200 f - 400 .. 200 f + 399];
//   |STFT| : 1025 magnitudes of the 2048-point DFT (the window's offset inside the frame is a phase factor and drops
//            out of the magnitude, so the DFT runs over the 800 live samples with twiddle index (k n) mod 2048);
//   mel    : Slaney-scale area-normalised triangles, 80 bands over 125-7600 Hz, applied to the MAGNITUDE;
//   output : clip((20 log10(max(1e-5, S)) + 140) / 140, 0, 1), time-major [B][F][80], F = 1 + L / 200.
// (`preemphasis` and `ref_level_db` are declared by the reference and never applied.)
//
// One workgroup = MEL_FR consecutive frames of one utterance.  The windowed samples of the frames (MEL_FR x 800
// floats) and one period of the twiddle table (2048 x (cos, sin)) sit in LDS; a thread owns DFT bins k, k + 256,
// k + 512, k + 768 and walks n with the table index advancing by k (mod 2048): one table read feeds 2 MEL_FR FMAs.
// Bin 1024 (twiddle (-1)^n) is a wave reduction.  The magnitudes go back to LDS and the 80 triangles are applied
// from a band table (first bin, count, weights) built on the host in double precision.  Arithmetic is fp32 with
// fp32 accumulation over the 800 samples; tests/test_mel.py holds it to the float64 restatement in oracle/mel_np.py
// and to closed-form tones / impulses.  2.5 MFLOP per frame: this is not a hot kernel and is written for clarity.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <vector>

#include "wn_internal.h"

namespace {

constexpr int MEL_SR = 16000;
constexpr int MEL_NFFT = 2048;
constexpr int MEL_NBIN = MEL_NFFT / 2 + 1;     // 1025
constexpr int MEL_HOP = 200;                   // 12.5 ms
constexpr int MEL_WIN = 800;                   // 50 ms
constexpr int MEL_NB = 80;
constexpr int MEL_FR = 8;                      // frames per workgroup
constexpr float MEL_MIN_AMP = 1e-5f;
constexpr float MEL_MIN_DB = -140.f;
constexpr int MEL_LDS_FLOATS = 2 * MEL_NFFT + MEL_FR * MEL_WIN + MEL_FR * (MEL_NBIN + 3);

struct MelTables {
    float* twiddle = nullptr;    // [2048][2] cos, sin of 2 pi i / 2048
    float* window = nullptr;     // [800] periodic Hann
    int* band = nullptr;         // [80][2] first bin, bin count
    float* weight = nullptr;     // [80][MEL_WMAX] triangle weights
    int wmax = 0;
};

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wav, int64_t L, int F,
                                                  const float* __restrict__ twiddle, const float* __restrict__ window,
                                                  const int* __restrict__ band, const float* __restrict__ weight,
                                                  int wmax, float* __restrict__ mel) {
    extern __shared__ float lds[];
    float* tw = lds;                                  // [2048][2]
    float* xs = tw + 2 * MEL_NFFT;                    // [MEL_FR][800]
    float* mag = xs + MEL_FR * MEL_WIN;               // [MEL_FR][1028]
    constexpr int MS = MEL_NBIN + 3;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * MEL_FR;
    const float* y = wav + (size_t)b * L;
    for (int i = threadIdx.x; i < 2 * MEL_NFFT; i += 256) tw[i] = twiddle[i];
    for (int i = threadIdx.x; i < MEL_FR * MEL_WIN; i += 256) {
        const int fr = i / MEL_WIN, n = i - fr * MEL_WIN;
        // sample of the reflect-padded signal (numpy 'reflect': the edge sample is not repeated)
        int64_t j = (int64_t)(f0 + fr) * MEL_HOP - MEL_WIN / 2 + n;
        if (j < 0) j = -j;
        if (j >= L) j = 2 * (L - 1) - j;
        xs[i] = (f0 + fr < F) ? y[j] * window[n] : 0.f;
    }
    __syncthreads();

    // bins k0 + 256 r, r = 0..3
    float re[4][MEL_FR], im[4][MEL_FR];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int fr = 0; fr < MEL_FR; ++fr) re[r][fr] = im[r][fr] = 0.f;
    int idx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) idx[r] = 0;
    for (int n = 0; n < MEL_WIN; ++n) {
        float x[MEL_FR];
#pragma unroll
        for (int fr = 0; fr < MEL_FR; ++fr) x[fr] = xs[fr * MEL_WIN + n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c = tw[2 * idx[r]], s = tw[2 * idx[r] + 1];
#pragma unroll
            for (int fr = 0; fr < MEL_FR; ++fr) {
                re[r][fr] = fmaf(x[fr], c, re[r][fr]);
                im[r][fr] = fmaf(x[fr], s, im[r][fr]);
            }
            idx[r] = (idx[r] + (int)threadIdx.x + 256 * r) & (MEL_NFFT - 1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int fr = 0; fr < MEL_FR; ++fr)
            mag[fr * MS + threadIdx.x + 256 * r] = sqrtf(re[r][fr] * re[r][fr] + im[r][fr] * im[r][fr]);
    // bin 1024: sum x[n] (-1)^n, one wave per two frames
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int fr = wave; fr < MEL_FR; fr += 4) {
            float a = 0.f;
            for (int n = lane; n < MEL_WIN; n += 64) a += (n & 1) ? -xs[fr * MEL_WIN + n] : xs[fr * MEL_WIN + n];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) mag[fr * MS + MEL_NFFT / 2] = fabsf(a);
        }
    }
    __syncthreads();

    for (int o = threadIdx.x; o < MEL_FR * MEL_NB; o += 256) {
        const int fr = o / MEL_NB, m = o - fr * MEL_NB;
        if (f0 + fr >= F) continue;
        const int k0 = band[2 * m], cnt = band[2 * m + 1];
        const float* wm = weight + (size_t)m * wmax;
        float s = 0.f;
        for (int i = 0; i < cnt; ++i) s = fmaf(wm[i], mag[fr * MS + k0 + i], s);
        const float db = 20.f * log10f(fmaxf(MEL_MIN_AMP, s));
        const float ns = fminf(fmaxf((db - MEL_MIN_DB) / -MEL_MIN_DB, 0.f), 1.f);
        mel[((size_t)b * F + f0 + fr) * MEL_NB + m] = ns;
    }
}

// Slaney mel scale (librosa.filters.mel defaults htk=False, norm='slaney'; mel_extractor.py:74-78)
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_hz / f_sp + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_mel = 1000.0 / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? 1000.0 * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

std::mutex g_mel_mu;
MelTables g_mel_tab[16];

int mel_tables(int dev, MelTables** out) {
    if (dev < 0 || dev >= 16) return wn_fail(nullptr, WN_EINVAL, "wn_mel_spectrogram: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_mel_mu);
    MelTables& t = g_mel_tab[dev];
    *out = &t;
    if (t.twiddle) return WN_OK;
    const double PI = 3.14159265358979323846;
    std::vector<float> tw(2 * MEL_NFFT), win(MEL_WIN);
    for (int i = 0; i < MEL_NFFT; ++i) {
        tw[2 * i] = (float)std::cos(2 * PI * i / MEL_NFFT);
        tw[2 * i + 1] = (float)std::sin(2 * PI * i / MEL_NFFT);
    }
    for (int n = 0; n < MEL_WIN; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2 * PI * n / MEL_WIN));
    // triangles over the FFT bin frequencies
    std::vector<double> mel_f(MEL_NB + 2);
    const double m_lo = hz_to_mel(125.0), m_hi = hz_to_mel(7600.0);
    for (int i = 0; i < MEL_NB + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (MEL_NB + 1));
    std::vector<std::vector<float>> wts(MEL_NB);
    std::vector<int> band(2 * MEL_NB);
    int wmax = 1;
    for (int m = 0; m < MEL_NB; ++m) {
        const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
        int first = -1, last = -1;
        std::vector<double> row(MEL_NBIN);
        for (int k = 0; k < MEL_NBIN; ++k) {
            const double fk = (double)k * (MEL_SR / 2.0) / (MEL_NBIN - 1);
            const double lower = (fk - mel_f[m]) / (mel_f[m + 1] - mel_f[m]);
            const double upper = (mel_f[m + 2] - fk) / (mel_f[m + 2] - mel_f[m + 1]);
            const double w = std::fmax(0.0, std::fmin(lower, upper)) * enorm;
            row[k] = w;
            if (w > 0) { if (first < 0) first = k; last = k; }
        }
        if (first < 0) { first = 0; last = -1; }
        band[2 * m] = first;
        band[2 * m + 1] = last - first + 1;
        for (int k = first; k <= last; ++k) wts[m].push_back((float)row[k]);
        wmax = std::max(wmax, last - first + 1);
    }
    std::vector<float> weight((size_t)MEL_NB * wmax, 0.f);
    for (int m = 0; m < MEL_NB; ++m)
        for (size_t i = 0; i < wts[m].size(); ++i) weight[(size_t)m * wmax + i] = wts[m][i];
    auto up = [&](auto** dst, const auto& src) -> bool {
        if (hipMalloc((void**)dst, src.size() * sizeof(src[0])) != hipSuccess) return false;
        return hipMemcpy(*dst, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&t.twiddle, tw) || !up(&t.window, win) || !up(&t.band, band) || !up(&t.weight, weight)) {
        t.twiddle = nullptr;
        return wn_fail(nullptr, WN_EIO, "wn_mel_spectrogram: cannot upload the featuriser tables");
    }
    t.wmax = wmax;
    // 74 880 B of dynamic LDS, above the 64 KB a kernel gets without asking (gfx950 has 160 KB per CU)
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(mel_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, MEL_LDS_FLOATS * (int)sizeof(float));
    if (ea != hipSuccess) {
        (void)hipFree(t.twiddle); (void)hipFree(t.window); (void)hipFree(t.band); (void)hipFree(t.weight);
        t.twiddle = nullptr;
        return wn_fail(nullptr, WN_EIO, "wn_mel_spectrogram: this device does not grant the %d bytes of LDS the "
                       "featuriser kernel needs (%s)", MEL_LDS_FLOATS * (int)sizeof(float), hipGetErrorString(ea));
    }
    return WN_OK;
}

}  // namespace

extern "C" int64_t wn_mel_frames(int64_t n_samples) { return n_samples < 0 ? -1 : 1 + n_samples / MEL_HOP; }

extern "C" int wn_mel_spectrogram(const float* wav, int B, int64_t L, float* mel, void* stream) {
    if (!wav || !mel) return wn_fail(nullptr, WN_EINVAL, "wn_mel_spectrogram: null pointer");
    if (B < 1) return wn_fail(nullptr, WN_EINVAL, "wn_mel_spectrogram: B must be >= 1");
    // the reference's reflect padding of n_fft / 2 samples needs a signal longer than the pad (numpy.pad raises)
    if (L <= MEL_NFFT / 2)
        return wn_fail(nullptr, WN_EINVAL, "wn_mel_spectrogram: %lld samples, need more than %d (reflect padding of the "
                       "centred 2048-point frames)", (long long)L, MEL_NFFT / 2);
    if (L > 0x7fffffff / 2) return wn_fail(nullptr, WN_EINVAL, "wn_mel_spectrogram: utterance too long");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return wn_fail(nullptr, WN_EIO, "wn_mel_spectrogram: no HIP device");
    MelTables* t = nullptr;
    if (int rc = mel_tables(dev, &t)) return rc;
    const int F = (int)(1 + L / MEL_HOP);
    dim3 grid((F + MEL_FR - 1) / MEL_FR, B);
    hipLaunchKernelGGL(mel_kernel, grid, dim3(256), MEL_LDS_FLOATS * sizeof(float), reinterpret_cast<hipStream_t>(stream),
                       wav, L, F, t->twiddle, t->window, t->band, t->weight, t->wmax, mel);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wn_fail(nullptr, WN_EIO, "wn_mel_spectrogram: launch failed: %s", hipGetErrorString(e));
    return WN_OK;
}
