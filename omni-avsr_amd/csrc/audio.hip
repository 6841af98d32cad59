// Whisper log-mel front-end on device (K10).  Restates WhisperFeatureExtractor's torch path
// (STFT n_fft 400 / hop 160 / periodic Hann / center+reflect, |.|^2, drop last frame, 80 slaney mels,
// log10(clamp 1e-10), clamp(max-8) per utterance, (x+4)/4) so the reference's device->host->numpy->device
// round trip (modeling_OmniAVSR.py:531-534) disappears.  The 201-bin DFT of each 400-sample frame is a
// direct transform with fp64 accumulation (3000 x 201 x 400 MACs per utterance: negligible next to the
// encoder), which is at least as accurate as the reference's fp32 FFT.
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kNfft = 400, kHop = 160, kBins = 201, kMels = 80, kFrames = 3000, kSamples = 480000;

__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ wav, int64_t ld, const int* __restrict__ lens,
                                                        const double* __restrict__ tw, const float* __restrict__ melfb,
                                                        float* __restrict__ mel_ws) {
  __shared__ float sx[kNfft];
  __shared__ float spow[kBins + 7];
  __shared__ double stw[kNfft * 2];
  const int frame = blockIdx.x, b = blockIdx.y;
  const int len = lens[b] < kSamples ? lens[b] : kSamples;
  const int start = frame * kHop - kNfft / 2;
  float* out = mel_ws + ((int64_t)b * kFrames + frame) * kMels;
  // a frame whose reflected support lies entirely in the zero padding has power 0 -> log10(1e-10) = -10
  const bool tail_reflect = start + kNfft > kSamples;     // only the very last frames reflect at the right edge
  if (start >= len && !tail_reflect) {
    for (int m = threadIdx.x; m < kMels; m += 256) out[m] = -10.f;
    return;
  }
  for (int n = threadIdx.x; n < kNfft; n += 256) {
    int i = start + n;
    if (i < 0) i = -i;
    if (i >= kSamples) i = 2 * (kSamples - 1) - i;
    float v = i < len ? wav[(int64_t)b * ld + i] : 0.f;
    // periodic Hann: 0.5 - 0.5 cos(2 pi n / 400) = 0.5 - 0.5 * tw[n].cos
    float w = (float)(0.5 - 0.5 * tw[2 * n]);
    sx[n] = v * w;
  }
  for (int n = threadIdx.x; n < kNfft * 2; n += 256) stw[n] = tw[n];
  __syncthreads();
  if (threadIdx.x < kBins) {
    const int f = threadIdx.x;
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int n = 0; n < kNfft; ++n) {
      double x = (double)sx[n];
      re += x * stw[2 * idx];
      im -= x * stw[2 * idx + 1];
      idx += f;
      if (idx >= kNfft) idx -= kNfft;
    }
    spow[f] = (float)(re * re + im * im);
  }
  __syncthreads();
  if (threadIdx.x < kMels) {
    const int m = threadIdx.x;
    float a = 0.f;
    for (int f = 0; f < kBins; ++f) a += melfb[f * kMels + m] * spow[f];
    out[m] = log10f(fmaxf(a, 1e-10f));
  }
}

__global__ __launch_bounds__(256) void utt_max_kernel(const float* __restrict__ mel_ws, float* __restrict__ max_ws) {
  __shared__ float sm[4];
  const float* p = mel_ws + (int64_t)blockIdx.x * kFrames * kMels;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < kFrames * kMels; i += 256) m = fmaxf(m, p[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) max_ws[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void mel_finalize_kernel(const float* __restrict__ mel_ws, const float* __restrict__ max_ws,
                                                            T* __restrict__ out, int64_t ldo, int B) {
  const int64_t total = (int64_t)B * kFrames * kMels;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int m = (int)(i % kMels);
    int64_t row = i / kMels;
    int b = (int)(row / kFrames);
    float v = fmaxf(mel_ws[i], max_ws[b] - 8.0f);
    out[row * ldo + m] = from_f32<T>((v + 4.0f) / 4.0f);
  }
}

}  // namespace

int launch_logmel(int dtype, const float* wav, int64_t ld, const int* lens, int B, const double* twiddle, const float* melfb,
                  float* mel_ws, float* max_ws, void* out, int64_t ldo, hipStream_t st) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(stft_mel_kernel, dim3(kFrames, B), dim3(256), 0, st, wav, ld, lens, twiddle, melfb, mel_ws);
  hipLaunchKernelGGL(utt_max_kernel, dim3(B), dim3(256), 0, st, mel_ws, max_ws);
  if (dtype == DT_F32) hipLaunchKernelGGL(mel_finalize_kernel<float>, dim3(2048), dim3(256), 0, st, mel_ws, max_ws, (float*)out, ldo, B);
  else hipLaunchKernelGGL(mel_finalize_kernel<bf16_t>, dim3(2048), dim3(256), 0, st, mel_ws, max_ws, (bf16_t*)out, ldo, B);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace oavsr
