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

// FR consecutive frames per workgroup: a thread owns one frequency bin and walks the 400 samples once for all FR frames, so
// every (conflict-prone) 16-byte twiddle read from LDS feeds 2 * FR fp64 FMAs and the sample reads are LDS broadcasts.
// Per (frame, bin) the accumulation order is the plain n = 0..399 loop.
constexpr int kFR = 8;
// order-preserving float <-> unsigned map (key 0 is below every float: a zeroed buffer is the identity of atomicMax)
__device__ __forceinline__ unsigned float_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ wav, int64_t ld, const int* __restrict__ lens,
                                                        const double* __restrict__ tw, const float* __restrict__ melfb,
                                                        float* __restrict__ mel_ws, unsigned* __restrict__ max_key, int round_bf16) {
  __shared__ __attribute__((aligned(16))) double sx[kNfft][kFR];   // windowed samples, already widened: [n][frame] -> broadcast 16-byte reads
  __shared__ float spow[kFR][kBins + 7];
  __shared__ float smax[4];
  __shared__ __attribute__((aligned(16))) double stw[kNfft * 2];
  const int frame0 = blockIdx.x * kFR, b = blockIdx.y;
  const int len = lens[b] < kSamples ? lens[b] : kSamples;
  float* out = mel_ws + ((int64_t)b * kFrames + frame0) * kMels;
  // a frame whose reflected support lies entirely in the zero padding has power 0 -> log10(1e-10) = -10
  // (only the very last frames reflect at the right edge); a workgroup whose first frame is already there has nothing to do
  const int start0 = frame0 * kHop - kNfft / 2;
  const bool tail_reflect = (frame0 + kFR - 1) * kHop - kNfft / 2 + kNfft > kSamples;
  if (start0 >= len && !tail_reflect) {
    for (int i = threadIdx.x; i < kFR * kMels; i += 256) out[i] = -10.f;
    if (threadIdx.x == 0) atomicMax(max_key + b, float_key(-10.f));
    return;
  }
  for (int i = threadIdx.x; i < kFR * kNfft; i += 256) {
    const int fr = i / kNfft, n = i - fr * kNfft;
    int j = (frame0 + fr) * kHop - kNfft / 2 + n;
    if (j < 0) j = -j;
    if (j >= kSamples) j = 2 * (kSamples - 1) - j;
    float v = j < len ? wav[(int64_t)b * ld + j] : 0.f;
    // bf16 mode = the reference's `bf16-true` evaluation: Lightning's HalfPrecision.convert_input hands forward() the WAVEFORM in bf16
    // (hence `audio.to(torch.float32)` at modeling_OmniAVSR.py:530); the front end itself then runs in fp32 / fp64 as there
    if (round_bf16) v = to_f32(from_f32<bf16_t>(v));
    // periodic Hann: 0.5 - 0.5 cos(2 pi n / 400) = 0.5 - 0.5 * tw[n].cos
    float w = (float)(0.5 - 0.5 * tw[2 * n]);
    sx[n][fr] = (double)(v * w);
  }
  for (int n = threadIdx.x; n < kNfft * 2; n += 256) stw[n] = tw[n];
  __syncthreads();
  if (threadIdx.x < kBins) {
    const int f = threadIdx.x;
    double re[kFR], im[kFR];
#pragma unroll
    for (int fr = 0; fr < kFR; ++fr) { re[fr] = 0.0; im[fr] = 0.0; }
    int idx = 0;
    for (int n = 0; n < kNfft; ++n) {
      const double c = stw[2 * idx], sn = stw[2 * idx + 1];
#pragma unroll
      for (int fr = 0; fr < kFR; ++fr) {
        const double x = sx[n][fr];
        re[fr] += x * c;
        im[fr] -= x * sn;
      }
      idx += f;
      if (idx >= kNfft) idx -= kNfft;
    }
#pragma unroll
    for (int fr = 0; fr < kFR; ++fr) spow[fr][f] = (float)(re[fr] * re[fr] + im[fr] * im[fr]);
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < kFR * kMels; i += 256) {
    const int fr = i / kMels, m = i - fr * kMels;
    float a = 0.f;
    for (int f = 0; f < kBins; ++f) a += melfb[f * kMels + m] * spow[fr][f];
    const float v = log10f(fmaxf(a, 1e-10f));
    out[i] = v;
    mx = fmaxf(mx, v);
  }
  // per-utterance maximum (the clamp to max - 8): one order-preserving-integer atomicMax per workgroup
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(max_key + b, float_key(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));
}

template <typename T>
__global__ __launch_bounds__(256) void mel_finalize_kernel(const float* __restrict__ mel_ws, const unsigned* __restrict__ max_key,
                                                            T* __restrict__ out, int64_t ldo, int B) {
  const int64_t total = (int64_t)B * kFrames * kMels;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int m = (int)(i % kMels);
    int64_t row = i / kMels;
    int b = (int)(row / kFrames);
    float v = fmaxf(mel_ws[i], key_float(max_key[b]) - 8.0f);
    out[row * ldo + m] = from_f32<T>((v + 4.0f) / 4.0f);
  }
}

}  // namespace

int launch_logmel(int dtype, const float* wav, int64_t ld, const int* lens, int B, const double* twiddle, const float* melfb,
                  float* mel_ws, float* max_ws, void* out, int64_t ldo, hipStream_t st) {
  if (B <= 0) return 0;
  static_assert(kFrames % kFR == 0, "frames per workgroup");
  unsigned* max_key = reinterpret_cast<unsigned*>(max_ws);
  OAVSR_CHECK_HIP(hipMemsetAsync(max_key, 0, sizeof(unsigned) * B, st));
  hipLaunchKernelGGL(stft_mel_kernel, dim3(kFrames / kFR, B), dim3(256), 0, st, wav, ld, lens, twiddle, melfb, mel_ws, max_key, dtype == DT_BF16 ? 1 : 0);
  if (dtype == DT_F32) hipLaunchKernelGGL(mel_finalize_kernel<float>, dim3(2048), dim3(256), 0, st, mel_ws, max_key, (float*)out, ldo, B);
  else hipLaunchKernelGGL(mel_finalize_kernel<bf16_t>, dim3(2048), dim3(256), 0, st, mel_ws, max_key, (bf16_t*)out, ldo, B);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace oavsr
