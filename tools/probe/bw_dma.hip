// HBM read-bandwidth probe for gfx950: LDS-DMA vs register loads, how many bytes in flight per CU.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int WAVES, int INFLIGHT, bool NT>
__global__ __launch_bounds__(WAVES * 64) void dma_read(const char* __restrict__ src, size_t bytes_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
  const size_t n_kb = bytes_per_wg / 1024;                       // 1-KiB pieces of this workgroup
  char* ring = smem + wave * (INFLIGHT * 1024);
  int slot = 0;
  for (size_t i = wave; i < n_kb; i += WAVES) {
    const char* p = base + i * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(ring + slot * 1024), 16, 0, NT ? 2 : 0);
    slot = slot + 1 == INFLIGHT ? 0 : slot + 1;
    if (slot == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);
}

template <int WAVES, int U, bool NT>
__global__ __launch_bounds__(WAVES * 64) void reg_read(const char* __restrict__ src, size_t bytes_per_wg, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
  const size_t n_kb = bytes_per_wg / 1024;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = wave; i < n_kb; i += WAVES * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + (size_t)u * WAVES;
      const u32x4* p = reinterpret_cast<const u32x4*>(base + (j < n_kb ? j : i) * 1024 + lane * 16);
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}

// the skinny GEMM's weight access pattern, loads only: workgroup = 8 waves, NT tiles of 16 rows (row = K*2 bytes), wave w takes the
// 64-byte k blocks w, w+8, ...; lane l reads 16 B of row (l & 15) at k block offset (l >> 4) * 16
template <int NT, int U, bool NT_LOAD, bool HALF>
__global__ __launch_bounds__(512) void skinny_pattern(const char* __restrict__ src, int K_bytes, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int rows_per_wg = NT * (HALF ? 8 : 16);
  const char* base = src + (size_t)blockIdx.x * rows_per_wg * K_bytes;
  const int nkb = K_bytes / (HALF ? 128 : 64);
  u32x4 acc = {0, 0, 0, 0};
  for (int kb0 = wave; kb0 < nkb; kb0 += 8 * U) {
    u32x4 v[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = kb0 + 8 * u;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const char* p = HALF ? base + (size_t)(j * 8 + (r16 & 7)) * K_bytes + kb * 128 + (r16 >> 3) * 64 + g * 16
                             : base + (size_t)(j * 16 + r16) * K_bytes + kb * 64 + g * 16;
        if (kb < nkb) v[u][j] = NT_LOAD ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
        else v[u][j] = u32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc ^= v[u][j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x & 1023] = 1;
}

// skinny pattern + what the real kernel adds: MODE 1 = also load the activation fragment of every k step from a small L2-resident
// buffer (global loads in the same queue), MODE 2 = activation fragments from LDS instead, MODE 3 = MODE 1 + one MFMA per fragment
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int NT, int U, int MODE>
__global__ __launch_bounds__(512) void skinny_plus(const char* __restrict__ src, const char* __restrict__ act, int K_bytes, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const char* base = src + (size_t)blockIdx.x * NT * 8 * K_bytes;
  const int nkb = K_bytes / 128;
  if (MODE == 2) {
    for (int i = threadIdx.x * 16; i < 8 * K_bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(lds + i) = *reinterpret_cast<const u32x4*>(act + i);
    __syncthreads();
  }
  u32x4 acc = {0, 0, 0, 0};
  f32x4_t facc[NT];
  for (int j = 0; j < NT; ++j) facc[j] = f32x4_t{0, 0, 0, 0};
  for (int kb0 = wave; kb0 < nkb; kb0 += 8 * U) {
    u32x4 v[U][NT], a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = kb0 + 8 * u;
      const size_t ko = (size_t)kb * 128 + (r16 >> 3) * 64 + g * 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const char* p = base + (size_t)(j * 8 + (r16 & 7)) * K_bytes + ko;
        v[u][j] = kb < nkb ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : u32x4{0, 0, 0, 0};
      }
      if (MODE == 1 || MODE == 3) a[u] = kb < nkb ? *reinterpret_cast<const u32x4*>(act + (size_t)(r16 & 7) * K_bytes + ko) : u32x4{0, 0, 0, 0};
      else if (MODE == 2) a[u] = *reinterpret_cast<const u32x4*>(lds + (size_t)(r16 & 7) * K_bytes + (kb < nkb ? ko : 0));
      else a[u] = u32x4{1, 2, 3, 4};
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (MODE == 3) facc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[u]), __builtin_bit_cast(bf16x8_t, v[u][j]), facc[j], 0, 0, 0);
        else acc ^= v[u][j] ^ a[u];
      }
  }
  float fs = 0;
  for (int j = 0; j < NT; ++j) fs += facc[j][0] + facc[j][1] + facc[j][2] + facc[j][3];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u || fs == 1.2345f) sink[blockIdx.x & 1023] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> float timeit(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main() {
  const size_t total = (size_t)2 << 30;   // 2 GiB: beyond the 256 MiB Infinity Cache
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4096 * 4));
  CK(hipMemset(buf, 1, total));
  int ncu = 256;
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = ncu * wgs_per_cu;
    const size_t per = total / grid / 1024 * 1024;
    auto rep = [&](const char* name, float ms) { printf("%-44s grid %4d: %8.1f us  %7.0f GB/s\n", name, grid, ms * 1e3, per * grid / ms / 1e6); };
#define DMA(W, I, NTF) { auto k = dma_read<W, I, NTF>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W * I * 1024)); \
    rep("dma waves=" #W " inflight=" #I " nt=" #NTF, timeit([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(W * 64), W * I * 1024, 0, buf, per, sink); }, 5)); }
    DMA(4, 8, false) DMA(4, 16, false) DMA(4, 16, true) DMA(8, 8, true) DMA(4, 32, true)
#define REG(W, U, NTF) rep("reg waves=" #W " unroll=" #U " nt=" #NTF, timeit([&] { hipLaunchKernelGGL((reg_read<W, U, NTF>), dim3(grid), dim3(W * 64), 0, 0, buf, per, sink); }, 5));
    REG(4, 8, false) REG(4, 8, true) REG(8, 8, true) REG(8, 16, true) REG(16, 8, true)
  }
  {
    const int Kb = 4096;   // K = 2048 bf16
    auto rep2 = [&](const char* name, int rows_per_wg, float ms) { int grid = (int)(total / Kb / rows_per_wg); printf("%-44s grid %5d: %8.1f us  %7.0f GB/s\n", name, grid, ms * 1e3, (double)grid * rows_per_wg * Kb / ms / 1e6); };
#define SK(NTT, U, NTF, H) rep2("skinny-pattern NT=" #NTT " U=" #U " nt=" #NTF " half=" #H, NTT * (H ? 8 : 16), timeit([&] { int grid = (int)(total / Kb / (NTT * (H ? 8 : 16))); hipLaunchKernelGGL((skinny_pattern<NTT, U, NTF, H>), dim3(grid), dim3(512), 0, 0, buf, Kb, sink); }, 5));
    // single-round launches of one gate/up-sized matrix (67 MB), as the decode step issues them
    { const size_t one = (size_t)16384 * Kb;
      auto rep3 = [&](const char* name, float ms) { printf("%-44s one 67 MB matrix: %8.1f us  %7.0f GB/s\n", name, ms * 1e3, one / ms / 1e6); };
      int it = 0;
      rep3("single round NT=4 U=2 half (512 WGs)", timeit([&] { hipLaunchKernelGGL((skinny_pattern<4, 2, true, true>), dim3(512), dim3(512), 0, 0, buf + (size_t)(it++ % 24) * one, Kb, sink); }, 48));
      rep3("single round NT=4 U=4 half (512 WGs)", timeit([&] { hipLaunchKernelGGL((skinny_pattern<4, 4, true, true>), dim3(512), dim3(512), 0, 0, buf + (size_t)(it++ % 24) * one, Kb, sink); }, 48));
      rep3("single round NT=2 U=4 half (1024 WGs)", timeit([&] { hipLaunchKernelGGL((skinny_pattern<2, 4, true, true>), dim3(1024), dim3(512), 0, 0, buf + (size_t)(it++ % 24) * one, Kb, sink); }, 48));
      rep3("single round NT=4 U=4 (256 WGs)", timeit([&] { hipLaunchKernelGGL((skinny_pattern<4, 4, true, false>), dim3(256), dim3(512), 0, 0, buf + (size_t)(it++ % 24) * one, Kb, sink); }, 48));
    }
    { char* act; CK(hipMalloc(&act, 8 * Kb)); CK(hipMemset(act, 0, 8 * Kb));
      auto rep4 = [&](const char* name, float ms) { int grid = (int)(total / Kb / 32); printf("%-44s grid %5d: %8.1f us  %7.0f GB/s\n", name, grid, ms * 1e3, (double)grid * 32 * Kb / ms / 1e6); };
#define SP(MODE, LDSB) rep4("skinny+ NT=4 U=2 half mode=" #MODE, timeit([&] { int grid = (int)(total / Kb / 32); hipLaunchKernelGGL((skinny_plus<4, 2, MODE>), dim3(grid), dim3(512), LDSB, 0, buf, act, Kb, sink); }, 5));
      SP(0, 0) SP(1, 0) SP(2, 8 * Kb) SP(3, 0)
    }
    SK(4, 4, true, false) SK(4, 4, false, false) SK(4, 8, true, false) SK(1, 8, true, false) SK(4, 4, true, true) SK(1, 8, true, true) SK(8, 4, true, false)
  }
  return 0;
}
