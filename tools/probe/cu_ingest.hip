// How fast can ONE CU take in a once-read HBM stream, as a function of the bytes it keeps in flight and of how many CUs stream at the
// same time?  One workgroup per CU (the 100 KB LDS request keeps a second one off the CU), WAVES waves, U 16-byte loads per lane in
// flight (WAVES x U KiB per CU).  hipcc --offload-arch=gfx950 -O3 cu_ingest.hip -o cu_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void stream_read(const char* __restrict__ src, size_t bytes_per_wg, unsigned* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * bytes_per_wg;
  const size_t n_kb = bytes_per_wg / 1024;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = wave; i < n_kb; i += (size_t)WAVES * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + (size_t)u * WAVES;
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (j < n_kb ? j : i) * 1024 + lane * 16));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) { sink[blockIdx.x] = 1; lds[0] = 1; }
}

template <int WAVES, int U>
void run(const char* buf, unsigned* sink, int ncu, size_t per_wg) {
  auto k = stream_read<WAVES, U>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    // a different 2 GiB window every repetition: nothing is served by the 256 MiB Infinity Cache
    const char* src = buf + (size_t)(rep & 1) * ((size_t)2 << 30);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(WAVES * 64), 100 * 1024, 0, src, per_wg, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  const double gbs = (double)ncu * per_wg / best / 1e6;
  printf("CUs %3d  waves %2d  U %2d  in flight %3d KiB/CU : %7.1f GB/s = %5.1f GB/s per CU\n", ncu, WAVES, U, WAVES * U, gbs, gbs / ncu);
}

// Decode-GEMV-shaped streams: `total` bytes of weights per launch, cut into contiguous slices of `per_wg` bytes, one workgroup (8 waves,
// 64 KiB in flight) per slice, no LDS padding (several workgroups per CU).  Launches walk through the 4 GiB buffer, so nothing repeats
// within 256 MiB.  Time per launch includes the launch itself, as in the decode loop.
template <int WAVES, int U>
void run_small(const char* buf, unsigned* sink, size_t total, size_t per_wg, const char* what) {
  auto k = stream_read<WAVES, U>;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = (int)(total / per_wg), iters = 24;
  size_t off = 0;
  auto step = [&]() {
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 0, 0, buf + off, per_wg, sink);
    off += total; if (off + total > ((size_t)4 << 30)) off = 0;
  };
  for (int i = 0; i < 3; ++i) step();
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) step();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  ms /= iters;
  printf("%-10s %6.1f MB in %5d slices of %4zu KiB (waves %d, U %d): %7.1f us per launch, %7.1f GB/s\n", what, total / 1e6, grid, per_wg >> 10, WAVES, U,
         ms * 1e3, total / ms / 1e6);
}

int main() {
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, (size_t)4 << 30)); CK(hipMalloc(&sink, 4096 * 4));
  CK(hipMemset(buf, 1, (size_t)4 << 30));
  const size_t per_wg = (size_t)8 << 20;                    // 8 MiB per CU: 2 GiB with 256 CUs
  const int cus[] = {16, 64, 128, 256};
  for (int ncu : cus) {
    run<1, 4>(buf, sink, ncu, per_wg);
    run<4, 2>(buf, sink, ncu, per_wg);
    run<4, 8>(buf, sink, ncu, per_wg);
    run<8, 8>(buf, sink, ncu, per_wg);
    run<16, 8>(buf, sink, ncu, per_wg);
    run<16, 16>(buf, sink, ncu, per_wg);
  }
  // the Llama-3.2-1B decode GEMVs (bf16): QKV 12.6 MB, o_proj 8.4 MB, gate/up 67 MB, down 33.5 MB, lm_head 525 MB
  struct { const char* what; size_t total; } shapes[] = {{"o_proj", (size_t)8 << 20}, {"qkv", (size_t)12 << 20}, {"down", (size_t)32 << 20},
                                                       {"gate_up", (size_t)64 << 20}, {"lm_head", (size_t)500 << 20}};
  for (auto& sh : shapes) {
    run_small<8, 8>(buf, sink, sh.total, (size_t)32 << 10, sh.what);
    run_small<8, 8>(buf, sink, sh.total, (size_t)64 << 10, sh.what);
    run_small<8, 8>(buf, sink, sh.total, (size_t)128 << 10, sh.what);
    run_small<4, 8>(buf, sink, sh.total, (size_t)32 << 10, sh.what);
    run_small<4, 16>(buf, sink, sh.total, (size_t)64 << 10, sh.what);
  }
  return 0;
}
