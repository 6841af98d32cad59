// Probe (developer tool, not part of the library): what one workgroup can stream from HBM into registers / LDS on MI355X, by access
// pattern, loads in flight per wave and workgroup count -- the numbers the decode kernels' tile shapes are chosen by.
//   build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o gpurun_out/stream_probe     run: gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A "weight matrix" of rows x K bytes.  Workgroup b owns rows [b * RPW, (b + 1) * RPW); its 8 (W) waves interleave 128-byte k blocks.
// PAT 0: a wave load = 8 rows x 128 B (lane: row = l & 7 ... the 8-column decode tiles); PAT 1: 16 rows x 64 B, the two halves of a line
// as two loads (the 16-column tiles); PAT 2: 1 KiB contiguous (what a repacked weight layout would give).
template <int PAT, int D, int W, bool NT>
__global__ __launch_bounds__(W * 64) void probe_kernel(const char* __restrict__ base, long krow_bytes, int rows_per_wg, int ntiles_per_wg, u32x4* out, int nkb_shift) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wg_off = (long)blockIdx.x * rows_per_wg * krow_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base + wg_off), 0, 0x7fffffff, 0x00020000);
  // a "tile" = 8 (PAT 0) or 16 (PAT 1) rows; a step = one 128-byte k block of the tile's rows (1 or 2 KiB per wave step)
  const int rows_per_tile = PAT == 1 ? 16 : 8;
  const int NKB = 1 << nkb_shift;               // 128-byte k blocks read per row (the row stride krow_bytes may be padded beyond them)
  unsigned lane_off;
  if (PAT == 0) lane_off = (unsigned)((lane & 7) * krow_bytes + (lane >> 3) * 16);           // 8 rows x 8 chunks
  else if (PAT == 1) lane_off = (unsigned)((lane & 15) * krow_bytes + (lane >> 4) * 16);      // 16 rows x 4 chunks (+64 for the second half)
  else lane_off = (unsigned)(lane * 16);
  u32x4 acc = {0, 0, 0, 0};
  constexpr int LPS = PAT == 1 ? 2 : 1;         // loads per step
  u32x4 buf[D][LPS];
  const int total = ntiles_per_wg * (NKB / W);  // steps of this wave
  const unsigned tile_bytes = (unsigned)(rows_per_tile * krow_bytes);
  auto addr = [&](int s) -> unsigned {
    const int i = wave + s * W, tile = i >> nkb_shift, kb = i & (NKB - 1);
    if (PAT == 2) return (unsigned)(i * 1024) + lane_off;
    return tile * tile_bytes + kb * 128 + lane_off;
  };
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const unsigned a = addr(d < total ? d : 0);
#pragma unroll
    for (int h = 0; h < LPS; ++h) buf[d][h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a + h * 64, 0, NT ? 2 : 0));
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int s0 = 0; s0 < total; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int h = 0; h < LPS; ++h) acc ^= buf[d][h];
      const int nx = s0 + d + D;
      const unsigned a = addr(nx < total ? nx : total - 1);
#pragma unroll
      for (int h = 0; h < LPS; ++h) buf[d][h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a + h * 64, 0, NT ? 2 : 0));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc[0] == 0x12345678u) out[blockIdx.x * 64 + lane] = acc;      // never true in practice: keeps the loads
}

template <int PAT, int D, int W, bool NT>
double run(const char* buf, size_t buf_bytes, int G, long krow_bytes, int rows_total, u32x4* out, const char* tag, int nkb_shift = 7) {
  const int rows_per_tile = PAT == 1 ? 16 : 8;
  int rows_per_wg = rows_total / G / rows_per_tile * rows_per_tile;
  const int ntiles = rows_per_wg / rows_per_tile;
  const size_t bytes = (size_t)G * rows_per_wg * krow_bytes;
  if (bytes > buf_bytes) { printf("skip %s\n", tag); return 0; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    // a different region each time: the previous pass must not survive in the Infinity Cache
    const size_t off = ((size_t)rep * bytes) % (buf_bytes - bytes + 1) / 4096 * 4096;
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_kernel<PAT, D, W, NT>), dim3(G), dim3(W * 64), 0, 0, buf + off, krow_bytes, rows_per_wg, ntiles, out, nkb_shift);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double gbs = (double)G * rows_per_wg * (128.0 * (1 << nkb_shift)) / (best * 1e-3) / 1e9;
  printf("%-34s G=%3d  %7.1f MB  %8.1f us  %7.1f GB/s  %6.1f GB/s per WG\n", tag, G, bytes / 1e6, best * 1e3, gbs, gbs / G);
  return gbs;
}

int main() {
  const size_t buf_bytes = 3ull << 30;
  char* buf; hipMalloc(&buf, buf_bytes); hipMemset(buf, 1, buf_bytes);
  u32x4* out; hipMalloc(&out, 1 << 20);
  // row-stride sweep for the strided patterns: K bytes read per row, stride = K + pad
  struct Cfg { int shift; long pad; };
  for (int G : {32, 128}) {
    for (Cfg c : {Cfg{7, 0}, Cfg{7, 128}, Cfg{7, 256}, Cfg{7, 512}, Cfg{7, 1024}, Cfg{7, 2048 + 128}, Cfg{5, 0}, Cfg{5, 128}, Cfg{5, 256}, Cfg{6, 0}, Cfg{6, 128}}) {
      const long kb = 128L << c.shift, stride = kb + c.pad;
      const int rows = (int)((long)G * (4 << 20) / stride / 16 * 16);      // ~4 MB per workgroup
      char tag[96];
      snprintf(tag, sizeof(tag), "8x128B D=4 K=%ldB stride+%ld", kb, c.pad);
      run<0, 4, 8, true>(buf, buf_bytes, G, stride, rows, out, tag, c.shift);
      snprintf(tag, sizeof(tag), "8x128B D=8 K=%ldB stride+%ld", kb, c.pad);
      run<0, 8, 8, true>(buf, buf_bytes, G, stride, rows, out, tag, c.shift);
      snprintf(tag, sizeof(tag), "16x64Bx2 D=4 K=%ldB stride+%ld no-nt", kb, c.pad);
      run<1, 4, 8, false>(buf, buf_bytes, G, stride, rows, out, tag, c.shift);
    }
    const int rows = G * 256;
    run<2, 2, 8, true>(buf, buf_bytes, G, 16384, rows, out, "1KiB contiguous D=2  W=8 nt");
    run<2, 4, 8, true>(buf, buf_bytes, G, 16384, rows, out, "1KiB contiguous D=4  W=8 nt");
    run<2, 8, 8, true>(buf, buf_bytes, G, 16384, rows, out, "1KiB contiguous D=8  W=8 nt");
    run<2, 8, 4, true>(buf, buf_bytes, G, 16384, rows, out, "1KiB contiguous D=8  W=4 nt");
    run<2, 8, 8, false>(buf, buf_bytes, G, 16384, rows, out, "1KiB contiguous D=8  W=8");
  }
  return 0;
}
