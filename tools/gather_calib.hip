// gather_calib -- calibration of rocprofv3's FETCH_SIZE on the access pattern of k_search's scoring step (dev tool).
//
// The guide's "x2" correction of FETCH_SIZE on gfx950 was measured on wide coalesced STREAMING reads and is declared
// uncalibrated for other patterns (MI355X_MICROARCH.md, section HBM).  k_search's bytes are a random GATHER of 512-byte point rows
// (a lane pair per row, 16 bytes per lane and load, 16 loads per lane) plus 256-byte adjacency rows (4 bytes per lane).  This
// program issues exactly that, with a known byte count, from a buffer far larger than the 256 MiB Infinity Cache:
//
//   gather_calib <buffer_GiB> <rows_per_launch> <row_bytes> <launches> [hot_MiB]      (row_bytes 4: single-dword random probes)
//
// rows are drawn uniformly from the whole buffer -- or, with hot_MiB > 0, from a window of that size (to see what the counter
// does when the working set fits the L2 / the Infinity Cache).  Prints one JSON line: bytes per launch (rows x row_bytes), mean
// launch time from HIP events, GB/s.  Run it under `rocprofv3 --pmc FETCH_SIZE` (and, in a second pass, `--pmc TCC_HIT_sum
// TCC_MISS_sum`) and divide: profiles/r04_fetch_size_calibration.json holds the result.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));             \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// one wave = 32 rows per step (a lane pair per row, lane h takes the 16-byte pieces 2t + h), like l2_pair_ct / mips_pair_ct
template <int PIECES>  // 16-byte pieces per lane: 16 = 512-B rows, 12 = 384-B, 14 = 448-B
__global__ __launch_bounds__(256) void k_gather(const float4 *buf, uint64_t nrows_buf, uint64_t rows, uint64_t seed, float *sink) {
  const int lane = threadIdx.x & 63, h = lane & 1;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  float acc = 0.f;
  for (uint64_t r0 = wave * 32; r0 < rows; r0 += nwaves * 32) {
    const uint64_t r = r0 + (lane >> 1);
    const uint64_t row = mix(seed + r) % nrows_buf;
    const float4 *p = buf + row * (2 * PIECES) + h;
    float4 v[PIECES];
#pragma unroll
    for (int t = 0; t < PIECES; t++) v[t] = p[2 * t];
#pragma unroll
    for (int t = 0; t < PIECES; t++) acc += v[t].x + v[t].y + v[t].z + v[t].w;
  }
  if (acc == 1234.5f) sink[0] = acc;  // (never true for the zero-filled buffer: keeps the loads alive)
}

// adjacency rows: one coalesced 256-byte row per wave and step (4 bytes per lane)
__global__ __launch_bounds__(256) void k_gather_rows256(const int *buf, uint64_t nrows_buf, uint64_t rows, uint64_t seed, int *sink) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  int acc = 0;
  for (uint64_t r0 = wave * 8; r0 < rows; r0 += nwaves * 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = buf[(mix(seed + r0 + u) % nrows_buf) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  if (acc == 12345) sink[0] = acc;
}

// filter probes: every lane reads ONE dword at a random address of its own (64 different 128-byte lines per wave-instruction) --
// the seen-filter probes of a mid-beam search (a table of 4 << bits bytes per search: 2 MiB at beam 1 280)
__global__ __launch_bounds__(256) void k_probe4(const int *buf, uint64_t nwords, uint64_t probes, uint64_t seed, int *sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
  int acc = 0;
  for (uint64_t r0 = t * 8; r0 < probes; r0 += nt * 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = buf[mix(seed + r0 + u) % nwords];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u];
  }
  if (acc == 12345) sink[0] = acc;
}

int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 2.0;
  const uint64_t rows = argc > 2 ? strtoull(argv[2], nullptr, 10) : 34000000ull;
  const int row_bytes = argc > 3 ? atoi(argv[3]) : 512;
  const int launches = argc > 4 ? atoi(argv[4]) : 10;
  const double hot_mib = argc > 5 ? atof(argv[5]) : 0.0;
  const size_t bytes = (size_t)(gib * (double)(1ull << 30));
  void *buf = nullptr, *sink = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipMalloc(&sink, 64));
  uint64_t nrows_buf = bytes / (size_t)row_bytes;
  if (hot_mib > 0) nrows_buf = (uint64_t)(hot_mib * 1048576.0) / (uint64_t)row_bytes;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 8;  // two waves per SIMD, as k_search runs
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto launch = [&](int i) {
    const uint64_t seed = 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1);
    if (row_bytes == 512) hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, nullptr, (const float4 *)buf, nrows_buf, rows, seed, (float *)sink);
    else if (row_bytes == 384) hipLaunchKernelGGL(k_gather<12>, dim3(blocks), dim3(256), 0, nullptr, (const float4 *)buf, nrows_buf, rows, seed, (float *)sink);
    else if (row_bytes == 448) hipLaunchKernelGGL(k_gather<14>, dim3(blocks), dim3(256), 0, nullptr, (const float4 *)buf, nrows_buf, rows, seed, (float *)sink);
    else if (row_bytes == 256) hipLaunchKernelGGL(k_gather_rows256, dim3(blocks), dim3(256), 0, nullptr, (const int *)buf, nrows_buf, rows, seed, (int *)sink);
    else if (row_bytes == 4) hipLaunchKernelGGL(k_probe4, dim3(blocks), dim3(256), 0, nullptr, (const int *)buf, nrows_buf, rows, seed, (int *)sink);
    else {
      fprintf(stderr, "row_bytes must be 512, 448, 384, 256 or 4 (single-dword probes)\n");
      exit(2);
    }
  };
  launch(-1);  // warm-up (page tables)
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < launches; i++) launch(i);
  CHECK(hipEventRecord(e1, nullptr));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per = (double)ms / launches;
  printf("{\"pattern\": \"random gather of %d-byte rows, 16 B (4 B for 256-byte rows) per lane and load\", \"buffer_gib\": %.2f, \"hot_mib\": %.1f, "
         "\"rows_per_launch\": %llu, \"bytes_per_launch\": %llu, \"launches\": %d, \"ms_per_launch\": %.4f, \"gb_per_s\": %.1f}\n",
         row_bytes, gib, hot_mib, (unsigned long long)rows, (unsigned long long)(rows * (uint64_t)row_bytes), launches, per,
         (double)rows * row_bytes / (per * 1e-3) / 1e9);
  return 0;
}
