// Synthetic access patterns with a KNOWN number of distinct 128-byte lines, for calibrating the
// rocprofv3 HBM counters (FETCH_SIZE = TCC_EA0_RDREQ based) on the access shapes of k_tab_pwg's
// global-memory waves: 2-byte per-lane gathers (model digest), random 16-byte rows (Q), 4 / 8-byte
// records (model, digest rows) — MI355X_MICROARCH.md only calibrates wide coalesced streams
// ("other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Test infrastructure of scripts/pmc_calibrate_gather.py, not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// every lane reads 16 consecutive bytes, a wave 1 KiB, the grid the whole table once
__global__ void calib_stream16(const uint4* __restrict__ t, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16;
       k += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = t[k];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// `per_lane` loads of W bytes per lane; load number j of the grid goes to line (j * odd) mod
// n_lines — a bijection for a power-of-two line count — at a W-aligned offset inside the line that
// depends on j: every 128-byte line of the table is touched by exactly ONE load of the launch, and
// the 64 lanes of a load instruction go to 64 lines far apart.
template <typename T>
__global__ void calib_touch(const unsigned char* __restrict__ t, uint32_t n_lines_log2,
                            uint32_t per_lane, uint32_t* sink) {
  const uint64_t first = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * per_lane;
  const uint64_t mask = ((uint64_t)1 << n_lines_log2) - 1;
  uint32_t acc = 0;
  for (uint32_t k = 0; k < per_lane; ++k) {
    const uint64_t j = first + k;
    const uint64_t line = (j * 2654435761ull) & mask;
    const uint32_t off = (uint32_t)((j * 40503u) >> 3) & (128u / sizeof(T) - 1u);
    const T* p = reinterpret_cast<const T*>(t + line * 128) + off;
    const T v = *p;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    if (sizeof(T) >= 4) {
      for (unsigned c = 0; c < sizeof(T) / 4; ++c) acc ^= w[c];
    } else {
      acc ^= (uint32_t)*reinterpret_cast<const uint16_t*>(&v);
    }
  }
  // (a value the accumulated bits can take, whatever the width: the loads stay)
  if (acc == (sizeof(T) >= 4 ? 0x12345678u : 0x1234u)) *sink = acc;
}

// ONE workgroup: byte 0 of each of `lines` lines (a region that fits the L2 of its XCD), a wait,
// then byte 64 of the same lines.  If the second pass sends no read to the fabric, a miss fills the
// whole 128-byte line; if it sends as many as the first, fills are 64-byte sectors.
__global__ void calib_halves(const unsigned char* __restrict__ t, uint32_t lines, uint32_t second,
                             uint32_t* sink) {
  uint32_t acc = 0;
  for (uint32_t l = threadIdx.x; l < lines; l += blockDim.x)
    acc ^= *reinterpret_cast<const uint32_t*>(t + (size_t)l * 128);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (second)
    for (uint32_t l = threadIdx.x; l < lines; l += blockDim.x)
      acc ^= *reinterpret_cast<const uint32_t*>(t + (size_t)l * 128 + 64);
  if (acc == 0x12345678u) *sink = acc;
}

struct row16 {
  uint32_t a, b, c, d;
};
struct rec8 {
  uint32_t a, b;
};

}  // namespace

extern "C" __attribute__((visibility("default"))) int calib_run(int pattern, const void* table,
                                                               unsigned long long bytes,
                                                               void* sink, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const unsigned char* t = static_cast<const unsigned char*>(table);
  uint32_t* s = static_cast<uint32_t*>(sink);
  uint32_t lg = 0;
  while (((unsigned long long)128 << (lg + 1)) <= bytes) ++lg;   // lines = 2^lg
  const uint32_t per_lane = 16;
  const unsigned long long loads = (unsigned long long)1 << lg;
  const unsigned grid = (unsigned)(loads / per_lane / 256);
  switch (pattern) {
    case 0: hipLaunchKernelGGL(calib_stream16, dim3(8192), dim3(256), 0, st,
                               static_cast<const uint4*>(table), (size_t)(bytes / 16), s); break;
    case 1: hipLaunchKernelGGL(calib_touch<uint16_t>, dim3(grid), dim3(256), 0, st, t, lg, per_lane, s); break;
    case 2: hipLaunchKernelGGL(calib_touch<uint32_t>, dim3(grid), dim3(256), 0, st, t, lg, per_lane, s); break;
    case 3: hipLaunchKernelGGL(calib_touch<rec8>, dim3(grid), dim3(256), 0, st, t, lg, per_lane, s); break;
    case 4: hipLaunchKernelGGL(calib_touch<row16>, dim3(grid), dim3(256), 0, st, t, lg, per_lane, s); break;
    case 5: hipLaunchKernelGGL(calib_halves, dim3(1), dim3(256), 0, st, t, 4096u, 0u, s); break;
    case 6: hipLaunchKernelGGL(calib_halves, dim3(1), dim3(256), 0, st, t, 4096u, 1u, s); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
