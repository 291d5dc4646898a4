// r5_mall_probe.hip -- does the 256 MiB Infinity Cache keep freshly WRITTEN data for a reader kernel that follows?
// (question behind the backward's dZ round trip: 1.23 GB written by the data chain, re-read by the weight-gradient GEMMs).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe profiles/experiments/r5_mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(uint4* p, size_t n16, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v o = {v, v + 1, v + 2, (unsigned)i};
    if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u4v*>(p + i)); else *reinterpret_cast<u4v*>(p + i) = o;
  }
}
template <bool REV>
__global__ __launch_bounds__(256) void read_kernel(const uint4* p, size_t n16, unsigned* out) {
  unsigned s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = p[REV ? n16 - 1 - i : i];
    s += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (s == 0x12345678u) *out = s;
}

int main() {
  const size_t MB = 1024 * 1024;
  const size_t maxb = 2048 * MB;
  uint4 *a, *b; unsigned* out;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  auto time_read = [&](const uint4* p, size_t bytes, bool rev) {
    CK(hipEventRecord(e0));
    if (rev) read_kernel<true><<<grid, 256>>>(p, bytes / 16, out); else read_kernel<false><<<grid, 256>>>(p, bytes / 16, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
  };
  auto time_write = [&](uint4* p, size_t bytes, bool nt) {
    CK(hipEventRecord(e0));
    if (nt) write_kernel<true><<<grid, 256>>>(p, bytes / 16, 7u); else write_kernel<false><<<grid, 256>>>(p, bytes / 16, 7u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
  };
  auto flush = [&]() { time_write(b, maxb, false); time_read(b, maxb, false); };   // 2 GiB of other traffic
  printf("size_MB  cold_read  write  read_after_write  read_after_write_nt  read_after_write_rev  read_after_read  write_then_pollute154_then_read   (TB/s)\n");
  const size_t sizes[] = {32, 64, 96, 128, 154, 192, 224, 256, 320, 384, 512, 1024};
  for (size_t s : sizes) {
    const size_t bytes = s * MB;
    double r[8];
    auto tb = [&](float ms) { return bytes / (ms * 1e-3) / 1e12; };
    for (int rep = 0; rep < 2; ++rep) {   // second repetition reported
      flush(); r[0] = tb(time_read(a, bytes, false));
      flush(); r[1] = tb(time_write(a, bytes, false)); r[2] = tb(time_read(a, bytes, false));
      flush(); time_write(a, bytes, true); r[3] = tb(time_read(a, bytes, false));
      flush(); time_write(a, bytes, false); r[4] = tb(time_read(a, bytes, true));
      flush(); time_read(a, bytes, false); r[5] = tb(time_read(a, bytes, false));
      flush(); time_write(a, bytes, false); time_read(b, 154 * MB, false); r[6] = tb(time_read(a, bytes, false));
    }
    printf("%6zu  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f\n", s, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
  }
  return 0;
}
