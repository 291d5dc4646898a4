// LDS port throughput on gfx950 for the access patterns of the MLP chain kernels: per-lane 16-byte fragment reads (ds_read_b128,
// lane L at base + 16 L, two 4-KB planes for the two lane halves), with and without the slab staging writes (ds_write_b128), at one
// and two waves per SIMD.  Prints bytes per clock per CU (clock64 cycles of one workgroup's loop, so independent of the clock state).
// hipcc --offload-arch=gfx950 -O3 -o lds_probe r5_lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int READS, int WRITES, bool B64>
__global__ __launch_bounds__(256) void probe(unsigned* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned s[2][2][2048];      // two buffers x two planes x 8 KB
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 2 * 2 * 2048; i += 256) (&s[0][0][0])[i] = i * 2654435761u;
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int r = 0; r < READS; ++r) {
      if (B64) {
        const uint2 a = *reinterpret_cast<const uint2*>(&s[buf][h][(r * 32 + m) * 4]);
        const uint2 b = *reinterpret_cast<const uint2*>(&s[buf][h][(r * 32 + m) * 4 + 2]);
        acc.x ^= a.x; acc.y ^= a.y; acc.z ^= b.x; acc.w ^= b.y;
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&s[buf][h][(r * 32 + m) * 4]);
        acc ^= v;
      }
    }
#pragma unroll
    for (int w = 0; w < WRITES; ++w) *reinterpret_cast<u32x4*>(&s[buf ^ 1][w & 1][threadIdx.x * 4]) = acc;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int READS, int WRITES, bool B64>
void run(const char* name, int wgs_per_cu) {
  int iters = 20000, cus = 256;
  unsigned* out; long long* cyc;
  hipMalloc(&out, sizeof(unsigned) * 256 * cus * wgs_per_cu); hipMalloc(&cyc, sizeof(long long) * cus * wgs_per_cu);
  hipLaunchKernelGGL((probe<READS, WRITES, B64>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, cyc, 100);
  hipLaunchKernelGGL((probe<READS, WRITES, B64>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> c(cus * wgs_per_cu);
  hipMemcpy(c.data(), cyc, sizeof(long long) * c.size(), hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : c) mean += (double)v; mean /= c.size();
  const double bytes_per_wg_iter = 4.0 * 64 * 16 * (READS + 0.0) + 256.0 * 16 * WRITES;      // 4 waves x 64 lanes x 16 B per read; 256 threads x 16 B per write
  printf("%-44s %d workgroup(s) per CU: %7.1f cycles per iteration, %6.1f B/clk per CU\n", name, wgs_per_cu, mean / iters,
         bytes_per_wg_iter * wgs_per_cu / (mean / iters));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 2; ++w) {
    run<8, 0, false>("8 x ds_read_b128 per wave", w);
    run<8, 2, false>("8 x ds_read_b128 + 2 x ds_write_b128 (staging)", w);
    run<6, 2, false>("6 x ds_read_b128 + 2 x ds_write_b128 (block kernel)", w);
    run<8, 0, true>("8 x (2 x ds_read_b64) per wave", w);
    run<16, 0, false>("16 x ds_read_b128 per wave", w);
  }
  return 0;
}
