// selftest.hip -- on-device checks of the wave64 primitives the kernels rely on.  Cheap insurance:
// DPP control codes, ballot widths and MFMA fragment layouts are easy to get subtly wrong and
// cannot be exercised in the GPU-less build container.
#include <string>

#include "common.h"

namespace trase {

// out[0]: wave_sum result from lane 63; out[1]: broadcast sum seen by lane 0; out[2..3]: ballot of
// odd lanes; out[4]: popcount(lanemask_lt) summed over lanes (= 2016); out[5]: lane_id sum
__global__ void selftest_wave_kernel(float* outf, unsigned long long* outu) {
  const unsigned lane = threadIdx.x;
  const float v = (float)(lane + 1);            // sum = 2080
  const float s63 = wave_sum_lane63(v);
  if (lane == 63) outf[0] = s63;
  const float all = wave_sum_all(v);
  if (lane == 0) outf[1] = all;
  const unsigned long long bal = __ballot(lane & 1u);
  if (lane == 0) outu[0] = bal;
  float pc = (float)__popcll(lanemask_lt());
  pc = wave_sum_all(pc);
  if (lane == 0) outf[2] = pc;
  float li = wave_sum_all((float)lane_id());
  if (lane == 0) outf[3] = li;
  // scans: inclusive add of 1..64 -> lane l holds (l+1)(l+2)/2; inclusive mul of 2 on even lanes (1 on odd)
  const float sa = wave_scan_add(v);
  float bad = (sa == 0.5f * (float)((lane + 1) * (lane + 2))) ? 0.f : 1.f;
  const float sm = wave_scan_mul((lane & 1u) ? 1.0f : 1.03125f);
  float want = 1.0f;
  for (unsigned k = 0; k <= lane; k += 2) want *= 1.03125f;
  bad += (fabsf(sm - want) <= 1e-5f * want) ? 0.f : 1.f;
  const float sa2 = wave_scan_add_asm(v);
  bad += (sa2 == sa) ? 0.f : 1.f;
  const float sm2 = wave_scan_mul_asm((lane & 1u) ? 1.0f : 1.03125f);
  bad += (sm2 == sm) ? 0.f : 1.f;
  const float sh = wave_shr1(v, -7.0f);
  bad += (sh == (lane == 0 ? -7.0f : (float)lane)) ? 0.f : 1.f;
  bad = wave_sum_all(bad);
  if (lane == 0) outf[4] = bad;
}

// MFMA f32 32x32x2 layout probe: A[i][k] = i + 100k, B[k][j] = (j+1) * (k ? 0.5 : 1)
//   D[i][j] = (i)*(j+1) + (i+100)*(j+1)*0.5
__global__ void selftest_mfma_kernel(float* out /* 32*32 */) {
  typedef float v16f __attribute__((ext_vector_type(16)));
  const unsigned lane = threadIdx.x;
  const int i = lane & 31, k = lane >> 5;
  const float a = (float)i + 100.f * (float)k;
  const float b = (float)(i + 1) * (k ? 0.5f : 1.0f);
  v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  // documented layout: column = lane % 32, row = 8*(r/4) + 4*(lane/32) + r%4
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r / 4) + 4 * (int)(lane >> 5) + (r % 4);
    const int col = lane & 31;
    out[row * 32 + col] = acc[r];
  }
}

}  // namespace trase

using namespace trase;

extern "C" int trase_selftest(int32_t device, trase_stream_t stream_, char* msg, size_t msg_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  float* df = nullptr;
  unsigned long long* du = nullptr;
  float* dm = nullptr;
  TRASE_CHECK(hipMalloc((void**)&df, 16 * sizeof(float)));
  TRASE_CHECK(hipMalloc((void**)&du, 4 * sizeof(unsigned long long)));
  TRASE_CHECK(hipMalloc((void**)&dm, 1024 * sizeof(float)));
  hipLaunchKernelGGL(selftest_wave_kernel, dim3(1), dim3(64), 0, stream, df, du);
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, stream, dm);
  float hf[16];
  unsigned long long hu[4];
  static float hm[1024];
  int rc = check_hip(hipMemcpyAsync(hf, df, sizeof(hf), hipMemcpyDeviceToHost, stream), "selftest copy");
  if (!rc) rc = check_hip(hipMemcpyAsync(hu, du, sizeof(hu), hipMemcpyDeviceToHost, stream), "selftest copy");
  if (!rc) rc = check_hip(hipMemcpyAsync(hm, dm, sizeof(hm), hipMemcpyDeviceToHost, stream), "selftest copy");
  if (!rc) rc = check_hip(hipStreamSynchronize(stream), "selftest sync");
  hipFree(df); hipFree(du); hipFree(dm);
  if (rc) return rc;
  std::string report;
  int bad = 0;
  auto expect = [&](const char* what, double got, double want) {
    char line[160];
    const bool ok = got == want;
    snprintf(line, sizeof(line), "%s: got %.6g want %.6g %s\n", what, got, want, ok ? "ok" : "FAIL");
    report += line;
    if (!ok) ++bad;
  };
  expect("wave_sum lane63", hf[0], 2080.0);
  expect("wave_sum broadcast", hf[1], 2080.0);
  expect("ballot odd lanes", (double)(hu[0] == 0xAAAAAAAAAAAAAAAAull), 1.0);
  expect("lanemask_lt popcount sum", hf[2], 2016.0);
  expect("lane_id sum", hf[3], 2016.0);
  expect("wave scans (builtin + asm add/mul, shr1) bad lanes", hf[4], 0.0);
  int mfma_bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const float want = (float)i * (float)(j + 1) + ((float)i + 100.f) * ((float)(j + 1) * 0.5f);
      if (hm[i * 32 + j] != want) ++mfma_bad;
    }
  expect("mfma 32x32x2 f32 layout mismatches", mfma_bad, 0.0);
  if (msg && msg_bytes) {
    snprintf(msg, msg_bytes, "%s", report.c_str());
  }
  return bad;
}
