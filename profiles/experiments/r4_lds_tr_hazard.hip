// r4_lds_tr_hazard.hip -- stand-alone reproducer (no torch, no library) for what scratch/two-stream runs of the rasterizer showed
// in round 4: while a wave on a CU executes ds_read_b64_tr_b16 (the transposing LDS read of gfx950), an ordinary ds_read_b32 of
// ANOTHER wave on that CU (another kernel, another HIP stream) can return wrong data in lanes 48..63.
//
//   hipcc --offload-arch=gfx950 -O2 -o r4_lds_tr_hazard r4_lds_tr_hazard.hip && ./r4_lds_tr_hazard
//
// victim: one wave per workgroup fills a wave-private LDS slab with a known pattern and reads it back with several access
// patterns, counting wrong words per lane.  aggressor: workgroups that loop over LDS reads, either transposing or plain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <bool TR>
__global__ __launch_bounds__(128) void aggressor(int iters, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[7424];      // 14 848 bytes, like the forward's tile
  for (int i = threadIdx.x; i < 7424; i += 128) tile[i] = (unsigned short)(i * 7 + blockIdx.x);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // 16 lanes cooperate on a 16 x 4 block: row = lane & 15, 8-byte column block = lane >> 4, row pitch 192 halves (384 bytes)
  const unsigned short* p = tile + (lane & 15) * 192 + (lane >> 4) * 4 + (threadIdx.x >> 6) * 3200;
  uint32_t acc = 0;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  f32x16 D = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      s16x4 v, w;
      if (TR) { v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * k)); w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * k + 16)); }
      else { v = *(const lds_s16x4*)(p + 16 * k); w = *(const lds_s16x4*)(p + 16 * k + 16); }
      const s16x8 q = {v[0], v[1], v[2], v[3], w[0], w[1], w[2], w[3]};
      const bf16x8 B = __builtin_bit_cast(bf16x8, q);
      D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, B, D, 0, 0, 0);      // the fragments feed the matrix pipe, as in the forward
    }
  }
  for (int k = 0; k < 16; ++k) acc += __float_as_uint(D[k]);
  if (acc == 0x12345678u) sink[0] = acc;
}

// patterns: 0 = b32 at a 45-word lane stride (the rasterizer's SH slab), 1 = b32 at stride 1, 2 = fp32 fmacs while 45 LDS reads are outstanding, 3 = 80 live registers incremented in place
__global__ __launch_bounds__(64) void victim(int iters, unsigned long long* bad /* [4][64] */) {
  __shared__ __attribute__((aligned(16))) uint32_t slab[2880];             // 11 520 bytes
  const int lane = threadIdx.x;
  unsigned long long nbad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const uint32_t salt = blockIdx.x * 2654435761u + it * 40503u;
    for (int i = lane; i < 2880; i += 64) slab[i] = salt + (uint32_t)i * 2246822519u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t v[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) v[k] = ((volatile uint32_t*)slab)[lane * 45 + k];
#pragma unroll
    for (int k = 0; k < 45; ++k) nbad[0] += v[k] != salt + (uint32_t)(lane * 45 + k) * 2246822519u;
#pragma unroll
    for (int k = 0; k < 45; ++k) v[k] = ((volatile uint32_t*)slab)[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 45; ++k) nbad[1] += v[k] != salt + (uint32_t)(k * 64 + lane) * 2246822519u;
    {
      // fp32 fmacs on live registers WHILE 45 LDS reads of this wave are still outstanding (the preprocess kernel evaluates the
      // SH basis while its coefficient reads are in flight), under a partial exec mask
      const bool act2 = (((salt >> 5) + lane * 2246822519u) >> 11 & 3u) != 0u;
      if (act2) {
        uint32_t w[45];
#pragma unroll
        for (int k = 0; k < 45; ++k) w[k] = slab[lane * 45 + k];
        float r[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) { r[k] = (float)((lane + k) & 15); asm volatile("" : "+v"(r[k])); }
#pragma unroll
        for (int rep = 0; rep < 6; ++rep) {
#pragma unroll
          for (int k = 0; k < 24; ++k) { r[k] = fmaf(r[k], 1.0f, 3.0f); asm volatile("" : "+v"(r[k])); }
        }
        uint32_t ok = 0;
#pragma unroll
        for (int k = 0; k < 45; ++k) ok += w[k] == salt + (uint32_t)(lane * 45 + k) * 2246822519u;
        nbad[2] += 45 - ok;
#pragma unroll
        for (int k = 0; k < 24; ++k) nbad[2] += r[k] != (float)((lane + k) & 15) + 18.0f;
      }
    }
    {
      // registers only: 80 live counters, each incremented 64 times by its own constant; a lost write, a foreign write or a
      // stale read shows as a wrong final value
      // (under a PARTIAL exec mask, like the colour of the visible Gaussians in the preprocess kernel, and with fp32 fmac
      // against literals, like its SH basis)
      const bool active = (((salt >> 7) + lane * 2654435761u) >> 13 & 3u) != 0u;
      if (active) {
        float r[80];
#pragma unroll
        for (int k = 0; k < 80; ++k) { r[k] = (float)((lane + k) & 15); asm volatile("" : "+v"(r[k])); }
        for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
          for (int k = 0; k < 80; ++k) { r[k] = fmaf(r[(k + 1) % 80 < 80 ? k : k], 1.0f, 3.0f); asm volatile("" : "+v"(r[k])); }
        }
#pragma unroll
        for (int k = 0; k < 80; ++k) nbad[3] += r[k] != (float)((lane + k) & 15) + 192.0f;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int p = 0; p < 4; ++p) if (nbad[p]) atomicAdd(&bad[p * 64 + lane], nbad[p]);
}

static void run(const char* label, int mode /* 0 none, 1 plain, 2 transposing */, hipStream_t sa, hipStream_t sb, unsigned long long* d_bad, uint32_t* sink) {
  CK(hipMemset(d_bad, 0, sizeof(unsigned long long) * 256));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, sa));
  for (int rep = 0; rep < 10; ++rep) {
    if (mode == 1) hipLaunchKernelGGL(aggressor<false>, dim3(32768), dim3(128), 0, sb, 300, sink);
    if (mode == 2) hipLaunchKernelGGL(aggressor<true>, dim3(32768), dim3(128), 0, sb, 300, sink);
    hipLaunchKernelGGL(victim, dim3(16384), dim3(64), 0, sa, 16, d_bad);
  }
  CK(hipEventRecord(e1, sa));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s victim stream: %.2f ms for 10 launches\n", label, ms);
  unsigned long long h[256];
  CK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
  const char* names[4] = {"b32 stride 45", "b32 stride 1", "fmac under reads", "80 registers"};
  for (int p = 0; p < 4; ++p) {
    unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += h[p * 64 + l]; q[l >> 4] += h[p * 64 + l]; }
    printf("%-28s victim %-14s wrong words %10llu   by lane quarter [%llu %llu %llu %llu]\n", label, names[p], tot, q[0], q[1], q[2], q[3]);
  }
}

int main() {
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  unsigned long long* d_bad; uint32_t* sink;
  CK(hipMalloc((void**)&d_bad, sizeof(unsigned long long) * 256)); CK(hipMalloc((void**)&sink, 64));
  run("alone", 0, sa, sb, d_bad, sink);
  run("beside plain ds_read_b64", 1, sa, sb, d_bad, sink);
  run("beside ds_read_b64_tr_b16", 2, sa, sb, d_bad, sink);
  run("alone again", 0, sa, sb, d_bad, sink);
  return 0;
}
