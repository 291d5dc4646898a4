// binning.hip -- builds the per-tile, depth-ordered Gaussian lists.
//
// Replaces the lineage's duplicateWithKeys + 64-bit cub::DeviceRadixSort +
// identifyTileRanges (SURVEY.md 2.2 kernel inventory) with a decomposition that
// moves ~4x fewer bytes and yields the *identical* order:
//   1. stable radix sort of the P Gaussians by float32 depth bits (ties keep
//      ascending Gaussian index)                       -- 4 passes over P items
//   2. inclusive scan of tiles-touched in depth-rank order
//   3. emit (tile, id) pairs rank-major                 -- R pairs, already depth-ordered
//   4. stable radix sort of the pairs by tile id only   -- 2 passes over R items
// A stable sort by tile of a depth-ordered sequence is exactly the lineage's
// (tile<<32 | depth) order.  All kernels take the pair count from device memory
// so the host never has to synchronise.
#include <stdlib.h>

#include "common.h"

namespace trase {

constexpr int RS_THREADS = 256;
#ifndef TRASE_RS_ITEMS
#define TRASE_RS_ITEMS 8
#endif
constexpr int RS_ITEMS = TRASE_RS_ITEMS;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // 2048 items per workgroup
constexpr int RS_SEG = WAVE * RS_ITEMS;          // 512 contiguous items per wave

__device__ __forceinline__ uint32_t dev_n(const uint32_t* n_ptr, uint32_t cap) {
  const uint32_t n = *n_ptr;
  return n < cap ? n : cap;
}

// ---- pass kernel 1: per-workgroup digit histograms --------------------------------------------
// DB = digit bits of a pass: 8 (the pair sort, KNN, 32-bit depth keys) or 9 (the default 27-bit depth keys in 3 passes)
template <int DB>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                                int shift, uint32_t mask, uint32_t* __restrict__ hist,
                                                                uint32_t* __restrict__ digit_total, int nb_max,
                                                                uint32_t flag_key = 0u, uint32_t* __restrict__ flag_word = nullptr,
                                                                uint32_t* __restrict__ zero_from = nullptr, uint32_t zero_words = 0u) {
  constexpr int ND = 1 << DB;
  // short sorts: the histograms of the LATER passes are filled by the scatter kernels with atomics -- cleared here (grid-stride,
  // by every workgroup of the launch, also the ones behind n)
  for (uint32_t i = blockIdx.x * RS_THREADS + threadIdx.x; i < zero_words; i += gridDim.x * RS_THREADS) zero_from[i] = 0u;
  const uint32_t n = dev_n(n_ptr, cap);
  const uint32_t base = blockIdx.x * RS_TILE;
  if (base >= n) return;
  __shared__ uint32_t h[ND];
#pragma unroll
  for (int d = threadIdx.x; d < ND; d += RS_THREADS) h[d] = 0;
  __syncthreads();
  // Keys arrive in runs (consecutive pairs of one Gaussian share the high bits of the sub-tile id):
  // aggregate runs inside the wave so that a 64-lane run costs one LDS atomic instead of 64 serialized ones.
  const unsigned lane = threadIdx.x & 63;
  // all keys of the thread are requested before any is used: with the load inside the loop the compiler waits for each one in
  // turn (the LDS atomics in between pin the order) -- eight dependent round trips per workgroup instead of one
  uint32_t kk[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
    kk[i] = idx < n ? keys[idx] : 0u;
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
    const bool valid = idx < n;
    if (flag_word && valid && kk[i] == flag_key) atomicOr(flag_word, 2u);      // (the depth sort's saturated-key watch: first pass only)
    const uint32_t dg = valid ? ((kk[i] >> shift) & mask) : 0xffffffffu;
    const uint32_t prev = (uint32_t)__shfl_up((int)dg, 1);
    const bool start = valid && (lane == 0 || prev != dg);
    const unsigned long long starts = __ballot(start);
    const unsigned long long vmask = __ballot(valid);
    if (start) {
      const unsigned long long later = (lane == 63) ? 0ull : (starts >> (lane + 1));
      const unsigned nvalid = (unsigned)__popcll(vmask);           // valid lanes are a prefix of the wave
      const unsigned next = later ? (lane + 1 + (unsigned)__builtin_ctzll(later)) : nvalid;
      atomicAdd(&h[dg], next - lane);
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = threadIdx.x; d < ND; d += RS_THREADS) hist[(size_t)d * nb_max + blockIdx.x] = h[d];
}

// inclusive scan over the 256 threads of a workgroup: a DPP scan inside every wave, then the three lower waves'
// totals through LDS (two barriers instead of the sixteen of a Hillis-Steele scan in LDS)
__device__ __forceinline__ uint32_t block256_incl_scan(uint32_t v, uint32_t* part /* [4] in LDS */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) part[wave] = x;
  __syncthreads();
  uint32_t add = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) add += (w < wave) ? part[w] : 0u;
  __syncthreads();
  return x + add;
}

// ---- pass kernel 2: one workgroup per digit scans its row of the histogram --------------------
__global__ __launch_bounds__(256) void radix_scan_kernel(const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                         uint32_t* __restrict__ hist,
                                                         uint32_t* __restrict__ digit_total, int nb_max) {
  const uint32_t n = dev_n(n_ptr, cap);
  const int nb = (int)((n + RS_TILE - 1) / RS_TILE);
  const int d = blockIdx.x;
  __shared__ uint32_t sh[256];
  __shared__ uint32_t carry;
  // exclusive scan of the row over the workgroups; the row total (= number of items with this digit) goes to
  // digit_total, whose scan over the digits is done by the scatter workgroups themselves
  if (threadIdx.x == 0) carry = 0u;
  __syncthreads();
  uint32_t* row = hist + (size_t)d * nb_max;
  // every thread owns a contiguous piece of the row: serial sum, one block-wide scan of the 256 sums, serial rewrite
  const int per = (nb + 255) / 256;
  const int b_lo = threadIdx.x * per, b_hi = min(b_lo + per, nb);
  uint32_t mine = 0, run;
  uint32_t incl;
  if (per <= 32) {
    // the thread's piece in registers, every load in flight at once (two loops of `per` dependent round trips were most of
    // this kernel's 12 us on the 6 M-pair sort)
    uint32_t v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = (b_lo + u < b_hi) ? row[b_lo + u] : 0u;
#pragma unroll
    for (int u = 0; u < 32; ++u) mine += v[u];
    incl = block256_incl_scan(mine, sh);
    run = incl - mine;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (b_lo + u < b_hi) row[b_lo + u] = run;
      run += v[u];
    }
  } else {
    for (int b = b_lo; b < b_hi; ++b) mine += row[b];
    incl = block256_incl_scan(mine, sh);
    run = incl - mine;
    for (int b = b_lo; b < b_hi; ++b) { const uint32_t v = row[b]; row[b] = run; run += v; }
  }
  if (threadIdx.x == 255) carry = incl;
  __syncthreads();
  if (threadIdx.x == 0) digit_total[d] = carry;
}

// ---- pass kernel 3: stable scatter -------------------------------------------------------------
// Each wave owns a contiguous 512-item segment and ranks it 64 items at a time with ballots:
// lanes holding the same digit find each other through DB bit-plane ballots; the rank inside the
// round is the number of lower lanes in the peer set, the rank across rounds comes from a
// per-wave running counter in LDS.
// SMALL (short sorts, see RS_SMALL_NB): `hist` holds the RAW per-workgroup counts of this pass -- every workgroup sums its digits' rows
// itself (counts of the workgroups before it, and of all); `hist_next` (when not null) receives the counts of the next pass's digit
// at the items' destinations
template <bool IOTA, int DB, bool SMALL = false>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t cap, int shift, uint32_t mask,
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ digit_total, int nb_max,
    uint32_t* __restrict__ hist_next = nullptr, int shift_next = 0, uint32_t mask_next = 0u) {
  constexpr int ND = 1 << DB;
  constexpr int DPT = ND / RS_THREADS;          // digits per thread in the per-digit steps: thread t owns [t DPT, (t + 1) DPT)
  const uint32_t n = dev_n(n_ptr, cap);
  const uint32_t base = blockIdx.x * RS_TILE;
  if (base >= n) return;
  __shared__ uint32_t wcount[RS_WAVES][ND];     // per-wave digit counts, then (in place) the counts of the earlier waves
  const int wave = threadIdx.x >> 6;
  const unsigned lane = threadIdx.x & 63;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w)
#pragma unroll
    for (int d = threadIdx.x; d < ND; d += RS_THREADS) wcount[w][d] = 0;
  __syncthreads();
  uint32_t k[RS_ITEMS], v[RS_ITEMS], rk[RS_ITEMS];
  const unsigned long long lt = lanemask_lt();
  volatile uint32_t* cnt = wcount[wave];
  // (thread t: the global totals of its digits and this workgroup's offsets inside them -- requested now, used after the ranking)
  uint32_t my_digit_total[DPT], my_hist[DPT];
  if constexpr (SMALL) {
    const int nb = (int)((n + RS_TILE - 1) / RS_TILE);
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      const uint32_t* row = hist + (size_t)(threadIdx.x * DPT + j) * nb_max;
      uint32_t before = 0, total = 0;
      for (int b0 = 0; b0 < nb; b0 += 8) {            // eight loads in flight
        uint32_t v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (b0 + u < nb) ? row[b0 + u] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) { total += v8[u]; before += (b0 + u < (int)blockIdx.x) ? v8[u] : 0u; }
      }
      my_digit_total[j] = total; my_hist[j] = before;
    }
  } else {
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    my_digit_total[j] = digit_total[threadIdx.x * DPT + j];
    my_hist[j] = hist[(size_t)(threadIdx.x * DPT + j) * nb_max + blockIdx.x];
  }
  }
  // every key and value of the thread is requested up front: inside the ranking loop (volatile LDS counters, wave barriers)
  // the compiler waited for each round's pair of loads before ranking it -- eight dependent round trips per workgroup
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint32_t idx = base + wave * RS_SEG + i * WAVE + lane;
    const bool valid = idx < n;
    k[i] = valid ? keys_in[idx] : 0xffffffffu;
    v[i] = valid ? (IOTA ? idx : vals_in[idx]) : 0u;
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint32_t idx = base + wave * RS_SEG + i * WAVE + lane;
    const bool valid = idx < n;
    const uint32_t dg = (k[i] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
    const uint32_t dg0 = __builtin_amdgcn_readfirstlane(dg);
    if (!__all(!valid || dg == dg0)) {      // fast path: every valid lane of the round holds the same digit
#pragma unroll
      for (int bit = 0; bit < DB; ++bit) {
        const bool set = (dg >> bit) & 1u;
        const unsigned long long bal = __ballot(set);
        peers &= set ? bal : ~bal;
      }
    }
    const uint32_t in_round = (uint32_t)__popcll(peers & lt);
    const uint32_t total = (uint32_t)__popcll(peers);
    uint32_t old = 0;
    if (valid) {
      old = cnt[dg];                         // every peer reads before the leader's update (in-order LDS)
      if (in_round == 0) cnt[dg] = old + total;
    }
    rk[i] = old + in_round;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // Block-local layout: items are first placed in LDS in their sorted order inside the block (digit-major, then
  // wave, then rank), then written out by consecutive threads -- a digit's run inside the block is contiguous in
  // the output too, so the stores are coalesced runs instead of 4-byte writes scattered over the whole array.
  __shared__ uint32_t gbase[ND], lstart[ND];
  __shared__ uint32_t st_k[RS_TILE], st_v[RS_TILE];
  __shared__ uint32_t part_a[4], part_b[4];
  {   // two exclusive scans over the digits: this block's counts -> LDS layout, the global digit totals -> number of
      // items with a smaller digit.  A thread's DPT digits are consecutive: serial inside the thread, one block scan across.
    uint32_t cnt_d[DPT], sum_l = 0, sum_g = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      const int d = threadIdx.x * DPT + j;
      uint32_t tot = 0;
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) {
        const uint32_t c = wcount[w][d];
        wcount[w][d] = tot;                          // items of digit d in earlier waves of this block
        tot += c;
      }
      cnt_d[j] = tot;
      sum_l += tot;
      sum_g += my_digit_total[j];
    }
    uint32_t run_l = block256_incl_scan(sum_l, part_a) - sum_l;
    uint32_t run_g = block256_incl_scan(sum_g, part_b) - sum_g;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      const int d = threadIdx.x * DPT + j;
      lstart[d] = run_l;
      gbase[d] = run_g + my_hist[j];
      run_l += cnt_d[j];
      run_g += my_digit_total[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint32_t idx = base + wave * RS_SEG + i * WAVE + lane;
    if (idx < n) {
      const uint32_t dg = (k[i] >> shift) & mask;
      const uint32_t lp = lstart[dg] + wcount[wave][dg] + rk[i];
      st_k[lp] = k[i];
      st_v[lp] = v[i];
    }
  }
  __syncthreads();
  const uint32_t nblk = min((uint32_t)RS_TILE, n - base);
  for (uint32_t j = threadIdx.x; j < nblk; j += RS_THREADS) {
    const uint32_t kk = st_k[j];
    const uint32_t dg = (kk >> shift) & mask;
    const uint32_t dst = gbase[dg] + (j - lstart[dg]);
    keys_out[dst] = kk;
    vals_out[dst] = st_v[j];
    if constexpr (SMALL) {
      if (hist_next) {
        // consecutive items of the block's sorted order mostly share (next digit, destination workgroup) -- depth keys of a small
        // scene agree in their upper digits, sub-tile ids come in runs: one atomic per RUN inside the wave (as radix_hist does),
        // not per item (1 000 atomics on ONE word cost 25 us)
        const uint32_t word = ((kk >> shift_next) & mask_next) * (uint32_t)nb_max + dst / RS_TILE;
        const uint32_t prev = (uint32_t)__shfl_up((int)word, 1);
        const bool start = lane == 0 || prev != word;
        const unsigned long long starts = __ballot(start), act = __ballot(true);
        if (start) {
          const unsigned long long later = (lane == 63) ? 0ull : (starts >> (lane + 1));
          const unsigned nact = (unsigned)__popcll(act);               // active lanes are a prefix of the wave (j ascending)
          const unsigned next = later ? (lane + 1 + (unsigned)__builtin_ctzll(later)) : nact;
          atomicAdd(hist_next + word, next - lane);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void zero_bytes_kernel(uint8_t* __restrict__ p, size_t bytes) {
  // 16-byte stores over the aligned middle, single bytes at the ragged ends
  const size_t head = ((size_t)(-(intptr_t)p) & 15u) < bytes ? ((size_t)(-(intptr_t)p) & 15u) : bytes;
  const size_t n16 = (bytes - head) >> 4;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  const size_t tail0 = head + (n16 << 4);
  if (t < head) p[t] = 0;
  if (t < bytes - tail0) p[tail0 + t] = 0;
}
int launch_zero_bytes(void* p, size_t bytes, hipStream_t stream) {
  if (!p || bytes == 0) return TRASE_OK;
  const size_t n16 = bytes >> 4;
  const unsigned blocks = (unsigned)(n16 / 256 / 4 + 1 < 4096 ? n16 / 256 / 4 + 1 : 4096);
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(blocks), dim3(256), 0, stream, (uint8_t*)p, bytes);
  return TRASE_OK;
}

// ---- two views in one depth sort (trase_rast_forward_raw_pair) -----------------------------------------------------------
__global__ void fill_u32_kernel(uint32_t* p, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }
int launch_fill_u32(uint32_t* p, uint32_t v, hipStream_t stream) {
  hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, stream, p, v);
  return TRASE_OK;
}
// the sorted (view, depth) order of 2 P keys back into the two views' own id lists: entries [0, P) are view 0's Gaussians in
// depth order, entries [P, 2 P) view 1's with P added to their ids
__global__ __launch_bounds__(256) void split_pair_ids_kernel(const uint32_t* __restrict__ sorted, int P, uint32_t* __restrict__ ids0,
                                                             uint32_t* __restrict__ ids1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * P) return;
  const uint32_t v = sorted[i];
  if (i < P) ids0[i] = v; else ids1[i - P] = v - (uint32_t)P;
}
int launch_split_pair_ids(const LaunchCtx& c, const uint32_t* sorted, int P, uint32_t* ids0, uint32_t* ids1) {
  ProfScope ps("split_pair_ids", c.stream);
  hipLaunchKernelGGL(split_pair_ids_kernel, dim3((2 * P + 255) / 256), dim3(256), 0, c.stream, sorted, P, ids0, ids1);
  return TRASE_OK;
}

int radix_passes(int bit_lo, int bit_hi, int digit_bits) { return (bit_hi - bit_lo + digit_bits - 1) / digit_bits; }

// digit_bits 8 or 9 (SortBufs::hist / digit_total must be sized for it: (1 << digit_bits) * nb_max and (1 << digit_bits) * passes);
// `start`: which of the ping-pong buffers holds the input
template <int DB>
static int radix_sort_pairs_t(const LaunchCtx& c, const SortBufs& t, const uint32_t* n_ptr, uint32_t n_cap, int bit_lo, int bit_hi,
                              bool vals_are_iota, int start, int* out_idx, uint32_t flag_key, uint32_t* flag_word) {
  constexpr int ND = 1 << DB;
  int cur = start;
  const int nb = (int)((n_cap + RS_TILE - 1) / RS_TILE);
  if (nb > t.nb_max) { set_error("radix_sort_pairs: nb %d > nb_max %d", nb, t.nb_max); return TRASE_ERR_WORKSPACE; }
  const int passes = radix_passes(bit_lo, bit_hi, DB);
  if (passes > 8) return TRASE_ERR_INVALID;
  static const bool small_off = [] { const char* e = getenv("TRASE_SORT_SMALL"); return e && atoi(e) == 0; }();
  if (nb <= RS_SMALL_NB && t.hist_copies >= passes && !small_off) {
    // short sort: 1 + passes launches (see RS_SMALL_NB in common.h)
    const size_t hw = (size_t)ND * t.nb_max;
    for (int p = 0; p < passes; ++p) {
      const int shift = bit_lo + DB * p;
      const int nbits = (bit_hi - shift) < DB ? (bit_hi - shift) : DB;
      const uint32_t mask = (1u << nbits) - 1u;
      const int shift_n = shift + DB;
      const int nbits_n = (bit_hi - shift_n) < DB ? (bit_hi - shift_n) : DB;
      uint32_t* hist_p = t.hist + hw * p;
      uint32_t* hist_n = (p + 1 < passes) ? t.hist + hw * (p + 1) : nullptr;
      if (p == 0) {
        ProfScope ps("radix_hist", c.stream);
        hipLaunchKernelGGL(radix_hist_kernel<DB>, dim3(nb), dim3(RS_THREADS), 0, c.stream, t.keys[cur], n_ptr, n_cap, shift, mask, hist_p,
                           t.digit_total, t.nb_max, flag_key, flag_word, t.hist + hw, (uint32_t)(hw * (passes - 1)));
      }
      TRASE_POST_LAUNCH("radix_hist", c.stream, c.debug);
      {
        ProfScope ps("radix_scatter", c.stream);
        const uint32_t mask_n = hist_n ? ((1u << nbits_n) - 1u) : 0u;
        if (vals_are_iota && p == 0)
          hipLaunchKernelGGL((radix_scatter_kernel<true, DB, true>), dim3(nb), dim3(RS_THREADS), 0, c.stream, t.keys[cur], t.vals[cur],
                             t.keys[cur ^ 1], t.vals[cur ^ 1], n_ptr, n_cap, shift, mask, hist_p, t.digit_total, t.nb_max, hist_n, shift_n, mask_n);
        else
          hipLaunchKernelGGL((radix_scatter_kernel<false, DB, true>), dim3(nb), dim3(RS_THREADS), 0, c.stream, t.keys[cur], t.vals[cur],
                             t.keys[cur ^ 1], t.vals[cur ^ 1], n_ptr, n_cap, shift, mask, hist_p, t.digit_total, t.nb_max, hist_n, shift_n, mask_n);
      }
      TRASE_POST_LAUNCH("radix_scatter", c.stream, c.debug);
      cur ^= 1;
    }
    *out_idx = cur;
    return TRASE_OK;
  }
  for (int p = 0; p < passes; ++p) {
    const int shift = bit_lo + DB * p;
    const int nbits = (bit_hi - shift) < DB ? (bit_hi - shift) : DB;
    const uint32_t mask = (1u << nbits) - 1u;
    uint32_t* dt = t.digit_total + ND * p;
    uint32_t* vout = t.vals[cur ^ 1];
    {
      ProfScope ps("radix_hist", c.stream);
      hipLaunchKernelGGL(radix_hist_kernel<DB>, dim3(nb), dim3(RS_THREADS), 0, c.stream, t.keys[cur], n_ptr, n_cap, shift,
                         mask, t.hist, dt, t.nb_max, flag_key, p == 0 ? flag_word : (uint32_t*)nullptr);
    }
    TRASE_POST_LAUNCH("radix_hist", c.stream, c.debug);
    {
      ProfScope ps("radix_scan", c.stream);
      hipLaunchKernelGGL(radix_scan_kernel, dim3(ND), dim3(256), 0, c.stream, n_ptr, n_cap, t.hist, dt, t.nb_max);
    }
    TRASE_POST_LAUNCH("radix_scan", c.stream, c.debug);
    {
      ProfScope ps("radix_scatter", c.stream);
      if (vals_are_iota && p == 0)
        hipLaunchKernelGGL((radix_scatter_kernel<true, DB>), dim3(nb), dim3(RS_THREADS), 0, c.stream, t.keys[cur],
                           t.vals[cur], t.keys[cur ^ 1], vout, n_ptr, n_cap, shift, mask, t.hist, dt, t.nb_max);
      else
        hipLaunchKernelGGL((radix_scatter_kernel<false, DB>), dim3(nb), dim3(RS_THREADS), 0, c.stream, t.keys[cur],
                           t.vals[cur], t.keys[cur ^ 1], vout, n_ptr, n_cap, shift, mask, t.hist, dt, t.nb_max);
    }
    TRASE_POST_LAUNCH("radix_scatter", c.stream, c.debug);
    cur ^= 1;
  }
  *out_idx = cur;
  return TRASE_OK;
}

int radix_sort_pairs(const LaunchCtx& c, const SortBufs& t, const uint32_t* n_ptr, uint32_t n_cap, int bit_lo, int bit_hi,
                     bool vals_are_iota, int* out_idx, int digit_bits, int start, uint32_t flag_key, uint32_t* flag_word) {
  if (digit_bits == 9) return radix_sort_pairs_t<9>(c, t, n_ptr, n_cap, bit_lo, bit_hi, vals_are_iota, start, out_idx, flag_key, flag_word);
  if (digit_bits == 8) return radix_sort_pairs_t<8>(c, t, n_ptr, n_cap, bit_lo, bit_hi, vals_are_iota, start, out_idx, flag_key, flag_word);
  set_error("radix_sort_pairs: digit_bits %d", digit_bits);
  return TRASE_ERR_INVALID;
}

// ---- inclusive scan of tiles touched, taken in depth-rank order --------------------------------
constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 4;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;

// ---- compaction of the Gaussians that have a pair (tile-row strips) ---------------------------------------------------
// A rank that renders one strip of a view (second axis) has pairs for ~1/world of the Gaussians; the depth
// sort, the scan, emit_pairs and the row reduction then only need THOSE.  Two launches turn the per-Gaussian depth keys into
// (keys, ids) of the live Gaussians in ascending id order -- a deterministic compaction, so that the stable sort still breaks
// depth ties by ascending Gaussian index -- followed by the ids without a pair (what "writes every Gaussian" consumers walk
// behind the live ones).  The live count replaces P in the header word the sort reads its length from.
__global__ __launch_bounds__(SC_THREADS) void compact_count_kernel(const uint32_t* __restrict__ tiles, int P,
                                                                   uint32_t* __restrict__ block_cnt) {
  const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
  uint32_t c = 0;
  if (base + SC_ITEMS <= P) {
    const uint4 t = *reinterpret_cast<const uint4*>(tiles + base);
    c = (t.x != 0) + (t.y != 0) + (t.z != 0) + (t.w != 0);
  } else {
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) c += (base + k < P && tiles[base + k] != 0) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
  __shared__ uint32_t part[SC_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(SC_THREADS) void compact_scatter_kernel(const uint32_t* __restrict__ tiles,
                                                                     const uint32_t* __restrict__ keys_raw, int P,
                                                                     const uint32_t* __restrict__ block_cnt, int nblocks,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ ids_out,
                                                                     uint32_t* __restrict__ live_ids,
                                                                     uint32_t* __restrict__ hdr) {
  __shared__ uint32_t sh[SC_THREADS / 64];
  __shared__ uint32_t sh2[SC_THREADS / 64];
  // live Gaussians in the blocks before this one, and in all blocks
  uint32_t before = 0, total = 0;
  for (int b = threadIdx.x; b < nblocks; b += SC_THREADS) {
    const uint32_t v = block_cnt[b];
    total += v;
    before += (b < (int)blockIdx.x) ? v : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { before += (uint32_t)__shfl_xor((int)before, o); total += (uint32_t)__shfl_xor((int)total, o); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = before; sh2[threadIdx.x >> 6] = total; }
  __syncthreads();
  before = sh[0] + sh[1] + sh[2] + sh[3];
  total = sh2[0] + sh2[1] + sh2[2] + sh2[3];
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[HDR_WORDS - 1] = total;       // the depth sort's length: live Gaussians only
  const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
  uint32_t t[SC_ITEMS], k[SC_ITEMS];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) {
    const bool in = base + j < P;
    t[j] = in ? tiles[base + j] : 0u;
    k[j] = in ? keys_raw[base + j] : 0u;
    mine += t[j] != 0;
  }
  __shared__ uint32_t part[4];
  const uint32_t incl = block256_incl_scan(mine, part);
  uint32_t live_before = before + incl - mine;                                 // live Gaussians with a smaller id
#pragma unroll
  for (int j = 0; j < SC_ITEMS; ++j) {
    const int i = base + j;
    if (i >= P) break;
    if (t[j] != 0) {
      keys_out[live_before] = k[j];
      ids_out[live_before] = (uint32_t)i;
      live_ids[live_before] = (uint32_t)i;        // (ids_out is re-ordered by the depth sort; this copy stays ascending)
      ++live_before;
    } else {
      ids_out[total + (uint32_t)i - live_before] = (uint32_t)i;               // behind the live ones, ascending
    }
  }
}

int launch_compact_live(const LaunchCtx& c, const GeomBuf& g, int P, const PreBuf& t, const uint32_t* keys_raw,
                        uint32_t* keys_out, uint32_t* ids_out) {
  const int nblocks = (P + SC_TILE - 1) / SC_TILE;
  {
    ProfScope ps("compact_live", c.stream);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(SC_THREADS), 0, c.stream, g.tiles, P, t.block_sums);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nblocks), dim3(SC_THREADS), 0, c.stream, g.tiles, keys_raw, P, t.block_sums,
                       nblocks, keys_out, ids_out, t.live_ids, g.hdr);
  }
  TRASE_POST_LAUNCH("compact_live", c.stream, c.debug);
  return TRASE_OK;
}

__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* sh /*[SC_THREADS]*/) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 1; s < SC_THREADS; s <<= 1) {
    const uint32_t add = (threadIdx.x >= (unsigned)s) ? sh[threadIdx.x - s] : 0u;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  return sh[threadIdx.x];
}

// Also totals the lineage's pair count R = sum of the 16x16-tile rect areas (what its num_rendered reports; here a
// statistic only): recomputed from the stored centre + radius in natural order and reduced per block -- one same-address
// atomic per wave in the preprocess kernel cost more than the rest of that kernel (0.04 ms at 300k Gaussians, 0.22 ms at
// 2.5 M: device-scope atomics on one address serialise at ~12 ns each).
__global__ __launch_bounds__(SC_THREADS) void scan_partial_kernel(const uint32_t* __restrict__ tiles,
                                                                  const uint32_t* __restrict__ ids, int P,
                                                                  uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ block_sums,
                                                                  const int32_t* __restrict__ radii,
                                                                  const float2* __restrict__ xy, int gx, int gy,
                                                                  uint32_t* __restrict__ block_R,
                                                                  uint32_t* __restrict__ block_max,
                                                                  const uint32_t* __restrict__ n_live_ptr) {
  __shared__ uint32_t sh[SC_THREADS];
  const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
  const int n_live = min(P, (int)*n_live_ptr);              // depth ranks that exist (all of them unless the ids were compacted)
  uint32_t area = 0;
  // every load of the thread is requested before the first is used (a load inside a branch per item was eight dependent round
  // trips; the centre of a culled Gaussian is never written -- whatever is read there is discarded by the select)
  int rad_[SC_ITEMS];
  float2 c_[SC_ITEMS];
  uint32_t id_[SC_ITEMS];
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    const int n = blockIdx.x * SC_TILE + i * SC_THREADS + threadIdx.x;     // natural order, coalesced
    rad_[i] = (n < P) ? radii[n] : 0;
    c_[i] = (n < P) ? xy[n] : make_float2(0.f, 0.f);
    id_[i] = (base + i < n_live) ? ids[base + i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    int x0, y0, x1, y1;
    tile_rect(c_[i].x, c_[i].y, rad_[i] > 0 ? rad_[i] : 1, gx, gy, x0, y0, x1, y1);
    area += rad_[i] > 0 ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u;
  }
  const uint32_t area_incl = block_incl_scan(area, sh);
  if (threadIdx.x == SC_THREADS - 1) block_R[blockIdx.x] = area_incl;
  __syncthreads();
  uint32_t v[SC_ITEMS];
  uint32_t sum = 0, mx = 0;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    const int r = base + i;
    v[i] = (r < n_live) ? tiles[id_[i]] : 0u;
    mx = max(mx, v[i]);
    sum += v[i];
    v[i] = sum;
  }
  {   // largest pair count of one Gaussian in this block (decides whether list values can be packed, HDR_PACK)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    __shared__ uint32_t shm[SC_THREADS / 64];
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t m = shm[0];
      for (int w = 1; w < SC_THREADS / 64; ++w) m = max(m, shm[w]);
      block_max[blockIdx.x] = m;
    }
  }
  const uint32_t incl = block_incl_scan(sum, sh);
  const uint32_t excl = incl - sum;
#pragma unroll
  for (int i = 0; i < SC_ITEMS; ++i) {
    const int r = base + i;
    if (r < n_live) offsets[r] = excl + v[i];
  }
  if (threadIdx.x == SC_THREADS - 1) block_sums[blockIdx.x] = incl;
}

__global__ __launch_bounds__(SC_THREADS) void scan_sums_kernel(uint32_t* __restrict__ block_sums, int nblocks,
                                                               uint32_t* __restrict__ hdr, uint32_t cap,
                                                               const uint32_t* __restrict__ block_R,
                                                               const uint32_t* __restrict__ block_max, int pack_bits) {
  __shared__ uint32_t sh[SC_THREADS];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += SC_THREADS) {
    const int b = b0 + threadIdx.x;
    const uint32_t v = (b < nblocks) ? block_sums[b] : 0u;
    const uint32_t incl = block_incl_scan(v, sh);
    const uint32_t c0 = carry;
    if (b < nblocks) block_sums[b] = c0 + incl - v;   // exclusive prefix of each block
    __syncthreads();
    if (threadIdx.x == SC_THREADS - 1) carry = c0 + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t R = carry;     // pairs after exact sub-tile culling (HDR_R holds the lineage count)
    hdr[HDR_R_EFF] = R;
    hdr[HDR_OVERFLOW] = (hdr[HDR_OVERFLOW] & 2u) | ((R > cap) ? 1u : 0u);   // (bit 1: a saturated depth key, set by the depth sort)
  }
  uint32_t part = 0;
  for (int b = threadIdx.x; b < nblocks; b += SC_THREADS) part += block_R[b];
  __syncthreads();
  const uint32_t total = block_incl_scan(part, sh);
  if (threadIdx.x == SC_THREADS - 1) hdr[HDR_R] = total;
  __syncthreads();
  uint32_t mx = 0;
  for (int b = threadIdx.x; b < nblocks; b += SC_THREADS) mx = max(mx, block_max[b]);
  sh[threadIdx.x] = mx;
  __syncthreads();
  for (int o = SC_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) hdr[HDR_PACK] = (pack_bits > 0 && sh[0] < (1u << pack_bits)) ? (uint32_t)pack_bits : 0u;
}

// (the third step of the scan -- adding a block's exclusive prefix to its offsets and recording every Gaussian's end slot -- is
// done by emit_pairs, which reads each offset exactly once anyway: one launch less)
int launch_scan_tiles(const LaunchCtx& c, const GeomBuf& g, const uint32_t* sorted_ids, int P, const PreBuf& t, uint32_t cap,
                      const int32_t* radii, int gx, int gy, int pack_bits) {
  const int nblocks = (P + SC_TILE - 1) / SC_TILE;
  uint32_t* const block_R = t.block_sums + ((size_t)P / 1024 + 2);
  uint32_t* const block_max = t.block_sums + 2 * ((size_t)P / 1024 + 2);
  {
    ProfScope ps("scan_tiles", c.stream);
    hipLaunchKernelGGL(scan_partial_kernel, dim3(nblocks), dim3(SC_THREADS), 0, c.stream, g.tiles, sorted_ids, P,
                       t.offsets, t.block_sums, radii, g.xy, gx, gy, block_R, block_max, g.hdr + (HDR_WORDS - 1));
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SC_THREADS), 0, c.stream, t.block_sums, nblocks, g.hdr, cap, block_R,
                       block_max, pack_bits);
  }
  TRASE_POST_LAUNCH("scan_tiles", c.stream, c.debug);
  return TRASE_OK;
}

// ---- emit (tile, id) pairs in depth-rank order --------------------------------------------------
// A DPP-quad-sized group of four lanes per depth rank writes the (sub-tile key, Gaussian id) pairs of its splat: per
// sub-tile row the live columns are an interval (subtile_row_live -- the SAME function, compiled with FP contraction off,
// that the preprocess counted with).  Lane q evaluates and writes the rows q, q+4, ...; the start of a row's run is a prefix
// over the quad.  A store instruction touches 16 splats' runs instead of 64.
// A splat's pairs are contiguous (its emit-order slots).
// Splats with many rows would make their wave wait for one lane: those (more than EMIT_BIG pairs) are handed to the whole
// wave afterwards, 64 rows at a time (lane = row, wave prefix sum of the row counts).
constexpr uint32_t EMIT_BIG = 160;
// The pairs of a workgroup's 64 consecutive depth ranks are one contiguous stretch of slots.  When it is short enough (and the
// list values are the packed form) the pairs are assembled in LDS and leave as fully coalesced runs: written straight from the
// traversal, a store instruction touches 16-32 different lines with four bytes each.
#ifndef TRASE_EMIT_STAGE
#define TRASE_EMIT_STAGE 3072
#endif
constexpr uint32_t EMIT_STAGE = TRASE_EMIT_STAGE;

__global__ __launch_bounds__(256) void emit_pairs_kernel(const uint32_t* __restrict__ sorted_ids, int P,
                                                         const uint32_t* __restrict__ offsets,
                                                         const float2* __restrict__ xy, const float4* __restrict__ conic_o,
                                                         const int32_t* __restrict__ radii,
                                                         const uint32_t* __restrict__ tiles, int W, int H, int gx, int gy,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ pair_gauss,
                                                         uint32_t cap, uint32_t* __restrict__ hdr, uint32_t trash_key,
                                                         uint2* __restrict__ ranges, int sy_lo, int sy_hi,
                                                         uint32_t* __restrict__ vals, uint32_t* __restrict__ geo_words,
                                                         const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ id_end) {
  // list value of a pair (what the sub-tile sort carries): its emit-order slot, or -- HDR_PACK -- (id << jb) | index among
  // the Gaussian's own pairs, from which the compositing kernels get the id with a shift instead of a load
  const uint32_t jb = hdr[HDR_PACK];
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = gt >> 2, q = gt & 3;
  // the sub-tile ranges (trash_key + 1 entries incl. the sentinel) are cleared here: tile_ranges runs after the sort
  if (ranges)
    for (uint32_t i = (uint32_t)gt; i <= trash_key; i += gridDim.x * blockDim.x) ranges[i] = make_uint2(0u, 0u);
  if (gt == 0) {
    if (hdr[HDR_R_EFF] > cap) hdr[HDR_OVERFLOW] |= 1u;  // pairs beyond the capacity are dropped (bit 1 stays)
    hdr[HDR_WORDS - 2] = cap;
  }
  const bool in = r < min(P, (int)hdr[HDR_WORDS - 1]);      // depth ranks that exist (live Gaussians only when compacted)
  const uint32_t id = in ? sorted_ids[r] : 0;
  const uint32_t nt = in ? tiles[id] : 0;
  // this Gaussian owns exactly [end - nt, end) -- never more, never less.  offsets[] holds block-local inclusive sums
  // (scan_partial_kernel); the block's exclusive prefix is added here and in reduce_rows, the per-id end slots are recorded here.
  const uint32_t end = in ? offsets[r] + block_sums[r / SC_TILE] : 0;
  const uint32_t off0 = end - nt;
  const int gx8 = (W + SUB - 1) / SUB;
  float2 p = make_float2(0.f, 0.f);
  float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  if (nt) {
    // centre, radius and conic from the Gaussian's 64-byte record: one cache line instead of three arrays' worth
    const float4 q0 = reinterpret_cast<const float4*>(geo_words)[4 * (size_t)id];
    co = reinterpret_cast<const float4*>(geo_words)[4 * (size_t)id + 1];
    p = make_float2(q0.x, q0.y);
    tile_rect(p.x, p.y, __float_as_int(q0.w), gx, gy, x0, y0, x1, y1);
    if (q == 0) geo_words[16 * (size_t)id + 2] = off0;        // first row slot, next to the geometry the backward fetches
  }
  // (behind the loads above: a store ahead of them would have them wait for its acknowledgement.  offsets[] itself stays as it
  // is: stage 2 may be repeated on one stage 1)
  if (in && q == 0) id_end[id] = end;
  const bool big = nt > EMIT_BIG;
  // ---- staging decision: the workgroup's stretch [w_lo, w_hi) --------------------------------------------------------
  __shared__ uint32_t s_lo_hi[2];
  __shared__ uint32_t s_key[EMIT_STAGE > 0 ? EMIT_STAGE : 1], s_val[EMIT_STAGE > 0 ? EMIT_STAGE : 1];
  if (threadIdx.x == 0) { s_lo_hi[0] = off0; s_lo_hi[1] = off0; }      // (rank r0 may not exist: then the stretch is empty)
  __syncthreads();
  if (in && q == 0) atomicMax(&s_lo_hi[1], end);                       // offsets are monotone in the rank: the last one wins
  __syncthreads();
  const uint32_t w_lo = s_lo_hi[0], w_hi = min(s_lo_hi[1], cap);
  const bool staged = EMIT_STAGE > 0 && jb != 0 && w_hi > w_lo && (w_hi - w_lo) <= EMIT_STAGE;
  auto put = [&](uint32_t w, uint32_t key, uint32_t val, uint32_t gid) {
    if (staged) { s_key[w - w_lo] = key; s_val[w - w_lo] = val; }
    else { keys[w] = key; vals[w] = val; if (!jb) pair_gauss[w] = gid; }
  };
  if (nt && !big) {
    const SubtileCull cull = subtile_cull_setup(p.x, p.y, co.x, co.y, co.z, co.w);
    // lane q of the quad evaluates the rows q, q+4, ... (one subtile_row_live per four rows and lane); where a row's run
    // starts = the pairs of the rows before it: an exclusive prefix over the quad (DPP quad broadcasts) + a running base
    uint32_t pos = off0;
    const int s_lo = max(2 * y0, sy_lo), s_hi = min(min(2 * y1, sy_hi), (H + SUB - 1) / SUB);
    for (int sb = s_lo; sb < s_hi; sb += 4) {                 // trip count is uniform over the quad
      const int sy = sb + q;
      int c0 = 0, c1 = 0;
      if (sy < s_hi) subtile_row_live(cull, sy, W, H, 2 * x0, 2 * x1, c0, c1);
      const int cnt = c1 - c0;
      const int n0 = __builtin_amdgcn_update_dpp(0, cnt, 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0]
      const int n1 = __builtin_amdgcn_update_dpp(0, cnt, 0x55, 0xf, 0xf, false);   // [1,1,1,1]
      const int n2 = __builtin_amdgcn_update_dpp(0, cnt, 0xAA, 0xf, 0xf, false);   // [2,2,2,2]
      const int n3 = __builtin_amdgcn_update_dpp(0, cnt, 0xFF, 0xf, 0xf, false);   // [3,3,3,3]
      uint32_t w = pos + (uint32_t)((q > 0 ? n0 : 0) + (q > 1 ? n1 : 0) + (q > 2 ? n2 : 0));
      for (int sx = c0; sx < c1; ++sx, ++w)
        if (w < end && w < cap) put(w, (uint32_t)(sy * gx8 + sx), jb ? ((id << jb) | (w - off0)) : w, id);
      pos += (uint32_t)(n0 + n1 + n2 + n3);
    }
    // belt and braces: should the re-evaluation ever find fewer live sub-tiles than were counted, the unused slots go
    // to the sentinel sub-tile `trash_key` that no kernel renders
    for (pos += q; pos < end; pos += 4)
      if (pos < cap) put(pos, trash_key, jb ? (id << jb) : pos, id);
  }
  // ---- big splats: the whole wave, one splat after the other -------------------------------------------------------
  unsigned long long todo = __ballot(big && q == 0);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1;
    const float bpx = __shfl(p.x, l), bpy = __shfl(p.y, l);
    const float ba = __shfl(co.x, l), bb = __shfl(co.y, l), bc = __shfl(co.z, l), bo = __shfl(co.w, l);
    const int bx0 = __shfl(x0, l), by0 = __shfl(y0, l), bx1 = __shfl(x1, l), by1 = __shfl(y1, l);
    const uint32_t bid = (uint32_t)__shfl((int)id, l), bend = (uint32_t)__shfl((int)end, l);
    const uint32_t boff0 = (uint32_t)__shfl((int)off0, l);
    uint32_t run = boff0;
    const SubtileCull cull = subtile_cull_setup(bpx, bpy, ba, bb, bc, bo);
    const int r_lo = max(2 * by0, sy_lo), r_hi = min(2 * by1, sy_hi);
    for (int rb = r_lo; rb < r_hi; rb += WAVE) {
      const int sy = rb + lane;
      int c0 = 0, c1 = 0;
      if (sy < r_hi && sy * SUB < H) subtile_row_live(cull, sy, W, H, 2 * bx0, 2 * bx1, c0, c1);
      const uint32_t cnt = (uint32_t)(c1 - c0);
      uint32_t incl = cnt;                              // inclusive prefix of the rows' counts over the wave
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= o) incl += y;
      }
      uint32_t pos = run + incl - cnt;
      for (int sx = c0; sx < c1; ++sx, ++pos)
        if (pos < bend && pos < cap) put(pos, (uint32_t)(sy * gx8 + sx), jb ? ((bid << jb) | (pos - boff0)) : pos, bid);
      run += (uint32_t)__shfl((int)incl, WAVE - 1);
    }
    for (uint32_t o = run + lane; o < bend; o += WAVE)
      if (o < cap) put(o, trash_key, jb ? (bid << jb) : o, bid);
  }
  if (staged) {                                                        // workgroup-uniform
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < w_hi - w_lo; i += blockDim.x) { keys[w_lo + i] = s_key[i]; vals[w_lo + i] = s_val[i]; }
  }
}

int launch_emit_pairs(const LaunchCtx& c, const TraseRastSettings& s, const GeomBuf& g, const int32_t* radii,
                      const uint32_t* sorted_ids, int P, const PreBuf& t, uint32_t* keys, uint32_t* pair_gauss, uint32_t cap,
                      uint2* ranges_to_clear, uint32_t* vals) {
  const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE;
  int sy_lo, sy_hi;
  strip_subtile_rows(s, sy_lo, sy_hi);
  {
    ProfScope ps("emit_pairs", c.stream);
    hipLaunchKernelGGL(emit_pairs_kernel, dim3((4 * P + 255) / 256), dim3(256), 0, c.stream, sorted_ids, P, t.offsets, g.xy,
                       g.conic_o, radii, g.tiles, s.image_width, s.image_height, gx, gy, keys, pair_gauss, cap, g.hdr,
                       (uint32_t)(((s.image_width + SUB - 1) / SUB) * ((s.image_height + SUB - 1) / SUB)), ranges_to_clear,
                       sy_lo, sy_hi, vals, (uint32_t*)g.geo, t.block_sums, t.id_end);
  }
  TRASE_POST_LAUNCH("emit_pairs", c.stream, c.debug);
  return TRASE_OK;
}

// ---- backward phase 2: per-Gaussian sum of its (contiguous) per-pair gradient rows ----------------
// Four depth ranks per wave: a 16-lane group owns one Gaussian and lane t of the group owns columns 4t..4t+3 of
// its rows (ROW/4 <= 11 lanes active), so a row is read with 16-byte loads, sums never cross lanes, and the chain
// of dependent loads (rank -> id -> pair range -> flags -> rows) is paid once per four Gaussians.  Writes every
// Gaussian (zeros where it has no pairs), so neither output needs a memset and the sums are bit-reproducible.
// GW = lanes per Gaussian: 16 (four Gaussians per wave; default) or 12 (five: a 44-column row is eleven 16-byte pieces, so 55 of
// the 64 lanes of a row-load instruction carry data instead of 44).  Measured, same box, two alternations: 0.148 / 0.149 ms with
// 16 against 0.150 / 0.150 with 12 -- the kernel is bound by the cache lines its sparse rows touch (1.9 M rows among 6.0 M slots:
// ~2.4 lines of 128 bytes per 176-byte row), not by the lanes an instruction keeps busy.
#ifndef TRASE_RR_GW
#define TRASE_RR_GW 16
#endif
template <int ROW, bool BY_ID, int GW, bool LOOP>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const uint32_t* __restrict__ sorted_ids,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ tiles, int p_begin, int P,
                                                          const uint32_t* __restrict__ hdr,
                                                          const float* __restrict__ rows,
                                                          const uint8_t* __restrict__ flags, float* __restrict__ acc,
                                                          float* __restrict__ d_feats,
                                                          const float* __restrict__ raw_feats, int norm_features,
                                                          const uint32_t* __restrict__ block_sums, int live_only) {
  constexpr int F = ROW - 12, Q = ROW / 4;
  const int lane = threadIdx.x & 63;
  constexpr int GPW = WAVE / GW, GPB = GPW * (256 / WAVE);   // Gaussians per wave / per workgroup
  static_assert(ROW / 4 <= GW, "a row's 16-byte pieces must fit the lane group");
  const int grp = lane / GW, t = lane - grp * GW;              // (GW = 12: lanes 60..63 idle, grp == 5)
  // BY_ID: `offsets` is PreBuf::id_end and the groups walk the Gaussian ids [p_begin, P) -- the gradients of an id range
  // are then complete (and can be exchanged) before the rest is reduced; otherwise depth ranks [0, P) through sorted_ids
  // (a loop: with live_only the launch is a fixed number of workgroups that stride over the live ranks -- the count lives on
  // the device, and workgroups that only find out that they have nothing to do still cost their dispatch, ~4 ns each)
  const int limit = live_only ? min(P, (int)hdr[HDR_WORDS - 1]) : P;
  // LOOP = false: one workgroup per GPB Gaussians, no loop in the code (the loop costs the common whole-image launch ~2 %)
  int wb = blockIdx.x;
  if (p_begin + wb * GPB >= limit) return;
  do {
    const int r = p_begin + (wb * (256 / WAVE) + (threadIdx.x >> 6)) * GPW + grp;
    // live_only (tile-row strips, depth-rank order): only the ranks of Gaussians that have a pair -- the ids behind them (no
    // pair: no rows) are NOT written; the caller zeroes what it needs of them (preprocess_bwd_raw does)
    const bool live = r < limit && grp < GPW;
    const uint32_t id = live ? (BY_ID ? (uint32_t)r : sorted_ids[r]) : 0;
    uint32_t k0 = 0, k1 = 0;
    if (live) {
      const uint32_t nt = tiles[id];
      k1 = offsets[r] + (BY_ID ? 0u : block_sums[r / SC_TILE]);   // depth-rank offsets are block-local sums (scan_partial_kernel)
      k0 = k1 - nt;
      const uint32_t cap = hdr[HDR_WORDS - 2];        // capacity the lists were built with
      if (k1 > cap) k1 = cap;                         // pairs dropped by an overflow have no row
      if (k0 > k1) k0 = k1;
    }
    // rows exist only for pairs that were blended somewhere (flag == 1): every group fetches 16 of its flags at a
    // time, the wave ballot is cut into the four group masks, and a group walks its set bits, four rows in flight
    constexpr int U = 4;                              // rows in flight per group
    float4 s[U];
  #pragma unroll
    for (int u = 0; u < U; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool col = t < Q;
    // the flags of the NEXT sixteen pairs are requested before the rows of the current ones (one dependent level less per trip)
    uint8_t f_cur = (k0 + t < k1) ? flags[k0 + t] : (uint8_t)0;
    for (uint32_t kb = k0; __any(kb < k1); kb += GW) {
      const unsigned long long wm = __ballot(f_cur != 0);
      const uint32_t kn = kb + GW + t;
      const uint8_t f_next = (kn < k1) ? flags[kn] : (uint8_t)0;
      uint32_t m = (uint32_t)(wm >> (GW * grp)) & ((1u << GW) - 1u);
      const float4* base = reinterpret_cast<const float4*>(rows + (size_t)kb * bwd_row_stride(F)) + t;
      while (__any(m != 0)) {
        int b[U];
  #pragma unroll
        for (int u = 0; u < U; ++u) { b[u] = m ? __builtin_ctz(m) : -1; m &= m - 1; }
        float4 v[U];
  #pragma unroll
        for (int u = 0; u < U; ++u)
          v[u] = (col && b[u] >= 0) ? ld_stream(base + (size_t)b[u] * (bwd_row_stride(F) / 4)) : make_float4(0.f, 0.f, 0.f, 0.f);   // read once
  #pragma unroll
        for (int u = 0; u < U; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
      }
      f_cur = f_next;
    }
    float4 tot;
    tot.x = (s[0].x + s[1].x) + (s[2].x + s[3].x); tot.y = (s[0].y + s[1].y) + (s[2].y + s[3].y);
    tot.z = (s[0].z + s[1].z) + (s[2].z + s[3].z); tot.w = (s[0].w + s[1].w) + (s[2].w + s[3].w);
    if (F > 0 && d_feats) {
      const bool fl = t < F / 4;                      // lanes holding feature columns
      float4 dx = tot;
      if (raw_feats) {
        // fused backward of y = x / (||x|| + 1e-9) (gaussian_renderer/__init__.py:120-121): tot holds dL/dy
        const float4 x = (live && fl) ? *reinterpret_cast<const float4*>(raw_feats + (size_t)id * F + 4 * t)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        if (norm_features) {
          float n2 = fl ? (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w) : 0.f;
          float dot = fl ? (x.x * tot.x + x.y * tot.y) + (x.z * tot.z + x.w * tot.w) : 0.f;
          if constexpr (GW == 16) {
  #pragma unroll
            for (int o = 1; o < 16; o <<= 1) {        // all-reduce inside the 16-lane group
              n2 += __shfl_xor(n2, o);
              dot += __shfl_xor(dot, o);
            }
          } else {                                    // the F / 4 feature lanes of this group, in lane order
            float a2 = 0.f, ad = 0.f;
  #pragma unroll
            for (int j = 0; j < F / 4; ++j) { a2 += __shfl(n2, grp * GW + j); ad += __shfl(dot, grp * GW + j); }
            n2 = a2; dot = ad;
          }
          const float n = sqrtf(n2), den = n + 1e-9f;
          const float k = (n > 0.f) ? dot / (n * den * den) : 0.f;
          dx.x = tot.x / den - x.x * k; dx.y = tot.y / den - x.y * k;
          dx.z = tot.z / den - x.z * k; dx.w = tot.w / den - x.w * k;
        }
      }
      if (live && fl) *reinterpret_cast<float4*>(d_feats + (size_t)id * F + 4 * t) = dx;
    }
    // columns F .. F+11 -> acc[0..11] (the last two are zero padding of the row)
    if (live && t >= F / 4 && t < F / 4 + 3) *reinterpret_cast<float4*>(acc + (size_t)id * BWD_ACC + 4 * (t - F / 4)) = tot;
    wb += gridDim.x;
  } while (LOOP && p_begin + wb * GPB < limit);
}

int launch_reduce_rows(const LaunchCtx& c, const GeomBuf& g, const PreBuf& pre, int P, int F, const float* rows,
                       const uint8_t* row_flags, float* acc, float* d_feats, const float* raw_feats, int norm_features,
                       int id_begin, int id_end, int live_only) {
  // id_begin < 0: every Gaussian, in depth-rank order (its rows are then read front to back); otherwise the ids
  // [id_begin, id_end) only
  const bool by_id = id_begin >= 0;
  const int first = by_id ? id_begin : 0, last = by_id ? id_end : P;
  if (last <= first) return TRASE_OK;
  constexpr int GW = TRASE_RR_GW, GPB = (WAVE / GW) * (256 / WAVE);
  // a bounded grid that strides (LOOP) where the launch would otherwise be very large: tile-row strips (the live count is on the
  // device; most workgroups of a P-sized grid would only find out that they have nothing to do) and scenes beyond ~0.5 M Gaussians
  // (S5: 156 k workgroups, 0.46 -> 0.43 ms with 16 k striding ones; no effect at 300 k)
  const int grid_cap = live_only ? 8192 : 16384;
  const int blocks = (last - first + GPB - 1) / GPB; // 4 waves x 4 (GW = 16) or 5 (GW = 12) Gaussians per block
  {
    ProfScope ps("reduce_rows", c.stream);
#define TRASE_RR(ROW)                                                                                                         \
  do {                                                                                                                        \
    if (by_id)                                                                                                                \
      hipLaunchKernelGGL((reduce_rows_kernel<ROW, true, GW, false>), dim3(blocks), dim3(256), 0, c.stream, pre.sort.vals[0], pre.id_end, \
                         g.tiles, first, last, g.hdr, rows, row_flags, acc, d_feats, raw_feats, norm_features, pre.block_sums, 0);   \
    else if (live_only ? blocks > grid_cap : blocks > 2 * grid_cap)                                                           \
      hipLaunchKernelGGL((reduce_rows_kernel<ROW, false, GW, true>), dim3(grid_cap), dim3(256), 0, c.stream, pre.sort.vals[0],            \
                         pre.offsets, g.tiles, first, last, g.hdr, rows, row_flags, acc, d_feats, raw_feats, norm_features, pre.block_sums, live_only);  \
    else                                                                                                                      \
      hipLaunchKernelGGL((reduce_rows_kernel<ROW, false, GW, false>), dim3(blocks), dim3(256), 0, c.stream, pre.sort.vals[0],            \
                         pre.offsets, g.tiles, first, last, g.hdr, rows, row_flags, acc, d_feats, raw_feats, norm_features, pre.block_sums, live_only);  \
  } while (0)
    switch (F) {
      case 0: TRASE_RR(12); break;
      case 16: TRASE_RR(28); break;
      case 32: TRASE_RR(44); break;
      default: set_error("reduce_rows: feature width %d not compiled in", F); return TRASE_ERR_UNSUPPORTED;
    }
#undef TRASE_RR
  }
  TRASE_POST_LAUNCH("reduce_rows", c.stream, c.debug);
  return TRASE_OK;
}

// ---- tile ranges ---------------------------------------------------------------------------------
// (grid-stride form: only for key buffers that are not 16-byte aligned)
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                          uint2* __restrict__ ranges, uint32_t T,
                                                          uint32_t* __restrict__ dbg) {
  const uint32_t n = dev_n(n_ptr, cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t t = keys[i];
    if (t >= T) {                                   // cannot happen unless the sort is broken: record, stay in bounds
      if (dbg) { dbg[0] = 1; dbg[1] = i; dbg[2] = t; }
      t = T - 1;
    }
    if (i == 0 || keys[i - 1] != t) ranges[t].x = i;
    if (i == n - 1 || keys[i + 1] != t) ranges[t].y = i + 1;
  }
}

// The default path: four consecutive list entries per thread, one 16-byte load
// plus the two neighbours -- a single pass with every load in flight at once (the grid-stride form above took six
// dependent trips per thread for 6 M entries).
__global__ __launch_bounds__(256) void tile_ranges4_kernel(const uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                           uint2* __restrict__ ranges, uint32_t T,
                                                           uint32_t* __restrict__ dbg) {
  const uint32_t n = dev_n(n_ptr, cap);
  const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (i0 >= n) return;
  uint32_t k[6];                                    // keys[i0 - 1 .. i0 + 4]
  const uint4 q = *reinterpret_cast<const uint4*>(keys + i0);     // the key buffers are capacity-sized and 16-byte aligned
  k[1] = q.x; k[2] = q.y; k[3] = q.z; k[4] = q.w;
  k[0] = (i0 > 0) ? keys[i0 - 1] : 0xffffffffu;
  k[5] = (i0 + 4 < n) ? keys[i0 + 4] : 0xffffffffu;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t i = i0 + e;
    if (i >= n) break;
    uint32_t t = k[1 + e];
    if (t >= T) {                                   // cannot happen unless the sort is broken: record, stay in bounds
      if (dbg) { dbg[0] = 1; dbg[1] = i; dbg[2] = t; }
      t = T - 1;
    }
    const uint32_t prev = k[e], next = (i + 1 < n) ? k[2 + e] : 0xffffffffu;
    if (i == 0 || prev != k[1 + e]) ranges[t].x = i;
    if (i == n - 1 || next != k[1 + e]) ranges[t].y = i + 1;
  }
}

int launch_tile_ranges(const LaunchCtx& c, const uint32_t* keys, const uint32_t* n_ptr, uint32_t cap, uint2* ranges, int T,
                       uint32_t* dbg, bool clear) {
  if (clear) launch_zero_bytes(ranges, sizeof(uint2) * (size_t)T, c.stream);
  int blocks = (int)((cap + 1023) / 1024);
  if (blocks < 1) blocks = 1;
  {
    ProfScope ps("tile_ranges", c.stream);
    if ((reinterpret_cast<uintptr_t>(keys) & 15) == 0)
      hipLaunchKernelGGL(tile_ranges4_kernel, dim3(blocks), dim3(256), 0, c.stream, keys, n_ptr, cap, ranges, (uint32_t)T, dbg);
    else
      hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, c.stream, keys, n_ptr, cap,
                         ranges, (uint32_t)T, dbg);
  }
  TRASE_POST_LAUNCH("tile_ranges", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
