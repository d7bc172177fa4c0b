// radix_sort.hip — see radix_sort.h.
#include "radix_sort.h"

namespace lcr {

__device__ __forceinline__ uint32_t digit_of(uint64_t k, int pass) { return static_cast<uint32_t>(k >> (RX_BITS * pass)) & (RX_D - 1u); }

__global__ __launch_bounds__(RX_T) void k_rx_hist(const RadixCtl* __restrict__ ctl, const uint64_t* __restrict__ kA,
                                                  const uint64_t* __restrict__ kB, int pass, int32_t* __restrict__ hist, int ntiles) {
  __shared__ int s_h[RX_D];
  if (pass >= ctl->num_passes) return;
  const int64_t n = ctl->n;
  const uint64_t* keys = (pass & 1) ? kB : kA;
  for (int d = threadIdx.x; d < RX_D; d += RX_T) s_h[d] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * RX_TILE;
  for (int k = 0; k < RX_I; ++k) {
    const int64_t i = base + k * RX_T + threadIdx.x;
    if (i < n) atomicAdd(&s_h[digit_of(keys[i], pass)], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RX_D; d += RX_T) hist[static_cast<int64_t>(d) * ntiles + blockIdx.x] = s_h[d];   // digit-major
}

__global__ __launch_bounds__(RX_T) void k_rx_scatter(const RadixCtl* __restrict__ ctl, const uint64_t* __restrict__ kA,
                                                     uint64_t* __restrict__ kB_, const uint32_t* __restrict__ vA,
                                                     uint32_t* __restrict__ vB_, int pass, const int32_t* __restrict__ hist_scan,
                                                     int ntiles) {
  __shared__ int s_cnt[RX_T / 64][RX_D];   // per-wave digit totals
  if (pass >= ctl->num_passes) return;
  const int64_t n = ctl->n;
  // ping-pong: even passes read A write B
  const uint64_t* kin = (pass & 1) ? kB_ : kA;
  uint64_t* kout = (pass & 1) ? const_cast<uint64_t*>(kA) : kB_;
  const uint32_t* vin = (pass & 1) ? vB_ : vA;
  uint32_t* vout = (pass & 1) ? const_cast<uint32_t*>(vA) : vB_;

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  for (int d = lane; d < RX_D; d += 64) s_cnt[w][d] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int64_t wbase = static_cast<int64_t>(blockIdx.x) * RX_TILE + static_cast<int64_t>(w) * RX_WSLICE;
  uint64_t key[RX_I];
  uint32_t val[RX_I];
  int rank[RX_I];
  const uint64_t lt = lanemask_lt();
#pragma unroll
  for (int k = 0; k < RX_I; ++k) {
    const int64_t i = wbase + k * 64 + lane;
    const bool ok = i < n;
    key[k] = ok ? kin[i] : ~0ull;
    val[k] = ok ? vin[i] : 0u;
    const uint32_t d = ok ? digit_of(key[k], pass) : static_cast<uint32_t>(RX_D);
    // lanes with the same digit
    uint64_t m = __ballot(ok);
#pragma unroll
    for (int bit = 0; bit < RX_BITS; ++bit) {
      const uint64_t bm = __ballot((d >> bit) & 1u);
      m &= ((d >> bit) & 1u) ? bm : ~bm;
    }
    int r = 0;
    if (ok) {
      const int before = s_cnt[w][d];
      r = before + __popcll(m & lt);
      // the highest lane of each digit group publishes the new running total
      if ((m >> lane) == 1ull) s_cnt[w][d] = before + __popcll(m);
    }
    rank[k] = r;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RX_I; ++k) {
    const int64_t i = wbase + k * 64 + lane;
    if (i < n) {
      const uint32_t d = digit_of(key[k], pass);
      int off = hist_scan[static_cast<int64_t>(d) * ntiles + blockIdx.x] + rank[k];
      for (int ww = 0; ww < w; ++ww) off += s_cnt[ww][d];
      kout[off] = key[k];
      vout[off] = val[k];
    }
  }
}

int radix_sort_pairs(const RadixCtl* ctl, uint64_t* keysA, uint64_t* keysB, uint32_t* valsA, uint32_t* valsB, int64_t n_cap,
                     int max_passes, int32_t* hist, void* scan_ws, hipStream_t st) {
  if (n_cap <= 0) return LCR_OK;
  const int ntiles = static_cast<int>((n_cap + RX_TILE - 1) / RX_TILE);
  for (int p = 0; p < max_passes; ++p) {
    hipLaunchKernelGGL(k_rx_hist, dim3(ntiles), dim3(RX_T), 0, st, ctl, keysA, keysB, p, hist, ntiles);
    // scanning stale histograms on skipped passes is harmless (the scatter exits first)
    int rc = exclusive_scan_i32(hist, hist, static_cast<int64_t>(ntiles) * RX_D, nullptr, scan_ws, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rx_scatter, dim3(ntiles), dim3(RX_T), 0, st, ctl, keysA, keysB, valsA, valsB, p, hist, ntiles);
  }
  return check_launch("radix_sort_pairs");
}

}  // namespace lcr
