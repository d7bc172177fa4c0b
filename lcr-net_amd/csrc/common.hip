// common.hip — error plumbing + device-wide scan used by the grid/sort stages.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace lcr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- scan: tile = 256 threads x 16 items ----------------------------------------------------------
constexpr int SCAN_T = 256;
constexpr int SCAN_I = 16;
constexpr int SCAN_TILE = SCAN_T * SCAN_I;

__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds /*>=4+1 ints*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < SCAN_T / 64; ++k) {
    int s = lds[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// n_dev (optional): device-side element count; the scan covers min(n, *n_dev + n_add) entries and the launches sized for
// the capacity n exit at once beyond it (grid cell arrays: capacity 32 cells per point slot, a fraction of it in use).
__global__ __launch_bounds__(SCAN_T) void k_scan_tile_sums(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ sums,
                                                           const int32_t* __restrict__ n_dev, int n_add) {
  __shared__ int lds[8];
  if (n_dev) n = min(n, static_cast<int64_t>(*n_dev) + n_add);
  if (static_cast<int64_t>(blockIdx.x) * SCAN_TILE >= n) return;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * SCAN_TILE + threadIdx.x * SCAN_I;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_I; ++k)
    if (base + k < n) s += in[base + k];
  int tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of tile sums (any count) in place; total -> *total
__global__ __launch_bounds__(SCAN_T) void k_scan_sums(int32_t* __restrict__ sums, int nt, int64_t* __restrict__ total,
                                                      const int32_t* __restrict__ n_dev, int n_add) {
  __shared__ int lds[8];
  if (n_dev) nt = min(nt, static_cast<int>((static_cast<int64_t>(*n_dev) + n_add + SCAN_TILE - 1) / SCAN_TILE));
  int64_t carry = 0;
  for (int b0 = 0; b0 < nt; b0 += SCAN_T) {
    int i = b0 + threadIdx.x;
    int v = i < nt ? sums[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot, lds);
    if (i < nt) sums[i] = static_cast<int32_t>(carry + ex);
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_apply(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                       const int32_t* __restrict__ sums, const int32_t* __restrict__ n_dev, int n_add) {
  __shared__ int lds[8];
  if (n_dev) n = min(n, static_cast<int64_t>(*n_dev) + n_add);
  if (static_cast<int64_t>(blockIdx.x) * SCAN_TILE >= n) return;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * SCAN_TILE + threadIdx.x * SCAN_I;
  int v[SCAN_I];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_I; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int run = block_excl_scan(s, &tot, lds) + sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_I; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

size_t scan_ws_bytes(int64_t n) { return align_up(sizeof(int32_t) * (static_cast<size_t>((n + SCAN_TILE - 1) / SCAN_TILE) + 1)); }

int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int64_t* total, void* ws, hipStream_t st) {
  return exclusive_scan_i32_dev(in, out, n, nullptr, 0, total, ws, st);
}

int exclusive_scan_i32_dev(const int32_t* in, int32_t* out, int64_t n, const int32_t* n_dev, int n_add, int64_t* total, void* ws,
                           hipStream_t st) {
  if (n <= 0) {
    if (total) hipMemsetAsync(total, 0, sizeof(int64_t), st);
    return LCR_OK;
  }
  const int nt = static_cast<int>((n + SCAN_TILE - 1) / SCAN_TILE);
  int32_t* sums = static_cast<int32_t*>(ws);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(nt), dim3(SCAN_T), 0, st, in, n, sums, n_dev, n_add);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, st, sums, nt, total, n_dev, n_add);
  hipLaunchKernelGGL(k_scan_apply, dim3(nt), dim3(SCAN_T), 0, st, in, out, n, sums, n_dev, n_add);
  return check_launch("exclusive_scan_i32");
}

}  // namespace lcr

extern "C" const char* lcr_last_error(void) { return lcr::g_err; }
extern "C" int lcr_version(void) { return 1; }
