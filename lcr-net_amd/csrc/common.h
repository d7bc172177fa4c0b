// common.h — shared host/device helpers for liblcr_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/lcr_hip.h"

namespace lcr {

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LCR_EHIP;
  }
  return LCR_OK;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace.
struct Carver {
  char*  base;
  size_t off = 0;
  size_t cap;
  Carver(void* p, size_t c) : base(static_cast<char*>(p)), cap(c) {}
  template <typename T>
  T* take(size_t n) {
    size_t o = off;
    off = align_up(off + n * sizeof(T));
    return reinterpret_cast<T*>(base ? base + o : nullptr);
  }
  bool ok() const { return off <= cap; }
};

inline int div_up(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

#ifdef __HIPCC__
// ---- exact fp32 arithmetic (never contracted into FMA) ------------------------------------------
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

// cloud that owns stacked row i, given exclusive prefix offsets off[0..B] (off[B] = total)
__device__ __forceinline__ int cloud_of(const int64_t* off, int B, int64_t i) {
  int b = 0;
  while (b + 1 < B && i >= off[b + 1]) ++b;
  return b;
}

// order-preserving float <-> uint mapping (for atomicMin/Max on floats)
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d);
    if (lane_id() >= d) v += o;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
#endif  // __HIPCC__

// ---- device-wide exclusive scan of int32 (n known on the host as a capacity) ---------------------
// out[i] = sum(in[0..i-1]); out may alias in; total (i64) written to *total if non-null.
// ws needs scan_ws_bytes(n) bytes.
size_t scan_ws_bytes(int64_t n);
int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int64_t* total, void* ws, hipStream_t st);

}  // namespace lcr
