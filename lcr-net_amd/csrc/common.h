// common.h — shared host/device helpers for liblcr_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
#include <atomic>
#include <mutex>
#endif

#include "../../include/lcr_hip.h"

namespace lcr {

constexpr int WAVE = 64;

// GroupNorm statistics are accumulated into GN_REPLICAS copies of the [S, groups, 2] fp64 table (copy = workgroup % R) so that
// same-address fp64 atomics — which serialise in the memory-side atomic unit — are spread out; consumers sum the copies.
constexpr int GN_REPLICAS = 8;

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

// Opt-in to more than 64 KB of dynamic LDS for one kernel.  Function attributes are PER DEVICE and this library is called from several
// host threads (pipeline workers), so the "already done" state is one atomic per (kernel, device): racing callers both set the
// attribute (idempotent), nobody launches before it is set on HIS device.  One static DynLds object per kernel at the call site.
struct DynLds {
  static constexpr int MAX_DEV = 64;
  std::atomic<int> have[MAX_DEV];
  DynLds() { for (auto& h : have) h.store(0, std::memory_order_relaxed); }
  hipError_t need(const void* kernel, size_t bytes) {
    int dev = 0;
    const int want = static_cast<int>(bytes);
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV)       // unknown device: set it every time (cheap, idempotent)
      return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (have[dev].load(std::memory_order_acquire) >= want) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (e == hipSuccess) {
      int cur = have[dev].load(std::memory_order_relaxed);
      while (cur < want && !have[dev].compare_exchange_weak(cur, want, std::memory_order_release)) {}
    }
    return e;
  }
};

#ifdef __HIPCC__
// ---- exact fp32 arithmetic (never contracted into FMA) ------------------------------------------
// `__fmul_rn` / `__fadd_rn` are plain `*` / `+` to the compiler: under hipcc's default (-ffp-contract=fast)
// fadd(fmul(dx, dx), fmul(dy, dy)) comes out as v_fma_f32(dy, dy, dx * dx) — one rounding where the reference (and the oracle) round
// twice, a d² that differs by an ulp for a tenth of all offsets.  The library is built with -ffp-contract=off (csrc/build.py), which
// is what has kept the rows exact; the pragma additionally strips the `contract` flag from these operations themselves, so they stay
// exact under any build flags (tests/test_ops_gpu.py::test_distance_arithmetic_is_not_contracted places supports where it matters).
__device__ __forceinline__ float fsub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float fadd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float fmul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float fdiv(float a, float b) {
#pragma clang fp contract(off)
  return __fdiv_rn(a, b);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// ballot of a bool without the int round trip of HIP's __ballot(int) (v_cndmask 0/1 + v_cmp_ne: two VALU instructions per ballot), and the
// number of set bits of a lane mask BELOW this lane on v_mbcnt_lo / v_mbcnt_hi (two instructions instead of and + and + bcnt + bcnt)
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int mbcnt_lt(uint64_t m) {
  return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
}
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

// Per-cloud bounding boxes of stacked points with few atomics: every workgroup takes a contiguous chunk of rows and, for
// each cloud that intersects it (one, except for the B-1 chunks that straddle a boundary), reduces the rows in registers ->
// shuffles -> LDS and issues 6 atomics.  (A per-row atomic fallback for the straddling chunks used to dominate the kernel:
// same-address atomics serialise at ~0.4 us each.)  bb_* hold order-preserving encodings (f2ord).
// rs = floats per input row (3: xyz; 4: KITTI velodyne x, y, z, intensity — the fourth column is never read)
__device__ __forceinline__ void bbox_accumulate(const float* __restrict__ xyz, int64_t n, const int64_t* __restrict__ off, int B,
                                                uint32_t (*bb_min)[3], uint32_t (*bb_max)[3], int rs = 3) {
  __shared__ uint32_t s_mn[16][3], s_mx[16][3];
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t hi = lo + chunk < n ? lo + chunk : n;
  if (lo >= hi) return;
  const int b_lo = cloud_of(off, B, lo), b_hi = cloud_of(off, B, hi - 1);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int b = b_lo; b <= b_hi; ++b) {          // block-uniform
    const int64_t r0 = lo > off[b] ? lo : off[b];
    const int64_t r1 = hi < off[b + 1] ? hi : off[b + 1];
    uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int64_t i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const uint32_t u = f2ord(xyz[rs * i + d]);
        mn[d] = u < mn[d] ? u : mn[d];
        mx[d] = u > mx[d] ? u : mx[d];
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) {
        const uint32_t a = __shfl_xor(mn[d], s), c = __shfl_xor(mx[d], s);
        mn[d] = a < mn[d] ? a : mn[d];
        mx[d] = c > mx[d] ? c : mx[d];
      }
    }
    if ((threadIdx.x & 63) == 0)
      for (int d = 0; d < 3; ++d) {
        s_mn[w][d] = mn[d];
        s_mx[w][d] = mx[d];
      }
    __syncthreads();
    if (threadIdx.x < 3 && r0 < r1) {
      const int d = threadIdx.x;
      uint32_t a = s_mn[0][d], c = s_mx[0][d];
      for (int k = 1; k < nw; ++k) {
        a = s_mn[k][d] < a ? s_mn[k][d] : a;
        c = s_mx[k][d] > c ? s_mx[k][d] : c;
      }
      atomicMin(&bb_min[b][d], a);
      atomicMax(&bb_max[b][d], c);
    }
    __syncthreads();
  }
}

// ---- wavefront reductions on the DPP data path ----------------------------------------------------
// __shfl_* compiles to ds_bpermute_b32 (the LDS crossbar: ~50-100 cycles per step, each step waiting for the previous), which
// made a six-step sum / scan cost ~500 cycles per call in the wavefront-per-query kernels.  Row-level steps (16 lanes) run as
// DPP modifiers of plain VALU instructions; the four row totals are combined through v_readlane (scalar).
template <int CTRL>
__device__ __forceinline__ int dpp0(int v) {                 // lanes without a source read 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;

template <typename T>
__device__ __forceinline__ T row_sum16(T v) {                // every lane of a 16-lane row gets the row's sum
  v += dpp0<DPP_QUAD_1032>(v);
  v += dpp0<DPP_QUAD_2301>(v);
  v += dpp0<DPP_ROW_HALF_MIRROR>(v);
  v += dpp0<DPP_ROW_MIRROR>(v);
  return v;
}
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ int wave_sum(int v) {
  v = row_sum16(v);
  return rdlane(v, 0) + rdlane(v, 16) + rdlane(v, 32) + rdlane(v, 48);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum16(v);
  return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp0<DPP_ROW_SHR1>(v);                                // inclusive scan inside each 16-lane row
  v += dpp0<DPP_ROW_SHR2>(v);
  v += dpp0<DPP_ROW_SHR4>(v);
  v += dpp0<DPP_ROW_SHR8>(v);
  const int r0 = rdlane(v, 15), r1 = rdlane(v, 31), r2 = rdlane(v, 47);
  const int row = lane_id() >> 4;
  return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}
// other types (int64 lengths, doubles): the generic cross-lane form
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

// FIFO gate for the native drivers' launch sequences.  Two host threads that issue launches back to back contend for the
// runtime's (unfair) locks: the encoder driver's 135-launch burst starved the pre-processing chain — a latency-bound sequence
// of dependent launches — and the streams ended up running one after the other.  Each driver passes the gate once per
// operation: a ticket lock, so the threads strictly alternate while both want to launch and neither waits when alone.
struct LaunchTurn {
  LaunchTurn();
  ~LaunchTurn();
};

// ---- opt-in launch timing (bench.py's roofline leg) ------------------------------------------------
// While enabled (lcr_ktimer_enable), instrumented entry points bracket their launch with HIP events on the launch stream and
// log (kind, 5 integers of shape metadata); lcr_ktimer_read turns the log into durations.  Off by default: one relaxed load.
constexpr int KT_GEMM = 0, KT_AGGREGATE = 1, KT_RADIUS = 2, KT_ATTENTION = 3, KT_KPCONV_FUSED = 4, KT_SINKHORN = 5, KT_RETRIEVAL = 6;
struct KernelTimerScope {
  int slot;
  hipStream_t st;
  KernelTimerScope(int kind, hipStream_t stream, int64_t m0, int64_t m1, int64_t m2, int64_t m3 = 0, int64_t m4 = 0);
  ~KernelTimerScope();
  // A second pair of events for the kernel's OWN begin / end timestamps (hipExtLaunchKernel's startEvent / stopEvent: what a
  // profiler reports as the kernel's duration).  The bracketing pair above also contains the time the launch waited in its queue
  // behind other streams' kernels; next to each other the two say how much of a launch's latency is the kernel.  False when
  // timing is off.
  bool kernel_events(hipEvent_t* a, hipEvent_t* b);
  static KernelTimerScope*& current();      // innermost live scope of this host thread (nullptr outside instrumented entry points)
  KernelTimerScope* outer;
};
// launch `kernel` so that an active timer scope also gets the kernel's own begin / end timestamps
#define LCR_LAUNCH_TIMED(kernel, grid, block, shmem, stream, ...)                                            \
  do {                                                                                                      \
    hipEvent_t _ka, _kb;                                                                                    \
    lcr::KernelTimerScope* _sc = lcr::KernelTimerScope::current();                                          \
    if (_sc && _sc->kernel_events(&_ka, &_kb)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, _ka, _kb, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                               \
  } while (0)

// ---- device-wide exclusive scan of int32 (n known on the host as a capacity) ---------------------
// out[i] = sum(in[0..i-1]); out may alias in; total (i64) written to *total if non-null.
// ws needs scan_ws_bytes(n) bytes.
size_t scan_ws_bytes(int64_t n);
int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int64_t* total, void* ws, hipStream_t st);
// A caller that issues many scans back to back (the native pre-processing driver) can lend them a region it has ALREADY zeroed
// for their tile-state words: each scan then skips its own fill launch.  Per host thread; nullptr / 0 ends the loan.
void scan_state_pool(void* zeroed, size_t bytes);
// same, over min(n, *n_dev + n_add) entries (n_dev: device-side count; launches are sized for n and exit early beyond it)
int exclusive_scan_i32_dev(const int32_t* in, int32_t* out, int64_t n, const int32_t* n_dev, int n_add, int64_t* total, void* ws,
                           hipStream_t st);

}  // namespace lcr
