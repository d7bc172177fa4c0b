// common.hip — error plumbing + device-wide scan used by the grid/sort stages.
#include <atomic>
#include <cstdarg>
#include <mutex>
#include <utility>
#include <vector>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace lcr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<unsigned> g_turn_next{0}, g_turn_serving{0};
static const bool g_turns_on = getenv("LCR_NO_LAUNCH_TURNS") == nullptr;
LaunchTurn::LaunchTurn() {
  if (!g_turns_on) return;
  const unsigned my = g_turn_next.fetch_add(1, std::memory_order_relaxed);
  while (g_turn_serving.load(std::memory_order_acquire) != my) {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}
LaunchTurn::~LaunchTurn() {
  if (g_turns_on) g_turn_serving.fetch_add(1, std::memory_order_release);
}

// ---- opt-in launch timing ---------------------------------------------------------------------------------------------
struct KtRec {
  hipEvent_t a, b;        // bracket the launch on its stream
  hipEvent_t ka, kb;      // the kernel's own begin / end (hipExtLaunchKernel), valid when has_k
  bool       has_k;
  int        kind;
  int64_t    meta[5];
};
struct KtEvents {
  hipEvent_t a, b, ka, kb;
};
static std::mutex g_kt_mu;
static std::vector<KtRec> g_kt_log;
static std::vector<KtEvents> g_kt_pool;
static std::atomic<int> g_kt_on{0};
static std::atomic<int> g_kt_sample{1};          // every n-th instrumented launch of a kind is timed
static std::atomic<unsigned> g_kt_kinds{~0u};    // bit k: launches of kind k are timed
static std::atomic<unsigned> g_kt_seen[8];

KernelTimerScope*& KernelTimerScope::current() {
  static thread_local KernelTimerScope* cur = nullptr;
  return cur;
}
KernelTimerScope::KernelTimerScope(int kind, hipStream_t stream, int64_t m0, int64_t m1, int64_t m2, int64_t m3, int64_t m4) : slot(-1), st(stream) {
  outer = current();
  current() = this;
  if (!g_kt_on.load(std::memory_order_relaxed) || !((g_kt_kinds.load(std::memory_order_relaxed) >> (kind & 31)) & 1u)) return;
  const int every = g_kt_sample.load(std::memory_order_relaxed);
  if (every > 1 && g_kt_seen[kind & 7].fetch_add(1, std::memory_order_relaxed) % static_cast<unsigned>(every) != 0) return;
  std::lock_guard<std::mutex> lk(g_kt_mu);
  KtRec r;
  if (!g_kt_pool.empty()) {
    r.a = g_kt_pool.back().a, r.b = g_kt_pool.back().b, r.ka = g_kt_pool.back().ka, r.kb = g_kt_pool.back().kb;
    g_kt_pool.pop_back();
  } else {
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventCreate(&r.ka);
    hipEventCreate(&r.kb);
  }
  r.has_k = false;
  r.kind = kind;
  r.meta[0] = m0, r.meta[1] = m1, r.meta[2] = m2, r.meta[3] = m3, r.meta[4] = m4;
  hipEventRecord(r.a, st);
  slot = static_cast<int>(g_kt_log.size());
  g_kt_log.push_back(r);
}
KernelTimerScope::~KernelTimerScope() {
  current() = outer;
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_kt_mu);
  if (slot < static_cast<int>(g_kt_log.size())) hipEventRecord(g_kt_log[slot].b, st);
}
bool KernelTimerScope::kernel_events(hipEvent_t* a, hipEvent_t* b) {
  if (slot < 0) return false;
  std::lock_guard<std::mutex> lk(g_kt_mu);
  if (slot >= static_cast<int>(g_kt_log.size())) return false;
  g_kt_log[slot].has_k = true;
  *a = g_kt_log[slot].ka, *b = g_kt_log[slot].kb;
  return true;
}

// ---- scan: tile = 256 threads x 16 items ----------------------------------------------------------
constexpr int SCAN_T = 256;
constexpr int SCAN_I = 16;
constexpr int SCAN_TILE = SCAN_T * SCAN_I;

__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds /*>=4+1 ints*/) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
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


// ---- single-pass scan (decoupled look-back) -----------------------------------------------------------------------
// One launch instead of three: tiles take their index from an atomic ticket (so a tile only ever waits for tiles that started
// earlier), publish (flag | value) as ONE 64-bit word with agent-scope atomics — flag and value travel together, so nothing
// else has to become visible across the per-XCD L2s — and wavefront 0 looks back over 64 predecessors at a time.
// state[t] = flag << 32 | uint32 value; flag 0 = nothing yet, 1 = tile aggregate, 2 = inclusive prefix.  The state words and
// the ticket must be zero at launch (one small memset per scan).
constexpr uint64_t SCAN_AGG = 1ull << 32, SCAN_INC = 2ull << 32;

__global__ __launch_bounds__(SCAN_T) void k_scan_lookback(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                          const int32_t* __restrict__ n_dev, int n_add, int64_t* __restrict__ total,
                                                          uint64_t* __restrict__ state, uint32_t* __restrict__ ticket) {
  __shared__ int lds[8];
  __shared__ int s_tile, s_prefix;
  if (n_dev) n = min(n, static_cast<int64_t>(*n_dev) + n_add);
  // launched for the capacity: workgroups beyond the tiles in use leave before touching the ticket (one same-address atomic per
  // workgroup is what a mostly empty launch would otherwise spend its time on); the others draw exactly the tickets in use
  if (static_cast<int64_t>(blockIdx.x) * SCAN_TILE >= n) return;
  if (threadIdx.x == 0) s_tile = static_cast<int>(atomicAdd(ticket, 1u));
  __syncthreads();
  const int tile = s_tile;
  const int64_t base = static_cast<int64_t>(tile) * SCAN_TILE + threadIdx.x * SCAN_I;
  int v[SCAN_I];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_I; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int run = block_excl_scan(s, &tot, lds);
  if (threadIdx.x < 64) {                                         // wavefront 0: publish, then look back
    const int lane = threadIdx.x;
    int prefix = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&state[0], SCAN_INC | static_cast<uint32_t>(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&state[tile], SCAN_AGG | static_cast<uint32_t>(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int j0 = tile - 1;; j0 -= 64) {
        const int j = j0 - lane;                                  // lane 0 = nearest predecessor
        uint64_t w = SCAN_INC;                                    // before tile 0: an empty inclusive prefix
        if (j >= 0) {
          do {
            w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((w >> 32) == 0);
        }
        const uint64_t inc = __ballot((w >> 32) == 2);
        const int val = static_cast<int>(static_cast<uint32_t>(w));
        if (inc) {
          const int first = __builtin_ctzll(inc);                 // nearest tile that already knows its inclusive prefix
          prefix += wave_sum(lane <= first ? val : 0);
          break;
        }
        prefix += wave_sum(val);
      }
      if (lane == 0) __hip_atomic_store(&state[tile], SCAN_INC | static_cast<uint32_t>(prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      s_prefix = prefix;
      if (total && static_cast<int64_t>(tile + 1) * SCAN_TILE >= n) *total = static_cast<int64_t>(prefix) + tot;
    }
  }
  __syncthreads();
  run += s_prefix;
#pragma unroll
  for (int k = 0; k < SCAN_I; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

static thread_local char* t_pool_base = nullptr;
static thread_local size_t t_pool_bytes = 0, t_pool_off = 0;
void scan_state_pool(void* zeroed, size_t bytes) {
  t_pool_base = static_cast<char*>(zeroed);
  t_pool_bytes = bytes;
  t_pool_off = 0;
}

size_t scan_ws_bytes(int64_t n) { return align_up(sizeof(uint64_t) * (static_cast<size_t>((n + SCAN_TILE - 1) / SCAN_TILE) + 2)); }

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
  static const bool three_pass = getenv("LCR_SCAN_3PASS") != nullptr;     // the previous three-launch form, kept for A/B
  if (!three_pass) {
    uint64_t* state = static_cast<uint64_t*>(ws);                           // [nt] words + the ticket
    const size_t need = sizeof(uint64_t) * (static_cast<size_t>(nt) + 1);
    if (t_pool_base && t_pool_off + need <= t_pool_bytes) {
      state = reinterpret_cast<uint64_t*>(t_pool_base + t_pool_off);        // lent, already zero
      t_pool_off += align_up(need);
    } else {
      hipMemsetAsync(ws, 0, need, st);
    }
    uint32_t* ticket = reinterpret_cast<uint32_t*>(state + nt);
    hipLaunchKernelGGL(k_scan_lookback, dim3(nt), dim3(SCAN_T), 0, st, in, out, n, n_dev, n_add, total, state, ticket);
    return check_launch("exclusive_scan_i32");
  }
  int32_t* sums = static_cast<int32_t*>(ws);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(nt), dim3(SCAN_T), 0, st, in, n, sums, n_dev, n_add);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, st, sums, nt, total, n_dev, n_add);
  hipLaunchKernelGGL(k_scan_apply, dim3(nt), dim3(SCAN_T), 0, st, in, out, n, sums, n_dev, n_add);
  return check_launch("exclusive_scan_i32");
}

}  // namespace lcr

extern "C" const char* lcr_last_error(void) { return lcr::g_err; }

// One wavefront that spins for `microseconds` (constant 100 MHz wall clock).  Used by the host side to find out which streams
// share a hardware queue: a short kernel on another stream finishes immediately unless it sits behind this one in the same queue.
__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int lcr_stream_spin(int microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 100000) return LCR_EARG;
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<long long>(microseconds) * 100);
  return hipGetLastError() == hipSuccess ? LCR_OK : LCR_EHIP;
}

// Time only every n-th launch of each kind (n >= 1).  A timed launch costs four event records and a profiled dispatch (its
// completion signal carries timestamps and the kernel ends with a system-scope release): with every launch timed the bench's
// pipeline ran 7 % slower; sampled 1 in 9 the cost disappears in the noise, and since 9 is coprime with the 35 GEMMs / 10
// aggregations of a step every shape is sampled equally often.
extern "C" void lcr_ktimer_sample(int every) { lcr::g_kt_sample.store(every < 1 ? 1 : every); }
// Time only the kinds whose bit is set (KT_GEMM = bit 0, ...): a tool that wants the attention launches does not pay for the others.
extern "C" void lcr_ktimer_kinds(unsigned mask) { lcr::g_kt_kinds.store(mask); }
extern "C" void lcr_ktimer_enable(int on) {
  std::lock_guard<std::mutex> lk(lcr::g_kt_mu);
  if (on) {
    for (auto& r : lcr::g_kt_log) lcr::g_kt_pool.push_back({r.a, r.b, r.ka, r.kb});
    lcr::g_kt_log.clear();
    for (auto& c : lcr::g_kt_seen) c.store(0);
  }
  lcr::g_kt_on.store(on ? 1 : 0);
}

// Durations (seconds) and metadata of the logged launches of one kind; synchronises on every logged stop event.  Returns the
// number of records of that kind (which may exceed max_records: only the first max_records are written).
extern "C" int lcr_ktimer_read(int kind, int max_records, double* seconds, int64_t* meta /* [max_records,5] */) {
  std::lock_guard<std::mutex> lk(lcr::g_kt_mu);
  int n = 0;
  for (auto& r : lcr::g_kt_log) {
    if (r.kind != kind) continue;
    if (n < max_records) {
      hipEventSynchronize(r.b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, r.a, r.b);
      if (seconds) seconds[n] = static_cast<double>(ms) * 1e-3;
      if (meta)
        for (int k = 0; k < 5; ++k) meta[static_cast<size_t>(n) * 5 + k] = r.meta[k];
    }
    ++n;
  }
  return n;
}
// The same records with the kernel's own duration next to the bracketed one (seconds_kernel[i] < 0 where the launch site does
// not provide it).
extern "C" int lcr_ktimer_read2(int kind, int max_records, double* seconds, double* seconds_kernel, int64_t* meta /* [max_records,5] */) {
  std::lock_guard<std::mutex> lk(lcr::g_kt_mu);
  int n = 0;
  for (auto& r : lcr::g_kt_log) {
    if (r.kind != kind) continue;
    if (n < max_records) {
      hipEventSynchronize(r.b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, r.a, r.b);
      if (seconds) seconds[n] = static_cast<double>(ms) * 1e-3;
      if (seconds_kernel) {
        seconds_kernel[n] = -1.0;
        if (r.has_k && hipEventSynchronize(r.kb) == hipSuccess && hipEventElapsedTime(&ms, r.ka, r.kb) == hipSuccess)
          seconds_kernel[n] = static_cast<double>(ms) * 1e-3;
      }
      if (meta)
        for (int k = 0; k < 5; ++k) meta[static_cast<size_t>(n) * 5 + k] = r.meta[k];
    }
    ++n;
  }
  return n;
}
extern "C" int lcr_version(void) { return 1; }
