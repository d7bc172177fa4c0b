// precompute.hip — a-3: the whole `precompute_data_stack_mode` of one batch (experiments/lcrnet/data.py:10-74) behind ONE C call.
//
// The reference runs this collate in DataLoader worker processes (utils/utils/torch.py:48-77).  On the GPU it is a chain of
// ~250 short dependent launches (3 grid subsamples, 4 support grids, 10 radius searches) plus a read-back of the voxel counts;
// driving that chain from Python costs more host time than the kernels take and fights the encoder's launching thread for
// the interpreter lock.  Here the chain is issued natively; optionally (LCR_PRE_FORK=1) fork-join over side streams:
//
//   main   : subsample 1 ──► subsample 2 ──► subsample 3 ───────────────────────────────► join ► lengths D2H ► sync
//   side i : (points i ready) grid i ► order i ► neighbors[i] ► upsampling[i-1] ► (points i+1 ready) subsampling[i]
//
// so a stage's grid build and searches overlap the next stage's subsampling.  Results land in one caller-provided arena whose
// layout lcr_precompute_layout reports; capacities are bounded by the stage-0 point count (every stage is a subset sample).
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

#define TURN(call) ([&]() { LaunchTurn turn_; return (call); }())

namespace lcr {

constexpr size_t PRE_SCAN_POOL = 2u << 20;        // >= 60 scans of up to 4 M entries (8 B per 4096-entry tile)

struct PreCtx {
  hipStream_t side[LCR_MAX_STAGES] = {};
  hipEvent_t  ready[LCR_MAX_STAGES] = {};
  hipEvent_t  done[LCR_MAX_STAGES] = {};
  int64_t*    pinned = nullptr;      // [LCR_MAX_STAGES * 64 + 8] host staging of the device-side lengths + status
  int         device = -1;
  bool        ok = false;
};

// Side streams, events and the pinned staging buffer are pooled: a call borrows one context (concurrent calls from different
// host threads get different ones) and returns it, so short-lived driver threads do not leak streams.
static std::mutex g_ctx_mu;
static std::vector<PreCtx*> g_ctx_pool;

static PreCtx* acquire_ctx() {
  int dev = 0;
  hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (size_t i = 0; i < g_ctx_pool.size(); ++i)
      if (g_ctx_pool[i]->device == dev) {
        PreCtx* c = g_ctx_pool[i];
        g_ctx_pool.erase(g_ctx_pool.begin() + static_cast<long>(i));
        return c;
      }
  }
  PreCtx* c = new PreCtx();
  for (int i = 0; i < LCR_MAX_STAGES; ++i) {
    // default priority on purpose: high-priority side streams made the whole pipeline 15-20 % slower (queue preemption)
    hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
    hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
  }
  hipHostMalloc(reinterpret_cast<void**>(&c->pinned), sizeof(int64_t) * (LCR_MAX_STAGES * 64 + 8), hipHostMallocDefault);
  c->device = dev;
  c->ok = true;
  return c;
}

struct CtxLease {
  PreCtx* c;
  CtxLease() : c(acquire_ctx()) {}
  ~CtxLease() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctx_pool.push_back(c);
  }
};

struct LenPtrs {
  const int64_t* p[LCR_MAX_STAGES];
};
// gathers the per-stage length vectors and the status word into one staging block: one D2H copy instead of num_stages + 1
__global__ void k_pack_lengths(LenPtrs lens, int S, int B, const uint32_t* __restrict__ status, int64_t* __restrict__ packed) {
  for (int i = threadIdx.x; i < S * B; i += blockDim.x) packed[(i / B) * 64 + (i % B)] = lens.p[i / B][i % B];
  if (threadIdx.x == 0) packed[LCR_MAX_STAGES * 64] = static_cast<int64_t>(*status);
}

struct PreWs {
  int64_t*  packed;                       // [LCR_MAX_STAGES * 64 + 8] device staging of lengths + status
  void*     scan_pool;                    // zero-filled tile-state words lent to the scans of the call (common.h)
  void*     raw_ws;                       // workspace of the raw-scan voxelisation (raw mode)
  size_t    raw_bytes;
  uint32_t* status;                       // [1] shared device status word
  void*     sub_ws[LCR_MAX_STAGES];       // workspace of the subsample that PRODUCES stage i (i >= 1)
  size_t    sub_bytes[LCR_MAX_STAGES];
  void*     grid_ws[LCR_MAX_STAGES];
  size_t    grid_bytes[LCR_MAX_STAGES];
  size_t    bytes;
};

static int carve_ws(void* ws, const LcrPrecomputeLayout& L, PreWs* W) {
  Carver c(ws, ~size_t(0));
  W->status = c.take<uint32_t>(64);
  W->packed = c.take<int64_t>(LCR_MAX_STAGES * 64 + 8);
  W->scan_pool = c.take<char>(PRE_SCAN_POOL);   // directly behind the status words: one fill covers both
  W->raw_ws = nullptr;
  W->raw_bytes = 0;
  if (L.n_raw > 0) {
    int rc = lcr_grid_subsample_ws_bytes(L.n_raw, L.B, &W->raw_bytes);
    if (rc) return rc;
    W->raw_ws = c.take<char>(W->raw_bytes);
  }
  for (int i = 0; i < L.num_stages; ++i) {
    W->sub_ws[i] = nullptr;
    W->sub_bytes[i] = 0;
    if (i > 0) {
      int rc = lcr_grid_subsample_ws_bytes(L.cap[i - 1], L.B, &W->sub_bytes[i]);
      if (rc) return rc;
      W->sub_ws[i] = c.take<char>(W->sub_bytes[i]);
    }
    int rc = lcr_support_grid_ws_bytes(L.cap[i], L.B, &W->grid_bytes[i]);
    if (rc) return rc;
    W->grid_ws[i] = c.take<char>(W->grid_bytes[i]);
  }
  W->bytes = c.off;
  return LCR_OK;
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_precompute_layout(int64_t n0, int B, int num_stages, const int* limits, int upsampling, int64_t n_raw,
                                     LcrPrecomputeLayout* L) {
  if (!L || !limits || n0 < 0 || n_raw < 0 || B < 1 || B > 64 || num_stages < 1 || num_stages > LCR_MAX_STAGES) {
    set_error("lcr_precompute_layout: bad argument (B <= 64, stages <= %d)", LCR_MAX_STAGES);
    return LCR_EARG;
  }
  std::memset(L, 0, sizeof(*L));
  L->num_stages = num_stages;
  L->B = B;
  L->upsampling = upsampling == 2 ? 2 : (upsampling ? 1 : 0);     // 2: nearest-only lists (one column)
  L->n_raw = n_raw;
  const int64_t cap = n0 > 0 ? n0 : 1;
  Carver c(nullptr, ~size_t(0));
  for (int i = 0; i < num_stages; ++i) {
    if (limits[i] < 1) {
      set_error("lcr_precompute_layout: neighbor limits must be >= 1");
      return LCR_EARG;
    }
    L->limits[i] = limits[i];
    L->cap[i] = cap;
  }
  for (int i = 0; i < num_stages; ++i) {
    const bool own = i > 0 || n_raw > 0;      // stage 0 is the caller's input unless it is produced here from raw scans
    L->off_points[i] = own ? c.off : 0;
    if (own) c.take<float>(3 * (i == 0 ? std::max<int64_t>(n_raw, cap) : cap));   // the voxelisation may emit up to n_raw rows
    L->off_lengths[i] = own ? c.off : 0;
    if (own) c.take<int64_t>(B);
    L->off_order[i] = c.off;
    c.take<int32_t>(cap);
    L->off_neighbors[i] = c.off;
    c.take<int32_t>(cap * limits[i]);
    if (i + 1 < num_stages) {
      L->off_subsampling[i] = c.off;
      c.take<int32_t>(cap * limits[i]);
      if (upsampling) {
        L->off_upsampling[i] = c.off;
        c.take<int32_t>(cap * (upsampling == 2 ? 1 : limits[i + 1]));
      }
    }
  }
  L->out_bytes = c.off;
  PreWs W;
  int rc = carve_ws(nullptr, *L, &W);
  if (rc) return rc;
  L->ws_bytes = W.bytes;
  return LCR_OK;
}

extern "C" int lcr_precompute_batch(const float* points0, const int64_t* lengths0, const LcrPrecomputeLayout* L, float voxel_size,
                                    float radius, float raw_voxel, int key_bits_hint, void* out, size_t out_bytes, void* ws, size_t ws_bytes,
                                    int64_t* lengths_host, uint32_t* status_host, void* stream) {
  return lcr_precompute_batch_rows(points0, 3, lengths0, L, voxel_size, radius, raw_voxel, key_bits_hint, out, out_bytes, ws, ws_bytes, lengths_host,
                                   status_host, stream);
}

extern "C" int lcr_precompute_batch_rows(const float* points0, int raw_row_floats, const int64_t* lengths0, const LcrPrecomputeLayout* L,
                                         float voxel_size, float radius, float raw_voxel, int key_bits_hint, void* out, size_t out_bytes, void* ws,
                                         size_t ws_bytes, int64_t* lengths_host, uint32_t* status_host, void* stream) {
  if (L && raw_row_floats != 3 && !(L->n_raw > 0)) {
    set_error("lcr_precompute_batch: rows of %d floats need raw mode (stage-0 points are then produced as [n,3] inside the call)", raw_row_floats);
    return LCR_EARG;
  }
  if (!points0 || !lengths0 || !L || !out || !ws || !lengths_host || !status_host || !(voxel_size > 0.f) || !(radius > 0.f)) {
    set_error("lcr_precompute_batch: bad argument");
    return LCR_EARG;
  }
  if (out_bytes < L->out_bytes || ws_bytes < L->ws_bytes) {
    set_error("lcr_precompute_batch: arena too small (out %zu < %zu or ws %zu < %zu)", out_bytes, L->out_bytes, ws_bytes, L->ws_bytes);
    return LCR_ESPACE;
  }
  const int S = L->num_stages, B = L->B;
  // LCR_PRE_HOST_STATS=1: where this call's host time goes (issuing the launches vs waiting for the stream), printed at exit
  struct HostStats {
    double launch_s = 0, wait_s = 0;
    long calls = 0;
    bool on = getenv("LCR_PRE_HOST_STATS") != nullptr;
    ~HostStats() {
      if (on && calls) fprintf(stderr, "lcr_precompute_batch: %ld calls, issuing %.3f ms, waiting %.3f ms per call\n", calls, launch_s / calls * 1e3, wait_s / calls * 1e3);
    }
  };
  static HostStats hs;
  const auto t_in = std::chrono::steady_clock::now();
  PreWs W;
  int rc = carve_ws(ws, *L, &W);
  if (rc) return rc;
  CtxLease lease;                      // returned to the pool on every exit path (all work is joined into `stream` first)
  PreCtx& C = *lease.c;
  hipStream_t main = static_cast<hipStream_t>(stream);
  char* o = static_cast<char*>(out);
  const float* pts[LCR_MAX_STAGES];
  const int64_t* lens[LCR_MAX_STAGES];
  const bool raw = L->n_raw > 0;
  if (raw && !(raw_voxel > 0.f)) {
    set_error("lcr_precompute_batch: raw mode needs raw_voxel > 0");
    return LCR_EARG;
  }
  pts[0] = points0;
  lens[0] = lengths0;
  for (int i = raw ? 0 : 1; i < S; ++i) {
    pts[i] = reinterpret_cast<const float*>(o + L->off_points[i]);
    lens[i] = reinterpret_cast<const int64_t*>(o + L->off_lengths[i]);
  }
  auto i32 = [&](size_t off) { return reinterpret_cast<int32_t*>(o + off); };

  hipMemsetAsync(W.status, 0, static_cast<size_t>(static_cast<char*>(W.scan_pool) - reinterpret_cast<char*>(W.status)) + PRE_SCAN_POOL, main);
  struct PoolLoan {
    PoolLoan(void* p, size_t n) { scan_state_pool(p, n); }
    ~PoolLoan() { scan_state_pool(nullptr, 0); }
  } loan(W.scan_pool, PRE_SCAN_POOL);
  if (raw) {
    rc = TURN(lcr_grid_subsample_rows(points0, raw_row_floats, lengths0, B, L->n_raw, raw_voxel, key_bits_hint, const_cast<float*>(pts[0]),
                                      const_cast<int64_t*>(lens[0]), W.status, W.raw_ws, W.raw_bytes, main));
    if (rc) return rc;
  }
  float v = voxel_size, r = radius;
  float radii[LCR_MAX_STAGES];
  // Default: the searches are collected and issued as ONE launch after the last grid is built (lcr_radius_query_multi: the small
  // coarse-stage searches ride in the tail of the large ones).  LCR_PRE_FORK=1 / LCR_PRE_SPLIT_SEARCHES=1: one launch per search.
  static const bool split_searches = getenv("LCR_PRE_FORK") != nullptr || getenv("LCR_PRE_SPLIT_SEARCHES") != nullptr;
  LcrRadiusQuery searches[3 * LCR_MAX_STAGES];
  int n_search = 0;
  auto search = [&](const float* q, const int64_t* ql, int64_t nq_cap, void* gws, int64_t ns_cap, float rad, int limit, int32_t* out,
                    const int32_t* order, hipStream_t s) -> int {
    if (split_searches) return TURN(lcr_radius_query_ordered(q, ql, B, nq_cap, gws, ns_cap, rad, limit, nullptr, out, nullptr, order, s));
    searches[n_search++] = LcrRadiusQuery{q, ql, nq_cap, gws, ns_cap, rad, limit, out, order};
    return LCR_OK;
  };
  for (int i = 0; i < S; ++i) {
    if (i > 0) {
      v *= 2.f;
      rc = TURN(lcr_grid_subsample_ex(pts[i - 1], lens[i - 1], B, L->cap[i - 1], v, key_bits_hint, const_cast<float*>(pts[i]),
                                 const_cast<int64_t*>(lens[i]), W.status, W.sub_ws[i], W.sub_bytes[i], main));
      if (rc) return rc;
    }
    radii[i] = r;
    hipEventRecord(C.ready[i], main);
    // Default: the whole chain on the caller's (high-priority) stream.  LCR_PRE_FORK=1 puts every stage's grid + searches on a
    // side stream (shorter when the GPU is otherwise idle: 1.9 vs 2.4 ms per batch); next to a busy encoder the side streams run
    // at normal priority behind its kernels and the single prioritised stream wins.
    static const bool no_fork = getenv("LCR_PRE_FORK") == nullptr;
    hipStream_t s = no_fork ? main : C.side[i];
    if (!no_fork) hipStreamWaitEvent(s, C.ready[i], 0);
    rc = TURN(lcr_support_grid_build_ex(pts[i], lens[i], B, L->cap[i], r, W.status, W.grid_ws[i], W.grid_bytes[i], i32(L->off_order[i]), s));
    if (rc) return rc;
    // every search walks its queries in the QUERY set's own cell order (written by that stage's grid build, earlier in the chain)
    rc = search(pts[i], lens[i], L->cap[i], W.grid_ws[i], L->cap[i], r, L->limits[i], i32(L->off_neighbors[i]), i32(L->off_order[i]), s);
    if (rc) return rc;
    if (i > 0 && L->upsampling) {
      // upsampling == 2: the decoder reads column 0 only (nearest_upsample, backbone4.py:355-367) — a limit-1 search takes the arg-min path
      rc = search(pts[i - 1], lens[i - 1], L->cap[i - 1], W.grid_ws[i], L->cap[i], r, L->upsampling == 2 ? 1 : L->limits[i], i32(L->off_upsampling[i - 1]),
                  no_fork ? i32(L->off_order[i - 1]) : nullptr, s);   // forked: order[i-1] is written on another side stream
      if (rc) return rc;
    }
    if (i > 0) {
      hipStream_t sp = no_fork ? main : C.side[i - 1];
      if (!no_fork) hipStreamWaitEvent(sp, C.ready[i], 0);
      rc = search(pts[i], lens[i], L->cap[i], W.grid_ws[i - 1], L->cap[i - 1], radii[i - 1], L->limits[i - 1], i32(L->off_subsampling[i - 1]),
                  no_fork ? i32(L->off_order[i]) : nullptr, sp);   // forked: order[i] is written on another stream
      if (rc) return rc;
    }
    r *= 2.f;
  }
  // up to 3*S - 2 searches (S self + S-1 subsampling + S-1 upsampling = 22 at LCR_MAX_STAGES): one launch per slice of
  // LCR_RADIUS_QUERY_MULTI_MAX — the reference's 4-stage collate (10 searches) is one launch
  for (int o = 0; o < n_search; o += LCR_RADIUS_QUERY_MULTI_MAX) {
    const int c = n_search - o < LCR_RADIUS_QUERY_MULTI_MAX ? n_search - o : LCR_RADIUS_QUERY_MULTI_MAX;
    rc = TURN(lcr_radius_query_multi(searches + o, c, B, main));
    if (rc) return rc;
  }
  for (int i = 0; i < S && getenv("LCR_PRE_FORK"); ++i) {
    hipEventRecord(C.done[i], C.side[i]);
    hipStreamWaitEvent(main, C.done[i], 0);
  }
  LenPtrs lp;
  for (int i = 0; i < LCR_MAX_STAGES; ++i) lp.p[i] = i < S ? lens[i] : nullptr;
  hipLaunchKernelGGL(k_pack_lengths, dim3(1), dim3(256), 0, main, lp, S, B, W.status, W.packed);
  hipMemcpyAsync(C.pinned, W.packed, sizeof(int64_t) * (LCR_MAX_STAGES * 64 + 1), hipMemcpyDeviceToHost, main);
  const auto t_issued = std::chrono::steady_clock::now();
  hipError_t e = hipStreamSynchronize(main);
  if (hs.on) {
    const auto t_done = std::chrono::steady_clock::now();
    hs.launch_s += std::chrono::duration<double>(t_issued - t_in).count();
    hs.wait_s += std::chrono::duration<double>(t_done - t_issued).count();
    ++hs.calls;
  }
  if (e != hipSuccess) {
    set_error("lcr_precompute_batch: %s", hipGetErrorString(e));
    return LCR_EHIP;
  }
  for (int i = 0; i < S; ++i) std::memcpy(lengths_host + static_cast<size_t>(i) * B, C.pinned + i * 64, sizeof(int64_t) * B);
  *status_host = static_cast<uint32_t>(C.pinned[LCR_MAX_STAGES * 64]);
  return check_launch("lcr_precompute_batch");
}
