// radius_search.hip — a-2: stack-mode radius neighbour search for gfx950.
//
// Replaces utils.ext.radius_neighbors (utils/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91; nanoflann
// kd-tree, extra/nanoflann/nanoflann.hpp:1280-1289) and the [:, :limit] slice of modules/ops/radius_search.py:25-26.
// Contract: same-cloud supports with d2 < r*r strictly, d2 = ((dx*dx)+dy*dy)+dz*dz in fp32 without FMA,
// ascending by (d2, index) — the canonical order of cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:125-208 —
// global support index, rows padded with sum(slen).
//
// MI355X design (not a kd-tree): per call a uniform grid over every support cloud (cell >= radius, so the 27-cell
// neighbourhood covers the ball), built with atomics + a device-wide scan into a cell-sorted float4 array (x,y,z,idx)
// so that candidate loads are coalesced 16-B reads.  One 64-lane wavefront owns one query: the nine x-runs of the
// neighbourhood are concatenated, lanes stride over the candidates, survivors are compacted into LDS with a
// ballot/popcount wavefront scan, and the row is ordered by an all-pairs rank over the LDS keys
// (key = d2 bits << 32 | index; n ~ 50 so n^2/64 work per lane beats a padded bitonic network).  Rows whose
// in-radius count exceeds the LDS capacity fall back to a storage-free rank by re-enumeration (exact, slow, rare).
// HBM-bound by its output rows (limit * 4 or 8 bytes per query); everything else stays in L2.
#include <cstdlib>

#include "common.h"

namespace lcr {

constexpr int RS_CAP = 512;      // LDS keys per wavefront (8 B each)
constexpr int RS_WAVES = 4;      // wavefronts (= queries in flight) per workgroup
constexpr int RS_UNROLL = 4;     // candidate chunks (of 64) in flight per wavefront
constexpr int GRID_MAX_B = 64;   // clouds per call
constexpr int CELL_PER_PT = 32;  // cell budget = CELL_PER_PT * ns_cap + CELL_MIN * B
constexpr int CELL_MIN = 4096;

struct GridCloud {
  double  org[3];
  double  inv_cell;
  int     dim[3];
  int     cell_base;   // first cell of this cloud in the global cell arrays
  int64_t s_start;     // first support row of this cloud
};

struct GridHeader {
  int       B;
  int       n_cells;     // cells in use (<= cell_cap)
  int64_t   ns_total;    // sum(slen)
  int64_t   ns_cap;
  int64_t   cell_cap;
  GridCloud cloud[GRID_MAX_B];
  uint32_t  bb_min[GRID_MAX_B][3];   // order-preserving encodings
  uint32_t  bb_max[GRID_MAX_B][3];
  int64_t   s_off[GRID_MAX_B + 1];
};

struct GridLayout {
  GridHeader* hdr;
  int32_t*    cell_cnt;     // [cell_cap]   (zero before and after build)
  int32_t*    cell_start;   // [cell_cap+1]
  int32_t*    pt_cell;      // [ns_cap]
  float4*     sorted;       // [ns_cap]  x,y,z,bits(idx global)
  void*       scan_ws;
  size_t      bytes;
};

static GridLayout grid_layout(void* ws, int64_t ns_cap, int B) {
  GridLayout L;
  Carver c(ws, ~size_t(0));
  const int64_t cell_cap = CELL_PER_PT * ns_cap + static_cast<int64_t>(CELL_MIN) * B;
  L.hdr = c.take<GridHeader>(1);
  L.cell_cnt = c.take<int32_t>(cell_cap);
  L.cell_start = c.take<int32_t>(cell_cap + 1);
  L.pt_cell = c.take<int32_t>(ns_cap > 0 ? ns_cap : 1);
  L.sorted = c.take<float4>(ns_cap > 0 ? ns_cap : 1);
  L.scan_ws = c.take<char>(scan_ws_bytes(cell_cap + 1));
  L.bytes = c.off;
  return L;
}

// ---- build ---------------------------------------------------------------------------------------------------------
__global__ void k_grid_init(GridHeader* h, const int64_t* __restrict__ slen, int B, int64_t ns_cap, int64_t cell_cap,
                            uint32_t* status) {
  if (threadIdx.x == 0) {
    // offsets clamped to the capacity (like k_gs_init): on an overrun — the expected raw-mode path when the capacity guess was too
    // small — the clouds beyond the cap become empty, every per-cloud cell budget stays inside cell_cap, and the host retries
    // after reading the status word
    int64_t o = 0;
    bool bad = false;
    for (int b = 0; b < B; ++b) {
      h->s_off[b] = o < ns_cap ? o : ns_cap;
      const int64_t l = slen[b];
      bad |= l < 0;
      o += l > 0 ? l : 0;
    }
    bad |= o > ns_cap;
    const int64_t tot = o < ns_cap ? o : ns_cap;
    h->s_off[B] = tot;
    h->ns_total = tot;
    h->B = B;
    h->ns_cap = ns_cap;
    h->cell_cap = cell_cap;
    if (bad && status) atomicOr(status, LCR_STATUS_LEN_MISMATCH);
  }
  for (int b = threadIdx.x; b < B; b += blockDim.x)
    for (int d = 0; d < 3; ++d) {
      h->bb_min[b][d] = 0xffffffffu;
      h->bb_max[b][d] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_grid_bbox(GridHeader* h, const float* __restrict__ s) {
  const int64_t n = h->ns_total < h->ns_cap ? h->ns_total : h->ns_cap;
  bbox_accumulate(s, n, h->s_off, h->B, h->bb_min, h->bb_max);
}

__global__ void k_grid_params(GridHeader* h, float radius) {
  // one thread: B is small; cell bases are a serial prefix anyway
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int B = h->B;
  int base = 0;
  for (int b = 0; b < B; ++b) {
    GridCloud& c = h->cloud[b];
    const int64_t nb = h->s_off[b + 1] - h->s_off[b];
    c.s_start = h->s_off[b];
    c.cell_base = base;
    if (nb <= 0) {
      c.dim[0] = c.dim[1] = c.dim[2] = 0;
      c.inv_cell = 0.0;
      c.org[0] = c.org[1] = c.org[2] = 0.0;
      continue;
    }
    double lo[3], ext[3];
    for (int d = 0; d < 3; ++d) {
      lo[d] = static_cast<double>(ord2f(h->bb_min[b][d]));
      ext[d] = static_cast<double>(ord2f(h->bb_max[b][d])) - lo[d];
      c.org[d] = lo[d];
    }
    // cell >= radius * (1 + 1e-6): cell indices are computed in fp64, so |dx| < r implies |cell delta| <= 1
    double cell = fmax(static_cast<double>(radius) * 1.000001, 1e-12);
    const double budget = static_cast<double>(CELL_PER_PT) * static_cast<double>(nb) + CELL_MIN;
    int dim[3];
    for (int it = 0; it < 64; ++it) {
      double tot = 1.0;
      for (int d = 0; d < 3; ++d) {
        double n = floor(ext[d] / cell) + 1.0;
        dim[d] = n > 2.0e9 ? 2000000000 : static_cast<int>(n);
        tot *= n;
      }
      if (tot <= budget) break;
      cell *= fmax(cbrt(tot / budget), 1.0) * 1.02;
    }
    const int64_t cells = static_cast<int64_t>(dim[0]) * dim[1] * dim[2];
    if (base + cells > h->cell_cap) {          // cannot happen with clamped offsets (sum of budgets == cell_cap); never write past the arrays
      c.dim[0] = c.dim[1] = c.dim[2] = 0;
      c.inv_cell = 0.0;
      continue;
    }
    for (int d = 0; d < 3; ++d) c.dim[d] = dim[d];
    c.inv_cell = 1.0 / cell;
    base += static_cast<int>(cells);
  }
  h->n_cells = base;
}

__device__ __forceinline__ int cell_coord(double p, double org, double inv_cell, int dim) {
  // clamp in floating point first: queries may lie far outside the support box
  double c = floor((p - org) * inv_cell);
  c = fmin(fmax(c, -2.0), static_cast<double>(dim) + 1.0);
  return static_cast<int>(c);
}

__global__ __launch_bounds__(256) void k_grid_zero(const GridHeader* __restrict__ h, int32_t* __restrict__ cell_cnt) {
  const int64_t n = static_cast<int64_t>(h->n_cells) + 1;
  int4* c4 = reinterpret_cast<int4*>(cell_cnt);            // workspace slices are 256-B aligned; the tail beyond n is scratch
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i * 4 < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    c4[i] = make_int4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_grid_count(GridHeader* h, const float* __restrict__ s, int32_t* __restrict__ cell_cnt,
                                                    int32_t* __restrict__ pt_cell) {
  const int B = h->B;
  const int64_t n = min(h->ns_total, h->ns_cap);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = cloud_of(h->s_off, B, i);
    const GridCloud& c = h->cloud[b];
    int cx = cell_coord(s[3 * i + 0], c.org[0], c.inv_cell, c.dim[0]);
    int cy = cell_coord(s[3 * i + 1], c.org[1], c.inv_cell, c.dim[1]);
    int cz = cell_coord(s[3 * i + 2], c.org[2], c.inv_cell, c.dim[2]);
    cx = min(max(cx, 0), c.dim[0] - 1);   // supports are inside the box by construction; guard rounding at the max face
    cy = min(max(cy, 0), c.dim[1] - 1);
    cz = min(max(cz, 0), c.dim[2] - 1);
    const int cell = c.cell_base + (cz * c.dim[1] + cy) * c.dim[0] + cx;
    pt_cell[i] = cell;
    atomicAdd(&cell_cnt[cell], 1);
  }
}

__global__ __launch_bounds__(256) void k_grid_scatter(GridHeader* h, const float* __restrict__ s, int32_t* __restrict__ cell_cnt,
                                                      const int32_t* __restrict__ cell_start, const int32_t* __restrict__ pt_cell,
                                                      float4* __restrict__ sorted, int32_t* __restrict__ order) {
  const int64_t n = min(h->ns_total, h->ns_cap);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int cell = pt_cell[i];
    const int slot = cell_start[cell] + atomicSub(&cell_cnt[cell], 1) - 1;   // counts return to zero
    sorted[slot] = make_float4(s[3 * i + 0], s[3 * i + 1], s[3 * i + 2], __uint_as_float(static_cast<uint32_t>(i)));
    if (order) order[slot] = static_cast<int32_t>(i);       // the cell-sorted processing order, for free (lcr_support_grid_order)
  }
}

// ---- query ---------------------------------------------------------------------------------------------------------
template <bool HAS64, bool HAS32>
__global__ __launch_bounds__(RS_WAVES * 64) void k_radius_query(const float* __restrict__ q, const int64_t* __restrict__ qlen, int B,
                                                                 int64_t nq_cap, const GridHeader* __restrict__ h,
                                                                 const int32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                                 float r2, int limit, int64_t* __restrict__ out64,
                                                                 int32_t* __restrict__ out32, int32_t* __restrict__ out_cnt,
                                                                 const int32_t* __restrict__ q_order) {
  __shared__ __attribute__((aligned(16))) uint64_t s_keys[RS_WAVES][RS_CAP + 16];   // + sentinel padding of the rank loop
  __shared__ int s_run_a[RS_WAVES][12];     // first sorted slot of each x-run
  __shared__ int s_run_p[RS_WAVES][12];     // exclusive prefix of run lengths
  __shared__ int64_t s_qoff[GRID_MAX_B + 1];

  if (threadIdx.x < 64) {                                   // prefix of the query lengths: one load + a wavefront scan
    const int64_t len_b = threadIdx.x < B ? qlen[threadIdx.x] : 0;
    const int64_t inc = wave_incl_scan(len_b);
    if (threadIdx.x < B) s_qoff[threadIdx.x] = inc - len_b;
    if (threadIdx.x == B - 1) s_qoff[B] = inc;
  }
  __syncthreads();
  const int64_t nq = min(s_qoff[B], nq_cap);
  const int64_t ns_total = h->ns_total;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index as a scalar: per-query values stay in SGPRs
  uint64_t* keys = s_keys[w];

  // With a processing order (the query set's own cell order) consecutive wavefronts search neighbouring cells, and every XCD
  // walks a contiguous eighth of that order: the candidate cells are re-used from L1 / the XCD's own L2 (-6 % on the ten searches
  // of a batch; the kernel is bound by its issue rate).  Rows are written at the query's own index either way.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;            // the grid is a multiple of 8
  const int64_t per_xcd = (nq + 7) / 8;
  const int64_t t_end = min(nq, (xcd + 1) * per_xcd), t_step = static_cast<int64_t>(nslots) * RS_WAVES;
  int64_t t_cur = xcd * per_xcd + static_cast<int64_t>(slot) * RS_WAVES + w;
  int64_t qi_next = t_cur < t_end ? (q_order ? static_cast<int64_t>(q_order[t_cur]) : t_cur) : 0;
  for (; t_cur < t_end; t_cur += t_step) {
    const int64_t qi = qi_next;
    {
      const int64_t tn = t_cur + t_step < t_end ? t_cur + t_step : t_cur;                      // the next query's index: one trip ahead
      qi_next = q_order ? static_cast<int64_t>(q_order[tn]) : tn;
    }
    const int b = cloud_of(s_qoff, B, qi);
    const GridCloud& c = h->cloud[b];
    const float qx = q[3 * qi + 0], qy = q[3 * qi + 1], qz = q[3 * qi + 2];

    // nine x-runs (dy, dz in {-1,0,1}); lanes 0..8 own one run each
    int len = 0, a = 0;
    if (lane < 9 && c.dim[0] > 0) {
      const int cx = cell_coord(qx, c.org[0], c.inv_cell, c.dim[0]);
      const int cy = cell_coord(qy, c.org[1], c.inv_cell, c.dim[1]) + (lane % 3) - 1;
      const int cz = cell_coord(qz, c.org[2], c.inv_cell, c.dim[2]) + (lane / 3) - 1;
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, c.dim[0] - 1);
      if (x0 <= x1 && cy >= 0 && cy < c.dim[1] && cz >= 0 && cz < c.dim[2]) {
        const int row = c.cell_base + (cz * c.dim[1] + cy) * c.dim[0];
        a = cell_start[row + x0];
        len = cell_start[row + x1 + 1] - a;
      }
    }
    const int incl = wave_incl_scan(len);
    const int total = __shfl(incl, 8);
    if (lane < 9) {
      s_run_a[w][lane] = a;
      s_run_p[w][lane] = incl - len;
    }
    // wave-private LDS: same-wave program order is enough, but keep the compiler honest
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    int p1 = s_run_p[w][1], p2 = s_run_p[w][2], p3 = s_run_p[w][3], p4 = s_run_p[w][4], p5 = s_run_p[w][5],
        p6 = s_run_p[w][6], p7 = s_run_p[w][7], p8 = s_run_p[w][8];

    auto candidate = [&](int t, float& d2, uint32_t& idx) {
      const int r = (t >= p1) + (t >= p2) + (t >= p3) + (t >= p4) + (t >= p5) + (t >= p6) + (t >= p7) + (t >= p8);
      const int slot = s_run_a[w][r] + (t - s_run_p[w][r]);
      const float4 P = sorted[slot];
      const float dx = fsub(qx, P.x), dy = fsub(qy, P.y), dz = fsub(qz, P.z);
      d2 = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz));
      idx = __float_as_uint(P.w);
    };

    // RS_UNROLL chunks of 64 candidates per trip: their loads are issued back to back, so a query pays the L2 round trip
    // once per 256 candidates instead of once per 64
    int n = 0;
    for (int t0 = 0; t0 < total; t0 += 64 * RS_UNROLL) {
      float d2[RS_UNROLL];
      uint32_t idx[RS_UNROLL];
      bool pass[RS_UNROLL];
      // unconditional loads with the candidate index clamped (lanes beyond `total` re-read the last candidate and are masked
      // out afterwards): a load inside a lane-conditional block made the compiler wait for it at the end of the block, so the
      // RS_UNROLL loads were never in flight together
      float4 P[RS_UNROLL];
#pragma unroll
      for (int u = 0; u < RS_UNROLL; ++u) {
        const int t = min(t0 + 64 * u + lane, total - 1);
        const int r = (t >= p1) + (t >= p2) + (t >= p3) + (t >= p4) + (t >= p5) + (t >= p6) + (t >= p7) + (t >= p8);
        P[u] = sorted[s_run_a[w][r] + (t - s_run_p[w][r])];
      }
#pragma unroll
      for (int u = 0; u < RS_UNROLL; ++u) {
        const float dx = fsub(qx, P[u].x), dy = fsub(qy, P[u].y), dz = fsub(qz, P[u].z);
        d2[u] = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz));
        idx[u] = __float_as_uint(P[u].w);
        pass[u] = (t0 + 64 * u + lane < total) && d2[u] < r2;
      }
#pragma unroll
      for (int u = 0; u < RS_UNROLL; ++u) {
        const uint64_t m = __ballot(pass[u]);
        const int off = n + __popcll(m & lanemask_lt());
        if (pass[u] && off < RS_CAP) keys[off] = (static_cast<uint64_t>(__float_as_uint(d2[u])) << 32) | idx[u];
        n += __popcll(m);
      }
    }
    if (out_cnt) {
      if (lane == 0) out_cnt[qi] = n;
    }
    if (limit <= 0) continue;

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    int64_t* row64 = HAS64 ? out64 + qi * static_cast<int64_t>(limit) : nullptr;
    int32_t* row32 = HAS32 ? out32 + qi * static_cast<int64_t>(limit) : nullptr;

    if (n <= RS_CAP) {
      // all-pairs rank over the LDS keys (keys are unique: the index is part of the key).  The list is padded to a multiple of
      // 16 with all-ones sentinels (never smaller than a key) so that the comparison loop runs in groups of 16 broadcast reads:
      // with 4 per group the loop was bound by LDS latency and took half of the kernel.
      if (lane < 16 && n + lane < ((n + 15) & ~15)) keys[n + lane] = ~0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int n16 = (n + 15) & ~15;
      for (int e = lane; e < n; e += 64) {
        const uint64_t k = keys[e];
        int rank = 0;
        for (int j = 0; j < n16; j += 16) {
          ulonglong2 kj[8];                       // 8 x ds_read_b128 (two keys each), all lanes the same address
#pragma unroll
          for (int u = 0; u < 8; ++u) kj[u] = *reinterpret_cast<const ulonglong2*>(&keys[j + 2 * u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) rank += (kj[u].x < k) + (kj[u].y < k);
        }
        if (rank < limit) {
          const int64_t v = static_cast<int64_t>(static_cast<uint32_t>(k));
          if (HAS64) row64[rank] = v;
          if (HAS32) row32[rank] = static_cast<int32_t>(v);
        }
      }
    } else {
      // exact fallback without storage: rank every in-radius candidate by re-enumerating the runs
      for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        float d2 = 0.f;
        uint32_t idx = 0;
        bool pass = false;
        if (t < total) {
          candidate(t, d2, idx);
          pass = d2 < r2;
        }
        const uint64_t k = (static_cast<uint64_t>(__float_as_uint(d2)) << 32) | idx;
        int rank = 0;
        for (int u = 0; u < total; ++u) {   // u is wave-uniform: one broadcast load per step
          float e2;
          uint32_t eidx;
          candidate(u, e2, eidx);
          const uint64_t ek = (static_cast<uint64_t>(__float_as_uint(e2)) << 32) | eidx;
          rank += (e2 < r2) && (ek < k);
        }
        if (pass && rank < limit) {
          if (HAS64) row64[rank] = static_cast<int64_t>(idx);
          if (HAS32) row32[rank] = static_cast<int32_t>(idx);
        }
      }
    }
    for (int col = n + lane; col < limit; col += 64) {
      if (HAS64) row64[col] = ns_total;
      if (HAS32) row32[col] = static_cast<int32_t>(ns_total);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_support_grid_ws_bytes(int64_t ns_cap, int B, size_t* bytes) {
  if (!bytes || ns_cap < 0 || B < 1 || B > GRID_MAX_B) return LCR_EARG;
  *bytes = grid_layout(nullptr, ns_cap, B).bytes;
  return LCR_OK;
}

extern "C" int lcr_support_grid_build(const float* s, const int64_t* slen, int B, int64_t ns_cap, float radius, uint32_t* status,
                                      void* grid_ws, size_t grid_ws_bytes, void* stream) {
  return lcr_support_grid_build_ex(s, slen, B, ns_cap, radius, status, grid_ws, grid_ws_bytes, nullptr, stream);
}

extern "C" int lcr_support_grid_build_ex(const float* s, const int64_t* slen, int B, int64_t ns_cap, float radius, uint32_t* status,
                                         void* grid_ws, size_t grid_ws_bytes, int32_t* order, void* stream) {
  if (!slen || !grid_ws || B < 1 || B > GRID_MAX_B || ns_cap < 0 || !(radius > 0.f)) {
    set_error("lcr_support_grid_build: bad argument");
    return LCR_EARG;
  }
  if (ns_cap > (int64_t(1) << 31) - 1) {
    set_error("lcr_support_grid_build: more than 2^31-1 support points");
    return LCR_EARG;
  }
  GridLayout L = grid_layout(grid_ws, ns_cap, B);
  if (L.bytes > grid_ws_bytes) {
    set_error("lcr_support_grid_build: workspace too small (%zu < %zu)", grid_ws_bytes, L.bytes);
    return LCR_ESPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t cell_cap = CELL_PER_PT * ns_cap + static_cast<int64_t>(CELL_MIN) * B;
  hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(64), 0, st, L.hdr, slen, B, ns_cap, cell_cap, status);
  const int nblk = ns_cap > 0 ? min(div_up(ns_cap, 256), 2048) : 1;
  hipLaunchKernelGGL(k_grid_bbox, dim3(ns_cap > 0 ? min(div_up(ns_cap, 1024), 512) : 1), dim3(256), 0, st, L.hdr, s);   // few workgroups: 6 same-address atomics each
  hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(64), 0, st, L.hdr, radius);
  // only the n_cells (device-side) cells in use are zeroed and scanned: the capacity is 32 cells per point SLOT, and the
  // coarse stages fill a small part of it
  hipLaunchKernelGGL(k_grid_zero, dim3(min(div_up(cell_cap + 1, 1024), 2048)), dim3(256), 0, st, L.hdr, L.cell_cnt);
  hipLaunchKernelGGL(k_grid_count, dim3(nblk), dim3(256), 0, st, L.hdr, s, L.cell_cnt, L.pt_cell);
  int rc = exclusive_scan_i32_dev(L.cell_cnt, L.cell_start, cell_cap + 1, &L.hdr->n_cells, 1, nullptr, L.scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_grid_scatter, dim3(nblk), dim3(256), 0, st, L.hdr, s, L.cell_cnt, L.cell_start, L.pt_cell, L.sorted, order);
  return check_launch("lcr_support_grid_build");
}

extern "C" int lcr_radius_query(const float* q, const int64_t* qlen, int B, int64_t nq_cap, const void* grid_ws, int64_t ns_cap,
                                float radius, int limit, int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt, void* stream) {
  return lcr_radius_query_ordered(q, qlen, B, nq_cap, grid_ws, ns_cap, radius, limit, out_idx64, out_idx32, out_cnt, nullptr, stream);
}

extern "C" int lcr_radius_query_ordered(const float* q, const int64_t* qlen, int B, int64_t nq_cap, const void* grid_ws, int64_t ns_cap,
                                        float radius, int limit, int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt,
                                        const int32_t* q_order, void* stream) {
  if (!qlen || !grid_ws || B < 1 || B > GRID_MAX_B || nq_cap < 0 || ns_cap < 0 || limit < 0 || !(radius > 0.f)) {
    set_error("lcr_radius_query: bad argument");
    return LCR_EARG;
  }
  if (limit == 0 && !out_cnt) {
    set_error("lcr_radius_query: limit == 0 needs out_cnt");
    return LCR_EARG;
  }
  if (limit > 0 && !out_idx64 && !out_idx32) {
    set_error("lcr_radius_query: limit > 0 needs an index output");
    return LCR_EARG;
  }
  if (nq_cap == 0) return LCR_OK;
  static const bool no_order = getenv("LCR_RS_NO_ORDER") != nullptr;      // A/B switch
  if (no_order) q_order = nullptr;
  GridLayout L = grid_layout(const_cast<void*>(grid_ws), ns_cap, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float r2 = radius * radius;   // fp32 product, as radius_neighbors_cpu.cpp:12
  // few, long-lived workgroups: the per-workgroup prologue (query offsets) and launch ramp were ~40 % of the kernel with one
  // workgroup per 4-16 queries
  const int nblk = (min(div_up(nq_cap, RS_WAVES), getenv("LCR_RS_NBLK") ? atoi(getenv("LCR_RS_NBLK")) : 256 * 8 * 4) + 7) / 8 * 8;
  const dim3 grid(nblk), block(RS_WAVES * 64);
  KernelTimerScope timed(KT_RADIUS, st, nq_cap, ns_cap, limit, out_idx64 ? 8 : 4, B);
  if (out_idx64 && out_idx32)
    hipLaunchKernelGGL((k_radius_query<true, true>), grid, block, 0, st, q, qlen, B, nq_cap, L.hdr, L.cell_start, L.sorted, r2, limit,
                       out_idx64, out_idx32, out_cnt, q_order);
  else if (out_idx64)
    hipLaunchKernelGGL((k_radius_query<true, false>), grid, block, 0, st, q, qlen, B, nq_cap, L.hdr, L.cell_start, L.sorted, r2, limit,
                       out_idx64, out_idx32, out_cnt, q_order);
  else
    hipLaunchKernelGGL((k_radius_query<false, true>), grid, block, 0, st, q, qlen, B, nq_cap, L.hdr, L.cell_start, L.sorted, r2, limit,
                       out_idx64, out_idx32, out_cnt, q_order);
  return check_launch("lcr_radius_query");
}

// order[i] = stacked row index of the i-th support in cell-sorted order: a spatially coherent processing order for any kernel
// that gathers neighbourhoods (consecutive entries share most of their neighbours, so gathered rows are re-used from L2).
__global__ __launch_bounds__(256) void k_grid_order(const GridHeader* __restrict__ h, const float4* __restrict__ sorted, int32_t* __restrict__ order) {
  const int64_t n = h->ns_total < h->ns_cap ? h->ns_total : h->ns_cap;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    order[i] = static_cast<int32_t>(__float_as_uint(sorted[i].w));
}

extern "C" int lcr_support_grid_order(const void* grid_ws, int64_t ns_cap, int B, int32_t* order, void* stream) {
  if (!grid_ws || !order || ns_cap < 0 || B < 1 || B > GRID_MAX_B) return LCR_EARG;
  if (ns_cap == 0) return LCR_OK;
  GridLayout L = grid_layout(const_cast<void*>(grid_ws), ns_cap, B);
  hipLaunchKernelGGL(k_grid_order, dim3(min(div_up(ns_cap, 256), 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), L.hdr, L.sorted, order);
  return check_launch("lcr_support_grid_order");
}

extern "C" int lcr_radius_search_ws_bytes(int64_t nq_cap, int64_t ns_cap, int B, size_t* bytes) {
  (void)nq_cap;
  return lcr_support_grid_ws_bytes(ns_cap, B, bytes);
}

extern "C" int lcr_radius_search(const float* q, const float* s, const int64_t* qlen, const int64_t* slen, int B, int64_t nq_cap,
                                 int64_t ns_cap, float radius, int limit, int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt,
                                 uint32_t* status, void* ws, size_t ws_bytes, void* stream) {
  int rc = lcr_support_grid_build(s, slen, B, ns_cap, radius, status, ws, ws_bytes, stream);
  if (rc) return rc;
  return lcr_radius_query(q, qlen, B, nq_cap, ws, ns_cap, radius, limit, out_idx64, out_idx32, out_cnt, stream);
}
