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
// so that candidate loads are coalesced 16-B reads.  One 64-lane wavefront owns one query at a time: the nine x-runs of the
// neighbourhood (sphere-culled) are concatenated, lanes stride over the candidates, survivors are compacted into LDS with a
// ballot/popcount wavefront scan, and the row is ordered by a counting sort on 64 monotone bins of d² (key = d2 bits << 32 |
// index).  Rows whose in-radius count exceeds the LDS capacity fall back to a storage-free rank by re-enumeration (exact,
// slow, rare).  Nominally HBM-bound by its output rows (limit * 4 or 8 bytes per query); measured, it is bound by instruction
// issue on three units at once — per query ≈350 VALU, ≈180 SALU and ≈25 LDS wavefront instructions, the LDS unit being shared
// by the four SIMDs of a CU (LABNOTES.md §4.1 has the ablation numbers).
#include <cstdlib>

#include "common.h"

namespace lcr {

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

// ---- query ---------------------------------------------------------------------------------------------------------------------
// One wavefront per query (round 1's kernel: ≈900 wavefront instructions for a stage-0 query — ≈180 of per-query setup, ≈250 per trip
// of 256 candidate SLOTS whether or not the query has that many, ≈200 for an all-pairs rank — 695 us for the ten searches of a bench
// batch).  This form, same contract and bit-identical rows, 455 us:
//  * a wavefront takes RQ_BLOCK CONSECUTIVE queries of the processing order (neighbours in the query set's own cell order), so the
//    owning cloud is the previous query's (one compare instead of a prefix walk);
//  * runs whose cell row lies farther than r from the query, and the outer x cells of a run when the query is farther than r from
//    them, are dropped before anything is loaded (sphere culling in cell units, conservative by 1e-5: ≈ -25 % candidates);
//  * the nine (start, prefix) pairs live in scalar registers (v_readlane) — no LDS round trip — and the slot of candidate t is a
//    compare/select chain on them;
//  * candidate chunks of 64 are issued 1..4 at a time as the query needs (a sparse far-field query costs one chunk, not four);
//  * the row is ordered by a counting sort on 64 monotone bins of d² (LDS atomics give the position inside a bin, one wavefront
//    scan gives the bin offsets, keys move to bin order, and every key only counts the smaller keys of ITS bin — a handful —
//    instead of all n): ≈ 85 instructions instead of ≈ 200 for n ≈ 50.
constexpr int RQ_CAP = 512;       // LDS keys per wavefront (denser balls take the exact storage-free path)
constexpr int RQ_BLOCK = 4;       // consecutive queries per wavefront turn (one per 16-lane DPP row)
constexpr int RQ_BINS = 64;

__device__ __forceinline__ int rq_bin(uint64_t key, float scale) {
  const float d2 = __uint_as_float(static_cast<uint32_t>(key >> 32));
  const int b = static_cast<int>(fmul(d2, scale));          // monotone non-decreasing in d2 (round-to-nearest product, truncation)
  return b < RQ_BINS - 1 ? b : RQ_BINS - 1;
}

// wave-uniform run table of one query: prefix p_k of the run lengths and c_k = first slot of run k - p_k.  Plain named scalars on
// purpose: with arrays the compiler turned the select chain into a select of ADDRESSES and fetched c_k from scratch memory per
// candidate (a dependent memory round trip in front of every candidate load: 13 us per wavefront turn).
struct RqRuns {
  int p1, p2, p3, p4, p5, p6, p7, p8;
  int c0, c1, c2, c3, c4, c5, c6, c7, c8;
};
__device__ __forceinline__ int rq_slot(const RqRuns R, int t) {
  int off = R.c0;
  off = t >= R.p1 ? R.c1 : off;
  off = t >= R.p2 ? R.c2 : off;
  off = t >= R.p3 ? R.c3 : off;
  off = t >= R.p4 ? R.c4 : off;
  off = t >= R.p5 ? R.c5 : off;
  off = t >= R.p6 ? R.c6 : off;
  off = t >= R.p7 ? R.c7 : off;
  off = t >= R.p8 ? R.c8 : off;
  return t + off;
}

template <int U>
__device__ __forceinline__ void rq_chunks(const RqRuns R, const float4* __restrict__ sorted, int t0, int total, float qx, float qy, float qz,
                                          float r2, float bin_scale, uint64_t* keys, int* cnt, int& n) {
  const int lane = threadIdx.x & 63;
  float4 P[U];
#pragma unroll
  for (int u = 0; u < U; ++u) P[u] = sorted[rq_slot(R, min(t0 + 64 * u + lane, total - 1))];     // all loads of the trip in flight together
#pragma unroll
  for (int u = 0; u < U; ++u)      // one 16-B load each: keep .w out of the conditional store block.  AFTER the last load is issued: an asm
    asm volatile("" : "+v"(P[u].x), "+v"(P[u].y), "+v"(P[u].z), "+v"(P[u].w));   // that "uses" P[u] right behind its load made every load wait for itself (vmcnt(0) per chunk)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float dx = fsub(qx, P[u].x), dy = fsub(qy, P[u].y), dz = fsub(qz, P[u].z);
    const float d2 = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz));
    const bool pass = (t0 + 64 * u + lane < total) && d2 < r2;
    const uint64_t m = wave_ballot(pass);
    const int off = n + mbcnt_lt(m);
    if (pass && off < RQ_CAP) {
      const uint64_t key = (static_cast<uint64_t>(__float_as_uint(d2)) << 32) | __float_as_uint(P[u].w);
      keys[off] = key;
      atomicAdd(&cnt[rq_bin(key, bin_scale)], 1);             // the counting pass of the sort, for free in this LDS round trip
    }
    n += __popcll(m);
  }
}

// One search: queries, the support grid it runs against, outputs.  Passed by value (single-search kernel) or as a list (multi).
struct RqSearch {
  const float*      q;
  const int64_t*    qlen;
  int64_t           nq_cap;
  const GridHeader* h;
  const int32_t*    cell_start;
  const float4*     sorted;
  float             r2;
  int               limit;
  int64_t*          out64;
  int32_t*          out32;
  int32_t*          out_cnt;
  const int32_t*    q_order;
  int               qb;        // queries set up per wavefront turn (1, 2 or 4)
};

// exclusive prefix of a search's query lengths into LDS (one wavefront)
// (32-bit: a search has fewer than 2^31 queries, checked on the host; lengths beyond that saturate)
__device__ __forceinline__ void rq_offsets(const int64_t* __restrict__ qlen, int B, int32_t* s_qoff) {
  const int lane = threadIdx.x & 63;
  const int64_t l64 = lane < B ? qlen[lane] : 0;
  const int len_b = static_cast<int>(l64 < 0 ? 0 : (l64 > 0x3fffffff ? 0x3fffffff : l64));
  const int inc = wave_incl_scan(len_b);
  if (lane < B) s_qoff[lane] = inc - len_b;
  if (lane == B - 1) s_qoff[B] = inc;
}
__device__ __forceinline__ int rq_cloud_of(const int32_t* off, int B, int i) {
  int b = 0;
  while (b + 1 < B && i >= off[b + 1]) ++b;
  return b;
}

// One wavefront turn: `nj` (1..4) CONSECUTIVE queries of the processing order starting at position t_blk, set up at once (query j in DPP
// row j) and then ranked one after the other.  b / off_b / off_b1: the lane's cached cloud lookup, carried from turn to turn.
// Wave-private LDS as for rq_search.
template <bool HAS64, bool HAS32>
__device__ __forceinline__ void rq_turn(const RqSearch& A, int B, const int32_t* s_qoff, uint64_t* keys, uint16_t* perm, int* cnt, int* fill, int* base,
                                        int64_t t_blk, int nj, int64_t ns_total, float bin_scale, int& b, int& off_b, int& off_b1) {
  const float* __restrict__ q = A.q;
  const GridHeader* __restrict__ h = A.h;
  const int32_t* __restrict__ cell_start = A.cell_start;
  const float4* __restrict__ sorted = A.sorted;
  const int32_t* __restrict__ q_order = A.q_order;
  int64_t* __restrict__ out64 = A.out64;
  int32_t* __restrict__ out32 = A.out32;
  int32_t* __restrict__ out_cnt = A.out_cnt;
  const float r2 = A.r2;
  const int limit = A.limit;
  const int lane = threadIdx.x & 63;
  const int row = lane >> 4, sub = lane & 15;
  {
    // ---- setup of the turn's queries, one per DPP row
    const bool mine = row < nj;
    const int64_t t_me = t_blk + (mine ? row : 0);
    const int64_t qi_me = q_order ? static_cast<int64_t>(q_order[t_me]) : t_me;
    if (qi_me < off_b || qi_me >= off_b1) {      // rare in cell order (a turn seldom straddles two clouds)
      b = rq_cloud_of(s_qoff, B, static_cast<int>(qi_me));
      off_b = s_qoff[b];
      off_b1 = s_qoff[b + 1];
    }
    const GridCloud& c = h->cloud[b];
    const float mx = q[3 * qi_me + 0], my = q[3 * qi_me + 1], mz = q[3 * qi_me + 2];
    int len = 0, a = 0;
    if (mine && sub < 9 && c.dim[0] > 0) {
      const double inv = c.inv_cell;
      const double ux = (static_cast<double>(mx) - c.org[0]) * inv, uy = (static_cast<double>(my) - c.org[1]) * inv,
                   uz = (static_cast<double>(mz) - c.org[2]) * inv;
      const double fx = floor(ux), fy = floor(uy), fz = floor(uz);
      const int dyl = (sub % 3) - 1, dzl = (sub / 3) - 1;
      const int cx = static_cast<int>(fmin(fmax(fx, -2.0), static_cast<double>(c.dim[0]) + 1.0));      // == cell_coord()
      const int cy = static_cast<int>(fmin(fmax(fy, -2.0), static_cast<double>(c.dim[1]) + 1.0)) + dyl;
      const int cz = static_cast<int>(fmin(fmax(fz, -2.0), static_cast<double>(c.dim[2]) + 1.0)) + dzl;
      // sphere culling in cell units.  A support filed in cell row cy+1 lies at least (1 - frac_y) cells away in y (its cell index
      // is the same fp64 floor; supports on the box's max face are clamped INTO the last cell, i.e. lie farther still), so a run
      // whose (y, z) gap alone exceeds r cannot hold a neighbour, and an outer x cell is out of reach when its x gap adds up
      // beyond r.  Margin 1e-5 relative on r² (fp32 d² of a true neighbour is below r² by construction; the fp64 gaps carry
      // ~1e-15).  An axis on which the query lies outside the box (clamped cell) is not culled.
      const double rc2 = static_cast<double>(r2) * inv * inv * 1.00001;
      const bool iny = fy >= 0.0 && fy < static_cast<double>(c.dim[1]), inz = fz >= 0.0 && fz < static_cast<double>(c.dim[2]),
                 inx = fx >= 0.0 && fx < static_cast<double>(c.dim[0]);
      const double ry = uy - fy, rz = uz - fz, rx = ux - fx;
      const double gy = (!iny || dyl == 0) ? 0.0 : (dyl < 0 ? ry : 1.0 - ry);
      const double gz = (!inz || dzl == 0) ? 0.0 : (dzl < 0 ? rz : 1.0 - rz);
      const double g2 = gy * gy + gz * gz;
      int x0 = cx - 1, x1 = cx + 1;
      if (inx) {
        if (rx * rx + g2 >= rc2) x0 = cx;
        if ((1.0 - rx) * (1.0 - rx) + g2 >= rc2) x1 = cx;
      }
      x0 = max(x0, 0);
      x1 = min(x1, c.dim[0] - 1);
      if (g2 < rc2 && x0 <= x1 && cy >= 0 && cy < c.dim[1] && cz >= 0 && cz < c.dim[2]) {
        const int crow = c.cell_base + (cz * c.dim[1] + cy) * c.dim[0];
        a = cell_start[crow + x0];
        len = cell_start[crow + x1 + 1] - a;
      }
    }
    int incl = len;                                            // inclusive scan inside every 16-lane row (lanes 9..15 hold zeros)
    incl += dpp0<DPP_ROW_SHR1>(incl);
    incl += dpp0<DPP_ROW_SHR2>(incl);
    incl += dpp0<DPP_ROW_SHR4>(incl);
    incl += dpp0<DPP_ROW_SHR8>(incl);
    const int pk_all = incl - len, ck_all = a - pk_all;
    const int qi_lo = static_cast<int>(qi_me);                 // nq < 2^31 (checked on the host)

    for (int j = 0; j < nj; ++j) {
      const int l0 = 16 * j;                                   // wave-uniform lane base of query j
      const int64_t qi = static_cast<int64_t>(__builtin_amdgcn_readlane(qi_lo, l0));
      const float qx = rdlane(mx, l0), qy = rdlane(my, l0), qz = rdlane(mz, l0);
      const int total = rdlane(incl, l0 + 8);
      RqRuns R;
      R.c0 = rdlane(ck_all, l0);
      R.p1 = rdlane(pk_all, l0 + 1), R.c1 = rdlane(ck_all, l0 + 1);
      R.p2 = rdlane(pk_all, l0 + 2), R.c2 = rdlane(ck_all, l0 + 2);
      R.p3 = rdlane(pk_all, l0 + 3), R.c3 = rdlane(ck_all, l0 + 3);
      R.p4 = rdlane(pk_all, l0 + 4), R.c4 = rdlane(ck_all, l0 + 4);
      R.p5 = rdlane(pk_all, l0 + 5), R.c5 = rdlane(ck_all, l0 + 5);
      R.p6 = rdlane(pk_all, l0 + 6), R.c6 = rdlane(ck_all, l0 + 6);
      R.p7 = rdlane(pk_all, l0 + 7), R.c7 = rdlane(ck_all, l0 + 7);
      R.p8 = rdlane(pk_all, l0 + 8), R.c8 = rdlane(ck_all, l0 + 8);

      int n = 0;
      for (int t0 = 0; t0 < total; t0 += 64 * RS_UNROLL) {
        const int rem = total - t0;                            // wave-uniform: issue as many chunks as the query still has
        if (rem > 192) rq_chunks<4>(R, sorted, t0, total, qx, qy, qz, r2, bin_scale, keys, cnt, n);
        else if (rem > 128) rq_chunks<3>(R, sorted, t0, total, qx, qy, qz, r2, bin_scale, keys, cnt, n);
        else if (rem > 64) rq_chunks<2>(R, sorted, t0, total, qx, qy, qz, r2, bin_scale, keys, cnt, n);
        else rq_chunks<1>(R, sorted, t0, total, qx, qy, qz, r2, bin_scale, keys, cnt, n);
      }
      if (out_cnt) {
        if (lane == 0) out_cnt[qi] = n;
      }
      if (limit <= 0) {                                        // count-only mode
        if (n > 0) cnt[lane] = 0;
        continue;
      }

      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      int64_t* row64 = HAS64 ? out64 + qi * static_cast<int64_t>(limit) : nullptr;
      int32_t* row32 = HAS32 ? out32 + qi * static_cast<int64_t>(limit) : nullptr;

      if (n == 0) {
        // nothing in range: the row is all padding
      } else if (limit == 1 && n <= RQ_CAP) {
        // nearest-only rows (the decoder's upsampling lists: only column 0 is ever read): the smallest (d², idx) key of the compacted
        // candidates — no bin scan, no slot dealing, no rank loop
        cnt[lane] = 0;                                         // bin counts taken during compaction: clean for the next query
        uint64_t best = ~0ull;
        for (int e = lane; e < n; e += 64) best = min(best, keys[e]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const uint32_t lo = __shfl_xor(static_cast<uint32_t>(best), d), hi = __shfl_xor(static_cast<uint32_t>(best >> 32), d);
          best = min(best, (static_cast<uint64_t>(hi) << 32) | lo);
        }
        if (lane == 0) {
          const int64_t v = static_cast<int64_t>(static_cast<uint32_t>(best));
          if (HAS64) row64[0] = v;
          if (HAS32) row32[0] = static_cast<int32_t>(v);
        }
      } else if (n <= RQ_CAP) {
        // counting sort on RQ_BINS monotone bins of d²: bin counts (taken while the keys were compacted) -> wavefront scan -> key
        // SLOTS dealt into bin order (an atomic per key hands out the positions of a bin; order inside a bin is arbitrary; 16-bit
        // slots, the keys stay put: a second key buffer costs two of the six resident workgroups per CU, measured 130 -> 154 us) -> every key counts the smaller keys of its own bin (1.8 keys on average for n = 50).
        {
          const int cb = cnt[lane];
          const int inc = wave_incl_scan(cb);
          base[lane + 1] = inc;
          if (lane == 0) base[0] = 0;
          fill[lane] = inc - cb;
          cnt[lane] = 0;                                       // clean for the next query
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < n; e += 64) perm[atomicAdd(&fill[rq_bin(keys[e], bin_scale)], 1)] = static_cast<uint16_t>(e);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < n; e += 64) {
          const uint64_t k = keys[e];
          const int bb = rq_bin(k, bin_scale);
          const int lo = base[bb], hi = base[bb + 1];
          int rank = lo;
          for (int j0 = lo; j0 < hi; j0 += 2) {                // two bin mates per trip (a bin holds 1.8 keys on average): the LDS unit,
            const int s0 = perm[j0], s1 = perm[min(j0 + 1, hi - 1)];   // shared by the CU's four SIMDs, is this kernel's scarce resource
            const uint64_t a0 = keys[s0], a1 = keys[s1];
            rank += (a0 < k) + (j0 + 1 < hi && a1 < k);        // keys are unique (the index is part of the key)
          }
          if (rank < limit) {
            const int64_t v = static_cast<int64_t>(static_cast<uint32_t>(k));
            if (HAS64) row64[rank] = v;
            if (HAS32) row32[rank] = static_cast<int32_t>(v);
          }
        }
      } else {
        // exact fallback without storage: rank every in-radius candidate by re-enumerating the runs
        cnt[lane] = 0;                                         // bin counts of the compacted part: not used on this path
        for (int t0 = 0; t0 < total; t0 += 64) {
          const int t = t0 + lane;
          float d2 = 0.f;
          uint32_t idx = 0;
          bool pass = false;
          if (t < total) {
            const float4 P = sorted[rq_slot(R, t)];
            const float dx = fsub(qx, P.x), dy = fsub(qy, P.y), dz = fsub(qz, P.z);
            d2 = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz));
            idx = __float_as_uint(P.w);
            pass = d2 < r2;
          }
          const uint64_t k = (static_cast<uint64_t>(__float_as_uint(d2)) << 32) | idx;
          int rank = 0;
          for (int u = 0; u < total; ++u) {   // u is wave-uniform: one broadcast load per step
            const float4 E = sorted[rq_slot(R, u)];
            const float ex = fsub(qx, E.x), ey = fsub(qy, E.y), ez = fsub(qz, E.z);
            const float e2 = fadd(fadd(fmul(ex, ex), fmul(ey, ey)), fmul(ez, ez));
            const uint64_t ek = (static_cast<uint64_t>(__float_as_uint(e2)) << 32) | __float_as_uint(E.w);
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
}

// this wavefront's share of one search (see the file header); wave-private LDS: keys, perm, cnt (all zero on entry and exit), fill, base
template <bool HAS64, bool HAS32>
__device__ __forceinline__ void rq_search(const RqSearch& A, int B, const int32_t* s_qoff, uint64_t* keys, uint16_t* perm, int* cnt, int* fill,
                                          int* base) {
  const float* __restrict__ q = A.q;
  const GridHeader* __restrict__ h = A.h;
  const int32_t* __restrict__ cell_start = A.cell_start;
  const float4* __restrict__ sorted = A.sorted;
  const int32_t* __restrict__ q_order = A.q_order;
  int64_t* __restrict__ out64 = A.out64;
  int32_t* __restrict__ out32 = A.out32;
  int32_t* __restrict__ out_cnt = A.out_cnt;
  const float r2 = A.r2;
  const int limit = A.limit, qb = A.qb;
  const int64_t nq_cap = A.nq_cap;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nq = min(static_cast<int64_t>(s_qoff[B]), nq_cap);
  const int64_t ns_total = h->ns_total;
  const float bin_scale = fdiv(static_cast<float>(RQ_BINS), r2);

  // every XCD walks a contiguous eighth of the processing order; inside it a wavefront takes `qb` (1, 2 or 4) CONSECUTIVE queries per
  // turn and sets all of them up AT ONCE: query j of the turn lives in DPP row j (lanes 16j .. 16j+8 own its nine runs), so the
  // header / cell-table loads, the fp64 cell arithmetic and the culling of up to four queries cost the instructions and the load
  // round trips of one.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int64_t per_xcd = (nq + 7) / 8;
  const int64_t x_lo = xcd * per_xcd, x_hi = min(nq, (xcd + 1) * per_xcd);
  const int64_t turn_step = static_cast<int64_t>(nslots) * RS_WAVES * qb;
  int b = 0;                                     // cloud of this lane's previous query
  int off_b = 0, off_b1 = 0;                     // its [first, last) rows (nq < 2^31): empty until the first lookup
  for (int64_t t_blk = x_lo + (static_cast<int64_t>(slot) * RS_WAVES + w) * qb; t_blk < x_hi; t_blk += turn_step) {
    const int nj = static_cast<int>(min(static_cast<int64_t>(qb), x_hi - t_blk));             // wave-uniform
    rq_turn<HAS64, HAS32>(A, B, s_qoff, keys, perm, cnt, fill, base, t_blk, nj, ns_total, bin_scale, b, off_b, off_b1);
  }
}

template <bool HAS64, bool HAS32>
__global__ __launch_bounds__(RS_WAVES * 64) void k_radius_query(RqSearch A, int B) {
  __shared__ __attribute__((aligned(16))) uint64_t s_keys[RS_WAVES][RQ_CAP];
  __shared__ uint16_t s_perm[RS_WAVES][RQ_CAP];             // bin-ordered position -> slot in s_keys
  __shared__ int s_cnt[RS_WAVES][RQ_BINS];
  __shared__ int s_fill[RS_WAVES][RQ_BINS];
  __shared__ int s_base[RS_WAVES][RQ_BINS + 1];
  __shared__ int32_t s_qoff[GRID_MAX_B + 1];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 64) rq_offsets(A.qlen, B, s_qoff);
  s_cnt[w][lane] = 0;
  __syncthreads();
  rq_search<HAS64, HAS32>(A, B, s_qoff, s_keys[w], s_perm[w], s_cnt[w], s_fill[w], s_base[w]);
}

// All searches of a collate in ONE launch (int32 rows): every wavefront works through its share of search 0, then of search 1, ...
// without a barrier in between, so the small coarse-stage searches — alone they sit on the ~10 us launch floor with a handful of
// workgroups — ride in the tail of the large ones.
constexpr int RQ_MAX_SEARCHES = LCR_RADIUS_QUERY_MULTI_MAX;
struct RqMulti {
  int      n, B;
  RqSearch s[RQ_MAX_SEARCHES];
};
__global__ __launch_bounds__(RS_WAVES * 64) void k_radius_query_multi(RqMulti m) {
  __shared__ __attribute__((aligned(16))) uint64_t s_keys[RS_WAVES][RQ_CAP];
  __shared__ uint16_t s_perm[RS_WAVES][RQ_CAP];
  __shared__ int s_cnt[RS_WAVES][RQ_BINS];
  __shared__ int s_fill[RS_WAVES][RQ_BINS];
  __shared__ int s_base[RS_WAVES][RQ_BINS + 1];
  __shared__ int32_t s_qoff[RQ_MAX_SEARCHES][GRID_MAX_B + 1];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = w; i < m.n; i += RS_WAVES) rq_offsets(m.s[i].qlen, m.B, s_qoff[i]);
  s_cnt[w][lane] = 0;
  __syncthreads();
  for (int i = 0; i < m.n; ++i) rq_search<false, true>(m.s[i], m.B, s_qoff[i], s_keys[w], s_perm[w], s_cnt[w], s_fill[w], s_base[w]);
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

// residency of the query kernels (workgroups per CU the grids are sized for) and the CU count, once per process
static int rq_wg_per_cu() {
  // One resident generation of workgroups, each with an equal share of the queries: with more workgroups than fit, the last
  // generation runs on a part-empty chip (2048 workgroups on 6-per-CU residency: 2 of 8 ran alone, measured 3.3 wavefronts per
  // SIMD on average instead of 6).  Residency from the kernel's register count (the occupancy API is one workgroup per CU high
  // for kernels with 97-112 SGPRs on this stack, MI355X_MICROARCH.md).
  static const int v = []() {
    if (getenv("LCR_RS_WG_PER_CU")) return atoi(getenv("LCR_RS_WG_PER_CU"));
    hipFuncAttributes fa;
    int api = 0;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_radius_query<false, true>)) != hipSuccess) return 4;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, reinterpret_cast<const void*>(&k_radius_query<false, true>), RS_WAVES * 64, 0);
    const int by_vgpr = 512 / ((fa.numRegs + 7) / 8 * 8);
    return max(1, min(min(api > 0 ? api : 8, by_vgpr), 6));
  }();
  return v;
}
static int rq_n_cu() {
  static const int v = []() {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  return v;
}
// queries per wavefront turn: four (one per DPP row) when the search is large enough to fill the chip anyway, fewer for the small
// coarse-stage searches, which are latency-bound and want as many wavefronts as they have queries
static int rq_qb(int64_t nq_cap) {
  static const int qb_env = getenv("LCR_RS_QB") ? atoi(getenv("LCR_RS_QB")) : 0;
  return qb_env ? qb_env : (nq_cap >= 65536 ? 4 : (nq_cap >= 16384 ? 2 : 1));
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
  if (nq_cap > (int64_t(1) << 31) - 1) {
    set_error("lcr_radius_query: more than 2^31-1 query points");
    return LCR_EARG;
  }
  if (nq_cap == 0) return LCR_OK;
  static const bool no_order = getenv("LCR_RS_NO_ORDER") != nullptr;      // A/B switch
  if (no_order) q_order = nullptr;
  GridLayout L = grid_layout(const_cast<void*>(grid_ws), ns_cap, B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  RqSearch A;
  A.q = q, A.qlen = qlen, A.nq_cap = nq_cap, A.h = L.hdr, A.cell_start = L.cell_start, A.sorted = L.sorted;
  A.r2 = radius * radius;   // fp32 product, as radius_neighbors_cpu.cpp:12
  A.limit = limit, A.out64 = out_idx64, A.out32 = out_idx32, A.out_cnt = out_cnt, A.q_order = q_order, A.qb = rq_qb(nq_cap);
  const dim3 block(RS_WAVES * 64);
  KernelTimerScope timed(KT_RADIUS, st, nq_cap, ns_cap, limit, out_idx64 ? 8 : 4, B);
  const int nblk = (min(div_up(nq_cap, RS_WAVES * A.qb), getenv("LCR_RS_NBLK") ? atoi(getenv("LCR_RS_NBLK")) : rq_n_cu() * rq_wg_per_cu()) + 7) / 8 * 8;
  const dim3 grid(nblk);
  if (out_idx64 && out_idx32) LCR_LAUNCH_TIMED((k_radius_query<true, true>), grid, block, 0, st, A, B);
  else if (out_idx64) LCR_LAUNCH_TIMED((k_radius_query<true, false>), grid, block, 0, st, A, B);
  else LCR_LAUNCH_TIMED((k_radius_query<false, true>), grid, block, 0, st, A, B);
  return check_launch("lcr_radius_query");
}

extern "C" int lcr_radius_query_multi(const LcrRadiusQuery* list, int n, int B, void* stream) {
  if (!list || n < 1 || n > RQ_MAX_SEARCHES || B < 1 || B > GRID_MAX_B) {
    set_error("lcr_radius_query_multi: bad argument (1 <= n <= %d searches)", RQ_MAX_SEARCHES);
    return LCR_EARG;
  }
  static const bool no_order = getenv("LCR_RS_NO_ORDER") != nullptr;
  RqMulti m;
  m.n = 0, m.B = B;
  int64_t turns = 0, nq_sum = 0;
  for (int i = 0; i < n; ++i) {
    const LcrRadiusQuery& e = list[i];
    if (!e.qlen || !e.grid_ws || !e.out_idx32 || e.nq_cap < 0 || e.ns_cap < 0 || e.limit < 1 || !(e.radius > 0.f) ||
        e.nq_cap > (int64_t(1) << 31) - 1) {
      set_error("lcr_radius_query_multi: bad search %d", i);
      return LCR_EARG;
    }
    if (e.nq_cap == 0) continue;
    GridLayout L = grid_layout(const_cast<void*>(e.grid_ws), e.ns_cap, B);
    RqSearch& A = m.s[m.n++];
    A.q = e.q, A.qlen = e.qlen, A.nq_cap = e.nq_cap, A.h = L.hdr, A.cell_start = L.cell_start, A.sorted = L.sorted;
    A.r2 = e.radius * e.radius;
    A.limit = e.limit, A.out64 = nullptr, A.out32 = e.out_idx32, A.out_cnt = nullptr, A.q_order = no_order ? nullptr : e.q_order;
    A.qb = rq_qb(e.nq_cap);
    turns += div_up(e.nq_cap, RS_WAVES * A.qb);
    nq_sum += e.nq_cap;
  }
  if (m.n == 0) return LCR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KernelTimerScope timed(KT_RADIUS, st, nq_sum, 0, 0, 4, B);
  const int64_t cap = static_cast<int64_t>(rq_n_cu()) * rq_wg_per_cu();
  const int nblk = static_cast<int>(((turns < cap ? turns : cap) + 7) / 8 * 8);
  LCR_LAUNCH_TIMED(k_radius_query_multi, dim3(nblk), dim3(RS_WAVES * 64), 0, st, m);
  return check_launch("lcr_radius_query_multi");
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
