// grid_subsample.hip — a-1: stack-mode voxel-barycentre subsampling for gfx950, bit-exact with the reference.
//
// Replaces utils.ext.grid_subsampling (utils/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75,
// grid_subsampling_cpu.h:7-21).  The reference walks the points once, accumulating fp32 sums per voxel in a
// std::unordered_map<size_t, SampledData> and emits the map in ITERATION order.  Three things must match exactly:
//   (1) the voxel key arithmetic  — origin = floor(min * float(1/v)) * v; i = floor((p - origin) / v) with a true
//       fp32 division; key = iX + NX*iY + NX*NY*iZ  (:11-35);
//   (2) the fp32 summation order  — input order within a voxel (.h:17-20), then sum * float(1.0 / count) (:46);
//   (3) the emission order        — libstdc++'s hash-table iteration order for keys inserted in first-occurrence order.
//
// MI355X design.  (1)+(2): keys are radix-sorted together with the row index (stable, so each voxel's run is in input
// order), one lane walks each run and adds in order.  (3): the unordered_map is never built; its iteration order is
// reproduced by a closed-form replay of libstdc++'s _Hashtable behaviour, phase by phase of its bucket-count schedule
// (13, 29, 59, ... — a property of libstdc++, measured with g++ 11.4): within a phase with nb buckets, give every
// element a timestamp t (its position in the previous phase's list, or its insertion rank if new); the list after the
// phase is the elements sorted by (min t of their bucket, descending; then t descending).  Each phase is therefore two
// block-wide scans plus bucket atomics — O(n) parallel work instead of a serial pointer chase.
// (Verified against std::unordered_map in tests/test_hashmap_order.py through the host mirror lcr_hashmap_order_host.)
#include <vector>

#include "common.h"
#include "radix_sort.h"

namespace lcr {

constexpr int GS_MAX_B = 64;
constexpr int HM_T = 1024;   // threads of the hash-order workgroup (one workgroup per cloud)

__constant__ const int64_t c_sched[] = {13,        29,        59,         127,        257,      541,      1109,     2357,
                                        5087,      10273,     20753,      42043,      85229,    172933,   351061,   712697,
                                        1447153,   2938679,   5967347,    12117689,   24607243, 49969847, 101473717,
                                        206062531, 418453099, 849745171, 1725584621, 3504127453};
constexpr int N_SCHED = 28;
static const int64_t h_sched[] = {13,        29,        59,         127,        257,      541,      1109,     2357,
                                  5087,      10273,     20753,      42043,      85229,    172933,   351061,   712697,
                                  1447153,   2938679,   5967347,    12117689,   24607243, 49969847, 101473717,
                                  206062531, 418453099, 849745171, 1725584621, 3504127453};

struct GsHeader {
  RadixCtl rx;                       // n, num_passes (must be first: radix kernels read it)
  int      B;
  int      kbits;                    // bits of the voxel key inside the composite sort key
  int64_t  n_cap;
  int64_t  in_off[GS_MAX_B + 1];
  int64_t  out_off[GS_MAX_B + 1];
  uint32_t bb_min[GS_MAX_B][3], bb_max[GS_MAX_B][3];
  float    org[GS_MAX_B][3];
  uint64_t NX[GS_MAX_B], NY[GS_MAX_B];
  uint64_t top[GS_MAX_B];           // NX * NY * NZ when the escape range below is in use, else 0
  int32_t  M[GS_MAX_B];              // distinct voxels per cloud
  int64_t  n_seg;                    // total distinct voxels
};

struct GsLayout {
  GsHeader* hdr;
  uint64_t *keyA, *keyB;
  uint32_t *valA, *valB;
  int32_t*  hist;
  void*     scan_ws;
  int32_t*  head;        // [n]   head flags -> segment index (scan)
  int32_t*  first;       // [n]   is-first-occurrence flags by input row -> rank (scan)
  float*    bary;        // [n,3] barycentre per segment
  uint64_t* seg_key;     // [n]
  uint32_t* seg_first;   // [n]
  int32_t*  seg_start;   // [n]   first sorted position of every run
  uint64_t* ins_key;     // [n]   keys in insertion (first-occurrence) order, per cloud at out_off[b]
  int32_t*  ins_seg;     // [n]
  int32_t*  hm_t;        // [n]   timestamps / positions
  int32_t*  hm_bk;       // [n]   bucket of each element
  int32_t*  hm_mem;      // [n]   bucket member lists
  int32_t*  hm_at;       // [n]
  int32_t*  hm_arr;      // [n]   arrival index of every element inside its bucket
  int32_t*  hm_gmin;     // [bucket_cap]
  int32_t*  hm_cnt;      // [bucket_cap]
  int32_t*  hm_start;    // [bucket_cap]
  size_t    bytes;
};

static inline int64_t bucket_cap_of(int64_t n, int b) { return (n * 9) / 4 + 64 * (b + 1); }

static GsLayout gs_layout(void* ws, int64_t n_cap, int B) {
  GsLayout L;
  Carver c(ws, ~size_t(0));
  const size_t n = static_cast<size_t>(n_cap > 0 ? n_cap : 1);
  L.hdr = c.take<GsHeader>(1);
  L.keyA = c.take<uint64_t>(n);
  L.keyB = c.take<uint64_t>(n);
  L.valA = c.take<uint32_t>(n);
  L.valB = c.take<uint32_t>(n);
  L.hist = c.take<int32_t>(radix_hist_elems(n_cap));
  L.scan_ws = c.take<char>(scan_ws_bytes(static_cast<int64_t>(radix_hist_elems(n_cap)) > n_cap ? radix_hist_elems(n_cap) : n_cap + 1));
  L.head = c.take<int32_t>(n + 1);
  L.first = c.take<int32_t>(n + 1);
  L.bary = c.take<float>(3 * n);
  L.seg_key = c.take<uint64_t>(n);
  L.seg_first = c.take<uint32_t>(n);
  L.seg_start = c.take<int32_t>(n);
  L.ins_key = c.take<uint64_t>(n);
  L.ins_seg = c.take<int32_t>(n);
  L.hm_t = c.take<int32_t>(n);
  L.hm_bk = c.take<int32_t>(n);
  L.hm_mem = c.take<int32_t>(n);
  L.hm_at = c.take<int32_t>(n + 1);
  L.hm_arr = c.take<int32_t>(n);
  const size_t bc = static_cast<size_t>(bucket_cap_of(n_cap, B));
  L.hm_gmin = c.take<int32_t>(bc);
  L.hm_cnt = c.take<int32_t>(bc);
  L.hm_start = c.take<int32_t>(bc + 1);
  L.bytes = c.off;
  return L;
}

// float -> uint64 exactly as x86-64 gcc does for in-range values (negatives wrap through int64)
__device__ __forceinline__ uint64_t f2u64(float v) { return static_cast<uint64_t>(static_cast<int64_t>(v)); }

__global__ void k_gs_init(GsHeader* h, const int64_t* __restrict__ len, int B, int64_t n_cap, uint32_t* status) {
  if (threadIdx.x == 0) {
    int64_t o = 0;
    for (int b = 0; b < B; ++b) {
      h->in_off[b] = o < n_cap ? o : n_cap;      // lengths that overrun the capacity are reported and clamped: every per-cloud
      o += len[b];                               // slice of the work arrays stays inside its allocation
    }
    if (o > n_cap) {
      atomicOr(status, LCR_STATUS_LEN_MISMATCH);
      o = n_cap;
    }
    h->in_off[B] = o;
    h->rx.n = o;
    h->B = B;
    h->n_cap = n_cap;
    h->n_seg = 0;
  }
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    h->M[b] = 0;
    for (int d = 0; d < 3; ++d) {
      h->bb_min[b][d] = 0xffffffffu;
      h->bb_max[b][d] = 0u;
    }
  }
}

__global__ __launch_bounds__(256) void k_gs_bbox(GsHeader* h, const float* __restrict__ xyz, int rs) {
  bbox_accumulate(xyz, h->rx.n, h->in_off, h->B, h->bb_min, h->bb_max, rs);
}

__device__ __forceinline__ int bits_of(uint64_t v) { return v ? 64 - __clzll(static_cast<long long>(v)) : 0; }

__global__ void k_gs_params(GsHeader* h, float voxel, float inv_voxel, int key_bits_hint, uint32_t* status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int B = h->B;
  int kbits = 1;
  for (int b = 0; b < B; ++b) {
    if (h->in_off[b + 1] <= h->in_off[b]) {
      h->NX[b] = h->NY[b] = 1;
      continue;
    }
    float mn[3], mx[3];
    for (int d = 0; d < 3; ++d) {
      mn[d] = ord2f(h->bb_min[b][d]);
      mx[d] = ord2f(h->bb_max[b][d]);
      h->org[b][d] = fmul(floorf(fmul(mn[d], inv_voxel)), voxel);                 // grid_subsampling_cpu.cpp:11
    }
    const uint64_t NX = f2u64(fadd(floorf(fdiv(fsub(mx[0], h->org[b][0]), voxel)), 1.f));   // :13-16
    const uint64_t NY = f2u64(fadd(floorf(fdiv(fsub(mx[1], h->org[b][1]), voxel)), 1.f));   // :17-20
    const uint64_t NZ = f2u64(fadd(floorf(fdiv(fsub(mx[2], h->org[b][2]), voxel)), 1.f));
    h->NX[b] = NX;
    h->NY[b] = NY;
    // largest key of this cloud (if the product overflows 64 bits the key is not sortable together with a cloud id)
    // (an axis on which the WHOLE cloud sits one cell below the origin — a plane or line at a constant coordinate c with
    // floor(c * fl(1/v)) * v > c — has extent 0 after the reference's (size_t) cast + 1; it still contributes the -1 index to every key, so it
    // counts as one cell for the key range.  Found by the op fuzz at seed 63843 of ~90 000: the call reported a key overflow.)
    const unsigned __int128 top = static_cast<unsigned __int128>(NX ? NX : 1) * (NY ? NY : 1) * (NZ ? NZ : 1);
    // A point can land one cell BELOW the origin: origin = floor(min * fl(1/v)) * v is rounded twice, so (min - origin) / v may come
    // out as -epsilon, its floor as -1, and the reference's (size_t) cast turns that into 2^64 - 1 (grid_subsampling_cpu.cpp:32-35;
    // the key then wraps mod 2^64 — a voxel of its own, found by the fuzz at 1 in ~2700 random stacks).  Such keys are -1 - NX - NX*NY
    // at the lowest, so they are sorted in an escape range [top, 3 top + 3) behind the regular keys (sort key = top + (-key)) and turned
    // back into the true 64-bit key for the hash-order replay (k_gs_reduce).  Two more key bits per cloud.
    h->top[b] = 0;
    int kb = 64;
    if ((top >> 64) != 0) kb = 65;
    else if (static_cast<uint64_t>(top) < (1ull << 61)) {
      h->top[b] = static_cast<uint64_t>(top);
      kb = bits_of(3 * static_cast<uint64_t>(top) + 3);
    } else kb = bits_of(static_cast<uint64_t>(top));
    kbits = max(kbits, kb);
  }
  const int cbits = bits_of(static_cast<uint64_t>(B - 1));
  int total = kbits + cbits;
  if (total > 64 || (key_bits_hint > 0 && total > key_bits_hint)) {
    atomicOr(status, LCR_STATUS_KEY_OVERFLOW);
    total = min(total, 64);
    if (key_bits_hint > 0) total = min(total, key_bits_hint);
    kbits = max(total - cbits, 1);
  }
  h->kbits = kbits;
  h->rx.num_passes = radix_passes(total);
}

__global__ __launch_bounds__(256) void k_gs_keys(GsHeader* h, const float* __restrict__ xyz, int rs, float voxel, uint64_t* __restrict__ keyA,
                                                 uint32_t* __restrict__ valA, uint32_t* status) {
  const int B = h->B;
  const int64_t n = h->rx.n;
  const int kbits = h->kbits;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = cloud_of(h->in_off, B, i);
    const uint64_t ix = f2u64(floorf(fdiv(fsub(xyz[rs * i + 0], h->org[b][0]), voxel)));   // :32
    const uint64_t iy = f2u64(floorf(fdiv(fsub(xyz[rs * i + 1], h->org[b][1]), voxel)));   // :33
    const uint64_t iz = f2u64(floorf(fdiv(fsub(xyz[rs * i + 2], h->org[b][2]), voxel)));   // :34
    uint64_t key = ix + h->NX[b] * iy + h->NX[b] * h->NY[b] * iz;                          // :35 (wraps mod 2^64 like size_t)
    const uint64_t top = h->top[b];
    if (top && key >= top) {                                                               // below the origin on some axis: escape range
      const uint64_t neg = 0ull - key;
      if (static_cast<int64_t>(key) < 0 && neg <= 2 * top + 2) key = top + neg;
    }
    // a key that does not fit the promised bits is reported AND truncated: the result of this call is then garbage (the host
    // retries), but the cloud field stays intact, so every cloud's voxels remain inside its own slice of the work arrays
    if (kbits < 64 && (key >> kbits) != 0) atomicOr(status, LCR_STATUS_KEY_OVERFLOW);
    const uint64_t kmask = kbits < 64 ? ((1ull << kbits) - 1ull) : ~0ull;
    keyA[i] = (kbits < 64 ? (static_cast<uint64_t>(b) << kbits) : 0ull) | (key & kmask);
    valA[i] = static_cast<uint32_t>(i);
  }
}

__device__ __forceinline__ const uint64_t* sorted_keys(const GsHeader* h, const uint64_t* a, const uint64_t* b) {
  return (h->rx.num_passes & 1) ? b : a;
}
__device__ __forceinline__ const uint32_t* sorted_vals(const GsHeader* h, const uint32_t* a, const uint32_t* b) {
  return (h->rx.num_passes & 1) ? b : a;
}

__global__ __launch_bounds__(256) void k_gs_heads(const GsHeader* __restrict__ h, const uint64_t* __restrict__ kA,
                                                  const uint64_t* __restrict__ kB, int32_t* __restrict__ head, int32_t* __restrict__ first,
                                                  int64_t n_cap) {
  const int64_t n = h->rx.n;
  const uint64_t* k = sorted_keys(h, kA, kB);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i <= n_cap; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    head[i] = (i < n && (i == 0 || k[i] != k[i - 1])) ? 1 : 0;
    first[i] = 0;                                 // the first-occurrence flags k_gs_reduce sets (saves a separate fill launch)
  }
}

__global__ __launch_bounds__(256) void k_gs_seg_starts(const GsHeader* __restrict__ h, const int32_t* __restrict__ head_scan,
                                                       int32_t* __restrict__ seg_start) {
  const int64_t n = h->rx.n;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    if (head_scan[i + 1] != head_scan[i]) seg_start[head_scan[i]] = static_cast<int32_t>(i);
}

// per-cloud output counts and offsets from the scanned run heads (one thread)
__device__ __forceinline__ void gs_offsets(GsHeader* h, const int32_t* __restrict__ head_scan, int64_t* __restrict__ out_len) {
  int64_t o = 0;
  for (int b = 0; b < h->B; ++b) {
    // the sort key carries the cloud id in its top bits, so cloud b owns sorted positions [in_off[b], in_off[b+1]) and its
    // number of distinct voxels is the number of run heads in that range (no per-run atomics)
    const int64_t lo = min(h->in_off[b], h->rx.n), hi = min(h->in_off[b + 1], h->rx.n);
    h->M[b] = head_scan ? head_scan[hi] - head_scan[lo] : 0;
    h->out_off[b] = o;
    out_len[b] = h->M[b];
    o += h->M[b];
  }
  h->out_off[h->B] = o;
  h->n_seg = o;
}

// one lane per voxel run: in-order fp32 sums (.h:17-20), barycentre = sum * float(1.0 / count) (:46)
__global__ __launch_bounds__(256) void k_gs_reduce(GsHeader* h, const float* __restrict__ xyz, int rs, const uint64_t* __restrict__ kA,
                                                   const uint64_t* __restrict__ kB, const uint32_t* __restrict__ vA,
                                                   const uint32_t* __restrict__ vB, const int32_t* __restrict__ head_scan,
                                                   const int32_t* __restrict__ seg_start, float* __restrict__ bary,
                                                   uint64_t* __restrict__ seg_key, uint32_t* __restrict__ seg_first,
                                                   int32_t* __restrict__ first_flag, int64_t* __restrict__ out_len) {
  if (blockIdx.x == 0 && threadIdx.x == 0) gs_offsets(h, head_scan, out_len);   // read by the insertion / hash-order kernels (later launches)
  const int64_t n = h->rx.n;
  const int64_t nseg = n > 0 ? head_scan[n] : 0;   // exclusive scan of the head flags: total at index n
  const uint64_t* k = sorted_keys(h, kA, kB);
  const uint32_t* v = sorted_vals(h, vA, vB);
  const int kbits = h->kbits;
  for (int64_t seg = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; seg < nseg; seg += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t i = seg_start[seg];
    const int64_t end = seg + 1 < nseg ? seg_start[seg + 1] : n;   // runs are contiguous in the sorted order
    const uint64_t key = k[i];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    // chunks of 8: the index / coordinate loads of a chunk are independent (issued together), only the fp32 additions
    // are serial — they must be, the order is part of the contract
    for (int64_t j = i; j < end; j += 8) {
      uint32_t r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = v[j + u < end ? j + u : end - 1];
      float px[8], py[8], pz[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        px[u] = xyz[rs * static_cast<int64_t>(r[u]) + 0];
        py[u] = xyz[rs * static_cast<int64_t>(r[u]) + 1];
        pz[u] = xyz[rs * static_cast<int64_t>(r[u]) + 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j + u < end) {
          sx = fadd(sx, px[u]);
          sy = fadd(sy, py[u]);
          sz = fadd(sz, pz[u]);
        }
      }
    }
    const int cnt = static_cast<int>(end - i);
    const float rc = static_cast<float>(1.0 / static_cast<double>(cnt));
    bary[3 * seg + 0] = fmul(sx, rc);
    bary[3 * seg + 1] = fmul(sy, rc);
    bary[3 * seg + 2] = fmul(sz, rc);
    const int b = kbits < 64 ? static_cast<int>(key >> kbits) : 0;
    uint64_t vk = kbits < 64 ? (key & ((1ull << kbits) - 1ull)) : key;
    if (h->top[b] && vk >= h->top[b]) vk = 0ull - (vk - h->top[b]);       // escape range -> the reference's wrapped 64-bit key
    seg_key[seg] = vk;
    const uint32_t f = v[i];   // stable sort => first element of the run is the first occurrence
    seg_first[seg] = f;
    first_flag[f] = 1;
  }
}

__global__ void k_gs_offsets(GsHeader* h, const int32_t* __restrict__ head_scan, int64_t* __restrict__ out_len) {   // empty input only
  if (threadIdx.x == 0 && blockIdx.x == 0) gs_offsets(h, head_scan, out_len);
}

__global__ __launch_bounds__(256) void k_gs_insertion(const GsHeader* __restrict__ h, const int32_t* __restrict__ first_scan,
                                                      const uint64_t* __restrict__ seg_key, const uint32_t* __restrict__ seg_first,
                                                      uint64_t* __restrict__ ins_key, int32_t* __restrict__ ins_seg) {
  const int64_t m = h->n_seg;
  const int B = h->B;
  for (int64_t s = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; s < m; s += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const uint32_t f = seg_first[s];
    const int b = cloud_of(h->in_off, B, static_cast<int64_t>(f));
    const int rank = first_scan[f] - first_scan[h->in_off[b]];   // insertion rank inside the cloud
    const int64_t o = h->out_off[b] + rank;
    ins_key[o] = seg_key[s];
    ins_seg[o] = static_cast<int32_t>(s);
  }
}

// ---- libstdc++ unordered_map iteration order, one workgroup per cloud -----------------------------------------------
__device__ __forceinline__ int hm_block_excl_scan(int v, int* total, int* lds) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const int inc = wave_incl_scan(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int k = 0; k < HM_T / 64; ++k) {
    const int s = lds[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// exclusive scan of a[0..n) in place (block-wide, arbitrary n); reversed => scan from the top index downwards.
// Tiles of 4*HM_T elements, four consecutive entries per thread: neighbouring lanes touch neighbouring 16-B pieces (a
// thread-contiguous chunking made every load instruction touch 64 different cache lines).
template <bool REVERSED, typename E>
__device__ void hm_scan_inplace(E* a, int n, int* lds) {
  int carry = 0;
  for (int base = 0; base < n; base += 4 * HM_T) {
    const int i0 = base + 4 * threadIdx.x;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k;
      v[k] = i < n ? static_cast<int>(a[REVERSED ? n - 1 - i : i]) : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s += v[k];
    int tot;
    int run = carry + hm_block_excl_scan(s, &tot, lds);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k;
      if (i < n) a[REVERSED ? n - 1 - i : i] = static_cast<E>(run);
      run += v[k];
    }
    carry += tot;
  }
  __syncthreads();
}

// Every pass below walks its index range in batches of HU entries per thread: the batch's loads are issued together, then
// consumed — one workgroup per cloud is latency-bound, so the number of DEPENDENT memory round trips per pass is what counts.

// key % nb for nb < 2^31: 32-bit remainder when the key fits, else floor(key * (1/nb)) in fp64 (exact quotient within +-1 for
// keys < 2^53, fixed up), else the 64-bit division.  The generic 64-bit urem costs ~100 instructions per element and phase.
__device__ __forceinline__ int hm_bucket(uint64_t key, uint64_t nb64, double inv_nb) {
  if ((key >> 32) == 0 && (nb64 >> 32) == 0) return static_cast<int>(static_cast<uint32_t>(key) % static_cast<uint32_t>(nb64));
  if ((key >> 52) == 0 && (nb64 >> 31) == 0) {
    int64_t q = static_cast<int64_t>(static_cast<double>(key) * inv_nb);
    int64_t r = static_cast<int64_t>(key) - q * static_cast<int64_t>(nb64);
    if (r < 0) r += static_cast<int64_t>(nb64);
    if (r >= static_cast<int64_t>(nb64)) r -= static_cast<int64_t>(nb64);
    return static_cast<int>(r);
  }
  return static_cast<int>(key % nb64);
}

// One phase of the replay with nb buckets: elements [0, hi) are in the table, those below `lo` carry their list position after
// the previous phase in t[], the rest are new (timestamp = insertion rank).  Leaves the new list positions in bk[] (the caller
// swaps t and bk).  The arrays may live in LDS (small phases) or in global memory; the code is the same.
// E = int32_t (arrays in global memory, any size) or uint16_t (LDS phases: every value — bucket id, list position, count — is
// below the phase's bucket count <= 5087); gmin / cnt stay 32-bit, they are the targets of LDS atomics.
// Arrays and their access pattern: key, t, bk, arrv are walked element by element (streamed); gmin, cnt (atomic targets), start, at
// and memt are hit at data-dependent addresses — ~11 of them per element and phase, which one CU issues ~30x faster to LDS than to
// L2/HBM.  Hence three placements of the same code: everything in LDS (phases <= 5087 buckets, all arrays 16-bit but the two atomic
// targets), the SCATTERED arrays in LDS and the streamed ones in global memory (the 10273-bucket phase: 16-bit start / at / memt next
// to 32-bit gmin / cnt = 14 B per bucket, 144 KB; the 20753-bucket phase of clouds up to ~17.6 k voxels: gmin 32-bit, cnt as packed
// 16-bit halves that turn into `start` in place, at 16-bit, memt left in global memory), or everything in global memory (any size).
// CNT16: cnt is a uint16_t array updated through 32-bit atomics on the word that holds it (counts stay below 2^16: a bucket holds
// at most `hi` <= 65535 elements there) and `start` ALIASES it (pass B reads cnt[i] and writes start[i] from the same thread).
template <bool CNT16>
struct HmCnt {
  using type = int32_t;
};
template <>
struct HmCnt<true> {
  using type = uint16_t;
};
__device__ __forceinline__ int hm_cnt_fetch_inc(int32_t* cnt, int i) { return atomicAdd(&cnt[i], 1); }
__device__ __forceinline__ int hm_cnt_fetch_inc(uint16_t* cnt, int i) {
  uint32_t* w = reinterpret_cast<uint32_t*>(cnt + (i & ~1));
  const uint32_t old = atomicAdd(w, (i & 1) ? 0x10000u : 1u);
  return static_cast<int>((old >> (16 * (i & 1))) & 0xffffu);
}

template <int HU, typename ET, typename EM, typename EA, typename ES, bool CNT16 = false>
__device__ __forceinline__ void hm_phase(const uint64_t* __restrict__ key, ET* __restrict__ t, ET* __restrict__ bk,
                                         EM* __restrict__ memt, ET* __restrict__ arrv, EA* __restrict__ at,
                                         int32_t* __restrict__ gmin, typename HmCnt<CNT16>::type* cnt, ES* start, int lo,
                                         int hi, int nb, uint64_t nb64, int* lds) {
  using E = ET;
  const int tid = threadIdx.x;
  const double inv_nb = 1.0 / static_cast<double>(nb64);
  // only buckets that can be hit matter, but all nb are scanned for the member offsets; nb <= 2.25 n + 64
  for (int i = tid; i < nb; i += HM_T) {
    gmin[i] = 0x7fffffff;
    cnt[i] = 0;
  }
  __syncthreads();
  // pass A: bucket of every element; per bucket the earliest member (group creation time) and the member count.  The
  // counting atomic returns the element's arrival index inside its bucket (its slot in the member list, pass C).
  for (int e0 = tid; e0 < hi; e0 += HU * HM_T) {
    uint64_t kk[HU];
    int tt[HU];
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) {
        kk[k] = key[e];
        tt[k] = e >= lo ? e : static_cast<int>(t[e]);            // new elements: timestamp = insertion rank
      }
    }
    int bb[HU], arr[HU];
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) {
        bb[k] = hm_bucket(kk[k], nb64, inv_nb);
        atomicMin(&gmin[bb[k]], tt[k]);
        arr[k] = hm_cnt_fetch_inc(cnt, bb[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) {
        bk[e] = static_cast<E>(bb[k]);
        if (e >= lo) t[e] = static_cast<E>(e);
        at[e] = 0;
        arrv[e] = static_cast<E>(arr[k]);
      }
    }
  }
  __syncthreads();
  // pass B: group sizes keyed by the group's creation time
  for (int i0 = tid; i0 < nb; i0 += HU * HM_T) {
    int c[HU], g[HU];
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int i = i0 + k * HM_T;
      if (i < nb) {
        c[k] = cnt[i];
        g[k] = gmin[i];
      }
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int i = i0 + k * HM_T;
      if (i < nb) {
        if (c[k] > 0) at[g[k]] = static_cast<EA>(c[k]);
        start[i] = static_cast<ES>(c[k]);
      }
    }
  }
  __syncthreads();
  hm_scan_inplace<true, EA>(at, hi, lds);     // at[tt] = number of elements in groups created AFTER time tt
  hm_scan_inplace<false, ES>(start, nb, lds);  // member-list offsets per bucket
  // pass C: member lists (timestamps), and per bucket the list position of its group (replaces gmin)
  for (int e0 = tid; e0 < hi; e0 += HU * HM_T) {
    int bb[HU], tt[HU], ar[HU], s0[HU];
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) {
        bb[k] = bk[e];
        tt[k] = t[e];
        ar[k] = arrv[e];
      }
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) s0[k] = start[bb[k]];
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) memt[s0[k] + ar[k]] = static_cast<EM>(tt[k]);
    }
  }
  for (int i0 = tid; i0 < nb; i0 += HU * HM_T) {
    int g[HU];
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int i = i0 + k * HM_T;
      g[k] = i < nb ? gmin[i] : 0x7fffffff;
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int i = i0 + k * HM_T;
      if (i < nb && g[k] != 0x7fffffff) gmin[i] = at[g[k]];
    }
  }
  __syncthreads();
  // pass D: new list position = (elements of groups created later) + (newer members of the own bucket)
  for (int e0 = tid; e0 < hi; e0 += HU * HM_T) {
    int bb[HU], te[HU], s0[HU], s1[HU], a0[HU], r[HU];
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      bb[k] = 0;
      te[k] = 0;
      if (e < hi) {
        bb[k] = bk[e];
        te[k] = t[e];
      }
    }
    int maxlen = 0;
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      s0[k] = s1[k] = a0[k] = 0;
      if (e < hi) {
        s0[k] = start[bb[k]];
        s1[k] = (bb[k] + 1 < nb) ? start[bb[k] + 1] : hi;
        a0[k] = gmin[bb[k]];
      }
      r[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) maxlen = max(maxlen, s1[k] - s0[k]);
    for (int j = 0; j < maxlen; ++j) {        // buckets hold one or two elements almost always
#pragma unroll
      for (int k = 0; k < HU; ++k)
        if (s0[k] + j < s1[k]) r[k] += static_cast<int>(memt[s0[k] + j]) > te[k];   // newer members of the bucket come first
    }
#pragma unroll
    for (int k = 0; k < HU; ++k) {
      const int e = e0 + k * HM_T;
      if (e < hi) bk[e] = static_cast<E>(a0[k] + r[k]);        // position in the list after this phase
    }
  }
  __syncthreads();
}

// Emit the barycentres in iteration order: pos[rank] = position of the rank-th inserted voxel.  Two dependent gathers per
// element (segment of the rank, then its barycentre): four elements per thread are requested together.
template <typename E>
__device__ __forceinline__ void hm_emit(const E* __restrict__ pos, const int32_t* __restrict__ ins_seg, const float* __restrict__ bary,
                                        float* __restrict__ out_xyz, int64_t o, int n) {
  constexpr int EU = 4;
  for (int e0 = threadIdx.x; e0 < n; e0 += EU * HM_T) {
    int seg[EU], dst[EU];
#pragma unroll
    for (int k = 0; k < EU; ++k) {
      const int e = e0 + k * HM_T < n ? e0 + k * HM_T : n - 1;
      seg[k] = ins_seg[o + e];
      dst[k] = static_cast<int>(pos[e]);
    }
    float b[EU][3];
#pragma unroll
    for (int k = 0; k < EU; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) b[k][d] = bary[3 * static_cast<int64_t>(seg[k]) + d];
#pragma unroll
    for (int k = 0; k < EU; ++k)
      if (e0 + k * HM_T < n) {
#pragma unroll
        for (int d = 0; d < 3; ++d) out_xyz[3 * (o + dst[k]) + d] = b[k][d];
      }
  }
}

// The first HM_LDS_PHASES phases (up to 5087 buckets) run on LDS-resident arrays: a phase is eight block-wide passes with a
// barrier between them, full of scattered accesses and atomics — one CU issues those ~30x faster to LDS than to L2/HBM.  Every
// value of these phases is below 5087, so six of the eight per-entry arrays are 16-bit (the two atomic targets stay 32-bit): 28 B
// per entry with the 8-B key, 142 KB of the CU's 160 KB.  (With 32-bit arrays only the phases up to 2357 buckets fitted: the 5087
// phase of every stage ran out of global memory.)
constexpr int HM_LDS_PHASES = 9;
constexpr int HM_LC = 5087 + 9;         // entries per LDS array (c_sched[HM_LDS_PHASES - 1], padded to a multiple of 8)
constexpr size_t HM_LDS_SMALL = static_cast<size_t>(HM_LC) * (2 * sizeof(int32_t) + 6 * sizeof(uint16_t) + sizeof(uint64_t));   // all arrays of the small phases
constexpr size_t HM_LDS_BYTES = 160 * 1024 - 1024;   // the whole CU's LDS but the static part: the 10273- / 20753-bucket phases keep their scattered arrays there
static_assert(HM_LDS_SMALL <= HM_LDS_BYTES, "LDS phases do not fit");

__global__ __launch_bounds__(HM_T) void k_gs_hashorder(const GsHeader* __restrict__ h, const uint64_t* __restrict__ ins_key,
                                                       const int32_t* __restrict__ ins_seg, const float* __restrict__ bary,
                                                       int32_t* __restrict__ hm_t, int32_t* __restrict__ hm_bk,
                                                       int32_t* __restrict__ hm_mem, int32_t* __restrict__ hm_at,
                                                       int32_t* __restrict__ hm_arr, int32_t* __restrict__ hm_gmin, int32_t* __restrict__ hm_cnt,
                                                       int32_t* __restrict__ hm_start, float* __restrict__ out_xyz) {
  __shared__ int lds[HM_T / 64];
  extern __shared__ __align__(16) unsigned char hm_dyn[];
  const int b = blockIdx.x;
  const int n = h->M[b];
  if (n <= 0) return;
  const int64_t o = h->out_off[b];
  const uint64_t* key = ins_key + o;
  const int tid = threadIdx.x;

  int lo = 0, p = 0;
  {
    uint64_t* lkey = reinterpret_cast<uint64_t*>(hm_dyn);
    int32_t* lgmin = reinterpret_cast<int32_t*>(lkey + HM_LC);
    int32_t* lcnt = lgmin + HM_LC;
    uint16_t* lt = reinterpret_cast<uint16_t*>(lcnt + HM_LC);
    uint16_t* lbk = lt + HM_LC;
    uint16_t* lmem = lbk + HM_LC;
    uint16_t* larr = lmem + HM_LC;
    uint16_t* lat = larr + HM_LC;
    uint16_t* lstart = lat + HM_LC;
    const int nl = min(n, static_cast<int>(c_sched[HM_LDS_PHASES - 1]));
    for (int e = tid; e < nl; e += HM_T) lkey[e] = key[e];
    __syncthreads();
    bool done = false;
    for (; p < HM_LDS_PHASES; ++p) {
      const int nb = static_cast<int>(c_sched[p]);
      const int hi = min(n, nb);
      hm_phase<4, uint16_t, uint16_t, uint16_t, uint16_t>(lkey, lt, lbk, lmem, larr, lat, lgmin, lcnt, lstart, lo, hi, nb, static_cast<uint64_t>(nb), lds);
      uint16_t* sw = lt;                         // the new positions become the next phase's timestamps
      lt = lbk;
      lbk = sw;
      lo = hi;
      if (hi >= n) {
        done = true;
        break;
      }
    }
    if (done) {
      hm_emit(lt, ins_seg, bary, out_xyz, o, n);
      return;
    }
    for (int e = tid; e < lo; e += HM_T) hm_t[o + e] = static_cast<int32_t>(lt[e]);
    __syncthreads();
  }

  int32_t* t = hm_t + o;         // list position of every element after the previous phase (ping-pongs with bk)
  int32_t* bk = hm_bk + o;       // bucket of every element in this phase, then its new list position
  int32_t* memt = hm_mem + o;    // member lists: the TIMESTAMPS of every bucket's elements, bucket after bucket
  int32_t* arrv = hm_arr + o;    // arrival index of every element inside its bucket (its slot in the member list)
  int32_t* at = hm_at + o;       // first: group size at the group's creation time; after the scan: elements in newer groups
  const int64_t bo = (h->in_off[b] * 9) / 4 + 64 * b;   // this cloud's slice of the bucket arrays
  int32_t* gmin = hm_gmin + bo;  // earliest member (group creation time); after pass B2: list position of the group's head
  int32_t* cnt = hm_cnt + bo;
  int32_t* start = hm_start + bo;
  for (; p < N_SCHED; ++p) {
    const int64_t nb64 = c_sched[p];
    const int hi = static_cast<int>(min(static_cast<int64_t>(n), nb64));
    const int nb = static_cast<int>(min(nb64, static_cast<int64_t>(2147483647)));
    // placement of the SCATTERED arrays (see hm_phase): all five in LDS, or gmin + packed cnt/start + at in LDS, or none
    const size_t nbe = static_cast<size_t>(nb + 2) & ~size_t(1), hie = static_cast<size_t>(hi + 2) & ~size_t(1);   // even entry counts: 4-B aligned arrays
    const size_t need_all = nbe * (4 + 4 + 2) + hie * (2 + 2);
    const size_t need_part = nbe * (4 + 2) + hie * 2;
    if (nb64 < 65536 && need_all <= HM_LDS_BYTES) {
      int32_t* lgmin = reinterpret_cast<int32_t*>(hm_dyn);
      int32_t* lcnt = lgmin + nbe;
      uint16_t* lstart = reinterpret_cast<uint16_t*>(lcnt + nbe);
      uint16_t* lat = lstart + nbe;
      uint16_t* lmem = lat + hie;
      hm_phase<4, int32_t, uint16_t, uint16_t, uint16_t>(key, t, bk, lmem, arrv, lat, lgmin, lcnt, lstart, lo, hi, nb, static_cast<uint64_t>(nb64), lds);
    } else if (nb64 < 65536 && need_part <= HM_LDS_BYTES) {
      int32_t* lgmin = reinterpret_cast<int32_t*>(hm_dyn);
      uint16_t* lcnt = reinterpret_cast<uint16_t*>(lgmin + nbe);     // becomes `start` in place
      uint16_t* lat = lcnt + nbe;
      hm_phase<4, int32_t, int32_t, uint16_t, uint16_t, true>(key, t, bk, memt, arrv, lat, lgmin, lcnt, lcnt, lo, hi, nb, static_cast<uint64_t>(nb64), lds);
    } else {
      hm_phase<4, int32_t, int32_t, int32_t, int32_t>(key, t, bk, memt, arrv, at, gmin, cnt, start, lo, hi, nb, static_cast<uint64_t>(nb64), lds);
    }
    int32_t* sw = t;
    t = bk;
    bk = sw;
    lo = hi;
    if (hi >= n) break;
  }
  hm_emit(t, ins_seg, bary, out_xyz, o, n);
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_grid_subsample_ws_bytes(int64_t n_cap, int B, size_t* bytes) {
  if (!bytes || n_cap < 0 || B < 1 || B > GS_MAX_B) return LCR_EARG;
  *bytes = gs_layout(nullptr, n_cap, B).bytes;
  return LCR_OK;
}

extern "C" int lcr_grid_subsample_rows(const float* xyz, int row_floats, const int64_t* len, int B, int64_t n_cap, float voxel, int key_bits_hint,
                                       float* out_xyz, int64_t* out_len, uint32_t* status, void* ws, size_t ws_bytes, void* stream);
extern "C" int lcr_grid_subsample_ex(const float* xyz, const int64_t* len, int B, int64_t n_cap, float voxel, int key_bits_hint,
                                     float* out_xyz, int64_t* out_len, uint32_t* status, void* ws, size_t ws_bytes, void* stream) {
  return lcr_grid_subsample_rows(xyz, 3, len, B, n_cap, voxel, key_bits_hint, out_xyz, out_len, status, ws, ws_bytes, stream);
}

// The same on rows of `row_floats` >= 3 floats whose first three are x, y, z: a KITTI velodyne scan (f32 [N, 4]: x, y, z, intensity;
// data/Kitti/downsample_pcd.py:21, dataset_overlap_online.py:245 takes [:, :3] on the host) is consumed as it lies in memory — the three
// kernels that touch the input (bounding box, voxel keys, in-order sums) step over the unused columns; the output is f32 [M, 3].
extern "C" int lcr_grid_subsample_rows(const float* xyz, int row_floats, const int64_t* len, int B, int64_t n_cap, float voxel, int key_bits_hint,
                                       float* out_xyz, int64_t* out_len, uint32_t* status, void* ws, size_t ws_bytes, void* stream) {
  if (row_floats < 3 || row_floats > 64) {
    set_error("lcr_grid_subsample: rows of %d floats (3 .. 64: x, y, z first)", row_floats);
    return LCR_EARG;
  }
  const int rs = row_floats;
  if (!len || !out_len || !status || !ws || B < 1 || B > GS_MAX_B || n_cap < 0 || !(voxel > 0.f) || key_bits_hint < 0 ||
      key_bits_hint > 64) {
    set_error("lcr_grid_subsample: bad argument");
    return LCR_EARG;
  }
  if (n_cap > (int64_t(1) << 31) - 2) {
    set_error("lcr_grid_subsample: more than 2^31-2 points");
    return LCR_EARG;
  }
  GsLayout L = gs_layout(ws, n_cap, B);
  if (L.bytes > ws_bytes) {
    set_error("lcr_grid_subsample: workspace too small (%zu < %zu)", ws_bytes, L.bytes);
    return LCR_ESPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float inv_voxel = static_cast<float>(1.0 / static_cast<double>(voxel));   // `1. / voxel_size` narrowed by operator*(PointXYZ,float)
  const int nblk = n_cap > 0 ? min(div_up(n_cap, 256), 2048) : 1;
  hipLaunchKernelGGL(k_gs_init, dim3(1), dim3(64), 0, st, L.hdr, len, B, n_cap, status);
  if (n_cap == 0) {
    hipLaunchKernelGGL(k_gs_offsets, dim3(1), dim3(64), 0, st, L.hdr, static_cast<const int32_t*>(nullptr), out_len);
    return check_launch("lcr_grid_subsample");
  }
  // one workgroup per CU at most: every workgroup ends with 6 atomics on its cloud's box, and same-address atomics serialise
  hipLaunchKernelGGL(k_gs_bbox, dim3(n_cap > 0 ? min(div_up(n_cap, 1024), 512) : 1), dim3(256), 0, st, L.hdr, xyz, rs);
  hipLaunchKernelGGL(k_gs_params, dim3(1), dim3(64), 0, st, L.hdr, voxel, inv_voxel, key_bits_hint, status);
  hipLaunchKernelGGL(k_gs_keys, dim3(nblk), dim3(256), 0, st, L.hdr, xyz, rs, voxel, L.keyA, L.valA, status);
  const int max_passes = radix_passes(key_bits_hint > 0 ? key_bits_hint : 64);
  int rc = radix_sort_pairs(&L.hdr->rx, L.keyA, L.keyB, L.valA, L.valB, n_cap, max_passes, L.hist, L.scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gs_heads, dim3(nblk), dim3(256), 0, st, L.hdr, L.keyA, L.keyB, L.head, L.first, n_cap);
  rc = exclusive_scan_i32(L.head, L.head, n_cap + 1, nullptr, L.scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gs_seg_starts, dim3(nblk), dim3(256), 0, st, L.hdr, L.head, L.seg_start);
  hipLaunchKernelGGL(k_gs_reduce, dim3(nblk), dim3(256), 0, st, L.hdr, xyz, rs, L.keyA, L.keyB, L.valA, L.valB, L.head, L.seg_start, L.bary,
                     L.seg_key, L.seg_first, L.first, out_len);
  rc = exclusive_scan_i32(L.first, L.first, n_cap + 1, nullptr, L.scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gs_insertion, dim3(nblk), dim3(256), 0, st, L.hdr, L.first, L.seg_key, L.seg_first, L.ins_key, L.ins_seg);
  static DynLds hm_opt_in;
  if (hm_opt_in.need(reinterpret_cast<const void*>(&k_gs_hashorder), HM_LDS_BYTES) != hipSuccess) {
    set_error("lcr_grid_subsample: cannot reserve %zu B of LDS for the hash-order kernel", HM_LDS_BYTES);
    return LCR_EHIP;
  }
  hipLaunchKernelGGL(k_gs_hashorder, dim3(B), dim3(HM_T), HM_LDS_BYTES, st, L.hdr, L.ins_key, L.ins_seg, L.bary, L.hm_t, L.hm_bk, L.hm_mem, L.hm_at, L.hm_arr,
                     L.hm_gmin, L.hm_cnt, L.hm_start, out_xyz);
  return check_launch("lcr_grid_subsample");
}

extern "C" int lcr_grid_subsample(const float* xyz, const int64_t* len, int B, int64_t n_cap, float voxel, float* out_xyz,
                                  int64_t* out_len, uint32_t* status, void* ws, size_t ws_bytes, void* stream) {
  return lcr_grid_subsample_ex(xyz, len, B, n_cap, voxel, 0, out_xyz, out_len, status, ws, ws_bytes, stream);
}

// Host mirror of the phase algorithm of k_gs_hashorder (same steps, serial): order[j] = insertion rank of the j-th
// element in std::unordered_map iteration order.  Lets the CPU-only test suite pin the algorithm against the real
// container without a GPU; it is not used by the device path.
extern "C" int lcr_hashmap_order_host(const uint64_t* keys, int64_t n, int64_t* order) {
  if (n < 0 || (n > 0 && (!keys || !order))) return LCR_EARG;
  if (n == 0) return LCR_OK;
  std::vector<int64_t> t(n), pos(n), bk(n), mem(n);
  int64_t lo = 0;
  for (int p = 0; p < N_SCHED; ++p) {
    const int64_t nb = h_sched[p];
    const int64_t hi = n < nb ? n : nb;
    for (int64_t e = lo; e < hi; ++e) t[e] = e;                       // new elements: timestamp = insertion rank
    std::vector<int64_t> gmin(nb, INT64_MAX), cnt(nb, 0), start(nb + 1, 0), at(hi + 1, 0);
    for (int64_t e = 0; e < hi; ++e) {
      bk[e] = static_cast<int64_t>(keys[e] % static_cast<uint64_t>(nb));
      if (t[e] < gmin[bk[e]]) gmin[bk[e]] = t[e];
      cnt[bk[e]]++;
    }
    for (int64_t i = 0; i < nb; ++i)
      if (cnt[i] > 0) at[gmin[i]] = cnt[i];
    int64_t run = 0;
    for (int64_t i = hi - 1; i >= 0; --i) {                            // reversed exclusive scan
      const int64_t v = at[i];
      at[i] = run;
      run += v;
    }
    run = 0;
    for (int64_t i = 0; i < nb; ++i) {                                 // member-list offsets
      start[i] = run;
      run += cnt[i];
    }
    start[nb] = run;
    std::vector<int64_t> cur(start.begin(), start.end() - 1);
    for (int64_t e = 0; e < hi; ++e) mem[cur[bk[e]]++] = e;
    for (int64_t e = 0; e < hi; ++e) {
      int64_t r = 0;
      for (int64_t s0 = start[bk[e]]; s0 < start[bk[e] + 1]; ++s0) r += t[mem[s0]] > t[e];
      pos[e] = at[gmin[bk[e]]] + r;
    }
    for (int64_t e = 0; e < hi; ++e) t[e] = pos[e];
    lo = hi;
    if (hi >= n) break;
  }
  for (int64_t e = 0; e < n; ++e) order[t[e]] = e;
  return LCR_OK;
}
