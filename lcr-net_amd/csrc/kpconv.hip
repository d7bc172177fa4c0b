// kpconv.hip — a-4: the gather / kernel-point-influence / aggregation half of rigid KPConv, plus the small ops around it.
//
// Reference: KPConv.forward, experiments/lcrnet/modules/kpconv/kpconv.py:79-122 — gathers (M,H,3) and (M,H,C), builds
// (M,H,15,3) differences, (M,15,H) influences, bmm -> (M,15,C), then the (15,C,Cout) contraction, every intermediate
// materialised in HBM (~2.6 GB per scan).  Here one wavefront owns one query point (k_kpconv_aggregate_vec):
//   1. lanes = neighbours: the valid ones (shadow rows skipped instead of a +1e6 pad, kpconv.py:91) are compacted in order into
//      LDS as (relative position, index) — 16 B each;
//   2. per 4 neighbours, D[16 kernel points x C] += W[16 x 4] * F[4 x C] on v_mfma_f32_16x16x4_f32 (exact fp32): lane l evaluates
//      the ONE linear influence max(0, 1 - |y - k|/sigma) (:96-99) it feeds to the matrix core — neighbour (l>>4), kernel point
//      (l&15) — and fetches 4 consecutive channels of that neighbour's feature row with one 16-B load (:104);
//   3. writes the (15*C) row that lcr_gemm_f32 contracts with the (15*C, Cout) weights (:108-110) and the neighbour
//      count used by its epilogue (:113-116: neighbours whose feature row sums to > 0).
// Only the (M, 15*C) aggregate goes through HBM/Infinity-Cache between the two halves.  Earlier variants (VALU accumulation with
// the influences parked in LDS; MFMA with scalar 4-B gathers) are described with their measurements in LABNOTES.md §4.1.
// encoder1_1 (C_in = 1, backbone4.py:15) is fully fused in k_kpconv_cin1.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace lcr {

constexpr int KP_K = 15;       // kernel points (cfg.backbone.kernel_size)
constexpr int KP_HMAX = 128;   // neighbour columns supported per query
constexpr int KP_WAVES = 4;

struct KPoints {
  float p[KP_K][3];
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// Query -> workgroup mapping of the gather kernels.  Workgroups go round-robin to the 8 XCDs (id % 8), each with its own,
// non-coherent L2.  With a plain grid-stride walk all XCDs sweep the (spatially ordered) query list side by side, so every XCD
// pulls the SAME support rows into its own L2 — the gathered table crosses the fabric up to eight times (PMC: max-pool fetched
// 445 MB for a 65 MB table).  Here XCD x walks the x-th CONTIGUOUS eighth of the list: its L2 only ever holds that region's rows.
struct XcdBand {
  int64_t begin, end, stride;   // this wavefront's first query, the end of its band, the step
};
__device__ __forceinline__ XcdBand xcd_band(int64_t M, int wave_in_block, int waves_per_block) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;      // the grid is a multiple of 8
  const int64_t per = (M + 7) / 8;
  XcdBand b;
  b.begin = xcd * per + static_cast<int64_t>(slot) * waves_per_block + wave_in_block;
  b.end = min(M, (xcd + 1) * per);
  b.stride = static_cast<int64_t>(nslots) * waves_per_block;
  return b;
}

// The aggregation kernel.  A lane fetches V = 4 (C >= 64)
// or 2 (C = 32) CONSECUTIVE channels of its neighbour's row with one 16-B / 8-B load, so a 4-neighbour step of C = 64 is one
// load instruction per lane (4 full 256-B rows per wavefront instruction) instead of four, and the V components feed V MFMA tiles
// whose column `col` is channel q*16V + V*col + v: the output row is then written with V-wide stores.  Loads run D steps ahead
// of the matrix core in a register ring (the kernel is bound by the L2 latency of the row gathers).
template <int V> struct VecOf;
template <> struct VecOf<4> { typedef float4 type; };
template <> struct VecOf<2> { typedef float2 type; };
__device__ __forceinline__ float vget(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ float vget(const float2& v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float4 vmake(const float (&a)[4]) { return make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ float2 vmake(const float (&a)[2]) { return make_float2(a[0], a[1]); }

// OFF32 (Ns * C * 4 < 2^32, chosen by the host): the list carries the BYTE offset of the neighbour's feature row instead of its index, and a
// step's gather is scalar base + (offset + lane column): one 32-bit add instead of a 64-bit multiply-add per step.
template <typename IdxT, int C, bool OFF32>
__global__ __launch_bounds__(KP_WAVES * 64) void k_kpconv_aggregate_vec(const float* __restrict__ s_feats, const uint8_t* __restrict__ s_pos,
                                                                        const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                                        const IdxT* __restrict__ idx, int64_t M, int64_t Ns, int H, KPoints kp,
                                                                        float sigma, float* __restrict__ A, float* __restrict__ nn,
                                                                        const int32_t* __restrict__ order, int valid_first) {
  constexpr int V = C >= 64 ? 4 : 2;
  constexpr int NL = C / (16 * V);           // vector loads per lane per 4-neighbour step
  constexpr int D = NL == 1 ? 4 : 2;         // steps in flight (register ring)
  typedef typename VecOf<V>::type VecT;
  constexpr int PAD = 4 * D + 4;             // list entries the register ring may read beyond the last neighbour
  __shared__ __attribute__((aligned(16))) float4 s_rel[KP_WAVES][KP_HMAX + PAD];   // (dx, dy, dz, bits(index)) per valid neighbour
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const int sub = lane >> 4, col = lane & 15;
  const float inv_sigma = 1.f / sigma;
  float kx = 0.f, ky = 0.f, kz = 0.f;
#pragma unroll
  for (int k = 0; k < KP_K; ++k)
    if (col == k) {
      kx = kp.p[k][0];
      ky = kp.p[k][1];
      kz = kp.p[k][2];
    }
  const XcdBand band = xcd_band(M, w, KP_WAVES);
  for (int64_t t = band.begin; t < band.end; t += band.stride) {
    const int64_t m = order ? order[t] : t;
    const float qx = q_pts[3 * m], qy = q_pts[3 * m + 1], qz = q_pts[3 * m + 2];
    int n = 0, cnt = 0;
    for (int h0 = 0; h0 < H; h0 += 64) {
      const int h = h0 + lane;
      bool ok, positive = false;
      uint64_t mk;
      if (OFF32 && sizeof(IdxT) == 4) {
        // 32-bit list building (host-checked: Ns * C * 4 < 2^32 and M * H < 2^30): one unsigned compare for the validity test, every gather
        // as scalar base + 32-bit byte offset — the 64-bit form below spends ~25 more VALU instructions per chunk on address arithmetic
        const uint32_t ju = h < H ? static_cast<uint32_t>(reinterpret_cast<const int32_t*>(idx)[static_cast<uint32_t>(m) * static_cast<uint32_t>(H) + static_cast<uint32_t>(h)])
                                  : static_cast<uint32_t>(Ns);
        ok = ju < static_cast<uint32_t>(Ns);
        mk = wave_ballot(ok);
        if (ok) {
          const int slot = n + mbcnt_lt(mk);
          const float* sp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_pts) + ju * 12u);
          s_rel[w][slot] = make_float4(sp[0] - qx, sp[1] - qy, sp[2] - qz, __uint_as_float(ju * static_cast<uint32_t>(C * 4)));
          positive = s_pos[ju] != 0;
        }
      } else {
        int64_t j = Ns;
        if (h < H) j = static_cast<int64_t>(idx[m * H + h]);
        ok = j >= 0 && j < Ns;
        mk = wave_ballot(ok);
        if (ok) {
          const int slot = n + mbcnt_lt(mk);
          s_rel[w][slot] = make_float4(s_pts[3 * j] - qx, s_pts[3 * j + 1] - qy, s_pts[3 * j + 2] - qz,
                                       __uint_as_float(OFF32 ? static_cast<uint32_t>(j) * static_cast<uint32_t>(C * 4) : static_cast<uint32_t>(j)));
          positive = s_pos[j] != 0;
        }
      }
      n += __popcll(mk);
      cnt += __popcll(wave_ballot(positive));      // scalar popcount of a lane mask instead of a six-step cross-lane sum
      // rows of a radius search are "valid first, padding last": a chunk with a hole is the row's last one — H = 65 ... 80 rows
      // (stages 1-3) whose ball holds <= 64 supports skip their second, all-padding chunk (LCR_KP_VALID_FIRST; wave-uniform)
      if (valid_first && mk != ~0ull) break;
    }
    // The list is padded with far-away neighbours (relative position 1e6: linear influence exactly 0 for every kernel point; feature
    // row 0, multiplied by that 0): the steps then need neither an index clamp nor an in-range mask — 6 of ~19 VALU instructions per
    // 4-neighbour step in a kernel that is VALU-issue-bound for C <= 64.  (Lane column 15 is no kernel point: its influence is
    // garbage, accumulates into row 15 of D and is never stored.)
    if (lane < PAD) s_rel[w][n + lane] = make_float4(1e6f, 1e6f, 1e6f, __uint_as_float(0u));
    wave_lds_sync();

    floatx4 acc[NL][V];
#pragma unroll
    for (int q = 0; q < NL; ++q)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[q][v] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int steps = (n + 3) >> 2;
    float4 p[D];
    VecT f[D][NL];
    auto fetch = [&](int s, float4& pp, VecT (&ff)[NL]) {
      pp = s_rel[w][4 * s + sub];
      if (OFF32) {
        const uint32_t off = __float_as_uint(pp.w) + static_cast<uint32_t>(V * col * 4);
#pragma unroll
        for (int q = 0; q < NL; ++q) ff[q] = *reinterpret_cast<const VecT*>(reinterpret_cast<const char*>(s_feats) + (off + static_cast<uint32_t>(q * 16 * V * 4)));
      } else {
        const float* r = s_feats + static_cast<int64_t>(__float_as_uint(pp.w)) * C + V * col;
#pragma unroll
        for (int q = 0; q < NL; ++q) ff[q] = *reinterpret_cast<const VecT*>(r + q * 16 * V);
      }
    };
    // Branch-free steady state: steps are rounded up to a multiple of D and every fetch is issued unconditionally (beyond the
    // list's end it reads padding: row 0, a cache hit, with influence 0).  One basic block per
    // trip keeps the s_waitcnt counts exact, so the D-step register ring really runs D steps ahead; with conditional fetches
    // the compiler waited for vmcnt(0) — the load it had just issued — before every MFMA group.
    if (n > 0) {
      auto compute = [&](int s, const float4& pp, const VecT (&ff)[NL]) {
        const float ex = pp.x - kx, ey = pp.y - ky, ez = pp.z - kz;
        const float wv = fmaxf(fmaf(-__builtin_amdgcn_sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex))), inv_sigma, 1.f), 0.f);
#pragma unroll
        for (int q = 0; q < NL; ++q)
#pragma unroll
          for (int v = 0; v < V; ++v) acc[q][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, vget(ff[q], v), acc[q][v], 0, 0, 0);
      };
#pragma unroll
      for (int d = 0; d < D; ++d) fetch(d, p[d], f[d]);
      int s0 = 0;
      for (; s0 + D <= steps; s0 += D) {                    // full trips: D x (compute, refill), no branch inside
#pragma unroll
        for (int d = 0; d < D; ++d) {
          compute(s0 + d, p[d], f[d]);
          fetch(s0 + d + D, p[d], f[d]);
        }
      }
#pragma unroll
      for (int d = 0; d < D - 1; ++d)                       // the last steps % D steps: their rows are already on the way
        if (s0 + d < steps) compute(s0 + d, p[d], f[d]);
    }
    float* out = A + m * (KP_K * C) + V * col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 4 * sub + r;
      if (k < KP_K) {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
          float a[V];
#pragma unroll
          for (int v = 0; v < V; ++v) a[v] = acc[q][v][r];
          *reinterpret_cast<VecT*>(out + k * C + q * 16 * V) = vmake(a);
        }
      }
    }
    if (lane == 0) nn[m] = static_cast<float>(cnt > 1 ? cnt : 1);
    wave_lds_sync();
  }
}

// ---- whole KPConv for C_in = C_out = 32 (the two widest query sets of the encoder: 1_2 at stage 0, 2_1 at stage 1) --------------
// The (M, 15*C) aggregate of these two blocks is the largest intermediate of the pass (480 floats per query: 245 MB written and
// read back at M = 128 k).  Here it only ever exists as a 16-query tile in LDS:
//   * a workgroup (4 wavefronts) takes 16 consecutive queries of its XCD band; every wavefront aggregates 4 of them exactly as
//     k_kpconv_aggregate_vec does (D[16 x 32] on v_mfma_f32_16x16x4_f32) and parks each D as one 480-float row of the tile;
//   * the (15*32) x 32 weights never touch LDS: wavefront w keeps the rows [120 w, 120 w + 120) in 60 registers per lane for the
//     whole launch (split-K over the wavefronts; inside a wavefront the MFMA k-slot g of step s is row 120 w + 30 g + s, so a
//     lane reads its 30 A values as 15 contiguous 8-byte LDS loads) and contracts the tile with 60 MFMAs;
//   * the four partial 16 x 32 tiles are folded through LDS by 256 threads (2 outputs each), divided by the neighbour count,
//     biased (kpconv.py:108-116) and stored; the GroupNorm sums of the output are kept per thread across the tiles of a workgroup
//     and leave as one fp64 atomic per (wavefront, column) at the end (earlier when the GroupNorm segment changes).
constexpr int KF_C = 32, KF_Q = 16, KF_QW = KF_Q / KP_WAVES, KF_KK = KP_K * KF_C, KF_LD = KF_KK + 4, KF_KW = KF_KK / KP_WAVES;
constexpr int KF_MAX_SEG = 64;

template <typename IdxT>
__global__ __launch_bounds__(KP_WAVES * 64, 3) void k_kpconv_fused32(const float* __restrict__ s_feats, const uint8_t* __restrict__ s_pos,
                                                                     const float* __restrict__ q_pts, const float* __restrict__ s_pts,
                                                                     const IdxT* __restrict__ idx, int64_t M, int64_t Ns, int H, KPoints kp,
                                                                     float sigma, const float* __restrict__ W, const float* __restrict__ bias,
                                                                     float* __restrict__ out, const int64_t* __restrict__ seg_len, int S,
                                                                     int groups, double* __restrict__ stats, const int32_t* __restrict__ order) {
  constexpr int C = KF_C, V = 2, D = 4;
  __shared__ __attribute__((aligned(16))) float s_D[KF_Q * KF_LD];
  __shared__ __attribute__((aligned(16))) float s_P[KP_WAVES][KF_Q][C];
  __shared__ __attribute__((aligned(16))) float4 s_rel[KP_WAVES][KP_HMAX + 8];
  __shared__ int64_t s_m[KF_Q], s_start[KF_MAX_SEG + 1];
  __shared__ float s_nn[KF_Q];
  __shared__ int s_seg[KF_Q];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane >> 4, col = lane & 15;
  const float inv_sigma = 1.f / sigma;
  float kx = 0.f, ky = 0.f, kz = 0.f;
#pragma unroll
  for (int k = 0; k < KP_K; ++k)
    if (col == k) {
      kx = kp.p[k][0];
      ky = kp.p[k][1];
      kz = kp.p[k][2];
    }
  const bool real_k = col < KP_K;
  if (threadIdx.x == 0) {
    int64_t o = 0;
    for (int i = 0; i < S; ++i) {
      s_start[i] = o;
      o += seg_len ? seg_len[i] : M;
    }
    s_start[S] = o;
  }
  // this wavefront's slice of the weights: row 120 w + 30 (lane / 16) + s, columns lane % 16 and 16 + lane % 16
  float breg[KF_KW / 4][2];
  {
    const float* wr = W + static_cast<int64_t>(KF_KW * w + (KF_KW / 4) * sub) * C + col;
#pragma unroll
    for (int st = 0; st < KF_KW / 4; ++st) {
      breg[st][0] = wr[st * C];
      breg[st][1] = wr[st * C + 16];
    }
  }
  const int eq = threadIdx.x >> 4, ec = (threadIdx.x & 15) * 2;      // epilogue: tile row, first of two columns
  const float2 bias2 = bias ? *reinterpret_cast<const float2*>(bias + ec) : make_float2(0.f, 0.f);
  int my_seg = -1;
  double rs[2] = {0.0, 0.0}, rss[2] = {0.0, 0.0};     // fp64: exact sums of fp32 values and squares, i.e. independent of the processing order
  const int gs = stats ? C / groups : 1;
  double* rep = stats ? stats + static_cast<int64_t>(blockIdx.x % GN_REPLICAS) * S * groups * 2 : nullptr;
  auto flush_thread = [&]() {
    if (my_seg < 0 || !rep) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double* d = rep + (static_cast<int64_t>(my_seg) * groups + (ec + j) / gs) * 2;
      atomicAdd(d, rs[j]);
      atomicAdd(d + 1, rss[j]);
    }
  };
  __syncthreads();

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int64_t per = (M + 7) / 8;
  const int64_t band_begin = xcd * per, band_end = min(M, band_begin + per);
  for (int64_t t0 = band_begin + static_cast<int64_t>(slot) * KF_Q; t0 < band_end; t0 += static_cast<int64_t>(nslots) * KF_Q) {
    // ---- 1. aggregation: 4 queries per wavefront, each D parked as a row of the tile
    for (int i = 0; i < KF_QW; ++i) {
      const int ql = w * KF_QW + i;
      const int64_t t = t0 + ql;
      if (t >= band_end) {                                  // wave-uniform
        if (lane == 0) s_m[ql] = -1;
        continue;
      }
      const int64_t m = order ? order[t] : t;
      const float qx = q_pts[3 * m], qy = q_pts[3 * m + 1], qz = q_pts[3 * m + 2];
      int n = 0, cnt = 0;
      for (int h0 = 0; h0 < H; h0 += 64) {
        const int h = h0 + lane;
        int64_t j = Ns;
        if (h < H) j = static_cast<int64_t>(idx[m * H + h]);
        const bool ok = j >= 0 && j < Ns;
        const uint64_t mk = wave_ballot(ok);
        bool positive = false;
        if (ok) {
          const int sl = n + mbcnt_lt(mk);
          s_rel[w][sl] = make_float4(s_pts[3 * j] - qx, s_pts[3 * j + 1] - qy, s_pts[3 * j + 2] - qz, __uint_as_float(static_cast<uint32_t>(j)));
          positive = s_pos[j] != 0;
        }
        n += __popcll(mk);
        cnt += __popcll(wave_ballot(positive));
      }
      wave_lds_sync();
      floatx4 acc[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = floatx4{0.f, 0.f, 0.f, 0.f};
      const int steps = (n + 3) >> 2;
      float4 p[D];
      float2 f[D];
      auto fetch = [&](int st, float4& pp, float2& ff) {
        const int h = 4 * st + sub;
        pp = s_rel[w][h < n ? h : n - 1];
        ff = *reinterpret_cast<const float2*>(s_feats + static_cast<int64_t>(__float_as_uint(pp.w)) * C + V * col);
      };
      if (n > 0) {
        auto compute = [&](int st, const float4& pp, const float2& ff) {
          const float ex = pp.x - kx, ey = pp.y - ky, ez = pp.z - kz;
          float wv = fmaxf(fmaf(-__builtin_amdgcn_sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex))), inv_sigma, 1.f), 0.f);
          wv = (real_k && 4 * st + sub < n) ? wv : 0.f;
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, ff.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, ff.y, acc[1], 0, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(d, p[d], f[d]);
        int s0 = 0;
        for (; s0 + D <= steps; s0 += D) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            compute(s0 + d, p[d], f[d]);
            fetch(s0 + d + D, p[d], f[d]);
          }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
          if (s0 + d < steps) compute(s0 + d, p[d], f[d]);
      }
      float* drow = s_D + ql * KF_LD + V * col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * sub + r;
        if (k < KP_K) *reinterpret_cast<float2*>(drow + k * C) = make_float2(acc[0][r], acc[1][r]);
      }
      if (lane == 0) {
        int sg = 0;
        while (sg + 1 < S && m >= s_start[sg + 1]) ++sg;
        s_m[ql] = m;
        s_nn[ql] = static_cast<float>(cnt > 1 ? cnt : 1);
        s_seg[ql] = sg;
      }
      wave_lds_sync();
    }
    __syncthreads();
    // ---- 2. contraction of the tile with this wavefront's 120 weight rows
    {
      const float* arow = s_D + col * KF_LD + KF_KW * w + (KF_KW / 4) * sub;
      floatx4 c[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) c[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < KF_KW / 4; st += 2) {
        const float2 a2 = *reinterpret_cast<const float2*>(arow + st);
        c[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, breg[st][0], c[0][0], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, breg[st][1], c[1][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, breg[st + 1][0], c[0][1], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, breg[st + 1][1], c[1][1], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_P[w][4 * sub + r][col] = c[0][0][r] + c[0][1][r];
        s_P[w][4 * sub + r][16 + col] = c[1][0][r] + c[1][1][r];
      }
    }
    __syncthreads();
    // ---- 3. fold the four partial tiles, count division + bias, store, GroupNorm sums
    {
      const int64_t m = s_m[eq];
      if (m >= 0) {
        float2 v = make_float2(0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < KP_WAVES; ++ww) {
          const float2 pv = *reinterpret_cast<const float2*>(&s_P[ww][eq][ec]);
          v.x += pv.x;
          v.y += pv.y;
        }
        const float nnv = s_nn[eq];
        v.x = v.x / nnv + bias2.x;
        v.y = v.y / nnv + bias2.y;
        *reinterpret_cast<float2*>(out + m * C + ec) = v;
        if (rep) {
          const int sg = s_seg[eq];
          if (sg != my_seg) {
            flush_thread();
            my_seg = sg;
            rs[0] = rs[1] = rss[0] = rss[1] = 0.0;
          }
          const double vx = v.x, vy = v.y;
          rs[0] += vx;
          rs[1] += vy;
          rss[0] += vx * vx;
          rss[1] += vy * vy;
        }
      }
    }
    // (the next tile's D rows are written after this point by wavefronts that have all passed the barrier above, i.e. finished
    //  reading the tile; its partial tiles are written after the next tile barrier, i.e. after everybody's fold)
  }
  if (rep) {
    // one fp64 atomic per (wavefront, column, statistic) when the wavefront's threads agree on the segment
    int mx = my_seg;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) mx = max(mx, __shfl_xor(mx, d));
    const bool agree = wave_ballot(my_seg >= 0 && my_seg != mx) == 0ull;
    if (agree && mx >= 0) {
      double ds[2], dss[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ds[j] = my_seg >= 0 ? rs[j] : 0.0;
        dss[j] = my_seg >= 0 ? rss[j] : 0.0;
        ds[j] += __shfl_xor(ds[j], 16);
        dss[j] += __shfl_xor(dss[j], 16);
        ds[j] += __shfl_xor(ds[j], 32);
        dss[j] += __shfl_xor(dss[j], 32);
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          double* d = rep + (static_cast<int64_t>(mx) * groups + (ec + j) / gs) * 2;
          atomicAdd(d, ds[j]);
          atomicAdd(d + 1, dss[j]);
        }
      }
    } else {
      flush_thread();
    }
  }
}

// encoder1_1: scalar input feature per point; out[m][o] = (sum_k (sum_h w[k][h] f[h]) W[k][o]) / count + bias[o]
// One wavefront per query, lanes = neighbours.  A query is three DEPENDENT memory round trips (order -> point + index row ->
// neighbour coordinates) and ~250 instructions, so the kernel is latency-bound: the header of query t+2 and the neighbour
// gathers of query t+1 are in flight while query t is computed (clamped, unconditional loads: one basic block per trip), and
// the register budget (MAXO output channels per lane) is sized by the launch: 91 VGPRs / 5 wavefronts per SIMD for C_out <= 64
// instead of 124 / 4.  With that the kernel sits at its VALU bound (15 influences x ~11 operations per neighbour, one lane each).
template <typename IdxT, int MAXO>
__global__ __launch_bounds__(KP_WAVES * 64) void k_kpconv_cin1(const float* __restrict__ s_feats, const float* __restrict__ q_pts,
                                                               const float* __restrict__ s_pts, const IdxT* __restrict__ idx, int64_t M,
                                                               int64_t Ns, int H, KPoints kp, float sigma, const float* __restrict__ W,
                                                               const float* __restrict__ bias, int Cout, float* __restrict__ out,
                                                               const int32_t* __restrict__ order) {
  __shared__ float s_a[KP_WAVES][16];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  // this lane's output channel(s): the (15, Cout) weights and the bias stay in registers for every query of the wavefront
  float wreg[MAXO][KP_K], breg[MAXO];
#pragma unroll
  for (int q = 0; q < MAXO; ++q) {
    const int o = lane + 64 * q;
    breg[q] = (bias && o < Cout) ? bias[o] : 0.f;
#pragma unroll
    for (int k = 0; k < KP_K; ++k) wreg[q][k] = o < Cout ? W[k * Cout + o] : 0.f;
  }
  const float inv_sigma = 1.f / sigma;
  const XcdBand band = xcd_band(M, w, KP_WAVES);         // a contiguous eighth of the (ordered) queries per XCD
  const int64_t stride = band.stride, M_end = band.end;
  int64_t t = band.begin;
  if (t >= M_end) return;                               // wavefront-level synchronisation only below
  struct Row {
    int64_t m, j;
    float   qx, qy, qz;
  };
  struct Nb {
    float f, x, y, z;
  };
  auto load_row = [&](int64_t tt, Row& r) {
    const int64_t tc = tt < M_end ? tt : M_end - 1;     // beyond the end of the band: a valid row whose values are never used
    r.m = order ? order[tc] : tc;
    r.qx = q_pts[3 * r.m];
    r.qy = q_pts[3 * r.m + 1];
    r.qz = q_pts[3 * r.m + 2];
    r.j = lane < H ? static_cast<int64_t>(idx[r.m * H + lane]) : Ns;
  };
  auto gather = [&](const Row& r, Nb& nb) {
    const int64_t jc = (r.j >= 0 && r.j < Ns) ? r.j : 0;
    nb.f = nb.x = nb.y = nb.z = 0.f;
    if (Ns > 0) {                                       // uniform
      nb.f = s_feats[jc];
      nb.x = s_pts[3 * jc];
      nb.y = s_pts[3 * jc + 1];
      nb.z = s_pts[3 * jc + 2];
    }
  };
  Row r0, r1;
  Nb n0, n1;
  load_row(t, r0);
  gather(r0, n0);
  load_row(t + stride, r1);
  for (; t < M_end; t += stride) {
    Row r2;
    gather(r1, n1);
    load_row(t + 2 * stride, r2);
    const int64_t m = r0.m;
    const float qx = r0.qx, qy = r0.qy, qz = r0.qz;
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 0.f;
    auto add = [&](float f, float px, float py, float pz) {
      a[15] += f > 0.f ? 1.f : 0.f;              // neighbour count rides along as the 16th value (exact: small integers)
      const float dx = px - qx, dy = py - qy, dz = pz - qz;
#pragma unroll
      for (int k = 0; k < KP_K; ++k) {
        const float ex = dx - kp.p[k][0], ey = dy - kp.p[k][1], ez = dz - kp.p[k][2];
        a[k] = fmaf(fmaxf(fmaf(-__builtin_amdgcn_sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex))), inv_sigma, 1.f), 0.f), f, a[k]);
      }
    };
    if (r0.j >= 0 && r0.j < Ns) add(n0.f, n0.x, n0.y, n0.z);
    for (int h = lane + 64; h < H; h += 64) {      // neighbour columns beyond 64: not prefetched
      const int64_t j = static_cast<int64_t>(idx[m * H + h]);
      if (j >= 0 && j < Ns) add(s_feats[j], s_pts[3 * j], s_pts[3 * j + 1], s_pts[3 * j + 2]);
    }
    // reduce-scatter of the 16 per-lane partials over the wavefront: each exchange halves the values a lane is responsible
    // for (8 + 4 + 2 + 1 exchanges, then two plain ones) instead of 16 full six-step reductions.  Lane l ends with the total
    // of value k(l) = bits 5..2 of l (bit 5 = 8, bit 4 = 4, bit 3 = 2, bit 2 = 1).
    {
      const bool u5 = lane & 32, u4 = lane & 16, u3 = lane & 8, u2 = lane & 4;
      float b8[8], b4[4], b2[2], b1;
#pragma unroll
      for (int i = 0; i < 8; ++i) b8[i] = (u5 ? a[i + 8] : a[i]) + __shfl_xor(u5 ? a[i] : a[i + 8], 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) b4[i] = (u4 ? b8[i + 4] : b8[i]) + __shfl_xor(u4 ? b8[i] : b8[i + 4], 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) b2[i] = (u3 ? b4[i + 2] : b4[i]) + __shfl_xor(u3 ? b4[i] : b4[i + 2], 8);
      b1 = (u2 ? b2[1] : b2[0]) + __shfl_xor(u2 ? b2[0] : b2[1], 4);
      b1 += __shfl_xor(b1, 2);
      b1 += __shfl_xor(b1, 1);
      if ((lane & 3) == 0) s_a[w][(u5 ? 8 : 0) + (u4 ? 4 : 0) + (u3 ? 2 : 0) + (u2 ? 1 : 0)] = b1;
    }
    wave_lds_sync();
    const float cntf = s_a[w][15];
    const float div = cntf > 1.f ? cntf : 1.f;
    float av[KP_K];
#pragma unroll
    for (int k = 0; k < KP_K; ++k) av[k] = s_a[w][k];
#pragma unroll
    for (int q = 0; q < MAXO; ++q) {
      const int o = lane + 64 * q;
      if (o < Cout) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KP_K; ++k) v = fmaf(av[k], wreg[q][k], v);
        v = v / div;
        if (bias) v += breg[q];
        out[m * Cout + o] = v;
      }
    }
    wave_lds_sync();
    r0 = r1;
    n0 = n1;
    r1 = r2;
  }
}

// maxpool over neighbours with a zero shadow row (kpconv/functional.py:54-67)
template <typename IdxT>
__global__ __launch_bounds__(256) void k_maxpool(const float* __restrict__ x, const IdxT* __restrict__ idx, int64_t M, int64_t Ns, int H, int C,
                                                 float* __restrict__ out, const int32_t* __restrict__ order) {
  __shared__ int32_t s_idx[4][KP_HMAX];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const XcdBand band = xcd_band(M, w, 4);
  for (int64_t t = band.begin; t < band.end; t += band.stride) {
    const int64_t m = order ? order[t] : t;
    int n = 0;
    bool any_shadow = false;
    for (int h0 = 0; h0 < H; h0 += 64) {
      const int h = h0 + lane;
      int64_t j = Ns;
      if (h < H) j = static_cast<int64_t>(idx[m * H + h]);
      const bool ok = j >= 0 && j < Ns;
      const uint64_t mk = wave_ballot(ok);
      if (ok) s_idx[w][n + mbcnt_lt(mk)] = static_cast<int32_t>(j);
      any_shadow |= (wave_ballot(h < H && !ok) != 0ull);
      n += __popcll(mk);
    }
    wave_lds_sync();
    const float init = any_shadow ? 0.f : -INFINITY;
    if ((C & 255) == 0) {
      // a lane owns 4 consecutive channels per 256-channel slab: one 16-B load per gathered row instead of four 4-B ones
      for (int c = lane * 4; c < C; c += 256) {
        float4 v = make_float4(init, init, init, init);
        for (int h0 = 0; h0 < n; h0 += 8) {        // 8 independent row gathers in flight
          float4 f[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            f[u] = *reinterpret_cast<const float4*>(x + static_cast<int64_t>(s_idx[w][h0 + u < n ? h0 + u : n - 1]) * C + c);
#pragma unroll
          for (int u = 0; u < 8; ++u) {             // duplicates of the last row do not change a max
            v.x = fmaxf(v.x, f[u].x);
            v.y = fmaxf(v.y, f[u].y);
            v.z = fmaxf(v.z, f[u].z);
            v.w = fmaxf(v.w, f[u].w);
          }
        }
        *reinterpret_cast<float4*>(out + m * C + c) = v;
      }
    } else if ((C & 127) == 0) {
      for (int c = lane * 2; c < C; c += 128) {
        float2 v = make_float2(init, init);
        for (int h0 = 0; h0 < n; h0 += 8) {
          float2 f[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            f[u] = *reinterpret_cast<const float2*>(x + static_cast<int64_t>(s_idx[w][h0 + u < n ? h0 + u : n - 1]) * C + c);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v.x = fmaxf(v.x, f[u].x);
            v.y = fmaxf(v.y, f[u].y);
          }
        }
        *reinterpret_cast<float2*>(out + m * C + c) = v;
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        float v = init;
        for (int h0 = 0; h0 < n; h0 += 8) {
          float f[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) f[u] = x[static_cast<int64_t>(s_idx[w][h0 + u < n ? h0 + u : n - 1]) * C + c];
#pragma unroll
          for (int u = 0; u < 8; ++u) v = fmaxf(v, f[u]);
        }
        out[m * C + c] = v;
      }
    }
    wave_lds_sync();
  }
}

// pos[n] = (sum_c x[n][c] > 0): the per-support flag behind KPConv's neighbour count (kpconv.py:113-114)
__global__ __launch_bounds__(256) void k_row_pos(const float* __restrict__ x, int64_t N, int C, uint8_t* __restrict__ pos) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * 4 + w; n < N; n += static_cast<int64_t>(gridDim.x) * 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += x[n * C + c];
    s = wave_sum(s);
    if (lane == 0) pos[n] = s > 0.f ? 1 : 0;
  }
}

static KPoints load_kp(const float* kp_host) {
  KPoints k;
  for (int i = 0; i < KP_K; ++i)
    for (int d = 0; d < 3; ++d) k.p[i][d] = kp_host[3 * i + d];
  return k;
}

static int grid_for(int64_t rows, int per_block) { return static_cast<int>(std::min<int64_t>((rows + per_block - 1) / per_block, 256 * 16)); }
// gather kernels (xcd_band): a multiple of 8 workgroups, enough for every XCD's eighth of the rows
static int grid_for_xcd(int64_t rows, int per_block) {
  const int64_t per_xcd = ((rows + 7) / 8 + per_block - 1) / per_block;
  return 8 * static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(per_xcd, 256 * 2)));
}

static int g_agg_force_off64 = 0;          // tests: run the 64-bit-offset form of the aggregation (tensors of 2^30 elements and more)

template <typename IdxT>
static int launch_aggregate(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts, const IdxT* idx, int64_t M,
                            int64_t Ns, int H, int C, const KPoints& kp, float sigma, float* A, float* nn, const int32_t* order, int vf,
                            hipStream_t st) {
  dim3 grid(grid_for_xcd(M, KP_WAVES)), block(KP_WAVES * 64);
  const bool off32 = Ns * C < (int64_t(1) << 30) && M * H < (int64_t(1) << 30) && !g_agg_force_off64;   // feature rows / index rows addressable with 32-bit byte offsets
#define LCR_AGG(CC)                                                                                                                          \
  if (off32) LCR_LAUNCH_TIMED((k_kpconv_aggregate_vec<IdxT, CC, true>), grid, block, 0, st, s_feats, s_pos, q_pts, s_pts, idx, M, Ns, H, kp, sigma, A, nn, order, vf); \
  else LCR_LAUNCH_TIMED((k_kpconv_aggregate_vec<IdxT, CC, false>), grid, block, 0, st, s_feats, s_pos, q_pts, s_pts, idx, M, Ns, H, kp, sigma, A, nn, order, vf);      \
  break
  switch (C) {
    case 32: LCR_AGG(32);
    case 64: LCR_AGG(64);
    case 128: LCR_AGG(128);
    case 256: LCR_AGG(256);
    default: set_error("lcr_kpconv_aggregate: C must be 32, 64, 128 or 256 (got %d)", C); return LCR_EARG;
  }
#undef LCR_AGG
  return check_launch("lcr_kpconv_aggregate");
}

}  // namespace lcr

using namespace lcr;

extern "C" void lcr_kpconv_debug_off64(int on) { g_agg_force_off64 = on; }

extern "C" int lcr_kpconv_aggregate(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts, const void* idx,
                                    int idx_is_64, int64_t M, int64_t Ns, int H, int C, const float* kernel_points_host, float sigma,
                                    float* A, float* nn, const int32_t* order, void* stream) {
  return lcr_kpconv_aggregate_ex(s_feats, s_pos, q_pts, s_pts, idx, idx_is_64, M, Ns, H, C, kernel_points_host, sigma, A, nn, order, 0, stream);
}

extern "C" int lcr_kpconv_aggregate_ex(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts, const void* idx,
                                       int idx_is_64, int64_t M, int64_t Ns, int H, int C, const float* kernel_points_host, float sigma,
                                       float* A, float* nn, const int32_t* order, int flags, void* stream) {
  if (!s_feats || !s_pos || !q_pts || !s_pts || !idx || !kernel_points_host || !A || !nn || M < 0 || Ns < 0 || H < 1 || H > KP_HMAX ||
      !(sigma > 0.f)) {
    set_error("lcr_kpconv_aggregate: bad argument (H must be in [1,%d])", KP_HMAX);
    return LCR_EARG;
  }
  if (M == 0) return LCR_OK;
  const KPoints kp = load_kp(kernel_points_host);
  hipStream_t st = static_cast<hipStream_t>(stream);
  KernelTimerScope timed(KT_AGGREGATE, st, M, Ns, H, C, idx_is_64 ? 8 : 4);
  const int vf = (flags & LCR_KP_VALID_FIRST) ? 1 : 0;
  return idx_is_64 ? launch_aggregate(s_feats, s_pos, q_pts, s_pts, static_cast<const int64_t*>(idx), M, Ns, H, C, kp, sigma, A, nn, order, vf, st)
                   : launch_aggregate(s_feats, s_pos, q_pts, s_pts, static_cast<const int32_t*>(idx), M, Ns, H, C, kp, sigma, A, nn, order, vf, st);
}

extern "C" int lcr_kpconv_fused(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts, const void* idx,
                                int idx_is_64, int64_t M, int64_t Ns, int H, int C, const float* kernel_points_host, float sigma,
                                const float* W, const float* bias, float* out, const int64_t* seg_len, int S, int groups, double* stats,
                                const int32_t* order, void* stream) {
  if (!s_feats || !s_pos || !q_pts || !s_pts || !idx || !kernel_points_host || !W || !out || M < 0 || Ns < 0 || H < 1 || H > KP_HMAX ||
      !(sigma > 0.f) || C != KF_C) {
    set_error("lcr_kpconv_fused: bad argument (C must be %d, H in [1,%d])", KF_C, KP_HMAX);
    return LCR_EARG;
  }
  if (stats && (!seg_len || S < 1 || S > KF_MAX_SEG || groups < 1 || C % groups != 0)) {
    set_error("lcr_kpconv_fused: statistics need seg_len, 1 <= S <= %d and groups dividing C", KF_MAX_SEG);
    return LCR_EARG;
  }
  if (M == 0) return LCR_OK;
  const KPoints kp = load_kp(kernel_points_host);
  hipStream_t st = static_cast<hipStream_t>(stream);
  KernelTimerScope timed(KT_KPCONV_FUSED, st, M, Ns, H, C, idx_is_64 ? 8 : 4);
  static const int n_cu = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  // residency-sized grid (3 workgroups per CU, a multiple of the 8 XCDs), never more workgroups than tiles per band
  const int64_t tiles_per_band = ((M + 7) / 8 + KF_Q - 1) / KF_Q;
  const int grid = 8 * static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(tiles_per_band, (3 * n_cu + 7) / 8)));
  const int Sx = stats ? S : 1;
  if (idx_is_64)
    hipLaunchKernelGGL((k_kpconv_fused32<int64_t>), dim3(grid), dim3(KP_WAVES * 64), 0, st, s_feats, s_pos, q_pts, s_pts,
                       static_cast<const int64_t*>(idx), M, Ns, H, kp, sigma, W, bias, out, stats ? seg_len : nullptr, Sx, groups, stats, order);
  else
    hipLaunchKernelGGL((k_kpconv_fused32<int32_t>), dim3(grid), dim3(KP_WAVES * 64), 0, st, s_feats, s_pos, q_pts, s_pts,
                       static_cast<const int32_t*>(idx), M, Ns, H, kp, sigma, W, bias, out, stats ? seg_len : nullptr, Sx, groups, stats, order);
  return check_launch("lcr_kpconv_fused");
}

extern "C" int lcr_kpconv_cin1(const float* s_feats, const float* q_pts, const float* s_pts, const void* idx, int idx_is_64, int64_t M,
                               int64_t Ns, int H, const float* kernel_points_host, float sigma, const float* W, const float* bias, int Cout,
                               float* out, const int32_t* order, void* stream) {
  if (!s_feats || !q_pts || !s_pts || !idx || !kernel_points_host || !W || !out || M < 0 || H < 1 || Cout < 1 || Cout > 256 || !(sigma > 0.f)) {
    set_error("lcr_kpconv_cin1: bad argument");
    return LCR_EARG;
  }
  if (M == 0) return LCR_OK;
  const KPoints kp = load_kp(kernel_points_host);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(grid_for_xcd(M, KP_WAVES)), block(KP_WAVES * 64);
#define LCR_CIN1(IDX, MAXO) \
  hipLaunchKernelGGL((k_kpconv_cin1<IDX, MAXO>), grid, block, 0, st, s_feats, q_pts, s_pts, static_cast<const IDX*>(idx), M, Ns, H, kp, sigma, W, bias, Cout, out, order)
  if (idx_is_64) {
    if (Cout <= 64) LCR_CIN1(int64_t, 1);
    else LCR_CIN1(int64_t, 4);
  } else {
    if (Cout <= 64) LCR_CIN1(int32_t, 1);
    else LCR_CIN1(int32_t, 4);
  }
#undef LCR_CIN1
  return check_launch("lcr_kpconv_cin1");
}

extern "C" int lcr_maxpool(const float* x, const void* idx, int idx_is_64, int64_t M, int64_t Ns, int H, int C, float* out,
                           const int32_t* order, void* stream) {
  if (!x || !idx || !out || M < 0 || H < 1 || H > KP_HMAX || C < 1) {
    set_error("lcr_maxpool: bad argument");
    return LCR_EARG;
  }
  if (M == 0) return LCR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(grid_for_xcd(M, 4)), block(256);
  if (idx_is_64) hipLaunchKernelGGL((k_maxpool<int64_t>), grid, block, 0, st, x, static_cast<const int64_t*>(idx), M, Ns, H, C, out, order);
  else hipLaunchKernelGGL((k_maxpool<int32_t>), grid, block, 0, st, x, static_cast<const int32_t*>(idx), M, Ns, H, C, out, order);
  return check_launch("lcr_maxpool");
}

extern "C" int lcr_row_positive(const float* x, int64_t N, int C, uint8_t* pos, void* stream) {
  if (!x || !pos || N < 0 || C < 1) {
    set_error("lcr_row_positive: bad argument");
    return LCR_EARG;
  }
  if (N == 0) return LCR_OK;
  hipLaunchKernelGGL(k_row_pos, dim3(grid_for(N, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, N, C, pos);
  return check_launch("lcr_row_positive");
}
