// gemm_f32.hip — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TFLOP/s peak)
// with the epilogues the KPConv encoder needs fused in:  C = (A·B) [/ rowdiv[m]] [+ bias[n]],  plus per-(segment, group)
// sum / sum-of-squares accumulation for the GroupNorm that always follows (modules/kpconv/modules.py:33-50, 53-84).
//
// Used for: nn.Linear of UnaryBlock (B = weight^T, TB=1), the kernel-point contraction of KPConv
// (kpconv.py:108-110: (M, 15*C) x (15*C, Cout), TB=0), NetVLAD's assignment / aggregation GEMMs (TA=1 for x^T·a).
// fp32 in, fp32 accumulate: bf16 would break the 1e-4 descriptor tolerance.
//
// Tiling: 256 threads = 4 wavefronts arranged WM x WN; block tile BM x BN x 32; every wavefront owns 32 rows x (BN/WN)
// columns = NT accumulators of 16 registers.  Operands are staged through LDS K-major (As[k][m], Bs[k][n]) so an MFMA
// operand fetch is one conflict-free ds_read_b32 per lane.  Because the fp32 MFMA is slow (64 cycles per 4 KFLOP) the
// only thing that matters is never exposing a latency: global loads are branch-free 16-B vectors issued one K-step ahead
// into registers (double-buffered LDS, one barrier per K-step), LDS fragments are prefetched one MFMA group ahead, and the
// tile shape is chosen so that >= ~400 workgroups exist (two per CU) even for the short-M stages.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace lcr {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GM_BK = 32;
constexpr int GM_T = 256;

// Optional batching over blockIdx.z: per-entry element offsets into A/B/C and per-entry K (NetVLAD's per-scan x^T·a products).
constexpr int GM_MAX_BATCH = 64;
struct GemmBatch {
  int     count;                 // 0 = plain GEMM
  int     strided;               // 1: entry z uses offsets z * {a,b,c}_off[0] and K = k[0] (uniform batch)
  int     k[GM_MAX_BATCH];
  int64_t a_off[GM_MAX_BATCH], b_off[GM_MAX_BATCH], c_off[GM_MAX_BATCH];
};

struct GemmEpilogue {
  const float*   bias;      // [N] or null
  const float*   rowdiv;    // [M] or null: C[m][:] /= rowdiv[m] (before the bias), KPConv neighbour-count normalisation
  const int64_t* seg_len;   // [S] rows per GroupNorm segment (device) or null
  int            S;
  int            groups;    // GroupNorm groups over N
  double*        stats;     // [GN_REPLICAS, S, groups, 2] (sum, sumsq), accumulated atomically; null = no statistics
};

// Normalise-on-load of the A operand: A holds the RAW output of the producing layer and the GroupNorm + LeakyReLU that the
// reference applies between the two layers (ResidualBlock: norm_conv + leaky_relu in front of unary2, modules.py:215-217) happens
// while the tile goes from registers to LDS — the normalised tensor is never written to memory (one stand-alone GroupNorm pass
// per residual block less).  Rows of A and rows of C share the segment table.
struct ANorm {
  const double* stats;    // [GN_REPLICAS, S, groups, 2] sums of the A producer (its GEMM epilogue)
  const float*  gamma;    // [K]
  const float*  beta;     // [K]
  int           groups;   // GroupNorm groups over K
  float         eps, slope;
};
constexpr int AN_KMAX = 256;   // K of the fused form (the light kernel: K <= 256)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// R x 32 tile of a row-major [rows_total x K] matrix -> S[k][r]   (transposing loader; src contiguous along k).
// Branch-free: out-of-range pieces read a valid clamped address and are zeroed by a select.
// RM: the LDS tile is kept ROW-major instead (S[r][k], LD = 36): the store is one 16-B write per piece instead of four
// transposing 4-B ones, and an MFMA operand fetch becomes one 16-B read per FOUR MFMAs (see the K pairing in k_gemm_f32).
template <int R, int LD, bool VEC, bool RM = false>
struct LoaderT {
  static constexpr int PIECES = R * 8 / GM_T;   // float4 pieces per thread
  float4 reg[PIECES];
  unsigned okmask;                              // bit j: piece j is inside K (zeroing is deferred to the LDS store, so that
                                                // nothing waits on the load between its issue and its use one K-step later)
  int kb;                                       // k0 of the loaded tile (for the normalising store)
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t rows_total, int K, int64_t r0, int k0) {
    kb = k0;
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int row = f >> 3, c4 = f & 7;
      int64_t gr = r0 + row;
      gr = gr < rows_total ? gr : rows_total - 1;          // duplicate rows only feed outputs that are never stored
      const int gk = k0 + c4 * 4;
      if (VEC) {
        const bool ok = gk < K;                               // K % 4 == 0: a piece is entirely in or out
        reg[j] = ld4(src + gr * K + (ok ? gk : 0));
        okmask = j == 0 ? (ok ? 1u : 0u) : (okmask | (ok ? (1u << j) : 0u));
      } else {
        const float* p = src + gr * K;
        float4 v;
        v.x = gk + 0 < K ? p[gk + 0] : 0.f;
        v.y = gk + 1 < K ? p[gk + 1] : 0.f;
        v.z = gk + 2 < K ? p[gk + 2] : 0.f;
        v.w = gk + 3 < K ? p[gk + 3] : 0.f;
        reg[j] = v;
        okmask = ~0u;
      }
    }
  }
  __device__ __forceinline__ void store_piece(float* __restrict__ S, int j) const {
    const int f = threadIdx.x + GM_T * j;
    const int row = f >> 3, c4 = f & 7;
    const bool ok = (okmask >> j) & 1u;
    if (RM) {
      *reinterpret_cast<float4*>(&S[row * LD + c4 * 4]) = ok ? reg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    S[(c4 * 4 + 0) * LD + row] = ok ? reg[j].x : 0.f;
    S[(c4 * 4 + 1) * LD + row] = ok ? reg[j].y : 0.f;
    S[(c4 * 4 + 2) * LD + row] = ok ? reg[j].z : 0.f;
    S[(c4 * 4 + 3) * LD + row] = ok ? reg[j].w : 0.f;
  }
  __device__ __forceinline__ void store(float* __restrict__ S) const {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) store_piece(S, j);
  }
  // row-major tiles only: y = leaky(x * scale[k] + shift[k]) with the (scale, shift) of the row's segment slot (tab[slot][0|1][k];
  // rows < split are slot 0, the others slot 1), then the same store as above
  __device__ __forceinline__ void store_norm(float* __restrict__ S, const float* __restrict__ tab, int split, float slope) const {
    static_assert(RM, "normalise-on-load is built for the row-major A tile");
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int row = f >> 3, c4 = f & 7;
      const bool ok = (okmask >> j) & 1u;
      const float* t = tab + (row >= split ? 2 * AN_KMAX : 0) + (ok ? kb + c4 * 4 : 0);
      const float4 sc = *reinterpret_cast<const float4*>(t), sh = *reinterpret_cast<const float4*>(t + AN_KMAX);
      float4 v;
      v.x = fmaf(reg[j].x, sc.x, sh.x), v.y = fmaf(reg[j].y, sc.y, sh.y), v.z = fmaf(reg[j].z, sc.z, sh.z), v.w = fmaf(reg[j].w, sc.w, sh.w);
      v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
      *reinterpret_cast<float4*>(&S[row * LD + c4 * 4]) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// 32 x W tile of a row-major [K x cols_total] matrix -> S[k][c]   (straight loader; src contiguous along c)
template <int W, int LD, bool VEC>
struct LoaderN {
  static constexpr int PIECES = 8 * W / GM_T;
  float4 reg[PIECES];
  unsigned okmask;
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t cols_total, int K, int64_t c0, int k0) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int kr = f / (W / 4), c4 = f % (W / 4);
      const int gk = k0 + kr;
      const int64_t gc = c0 + c4 * 4;
      const bool kok = gk < K;
      const float* p = src + static_cast<int64_t>(kok ? gk : 0) * cols_total;
      if (VEC) {
        const bool ok = kok && gc < cols_total;              // cols_total % 4 == 0
        reg[j] = ld4(p + (gc < cols_total ? gc : 0));
        okmask = j == 0 ? (ok ? 1u : 0u) : (okmask | (ok ? (1u << j) : 0u));
      } else {
        float4 v;
        v.x = (kok && gc + 0 < cols_total) ? p[gc + 0] : 0.f;
        v.y = (kok && gc + 1 < cols_total) ? p[gc + 1] : 0.f;
        v.z = (kok && gc + 2 < cols_total) ? p[gc + 2] : 0.f;
        v.w = (kok && gc + 3 < cols_total) ? p[gc + 3] : 0.f;
        reg[j] = v;
        okmask = ~0u;
      }
    }
  }
  __device__ __forceinline__ void store_piece(float* __restrict__ S, int j) const {
    const int f = threadIdx.x + GM_T * j;
    const int kr = f / (W / 4), c4 = f % (W / 4);
    const bool ok = (okmask >> j) & 1u;
    *reinterpret_cast<float4*>(&S[kr * LD + c4 * 4]) = ok ? reg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void store(float* __restrict__ S) const {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) store_piece(S, j);
  }
};

__device__ __forceinline__ int seg_of_row(const int64_t* __restrict__ seg_len, int S, int64_t row) {
  int s = 0;
  int64_t end = seg_len[0];
  while (s + 1 < S && row >= end) {
    ++s;
    end += seg_len[s];
  }
  return s;
}

// Epilogue shared by the tile-per-workgroup kernel and the stream-K kernel: C = acc [/ rowdiv] [+ bias], GroupNorm statistics.
template <int BM, int BN, int WM, int WN, int NT>
__device__ __forceinline__ void gemm_epilogue(floatx16 (&acc)[NT], float* __restrict__ C, int64_t M, int N, int64_t m0, int n0, int64_t m_tile,
                                              const GemmEpilogue& ep, const float (&bias_v)[NT], int blk_first, int64_t blk_seg_start,
                                              int64_t blk_seg_end, int wm, int wn, int lane) {
  const bool want_stats = ep.stats != nullptr;
  const int gs = want_stats ? N / ep.groups : 1;   // channels per group
  const int64_t wrow0 = m0 + wm * 32;

  // store (and keep the final values in the accumulators for the statistics pass)
  // Interior tiles (every row and column of the tile inside the matrix — all but the last row / column block; element offsets below 2^30):
  // one 32-bit offset per lane and value, a compile-time multiple of N apart, on the scalar base — 2-3 VALU instructions per stored value.
  // The general form below costs ~16 (64-bit row, bounds, 64-bit address) x 16 values x 4 wavefronts per tile: ~50 M of a step's 600 M VALU
  // instructions (PMC, profiles/r04_pmc_insts*.md).  Same values, same order.
  const bool interior = m0 + BM <= M && n0 + BN <= N && M * static_cast<int64_t>(N) < (int64_t(1) << 30);   // workgroup-uniform
  if (interior) {
    const unsigned row_l = static_cast<unsigned>(wrow0) + 4u * static_cast<unsigned>(lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const unsigned col = static_cast<unsigned>(n0 + wn * (32 * NT) + j * 32 + (lane & 31));
      const float bv = bias_v[j];
      const unsigned off0 = row_l * static_cast<unsigned>(N) + col;
      if (ep.rowdiv) {                                 // uniform: the KPConv contractions
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned dr = static_cast<unsigned>((r & 3) + 8 * (r >> 2));
          const float v = acc[j][r] / ep.rowdiv[row_l + dr] + bv;
          C[off0 + dr * static_cast<unsigned>(N)] = v;
          acc[j][r] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned dr = static_cast<unsigned>((r & 3) + 8 * (r >> 2));
          const float v = acc[j][r] + bv;
          C[off0 + dr * static_cast<unsigned>(N)] = v;
          acc[j][r] = v;
        }
      }
    }
  } else
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * (32 * NT) + j * 32 + (lane & 31);
    const float bv = bias_v[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float v = acc[j][r];
      if (row < M) {
        if (ep.rowdiv) v = v / ep.rowdiv[row];
        v += bv;
        if (col < N) C[row * N + col] = v;
      } else {
        v = 0.f;
      }
      acc[j][r] = v;
    }
  }

  // GroupNorm statistics.  Fast path (all BM rows of the workgroup in one segment — all but B-1 workgroups): every lane parks
  // the fp32 sums of its 16 values per column in LDS, one thread per column folds the 2*WM row slices in fp64, the gs lanes of
  // a group are folded with shuffles in the BN/64 wavefronts that hold the columns, and the workgroup issues one pair of
  // global fp64 atomics per group into the statistics replica (row block % GN_REPLICAS).  (A version with fp64 shuffles in
  // every wavefront + LDS fp64 atomics cost 8-9 us per unary GEMM.)  Slow path (a segment boundary inside the tile): per
  // wavefront, per segment present in its 32 rows, straight to global memory.
  if (want_stats) {
    __shared__ float s_part[2][2 * WM][BN];
    double* rep = ep.stats + static_cast<int64_t>(m_tile % GN_REPLICAS) * ep.S * ep.groups * 2;
    int64_t seg_start = blk_seg_start, seg_end = blk_seg_end;
    const int64_t blk_last_row = min(m0 + BM - 1, M - 1);
    const bool uniform = blk_last_row < seg_end;      // block-uniform
    if (uniform) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float s = 0.f, ss = 0.f;                      // rows beyond M hold zeros in the accumulators
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s += acc[j][r];
          ss = fmaf(acc[j][r], acc[j][r], ss);
        }
        const int cl = wn * (32 * NT) + j * 32 + (lane & 31);
        s_part[0][2 * wm + (lane >> 5)][cl] = s;
        s_part[1][2 * wm + (lane >> 5)][cl] = ss;
      }
      __syncthreads();
      if (threadIdx.x < BN) {
        const int cl = threadIdx.x;
        double ds = 0.0, dss = 0.0;
#pragma unroll
        for (int r = 0; r < 2 * WM; ++r) {
          ds += static_cast<double>(s_part[0][r][cl]);
          dss += static_cast<double>(s_part[1][r][cl]);
        }
        const int span = gs < 64 ? gs : 64;           // gs is a power of two; groups wider than 64 columns add per wavefront
        for (int d = 1; d < span; d <<= 1) {
          ds += __shfl_xor(ds, d);
          dss += __shfl_xor(dss, d);
        }
        const int col = n0 + cl;
        if ((cl & (span - 1)) == 0 && col < N) {
          double* d = rep + (static_cast<int64_t>(blk_first) * ep.groups + col / gs) * 2;
          atomicAdd(d, ds);
          atomicAdd(d + 1, dss);
        }
      }
    } else if (wrow0 < M) {
      const int64_t wlast = min(wrow0 + 31, M - 1);
      int sg = blk_first;
      while (sg + 1 < ep.S && wrow0 >= seg_end) {
        ++sg;
        seg_start = seg_end;
        seg_end += ep.seg_len[sg];
      }
      while (true) {   // wave-uniform loop over the segments that intersect [wrow0, wlast]
        const bool whole = seg_start <= wrow0 && wlast < seg_end;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = n0 + wn * (32 * NT) + j * 32 + (lane & 31);
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float v = (whole || (row >= seg_start && row < seg_end)) ? acc[j][r] : 0.f;
            s += v;
            ss = fmaf(v, v, ss);
          }
          double ds = s, dss = ss;
          // fold the two row-halves, then the lanes of one group (gs consecutive columns, capped at the 32-column tile)
          ds += __shfl_xor(ds, 32);
          dss += __shfl_xor(dss, 32);
          const int span = gs < 32 ? gs : 32;
          for (int d = 1; d < span; d <<= 1) {
            ds += __shfl_xor(ds, d);
            dss += __shfl_xor(dss, d);
          }
          if (lane < 32 && (lane & (span - 1)) == 0 && col < N) {
            double* d = rep + (static_cast<int64_t>(sg) * ep.groups + col / gs) * 2;
            atomicAdd(d, ds);
            atomicAdd(d + 1, dss);
          }
        }
        if (wlast < seg_end || sg + 1 >= ep.S) break;
        ++sg;
        seg_start = seg_end;
        seg_end += ep.seg_len[sg];
      }
    }
  }
}

// SHORT = the light form for K <= 128 (the unary Linears of the big stages): such a problem is all prologue — load, one to four
// K-steps, epilogue — so what hides the load latency is the number of workgroups a CU holds, not prefetch depth: one LDS
// buffer (17 KB), one register stage, shallow fragment prefetch, and a register budget that lets 6-8 workgroups share a CU
// instead of 4.
template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC, bool SHORT = false, bool ANORM = false>
__global__ __launch_bounds__(GM_T, SHORT ? 6 : 2) void k_gemm_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int64_t M, int N, int K, GemmEpilogue ep, GemmBatch batch, ANorm an) {
  static_assert(!ANORM || (SHORT && !TA && VEC && BN / (32 * WN) == 1), "normalise-on-load: light form, row-major A");
  static_assert(WM * WN == 4 && BM == 32 * WM, "one 32-row MFMA tile per wavefront along M");
  constexpr int k_begin = 0;
  if (batch.count) {
    const int z = blockIdx.z;
    if (batch.strided) {
      A += static_cast<int64_t>(z) * batch.a_off[0];
      B += static_cast<int64_t>(z) * batch.b_off[0];
      C += static_cast<int64_t>(z) * batch.c_off[0];
      K = batch.k[0];
    } else {
      A += batch.a_off[z];
      B += batch.b_off[z];
      C += batch.c_off[z];
      K = batch.k[z];
    }
  }
  constexpr int NT = BN / (32 * WN);
  constexpr int NBUF = SHORT ? 1 : 2;
  // K pairing.  A 32x32x2 MFMA consumes two K indices (lanes 0-31 one, lanes 32-63 the other); which two is free as long as A
  // and B agree.  With one accumulator per wavefront (NT == 1: every encoder GEMM) MFMA j of a 32-deep K-step takes (j, j + 16):
  // a lane then needs 16 CONSECUTIVE k of its row, so an operand whose global layout is k-contiguous (A [M,K]; B = nn.Linear
  // weight [N,K]) stays row-major in LDS — stored with one 16-B write per piece, fetched with four 16-B reads per K-step instead
  // of sixteen 4-B ones.  (Before: (2j, 2j+1) on K-major tiles, 48 LDS instructions per wavefront and K-step; now 12 for the
  // unary Linears, 24 for the KPConv contraction whose B [K,N] stays K-major.)  Multi-accumulator tiles keep the old scheme.
  constexpr bool NEWP = NT == 1;
  constexpr bool ARM = NEWP && !TA, BRM = NEWP && TB;
  constexpr int LDR = GM_BK + 4;
  constexpr int LDA = ARM ? LDR : (TA ? BM + 4 : BM + 1);
  constexpr int LDB = BRM ? LDR : (TB ? BN + 1 : BN + 4);
  __shared__ __attribute__((aligned(16))) float As[NBUF][ARM ? BM * LDR : GM_BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BRM ? BN * LDR : GM_BK * LDB];

  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const int wm = w / WN, wn = w % WN;
  // XCD-aware tile order.  Workgroups go round-robin to the 8 XCDs (linear id % 8), each with its own L2; the column tiles
  // of one row block all read the same A rows, so they are given ids that land on ONE XCD back to back: the A tile is then
  // fetched from HBM / Infinity Cache once instead of once per column tile (PMC: GEMM fetch traffic was ~2x algorithmic).
  const int ntn = (N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m_tile = static_cast<int64_t>(slot / ntn) * 8 + xcd;
  const int64_t m0 = m_tile * BM;
  const int n0 = (slot % ntn) * BN;
  if (m0 >= M) return;                     // padding of the row-block count to a multiple of 8

  // two register stages: the tile for K-step t+2 is requested while step t computes and step t+1 already sits in registers,
  // so a global load has two full MFMA phases (~2 x 1024 cycles) to land before it is needed for the LDS store
  LoaderT<BM, LDA, VEC, ARM> la_t[NBUF];
  LoaderN<BM, LDA, VEC> la_n[NBUF];
  LoaderT<BN, LDB, VEC, BRM> lb_t[NBUF];
  LoaderN<BN, LDB, VEC> lb_n[NBUF];

  auto gload = [&](int k0, int r) {
    if (TA) la_n[r].load(A, M, K, m0, k0);
    else la_t[r].load(A, M, K, m0, k0);
    if (TB) lb_t[r].load(B, N, K, n0, k0);
    else lb_n[r].load(B, N, K, n0, k0);
  };
  auto sstore = [&](int buf, int r) {
    if (TA) la_n[r].store(As[buf]);
    else la_t[r].store(As[buf]);
    if (TB) lb_t[r].store(Bs[buf]);
    else lb_n[r].store(Bs[buf]);
  };

  // The 128x32 tile (N = 32: the stage-1 KPConv contraction) chains every MFMA on one accumulator, where any issue bubble
  // costs a full 64-cycle latency: it alternates between two accumulator sets (even / odd K pairs), +10 % measured.  The
  // 64x64 tile measured slower with the extra registers, so it keeps one.
  constexpr int NACC = (NT == 1 && BM == 128) ? 2 : 1;
  floatx16 acc[NT], acc2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[j][r] = 0.f;
      acc2[j][r] = 0.f;
    }

  const int nk = (K + GM_BK - 1) / GM_BK;                 // K may have been replaced by a per-entry value above
  // Both prologue tiles are requested before anything waits (for nk == 1 the second request re-reads tile 0: its pieces are
  // out of K, so their addresses fall back to column 0 — cache hits, never stored).  The epilogue's per-column bias and the
  // GroupNorm segment of this row block (a chain of dependent scalar loads) are fetched here too, under the same latency.
  gload(k_begin, 0);
  if (!SHORT) gload(k_begin + GM_BK, NBUF - 1);
  const bool want_stats = ep.stats != nullptr;
  float bias_v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + (w % WN) * (32 * NT) + j * 32 + (lane & 31);
    bias_v[j] = (ep.bias && col < N) ? ep.bias[col] : 0.f;
  }
  float an_g[2] = {0.f, 0.f}, an_b[2] = {0.f, 0.f};           // normalise-on-load: this thread's (gamma, beta) of table entries
  if constexpr (ANORM) {                                       // e = tid and tid + 256 (2 K <= 512), requested before the segment scan
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = threadIdx.x + u * GM_T;
      if (e < 2 * K) {
        const int k = e >= K ? e - K : e;
        an_g[u] = an.gamma[k];
        an_b[u] = an.beta[k];
      }
    }
  }
  int blk_first = 0;
  int64_t blk_seg_start = 0, blk_seg_end = 0;
  if (want_stats || ANORM) {
    blk_seg_end = ep.seg_len[0];
    while (blk_first + 1 < ep.S && m0 >= blk_seg_end) {
      ++blk_first;
      blk_seg_start = blk_seg_end;
      blk_seg_end += ep.seg_len[blk_first];
    }
  }
  __shared__ __attribute__((aligned(16))) float s_an[ANORM ? 4 * AN_KMAX : 4];   // [slot][scale | shift][k]
  int an_split = BM;
  if constexpr (ANORM) {
    // (scale, shift) per channel for the one or two segments this row block touches (the host guarantees that no segment is
    // shorter than the block), finalised in fp64 from the producer's statistics replicas exactly as lcr_groupnorm_apply does
    an_split = static_cast<int>(min(static_cast<int64_t>(BM), blk_seg_end - m0));
    const int gs = K / an.groups;
    __shared__ float2 s_mr[2 * 64];                        // (mean, rstd) of [slot][group]
    for (int e = threadIdx.x; e < 2 * an.groups; e += GM_T) {
      const int slot = e >= an.groups ? 1 : 0, g = e - slot * an.groups;
      const int sg = min(blk_first + slot, ep.S - 1);
      const double cnt = static_cast<double>(ep.seg_len[sg]) * gs;
      double sx = 0.0, sxx = 0.0;
      for (int rep = 0; rep < GN_REPLICAS; ++rep) {
        const int64_t o = ((static_cast<int64_t>(rep) * ep.S + sg) * an.groups + g) * 2;
        sx += an.stats[o];
        sxx += an.stats[o + 1];
      }
      const double mean = sx / cnt;
      const double var = fmax(sxx / cnt - mean * mean, 0.0);
      s_mr[e] = make_float2(static_cast<float>(mean), static_cast<float>(1.0 / sqrt(var + static_cast<double>(an.eps))));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = threadIdx.x + u * GM_T;
      if (e < 2 * K) {
        const int slot = e >= K ? 1 : 0, k = e - slot * K;
        const float2 mr = s_mr[slot * an.groups + k / gs];
        const float sc = mr.y * an_g[u];
        s_an[slot * 2 * AN_KMAX + k] = sc;
        s_an[slot * 2 * AN_KMAX + AN_KMAX + k] = fmaf(-mr.x, sc, an_b[u]);
      }
    }
    __syncthreads();
  }
  auto sstore_a0 = [&]() {                                 // the light form's (only) A store
    if constexpr (ANORM) la_t[0].store_norm(As[0], s_an, an_split, an.slope);
  };
  if constexpr (ANORM) {
    sstore_a0();
    lb_t[0].store(Bs[0]);
  } else {
    sstore(0, 0);
  }
  __syncthreads();
  const int half = lane >> 5;
  const int a_off = ARM ? (wm * 32 + (lane & 31)) * LDR + half * 16 : half * (NEWP ? 16 : 1) * LDA + wm * 32 + (lane & 31);
  const int b_off = BRM ? (wn * (32 * NT) + (lane & 31)) * LDR + half * 16 : half * (NEWP ? 16 : 1) * LDB + wn * (32 * NT) + (lane & 31);
  // operands of MFMAs 4q .. 4q+3 of a K-step (NEWP only)
  auto fetch4_a = [&](const float* as, int q, float (&o)[4]) {
    if (ARM) {
      const float4 v = *reinterpret_cast<const float4*>(as + 4 * q);
      o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = as[(4 * q + i) * LDA];
    }
  };
  auto fetch4_b = [&](const float* bs, int q, float (&o)[4]) {
    if (BRM) {
      const float4 v = *reinterpret_cast<const float4*>(bs + 4 * q);
      o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = bs[(4 * q + i) * LDB];
    }
  };
  if constexpr (SHORT) {
    constexpr int KKS = GM_BK / 2, PFS = 4;
    for (int t = 0; t < nk; ++t) {
      if (t > 0) {
        gload(k_begin + t * GM_BK, 0);
        __syncthreads();                                        // every wavefront has read step t-1's fragments
        if constexpr (ANORM) {
          sstore_a0();
          lb_t[0].store(Bs[0]);
        } else {
          sstore(0, 0);
        }
        __syncthreads();
      }
      const float* as = As[0] + a_off;
      const float* bs = Bs[0] + b_off;
      static_assert(NEWP, "the light form is only instantiated for single-accumulator tiles");
      float af[2][4], bf[2][4];                               // two groups of four MFMAs in flight
      fetch4_a(as, 0, af[0]);
      fetch4_b(bs, 0, bf[0]);
      fetch4_a(as, 1, af[1]);
      fetch4_b(bs, 1, bf[1]);
#pragma unroll
      for (int q = 0; q < KKS / 4; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i], bf[q & 1][i], acc[0], 0, 0, 0);
        if (q + 2 < KKS / 4) {
          fetch4_a(as, q + 2, af[q & 1]);
          fetch4_b(bs, q + 2, bf[q & 1]);
        }
      }
    }
  } else {
  // One K-step: compute LDS buffer BUF while (a) the tile for step t+2 is requested into register set BUF and (b) the tile
  // for step t+1 (register set 1-BUF) is written to LDS buffer 1-BUF — its ds_writes are issued BETWEEN the MFMAs, and the
  // MFMA operand fragments are read PF K-pairs ahead: LDS latency (~100+ cycles) exceeds one 64-cycle MFMA, so a one-deep
  // prefetch exposes a bubble on every MFMA when a SIMD holds a single wavefront (measured: the loop skeleton, the MFMAs and
  // the global-load stalls simply added up).
  constexpr int KK = GM_BK / 2;
  constexpr int PF = NT == 1 ? KK : (NT == 2 ? 8 : 4);           // fragment prefetch depth (registers: PF * (1 + NT))
  constexpr int PA = BM * 8 / GM_T, PB = BN * 8 / GM_T;           // LDS-store pieces per thread for the A / B tile
  constexpr int ST0 = KK - (PA + PB) - 1 > 2 ? (KK - (PA + PB)) / 2 : 1;   // first K-pair that carries a store piece
  // FULL = steps t+1 and t+2 exist: no branches in the body, which keeps the whole K-step one basic block (the waitcnt
  // insertion is only exact inside a block: with the conditional stores it waited for the loads it had just issued).
  auto step = [&](auto buf_c, auto full_c, int t) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    if (FULL || t + 2 < nk) gload(k_begin + (t + 2) * GM_BK, BUF);
    const bool do_store = FULL || t + 1 < nk;                     // block-uniform
    const float* as = As[BUF] + a_off;
    const float* bs = Bs[BUF] + b_off;
    float af[PF], bf[PF][NT];
    if constexpr (NEWP) {                                         // PF == KK: the whole K-step's operands, in 16-B reads where row-major
#pragma unroll
      for (int q = 0; q < KK / 4; ++q) {
        float a4[4], b4[4];
        fetch4_a(as, q, a4);
        fetch4_b(bs, q, b4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[4 * q + i] = a4[i];
          bf[4 * q + i][0] = b4[i];
        }
      }
    } else {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        af[d] = as[d * 2 * LDA];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[d][j] = bs[d * 2 * LDB + j * 32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                            // keep the fragment reads ahead of the MFMA chain
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int sl = kk % PF;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (NACC == 2 && (kk & 1)) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sl], bf[sl][j], acc2[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sl], bf[sl][j], acc[j], 0, 0, 0);
      }
      if (kk + PF < KK) {
        af[sl] = as[(kk + PF) * 2 * LDA];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[sl][j] = bs[(kk + PF) * 2 * LDB + j * 32];
      }
      const int piece = kk - ST0;                                 // compile-time after unrolling
      if (do_store && piece >= 0 && piece < PA + PB) {
        if (piece < PA) {
          if (TA) la_n[1 - BUF].store_piece(As[1 - BUF], piece);
          else la_t[1 - BUF].store_piece(As[1 - BUF], piece);
        } else {
          if (TB) lb_t[1 - BUF].store_piece(Bs[1 - BUF], piece - PA);
          else lb_n[1 - BUF].store_piece(Bs[1 - BUF], piece - PA);
        }
      }
    }
    __syncthreads();
  };
  int t = 0;
  for (; t + 3 < nk; t += 2) {
    step(std::integral_constant<int, 0>{}, std::true_type{}, t);
    step(std::integral_constant<int, 1>{}, std::true_type{}, t + 1);
  }
  for (; t < nk; t += 2) {
    step(std::integral_constant<int, 0>{}, std::false_type{}, t);
    if (t + 1 >= nk) break;
    step(std::integral_constant<int, 1>{}, std::false_type{}, t + 1);
  }

  }

  if (NACC == 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += acc2[j][r];
  }

  gemm_epilogue<BM, BN, WM, WN, NT>(acc, C, M, N, m0, n0, m_tile, ep, bias_v, blk_first, blk_seg_start, blk_seg_end, wm, wn, lane);
}

// ---- K-deep form: LDS-direct loads three tiles ahead, fragments of step t+1 fetched under the MFMAs of step t -----------------
// For C = A[M,K] · B[N,K]^T with K % 32 == 0 and K beyond the light form (the KPConv contractions with pre-transposed weights,
// the unary Linears of stages 3-4).  What the register-staged kernel above loses per K-step is (a) the LDS latency of the
// fragment reads it issues right after its barrier, in front of the first MFMA, and (b) ~40 VALU / LDS-store instructions that
// move a tile from registers to LDS.  Here
//   * tiles go from global memory straight into LDS (global_load_lds_dwordx4: the wavefront's 64 x 16 B land in 1 KB of
//     consecutive LDS, no staging registers, no ds_write, no zero-select); the LDS image of a tile is row-major, 128 B per row,
//     with the eight 16-B chunks of row r XOR-ed by (r >> 1) & 7 — applied on the SOURCE side (which chunk a lane fetches) and on
//     the fragment read, so a ds_read_b128 of 16 different rows touches 16 different bank groups;
//   * a ring of three LDS stages: the loads of tile t+3 are issued during step t and have two full steps (~2 x 1024 cycles of
//     MFMA) to land.  They are issued from inline assembly and retired with COUNTED waits (s_waitcnt vmcnt(loads of one tile):
//     the next tile stays in flight across the barrier) — the compiler orders every LDS read behind ALL outstanding LDS-direct
//     loads it knows of (vmcnt(0)), which with one step of cover measured slower than the register-staged kernel whenever a CU
//     holds more than one workgroup;
//   * the MFMA operand fragments of step t+1 are read from LDS during the MFMAs of step t (two fragment register sets), so a
//     step starts with its operands in registers;
//   * one barrier per step, after the step's first MFMA: it closes tile t+1 (every wavefront has waited for its own share) and
//     frees stage t % 3 — whose fragments are in registers by then — for tile t+3.
// Same K pairing and summation order as k_gemm_f32's single-accumulator tiles: results are bit-identical to its 64x64 tile.
template <int N_OUTSTANDING>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_OUTSTANDING) : "memory");
}
// one wavefront-wide 1-KB LDS-direct load: lane i's 16 B at base + off[i] land at lds_dst + 16 i (M0 = the wave-uniform LDS
// address; written in the same statement that uses it, saved and restored around it: the compiler owns M0).  Scalar base + 32-bit
// lane offset: the K-step advance is one scalar add for all loads.
__device__ __forceinline__ void glds16(const float* base, unsigned off, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(off), "s"(base), "s"(lds_dst)
               : "memory");
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(GM_T, (BM == 64 ? 3 : 2)) void k_gemm_f32_deep(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                            int64_t M, int N, int K, GemmEpilogue ep) {
  static_assert(WM * WN == 4 && BM == 32 * WM && BN == 32 * WN, "one 32x32 accumulator per wavefront");
  constexpr int LA = BM / 32, LB = BN / 32, LPW = LA + LB;   // 1-KB load instructions per wavefront and tile (8 rows each)
  constexpr int STAGE_A = BM * 128, STAGE = (BM + BN) * 128;  // bytes
  constexpr int NS = 3;
  __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w / WN, wn = w % WN;
  const int ntn = (N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;    // XCD-aware tile order, as in k_gemm_f32
  const int64_t m_tile = static_cast<int64_t>(slot / ntn) * 8 + xcd;
  const int64_t m0 = m_tile * BM;
  const int n0 = (slot % ntn) * BN;
  if (m0 >= M) return;
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));   // LDS byte address (low 32 bits of the generic pointer)

  // source byte offset of this lane for each of its load instructions: LDS slot (row, physical chunk pc) <- global chunk pc ^ ((row >> 1) & 7)
  unsigned oa[LA], ob[LB];                                     // the launcher guarantees M*K*4 and N*K*4 < 2^32
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int row = (w * LA + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int64_t gr = m0 + row;
    gr = gr < M ? gr : M - 1;                                // duplicate rows only feed outputs that are never stored
    oa[j] = static_cast<unsigned>((gr * K + c * 4) * 4);
  }
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int row = (w * LB + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int gr = n0 + row;
    gr = gr < N ? gr : N - 1;
    ob[j] = static_cast<unsigned>((static_cast<int64_t>(gr) * K + c * 4) * 4);
  }
  const unsigned dst_a = lds_base + w * LA * 1024, dst_b = lds_base + STAGE_A + w * LB * 1024;
  const float* Ak = A;                                         // scalar bases, advanced one K-step per tile issued
  const float* Bk = B;
  auto issue_one = [&](int stage, int j) {                     // j-th load instruction of the tile going to `stage` (A first, then B)
    if (j < LA) glds16(Ak, oa[j], dst_a + stage * STAGE + j * 1024);
    else glds16(Bk, ob[j - LA], dst_b + stage * STAGE + (j - LA) * 1024);
  };
  auto advance = [&] {                                         // after ALL load instructions of a tile
    Ak += GM_BK;
    Bk += GM_BK;
  };
  auto issue = [&](int stage) {
#pragma unroll
    for (int j = 0; j < LPW; ++j) issue_one(stage, j);
    advance();
  };

  // fragment addresses: lane (l & 31) owns row wm*32 + (l & 31) of A (wn*32 + .. of B), half l >> 5 the k range [16 half, 16 half + 16)
  const int half = lane >> 5;
  const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31);
  int offa[4], offb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    offa[q] = ra * 128 + (((4 * half + q) ^ ((ra >> 1) & 7)) << 4);
    offb[q] = STAGE_A + rb * 128 + (((4 * half + q) ^ ((rb >> 1) & 7)) << 4);
  }
  float4 fa[4], fb[4];                                         // the current step's fragments (k = 16 half + 4 q .. + 3 in fa[q])
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nk = K / GM_BK;

  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  // epilogue operands under the first tile's latency
  float bias_v[1];
  {
    const int col = n0 + wn * 32 + (lane & 31);
    bias_v[0] = (ep.bias && col < N) ? ep.bias[col] : 0.f;
  }
  int blk_first = 0;
  int64_t blk_seg_start = 0, blk_seg_end = 0;
  if (ep.stats != nullptr) {
    blk_seg_end = ep.seg_len[0];
    while (blk_first + 1 < ep.S && m0 >= blk_seg_end) {
      ++blk_first;
      blk_seg_start = blk_seg_end;
      blk_seg_end += ep.seg_len[blk_first];
    }
  }
  if (nk > 2) wait_vmcnt<2 * LPW>();                           // tile 0 has landed (tiles 1 and 2 may still be in flight)
  else if (nk > 1) wait_vmcnt<LPW>();
  else wait_vmcnt<0>();
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    fa[q] = *reinterpret_cast<const float4*>(lds + offa[q]);
    fb[q] = *reinterpret_cast<const float4*>(lds + offb[q]);
  }

  auto mfma = [&](float a, float b) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); };
  auto mfma4 = [&](const float4& a, const float4& b) {
    mfma(a.x, b.x);
    mfma(a.y, b.y);
    mfma(a.z, b.z);
    mfma(a.w, b.w);
  };
  auto pin = [] { __builtin_amdgcn_sched_barrier(0); };
  // step t computes tile t (its fragments are in fa / fb; its LDS stage is ST = t % 3), fetches the fragments of tile t+1 and
  // issues the loads of tile t+3.  FULL: tiles t+1 .. t+3 exist — no branches, the three steps of a ring turn are one basic block.
  // The 8 fragment reads and the LPW loads are dealt out one or two per MFMA (a 64-cycle MFMA hides ~10 issue slots; in one clump
  // behind the barrier they were ~40 instructions during which this wavefront fed the matrix pipe nothing), every position pinned:
  // left alone, the scheduler sinks the reads to their first use in the NEXT step and hoists the barrier above this step's MFMAs.
  auto step = [&](auto st_c, auto full_c, int t) {
    constexpr int ST = decltype(st_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int NXT = (ST + 1) % NS;
    float4 na[4], nb[4];
    mfma4(fa[0], fb[0]);
    const bool more = FULL || t + 1 < nk;                     // block-uniform
    const bool load = FULL || t + 3 < nk;
    pin();
    if (more) {
      if (FULL || t + 2 < nk) wait_vmcnt<LPW>();              // this wavefront's share of tile t+1 has landed; tile t+2 stays in flight
      else wait_vmcnt<0>();
      __syncthreads();                                        // ... and everybody else's; every wavefront holds its stage-ST fragments
    }
    const float a1[4] = {fa[1].x, fa[1].y, fa[1].z, fa[1].w}, b1[4] = {fb[1].x, fb[1].y, fb[1].z, fb[1].w};
    const float a2[4] = {fa[2].x, fa[2].y, fa[2].z, fa[2].w}, b2[4] = {fb[2].x, fb[2].y, fb[2].z, fb[2].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (more) {
        na[q] = *reinterpret_cast<const float4*>(lds + NXT * STAGE + offa[q]);
        nb[q] = *reinterpret_cast<const float4*>(lds + NXT * STAGE + offb[q]);
      }
      pin();
      mfma(a1[q], b1[q]);
      pin();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (more && load) {                                     // tile t+3 -> the stage this step's fragments came from
#pragma unroll
        for (int j = q; j < LPW; j += 4) issue_one(ST, j);
      }
      pin();
      mfma(a2[q], b2[q]);
      pin();
    }
    if (more && load) advance();
    mfma4(fa[3], fb[3]);
    if (more) {
#pragma unroll
      for (int q = 0; q < 4; ++q) fa[q] = na[q], fb[q] = nb[q];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int t = 0;
  for (; t + 6 <= nk; t += 3) {                                // every step of this turn has three successors
    step(I0{}, std::true_type{}, t);
    step(I1{}, std::true_type{}, t + 1);
    step(I2{}, std::true_type{}, t + 2);
  }
  for (; t < nk; t += 3) {
    step(I0{}, std::false_type{}, t);
    if (t + 1 < nk) step(I1{}, std::false_type{}, t + 1);
    if (t + 2 < nk) step(I2{}, std::false_type{}, t + 2);
  }

  floatx16 accs[1] = {acc};
  gemm_epilogue<BM, BN, WM, WN, 1>(accs, C, M, N, m0, n0, m_tile, ep, bias_v, blk_first, blk_seg_start, blk_seg_end, wm, wn, lane);
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm_deep(const float* A, const float* B, float* C, int64_t M, int N, int K, const GemmEpilogue& ep, hipStream_t st) {
  const int mt8 = (div_up(M, BM) + 7) / 8 * 8;
  LCR_LAUNCH_TIMED((k_gemm_f32_deep<BM, BN, WM, WN>), dim3(mt8 * div_up(N, BN)), dim3(GM_T), 0, st, A, B, C, M, N, K, ep);
  return check_launch("lcr_gemm_f32");
}

// ---- split-bf16 form of the K-deep contractions: fp32 operands as three bf16 terms, six cross products on the bf16 matrix cores ----
// The K-deep fp32 form above is bound by the matrix pipe (mfma_busy 0.82) on an instruction that runs at 1/16 of the bf16 rate
// (v_mfma_f32_32x32x2_f32: 64 cycles for 4 096 flop; v_mfma_f32_32x32x16_bf16: 32 cycles for 32 768).  An fp32 number is EXACTLY the sum
// of three bf16 numbers (8 + 8 + 8 significand bits: h1 = rn(x), h2 = rn(x - h1), h3 = rn(x - h1 - h2); both subtractions are exact in
// fp32), so a·b = Σ_{i,j} a_i b_j over nine exact bf16 x bf16 products.  Kept: the six with i + j <= 4 (a1b1; a1b2, a2b1; a1b3, a2b2,
// a3b1); dropped: a2b3, a3b2 (each <= 2^-24 |ab|) and a3b3 (2^-32): a product is carried to <= 2^-23 relative, accumulation in fp32 inside the
// MFMA — six K=16 instructions (192 cycles) replace eight fp32 ones (512) per 32 x 32 x 16 block.
//   * B (weights, constant) arrives pre-split: three bf16 planes [3][N][K] made once per weight by lcr_split_bf16x3;
//   * A (activations) is split ONCE PER WORKGROUP on its way from registers to LDS (8 values per thread and K-step: 44 VALU), not per
//     wavefront at the fragment read; planes are row-major bf16 tiles, 64-B rows with XOR-ed 16-B chunks (k_gemm_f32_bsplit_p);
//   * LDS double-buffered, loads two tiles ahead in registers: one barrier per K-step.
// Results are NOT bit-identical to the fp32 form (different rounding points); tests/test_gemm_split_gpu.py holds both against fp64.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));     // plain vector type: arrays of HIP's uint4 struct did not leave scratch memory

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32 (round to nearest even): a -> low half
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// two fp32 values -> their three bf16 terms (packed pairs)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h1, uint32_t& h2, uint32_t& h3) {
  h1 = pk_bf16(x0, x1);
  const float r0 = fsub(x0, __uint_as_float(h1 << 16)), r1 = fsub(x1, __uint_as_float(h1 & 0xffff0000u));     // exact
  h2 = pk_bf16(r0, r1);
  const float s0 = fsub(r0, __uint_as_float(h2 << 16)), s1 = fsub(r1, __uint_as_float(h2 & 0xffff0000u));     // exact
  h3 = pk_bf16(s0, s1);
}

__global__ __launch_bounds__(256) void k_split_bf16x3(const float* __restrict__ w, int64_t n, uint16_t* __restrict__ planes) {
  for (int64_t i = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) * 2; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x * 2) {
    const float x0 = w[i], x1 = i + 1 < n ? w[i + 1] : 0.f;
    uint32_t h1, h2, h3;
    split2(x0, x1, h1, h2, h3);
    planes[i] = static_cast<uint16_t>(h1), planes[n + i] = static_cast<uint16_t>(h2), planes[2 * n + i] = static_cast<uint16_t>(h3);
    if (i + 1 < n) planes[i + 1] = static_cast<uint16_t>(h1 >> 16), planes[n + i + 1] = static_cast<uint16_t>(h2 >> 16), planes[2 * n + i + 1] = static_cast<uint16_t>(h3 >> 16);
  }
}

// The planes of a constant operand in the layout the GEMM stages them in: for every (64-column tile ct, K-step ks of 32) one contiguous
// 12 KB block = [3 planes][64 rows][4 chunks of 8 bf16, chunk index XOR-ed with (row >> 2) & 3] — byte for byte the LDS image of the step,
// so the kernel's B loads are full, consecutive 128-B lines (thread t of 256 fetches bytes [16 t, 16 t + 16) of each 4 KB plane block).
// Rows beyond N are zero.  One thread per (row, chunk) pair of a block.
__global__ __launch_bounds__(256) void k_split_bf16x3_tiles(const float* __restrict__ w, int N, int K, uint16_t* __restrict__ tiles) {
  const int KS = K / GM_BK;
  const int64_t blk = blockIdx.x;                               // ct * KS + ks
  const int ct = static_cast<int>(blk / KS), ks = static_cast<int>(blk % KS);
  const int row = threadIdx.x >> 2, kc = threadIdx.x & 3;
  const int n = ct * 64 + row;
  u32x4_t p1 = {0u, 0u, 0u, 0u}, p2 = p1, p3 = p1;
  if (n < N) {
    const float* src = w + static_cast<int64_t>(n) * K + ks * GM_BK + kc * 8;
    const float4 x0 = ld4(src), x1 = ld4(src + 4);
    uint32_t h1, h2, h3;
    split2(x0.x, x0.y, h1, h2, h3), p1.x = h1, p2.x = h2, p3.x = h3;
    split2(x0.z, x0.w, h1, h2, h3), p1.y = h1, p2.y = h2, p3.y = h3;
    split2(x1.x, x1.y, h1, h2, h3), p1.z = h1, p2.z = h2, p3.z = h3;
    split2(x1.z, x1.w, h1, h2, h3), p1.w = h1, p2.w = h2, p3.w = h3;
  }
  char* dst = reinterpret_cast<char*>(tiles) + blk * 12288 + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4);
  *reinterpret_cast<u32x4_t*>(dst) = p1;
  *reinterpret_cast<u32x4_t*>(dst + 4096) = p2;
  *reinterpret_cast<u32x4_t*>(dst + 8192) = p3;
}

// 64 x 64 tile, 2 x 2 wavefronts of 32 x 32 (the stage-3/4 contractions are 200-600 such tiles on 256 CUs).  Both operands' planes in LDS,
// DOUBLE-buffered: the split + plane writes of tile t+1 run under the MFMAs of tile t and a K-step has ONE barrier; tiles t+1 and t+2 in flight /
// parked in two register stages; every load instruction fetches full 128-B lines (A: 8 lanes per row; B: the pre-tiled planes, verbatim).
// What was measured on the way (tools/gemm_split_bench.py, LABNOTES.md §4.4), each against the fp32 K-deep kernel over the 13 bench shapes:
// single-buffered LDS with two barriers per step 1.00x; larger tiles (128 x 64, 128 x 128 on 4 wavefronts; 128 x 64 on 8) lose on the encoder's
// shapes (too few tiles) and win on 8192 x 1024 x 1024 (1.45x); A straight from global memory into fragment registers 0.92x; four register
// stages: no change; 80-byte padded LDS rows: a third of the LDS cycles were bank conflicts of the 16-B WRITES (SQ_LDS_BANK_CONFLICT) — the XOR
// layout: 1.07x -> 1.17x; two alternating accumulators: same speed, half the error; producer / consumer wavefronts (512 threads): same time;
// ablations (LCR_SPLIT_ABL): loads alone 75 us and MFMAs alone 47 us of 4_2's 98 — the loads were bound by LINE REQUESTS (every A line asked
// for twice, B rows half lines): full-line loads + tiled planes: loads alone 53 us — and the whole kernel still 100 (1.18x overall,
// 1.23-1.28x on 2_2 / 3_2 / 4_2 / the stage-3/4 Linears): every ablation removes its own 10-25 %, i.e. the K-step is a latency chain
// (arrival -> split -> park -> barrier -> fragment reads -> 12 MFMAs) that two to three workgroups per CU do not cover; matrix pipe 38 % busy.
// Raising the wavefront's priority over its MFMA section (s_setprio) 0.98x, parking the next tile before the MFMAs instead of between them 1.00x;
// B planes global -> LDS directly (global_load_lds_dwordx4, three B stages, no staging registers / ds_write for B): 0.89x (60 KB of LDS: two
// workgroups per CU instead of three).  A 64 x 128 form of THIS pipeline (wavefront = 32 x 64, two accumulators; half the splits, 9 fragment
// reads per 12 MFMAs instead of 6 per 6, 16 instead of 20 KB loaded per 64 x 64 x 32): 160 TFLOP/s-equivalent on 8192 x 1024 x 1024 against this
// form's 146 (so instructions per flop ARE the bound once the chip is full) but 0.93x over the 13 shapes: 4_2 is 204 such tiles, 3_2 298 on
// 512 workgroup slots (4_2 104 us against 97, 4_1 56 against 36) — removed.
// ABL: timing ablations (tools/gemm_split_bench.py --abl): 1 no MFMAs, 2 no split arithmetic, 4 no plane stores, 8 no global loads in the
// loop, 16 no fragment reads.  0 = the product; any other value computes garbage.
// NWM: 32-row wavefront rows of the tile (2: 64 x 64, 256 threads — the product; 4: 128 x 64, 512 threads: measured 1.08x, not instantiated).
template <int ABL, int NWM, int D = 2>
__global__ __launch_bounds__(128 * NWM, 2) void k_gemm_f32_bsplit_p(const float* __restrict__ A, const uint16_t* __restrict__ Bs, float* __restrict__ C,
                                                             int64_t M, int N, int K, GemmEpilogue ep) {
  // LDS plane tile: 64 rows x 64 B (32 bf16), no padding; the four 16-B chunks of row r are XOR-ed with (r >> 2) & 3.  A ds_write_b128 is
  // serviced in groups of 8 consecutive lanes over 32 banks (two rows x four chunks: 128 distinct bytes), a ds_read_b128 in groups of 16 lanes
  // over 64 banks (16 different rows, one logical chunk: rows r, r+4, r+8, r+12 share a 64-B bank quadrant and get four different chunks).
  // (The 80-byte padded rows of the first version were conflict-free for the reads only: SQ_LDS_BANK_CONFLICT = a third of SQ_LDS_IDX_ACTIVE.)
  constexpr int BM = 32 * NWM, BN = 64, RS = 64, T = 128 * NWM, PLA = BM * RS, STGA = 3 * PLA, PL = 64 * RS;
  constexpr int NPB = (768 + T - 1) / T;                        // 16-B pieces of the B planes per thread and K-step (3 planes x 64 rows x 4 chunks)
  constexpr int STG = 3 * PL + (NPB * T - 768) * 16;            // a B stage + a dump area for the surplus pieces of the last round
  __shared__ __attribute__((aligned(16))) char sA[2 * STGA];
  __shared__ __attribute__((aligned(16))) char sB[2 * STG];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntn = (N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t m_tile = static_cast<int64_t>(slot / ntn) * 8 + xcd;
  const int64_t m0 = m_tile * BM;
  const int n0 = (slot % ntn) * BN;
  if (m0 >= M) return;
  // A staging: the tile is BM rows x eight 16-B pieces; thread t takes pieces t and t + T, i.e. (row t >> 3 (+ T/8), piece t & 7): the 8 lanes
  // of a row fetch one full 128-B line per load instruction (with 4 lanes x two strided 16-B pieces per row, every line was requested twice,
  // and the kernel is bound by line requests: see the ablations).  A piece = 4 floats -> 8 bytes per plane (ds_write_b64).
  const int srow = threadIdx.x >> 3, spc = threadIdx.x & 7;
  const float* ap[2];
  int soff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = srow + j * (T / 8);
    int64_t ga = m0 + r;
    ga = ga < M ? ga : M - 1;                                  // duplicate rows only feed outputs that are never stored
    ap[j] = A + ga * K + spc * 4;
    soff[j] = r * RS + (((spc >> 1) ^ ((r >> 2) & 3)) << 4) + (spc & 1) * 8;
  }
  // B staging: the step's 12 KB block of the tiled planes, copied verbatim (it IS the LDS image): thread t takes bytes [16 (t + T j), + 16).
  // 768 pieces over T threads: with T = 512 the second round has 256 surplus threads — they re-fetch piece t and park it in a dump area
  // behind the stages instead of branching (a conditional load or store splits the K-step into basic blocks, and the compiler's s_waitcnt
  // counting across blocks waits for the NEXT tile's loads too: the two-tile prefetch collapses to one).
  const char* bp = reinterpret_cast<const char*>(Bs) + static_cast<int64_t>(n0 / 64) * (K / GM_BK) * 12288;
  int bsrc[NPB], bdst[NPB];
#pragma unroll
  for (int j = 0; j < NPB; ++j) {
    const int q = threadIdx.x + T * j;
    bsrc[j] = (q < 768 ? q : threadIdx.x) * 16;
    bdst[j] = q * 16;                                            // q >= 768 lands in the stage's dump area
  }
  const int fsw = ((lane & 31) >> 2) & 3, fc0 = ((lane >> 5) ^ fsw) << 4, fc1 = (((lane >> 5) + 2) ^ fsw) << 4;     // chunks of K16 block 0 / 1
  const int fa = (wm * 32 + (lane & 31)) * RS, fb = (wn * 32 + (lane & 31)) * RS;
  floatx16 acc, acc2;                                          // two accumulators, alternating: consecutive MFMAs are independent
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc2[r] = 0.f;
  const int nk = K / GM_BK;
  f32x4_t ra[D][2];                                            // D register stages: tiles t+1 .. t+D in flight / parked (two pieces each)
  u32x4_t rb[D][NPB];
  auto fetch = [&](auto st_c, int t) {
    constexpr int st = decltype(st_c)::value;
    const int tt = t < nk ? t : nk - 1;                         // the tail re-reads the last tile
    ra[st][0] = *reinterpret_cast<const f32x4_t*>(ap[0] + tt * GM_BK), ra[st][1] = *reinterpret_cast<const f32x4_t*>(ap[1] + tt * GM_BK);
#pragma unroll
    for (int j = 0; j < NPB; ++j) rb[st][j] = *reinterpret_cast<const u32x4_t*>(bp + static_cast<int64_t>(tt) * 12288 + bsrc[j]);
  };
  auto park = [&](auto st_c, int lds_stage) {                  // register stage -> (split) -> LDS stage
    constexpr int st = decltype(st_c)::value;
    char* da = sA + lds_stage * STGA;
    char* db = sB + lds_stage * STG;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t h1, h2, h3, g1, g2, g3;
      if constexpr (ABL & 2) {
        h1 = __float_as_uint(ra[st][j].x), h2 = __float_as_uint(ra[st][j].y), h3 = h1, g1 = __float_as_uint(ra[st][j].z), g2 = __float_as_uint(ra[st][j].w), g3 = g1;
      } else {
        split2(ra[st][j].x, ra[st][j].y, h1, h2, h3);
        split2(ra[st][j].z, ra[st][j].w, g1, g2, g3);
      }
      if constexpr (ABL & 4) {
        asm volatile("" ::"v"(h1), "v"(h2), "v"(h3), "v"(g1), "v"(g2), "v"(g3));
      } else {
        *reinterpret_cast<uint2*>(da + soff[j]) = make_uint2(h1, g1);
        *reinterpret_cast<uint2*>(da + PLA + soff[j]) = make_uint2(h2, g2);
        *reinterpret_cast<uint2*>(da + 2 * PLA + soff[j]) = make_uint2(h3, g3);
      }
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const u32x4_t v = rb[st][j];
      if constexpr (ABL & 4) {
        asm volatile("" ::"v"(v));
      } else {
        *reinterpret_cast<u32x4_t*>(db + bdst[j]) = v;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  // register stage of tile t is t % D; LDS stage t & 1
  fetch(I0{}, 0);
  if constexpr (D > 1) fetch(std::integral_constant<int, 1 % D>{}, 1);
  if constexpr (D > 2) fetch(std::integral_constant<int, 2 % D>{}, 2);
  if constexpr (D > 3) fetch(std::integral_constant<int, 3 % D>{}, 3);
  float bias_v[1];
  {
    const int col = n0 + wn * 32 + (lane & 31);
    bias_v[0] = (ep.bias && col < N) ? ep.bias[col] : 0.f;
  }
  int blk_first = 0;
  int64_t blk_seg_start = 0, blk_seg_end = 0;
  if (ep.stats != nullptr) {
    blk_seg_end = ep.seg_len[0];
    while (blk_first + 1 < ep.S && m0 >= blk_seg_end) {
      ++blk_first;
      blk_seg_start = blk_seg_end;
      blk_seg_end += ep.seg_len[blk_first];
    }
  }
  park(I0{}, 0);
  fetch(I0{}, D);
  __syncthreads();
  // step t: LDS stage t & 1 holds tile t; register stage (t+1) % D holds tile t+1, which is split into the other LDS stage under this
  // step's MFMAs; its registers then take tile t+1+D
  auto step = [&](auto lds_c, auto reg_c, int t) {
    constexpr int P = decltype(lds_c)::value;                  // t & 1
    using NX = std::integral_constant<int, (decltype(reg_c)::value + 1) % D>;      // register stage of tile t+1
    const char* la = sA + P * STGA + fa;
    const char* lb = sB + P * STG + fb;
    bf16x8_t a0[3], b0[3], a1[3], b1[3];
    auto mm = [&](const bf16x8_t& x, const bf16x8_t& y, floatx16& c) {
      if constexpr (ABL & 1) {
        asm volatile("" ::"v"(x), "v"(y));                        // operands stay live (their loads are not dead), no matrix instruction
      } else {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
      }
    };
    if constexpr (ABL & 16) {
      u32x4_t z = {static_cast<uint32_t>(t), 0u, 0u, 0u};
      asm volatile("" : "+v"(z));
#pragma unroll
      for (int p = 0; p < 3; ++p) a0[p] = b0[p] = a1[p] = b1[p] = __builtin_bit_cast(bf16x8_t, z);
      (void)la, (void)lb;
    } else {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a0[p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(la + p * PLA + fc0));
        b0[p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(lb + p * PL + fc0));
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a1[p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(la + p * PLA + fc1));
        b1[p] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(lb + p * PL + fc1));
      }
    }
    mm(a0[2], b0[0], acc);
    mm(a0[0], b0[2], acc2);
    mm(a0[1], b0[1], acc);
    mm(a0[1], b0[0], acc2);
    mm(a0[0], b0[1], acc);
    mm(a0[0], b0[0], acc2);
    park(NX{}, P ^ 1);                                         // tile t+1 -> the other LDS stage (its last readers passed the barrier of step t-1)
    mm(a1[2], b1[0], acc);
    mm(a1[0], b1[2], acc2);
    mm(a1[1], b1[1], acc);
    mm(a1[1], b1[0], acc2);
    mm(a1[0], b1[1], acc);
    mm(a1[0], b1[0], acc2);
    if constexpr (!(ABL & 8)) fetch(NX{}, t + 1 + D);
    __syncthreads();
  };
  static_assert(D == 2 || D == 4, "register stages");
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  int t = 0;
  if constexpr (D == 2) {
    for (; t + 2 <= nk; t += 2) {
      step(I0{}, I0{}, t);
      step(I1{}, I1{}, t + 1);
    }
    if (t < nk) step(I0{}, I0{}, t);
  } else {
    for (; t + 4 <= nk; t += 4) {
      step(I0{}, I0{}, t);
      step(I1{}, I1{}, t + 1);
      step(I0{}, I2{}, t + 2);
      step(I1{}, I3{}, t + 3);
    }
    if (t < nk) step(I0{}, I0{}, t);
    if (t + 1 < nk) step(I1{}, I1{}, t + 1);
    if (t + 2 < nk) step(I0{}, I2{}, t + 2);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  floatx16 accs[1] = {acc};
  gemm_epilogue<BM, BN, NWM, 2, 1>(accs, C, M, N, m0, n0, m_tile, ep, bias_v, blk_first, blk_seg_start, blk_seg_end, wm, wn, lane);
}

// ---- stream-K form of the K-deep contractions ---------------------------------------------------------------------------
// With one tile per workgroup, 408 / 596 / 806 tiles of 64x64 on 256 CUs (<= 3 resident workgroups each) leave some CUs with one
// tile more than others: the deepest KPConv contractions lose ~20 % to that quantisation.  Here the grid is a fixed number of
// persistent workgroups (a multiple of the CU count) and every workgroup takes an equal, CONTIGUOUS range of (tile, K-step)
// iterations, so it touches at most two partial tiles.  Nobody waits for anybody: a workgroup parks every partial tile it
// computes (16 KB, device-coherent stores), then bumps the tile's arrival counter; the workgroup that arrives LAST folds all the
// pieces of the tile in ascending workgroup order (its own included, read back like the others: the sum is the same whoever
// folds), runs the epilogue and resets the counter for the next launch.  (Two earlier protocols with an owner waiting for the
// others' flags: in range order the waits chain up — 20x slower; with the shared pieces computed first an owner still idles
// whenever its neighbour's piece is the longer one — erratic, 0.8-1.4x.)  Tiles are dealt to the XCDs as in k_gemm_f32 (row
// block % 8) and the iteration ranges are cut inside one XCD's band, so the column tiles of a row block stay on one L2.
struct GemmStreamK {
  float*    partial;   // [grid][2][NT * 16][GM_T]: slot 0 = the piece a workgroup's range starts with, slot 1 = any later piece
  uint32_t* arrived;   // [tiles] pieces parked so far; zero between launches
  int       U;         // workgroups per XCD band (grid = 8 * U)
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(GM_T, 3) void k_gemm_f32_sk(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int64_t M, int N, int K, GemmEpilogue ep, GemmStreamK sk) {
  static_assert(WM * WN == 4 && BM == 32 * WM, "one 32-row MFMA tile per wavefront along M");
  constexpr int LDA = BM + 1, LDB = BN + 4;
  constexpr int NT = BN / (32 * WN);
  __shared__ __attribute__((aligned(16))) float As[2][GM_BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][GM_BK * LDB];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w / WN, wn = w % WN;
  const int ntn = (N + BN - 1) / BN;
  const int64_t mt = (M + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
  const int64_t mt_x = mt > xcd ? (mt - xcd + 7) / 8 : 0;          // row blocks of this XCD's band: m_tile = 8 * i + xcd
  const int nk = (K + GM_BK - 1) / GM_BK;
  const int64_t I = mt_x * ntn * nk;                                // iterations of the band
  auto first_it = [&](int uu) { return static_cast<int64_t>(uu) * I / sk.U; };
  const int64_t it_begin = first_it(u), it_end = first_it(u + 1);
  LoaderT<BM, LDA, true> la[2];
  LoaderN<BN, LDB, true> lb[2];
  const int a_off = (lane >> 5) * LDA + wm * 32 + (lane & 31);
  const int b_off = (lane >> 5) * LDB + wn * (32 * NT) + (lane & 31);
  constexpr int PIECE = NT * 16 * GM_T;                             // floats of one parked partial tile
  __shared__ unsigned s_arrived;

  int64_t it = it_begin;
  while (it < it_end) {
    const int64_t lt = it / nk;
    const int ks = static_cast<int>(it - lt * nk);
    const int ns = static_cast<int>(min(static_cast<int64_t>(nk - ks), it_end - it));        // K-steps of this segment
    const int64_t m_tile = (lt / ntn) * 8 + xcd;
    const int64_t m0 = m_tile * BM;
    const int n0 = static_cast<int>(lt % ntn) * BN;
    const int k_begin = ks * GM_BK;
    auto gload = [&](int k0, int r) {
      la[r].load(A, M, K, m0, k0);
      lb[r].load(B, N, K, n0, k0);
    };
    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    gload(k_begin, 0);
    gload(k_begin + GM_BK, 1);
    const bool covers_end = ks + ns == nk, covers_start = ks == 0;
    float bias_v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * (32 * NT) + j * 32 + (lane & 31);
      bias_v[j] = (ep.bias && col < N) ? ep.bias[col] : 0.f;
    }
    int blk_first = 0;
    int64_t blk_seg_start = 0, blk_seg_end = 0;
    la[0].store(As[0]);
    lb[0].store(Bs[0]);
    __syncthreads();

    // the K-step of k_gemm_f32 (see there): fragments PF K-pairs ahead, next tile's LDS stores between the MFMAs
    constexpr int KK = GM_BK / 2;
    constexpr int PF = NT == 1 ? KK : (NT == 2 ? 8 : 4);
    constexpr int PA = BM * 8 / GM_T, PB = BN * 8 / GM_T;
    constexpr int ST0 = KK - (PA + PB) - 1 > 2 ? (KK - (PA + PB)) / 2 : 1;
    auto step = [&](auto buf_c, auto full_c, int t) {
      constexpr int BUF = decltype(buf_c)::value;
      constexpr bool FULL = decltype(full_c)::value;
      if (FULL || t + 2 < ns) gload(k_begin + (t + 2) * GM_BK, BUF);
      const bool do_store = FULL || t + 1 < ns;
      const float* as = As[BUF] + a_off;
      const float* bs = Bs[BUF] + b_off;
      float af[PF], bf[PF][NT];
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        af[d] = as[d * 2 * LDA];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[d][j] = bs[d * 2 * LDB + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int sl = kk % PF;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sl], bf[sl][j], acc[j], 0, 0, 0);
        if (kk + PF < KK) {
          af[sl] = as[(kk + PF) * 2 * LDA];
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[sl][j] = bs[(kk + PF) * 2 * LDB + j * 32];
        }
        const int piece = kk - ST0;
        if (do_store && piece >= 0 && piece < PA + PB) {
          if (piece < PA) la[1 - BUF].store_piece(As[1 - BUF], piece);
          else lb[1 - BUF].store_piece(Bs[1 - BUF], piece - PA);
        }
      }
      __syncthreads();
    };
    int t = 0;
    for (; t + 3 < ns; t += 2) {
      step(std::integral_constant<int, 0>{}, std::true_type{}, t);
      step(std::integral_constant<int, 1>{}, std::true_type{}, t + 1);
    }
    for (; t < ns; t += 2) {
      step(std::integral_constant<int, 0>{}, std::false_type{}, t);
      if (t + 1 >= ns) break;
      step(std::integral_constant<int, 1>{}, std::false_type{}, t + 1);
    }

    bool finish = true;
    if (!(covers_start && covers_end)) {
      // park this piece: device-coherent stores (a device-scope release FENCE per workgroup would write back the XCD's whole L2:
      // measured 2x slower overall), then count it in once every lane's stores are acknowledged
      float* mine = sk.partial + (static_cast<int64_t>(blockIdx.x) * 2 + (it == it_begin ? 0 : 1)) * PIECE;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __hip_atomic_store(&mine[(j * 16 + r) * GM_T + threadIdx.x], acc[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the workgroups that hold pieces of this tile: uf .. ul of this band (those with a non-empty range)
      const int64_t tile_it0 = lt * nk, tile_it1 = tile_it0 + nk;
      int uf = static_cast<int>(tile_it0 * sk.U / I);
      while (uf > 0 && first_it(uf) > tile_it0) --uf;
      while (first_it(uf + 1) <= tile_it0) ++uf;
      int ul = uf, pieces = 0;
      for (int uu = uf; uu < sk.U && first_it(uu) < tile_it1; ++uu)
        if (first_it(uu + 1) > first_it(uu)) {
          ul = uu;
          ++pieces;
        }
      __builtin_amdgcn_s_waitcnt(0);                                       // THIS wavefront's parked values are acknowledged ...
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();                                                     // ... and so are the other three's, before the count
      uint32_t* cnt = sk.arrived + (m_tile * ntn + lt % ntn);
      if (threadIdx.x == 0) s_arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      finish = s_arrived == static_cast<unsigned>(pieces - 1);           // block-uniform: the last piece to arrive folds the tile
      if (finish) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int uu = uf; uu <= ul; ++uu) {
          if (first_it(uu + 1) <= first_it(uu)) continue;
          const float* pp = sk.partial + (static_cast<int64_t>(uu * 8 + xcd) * 2 + (first_it(uu) >= tile_it0 ? 0 : 1)) * PIECE;
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[j][r] += __hip_atomic_load(&pp[(j * 16 + r) * GM_T + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (finish) {
      if (ep.stats != nullptr) {                                           // GroupNorm segment of this row block
        blk_first = 0;
        blk_seg_start = 0;
        blk_seg_end = ep.seg_len[0];
        while (blk_first + 1 < ep.S && m0 >= blk_seg_end) {
          ++blk_first;
          blk_seg_start = blk_seg_end;
          blk_seg_end += ep.seg_len[blk_first];
        }
      }
      gemm_epilogue<BM, BN, WM, WN, NT>(acc, C, M, N, m0, n0, m_tile, ep, bias_v, blk_first, blk_seg_start, blk_seg_end, wm, wn, lane);
    }
    it += ns;
    __syncthreads();                                                       // LDS is reused by the next segment
  }
}

template <int BM, int BN, int WM, int WN, bool VEC, bool SHORT = false>
static int launch_gemm(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const GemmEpilogue& ep,
                       hipStream_t st, const GemmBatch* batch = nullptr) {
  static const GemmBatch no_batch = {};
  const GemmBatch& bt = batch ? *batch : no_batch;
  const int mt8 = (div_up(M, BM) + 7) / 8 * 8;
  dim3 grid(mt8 * div_up(N, BN), 1, bt.count ? bt.count : 1);     // see the XCD-aware tile order in the kernel
  dim3 block(GM_T);
  if (!transA && !transB) LCR_LAUNCH_TIMED((k_gemm_f32<BM, BN, WM, WN, false, false, VEC, SHORT>), grid, block, 0, st, A, B, C, M, N, K, ep, bt, ANorm{});
  else if (!transA && transB) LCR_LAUNCH_TIMED((k_gemm_f32<BM, BN, WM, WN, false, true, VEC, SHORT>), grid, block, 0, st, A, B, C, M, N, K, ep, bt, ANorm{});
  else if (transA && !transB) LCR_LAUNCH_TIMED((k_gemm_f32<BM, BN, WM, WN, true, false, VEC, SHORT>), grid, block, 0, st, A, B, C, M, N, K, ep, bt, ANorm{});
  else LCR_LAUNCH_TIMED((k_gemm_f32<BM, BN, WM, WN, true, true, VEC, SHORT>), grid, block, 0, st, A, B, C, M, N, K, ep, bt, ANorm{});
  return check_launch("lcr_gemm_f32");
}

// Scratch of the stream-K launches, one per stream (launches on one stream are ordered, so they can share it): the parked
// partial tiles (two per workgroup) and the per-tile arrival counters (left at zero by every launch).
struct SkScratch {
  float*    partial = nullptr;
  uint32_t* arrived = nullptr;
};
static std::mutex g_sk_mu;
static std::map<std::pair<int, hipStream_t>, SkScratch> g_sk_scratch;
constexpr int SK_MAX_GRID = 1024;
constexpr int SK_MAX_TILES = 8192;

static int sk_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }();
  return n;
}

// Stream-K pays when whole-tile scheduling leaves the busiest CU with >= 10 % more work than the average one.
static bool sk_worth(int64_t M, int N, int K) {
  const int cus = sk_cu_count();
  const int64_t tiles = static_cast<int64_t>(div_up(M, 64)) * div_up(N, 64);
  if (K < 480 || tiles < cus / 2 || tiles > 6 * static_cast<int64_t>(cus) || tiles > SK_MAX_TILES) return false;
  const double avg = static_cast<double>(tiles) / cus;
  const double worst = static_cast<double>((tiles + cus - 1) / cus);
  return worst / avg >= 1.10;
}

static int launch_gemm_sk(const float* A, const float* B, float* C, int64_t M, int N, int K, const GemmEpilogue& ep, hipStream_t st) {
  const int cus = sk_cu_count();
  const int64_t iters = static_cast<int64_t>(div_up(M, 64)) * div_up(N, 64) * div_up(K, GM_BK);
  // persistent workgroups: 3, 2 or 1 per CU — as many as still get >= 24 K-steps each
  int per_cu = 3;
  while (per_cu > 1 && iters / (static_cast<int64_t>(cus) * per_cu) < 24) --per_cu;
  if (getenv("LCR_SK_PER_CU")) per_cu = atoi(getenv("LCR_SK_PER_CU"));
  int U = cus * per_cu / 8;
  if (U < 1) U = 1;
  if (8 * U > SK_MAX_GRID) U = SK_MAX_GRID / 8;
  int dev = 0;
  hipGetDevice(&dev);
  GemmStreamK sk;
  {
    std::lock_guard<std::mutex> lk(g_sk_mu);
    SkScratch& sc = g_sk_scratch[{dev, st}];
    if (!sc.partial) {
      if (hipMalloc(reinterpret_cast<void**>(&sc.partial), sizeof(float) * 2 * 16 * GM_T * SK_MAX_GRID) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&sc.arrived), sizeof(uint32_t) * SK_MAX_TILES) != hipSuccess) {
        set_error("lcr_gemm_f32: cannot allocate the stream-K scratch");
        return LCR_EHIP;
      }
      hipMemsetAsync(sc.arrived, 0, sizeof(uint32_t) * SK_MAX_TILES, st);     // once: every launch leaves the counters at zero
    }
    sk.partial = sc.partial;
    sk.arrived = sc.arrived;
    sk.U = U;
  }
  LCR_LAUNCH_TIMED((k_gemm_f32_sk<64, 64, 2, 2>), dim3(8 * U), dim3(GM_T), 0, st, A, B, C, M, N, K, ep, sk);
  return check_launch("lcr_gemm_f32");
}

}  // namespace lcr

using namespace lcr;

// tuning hook (tools/gemm_bench.py): 0 = heuristic, 1..5 = force a tile shape
static int g_force_tile = 0;
extern "C" void lcr_gemm_debug_force_tile(int t) { g_force_tile = t; }
// tuning / test hook: -1 = LCR_GEMM_STREAMK (default 1 = heuristic), 0 = never, 2 = whenever the stream-K kernel is legal
static int g_force_streamk = -1;
// tuning / test hook: -1 = LCR_GEMM_DEEP (default on), 0 = never, 1 = wherever legal
static int g_force_deep = -1;
extern "C" void lcr_gemm_debug_deep(int mode) { g_force_deep = mode; }
extern "C" void lcr_gemm_debug_streamk(int mode) { g_force_streamk = mode; }

static int gemm_impl(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const float* bias,
                     const float* rowdiv, const int64_t* seg_len, int S, int groups, double* stats, void* stream);

// C = leaky(GroupNorm(A)) · B^T (+ bias, + statistics of C): the light GEMM with normalise-on-load (ANorm above).  The caller
// guarantees that every segment holds at least 64 rows (a row block then touches at most two segments); shapes outside the
// light form are an argument error — the caller normalises with lcr_groupnorm_apply and calls lcr_gemm_f32 instead.
extern "C" int lcr_gemm_f32_anorm(const float* A, const float* B, float* C, int64_t M, int N, int K, const float* bias,
                                  const double* a_stats, const float* a_gamma, const float* a_beta, int a_groups, float a_eps,
                                  float a_slope, const int64_t* seg_len, int S, int groups, double* stats, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 32 || K <= 0 || K > AN_KMAX || K % 4 != 0 || !a_stats || !a_gamma || !a_beta || a_groups < 1 ||
      K % a_groups != 0 || a_groups > 64 || !seg_len || S < 1 || reinterpret_cast<uintptr_t>(A) % 16 != 0 || reinterpret_cast<uintptr_t>(B) % 16 != 0) {
    set_error("lcr_gemm_f32_anorm: needs 32 < N, K <= 256, K % 4 == 0, a_groups dividing K, a segment table and 16-byte aligned operands");
    return LCR_EARG;
  }
  if (stats) {
    const int gs = groups >= 1 && N % groups == 0 ? N / groups : 3;
    if ((gs & (gs - 1)) != 0) {
      set_error("lcr_gemm_f32_anorm: statistics need groups dividing N into power-of-two sized groups");
      return LCR_EARG;
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  KernelTimerScope timed(KT_GEMM, st, M, N, K);
  if (M == 0) return LCR_OK;
  GemmEpilogue ep{bias, nullptr, seg_len, S, groups, stats};
  ANorm an{a_stats, a_gamma, a_beta, a_groups, a_eps, a_slope};
  const int mt8 = (div_up(M, 64) + 7) / 8 * 8;
  LCR_LAUNCH_TIMED((k_gemm_f32<64, 64, 2, 2, false, true, true, true, true>), dim3(mt8 * div_up(N, 64)), dim3(GM_T), 0, st, A, B, C, M,
                     N, K, ep, GemmBatch{}, an);
  return check_launch("lcr_gemm_f32_anorm");
}

extern "C" int lcr_gemm_f32(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const float* bias,
                            const float* rowdiv, const int64_t* seg_len, int S, int groups, double* stats, void* stream) {
  return gemm_impl(A, B, C, M, N, K, transA, transB, bias, rowdiv, seg_len, S, groups, stats, stream);
}

static int gemm_impl(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const float* bias,
                     const float* rowdiv, const int64_t* seg_len, int S, int groups, double* stats, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) {
    set_error("lcr_gemm_f32: bad argument");
    return LCR_EARG;
  }
  if (stats && (!seg_len || S < 1 || groups < 1 || N % groups != 0)) {
    set_error("lcr_gemm_f32: statistics need seg_len, S >= 1 and groups dividing N");
    return LCR_EARG;
  }
  if (stats) {
    const int gs = N / groups;
    if ((gs & (gs - 1)) != 0) {
      set_error("lcr_gemm_f32: channels per group must be a power of two");
      return LCR_EARG;
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  KernelTimerScope timed(KT_GEMM, st, M, N, K);
  // `stats` ACCUMULATES: the caller passes a zeroed table (one fill per forward pass for all layers, instead of a fill
  // launch in front of every GEMM)
  if (M == 0) return LCR_OK;
  GemmEpilogue ep{bias, rowdiv, seg_len, S, groups, stats};
  // 16-byte vector loads need leading dimensions that are multiples of 4 floats (bases: torch allocations are >= 256-B aligned,
  // row offsets inside them are multiples of the leading dimension)
  const int64_t lda = transA ? M : K, ldb = transB ? K : N;
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
  if (!vec) return launch_gemm<128, 64, 4, 1, false>(A, B, C, M, N, K, transA, transB, ep, st);
  switch (g_force_tile) {
    case 1: return launch_gemm<128, 128, 4, 1, true>(A, B, C, M, N, K, transA, transB, ep, st);
    case 2: return launch_gemm<128, 64, 4, 1, true>(A, B, C, M, N, K, transA, transB, ep, st);
    case 3: return launch_gemm<128, 32, 4, 1, true>(A, B, C, M, N, K, transA, transB, ep, st);
    case 4: return launch_gemm<64, 64, 2, 2, true>(A, B, C, M, N, K, transA, transB, ep, st);
    case 5: return launch_gemm<64, 128, 2, 2, true>(A, B, C, M, N, K, transA, transB, ep, st);
    default: break;
  }
  // Tile choice from tools/gemm_bench.py --tiles on MI355X: 64x64 (2x2 wavefronts) wins or ties on every encoder shape with
  // N >= 64 (enough workgroups for two per CU matters more than per-tile reuse at these M); N = 32 wants the 128-row tile;
  // only large, deep problems (>= 512 tiles of 128x128 and K >= 512) pay for the big tile.
  const int64_t b128 = (M + 127) / 128, nb128 = (N + 127) / 128;
  static const bool no_short = getenv("LCR_GEMM_NO_SHORT") != nullptr;   // A/B switch while the light form is being evaluated
  static const int short_k = getenv("LCR_GEMM_SHORT_K") ? atoi(getenv("LCR_GEMM_SHORT_K")) : 256;
  // K-deep form (LDS-direct loads + cross-step fragment prefetch) wherever both operands are k-contiguous and K is a multiple of the
  // K-step (LCR_GEMM_DEEP=0: off; 1: K beyond the light form; 2: every such K)
  static const int deep_env = getenv("LCR_GEMM_DEEP") ? atoi(getenv("LCR_GEMM_DEEP")) : 1;
  const int deep_mode = g_force_deep >= 0 ? g_force_deep : deep_env;
  const bool deep_ok = !transA && transB && K % GM_BK == 0 && !g_force_tile && M * K < (int64_t(1) << 30) && static_cast<int64_t>(N) * K < (int64_t(1) << 30);
  auto deep = [&] {
    if (N <= 32) return launch_gemm_deep<128, 32, 4, 1>(A, B, C, M, N, K, ep, st);
    return launch_gemm_deep<64, 64, 2, 2>(A, B, C, M, N, K, ep, st);
  };
  if (deep_mode == 2 && deep_ok) return deep();
  if (!no_short && K <= short_k && !transA) {
    if (N <= 32) return launch_gemm<128, 32, 4, 1, true, true>(A, B, C, M, N, K, transA, transB, ep, st);
    return launch_gemm<64, 64, 2, 2, true, true>(A, B, C, M, N, K, transA, transB, ep, st);
  }
  if (deep_mode && deep_ok) return deep();
  if (N <= 32) return launch_gemm<128, 32, 4, 1, true>(A, B, C, M, N, K, transA, transB, ep, st);
  // Stream-K is opt-in: measured +3 / +5 / +7 % on the three deepest KPConv contractions alone (93.6 vs 96, 121 vs 128, 153 vs
  // 164 us) — far from the 20 % a CU-count model predicts, because one or two workgroups per CU already reach 40 / 60 % of the
  // matrix-pipe rate that three reach (65 %), so an unevenly loaded CU is slower per tile but not idle pro rata.
  static const int sk_env = getenv("LCR_GEMM_STREAMK") ? atoi(getenv("LCR_GEMM_STREAMK")) : 0;   // 0 off, 1 heuristic, 2 whenever legal
  const int sk_mode = g_force_streamk >= 0 ? g_force_streamk : sk_env;
  const bool sk_legal = !transA && !transB && K >= 64 && static_cast<int64_t>(div_up(M, 64)) * div_up(N, 64) <= SK_MAX_TILES;
  if (sk_mode && sk_legal && (sk_mode == 2 || sk_worth(M, N, K))) return launch_gemm_sk(A, B, C, M, N, K, ep, st);
  if (K >= 512 && b128 * nb128 >= 512) return launch_gemm<128, 128, 4, 1, true>(A, B, C, M, N, K, transA, transB, ep, st);
  return launch_gemm<64, 64, 2, 2, true>(A, B, C, M, N, K, transA, transB, ep, st);
}

// fp32 array -> its three bf16 terms, planes [3][n] (bf16 bit patterns); once per constant operand (weights)
extern "C" int lcr_split_bf16x3(const float* w, int64_t n, uint16_t* planes, void* stream) {
  if (!w || !planes || n < 0) {
    set_error("lcr_split_bf16x3: bad argument");
    return LCR_EARG;
  }
  if (n == 0) return LCR_OK;
  hipLaunchKernelGGL(k_split_bf16x3, dim3(min(div_up(n, 512), 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), w, n, planes);
  return check_launch("lcr_split_bf16x3");
}

// weights [N,K] (K % 32 == 0) -> the tiled planes lcr_gemm_f32_bsplit takes: u16[ceil(N/64)][K/32][3][64][32] (12 KB per block)
extern "C" int lcr_split_bf16x3_tiles(const float* w, int N, int K, uint16_t* tiles, void* stream) {
  if (!w || !tiles || N < 1 || K < GM_BK || K % GM_BK != 0 || reinterpret_cast<uintptr_t>(w) % 16 != 0 || reinterpret_cast<uintptr_t>(tiles) % 16 != 0) {
    set_error("lcr_split_bf16x3_tiles: needs K %% 32 == 0 and 16-byte aligned buffers");
    return LCR_EARG;
  }
  hipLaunchKernelGGL(k_split_bf16x3_tiles, dim3(div_up(N, 64) * (K / GM_BK)), dim3(256), 0, static_cast<hipStream_t>(stream), w, N, K, tiles);
  return check_launch("lcr_split_bf16x3_tiles");
}

// C = A[M,K] · B[N,K]^T with B given as the TILED bf16 planes of lcr_split_bf16x3_tiles; epilogue as lcr_gemm_f32.
extern "C" int lcr_gemm_f32_bsplit(const float* A, const uint16_t* Bs, float* C, int64_t M, int N, int K, const float* bias, const float* rowdiv,
                                   const int64_t* seg_len, int S, int groups, double* stats, void* stream) {
  if (!A || !Bs || !C || M < 0 || N <= 0 || K <= 0 || K % GM_BK != 0 || reinterpret_cast<uintptr_t>(A) % 16 != 0 || reinterpret_cast<uintptr_t>(Bs) % 16 != 0 ||
      K < GM_BK) {
    set_error("lcr_gemm_f32_bsplit: needs K %% 32 == 0 and 16-byte aligned operands");
    return LCR_EARG;
  }
  if (stats && (!seg_len || S < 1 || groups < 1 || N % groups != 0 || ((N / groups) & (N / groups - 1)) != 0)) {
    set_error("lcr_gemm_f32_bsplit: statistics need seg_len, S >= 1 and groups dividing N into power-of-two sized groups");
    return LCR_EARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  KernelTimerScope timed(KT_GEMM, st, M, N, K);
  if (M == 0) return LCR_OK;
  GemmEpilogue ep{bias, rowdiv, seg_len, S, groups, stats};
  const int mt8 = (div_up(M, 64) + 7) / 8 * 8;
  static const int abl = getenv("LCR_SPLIT_ABL") ? atoi(getenv("LCR_SPLIT_ABL")) : 0;      // timing ablations only (garbage results)
  const dim3 grid(mt8 * div_up(N, 64)), block(256);
  switch (abl) {
    case 1: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<1, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    case 2: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<2, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    case 4: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<4, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    case 8: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<8, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    case 16: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<16, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    case 23: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<23, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    case 30: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<30, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
    default: LCR_LAUNCH_TIMED((k_gemm_f32_bsplit_p<0, 2>), grid, block, 0, st, A, Bs, C, M, N, K, ep); break;
  }
  return check_launch("lcr_gemm_f32_bsplit");
}

// Batched C_z = A_z^T·B_z (transA) with per-entry K — NetVLAD's per-scan aggregation (NetVlad.py:68): one launch for S scans.
extern "C" int lcr_gemm_f32_batched_ta(const float* A, const float* B, float* C, int64_t M, int N, int count, const int* k_host,
                                       const int64_t* a_off_host, const int64_t* b_off_host, const int64_t* c_off_host, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || count < 1 || count > GM_MAX_BATCH || !k_host || !a_off_host || !b_off_host || !c_off_host) {
    set_error("lcr_gemm_f32_batched_ta: bad argument (count <= %d)", GM_MAX_BATCH);
    return LCR_EARG;
  }
  GemmBatch bt = {};
  bt.count = count;
  for (int i = 0; i < count; ++i) {
    bt.k[i] = k_host[i];
    bt.a_off[i] = a_off_host[i];
    bt.b_off[i] = b_off_host[i];
    bt.c_off[i] = c_off_host[i];
    if (a_off_host[i] % 4 || b_off_host[i] % 4 || k_host[i] < 1) {
      set_error("lcr_gemm_f32_batched_ta: offsets must be multiples of 4 floats and K >= 1");
      return LCR_EARG;
    }
  }
  if ((M % 4) || (N % 4)) {
    set_error("lcr_gemm_f32_batched_ta: M and N must be multiples of 4");
    return LCR_EARG;
  }
  GemmEpilogue ep{nullptr, nullptr, nullptr, 0, 0, nullptr};
  return launch_gemm<64, 64, 2, 2, true>(A, B, C, M, N, /*K (per entry)*/ 1, 1, 0, ep, static_cast<hipStream_t>(stream), &bt);
}

// Uniform batch: C_z[M,N] = op(A_z)·op(B_z), z < count (<= 65535), constant strides (in floats, multiples of 4) — the per-patch
// feature products of the dense matching stage (LCRNet.py:237-238: einsum('bnd,bmd->bnm')).
extern "C" int lcr_gemm_f32_strided_batched(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB,
                                            int64_t strideA, int64_t strideB, int64_t strideC, int count, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || count < 1 || count > 65535 || (strideA % 4) || (strideB % 4)) {
    set_error("lcr_gemm_f32_strided_batched: bad argument");
    return LCR_EARG;
  }
  const int64_t lda = transA ? M : K, ldb = transB ? K : N;
  if ((lda % 4) || (ldb % 4)) {
    set_error("lcr_gemm_f32_strided_batched: leading dimensions must be multiples of 4");
    return LCR_EARG;
  }
  GemmBatch bt = {};
  bt.count = count;
  bt.strided = 1;
  bt.k[0] = K;
  bt.a_off[0] = strideA;
  bt.b_off[0] = strideB;
  bt.c_off[0] = strideC;
  GemmEpilogue ep{nullptr, nullptr, nullptr, 0, 0, nullptr};
  // K <= 256 (the 128 x 128 x 256 patch products: thousands of two-by-two-tile problems of eight K-steps each): the light form — one
  // register stage, six to eight workgroups per CU — like the un-batched Linears of that depth (LCR_GEMM_BATCH_SHORT=0: the deep-pipeline form)
  // LCR_GEMM_BATCH_TILE=128: ONE 128 x 128 tile per problem instead of four 64 x 64 ones for the 128 x 128 x 256 patch products.  Looked like
  // +4 % pairs/s across two gpurun sessions (590 -> 616), is -3 % in a same-session A/B (648 / 644 vs 632 / 623 pairs/s at 16 pairs per call) and
  // 1.29 vs 0.96 ms per call alone: off by default, kept as the switch that measured it
  static const int batch_tile = getenv("LCR_GEMM_BATCH_TILE") ? atoi(getenv("LCR_GEMM_BATCH_TILE")) : 64;
  if (batch_tile == 128 && M >= 128 && N >= 128) return launch_gemm<128, 128, 4, 1, true>(A, B, C, M, N, K, transA, transB, ep, static_cast<hipStream_t>(stream), &bt);
  static const bool batch_short = !(getenv("LCR_GEMM_BATCH_SHORT") && atoi(getenv("LCR_GEMM_BATCH_SHORT")) == 0);
  if (batch_short && K <= 256 && !transA) return launch_gemm<64, 64, 2, 2, true, true>(A, B, C, M, N, K, transA, transB, ep, static_cast<hipStream_t>(stream), &bt);
  return launch_gemm<64, 64, 2, 2, true>(A, B, C, M, N, K, transA, transB, ep, static_cast<hipStream_t>(stream), &bt);
}
