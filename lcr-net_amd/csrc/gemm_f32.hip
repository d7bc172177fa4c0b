// gemm_f32.hip — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TFLOP/s peak)
// with the epilogues the KPConv encoder needs fused in:  C = (A·B) [/ rowdiv[m]] [+ bias[n]],  plus per-(segment, group)
// sum / sum-of-squares accumulation for the GroupNorm that always follows (modules/kpconv/modules.py:33-50, 53-84).
//
// Used for: nn.Linear of UnaryBlock (B = weight^T, TB=1), the kernel-point contraction of KPConv
// (kpconv.py:108-110: (M, 15*C) x (15*C, Cout), TB=0), NetVLAD's assignment / aggregation GEMMs (TA=1 for x^T·a).
// fp32 in, fp32 accumulate: bf16 would break the 1e-4 descriptor tolerance.
//
// Tiling: 256 threads = 4 wavefronts stacked along M; block tile 128 x BN x 32; each wavefront owns 32 rows x BN columns
// = BN/32 accumulators of 16 VGPRs.  Operands are staged through LDS K-major (As[k][m], Bs[k][n]) so that an MFMA
// operand fetch is one conflict-free ds_read_b32 per lane; global loads are 16-B vectors, double-buffered in LDS with
// register prefetch (one barrier per K-step).
#include "common.h"

namespace lcr {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GM_BM = 128;
constexpr int GM_BK = 32;
constexpr int GM_T = 256;

struct GemmEpilogue {
  const float*   bias;      // [N] or null
  const float*   rowdiv;    // [M] or null: C[m][:] /= rowdiv[m] (before the bias), KPConv neighbour-count normalisation
  const int64_t* seg_len;   // [S] rows per GroupNorm segment (device) or null
  int            S;
  int            groups;    // GroupNorm groups over N
  double*        stats;     // [S, groups, 2] (sum, sumsq), accumulated atomically; null = no statistics
};

// rows x 32 tile of a row-major [rows_total x K] matrix -> S[k][r]   (transposing loader; src contiguous along k)
template <int R, int LD>
struct LoaderT {
  static constexpr int PIECES = R * 8 / GM_T;   // float4 pieces per thread
  float4 reg[PIECES];
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t rows_total, int K, int64_t r0, int k0, bool vec) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int row = f >> 3, c4 = f & 7;
      const int64_t gr = r0 + row;
      const int gk = k0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows_total) {
        const float* p = src + gr * K + gk;
        if (vec && gk + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk + 0 < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
      reg[j] = v;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ S) const {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int row = f >> 3, c4 = f & 7;
      S[(c4 * 4 + 0) * LD + row] = reg[j].x;
      S[(c4 * 4 + 1) * LD + row] = reg[j].y;
      S[(c4 * 4 + 2) * LD + row] = reg[j].z;
      S[(c4 * 4 + 3) * LD + row] = reg[j].w;
    }
  }
};

// 32 x W tile of a row-major [K x cols_total] matrix -> S[k][c]   (straight loader; src contiguous along c)
template <int W, int LD>
struct LoaderN {
  static constexpr int PIECES = 8 * W / GM_T;
  float4 reg[PIECES];
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t cols_total, int K, int64_t c0, int k0, bool vec) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int kr = f / (W / 4), c4 = f % (W / 4);
      const int gk = k0 + kr;
      const int64_t gc = c0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < K) {
        const float* p = src + static_cast<int64_t>(gk) * cols_total + gc;
        if (vec && gc + 3 < cols_total) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gc + 0 < cols_total) v.x = p[0];
          if (gc + 1 < cols_total) v.y = p[1];
          if (gc + 2 < cols_total) v.z = p[2];
          if (gc + 3 < cols_total) v.w = p[3];
        }
      }
      reg[j] = v;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ S) const {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int f = threadIdx.x + GM_T * j;
      const int kr = f / (W / 4), c4 = f % (W / 4);
      *reinterpret_cast<float4*>(&S[kr * LD + c4 * 4]) = reg[j];
    }
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

template <int BN, bool TA, bool TB>
__global__ __launch_bounds__(GM_T) void k_gemm_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    int64_t M, int N, int K, GemmEpilogue ep) {
  constexpr int LDA = TA ? GM_BM + 4 : GM_BM + 1;
  constexpr int LDB = TB ? BN + 1 : BN + 4;
  constexpr int NT = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[2][GM_BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][GM_BK * LDB];
  __shared__ double s_red[BN][2];

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * GM_BM;
  const int n0 = blockIdx.y * BN;

  // 16-byte vector loads need an aligned leading dimension (and base: torch allocations are 256-B aligned)
  const bool vecA = TA ? ((M & 3) == 0) : ((K & 3) == 0);
  const bool vecB = TB ? ((K & 3) == 0) : ((N & 3) == 0);

  LoaderT<GM_BM, LDA> la_t;
  LoaderN<GM_BM, LDA> la_n;
  LoaderT<BN, LDB>    lb_t;
  LoaderN<BN, LDB>    lb_n;

  auto gload = [&](int k0) {
    if (TA) la_n.load(A, M, K, m0, k0, vecA);
    else la_t.load(A, M, K, m0, k0, vecA);
    if (TB) lb_t.load(B, N, K, n0, k0, vecB);
    else lb_n.load(B, N, K, n0, k0, vecB);
  };
  auto sstore = [&](int buf) {
    if (TA) la_n.store(As[buf]);
    else la_t.store(As[buf]);
    if (TB) lb_t.store(Bs[buf]);
    else lb_n.store(Bs[buf]);
  };

  floatx16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nk = (K + GM_BK - 1) / GM_BK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int a_off = (lane >> 5) * LDA + w * 32 + (lane & 31);
  const int b_off = (lane >> 5) * LDB + (lane & 31);
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) gload((t + 1) * GM_BK);
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int kk = 0; kk < GM_BK / 2; ++kk) {
      const float a = as[kk * 2 * LDA + a_off];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float b = bs[kk * 2 * LDB + b_off + j * 32];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
    }
    if (t + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------
  const bool want_stats = ep.stats != nullptr;
  const int gs = want_stats ? N / ep.groups : 1;   // channels per group
  if (want_stats) {
    for (int i = threadIdx.x; i < BN * 2; i += GM_T) (&s_red[0][0])[i] = 0.0;
    __syncthreads();
  }
  const int64_t wrow0 = m0 + w * 32;
  int seg_first = 0, seg_last = 0;
  if (want_stats && wrow0 < M) {
    seg_first = seg_of_row(ep.seg_len, ep.S, wrow0);
    seg_last = seg_of_row(ep.seg_len, ep.S, min(wrow0 + 31, M - 1));
  }
  // all rows of the BLOCK in one segment?  (block-uniform decision so the LDS reduction below is valid)
  int blk_seg_first = 0, blk_seg_last = 0;
  if (want_stats) {
    blk_seg_first = seg_of_row(ep.seg_len, ep.S, m0);
    blk_seg_last = seg_of_row(ep.seg_len, ep.S, min(m0 + GM_BM - 1, M - 1));
  }
  const bool uniform_seg = blk_seg_first == blk_seg_last;
  (void)seg_first;
  (void)seg_last;

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + (lane & 31);
    const float bv = (ep.bias && col < N) ? ep.bias[col] : 0.f;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M && col < N) {
        float v = acc[j][r];
        if (ep.rowdiv) v = v / ep.rowdiv[row];
        v += bv;
        C[row * N + col] = v;
        if (want_stats) {
          if (uniform_seg) {
            s += v;
            ss = fmaf(v, v, ss);
          } else {
            const int sg = seg_of_row(ep.seg_len, ep.S, row);
            double* d = ep.stats + (static_cast<int64_t>(sg) * ep.groups + col / gs) * 2;
            atomicAdd(d, static_cast<double>(v));
            atomicAdd(d + 1, static_cast<double>(v) * static_cast<double>(v));
          }
        }
      }
    }
    if (want_stats && uniform_seg) {
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
        const int gl = (col - n0) / gs;   // group index local to this block's column range
        atomicAdd(&s_red[gl][0], ds);
        atomicAdd(&s_red[gl][1], dss);
      }
    }
  }
  if (want_stats && uniform_seg) {
    __syncthreads();
    const int ngl = (BN + gs - 1) / gs;
    for (int i = threadIdx.x; i < ngl; i += GM_T) {
      const int g = n0 / gs + i;
      if (g < ep.groups && (n0 + i * gs) < N) {
        double* d = ep.stats + (static_cast<int64_t>(blk_seg_first) * ep.groups + g) * 2;
        atomicAdd(d, s_red[i][0]);
        atomicAdd(d + 1, s_red[i][1]);
      }
    }
  }
}

template <int BN>
static int launch_gemm(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const GemmEpilogue& ep,
                       hipStream_t st) {
  dim3 grid(div_up(M, GM_BM), div_up(N, BN));
  dim3 block(GM_T);
  if (!transA && !transB) hipLaunchKernelGGL((k_gemm_f32<BN, false, false>), grid, block, 0, st, A, B, C, M, N, K, ep);
  else if (!transA && transB) hipLaunchKernelGGL((k_gemm_f32<BN, false, true>), grid, block, 0, st, A, B, C, M, N, K, ep);
  else if (transA && !transB) hipLaunchKernelGGL((k_gemm_f32<BN, true, false>), grid, block, 0, st, A, B, C, M, N, K, ep);
  else hipLaunchKernelGGL((k_gemm_f32<BN, true, true>), grid, block, 0, st, A, B, C, M, N, K, ep);
  return check_launch("lcr_gemm_f32");
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_gemm_f32(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const float* bias,
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
    if ((gs & (gs - 1)) != 0 || (gs < 32 && 32 % gs != 0) || (gs > 32 && gs % 32 != 0)) {
      set_error("lcr_gemm_f32: channels per group must be a power of two");
      return LCR_EARG;
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats) hipMemsetAsync(stats, 0, sizeof(double) * 2 * S * groups, st);
  if (M == 0) return LCR_OK;
  GemmEpilogue ep{bias, rowdiv, seg_len, S, groups, stats};
  if (N <= 32) return launch_gemm<32>(A, B, C, M, N, K, transA, transB, ep, st);
  if (N <= 64) return launch_gemm<64>(A, B, C, M, N, K, transA, transB, ep, st);
  return launch_gemm<128>(A, B, C, M, N, K, transA, transB, ep, st);
}
