// groupnorm.hip — a-5: segmented GroupNorm (+ LeakyReLU, + residual) applied from pre-reduced statistics.
//
// Reference: GroupNorm.forward, experiments/lcrnet/modules/kpconv/modules.py:33-50 — nn.GroupNorm(32, C) over (1, C, N):
// mean / biased variance per group over (C/32 channels) x (ALL points of the stack), eps 1e-5.  Scan-parallel batching
// needs the statistics to restart per segment (= what the reference stacked: one scan, or one pair); the
// sums are produced by lcr_gemm_f32's epilogue in fp64, this kernel turns them into mean / rstd and applies
//     y = act( GN(x) [+ res | + GN_res(res)] ),   act = LeakyReLU(slope) or identity,
// which covers UnaryBlock / ConvBlock (modules.py:78-84, 140-145) and the tail of ResidualBlock (:207-225) in one pass.
// Optionally emits pos[n] = (sum_c y[n][c] > 0), the flag KPConv's neighbour count is built from (kpconv.py:113-114).
#include <algorithm>

#include <type_traits>

#include "common.h"

namespace lcr {

struct GnSide {
  const double* stats;   // [S, groups, 2] or null (= identity, plain residual)
  const float*  gamma;
  const float*  beta;
};

__device__ __forceinline__ int seg_of(const int64_t* __restrict__ seg_len, int S, int64_t row, int64_t* len) {
  int s = 0;
  int64_t end = seg_len[0];
  while (s + 1 < S && row >= end) {
    ++s;
    end += seg_len[s];
  }
  *len = seg_len[s];
  return s;
}

// Flat float4 mapping: thread t handles 4 consecutive channels of one row (C % 4 == 0), so loads/stores are 16-B vectors and
// all 64 lanes are busy for every C.  Row flags (pos) are reduced across the C/4 lanes of a row (C <= 256).
constexpr int GN_TABLE = 2048;   // (segment, group) pairs whose mean / rstd fit the LDS table
constexpr int GN_MAX_SEG = 256;  // segments per call

// RM: residual mode, compile-time (0 none, 1 plain residual, 2 GroupNorm-ed residual): no per-element branches on it
template <bool POS, int RM>
__global__ __launch_bounds__(256) void k_gn_apply(const float* __restrict__ x, GnSide gx, const float* __restrict__ res, GnSide gr,
                                                  float* __restrict__ y, int64_t N, int C, int groups, const int64_t* __restrict__ seg_len, int S,
                                                  float eps, float slope, int act, uint8_t* __restrict__ pos) {
  // Every workgroup owns a CONTIGUOUS range of rows, so it touches one segment (two or three at scan boundaries) and only
  // folds the statistics replicas of those: (mean, rstd) per (segment, group) of the range, finalised in fp64 once per block.
  extern __shared__ __attribute__((aligned(16))) float2 s_tab[];   // [2][S * groups]: sized by the launch, 4 KB for 8 scans
  float2* s_x = s_tab;
  float2* s_r = s_tab + S * groups;
  __shared__ int64_t s_start[GN_MAX_SEG + 1];       // first row of every segment (prefix of seg_len)
  const int gs = C / groups;
  const int c4n = C >> 2;                           // float4 pieces per row
  const int64_t rows_per_blk = (N + gridDim.x - 1) / gridDim.x;
  const int64_t row_lo = static_cast<int64_t>(blockIdx.x) * rows_per_blk;
  const int64_t row_hi = min(row_lo + rows_per_blk, N);
  if (row_lo >= row_hi) return;
  if (threadIdx.x == 0) {
    int64_t o = 0;
    for (int i = 0; i < S; ++i) {
      s_start[i] = o;
      o += seg_len[i];
    }
    s_start[S] = o;
  }
  __syncthreads();
  int seg_lo = 0, seg_hi = 0;                        // segments of the first / last row (the last segment absorbs rows beyond the sum)
  while (seg_lo + 1 < S && row_lo >= s_start[seg_lo + 1]) ++seg_lo;
  seg_hi = seg_lo;
  while (seg_hi + 1 < S && row_hi - 1 >= s_start[seg_hi + 1]) ++seg_hi;
  const int nseg = seg_hi - seg_lo + 1;
  for (int i = threadIdx.x; i < nseg * groups; i += blockDim.x) {
    const int sg = seg_lo + i / groups, g = i - (i / groups) * groups;
    const double cnt = static_cast<double>(seg_len[sg]) * gs;
    double sx = 0.0, sxx = 0.0, rx = 0.0, rxx = 0.0;
    for (int rep = 0; rep < GN_REPLICAS; ++rep) {     // fold the statistics replicas (fixed order: deterministic given the sums)
      const int64_t o = ((static_cast<int64_t>(rep) * S + sg) * groups + g) * 2;
      sx += gx.stats[o];
      sxx += gx.stats[o + 1];
      if (RM == 2) {
        rx += gr.stats[o];
        rxx += gr.stats[o + 1];
      }
    }
    const double m = sx / cnt;
    const double var = fmax(sxx / cnt - m * m, 0.0);
    s_x[i] = make_float2(static_cast<float>(m), static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps))));
    if (RM == 2) {
      const double rm = rx / cnt;
      const double rv = fmax(rxx / cnt - rm * rm, 0.0);
      s_r[i] = make_float2(static_cast<float>(rm), static_cast<float>(1.0 / sqrt(rv + static_cast<double>(eps))));
    }
  }
  __syncthreads();
  const int64_t t_lo = row_lo * c4n, t_hi = row_hi * c4n;
  // One element = 4 consecutive channels of one row.  The per-channel parameters depend only on the channel offset, which is
  // the SAME in every iteration of a thread when the block size is a multiple of C/4 (every C this model uses): they are
  // loaded once; the row data of GU iterations is requested before any of it is used (the kernel is a pure stream).
  constexpr int GU = 4;
  const bool fixed_c = (blockDim.x % c4n) == 0;
  float4 gam_f = make_float4(0.f, 0.f, 0.f, 0.f), bet_f = gam_f, rgam_f = gam_f, rbet_f = gam_f;
  const int c0_f = static_cast<int>(threadIdx.x % c4n) * 4;
  if (fixed_c) {
    gam_f = *reinterpret_cast<const float4*>(gx.gamma + c0_f);
    bet_f = *reinterpret_cast<const float4*>(gx.beta + c0_f);
    if (RM == 2) {
      rgam_f = *reinterpret_cast<const float4*>(gr.gamma + c0_f);
      rbet_f = *reinterpret_cast<const float4*>(gr.beta + c0_f);
    }
  }
  const int64_t t_end = ((t_hi - t_lo + 63) & ~int64_t(63)) + t_lo;
  // FAST (block-uniform; every block of this model except the B - 1 that straddle a scan boundary): one segment in the range, C / 4 and the
  // group size powers of two, fixed channel offset — the row is a shift of the element index, and the thread's four (mean, rstd) pairs sit in
  // registers.  The general form spends ~170 VALU instructions per 16-B element on a 64-bit division (row = t / (C/4)), four 32-bit
  // divisions (group = channel / gs), the segment walk and eight LDS reads (PMC: 97 M of a step's 601 M VALU instructions for 2 % of its
  // flops); the arithmetic on the values is the same in both forms, operation for operation.
  const bool fast = fixed_c && nseg == 1 && (c4n & (c4n - 1)) == 0 && (gs & (gs - 1)) == 0 && N * C < (int64_t(1) << 30) && !(act & 256);   // act bit 8: tests force the general form
  act &= 255;
  const int sh4 = 31 - __builtin_clz(c4n), shg = 31 - __builtin_clz(gs);
  auto run = [&](auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    float2 mx[4], mrs[4];
    if (FAST) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {                  // (scale, shift) of this thread's four channels
        const int gi = (c0_f + u) >> shg;
        const float g = u == 0 ? gam_f.x : u == 1 ? gam_f.y : u == 2 ? gam_f.z : gam_f.w, b = u == 0 ? bet_f.x : u == 1 ? bet_f.y : u == 2 ? bet_f.z : bet_f.w;
        const float2 mr = s_x[gi];
        const float a = mr.y * g;
        mx[u] = make_float2(a, fmaf(-mr.x, a, b));
        mrs[u] = make_float2(1.f, 0.f);
        if (RM == 2) {
          const float rg = u == 0 ? rgam_f.x : u == 1 ? rgam_f.y : u == 2 ? rgam_f.z : rgam_f.w, rbv = u == 0 ? rbet_f.x : u == 1 ? rbet_f.y : u == 2 ? rbet_f.z : rbet_f.w;
          const float2 rr = s_r[gi];
          const float a2 = rr.y * rg;
          mrs[u] = make_float2(a2, fmaf(-rr.x, a2, rbv));
        }
      }
    }
    for (int64_t tb = t_lo + threadIdx.x; tb < t_end; tb += static_cast<int64_t>(GU) * blockDim.x) {
      float4 xv[GU], rv[GU];
      int64_t nrow[GU];
      typename std::conditional<FAST, unsigned, int64_t>::type eoff[GU];
      int c0s[GU];
      bool live[GU];
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int64_t t = tb + static_cast<int64_t>(k) * blockDim.x;
        live[k] = t < t_hi;
        if (FAST) {                                   // N * C < 2^30: element offsets are 32-bit (scalar base + vector offset addressing)
          const unsigned r32 = live[k] ? static_cast<unsigned>(t) >> sh4 : static_cast<unsigned>(row_hi - 1);
          nrow[k] = r32;
          c0s[k] = c0_f;
          eoff[k] = r32 * static_cast<unsigned>(C) + static_cast<unsigned>(c0_f);
        } else {
          nrow[k] = live[k] ? t / c4n : row_hi - 1;
          c0s[k] = live[k] ? static_cast<int>(t - nrow[k] * c4n) * 4 : 0;
          eoff[k] = nrow[k] * C + c0s[k];
        }
        xv[k] = *reinterpret_cast<const float4*>(x + eoff[k]);
        rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RM > 0) rv[k] = *reinterpret_cast<const float4*>(res + eoff[k]);
      }
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int64_t t = tb + static_cast<int64_t>(k) * blockDim.x;
        if (t >= t_end) break;                       // wave-uniform: t_end - t_lo is a multiple of 64
        const int64_t n = nrow[k];
        const int c0 = c0s[k];
        int s = seg_lo;
        if (!FAST) {
          while (s < seg_hi && n >= s_start[s + 1]) ++s;
        }
        float4 gam = gam_f, bet = bet_f, rgam = rgam_f, rbet = rbet_f;
        if (!FAST && !fixed_c) {
          gam = *reinterpret_cast<const float4*>(gx.gamma + c0);
          bet = *reinterpret_cast<const float4*>(gx.beta + c0);
          if (RM == 2) {
            rgam = *reinterpret_cast<const float4*>(gr.gamma + c0);
            rbet = *reinterpret_cast<const float4*>(gr.beta + c0);
          }
        }
        const float xin[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w}, g4[4] = {gam.x, gam.y, gam.z, gam.w}, b4[4] = {bet.x, bet.y, bet.z, bet.w};
        const float rin[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w}, rg4[4] = {rgam.x, rgam.y, rgam.z, rgam.w},
                    rb4[4] = {rbet.x, rbet.y, rbet.z, rbet.w};
        float out[4];
        float rowsum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // y = x * (rstd * gamma) + (beta - mean * rstd * gamma): the scale / shift form of lcr_gemm_f32_anorm's table (and of torch's
          // own GroupNorm kernels); FAST: the thread's four (scale, shift) pairs are loop constants
          float xa, xb, ra = 1.f, rb = 0.f;
          if (FAST) {
            xa = mx[u].x, xb = mx[u].y;
            ra = mrs[u].x, rb = mrs[u].y;
          } else {
            const int gi = (s - seg_lo) * groups + (c0 + u) / gs;
            const float2 mr = s_x[gi];
            xa = mr.y * g4[u];
            xb = fmaf(-mr.x, xa, b4[u]);
            if (RM == 2) {
              const float2 rr = s_r[gi];
              ra = rr.y * rg4[u];
              rb = fmaf(-rr.x, ra, rb4[u]);
            }
          }
          float v = fmaf(xin[u], xa, xb);
          if (RM == 2) v += fmaf(rin[u], ra, rb);
          else if (RM == 1) v += rin[u];
          if (act) v = v > 0.f ? v : v * slope;
          out[u] = v;
          rowsum += v;
        }
        if (live[k]) *reinterpret_cast<float4*>(y + eoff[k]) = make_float4(out[0], out[1], out[2], out[3]);
        if (POS) {
          // the c4n (<= 64, power of two) lanes of a row are consecutive and aligned inside the wavefront
          for (int d = 1; d < c4n; d <<= 1) rowsum += __shfl_xor(rowsum, d);
          if (live[k] && c0 == 0) pos[n] = rowsum > 0.f ? 1 : 0;
        }
      }
    }
  };
  if (fast) run(std::true_type{});
  else run(std::false_type{});
}

// plain segmented statistics for tensors that do not come out of lcr_gemm_f32 (e.g. the fused C_in = 1 KPConv):
// a wavefront walks 64 consecutive rows with lanes = channels (coalesced rows), folds the lanes of a group and issues one
// pair of fp64 atomics per (segment, group, wavefront).
__global__ __launch_bounds__(256) void k_gn_stats(const float* __restrict__ x, int64_t N, int C, int groups, const int64_t* __restrict__ seg_len,
                                                  int S, double* __restrict__ stats) {
  const int gs = C / groups;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * 4 + w) * 64;
  const int64_t r1 = r0 + 64 < N ? r0 + 64 : N;
  if (r0 >= N) return;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    float s = 0.f, ss = 0.f;
    int64_t slen;
    int cur = seg_of(seg_len, S, r0, &slen);
    int64_t seg_end = 0;
    for (int i = 0; i <= cur; ++i) seg_end += seg_len[i];
    auto flush = [&](int sg) {
      double ds = s, dss = ss;
      const int span = gs < 64 ? gs : 64;
      for (int d = 1; d < span; d <<= 1) {
        ds += __shfl_xor(ds, d);
        dss += __shfl_xor(dss, d);
      }
      if (c < C && (lane & (span - 1)) == 0) {
        double* rep = stats + static_cast<int64_t>(blockIdx.x % GN_REPLICAS) * S * groups * 2;
        atomicAdd(&rep[(static_cast<int64_t>(sg) * groups + c / gs) * 2], ds);
        atomicAdd(&rep[(static_cast<int64_t>(sg) * groups + c / gs) * 2 + 1], dss);
      }
      s = ss = 0.f;
    };
    constexpr int RB = 8;                       // rows requested per trip (clamped addresses, masked afterwards: branch-free loads)
    const int cc = c < C ? c : C - 1;
    for (int64_t n0 = r0; n0 < r1; n0 += RB) {
      float vv[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int64_t nn = n0 + u < r1 ? n0 + u : r1 - 1;
        vv[u] = x[nn * C + cc];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int64_t n = n0 + u;
        if (n < r1) {                           // wave-uniform
          while (n >= seg_end && cur + 1 < S) {
            flush(cur);
            ++cur;
            seg_end += seg_len[cur];
          }
          const float v = c < C ? vv[u] : 0.f;
          s += v;
          ss = fmaf(v, v, ss);
        }
      }
    }
    flush(cur);
  }
}

}  // namespace lcr

using namespace lcr;

static int g_gn_force_general = 0;          // tests: every block takes the general (division) form
extern "C" void lcr_groupnorm_debug_general(int on) { g_gn_force_general = on; }

extern "C" int lcr_groupnorm_apply(const float* x, const double* stats, const float* gamma, const float* beta, const float* res,
                                   const double* res_stats, const float* res_gamma, const float* res_beta, float* y, int64_t N, int C,
                                   int groups, const int64_t* seg_len, int S, float eps, float slope, int act, uint8_t* pos, void* stream) {
  if (!x || !stats || !gamma || !beta || !y || !seg_len || N < 0 || C < 1 || groups < 1 || C % groups != 0 || S < 1 ||
      (res_stats && (!res || !res_gamma || !res_beta))) {
    set_error("lcr_groupnorm_apply: bad argument");
    return LCR_EARG;
  }
  if (N == 0) return LCR_OK;
  GnSide gx{stats, gamma, beta}, gr{res_stats, res_gamma, res_beta};
  if (C % 4 != 0 || S * groups > GN_TABLE || S > GN_MAX_SEG) {
    set_error("lcr_groupnorm_apply: C must be a multiple of 4, S*groups <= %d and S <= %d", GN_TABLE, GN_MAX_SEG);
    return LCR_EARG;
  }
  const int c4n = C / 4;
  if (pos && (c4n > 64 || (c4n & (c4n - 1)) != 0)) {
    set_error("lcr_groupnorm_apply: row flags need C/4 to be a power of two <= 64");
    return LCR_EARG;
  }
  // contiguous row ranges per workgroup, >= 2048 float4 pieces each (the per-block statistics fold is amortised over them)
  const int nblk = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((N * c4n + 2047) / 2048, 256 * 8)));
  const size_t tab_bytes = sizeof(float2) * 2 * static_cast<size_t>(S) * groups;
  const int rm = res ? (res_stats ? 2 : 1) : 0;
  act = (act & 255) | (g_gn_force_general ? 256 : 0);
#define LCR_GN(P, R)                                                                                                                       \
  hipLaunchKernelGGL((k_gn_apply<P, R>), dim3(nblk), dim3(256), tab_bytes, static_cast<hipStream_t>(stream), x, gx, res, gr, y, N, C, groups, \
                     seg_len, S, eps, slope, act, pos)
  if (pos) {
    if (rm == 2) LCR_GN(true, 2);
    else if (rm == 1) LCR_GN(true, 1);
    else LCR_GN(true, 0);
  } else {
    if (rm == 2) LCR_GN(false, 2);
    else if (rm == 1) LCR_GN(false, 1);
    else LCR_GN(false, 0);
  }
#undef LCR_GN
  return check_launch("lcr_groupnorm_apply");
}

extern "C" int lcr_groupnorm_stats(const float* x, int64_t N, int C, int groups, const int64_t* seg_len, int S, double* stats, void* stream) {
  if (!x || !stats || !seg_len || N < 0 || C < 1 || groups < 1 || C % groups != 0 || S < 1) {
    set_error("lcr_groupnorm_stats: bad argument");
    return LCR_EARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // `stats` accumulates into a caller-zeroed table, like lcr_gemm_f32
  if (N == 0) return LCR_OK;
  hipLaunchKernelGGL(k_gn_stats, dim3(static_cast<int>((N + 255) / 256)), dim3(256), 0, st, x, N, C, groups, seg_len, S, stats);   // 4 waves x 64 rows
  return check_launch("lcr_groupnorm_stats");
}
