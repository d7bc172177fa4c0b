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

// one wavefront per row; lanes stride the channels
__global__ __launch_bounds__(256) void k_gn_apply(const float* __restrict__ x, GnSide gx, const float* __restrict__ res, GnSide gr,
                                                  float* __restrict__ y, int64_t N, int C, int groups, const int64_t* __restrict__ seg_len, int S,
                                                  float eps, float slope, int act, uint8_t* __restrict__ pos) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gs = C / groups;
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * 4 + w; n < N; n += static_cast<int64_t>(gridDim.x) * 4) {
    int64_t slen;
    const int s = seg_of(seg_len, S, n, &slen);
    const double cnt = static_cast<double>(slen) * gs;
    float rowsum = 0.f;
    for (int c = lane; c < C; c += 64) {
      const int g = c / gs;
      const double* st = gx.stats + (static_cast<int64_t>(s) * groups + g) * 2;
      const double mean = st[0] / cnt;
      const double var = fmax(st[1] / cnt - mean * mean, 0.0);
      const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
      float v = (x[n * C + c] - static_cast<float>(mean)) * rstd * gx.gamma[c] + gx.beta[c];
      if (res) {
        float r = res[n * C + c];
        if (gr.stats) {
          const double* rt = gr.stats + (static_cast<int64_t>(s) * groups + g) * 2;
          const double rm = rt[0] / cnt;
          const double rv = fmax(rt[1] / cnt - rm * rm, 0.0);
          r = (r - static_cast<float>(rm)) * static_cast<float>(1.0 / sqrt(rv + static_cast<double>(eps))) * gr.gamma[c] + gr.beta[c];
        }
        v += r;
      }
      if (act) v = v > 0.f ? v : v * slope;
      y[n * C + c] = v;
      rowsum += v;
    }
    if (pos) {
      rowsum = wave_sum(rowsum);
      if (lane == 0) pos[n] = rowsum > 0.f ? 1 : 0;
    }
  }
}

// plain segmented statistics for tensors that do not come out of lcr_gemm_f32 (e.g. the fused C_in = 1 KPConv)
__global__ __launch_bounds__(256) void k_gn_stats(const float* __restrict__ x, int64_t N, int C, int groups, const int64_t* __restrict__ seg_len,
                                                  int S, double* __restrict__ stats) {
  // block = 256 consecutive rows; thread t owns channels t, t+256, ...
  const int gs = C / groups;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 256;
  const int64_t r1 = r0 + 256 < N ? r0 + 256 : N;
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, ss = 0.0;
    int cur = -1;
    for (int64_t n = r0; n < r1; ++n) {
      int64_t slen;
      const int sg = seg_of(seg_len, S, n, &slen);
      if (sg != cur) {
        if (cur >= 0) {
          atomicAdd(&stats[(static_cast<int64_t>(cur) * groups + c / gs) * 2], s);
          atomicAdd(&stats[(static_cast<int64_t>(cur) * groups + c / gs) * 2 + 1], ss);
        }
        s = ss = 0.0;
        cur = sg;
      }
      const double v = x[n * C + c];
      s += v;
      ss += v * v;
    }
    if (cur >= 0) {
      atomicAdd(&stats[(static_cast<int64_t>(cur) * groups + c / gs) * 2], s);
      atomicAdd(&stats[(static_cast<int64_t>(cur) * groups + c / gs) * 2 + 1], ss);
    }
  }
}

}  // namespace lcr

using namespace lcr;

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
  const int nblk = static_cast<int>(std::min<int64_t>((N + 3) / 4, 256 * 16));
  hipLaunchKernelGGL(k_gn_apply, dim3(nblk), dim3(256), 0, static_cast<hipStream_t>(stream), x, gx, res, gr, y, N, C, groups, seg_len, S, eps,
                     slope, act, pos);
  return check_launch("lcr_groupnorm_apply");
}

extern "C" int lcr_groupnorm_stats(const float* x, int64_t N, int C, int groups, const int64_t* seg_len, int S, double* stats, void* stream) {
  if (!x || !stats || !seg_len || N < 0 || C < 1 || groups < 1 || C % groups != 0 || S < 1) {
    set_error("lcr_groupnorm_stats: bad argument");
    return LCR_EARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipMemsetAsync(stats, 0, sizeof(double) * 2 * S * groups, st);
  if (N == 0) return LCR_OK;
  hipLaunchKernelGGL(k_gn_stats, dim3(static_cast<int>((N + 255) / 256)), dim3(256), 0, st, x, N, C, groups, seg_len, S, stats);
  return check_launch("lcr_groupnorm_stats");
}
