// netvlad.hip — a-7: global-descriptor head (NetVLAD + context gating) for a batch of scans, fp32.
//
// Reference: GlobalDescritionHEAD (model_family/LCRNet_GlobalDescrition.py:34-38, LCRNet.py:115-122) =
//   F.normalize(feats_c, dim=2) -> NetVLADLoupe2.forward (modules/netvlad/NetVlad.py:49-87, eval: BatchNorm1d uses running
//   statistics) -> GatingContext (:165-201) -> F.normalize(dim=1).
// The reference runs it with batch 1 (one scan per stack); here S scans of a batch are stacked (seg_len rows each) and every
// step is segment-aware, so the 67 MB hidden1_weights matrix — the only HBM-significant operand — is streamed ONCE per batch
// by a split-K kernel (256 K-slices x all segments), instead of once per scan.
//
// Steps: row L2-normalise (wave per row) -> assignment GEMM (lcr_gemm_f32, K=1024, N=64) -> BN + softmax over the 64 clusters
// (lane = cluster) -> per-segment x^T·a GEMM (TA) and column sums -> residual / intra-normalise / global normalise ->
// split-K hidden projection -> BN2, gating GEMV, sigmoid, final L2 normalise (one workgroup per scan).
#include <algorithm>
#include <vector>

#include "common.h"

extern "C" int lcr_gemm_f32(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const float* bias,
                            const float* rowdiv, const int64_t* seg_len, int S, int groups, double* stats, void* stream);
extern "C" int lcr_gemm_f32_batched_ta(const float* A, const float* B, float* C, int64_t M, int N, int count, const int* k_host,
                                       const int64_t* a_off_host, const int64_t* b_off_host, const int64_t* c_off_host, void* stream);

namespace lcr {

constexpr int NV_F = 1024;   // feature size
constexpr int NV_K = 64;     // clusters
constexpr int NV_D = 256;    // output dim
constexpr int NV_SPLIT = 256;   // K-slices of the hidden projection (65536 / 256 = 256 rows each)

struct BnParams {
  const float *w, *b, *mean, *var;
};

__global__ __launch_bounds__(256) void k_row_l2norm(const float* __restrict__ x, int64_t N, int C, float eps, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * 4 + w; n < N; n += static_cast<int64_t>(gridDim.x) * 4) {
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float v = x[n * C + c];
      ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    const float d = fmaxf(sqrtf(ss), eps);
    for (int c = lane; c < C; c += 64) y[n * C + c] = x[n * C + c] / d;
  }
}

// act[n][k] <- softmax_k( BN_eval(act[n][k]) ), lane = cluster (64 clusters = one wavefront)
__global__ __launch_bounds__(256) void k_bn_softmax64(float* __restrict__ act, int64_t N, BnParams bn, float eps) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const float scale = bn.w[lane] / sqrtf(bn.var[lane] + eps);
  const float mean = bn.mean[lane], beta = bn.b[lane];
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * 4 + w; n < N; n += static_cast<int64_t>(gridDim.x) * 4) {
    const float v = (act[n * NV_K + lane] - mean) * scale + beta;
    float mx = v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    const float e = expf(v - mx);
    const float s = wave_sum(e);
    act[n * NV_K + lane] = e / s;
  }
}

struct SegRows {
  int64_t off[65];   // row offsets of up to 64 segments (+ total)
};

// a_sum[s][k] = sum over the rows of segment s = blockIdx.x (deterministic: fixed row partition + fixed combine order)
__global__ __launch_bounds__(1024) void k_colsum64(const float* __restrict__ act, SegRows seg, float* __restrict__ out) {
  __shared__ float part[16][NV_K];
  const int k = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t row0 = seg.off[blockIdx.x], rows = seg.off[blockIdx.x + 1] - row0;
  float s = 0.f;
  for (int64_t n = q; n < rows; n += 16) s += act[(row0 + n) * NV_K + k];
  part[q][k] = s;
  __syncthreads();
  if (q == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][k];
    out[blockIdx.x * NV_K + k] = t;
  }
}

// V[c][k] (1024 x 64 per segment): subtract a_sum * Wc2, normalise each cluster column over c (eps 1e-6), then the whole
// 65536-vector (eps 1e-6); output flattened c-major (index c*64 + k) = vlad.view(-1, 65536) of NetVlad.py:73-75.
__global__ __launch_bounds__(1024) void k_vlad_finalize(float* __restrict__ V, const float* __restrict__ a_sum, const float* __restrict__ Wc2) {
  __shared__ float s_col[16][NV_K];
  __shared__ float s_inv[NV_K];
  __shared__ float s_tot;
  float* v = V + static_cast<int64_t>(blockIdx.x) * NV_F * NV_K;
  const float* as = a_sum + blockIdx.x * NV_K;
  const int k = threadIdx.x & 63, q = threadIdx.x >> 6;   // 16 row-groups x 64 clusters
  const float ak = as[k];
  float ss = 0.f;
  constexpr int FB = 8;                           // rows requested per trip: one workgroup per scan, latency is all there is to hide
  static_assert((NV_F / 16) % FB == 0, "row batches");
  for (int c0 = q; c0 < NV_F; c0 += 16 * FB) {
    float vv[FB], ww[FB];
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      vv[u] = v[(c0 + 16 * u) * NV_K + k];
      ww[u] = Wc2[(c0 + 16 * u) * NV_K + k];
    }
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      const float t = vv[u] - ak * ww[u];
      v[(c0 + 16 * u) * NV_K + k] = t;
      ss = fmaf(t, t, ss);
    }
  }
  s_col[q][k] = ss;
  __syncthreads();
  if (q == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s_col[i][k];
    const float nrm = sqrtf(t);
    const float inv = 1.f / fmaxf(nrm, 1e-6f);
    s_inv[k] = inv;
    const float u = nrm * inv;   // norm of the normalised column
    float tot = u * u;
    tot = wave_sum(tot);
    if (k == 0) s_tot = tot;
  }
  __syncthreads();
  const float g = 1.f / fmaxf(sqrtf(s_tot), 1e-6f);
  const float f = s_inv[k] * g;
  for (int c0 = q; c0 < NV_F; c0 += 16 * FB) {
    float vv[FB];
#pragma unroll
    for (int u = 0; u < FB; ++u) vv[u] = v[(c0 + 16 * u) * NV_K + k];
#pragma unroll
    for (int u = 0; u < FB; ++u) v[(c0 + 16 * u) * NV_K + k] = vv[u] * f;
  }
}

// partial[slice][s][j] = sum_{i in slice} v[s][i] * H[i][j];  one workgroup per K-slice, thread = output column j
template <int SMAX>
__global__ __launch_bounds__(NV_D) void k_hidden_splitk(const float* __restrict__ V, const float* __restrict__ H, int S, int s0,
                                                         float* __restrict__ partial) {
  constexpr int ROWS = NV_F * NV_K / NV_SPLIT;   // 256
  __shared__ float s_v[SMAX][ROWS];
  const int j = threadIdx.x;
  const int i0 = blockIdx.x * ROWS;
  const int ns = min(SMAX, S - s0);
  for (int s = 0; s < ns; ++s) s_v[s][j] = V[static_cast<int64_t>(s0 + s) * NV_F * NV_K + i0 + j];
  __syncthreads();
  float acc[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) acc[s] = 0.f;
  // the 67 MB weight matrix is streamed once: 16 rows requested per trip (one wavefront per SIMD has nothing else to hide the
  // latency with); the additions keep the row order
  constexpr int HB = 16;
  static_assert(ROWS % HB == 0, "row batches");
  const float* hp = H + static_cast<int64_t>(i0) * NV_D + j;
  for (int i = 0; i < ROWS; i += HB) {
    float h[HB];
#pragma unroll
    for (int u = 0; u < HB; ++u) h[u] = hp[static_cast<int64_t>(i + u) * NV_D];
#pragma unroll
    for (int u = 0; u < HB; ++u)
#pragma unroll
      for (int s = 0; s < SMAX; ++s) acc[s] = fmaf(s_v[s][i + u], h[u], acc[s]);
  }
  for (int s = 0; s < ns; ++s) partial[(static_cast<int64_t>(blockIdx.x) * S + s0 + s) * NV_D + j] = acc[s];
}

__device__ __forceinline__ float block256_sum(float v, float* lds) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  __syncthreads();
  return t;
}

// one workgroup per scan: reduce the K-slices, BN2, context gating (NetVlad.py:187-199), final F.normalize(dim=1)
__global__ __launch_bounds__(NV_D) void k_netvlad_tail(const float* __restrict__ partial, int S, BnParams bn2, const float* __restrict__ Wg,
                                                        BnParams bng, float eps, float* __restrict__ out) {
  __shared__ float s_o[NV_D];
  __shared__ float s_red[4];
  const int s = blockIdx.x, j = threadIdx.x;
  float o = 0.f;
  constexpr int TB = 16;                          // loads requested per trip (8 workgroups: nothing else hides the latency)
  static_assert(NV_SPLIT % TB == 0 && NV_D % TB == 0, "load batches");
  for (int b = 0; b < NV_SPLIT; b += TB) {
    float p[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) p[u] = partial[(static_cast<int64_t>(b + u) * S + s) * NV_D + j];
#pragma unroll
    for (int u = 0; u < TB; ++u) o += p[u];
  }
  o = (o - bn2.mean[j]) / sqrtf(bn2.var[j] + eps) * bn2.w[j] + bn2.b[j];
  s_o[j] = o;
  __syncthreads();
  float g = 0.f;
  for (int i = 0; i < NV_D; i += TB) {
    float wv[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) wv[u] = Wg[(i + u) * NV_D + j];
#pragma unroll
    for (int u = 0; u < TB; ++u) g = fmaf(s_o[i + u], wv[u], g);
  }
  g = (g - bng.mean[j]) / sqrtf(bng.var[j] + eps) * bng.w[j] + bng.b[j];
  g = 1.f / (1.f + expf(-g));
  const float a = o * g;
  const float tot = block256_sum(a * a, s_red);
  out[s * NV_D + j] = a / fmaxf(sqrtf(tot), 1e-12f);
}

}  // namespace lcr

using namespace lcr;



extern "C" int lcr_netvlad_ws_bytes(int64_t n_rows, int S, size_t* bytes) {
  if (!bytes || n_rows < 0 || S < 1) return LCR_EARG;
  Carver c(nullptr, ~size_t(0));
  c.take<float>(static_cast<size_t>(n_rows > 0 ? n_rows : 1) * NV_F);
  c.take<float>(static_cast<size_t>(n_rows > 0 ? n_rows : 1) * NV_K);
  c.take<float>(static_cast<size_t>(S) * NV_F * NV_K);
  c.take<float>(static_cast<size_t>(S) * NV_K);
  c.take<float>(static_cast<size_t>(NV_SPLIT) * S * NV_D);
  *bytes = c.off;
  return LCR_OK;
}

// feats [sum(seg_len), 1024] (coarse node features, stacked) -> out [S, 256] unit-norm descriptors
extern "C" int lcr_netvlad_forward(const float* feats, const int64_t* seg_len_host, int S, const LcrNetvladWeights* wt, float* out,
                                   void* ws, size_t ws_bytes, void* stream) {
  if (!feats || !seg_len_host || !wt || !out || !ws || S < 1) {
    set_error("lcr_netvlad_forward: bad argument");
    return LCR_EARG;
  }
  int64_t N = 0;
  for (int s = 0; s < S; ++s) {
    if (seg_len_host[s] <= 0) {
      set_error("lcr_netvlad_forward: empty segment %d", s);
      return LCR_EARG;
    }
    N += seg_len_host[s];
  }
  size_t need = 0;
  lcr_netvlad_ws_bytes(N, S, &need);
  if (need > ws_bytes) {
    set_error("lcr_netvlad_forward: workspace too small (%zu < %zu)", ws_bytes, need);
    return LCR_ESPACE;
  }
  Carver c(ws, ws_bytes);
  float* xn = c.take<float>(static_cast<size_t>(N) * NV_F);
  float* act = c.take<float>(static_cast<size_t>(N) * NV_K);
  float* V = c.take<float>(static_cast<size_t>(S) * NV_F * NV_K);
  float* asum = c.take<float>(static_cast<size_t>(S) * NV_K);
  float* partial = c.take<float>(static_cast<size_t>(NV_SPLIT) * S * NV_D);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float bn_eps = 1e-5f;
  const int nblk = static_cast<int>(std::min<int64_t>((N + 3) / 4, 256 * 16));
  hipLaunchKernelGGL(k_row_l2norm, dim3(nblk), dim3(256), 0, st, feats, N, NV_F, 1e-12f, xn);
  int rc = lcr_gemm_f32(xn, wt->cluster_weights, act, N, NV_K, NV_F, 0, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bn_softmax64, dim3(nblk), dim3(256), 0, st, act, N, BnParams{wt->bn1_w, wt->bn1_b, wt->bn1_mean, wt->bn1_var}, bn_eps);
  // V_s (1024 x 64) = xn_s^T (1024 x n_s) · act_s (n_s x 64) for all scans, 64 scans per launch: A is stored K-major -> transA
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int cnt = std::min(64, S - s0);
    int kk[64];
    int64_t ao[64], bo[64], co[64];
    SegRows seg;
    int64_t r0 = 0;
    for (int s = 0; s < s0; ++s) r0 += seg_len_host[s];
    for (int i = 0; i < cnt; ++i) {
      const int64_t n = seg_len_host[s0 + i];
      if (n > 2147483647LL) return LCR_EARG;
      kk[i] = static_cast<int>(n);
      ao[i] = r0 * NV_F;
      bo[i] = r0 * NV_K;
      co[i] = static_cast<int64_t>(s0 + i) * NV_F * NV_K;
      seg.off[i] = r0;
      r0 += n;
    }
    seg.off[cnt] = r0;
    rc = lcr_gemm_f32_batched_ta(xn, act, V, NV_F, NV_K, cnt, kk, ao, bo, co, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_colsum64, dim3(cnt), dim3(1024), 0, st, act, seg, asum + s0 * NV_K);
  }
  hipLaunchKernelGGL(k_vlad_finalize, dim3(S), dim3(1024), 0, st, V, asum, wt->cluster_weights2);
  for (int s0 = 0; s0 < S; s0 += 8)
    hipLaunchKernelGGL((k_hidden_splitk<8>), dim3(NV_SPLIT), dim3(NV_D), 0, st, V, wt->hidden1_weights, S, s0, partial);
  hipLaunchKernelGGL(k_netvlad_tail, dim3(S), dim3(NV_D), 0, st, partial, S, BnParams{wt->bn2_w, wt->bn2_b, wt->bn2_mean, wt->bn2_var},
                     wt->gating_weights, BnParams{wt->gbn_w, wt->gbn_b, wt->gbn_mean, wt->gbn_var}, bn_eps, out);
  return check_launch("lcr_netvlad_forward");
}
