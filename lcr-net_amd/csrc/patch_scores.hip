// patch_scores.hip — a-10, dense point matching: the patch score matrices of DenseMatchingHEAD (model_family/LCRNet.py:236-250) in ONE kernel.
//
// The reference gathers the K = 128 point features of every matched patch on both sides (index_select on the zero-padded feature
// tensor, :236-239), multiplies them (einsum 'bnd,bmd->bnm', :247), scales by 1/sqrt(C) (:248) and hands the (P,128,128) products to
// LearnableLogOptimalTransport, which pads them with the dustbin row / column alpha and masks invalid rows / columns to -inf
// (learnable_sinkhorn.py:38-49).  As separate launches that is two gathers writing 2 x P x 128 x C floats (1 GB each for a 16-pair call),
// a batched product reading them back, and a pass over the (P,129,129) scores.  Here one workgroup per patch pair gathers the rows
// straight from the point-feature tensor into LDS (a patch's 128 rows are re-used by neighbouring patches: L2 hits), multiplies on
// v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, k ascending: the arithmetic of lcr_gemm_f32) and writes the PADDED, MASKED, SCALED
// score matrix the transport kernel starts from.  The gathered feature copies never exist.
//
//   tile: 128 x 128 outputs per workgroup, 4 wavefronts x (64 x 64 = 2 x 2 MFMA tiles), K-steps of 32 channels, k-major LDS tiles
//   (stride 130 floats: the transposing 4-B stores of a wavefront — 8 rows x 8 channel quads — hit 64 different banks), double-buffered,
//   the next step's 16-B gathers in flight under the 64 MFMAs of the current one; 8 consecutive lanes fetch one row's full 128-B line.
#include "common.h"

namespace lcr {

typedef float ps_floatx16 __attribute__((ext_vector_type(16)));
constexpr int PS_K = 128;      // points per patch (cfg.model.num_points_in_patch)
constexpr int PS_BK = 32;      // channels per K-step
constexpr int PS_LD = 130;     // LDS row stride of the k-major tiles (floats)
constexpr int PS_T = 256;

__global__ __launch_bounds__(PS_T, 2) void k_patch_scores(const float* __restrict__ fa, int64_t Na, const float* __restrict__ fb, int64_t Nb, int C,
                                                          const int64_t* __restrict__ idx_a, const int64_t* __restrict__ idx_b,
                                                          const uint8_t* __restrict__ mask_a, const uint8_t* __restrict__ mask_b, float scale,
                                                          const float* __restrict__ alpha, float inf_val, float* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) float ps_lds[];      // 4 tiles of 32 x 130 floats = 66 560 B: dynamic (above the 64 KB static limit)
  constexpr int TILE = PS_BK * PS_LD;
  auto sA = [&](int buf) { return ps_lds + buf * TILE; };
  auto sB = [&](int buf) { return ps_lds + (2 + buf) * TILE; };
  const int64_t p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  // staging: piece q = tid + 256 i (i = 0..3) is (row q >> 3, channel quad q & 7): the 8 lanes of a row fetch one full line per K-step
  const int k4 = tid & 7;
  const float* pa[4];
  const float* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid + PS_T * i) >> 3;
    const int64_t ja = idx_a[p * PS_K + row], jb = idx_b[p * PS_K + row];
    pa[i] = (ja >= 0 && ja < Na) ? fa + ja * C + k4 * 4 : nullptr;       // the shadow index (== N): a zero row (index_select on the padded tensor)
    pb[i] = (jb >= 0 && jb < Nb) ? fb + jb * C + k4 * 4 : nullptr;
  }
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = pa[i] ? *reinterpret_cast<const float4*>(pa[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = pb[i] ? *reinterpret_cast<const float4*>(pb[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid + PS_T * i) >> 3;
      float* da = sA(buf) + (k4 * 4) * PS_LD + row;
      float* db = sB(buf) + (k4 * 4) * PS_LD + row;
      da[0] = ra[i].x, da[PS_LD] = ra[i].y, da[2 * PS_LD] = ra[i].z, da[3 * PS_LD] = ra[i].w;
      db[0] = rb[i].x, db[PS_LD] = rb[i].y, db[2 * PS_LD] = rb[i].z, db[3 * PS_LD] = rb[i].w;
    }
  };
  ps_floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = C / PS_BK;
  gload(0);
  park(0);
  __syncthreads();
  const int a_off = (lane >> 5) * PS_LD + wm * 64 + (lane & 31);
  const int b_off = (lane >> 5) * PS_LD + wn * 64 + (lane & 31);
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) gload((t + 1) * PS_BK);
    const float* as = sA(buf) + a_off;
    const float* bs = sB(buf) + b_off;
#pragma unroll
    for (int kk = 0; kk < PS_BK / 2; ++kk) {
      const float a0 = as[kk * 2 * PS_LD], a1 = as[kk * 2 * PS_LD + 32];
      const float b0 = bs[kk * 2 * PS_LD], b1 = bs[kk * 2 * PS_LD + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (t + 1 < nk) park(buf ^ 1);                       // its last readers passed the barrier of step t - 1
    __syncthreads();
  }
  // ---- the padded score matrix (learnable_sinkhorn.py:38-49): scaled products, dustbin row / column = alpha, masked entries = -inf
  const float al = alpha[0];
  float* Sp = S + p * (PS_K + 1) * (PS_K + 1);
  const uint8_t* ma = mask_a + p * PS_K;
  const uint8_t* mb = mask_b + p * PS_K;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int col = wn * 64 + tj * 32 + (lane & 31);
    const bool col_ok = mb[col] != 0;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bool ok = col_ok && ma[row] != 0;
        Sp[row * (PS_K + 1) + col] = ok ? acc[ti][tj][r] * scale : -inf_val;
      }
    }
  }
  if (tid <= PS_K) {
    const bool in = tid < PS_K;
    Sp[PS_K * (PS_K + 1) + tid] = (in && !mb[tid]) ? -inf_val : al;          // dustbin row: masked columns stay masked
    Sp[tid * (PS_K + 1) + PS_K] = (in && !ma[tid]) ? -inf_val : al;          // dustbin column (the corner is written twice with alpha)
  }
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_patch_scores(const float* feats_a, int64_t Na, const float* feats_b, int64_t Nb, int C, const int64_t* idx_a, const int64_t* idx_b,
                                const uint8_t* mask_a, const uint8_t* mask_b, int64_t P, int K, float scale, const float* alpha, float inf_val,
                                float* S, void* stream) {
  if (!feats_a || !feats_b || !idx_a || !idx_b || !mask_a || !mask_b || !alpha || !S || P < 0 || Na < 0 || Nb < 0 || K != PS_K || C < PS_BK ||
      C % PS_BK != 0 || (reinterpret_cast<uintptr_t>(feats_a) | reinterpret_cast<uintptr_t>(feats_b)) % 16 != 0 || P > 2147483647) {
    set_error("lcr_patch_scores: bad argument (patches of %d points, C a multiple of %d, 16-byte aligned features)", PS_K, PS_BK);
    return LCR_EARG;
  }
  if (P == 0) return LCR_OK;
  constexpr size_t lds = sizeof(float) * 4 * PS_BK * PS_LD;
  static DynLds opt_in;
  if (opt_in.need(reinterpret_cast<const void*>(&k_patch_scores), lds) != hipSuccess) {
    set_error("lcr_patch_scores: cannot reserve %zu B of dynamic LDS", lds);
    return LCR_EHIP;
  }
  hipLaunchKernelGGL(k_patch_scores, dim3(static_cast<unsigned>(P)), dim3(PS_T), lds, static_cast<hipStream_t>(stream), feats_a, Na, feats_b, Nb, C, idx_a,
                     idx_b, mask_a, mask_b, scale, alpha, inf_val, S);
  return check_launch("lcr_patch_scores");
}
