// attention.hip — a-8: the 3D-RoFormer attention block (rotary self-attention / vanilla cross-attention), fp32.
//
// Reference: RPEMultiHeadAttention.forward (experiments/lcrnet/modules/thdroformer/rpetransformer.py:77-108),
// RotaryPositionalEmbedding.forward (:45-54), dynamic_attention with k=None (:19-39: full softmax),
// MultiHeadAttention.forward (vanilla_transformer.py:46-85), AttentionLayer / AttentionOutput residual LayerNorms (:13-28, 88-118).
// The reference materialises (1,4,N,M) score tensors and runs ~15 small torch ops per layer; here:
//   k_rotary     : q,k <- q·cos(theta) + rot(q)·sin(theta) in place, theta = learned 3-D position code (one angle per
//                  adjacent channel pair; the SAME theta rotates q and k), one pass, sincosf once per pair;
//   k_attention  : flash-style fused QK^T / softmax / PV on the fp32 matrix cores — the only dense QK^T·V on the path, so
//                  this is where MFMA is spent (v_mfma_f32_32x32x2_f32; fp32 inputs are required for the 1e-4 tolerance).
//                  One workgroup of four wavefronts per (head, 32-query tile), the key blocks dealt round-robin to the wavefronts and
//                  the partial softmax states merged through LDS (round 3: 72 -> 26 us per launch for one pair).  Scores are computed TRANSPOSED (S^T = K·Q^T) so that a lane
//                  owns one query column: its running max / sum / rescale are per-lane scalars, the 32 keys of a tile sit
//                  in the lane's 16 accumulator registers + its partner lane's (lane ^ 32), and P feeds the second MFMA
//                  (O^T = V^T·P^T) straight from registers with one cross-half swap per step — no LDS round trip for P,
//                  no score matrix in memory;
//   k_add_layernorm : y = LayerNorm(a + b) (d_model = 128: one wavefront per row).
#include <algorithm>
#include <cmath>

#include "common.h"

namespace lcr {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int AT_D = 32;    // head dim (d_model 128 / 4 heads)
constexpr int AT_LD = 33;   // LDS row stride: conflict-free ds_read_b32 for both "row = lane" operand patterns

// x [N, H*32] in place; theta [N, H*16]
__global__ __launch_bounds__(256) void k_rotary(float* __restrict__ x, const float* __restrict__ theta, int64_t N, int heads) {
  const int pairs = heads * 16;
  const int64_t total = N * pairs;
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < total; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t n = t / pairs;
    const int p = static_cast<int>(t - n * pairs);
    float s, c;
    sincosf(theta[n * pairs + p], &s, &c);
    float2* q = reinterpret_cast<float2*>(x + n * (pairs * 2) + 2 * p);
    const float2 v = *q;
    *q = make_float2(v.x * c - v.y * s, v.y * c + v.x * s);
  }
}

// register-staged variant: global -> registers (issued one tile ahead), registers -> LDS
__device__ __forceinline__ void tile_to_regs(const float* __restrict__ src, int64_t rows, int64_t r0, int ld_src, int col0, float4 (&reg)[4]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = lane + 64 * i;
    const int r = f >> 3, c4 = f & 7;
    const int64_t rr = r0 + r < rows ? r0 + r : rows - 1;          // clamped, branch-free; rows past the end are masked by the caller
    reg[i] = *reinterpret_cast<const float4*>(src + rr * ld_src + col0 + c4 * 4);
  }
}
__device__ __forceinline__ void regs_to_lds(const float4 (&reg)[4], float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = lane + 64 * i;
    const int r = f >> 3, c4 = f & 7;
    float* d = dst + r * AT_LD + c4 * 4;
    d[0] = reg[i].x;
    d[1] = reg[i].y;
    d[2] = reg[i].z;
    d[3] = reg[i].w;
  }
}

__device__ __forceinline__ void load_tile(const float* __restrict__ src, int64_t rows, int64_t r0, int ld_src, int col0, float* __restrict__ dst) {
  // 32 x 32 tile (rows r0.., columns col0..col0+31) of a row-major matrix -> dst[32][AT_LD]; rows beyond `rows` -> 0
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = lane + 64 * i;          // 256 float4 pieces
    const int r = f >> 3, c4 = f & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows) v = *reinterpret_cast<const float4*>(src + (r0 + r) * ld_src + col0 + c4 * 4);
    float* d = dst + r * AT_LD + c4 * 4;
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  }
}

// one (head, 32-query tile) of one attention problem: q [Nq, H*32], k/v [Nk, H*32] -> out [Nq, H*32]; 64 threads
// Key split: the AT_SPLIT wavefronts of a workgroup share the query tile and take the 32-key blocks w, w + AT_SPLIT, ... each with its
// own running (max, sum, O^T); the partial states are merged through LDS at the end (O = sum_w O_w e^{m_w - m} / sum_w l_w e^{m_w - m}).
// One wavefront per tile walked all ~26 key blocks of a cloud one after the other, every block a dependent chain (S^T MFMAs -> softmax
// -> P·V MFMAs), and a pair's layer was 108 wavefronts on 1 024 SIMDs: the launch took as long as that one chain (68-90 us).
constexpr int AT_SPLIT = 4;
struct AttnMerge {
  float m[AT_SPLIT][32], l[AT_SPLIT][32];
  float o[AT_SPLIT][16][64];
};
__device__ __forceinline__ void attention_tile(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                               int64_t Nq, int64_t Nk, int heads, float scale, float* __restrict__ out, int64_t q0, int head,
                                               float* s_q, float* s_k, float* s_v, AttnMerge* mg) {
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  const int wsp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // this wavefront's share of the key blocks
  const int ld = heads * AT_D;
  if (wsp == 0) load_tile(q, Nq, q0, ld, head * AT_D, s_q);
  __syncthreads();
  // B operand of S^T = K·Q^T: B[kd][j=query] = Q[query][kd]; hoisted: 16 values per lane
  float qf[16];
  const float scale2 = scale * 1.44269504088896341f;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) qf[kk] = s_q[col * AT_LD + 2 * kk + half] * scale2;     // scores in base-2 units: exp2 below is the bare v_exp_f32

  floatx16 o;   // O^T[d = row][query = col]
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -INFINITY, l = 0.f;

  float4 kreg[4], vreg[4];
  const int64_t kb0 = 32 * static_cast<int64_t>(wsp), kstep = 32 * AT_SPLIT;
  if (kb0 < Nk) {
    tile_to_regs(k, Nk, kb0, ld, head * AT_D, kreg);
    tile_to_regs(v, Nk, kb0, ld, head * AT_D, vreg);
  }
  for (int64_t k0 = kb0; k0 < Nk; k0 += kstep) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                      // everyone is done reading the previous tile
    regs_to_lds(kreg, s_k);
    regs_to_lds(vreg, s_v);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (k0 + kstep < Nk) {                                // next tile's loads fly while this tile's 32 MFMAs run
      tile_to_regs(k, Nk, k0 + kstep, ld, head * AT_D, kreg);
      tile_to_regs(v, Nk, k0 + kstep, ld, head * AT_D, vreg);
    }
    // S^T[key][query] = sum_d K[key][d] * Q[query][d]
    floatx16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float a = s_k[col * AT_LD + 2 * kk + half];   // A[i=key][kd]
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qf[kk], s, 0, 0, 0);
    }
    // this lane: query `col`, keys key(r) = (r&3) + 8*(r>>2) + 4*half; partner lane (lane^32) holds the other 16 keys
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= Nk) s[r] = -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);                 // finite: every tile holds at least one valid key
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // exp2(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    // O^T[d][query] += sum_key V[key][d] * P[key][query]:  A[i=d][kk] = V[key][d],  B[kk][j=query] = P[key][query]
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      // the key this lane must supply is 2*kk + half; it lives in half hp = (kk>>1)&1, register 2*(kk&1) + (key&1) + 4*(kk>>2)
      constexpr int dummy = 0;
      (void)dummy;
      const int hp = (kk >> 1) & 1;
      const int rbase = 2 * (kk & 1) + 4 * (kk >> 2);
      const float own = (half == 0) ? s[rbase + 0] : s[rbase + 1];       // value for a requester in my own half
      const float other = (half == 0) ? s[rbase + 1] : s[rbase + 0];     // value my partner (other half) needs
      const float recv = __shfl_xor(other, 32);
      const float b = (half == hp) ? own : recv;
      const float a = s_v[(2 * kk + half) * AT_LD + col];                 // A[i=d=col][key]
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, o, 0, 0, 0);
    }
  }
  // merge the AT_SPLIT partial states: wavefront w finishes accumulator rows 4w .. 4w+3 of every lane
  if (half == 0) {
    mg->m[wsp][col] = m;                                    // -inf / 0 for a wavefront without key blocks (fewer than 32 w keys)
    mg->l[wsp][col] = l;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) mg->o[wsp][r][lane] = o[r];
  __syncthreads();
  float mm = -INFINITY;
#pragma unroll
  for (int u = 0; u < AT_SPLIT; ++u) mm = fmaxf(mm, mg->m[u][col]);
  float lsum = 0.f, wgt[AT_SPLIT];
#pragma unroll
  for (int u = 0; u < AT_SPLIT; ++u) {
    wgt[u] = __builtin_amdgcn_exp2f(mg->m[u][col] - mm);                      // exp(-inf) = 0; wavefront 0 always has a block, so mm is finite
    lsum += mg->l[u][col] * wgt[u];
  }
  // O^T[d][query]: lane = query `col`, rows d = (r&3) + 8*(r>>2) + 4*half
  const int64_t qi = q0 + col;
  if (qi < Nq) {
    const float inv = 1.f / lsum;
#pragma unroll
    for (int rr = 0; rr < 16 / AT_SPLIT; ++rr) {
      const int r = wsp * (16 / AT_SPLIT) + rr;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < AT_SPLIT; ++u) acc += mg->o[u][r][lane] * wgt[u];
      const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
      out[qi * ld + head * AT_D + d] = acc * inv;
    }
  }
}

// grid (ceil(Nq/32), H)
__global__ __launch_bounds__(64 * AT_SPLIT) void k_attention(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             int64_t Nq, int64_t Nk, int heads, float scale, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_q[32 * AT_LD], s_k[AT_SPLIT][32 * AT_LD], s_v[AT_SPLIT][32 * AT_LD];
  __shared__ AttnMerge s_mg;
  const int wsp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  attention_tile(q, k, v, Nq, Nk, heads, scale, out, static_cast<int64_t>(blockIdx.x) * 32, blockIdx.y, s_q, s_k[wsp], s_v[wsp], &s_mg);
}

// Several independent attention problems in ONE launch (registration pairs batched per call: the self layers of 2P clouds, the
// cross layers of P pairs): problem p attends queries [q_off[p], q_off[p+1]) to keys [k_off[p], k_off[p+1]) of the stacked
// tensors.  One pair alone is 27 query tiles x 4 heads = 108 wavefronts on 256 CUs; P pairs fill the chip.
constexpr int AT_MAX_P = 64;
struct AttnSeg {
  int P;
  int q_off[AT_MAX_P + 1], k_off[AT_MAX_P + 1], tile_off[AT_MAX_P + 1];   // row offsets; first query tile of every problem
};
__global__ __launch_bounds__(64 * AT_SPLIT) void k_attention_seg(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                 AttnSeg seg, int heads, float scale, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_q[32 * AT_LD], s_k[AT_SPLIT][32 * AT_LD], s_v[AT_SPLIT][32 * AT_LD];
  __shared__ AttnMerge s_mg;
  const int wsp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int p = 0;
  while (p + 1 < seg.P && static_cast<int>(blockIdx.x) >= seg.tile_off[p + 1]) ++p;      // block-uniform, scalar
  const int ld = heads * AT_D;
  const int64_t qo = seg.q_off[p], ko = seg.k_off[p];
  attention_tile(q + qo * ld, k + ko * ld, v + ko * ld, seg.q_off[p + 1] - qo, seg.k_off[p + 1] - ko, heads, scale, out + qo * ld,
                 static_cast<int64_t>(static_cast<int>(blockIdx.x) - seg.tile_off[p]) * 32, blockIdx.y, s_q, s_k[wsp], s_v[wsp], &s_mg);
}

// ---- top-k sparsified attention: dynamic_attention with k != None (rpetransformer.py:19-39) --------------------------------------
// The shipped configuration has cfg.GAT.k = None (full softmax, the fused kernel above); with a fraction per self layer the reference keeps,
// per query and head, the kk = int(n_queries * k) largest scores, soft-maxes THOSE and zeroes the rest.  Built for completeness, not for
// speed (one wavefront per (query, head); scores of the row parked in LDS, the kk-th largest found by a 4-pass radix select on the
// order-preserving bits, ties at the threshold taken in index order — torch.topk leaves that order unspecified).
constexpr int ATK_MAXK = 4096;        // keys per problem (16 KB of LDS per wavefront)
struct AttnTopkSeg {
  int P;
  int q_off[AT_MAX_P + 1], k_off[AT_MAX_P + 1], kk[AT_MAX_P];
};
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
__global__ __launch_bounds__(256) void k_attention_topk(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        AttnTopkSeg seg, int heads, float scale, float* __restrict__ out) {
  __shared__ float s_sc[4][ATK_MAXK];
  __shared__ int s_hist[4][256];
  __shared__ float s_o[4][AT_D];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t qi = static_cast<int64_t>(blockIdx.x) * 4 + w;
  if (qi >= seg.q_off[seg.P]) return;                           // whole wavefront (no block-wide barriers below)
  int p = 0;
  while (p + 1 < seg.P && qi >= seg.q_off[p + 1]) ++p;
  const int head = blockIdx.y, ld = heads * AT_D;
  const int64_t ko = seg.k_off[p];
  const int m = seg.k_off[p + 1] - seg.k_off[p], kk = min(seg.kk[p], m);
  float* sc = s_sc[w];
  int* hist = s_hist[w];
  float* op = out + qi * ld + head * AT_D;
  if (kk <= 0) {                                                // topk(0): nothing kept, the output row is zero
    if (lane < AT_D) op[lane] = 0.f;
    return;
  }
  float qv[AT_D];
  {
    const float4* qp = reinterpret_cast<const float4*>(q + qi * ld + head * AT_D);
#pragma unroll
    for (int c = 0; c < AT_D / 4; ++c) {
      const float4 t = qp[c];
      qv[4 * c] = t.x, qv[4 * c + 1] = t.y, qv[4 * c + 2] = t.z, qv[4 * c + 3] = t.w;
    }
  }
  float mx = -INFINITY;
  for (int j = lane; j < m; j += 64) {
    const float4* kp = reinterpret_cast<const float4*>(k + (ko + j) * ld + head * AT_D);
    float sdot = 0.f;
#pragma unroll
    for (int c = 0; c < AT_D / 4; ++c) {
      const float4 t = kp[c];
      sdot = fmaf(qv[4 * c], t.x, sdot);
      sdot = fmaf(qv[4 * c + 1], t.y, sdot);
      sdot = fmaf(qv[4 * c + 2], t.z, sdot);
      sdot = fmaf(qv[4 * c + 3], t.w, sdot);
    }
    sdot *= scale;
    sc[j] = sdot;
    mx = fmaxf(mx, sdot);
  }
  mx = wave_max_f(mx);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // kk-th largest score: radix select over the order-preserving 32-bit keys, most significant byte first
  uint32_t prefix = 0u, mask = 0u;
  int remaining = kk;                                           // how many of the keys matching `prefix` are still to be taken from the top
  if (kk < m) {
    for (int pass = 3; pass >= 0; --pass) {
      const int sh = 8 * pass;
      for (int i = lane; i < 256; i += 64) hist[i] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int j = lane; j < m; j += 64) {
        const uint32_t u = f2ord(sc[j]);
        if ((u & mask) == prefix) atomicAdd(&hist[(u >> sh) & 255u], 1);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // lane L owns bins 255-4L .. 252-4L (descending), so an inclusive scan over the lanes counts the keys in HIGHER bins
      const int d0 = 255 - 4 * lane;
      const int c0 = hist[d0], c1 = hist[d0 - 1], c2 = hist[d0 - 2], c3 = hist[d0 - 3];
      const int tot = c0 + c1 + c2 + c3;
      const int incl = wave_incl_scan(tot), excl = incl - tot;
      const bool mine = excl < remaining && remaining <= incl;  // exactly one lane
      int bsel = 0, above = 0;
      if (mine) {
        int acc = excl;
        if (remaining <= acc + c0) bsel = d0, above = acc;
        else if (remaining <= acc + c0 + c1) bsel = d0 - 1, above = acc + c0;
        else if (remaining <= acc + c0 + c1 + c2) bsel = d0 - 2, above = acc + c0 + c1;
        else bsel = d0 - 3, above = acc + c0 + c1 + c2;
      }
      const int src = __builtin_ctzll(wave_ballot(mine));
      bsel = rdlane(bsel, src);
      above = rdlane(above, src);
      prefix |= static_cast<uint32_t>(bsel) << sh;
      mask |= 0xffu << sh;
      remaining -= above;
    }
  }
  // soft-max over the kept scores (their largest is the row's largest) and the weighted sum of their value rows
  float acc[AT_D];
#pragma unroll
  for (int d = 0; d < AT_D; ++d) acc[d] = 0.f;
  float sum = 0.f;
  int eq_taken = 0;
  for (int j0 = 0; j0 < m; j0 += 64) {
    const int j = j0 + lane;
    const bool live = j < m;
    const float sj = live ? sc[j] : 0.f;
    const uint32_t u = f2ord(sj);
    bool sel = live;
    if (kk < m) {
      const bool eq = live && u == prefix;
      const uint64_t be = wave_ballot(eq);
      sel = live && (u > prefix || (eq && eq_taken + mbcnt_lt(be) < remaining));
      eq_taken += __popcll(be);
    }
    if (sel) {
      const float pj = expf(sj - mx);
      sum += pj;
      const float4* vp = reinterpret_cast<const float4*>(v + (ko + j) * ld + head * AT_D);
#pragma unroll
      for (int c = 0; c < AT_D / 4; ++c) {
        const float4 t = vp[c];
        acc[4 * c] = fmaf(pj, t.x, acc[4 * c]);
        acc[4 * c + 1] = fmaf(pj, t.y, acc[4 * c + 1]);
        acc[4 * c + 2] = fmaf(pj, t.z, acc[4 * c + 2]);
        acc[4 * c + 3] = fmaf(pj, t.w, acc[4 * c + 3]);
      }
    }
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int d = 0; d < AT_D; ++d) {
    const float t = wave_sum(acc[d]);
    if (lane == 0) s_o[w][d] = t / sum;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < AT_D) op[lane] = s_o[w][lane];
}

// y = LayerNorm(a + b) * gamma + beta, rows of D (<= 1024) features; one wavefront per row
__global__ __launch_bounds__(256) void k_add_layernorm(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int64_t N, int D, float eps, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * 4 + w; n < N; n += static_cast<int64_t>(gridDim.x) * 4) {
    float vals[16];
    float s = 0.f;
    int cnt = 0;
    for (int c = lane; c < D; c += 64, ++cnt) {
      const float t = a[n * D + c] + (b ? b[n * D + c] : 0.f);
      vals[cnt] = t;
      s += t;
    }
    s = wave_sum(s);
    const float mean = s / D;
    float ss = 0.f;
    for (int i = 0; i < cnt; ++i) {
      const float d = vals[i] - mean;
      ss = fmaf(d, d, ss);
    }
    ss = wave_sum(ss);
    const float rstd = 1.f / sqrtf(ss / D + eps);
    cnt = 0;
    for (int c = lane; c < D; c += 64, ++cnt) y[n * D + c] = (vals[cnt] - mean) * rstd * gamma[c] + beta[c];
  }
}

__global__ __launch_bounds__(256) void k_relu_inplace(float* __restrict__ x, int64_t n) {
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    x[i] = fmaxf(x[i], 0.f);
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_rotary_embed(float* x, const float* theta, int64_t N, int heads, void* stream) {
  if (!x || !theta || N < 0 || heads < 1) {
    set_error("lcr_rotary_embed: bad argument");
    return LCR_EARG;
  }
  if (N == 0) return LCR_OK;
  const int64_t total = N * heads * 16;
  hipLaunchKernelGGL(k_rotary, dim3(static_cast<int>(std::min<int64_t>((total + 255) / 256, 4096))), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     theta, N, heads);
  return check_launch("lcr_rotary_embed");
}

extern "C" int lcr_attention_f32(const float* q, const float* k, const float* v, int64_t Nq, int64_t Nk, int heads, int head_dim, float* out,
                                 void* stream) {
  if (!q || !k || !v || !out || Nq < 0 || Nk < 1 || heads < 1 || head_dim != AT_D) {
    set_error("lcr_attention_f32: bad argument (head_dim must be %d, Nk >= 1)", AT_D);
    return LCR_EARG;
  }
  if (Nq == 0) return LCR_OK;
  const float scale = 1.f / sqrtf(static_cast<float>(head_dim));
  KernelTimerScope timed(KT_ATTENTION, static_cast<hipStream_t>(stream), Nq * Nk, 1, heads, head_dim);
  LCR_LAUNCH_TIMED(k_attention, dim3(static_cast<int>((Nq + 31) / 32), heads), dim3(64 * AT_SPLIT), 0, static_cast<hipStream_t>(stream), q, k, v, Nq, Nk,
                   heads, scale, out);
  return check_launch("lcr_attention_f32");
}

extern "C" int lcr_attention_seg_f32(const float* q, const float* k, const float* v, const int64_t* q_len_host, const int64_t* k_len_host,
                                     int P, int heads, int head_dim, float* out, void* stream) {
  if (!q || !k || !v || !out || !q_len_host || !k_len_host || P < 1 || P > AT_MAX_P || heads < 1 || head_dim != AT_D) {
    set_error("lcr_attention_seg_f32: bad argument (head_dim must be %d, 1 <= P <= %d)", AT_D, AT_MAX_P);
    return LCR_EARG;
  }
  AttnSeg seg;
  seg.P = P;
  int64_t qo = 0, ko = 0, to = 0;
  for (int p = 0; p < P; ++p) {
    if (q_len_host[p] < 0 || k_len_host[p] < 1) {
      set_error("lcr_attention_seg_f32: problem %d has %lld queries / %lld keys (keys >= 1)", p, static_cast<long long>(q_len_host[p]),
                static_cast<long long>(k_len_host[p]));
      return LCR_EARG;
    }
    seg.q_off[p] = static_cast<int>(qo);
    seg.k_off[p] = static_cast<int>(ko);
    seg.tile_off[p] = static_cast<int>(to);
    qo += q_len_host[p];
    ko += k_len_host[p];
    to += (q_len_host[p] + 31) / 32;
    if (qo > 2147483647 || ko > 2147483647) {
      set_error("lcr_attention_seg_f32: more than 2^31-1 stacked rows");
      return LCR_EARG;
    }
  }
  seg.q_off[P] = static_cast<int>(qo);
  seg.k_off[P] = static_cast<int>(ko);
  seg.tile_off[P] = static_cast<int>(to);
  if (to == 0) return LCR_OK;
  const float scale = 1.f / sqrtf(static_cast<float>(head_dim));
  int64_t qk = 0;
  for (int p = 0; p < P; ++p) qk += q_len_host[p] * k_len_host[p];
  KernelTimerScope timed(KT_ATTENTION, static_cast<hipStream_t>(stream), qk, P, heads, head_dim);
  LCR_LAUNCH_TIMED(k_attention_seg, dim3(static_cast<int>(to), heads), dim3(64 * AT_SPLIT), 0, static_cast<hipStream_t>(stream), q, k, v, seg, heads, scale, out);
  return check_launch("lcr_attention_seg_f32");
}

// top-k sparsified attention (rpetransformer.py:19-39 with k != None): problem p keeps, per query and head, its kk_host[p] largest scores
extern "C" int lcr_attention_topk_f32(const float* q, const float* k, const float* v, const int64_t* q_len_host, const int64_t* k_len_host,
                                      const int* kk_host, int P, int heads, int head_dim, float* out, void* stream) {
  if (!q || !k || !v || !out || !q_len_host || !k_len_host || !kk_host || P < 1 || P > AT_MAX_P || heads < 1 || head_dim != AT_D) {
    set_error("lcr_attention_topk_f32: bad argument (head_dim must be %d, 1 <= P <= %d)", AT_D, AT_MAX_P);
    return LCR_EARG;
  }
  AttnTopkSeg seg;
  seg.P = P;
  int64_t qo = 0, ko = 0;
  for (int p = 0; p < P; ++p) {
    if (q_len_host[p] < 0 || k_len_host[p] < 1 || k_len_host[p] > ATK_MAXK || kk_host[p] < 0) {
      set_error("lcr_attention_topk_f32: problem %d has %lld queries / %lld keys (1 <= keys <= %d), k = %d", p, static_cast<long long>(q_len_host[p]),
                static_cast<long long>(k_len_host[p]), ATK_MAXK, kk_host[p]);
      return LCR_EARG;
    }
    seg.q_off[p] = static_cast<int>(qo);
    seg.k_off[p] = static_cast<int>(ko);
    seg.kk[p] = kk_host[p];
    qo += q_len_host[p];
    ko += k_len_host[p];
    if (qo > 2147483647 || ko > 2147483647) {
      set_error("lcr_attention_topk_f32: more than 2^31-1 stacked rows");
      return LCR_EARG;
    }
  }
  seg.q_off[P] = static_cast<int>(qo);
  seg.k_off[P] = static_cast<int>(ko);
  if (qo == 0) return LCR_OK;
  const float scale = 1.f / sqrtf(static_cast<float>(head_dim));
  hipLaunchKernelGGL(k_attention_topk, dim3(static_cast<int>((qo + 3) / 4), heads), dim3(256), 0, static_cast<hipStream_t>(stream), q, k, v, seg, heads,
                     scale, out);
  return check_launch("lcr_attention_topk_f32");
}

extern "C" int lcr_add_layernorm(const float* a, const float* b, const float* gamma, const float* beta, int64_t N, int D, float eps, float* y,
                                 void* stream) {
  if (!a || !gamma || !beta || !y || N < 0 || D < 1 || D > 1024) {
    set_error("lcr_add_layernorm: bad argument (D <= 1024)");
    return LCR_EARG;
  }
  if (N == 0) return LCR_OK;
  hipLaunchKernelGGL(k_add_layernorm, dim3(static_cast<int>(std::min<int64_t>((N + 3) / 4, 4096))), dim3(256), 0, static_cast<hipStream_t>(stream), a, b,
                     gamma, beta, N, D, eps, y);
  return check_launch("lcr_add_layernorm");
}

extern "C" int lcr_relu_inplace(float* x, int64_t n, void* stream) {
  if (!x || n < 0) return LCR_EARG;
  if (n == 0) return LCR_OK;
  hipLaunchKernelGGL(k_relu_inplace, dim3(static_cast<int>(std::min<int64_t>((n + 255) / 256, 4096))), dim3(256), 0, static_cast<hipStream_t>(stream), x, n);
  return check_launch("lcr_relu_inplace");
}
