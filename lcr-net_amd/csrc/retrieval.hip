// retrieval.hip — a-9: descriptor nearest-neighbour search for loop detection (exhaustive squared-L2 top-k with the
// temporal exclusion window), for one shard of query rows against the whole (all-gathered) descriptor set.
//
// Reference: experiments/loop_detection/eval_loop_detection_overlap_dataset.py:183-214 (and the inference twin
// experiments/inference/infer_loop_detection_find_top1.py:79-101): for query i in [101, C-1) a faiss IndexIVFFlat with
// nlist=1 (= exhaustive) is re-built over descriptors [0, i-100) and searched for k=50; rows (i, j, d2) ascending in d2.
// faiss is an un-vendored dependency (no version pinned): its IndexFlat L2 metric is the plain squared
// Euclidean distance; only the order inside exact-distance ties is unspecified — here ties are broken by ascending index.
//
// MI355X design: the inner products come from lcr_gemm_f32 (fp32 MFMA, TB=1: Q·D^T), so the per-query faiss rebuild loop
// (O(C^2) adds, minutes on the CPU) becomes dense contractions — over blocks of RT_ROWS query rows, each against only the
// database columns its LAST row may see (j < i - exclude: the causal window halves the work and the product buffer is one block,
// not Q x C).  k_row_topk then selects per row, one workgroup per query, in ONE pass over the row's products:
//   * d2 = |q_i|^2 + |d_j|^2 - 2 q_i·d_j (clamped at 0) is formed on the fly — no masked-distance matrix is written;
//   * the first TK_FIRST columns are staged in LDS and their k-th smallest key (d2 bits << 32 | j) found with an 8-bit radix
//     select; that key is the admission threshold for the rest of the row, which is only STREAMED: a column is kept iff its key
//     beats the threshold (on average k * (C / TK_FIRST - 1) of them), admitted keys collect in an LDS list that is re-selected
//     (tightening the threshold) only when it could overflow, and once at the end;
//   * the <= k survivors are ordered by an all-pairs rank.
// (Round 2 materialised the masked Q x C distances and radix-selected every 16 K-column chunk of every row: 16.4 ms for the
// 23 201-frame corpus of KITTI 00-10; this form: see profiles/r03_retrieval_bench.json.)
#include <algorithm>
#include <cmath>

#include "common.h"

namespace lcr {

constexpr int TK_T = 512;        // threads per top-k workgroup
constexpr int TK_FIRST = 2048;   // columns staged for the first selection
constexpr int TK_CAND = 4096;    // admitted keys collected before a re-selection
constexpr int TK_STEP = 2048;    // columns streamed between two overflow checks (TK_CAND - TK_STEP - TK_KMAX keys always fit)
constexpr int TK_KMAX = 128;     // largest k supported
constexpr int RT_ROWS = 2048;    // query rows per product block

__global__ __launch_bounds__(256) void k_row_sqnorm(const float* __restrict__ x, int64_t N, int D, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * 4 + w; n < N; n += static_cast<int64_t>(gridDim.x) * 4) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float v = x[n * D + c];
      s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) out[n] = s;
  }
}

__device__ __forceinline__ uint64_t make_key(float d2, uint32_t j) { return (static_cast<uint64_t>(__float_as_uint(d2)) << 32) | j; }

// Block-wide: the k-th smallest of keys[0..n) (unique keys, n > k) by an MSB-first 8-bit radix select.
__device__ uint64_t tk_select(const uint64_t* keys, int n, int k, int* s_hist, uint64_t* s_prefix, int* s_need) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    *s_prefix = 0ull;
    *s_need = k;
  }
  __syncthreads();
  for (int pass = 7; pass >= 0; --pass) {
    for (int i = tid; i < 256; i += TK_T) s_hist[i] = 0;
    __syncthreads();
    const uint64_t prefix = *s_prefix;
    const uint64_t himask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
    for (int i = tid; i < n; i += TK_T) {
      const uint64_t key = keys[i];
      if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> (8 * pass)) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int need = *s_need, b = 0;
      for (; b < 256; ++b) {
        if (s_hist[b] >= need) break;
        need -= s_hist[b];
      }
      *s_need = need;
      *s_prefix = prefix | (static_cast<uint64_t>(b) << (8 * pass));
    }
    __syncthreads();
  }
  return *s_prefix;
}

// k smallest (d2, j) of every row of a product block, ascending; rows with fewer than k admissible columns are padded with
// (-1, +inf).  Row r of the block is global frame g = frame0 + r and sees columns j < min(ncols, g - exclude); dots[r][j] = q_g·d_j.
__global__ __launch_bounds__(TK_T) void k_row_topk(const float* __restrict__ dots, int64_t ld, const float* __restrict__ qn, const float* __restrict__ dn,
                                                   int64_t frame0, int exclude, int64_t ncols, int k, int32_t* __restrict__ out_idx,
                                                   float* __restrict__ out_d2) {
  __shared__ uint64_t s_key[TK_CAND];      // first: the staged columns; then: [0, k) the best so far, [k, ..) admitted keys
  __shared__ uint64_t s_best[TK_KMAX];
  __shared__ int s_hist[256];
  __shared__ int s_cnt;
  __shared__ uint64_t s_prefix;
  __shared__ int s_need;
  const int64_t row = blockIdx.x;
  const float* r = dots + row * ld;
  const int tid = threadIdx.x;
  const int64_t lim = max(static_cast<int64_t>(0), min(ncols, frame0 + row - exclude));
  const float qq = qn[row];
  auto key_of = [&](int64_t j) { return make_key(fmaxf(qq + dn[j] - 2.f * r[j], 0.f), static_cast<uint32_t>(j)); };
  // ---- first selection
  const int n1 = static_cast<int>(min(lim, static_cast<int64_t>(TK_FIRST)));
  for (int i = tid; i < n1; i += TK_T) s_key[i] = key_of(i);
  __syncthreads();
  int nbest = n1;                          // keys held in s_key[0, nbest)
  uint64_t thr = ~0ull;                    // admission threshold: keys <= thr may still be among the k smallest
  auto reselect = [&](int n) {             // keep the min(n, k) smallest of s_key[0, n) in s_key[0, nbest); block-uniform
    if (n > k) {
      thr = tk_select(s_key, n, k, s_hist, &s_prefix, &s_need);
      // keys <= thr (exactly k: keys are unique) move to the front, through a side buffer (no key is overwritten before it is read)
      if (tid == 0) s_cnt = 0;
      __syncthreads();
      for (int i = tid; i < n; i += TK_T) {
        const uint64_t key = s_key[i];
        if (key <= thr) s_best[atomicAdd(&s_cnt, 1)] = key;
      }
      __syncthreads();
      for (int i = tid; i < k; i += TK_T) s_key[i] = s_best[i];
      __syncthreads();
      nbest = k;
    } else {
      nbest = n;
    }
  };
  reselect(n1);
  // ---- stream the rest of the row: admit what beats the threshold
  if (tid == 0) s_cnt = nbest;
  __syncthreads();
  for (int64_t c0 = n1; c0 < lim; c0 += TK_STEP) {
    const int64_t c1 = min(lim, c0 + TK_STEP);
    for (int64_t j = c0 + tid; j < c1; j += TK_T) {
      const uint64_t key = key_of(j);
      if (key <= thr) s_key[atomicAdd(&s_cnt, 1)] = key;
    }
    __syncthreads();
    const int n = s_cnt;
    __syncthreads();                       // every thread holds the SAME n before the next chunk's admissions move s_cnt again:
                                           // a late reader would otherwise take the re-selection branch (barriers inside) alone
    if (n + TK_STEP > TK_CAND || c1 >= lim) {
      reselect(n);
      if (tid == 0) s_cnt = nbest;
      __syncthreads();
    }
  }
  // ---- order the survivors (all-pairs rank, keys unique)
  for (int e = tid; e < k; e += TK_T) {
    if (e < nbest) {
      const uint64_t key = s_key[e];
      int rank = 0;
      for (int j = 0; j < nbest; ++j) rank += s_key[j] < key;
      out_idx[row * k + rank] = static_cast<int32_t>(static_cast<uint32_t>(key));
      out_d2[row * k + rank] = __uint_as_float(static_cast<uint32_t>(key >> 32));
    } else {
      out_idx[row * k + e] = -1;
      out_d2[row * k + e] = INFINITY;
    }
  }
}

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_gemm_f32(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB, const float* bias,
                            const float* rowdiv, const int64_t* seg_len, int S, int groups, double* stats, void* stream);

extern "C" int lcr_retrieval_ws_bytes(int64_t Q, int64_t C, size_t* bytes) {
  if (!bytes || Q < 0 || C < 0) return LCR_EARG;
  Carver c(nullptr, ~size_t(0));
  c.take<float>(static_cast<size_t>(std::max<int64_t>(std::min<int64_t>(Q, RT_ROWS) * C, 1)));   // one block of products
  c.take<float>(static_cast<size_t>(std::max<int64_t>(Q, 1)));
  c.take<float>(static_cast<size_t>(std::max<int64_t>(C, 1)));
  *bytes = c.off;
  return LCR_OK;
}

// queries [Q, D] (global frames q0 .. q0+Q-1), database [C, D] (frames 0 .. C-1) -> out_idx i32[Q,k], out_d2 f32[Q,k]
extern "C" int lcr_retrieval_topk(const float* queries, int64_t Q, int64_t q0, const float* database, int64_t C, int D, int k, int exclude,
                                  int32_t* out_idx, float* out_d2, void* ws, size_t ws_bytes, void* stream) {
  if (!queries || !database || !out_idx || !out_d2 || !ws || Q < 0 || C < 1 || D < 1 || k < 1 || k > TK_KMAX || exclude < 0) {
    set_error("lcr_retrieval_topk: bad argument (k <= %d)", TK_KMAX);
    return LCR_EARG;
  }
  if (C > (int64_t(1) << 31) - 1) {
    set_error("lcr_retrieval_topk: more than 2^31-1 database rows");
    return LCR_EARG;
  }
  size_t need = 0;
  lcr_retrieval_ws_bytes(Q, C, &need);
  if (need > ws_bytes) {
    set_error("lcr_retrieval_topk: workspace too small (%zu < %zu)", ws_bytes, need);
    return LCR_ESPACE;
  }
  if (Q == 0) return LCR_OK;
  Carver c(ws, ws_bytes);
  float* dots = c.take<float>(static_cast<size_t>(std::min<int64_t>(Q, RT_ROWS) * C));
  float* qn = c.take<float>(static_cast<size_t>(Q));
  float* dn = c.take<float>(static_cast<size_t>(C));
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_row_sqnorm, dim3(static_cast<int>(std::min<int64_t>((Q + 3) / 4, 4096))), dim3(256), 0, st, queries, Q, D, qn);
  hipLaunchKernelGGL(k_row_sqnorm, dim3(static_cast<int>(std::min<int64_t>((C + 3) / 4, 4096))), dim3(256), 0, st, database, C, D, dn);
  for (int64_t r0 = 0; r0 < Q; r0 += RT_ROWS) {
    const int64_t rows = std::min<int64_t>(RT_ROWS, Q - r0);
    // columns the block's last row may see (the window is causal: earlier rows see fewer; every row masks by its own bound)
    const int64_t nb = std::max<int64_t>(0, std::min<int64_t>(C, q0 + r0 + rows - 1 - exclude));
    if (nb > 0) {
      int rc = lcr_gemm_f32(queries + r0 * D, database, dots, rows, static_cast<int>(nb), D, 0, 1, nullptr, nullptr, nullptr, 0, 0, nullptr, stream);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_row_topk, dim3(static_cast<int>(rows)), dim3(TK_T), 0, st, dots, nb, qn + r0, dn, q0 + r0, exclude, nb, k, out_idx + r0 * k,
                       out_d2 + r0 * k);
  }
  return check_launch("lcr_retrieval_topk");
}
