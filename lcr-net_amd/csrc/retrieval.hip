// retrieval.hip — a-9: descriptor nearest-neighbour search for loop detection (exhaustive squared-L2 top-k with the
// temporal exclusion window), for one shard of query rows against the whole (all-gathered) descriptor set.
//
// Reference: experiments/loop_detection/eval_loop_detection_overlap_dataset.py:183-214 (and the inference twin
// experiments/inference/infer_loop_detection_find_top1.py:79-101): for query i in [101, C-1) a faiss IndexIVFFlat with
// nlist=1 (= exhaustive) is re-built over descriptors [0, i-100) and searched for k=50; rows (i, j, d2) ascending in d2.
// faiss is an un-vendored dependency (no version pinned): its IndexFlat L2 metric is the plain squared
// Euclidean distance; only the order inside exact-distance ties is unspecified — here ties are broken by ascending index.
//
// MI355X design: the (Q x C) inner products come from lcr_gemm_f32 (fp32 MFMA, TB=1: Q·D^T), so the per-query faiss
// rebuild loop (O(C^2) adds, minutes on the CPU) becomes one dense contraction; this file turns products into masked
// distances and selects the k smallest per row:
//   k_l2_mask  : d2[i][j] = |q_i|^2 + |d_j|^2 - 2 q_i·d_j, clamped at 0, +inf outside the window j < i - exclude;
//   k_row_topk : one workgroup per query; the row (<= 160 KB / 4 = 40 K columns per pass, longer rows are chunked with
//                a carried candidate list) is staged in LDS, the k-th smallest key is found with an 8-bit radix select
//                over (d2 bits << 32 | j), and the <= k survivors are ordered by an all-pairs rank (k = 50).
#include <algorithm>
#include <cmath>

#include "common.h"

namespace lcr {

constexpr int TK_T = 512;        // threads per top-k workgroup
constexpr int TK_CHUNK = 16384;  // columns staged in LDS per pass (keys are 8 B)
constexpr int TK_KMAX = 128;     // largest k supported

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

// in place: dots[i][j] -> masked squared distance.  Query row i is global frame q0 + i; database column j is frame j.
__global__ __launch_bounds__(256) void k_l2_mask(float* __restrict__ dots, const float* __restrict__ qn, const float* __restrict__ dn, int64_t Q,
                                                 int64_t C, int64_t q0, int exclude) {
  const int64_t total = Q * C;
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < total; t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t i = t / C, j = t - i * C;
    const float d2 = fmaxf(qn[i] + dn[j] - 2.f * dots[t], 0.f);
    dots[t] = (j < q0 + i - exclude) ? d2 : INFINITY;
  }
}

__device__ __forceinline__ uint64_t make_key(float d2, uint32_t j) { return (static_cast<uint64_t>(__float_as_uint(d2)) << 32) | j; }

// k smallest (d2, j) of every row, ascending; rows with fewer than k finite entries are padded with (-1, +inf).
__global__ __launch_bounds__(TK_T) void k_row_topk(const float* __restrict__ d2, int64_t C, int k, int32_t* __restrict__ out_idx,
                                                   float* __restrict__ out_d2) {
  __shared__ uint64_t s_key[TK_CHUNK + TK_KMAX];   // chunk keys, followed by the carried best-k of earlier chunks
  __shared__ uint64_t s_best[TK_KMAX];
  __shared__ int s_hist[256];
  __shared__ int s_cnt;
  __shared__ uint64_t s_prefix;
  __shared__ int s_need;
  const int64_t row = blockIdx.x;
  const float* r = d2 + row * C;
  const int tid = threadIdx.x;
  int nbest = 0;
  const uint64_t INF_KEY = make_key(INFINITY, 0);   // every finite distance sorts below this
  for (int64_t c0 = 0; c0 < C; c0 += TK_CHUNK) {
    const int nc = static_cast<int>(std::min<int64_t>(TK_CHUNK, C - c0));
    for (int i = tid; i < nc; i += TK_T) s_key[i] = make_key(r[c0 + i], static_cast<uint32_t>(c0 + i));
    for (int i = tid; i < nbest; i += TK_T) s_key[nc + i] = s_best[i];
    const int n = nc + nbest;
    __syncthreads();
    // radix select (MSB first) of the k-th smallest key among the n staged keys
    if (tid == 0) {
      s_prefix = 0ull;
      s_need = k;
    }
    __syncthreads();
    if (n > k) {
      for (int pass = 7; pass >= 0; --pass) {
        for (int i = tid; i < 256; i += TK_T) s_hist[i] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix;
        const uint64_t himask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
        for (int i = tid; i < n; i += TK_T) {
          const uint64_t key = s_key[i];
          if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> (8 * pass)) & 255], 1);
        }
        __syncthreads();
        if (tid == 0) {
          int need = s_need, b = 0;
          for (; b < 256; ++b) {
            if (s_hist[b] >= need) break;
            need -= s_hist[b];
          }
          s_need = need;
          s_prefix = prefix | (static_cast<uint64_t>(b) << (8 * pass));
        }
        __syncthreads();
      }
    } else if (tid == 0) {
      s_prefix = ~0ull;   // keep everything
    }
    __syncthreads();
    // keys are unique (the column index is part of the key): exactly min(n, k) keys are <= the k-th key
    const uint64_t kth = s_prefix;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += TK_T) {
      const uint64_t key = s_key[i];
      if (key <= kth && key < INF_KEY) s_best[atomicAdd(&s_cnt, 1)] = key;
    }
    __syncthreads();
    nbest = s_cnt;
    __syncthreads();
  }
  // order the survivors (all-pairs rank, keys unique)
  for (int e = tid; e < k; e += TK_T) {
    if (e < nbest) {
      const uint64_t key = s_best[e];
      int rank = 0;
      for (int j = 0; j < nbest; ++j) rank += s_best[j] < key;
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
  c.take<float>(static_cast<size_t>(std::max<int64_t>(Q * C, 1)));
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
  float* d2 = c.take<float>(static_cast<size_t>(Q * C));
  float* qn = c.take<float>(static_cast<size_t>(Q));
  float* dn = c.take<float>(static_cast<size_t>(C));
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_row_sqnorm, dim3(static_cast<int>(std::min<int64_t>((Q + 3) / 4, 4096))), dim3(256), 0, st, queries, Q, D, qn);
  hipLaunchKernelGGL(k_row_sqnorm, dim3(static_cast<int>(std::min<int64_t>((C + 3) / 4, 4096))), dim3(256), 0, st, database, C, D, dn);
  int rc = lcr_gemm_f32(queries, database, d2, Q, static_cast<int>(C), D, 0, 1, nullptr, nullptr, nullptr, 0, 0, nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_l2_mask, dim3(static_cast<int>(std::min<int64_t>((Q * C + 255) / 256, 8192))), dim3(256), 0, st, d2, qn, dn, Q, C, q0,
                     exclude);
  hipLaunchKernelGGL(k_row_topk, dim3(static_cast<int>(Q)), dim3(TK_T), 0, st, d2, C, k, out_idx, out_d2);
  return check_launch("lcr_retrieval_topk");
}
