// encoder.hip — a-6: KPEncoder.forward (experiments/lcrnet/backbone4.py:60-89) as ONE native call.
//
// The encoder is ~135 dependent launches per pass; issued from Python each costs ~12 us of host time, more than many of the
// stage-3/4 kernels run, so the encoder streams kept draining (GPU idle 18 % of the pipeline's time).  This file is the
// host-side sequencer only: it walks the 11 blocks in the reference's order and calls the library's own entry points
// (lcr_kpconv_cin1 / lcr_gemm_f32 / lcr_groupnorm_* / lcr_kpconv_aggregate / lcr_maxpool) — the same launches, arguments and
// order as lcr-net_amd/modules/kpconv/modules.py, hence bit-identical outputs — with intermediates bump-allocated from one
// workspace (reset per block) and every GroupNorm statistics table carved from one zero-filled arena.
//
//   ConvBlock      (modules.py:104-145): KPConv(C_in = 1) -> GN -> LeakyReLU(0.1)
//   ResidualBlock  (modules.py:148-225): unary1 (Linear+GN+LeakyReLU) -> KPConv -> GN -> LeakyReLU -> unary2 (Linear, GN fused
//                  below) ; shortcut = [maxpool] -> [Linear + GN] ; out = LeakyReLU(GN(unary2) + shortcut)
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

// every library call of the sequence passes the FIFO gate (common.h) for the duration of its own launches
#define TURN(call) ([&]() { LaunchTurn turn_; return (call); }())

namespace lcr {

constexpr float ENC_GN_EPS = 1e-5f;
constexpr float ENC_SLOPE = 0.1f;

struct Arena {
  char*  base;
  size_t off, cap;
  template <typename T>
  T* take(size_t n) {
    const size_t o = off;
    off = align_up(off + n * sizeof(T));
    return (base && off <= cap) ? reinterpret_cast<T*>(base + o) : nullptr;
  }
};

struct StageIO {
  const float*   pts;
  const int64_t* seg;     // GroupNorm segment lengths of this stage (device)
  const int32_t* order;   // optional processing order
  int64_t        n;
  int64_t        min_rows;   // rows of the shortest GroupNorm segment (host knowledge; 0 = unknown)
};

struct StatsPool {
  double* base;
  size_t  off, cap;       // in doubles
  int     S, groups;
  double* take() {
    const size_t n = static_cast<size_t>(GN_REPLICAS) * S * groups * 2;
    if (off + n > cap) return nullptr;
    double* p = base + off;
    off += n;
    return p;
  }
};

static size_t stats_doubles(int S, int groups) { return static_cast<size_t>(GN_REPLICAS) * S * groups * 2 * 44; }   // <= 1 + 4 * 10 tables per pass

// unary block: y = Linear(x) (+ GroupNorm sums); returns the raw output and its statistics table
static int unary_raw(const LcrUnaryW& u, const float* x, int64_t n, int cin, int cout, const StageIO& st, StatsPool& sp, Arena& ws,
                     float** y_out, double** stats_out, hipStream_t s) {
  float* y = ws.take<float>(static_cast<size_t>(n) * cout);
  double* stats = sp.take();
  if (!y || !stats) return LCR_ESPACE;
  // K-deep Linears (stages 3-4) on the bf16 matrix cores when the caller provides the weight's three bf16 terms (fp32-faithful, gemm_f32.hip)
  const bool split = u.w_split && cin >= 288 && cin % 32 == 0 && cout >= 64;
  int rc = split ? TURN(lcr_gemm_f32_bsplit(x, u.w_split, y, n, cout, cin, u.b, nullptr, st.seg, sp.S, sp.groups, stats, s))
                 : TURN(lcr_gemm_f32(x, u.w, y, n, cout, cin, 0, 1, u.b, nullptr, st.seg, sp.S, sp.groups, stats, s));
  *y_out = y;
  *stats_out = stats;
  return rc;
}

static int residual_block(const LcrBlockW& b, const float* s_feats, const StageIO& q, const StageIO& sup, const void* idx, int H,
                          StatsPool& sp, Arena& ws, float* out, unsigned enc_flags, hipStream_t s) {
  const int mid = b.cout / 4, g = sp.groups;
  const int64_t M = q.n, Ns = sup.n;
  int rc;
  // 1. unary1 on the SUPPORT rows (+ the positive-row flags KPConv's neighbour count needs)
  const float* x = s_feats;
  uint8_t* pos = ws.take<uint8_t>(static_cast<size_t>(Ns));
  if (!pos) return LCR_ESPACE;
  if (b.unary1.w) {
    float* y1;
    double* st1;
    if ((rc = unary_raw(b.unary1, s_feats, Ns, b.cin, mid, sup, sp, ws, &y1, &st1, s))) return rc;
    float* x1 = ws.take<float>(static_cast<size_t>(Ns) * mid);
    if (!x1) return LCR_ESPACE;
    if ((rc = TURN(lcr_groupnorm_apply(y1, st1, b.unary1.gn_w, b.unary1.gn_b, nullptr, nullptr, nullptr, nullptr, x1, Ns, mid, g, sup.seg, sp.S,
                                  ENC_GN_EPS, ENC_SLOPE, 1, pos, s))))
      return rc;
    x = x1;
  } else if ((rc = TURN(lcr_row_positive(s_feats, Ns, b.cin, pos, s)))) {
    return rc;
  }
  // 2. KPConv: aggregate + kernel-point contraction (count division, bias, GroupNorm sums in the GEMM epilogue)
  float* A = ws.take<float>(static_cast<size_t>(M) * 15 * mid);
  float* nn = ws.take<float>(static_cast<size_t>(M));
  float* kpo = ws.take<float>(static_cast<size_t>(M) * mid);
  double* stc = sp.take();
  if (!A || !nn || !kpo || !stc) return LCR_ESPACE;
  // lists that come from a radius search (the reference's radius_neighbors and ours alike) hold their valid entries first, padding (= Ns)
  // behind them: the CALLER says so with LCR_ENC_LISTS_VALID_FIRST and the aggregation may then stop at the first chunk with a hole.
  // Without the flag every chunk is scanned (rows with interior padding are legal input).  LCR_KP_VALID_FIRST=0 ignores the flag.
  static const bool vf_env_off = getenv("LCR_KP_VALID_FIRST") && atoi(getenv("LCR_KP_VALID_FIRST")) == 0;
  const int vf_flag = ((enc_flags & LCR_ENC_LISTS_VALID_FIRST) && !vf_env_off) ? LCR_KP_VALID_FIRST : 0;
  static const bool fused32 = getenv("LCR_KPCONV_FUSED") != nullptr;       // opt-in, like KPConv.forward_raw (LABNOTES.md §4.2)
  if (fused32 && mid == 32 && sp.S <= 64) {
    if ((rc = TURN(lcr_kpconv_fused(x, pos, q.pts, sup.pts, idx, 0, M, Ns, H, mid, b.kernel_points_host, b.sigma, b.kp_w, b.kp_b, kpo, q.seg, sp.S, g,
                                    stc, q.order, s))))
      return rc;
  } else {
    if ((rc = TURN(lcr_kpconv_aggregate_ex(x, pos, q.pts, sup.pts, idx, 0, M, Ns, H, mid, b.kernel_points_host, b.sigma, A, nn, q.order, vf_flag, s)))) return rc;
    // weights pre-transposed by the caller ([mid, 15 mid]): both operands k-contiguous -> the K-deep GEMM form
    if (b.kp_wt_split && mid >= 64) {      // N = 32 contractions stream A at the HBM rate on the fp32 form already
      if ((rc = TURN(lcr_gemm_f32_bsplit(A, b.kp_wt_split, kpo, M, mid, 15 * mid, b.kp_b, nn, q.seg, sp.S, g, stc, s)))) return rc;
    } else if ((rc = TURN(b.kp_wt ? lcr_gemm_f32(A, b.kp_wt, kpo, M, mid, 15 * mid, 0, 1, b.kp_b, nn, q.seg, sp.S, g, stc, s)
                                  : lcr_gemm_f32(A, b.kp_w, kpo, M, mid, 15 * mid, 0, 0, b.kp_b, nn, q.seg, sp.S, g, stc, s)))) return rc;
  }
  // 3. + 4. norm_conv + LeakyReLU + unary2 (normalised in step 7).  With segments of >= 64 rows and the light GEMM form, the
  // normalisation happens while unary2's GEMM stages its A tiles (lcr_gemm_f32_anorm) — same rule as ResidualBlock.forward.
  float* y;
  double* sty;
  static const bool no_anorm = getenv("LCR_NO_NORM_ON_LOAD") != nullptr;
  if (!no_anorm && mid <= 256 && mid % 4 == 0 && b.cout > 32 && q.min_rows >= 64) {
    y = ws.take<float>(static_cast<size_t>(M) * b.cout);
    sty = sp.take();
    if (!y || !sty) return LCR_ESPACE;
    if ((rc = TURN(lcr_gemm_f32_anorm(kpo, b.unary2.w, y, M, b.cout, mid, b.unary2.b, stc, b.normconv_w, b.normconv_b, g, ENC_GN_EPS, ENC_SLOPE,
                                      q.seg, sp.S, g, sty, s))))
      return rc;
  } else {
    float* x2 = ws.take<float>(static_cast<size_t>(M) * mid);
    if (!x2) return LCR_ESPACE;
    if ((rc = TURN(lcr_groupnorm_apply(kpo, stc, b.normconv_w, b.normconv_b, nullptr, nullptr, nullptr, nullptr, x2, M, mid, g, q.seg, sp.S, ENC_GN_EPS,
                                  ENC_SLOPE, 1, nullptr, s))))
      return rc;
    if ((rc = unary_raw(b.unary2, x2, M, mid, b.cout, q, sp, ws, &y, &sty, s))) return rc;
  }
  // 5./6. shortcut
  const float* res = s_feats;
  if (b.strided) {
    float* pooled = ws.take<float>(static_cast<size_t>(M) * b.cin);
    if (!pooled) return LCR_ESPACE;
    if ((rc = TURN(lcr_maxpool(s_feats, idx, 0, M, Ns, H, b.cin, pooled, q.order, s)))) return rc;
    res = pooled;
  }
  const double* str = nullptr;
  if (b.shortcut.w) {
    float* r;
    double* st;
    if ((rc = unary_raw(b.shortcut, res, M, b.cin, b.cout, q, sp, ws, &r, &st, s))) return rc;
    res = r;
    str = st;
  }
  // 7. out = LeakyReLU(GN(unary2) + [GN](shortcut))
  return TURN(lcr_groupnorm_apply(y, sty, b.unary2.gn_w, b.unary2.gn_b, res, str, str ? b.shortcut.gn_w : nullptr, str ? b.shortcut.gn_b : nullptr, out,
                             M, b.cout, g, q.seg, sp.S, ENC_GN_EPS, ENC_SLOPE, 1, nullptr, s));
}

// workspace of one block (worst case over the blocks), + two ping-pong block outputs + the statistics arena
static size_t block_ws_bytes(const LcrBlockW& b, int64_t M, int64_t Ns) {
  Arena a{nullptr, 0, ~size_t(0)};
  const int mid = b.cout / 4;
  a.take<uint8_t>(Ns);
  a.take<float>(static_cast<size_t>(Ns) * mid);
  a.take<float>(static_cast<size_t>(Ns) * mid);
  a.take<float>(static_cast<size_t>(M) * 15 * mid);
  a.take<float>(M);
  a.take<float>(static_cast<size_t>(M) * mid);
  a.take<float>(static_cast<size_t>(M) * mid);
  a.take<float>(static_cast<size_t>(M) * b.cout);
  a.take<float>(static_cast<size_t>(M) * b.cin);
  a.take<float>(static_cast<size_t>(M) * b.cout);
  return a.off;
}

// blocks[i] runs with queries at stage BLOCK_Q[i] and supports at stage BLOCK_S[i] (backbone4.py:66-87)
static const int BLOCK_Q[LCR_ENC_BLOCKS] = {0, 1, 1, 1, 2, 2, 2, 3, 3, 3};
static const int BLOCK_S[LCR_ENC_BLOCKS] = {0, 0, 1, 1, 1, 2, 2, 2, 3, 3};

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_encoder_ws_bytes(const LcrEncoderW* W, const int64_t* n_host, int S, size_t* bytes) {
  if (!W || !n_host || !bytes || S < 1) return LCR_EARG;
  size_t blk = 0, pp = 0;
  for (int i = 0; i < LCR_ENC_BLOCKS; ++i) {
    blk = std::max(blk, block_ws_bytes(W->blocks[i], n_host[BLOCK_Q[i]], n_host[BLOCK_S[i]]));
    pp = std::max(pp, align_up(sizeof(float) * static_cast<size_t>(n_host[BLOCK_Q[i]]) * W->blocks[i].cout));
  }
  pp = std::max(pp, align_up(sizeof(float) * static_cast<size_t>(n_host[0]) * W->c1_cout * 2));     // encoder1_1: raw + normalised
  *bytes = blk + 2 * pp + align_up(sizeof(double) * stats_doubles(S, W->groups)) + 4096;
  return LCR_OK;
}

extern "C" int lcr_encoder_forward(const LcrEncoderW* W, const float* feats0, const float* const* points, const int32_t* const* neighbors,
                                   const int32_t* const* subsampling, const int32_t* const* order, const int64_t* const* seg_len, int S,
                                   const int64_t* n_host, const int64_t* seg_min_rows_host, const int* limits, float* const* out_feats,
                                   void* ws, size_t ws_bytes, void* stream) {
  return lcr_encoder_forward_ex(W, feats0, points, neighbors, subsampling, order, seg_len, S, n_host, seg_min_rows_host, limits, out_feats, 0u, ws,
                                ws_bytes, stream);
}

extern "C" int lcr_encoder_forward_ex(const LcrEncoderW* W, const float* feats0, const float* const* points, const int32_t* const* neighbors,
                                      const int32_t* const* subsampling, const int32_t* const* order, const int64_t* const* seg_len, int S,
                                      const int64_t* n_host, const int64_t* seg_min_rows_host, const int* limits, float* const* out_feats,
                                      unsigned flags, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !feats0 || !points || !neighbors || !subsampling || !seg_len || !n_host || !limits || !out_feats || !ws || S < 1) {
    set_error("lcr_encoder_forward: bad argument");
    return LCR_EARG;
  }
  size_t need = 0;
  lcr_encoder_ws_bytes(W, n_host, S, &need);
  if (ws_bytes < need) {
    set_error("lcr_encoder_forward: workspace too small (%zu < %zu)", ws_bytes, need);
    return LCR_ESPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  StageIO st[4];
  for (int i = 0; i < 4; ++i) st[i] = StageIO{points[i], seg_len[i], order ? order[i] : nullptr, n_host[i], seg_min_rows_host ? seg_min_rows_host[i] : 0};
  // layout: [statistics arena][ping][pong][block scratch]
  Arena top{static_cast<char*>(ws), 0, ws_bytes};
  StatsPool sp{top.take<double>(stats_doubles(S, W->groups)), 0, stats_doubles(S, W->groups), S, W->groups};
  size_t pp = 0;
  for (int i = 0; i < LCR_ENC_BLOCKS; ++i) pp = std::max(pp, sizeof(float) * static_cast<size_t>(n_host[BLOCK_Q[i]]) * W->blocks[i].cout);
  pp = std::max(pp, sizeof(float) * static_cast<size_t>(n_host[0]) * W->c1_cout * 2);
  float* ping = reinterpret_cast<float*>(top.take<char>(pp));
  float* pong = reinterpret_cast<float*>(top.take<char>(pp));
  const size_t scratch0 = top.off;
  if (!sp.base || !ping || !pong) return LCR_ESPACE;
  hipMemsetAsync(sp.base, 0, sizeof(double) * sp.cap, s);

  int rc;
  // encoder1_1: ConvBlock with C_in = 1
  float* cur;
  {
    const int64_t n0 = n_host[0];
    float* raw = ping;
    float* f = ping + static_cast<size_t>(n0) * W->c1_cout;
    double* stt = sp.take();
    if ((rc = TURN(lcr_kpconv_cin1(feats0, st[0].pts, st[0].pts, neighbors[0], 0, n0, n0, limits[0], W->c1_kernel_points_host, W->c1_sigma, W->c1_w,
                              W->c1_b, W->c1_cout, raw, st[0].order, s))))
      return rc;
    if ((rc = TURN(lcr_groupnorm_stats(raw, n0, W->c1_cout, W->groups, st[0].seg, S, stt, s)))) return rc;
    if ((rc = TURN(lcr_groupnorm_apply(raw, stt, W->c1_gn_w, W->c1_gn_b, nullptr, nullptr, nullptr, nullptr, f, n0, W->c1_cout, W->groups, st[0].seg, S,
                                  ENC_GN_EPS, ENC_SLOPE, 1, nullptr, s))))
      return rc;
    cur = f;
  }
  // the ten residual blocks; the last block of every stage writes the caller's stage output
  static const int STAGE_LAST[4] = {0, 3, 6, 9};
  for (int i = 0; i < LCR_ENC_BLOCKS; ++i) {
    const int qs = BLOCK_Q[i], ss = BLOCK_S[i];
    const bool strided = qs != ss;
    const void* idx = strided ? static_cast<const void*>(subsampling[ss]) : static_cast<const void*>(neighbors[qs]);
    const int H = limits[ss];            // subsampling[ss] and neighbors[ss] are both `limits[ss]` wide (data.py:36-47)
    float* out = nullptr;
    for (int k = 0; k < 4; ++k)
      if (STAGE_LAST[k] == i) out = out_feats[k];
    if (!out) out = (cur >= ping && cur < ping + pp / sizeof(float)) ? pong : ping;
    Arena scratch{static_cast<char*>(ws), scratch0, ws_bytes};
    if ((rc = residual_block(W->blocks[i], cur, st[qs], st[ss], idx, H, sp, scratch, out, flags, s))) {
      if (rc == LCR_ESPACE) set_error("lcr_encoder_forward: block %d ran out of workspace", i);
      return rc;
    }
    cur = out;
  }
  return check_launch("lcr_encoder_forward");
}
