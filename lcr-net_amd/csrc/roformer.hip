// roformer.hip — a-8: ThDRoFormer.forward (experiments/lcrnet/modules/thdroformer/thdroformer_linear.py:60-97, rpetransformer.py:57-220,
// vanilla_transformer.py:13-144, Rotary3DPosEmb.py:27-38) as ONE native call.
//
// The transformer of a registration pair is ~130 dependent launches on a few hundred coarse nodes: every kernel sits on its launch floor,
// and issued from Python (≈ 12 us of interpreter time per launch) the pass cost 1.8 ms of host time per pair — a quarter of what one pair per
// call spends on the host, under the interpreter lock the second worker needs (profiles/r06_pair_host_profile.log).  Like encoder.hip this
// file is the host-side sequencer only: it issues the library's own entry points (lcr_gemm_f32 / lcr_rotary_embed / lcr_attention_seg_f32 /
// lcr_add_layernorm / lcr_relu_inplace) with the arguments and in the order of lcr-net_amd/modules/thdroformer/thdroformer_linear.py,
// hence bit-identical outputs, with the intermediates bump-allocated from one workspace.
//
// Row layout (the module's own): the rows of all FIRST clouds of the P pairs stacked, then the rows of all SECOND clouds; a self layer runs
// over all rows as 2P attention problems, a cross layer as P problems per direction, cloud 1 attending to the UPDATED cloud 0
// (rpetransformer.py:213-214).
#include <algorithm>
#include <cstring>

#include "common.h"

#define TURN(call) ([&]() { LaunchTurn turn_; return (call); }())

namespace lcr {

struct RfArena {
  char*  base;
  size_t off, cap;
  float* take(size_t n) {
    const size_t o = off;
    off = align_up(off + n * sizeof(float));
    return (base && off <= cap) ? reinterpret_cast<float*>(base + o) : nullptr;
  }
};

static int rf_linear(const LcrLinearW& l, const float* x, int64_t n, int in, int out, float* y, bool relu, hipStream_t s) {
  int rc = TURN(lcr_gemm_f32(x, l.w, y, n, out, in, 0, 1, l.b, nullptr, nullptr, 0, 0, nullptr, s));
  if (rc) return rc;
  return relu ? TURN(lcr_relu_inplace(y, n * out, s)) : LCR_OK;
}

// _TransformerLayer.forward (thdroformer_linear.py:56-91): x [nx, d] attends to mem [nm, d]; theta != null: rotary on q and k (self layers,
// x == mem).  y [nx, d].  Scratch from `a` (reset by the caller per layer).
static int rf_layer(const LcrRoformerLayerW& L, int d, int heads, const float* x, int64_t nx, const float* mem, int64_t nm, const float* theta,
                    const int64_t* xl, const int64_t* ml, int P, float* y, RfArena a, hipStream_t s) {
  float* q = a.take(static_cast<size_t>(nx) * d);
  float* k = a.take(static_cast<size_t>(nm) * d);
  float* v = a.take(static_cast<size_t>(nm) * d);
  float* h = a.take(static_cast<size_t>(nx) * d);
  float* h2 = a.take(static_cast<size_t>(nx) * d);
  float* y1 = a.take(static_cast<size_t>(nx) * d);
  float* e = a.take(static_cast<size_t>(nx) * 2 * d);
  float* sq = a.take(static_cast<size_t>(nx) * d);
  if (!q || !k || !v || !h || !h2 || !y1 || !e || !sq) return LCR_ESPACE;
  int rc;
  if ((rc = rf_linear(L.q, x, nx, d, d, q, false, s))) return rc;
  if ((rc = rf_linear(L.k, mem, nm, d, d, k, false, s))) return rc;
  if ((rc = rf_linear(L.v, mem, nm, d, d, v, false, s))) return rc;
  if (theta) {
    if ((rc = TURN(lcr_rotary_embed(q, theta, nx, heads, s)))) return rc;
    if ((rc = TURN(lcr_rotary_embed(k, theta, nm, heads, s)))) return rc;
  }
  if ((rc = TURN(lcr_attention_seg_f32(q, k, v, xl, ml, P, heads, d / heads, h, s)))) return rc;
  if ((rc = rf_linear(L.lin, h, nx, d, d, h2, false, s))) return rc;
  if ((rc = TURN(lcr_add_layernorm(h2, x, L.ln1_w, L.ln1_b, nx, d, L.ln1_eps, y1, s)))) return rc;          // norm(hidden + input)
  if ((rc = rf_linear(L.expand, y1, nx, d, 2 * d, e, true, s))) return rc;
  if ((rc = rf_linear(L.squeeze, e, nx, 2 * d, d, sq, false, s))) return rc;
  return TURN(lcr_add_layernorm(y1, sq, L.ln2_w, L.ln2_b, nx, d, L.ln2_eps, y, s));                           // norm(input + squeeze)
}

static size_t rf_layer_floats(int64_t n, int d) { return static_cast<size_t>(n) * d * 9 + 8 * 64; }          // q k v h h2 y1 e(2) sq + alignment slack

}  // namespace lcr

using namespace lcr;

extern "C" int lcr_roformer_ws_bytes(const LcrRoformerW* W, int64_t n_rows, size_t* bytes) {
  if (!W || !bytes || n_rows < 0 || W->d_model < 1) return LCR_EARG;
  const size_t n = static_cast<size_t>(std::max<int64_t>(n_rows, 1));
  // embedding hidden [n, d] + ping + pong [n, d] + the layer scratch
  *bytes = align_up(sizeof(float) * n * W->d_model) * 3 + align_up(sizeof(float) * rf_layer_floats(n_rows, W->d_model)) + 4096;
  return LCR_OK;
}

extern "C" int lcr_roformer_forward(const LcrRoformerW* W, const float* points, const float* feats, const int64_t* lens0_host, const int64_t* lens1_host,
                                    int P, float* out, float* theta_out, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !points || !feats || !lens0_host || !lens1_host || !out || !theta_out || !ws || P < 1 || 2 * P > 64 || W->num_blocks < 0 ||
      W->num_blocks > LCR_ROFORMER_MAX_BLOCKS || W->heads < 1 || W->d_model % W->heads != 0 || W->d_model / W->heads != 32) {
    set_error("lcr_roformer_forward: bad argument (1 <= pairs <= 32, head dim 32, <= %d blocks)", LCR_ROFORMER_MAX_BLOCKS);
    return LCR_EARG;
  }
  int64_t n0 = 0, n1 = 0;
  int64_t cat[64];
  for (int p = 0; p < P; ++p) {
    if (lens0_host[p] < 1 || lens1_host[p] < 1) {
      set_error("lcr_roformer_forward: pair %d has an empty cloud", p);
      return LCR_EARG;
    }
    cat[p] = lens0_host[p];
    cat[P + p] = lens1_host[p];
    n0 += lens0_host[p];
    n1 += lens1_host[p];
  }
  const int64_t n = n0 + n1;
  size_t need = 0;
  lcr_roformer_ws_bytes(W, n, &need);
  if (ws_bytes < need) {
    set_error("lcr_roformer_forward: workspace too small (%zu < %zu)", ws_bytes, need);
    return LCR_ESPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int d = W->d_model;
  RfArena top{static_cast<char*>(ws), 0, ws_bytes};
  float* emb_h = top.take(static_cast<size_t>(n) * d);
  float* ping = top.take(static_cast<size_t>(n) * d);
  float* pong = top.take(static_cast<size_t>(n) * d);
  if (!emb_h || !ping || !pong) return LCR_ESPACE;
  int rc;
  // LinearLearnablePosEmbedding: two Linears, no activation (Rotary3DPosEmb.py:34-38) -> theta [n, d/2]; in_proj
  if ((rc = rf_linear(W->emb1, points, n, 3, d, emb_h, false, s))) return rc;
  if ((rc = rf_linear(W->emb2, emb_h, n, d, d / 2, theta_out, false, s))) return rc;
  if ((rc = rf_linear(W->in_proj, feats, n, W->d_in, d, ping, false, s))) return rc;
  float* cur = ping;
  float* nxt = pong;
  for (int i = 0; i < W->num_blocks; ++i) {
    const LcrRoformerLayerW& L = W->layers[i];
    if (W->block_is_self[i]) {
      if ((rc = rf_layer(L, d, W->heads, cur, n, cur, n, theta_out, cat, cat, 2 * P, nxt, top, s))) return rc;
    } else {
      // sequential cross (rpetransformer.py:213-214): cloud 0 attends to cloud 1, then cloud 1 to the UPDATED cloud 0
      if ((rc = rf_layer(L, d, W->heads, cur, n0, cur + n0 * d, n1, nullptr, lens0_host, lens1_host, P, nxt, top, s))) return rc;
      if ((rc = rf_layer(L, d, W->heads, cur + n0 * d, n1, nxt, n0, nullptr, lens1_host, lens0_host, P, nxt + n0 * d, top, s))) return rc;
    }
    std::swap(cur, nxt);
  }
  if ((rc = rf_linear(W->out_proj, cur, n, d, W->d_out, out, false, s))) return rc;
  return check_launch("lcr_roformer_forward");
}
