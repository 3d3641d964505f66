// Context, weight packing and stage orchestration (host side of the library).
#pragma once
#include <math.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/stylish_hip.h"
#include "common.h"
#include "elementwise.hip.h"
#include "gemm.hip.h"
#include "gemm16.hip.h"
#include "signal.hip.h"
#include "phoneme.hip.h"
#include "wn_layer.hip.h"
#include "wn_layer_small.hip.h"
#include "wn_fused.hip.h"
#include "wn_fused16.hip.h"
#include "wn_block16.hip.h"
#include "wn_fused_x3.hip.h"
#include "wn_block_x3.hip.h"
#include "winograd.hip.h"

namespace stts {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t dim(int i) const { return shape[i]; }
};

struct PackedConv {
  float* W = nullptr;
  unsigned short* W16 = nullptr;  // bf16 / fp16 copy (16-bit operand modes), same layout; fp32 mode: the three bf16 planes of the exact split (gemm.hip.h, PREC_X3)
  long w16_plane = 0;             // fp32 mode: elements between two of those planes (0: no split form)
  int prec = 0;                   // PREC_* the copy was rounded to
  float* bias = nullptr;
  int npad = 0, N = 0, kc = 0, ntaps = 1;
  int cin_real = 0, rows_real = 0;  // un-padded sizes (FLOP accounting)
};

// Winograd F(6, r) form of a 'same' conv (winograd.hip.h): the n weight planes G_j g as one packed tensor [n][npad][1][kc].
struct WinoConv {
  PackedConv planes;  // W: n planes of npad x kc (plane stride npad * kc); bias: the conv's own bias [npad]
  WinoMats mats;
  int pad = 0;
  bool ready = false;
};

// one entry of a style-projection table
struct StyleSlot {
  int col0 = 0, C = 0;
};

struct StyleTable {
  std::vector<float> hW, hb;  // host staging [J][K], [J]
  float* W = nullptr;
  float* b = nullptr;
  int J = 0, K = 64;
  int add(const HostTensor& w, const HostTensor& bias) {  // returns col0 (4-aligned)
    while (J % 4) {
      hW.insert(hW.end(), K, 0.f);
      hb.push_back(0.f);
      ++J;
    }
    const int c0 = J;
    hW.insert(hW.end(), w.data.begin(), w.data.end());
    hb.insert(hb.end(), bias.data.begin(), bias.data.end());
    J += (int)bias.data.size();
    return c0;
  }
  int ld() const { return round_up(J, 4); }
};

struct AdainBlockW {
  int cin = 0, cout = 0, kcin = 0;  // kcin = cin padded to 32
  PackedConv conv1, conv2, sc;      // sc.W == nullptr: identity shortcut
  WinoConv w1, w2;                  // conv1 / conv2 (identity-shortcut blocks only) in Winograd form (fp32 mode, k = 3), large batches
  StyleSlot n1, n2;
};

// fragment-order weights of one coupling layer for wn_fused_kernel (fp32 mode; wn_fused.hip.h)
struct WnFusedW {
  float* W1[3][4] = {};  // [0: F(2,5), 1: F(4,5), 2: direct (M = 1)][WaveNet layer]: transformed in_layers planes
  float* b1[4] = {};     // in_layers bias, natural order [256]
  float* W2[4] = {};     // res_skip_layers
  float* b2[4] = {};
  float* W3 = nullptr;   // post: proj_mean | proj_logstd
  float* b3m = nullptr;
  float* b3s = nullptr;
  float* W4 = nullptr;   // this layer's own `pre` (run by the tail of the layer before it in reverse order)
  float* b4 = nullptr;
  // 16-bit operand modes (wn_fused16_kernel): the same matrices as 16-bit fragments, conv in direct (tap-major) form
  unsigned short* H1[4] = {};
  unsigned short* H2[4] = {};
  unsigned short* H2b[4] = {};  // res_skip in wn_block16_kernel's tile order (wave w: res 32w.., skip 32w..; layer 3: skip only)
  unsigned short* H3 = nullptr;
  unsigned short* H4 = nullptr;
  // split fp32 (wn_fused_x3_kernel): the 16-bit kernels' fragment arrays as the three bf16 planes of the exact split, direct (tap-major) conv
  unsigned short* X1[4] = {};
  unsigned short* X2[4] = {};
  unsigned short* X2b[4] = {};  // res_skip in wn_block_x3_kernel's tile order (as H2b)
  unsigned short* X3 = nullptr;
  unsigned short* X4 = nullptr;
  long xp1 = 0, xp2[4] = {}, xp3 = 0, xp4 = 0;  // f32x4 units between two planes
  bool ready = false;    // fp32 fragments packed
  bool ready16 = false;  // 16-bit fragments packed
  bool ready_x3 = false; // split-fp32 fragments packed
};

struct FlowLayerW {
  PackedConv pre, in[4], rs[4], proj;
  WnFusedW fused;
  int cond_col0 = 0;
};

struct ConvNextW {
  float* dw_wt = nullptr;  // [K][C] tap-major
  float* dw_b = nullptr;
  int K = 0;
  StyleSlot norm;
  PackedConv pw1, pw2;  // pw2.bias already includes W2 @ grn.beta
  float* grn_gamma = nullptr;
};

struct MrfW {
  PackedConv c1[3], c2[3];
  StyleSlot a1[3], a2[3];
  float* alpha1[3];
  float* alpha2[3];
  int dil[3] = {1, 3, 5};
  int channels = 0, kernel = 0;
  StyleTable table;
};

}  // namespace stts

struct stts_ctx {
  stts_model_dims d;
  int device = 0;
  std::map<std::string, stts::HostTensor> host;
  std::vector<void*> allocs;
  std::vector<int> alloc_tag;  // STTS_W_* component a device allocation belongs to (0: context lifetime); same length as allocs
  int cur_tag = 0;             // tag of the allocations made right now (set by the finalize sections)
  int* d_err = nullptr;
  // side streams of stts_frame_path (fp32, large batches): one per caller stream (several host threads may run the path on their own streams)
  struct SideLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    // second side stream: short independent contractions inside a stage (the decoder blocks' learned shortcuts, stts::SideWork)
    hipStream_t stream2 = nullptr;
    hipEvent_t fork2 = nullptr, join2 = nullptr;
  };
  std::map<hipStream_t, SideLane> side_lanes;
  std::mutex side_mu;
  int ready = 0;  // STTS_W_* components finalized
  int prec = 0;   // contraction operand precision (stts::PREC_*), fixed before the first finalize
  bool allow_x3 = true;  // false: STTS_PREC_F32_NATIVE - fp32 contractions on the f32 matrix cores only
  bool pack_x3 = true;  // fp32 mode: pack_rows also writes the three bf16 planes of every weight (split-fp32 contractions, gemm.hip.h PREC_X3); off while the phoneme-rate / CFM models are packed
  int kc_align = 32;  // input channels of a packed conv are padded to this (64 while the frame path is packed for a 16-bit mode: conv_gemm16_kernel's K tile)
  // shared tables
  float* hann = nullptr;      // periodic Hann(win)
  float2* twiddle = nullptr;  // exp(-2 pi i m / n_fft), m < n_fft/2
  double2* twiddle64 = nullptr;
  // decoder
  float front_wf[3], front_bf, front_wn[3], front_bn;
  stts::PackedConv asr_res;
  stts::AdainBlockW dec[5];
  stts::StyleTable dec_style;
  // prior + flow
  stts::PackedConv prior, post_flow;
  stts::FlowLayerW flow[8];
  stts::StyleTable flow_style;
  // generator
  stts::PackedConv amp_prior, phase_prior, proj_mel, proj_la, proj_ph, amp_out, phase_out;  // *_out: the first n_fft/2 channels
  stts::WinoConv wino_prior[2], wino_out[2];  // the four k = 7 convs of the vocoder in Winograd form (fp32 mode)
  float* nyq_w[2] = {nullptr, nullptr};  // last (Nyquist) output channel of amp/phase output convs, [taps][cin]
  float nyq_b[2] = {0.f, 0.f};
  stts::ConvNextW cnx[4];
  stts::StyleSlot head_amp, head_phase;
  stts::StyleTable gen_style;
  // on-demand op caches (tests)
  std::map<std::string, std::unique_ptr<stts::AdainBlockW>> op_blocks;
  std::map<std::string, std::unique_ptr<stts::StyleTable>> op_tables;
  std::map<std::string, std::unique_ptr<stts::MrfW>> op_mrf;
  std::shared_ptr<void> phoneme;  // stts::PhonemeModel (phoneme_model.hip.h)
  std::shared_ptr<void> cfm;      // stts::CfmModel (cfm.hip.h)
};

namespace stts {

// ------------------------------------------------------------------------------------------------
// small host utilities
// ------------------------------------------------------------------------------------------------
template <typename T>
inline int dev_upload(stts_ctx* c, const std::vector<T>& h, T** out) {
  void* p = nullptr;
  STTS_HIP(hipMalloc(&p, std::max<size_t>(h.size(), 1) * sizeof(T)));
  c->allocs.push_back(p);
  c->alloc_tag.push_back(c->cur_tag);
  if (!h.empty()) STTS_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return 0;
}

// Re-finalizing a component (a shim re-bound with other weights) releases the device buffers of its previous packing.
inline void free_component_allocs(stts_ctx* c, int mask) {
  bool any = false;
  for (size_t i = 0; i < c->allocs.size(); ++i) any = any || (c->alloc_tag[i] & mask);
  if (!any) return;
  (void)hipDeviceSynchronize();  // kernels that read the old buffers may still be queued
  size_t k = 0;
  for (size_t i = 0; i < c->allocs.size(); ++i) {
    if (c->alloc_tag[i] & mask) {
      (void)hipFree(c->allocs[i]);
    } else {
      c->allocs[k] = c->allocs[i];
      c->alloc_tag[k++] = c->alloc_tag[i];
    }
  }
  c->allocs.resize(k);
  c->alloc_tag.resize(k);
}

inline const HostTensor* find(stts_ctx* c, const std::string& name) {
  auto it = c->host.find(name);
  return it == c->host.end() ? nullptr : &it->second;
}
#define STTS_GET(var, name)                                   \
  const stts::HostTensor* var = stts::find(c, (name));        \
  if (!var) return stts::fail("missing weight '%s'", std::string(name).c_str())

// w = g * v / ||v||, norm over all dims but 0 (torch weight_norm dim=0; reference: models/decoder.py:35-45
// parametrization keys original0/original1, models/flow.py:40,52,60 legacy keys weight_g/weight_v)
inline HostTensor fold_weight_norm(const HostTensor& g, const HostTensor& v) {
  HostTensor w;
  w.shape = v.shape;
  w.data.resize(v.data.size());
  const int64_t rows = v.shape[0], per = (int64_t)v.data.size() / rows;
  for (int64_t r = 0; r < rows; ++r) {
    double s = 0;
    for (int64_t i = 0; i < per; ++i) s += (double)v.data[r * per + i] * v.data[r * per + i];
    const float scale = g.data[r] / (float)sqrt(s);
    for (int64_t i = 0; i < per; ++i) w.data[r * per + i] = v.data[r * per + i] * scale;
  }
  return w;
}

// fetch a conv/linear weight by module path, folding whichever weight-norm flavour is present
inline int get_weight(stts_ctx* c, const std::string& p, HostTensor* w) {
  if (const HostTensor* g = find(c, p + ".parametrizations.weight.original0")) {
    STTS_GET(v, p + ".parametrizations.weight.original1");
    *w = fold_weight_norm(*g, *v);
  } else if (const HostTensor* g2 = find(c, p + ".weight_g")) {
    STTS_GET(v, p + ".weight_v");
    *w = fold_weight_norm(*g2, *v);
  } else {
    STTS_GET(v, p + ".weight");
    *w = *v;
  }
  if (w->shape.size() == 2) w->shape.push_back(1);  // Linear == conv k=1
  return 0;
}

// Pack rows of a [cout][cin][k] weight: out[n'][tap][ci] for ci in [0,kc) taking input channel cin_lo+ci.
// row_of[n'] = source row (or -1 for zero padding); bias follows the same row order.
// host-side round-to-nearest-even conversions for the 16-bit weight copies
inline unsigned short f32_to_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// exact three-term bf16 split of an fp32 value (gemm.hip.h, PREC_X3): f = p0 + p1 + p2, every term the RNE bf16 of what is left
inline void split3_host(float f, unsigned short* p0, unsigned short* p1, unsigned short* p2) {
  *p0 = f32_to_bf16(f);
  const float r1 = f - bf16_to_f32(*p0);
  *p1 = f32_to_bf16(r1);
  *p2 = f32_to_bf16(r1 - bf16_to_f32(*p1));
}
inline unsigned short f32_to_f16(float f) {
  const _Float16 h = (_Float16)f;  // IEEE RNE, saturates to inf
  unsigned short r;
  memcpy(&r, &h, 2);
  return r;
}

inline int pack_rows(stts_ctx* c, const HostTensor& w, const HostTensor* bias, const std::vector<int>& row_of, int cin_lo, int cin_n,
                     int kc, int N, PackedConv* out, float scale = 1.0f) {
  const int cin = (int)w.shape[1], k = (int)w.shape[2];
  const int npad = (int)row_of.size();
  std::vector<float> pw((size_t)npad * k * kc, 0.f), pb(npad, 0.f);
  for (int n = 0; n < npad; ++n) {
    const int r = row_of[n];
    if (r < 0) continue;
    for (int t = 0; t < k; ++t)
      for (int ci = 0; ci < cin_n; ++ci) pw[((size_t)n * k + t) * kc + ci] = w.data[((size_t)r * cin + cin_lo + ci) * k + t] * scale;
    if (bias) pb[n] = bias->data[r] * scale;
  }
  STTS_TRY(dev_upload(c, pw, &out->W));
  STTS_TRY(dev_upload(c, pb, &out->bias));
  out->prec = c->prec;
  out->W16 = nullptr;
  out->w16_plane = 0;
  if (c->prec != PREC_F32 || (c->allow_x3 && c->pack_x3 && x3_enabled())) {
    const bool split = c->prec == PREC_F32;
    std::vector<unsigned short> h(pw.size() * (split ? 3 : 1));
    if (split) {
      out->w16_plane = (long)pw.size();
      for (size_t i = 0; i < pw.size(); ++i) split3_host(pw[i], &h[i], &h[pw.size() + i], &h[2 * pw.size() + i]);
    } else
    for (size_t i = 0; i < pw.size(); ++i) h[i] = c->prec == PREC_BF16 ? f32_to_bf16(pw[i]) : f32_to_f16(pw[i]);
    void* d = nullptr;
    STTS_HIP(hipMalloc(&d, h.size() * sizeof(unsigned short)));
    c->allocs.push_back(d);
    c->alloc_tag.push_back(c->cur_tag);
    STTS_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    out->W16 = (unsigned short*)d;
  }
  out->npad = npad;
  out->N = N;
  out->kc = kc;
  out->ntaps = k;
  out->cin_real = cin_n;
  out->rows_real = 0;
  for (int r : row_of) out->rows_real += r >= 0;
  return 0;
}

inline std::vector<int> plain_rows(int cout) {
  std::vector<int> r(round_up(cout, 128), -1);
  for (int i = 0; i < cout; ++i) r[i] = i;
  return r;
}
// paired packing: result channel ch has its 'a' row at a_lo+ch and its 'b' row at b_lo+ch; within every 64 packed rows
// the first 32 are 'a' rows and the next 32 the matching 'b' rows (see EPI_GATE in gemm.hip.h).
inline std::vector<int> paired_rows(int nres, int a_lo, int b_lo) {
  std::vector<int> r(round_up(2 * nres, 128), -1);
  for (int ch = 0; ch < nres; ++ch) {
    const int g = ch / 32, q = ch % 32;
    r[g * 64 + q] = a_lo + ch;
    r[g * 64 + 32 + q] = b_lo + ch;
  }
  return r;
}

inline int pack_plain(stts_ctx* c, const std::string& p, bool has_bias, int cin_lo, int cin_n, PackedConv* out, float scale = 1.0f) {
  HostTensor w;
  STTS_TRY(get_weight(c, p, &w));
  const HostTensor* b = has_bias ? find(c, p + ".bias") : nullptr;
  if (has_bias && !b) return fail("missing weight '%s.bias'", p.c_str());
  if (cin_n < 0) cin_n = (int)w.shape[1] - cin_lo;
  return pack_rows(c, w, b, plain_rows((int)w.shape[0]), cin_lo, cin_n, round_up(cin_n, c->kc_align), (int)w.shape[0], out, scale);
}

inline int upload_table(stts_ctx* c, StyleTable* t) {
  while (t->J % 4) {
    t->hW.insert(t->hW.end(), t->K, 0.f);
    t->hb.push_back(0.f);
    ++t->J;
  }
  STTS_TRY(dev_upload(c, t->hW, &t->W));
  STTS_TRY(dev_upload(c, t->hb, &t->b));
  return 0;
}

inline int add_style(stts_ctx* c, StyleTable* t, const std::string& p, int C, StyleSlot* slot) {
  STTS_GET(w, p + ".fc.weight");
  STTS_GET(b, p + ".fc.bias");
  STTS_CHECK((int)b->data.size() == 2 * C, "%s.fc has %zu outputs, expected %d", p.c_str(), b->data.size(), 2 * C);
  slot->col0 = t->add(*w, *b);
  slot->C = C;
  return 0;
}

inline int pack_winograd(stts_ctx* c, const HostTensor& w, const HostTensor* bias, int cin_lo, int cin_n, int cout_used, WinoConv* out) {
  static const bool disabled = getenv("STTS_NO_WINOGRAD") != nullptr;  // debugging aid: every conv then runs in its direct form
  if (disabled) return 0;  // out->ready stays false
  const int cin = (int)w.shape[1], r = (int)w.shape[2];
  STTS_CHECK(wino_matrices(r, &out->mats), "winograd: unsupported kernel size %d (or self-check failed)", r);
  const int n = out->mats.n, npad = round_up(cout_used, 128), kc = round_up(cin_n, 32);
  HostTensor wp;
  wp.shape = {(int64_t)n * cout_used, cin, 1};
  wp.data.assign((size_t)n * cout_used * cin, 0.f);
  for (int j = 0; j < n; ++j)
    for (int co = 0; co < cout_used; ++co)
      for (int ci = 0; ci < cin; ++ci) {
        double acc = 0;
        for (int k = 0; k < r; ++k) acc += out->mats.G[j][k] * (double)w.data[((size_t)co * cin + ci) * r + k];
        wp.data[((size_t)j * cout_used + co) * cin + ci] = (float)acc;
      }
  std::vector<int> row_of((size_t)n * npad, -1);
  for (int j = 0; j < n; ++j)
    for (int co = 0; co < cout_used; ++co) row_of[(size_t)j * npad + co] = j * cout_used + co;
  const int saved_prec = c->prec;
  c->prec = PREC_F32;  // the transformed weights span ~3 decades (G up to 729): fp32 operands only
  const int rc = pack_rows(c, wp, nullptr, row_of, cin_lo, cin_n, kc, cout_used, &out->planes);
  c->prec = saved_prec;
  STTS_TRY(rc);
  out->planes.npad = npad;
  out->planes.cin_real = cin_n;
  out->planes.rows_real = cout_used;
  if (bias) {
    std::vector<float> pb(npad, 0.f);
    for (int co = 0; co < cout_used; ++co) pb[co] = bias->data[co];
    STTS_TRY(dev_upload(c, pb, &out->planes.bias));
  } else {
    out->planes.bias = nullptr;
  }
  out->pad = (r - 1) / 2;
  out->ready = true;
  return 0;
}

inline int pack_adain_block(stts_ctx* c, const std::string& p, int cin, int cout, StyleTable* table, AdainBlockW* o) {
  o->cin = cin;
  o->cout = cout;
  o->kcin = round_up(cin, c->kc_align);
  STTS_TRY(pack_plain(c, p + ".conv1", true, 0, cin, &o->conv1));
  o->w1 = WinoConv();
  if (c->prec == PREC_F32) {
    HostTensor w1;
    STTS_TRY(get_weight(c, p + ".conv1", &w1));
    if (w1.shape[2] == 3) STTS_TRY(pack_winograd(c, w1, find(c, p + ".conv1.bias"), 0, cin, cout, &o->w1));
  }
  STTS_TRY(pack_plain(c, p + ".conv2", true, 0, cout, &o->conv2));
  o->w2 = WinoConv();
  if (find(c, p + ".conv1x1.parametrizations.weight.original0") || find(c, p + ".conv1x1.weight")) STTS_TRY(pack_plain(c, p + ".conv1x1", false, 0, cin, &o->sc));
  if (c->prec == PREC_F32) {  // conv2 is a plain conv + residual (the block's input, or its learned 1x1 shortcut computed first): it has a Winograd form too
    HostTensor w2;
    STTS_TRY(get_weight(c, p + ".conv2", &w2));
    if (w2.shape[2] == 3) STTS_TRY(pack_winograd(c, w2, find(c, p + ".conv2.bias"), 0, cout, cout, &o->w2));
  }
  STTS_TRY(add_style(c, table, p + ".norm1", cin, &o->n1));
  STTS_TRY(add_style(c, table, p + ".norm2", cout, &o->n2));
  return 0;
}

// wn_fused_kernel operands of coupling layer `q` (prefix "...flow.flows.N."): every matrix in MFMA-fragment order
// (wn_fused.hip.h: pack_fragments), the k = 5 conv as F(2,5) and F(4,5) planes G_j g computed in double.
inline int pack_wn_fused(stts_ctx* c, const std::string& q, const HostTensor& pm, const HostTensor& pl, const HostTensor& pmb, const HostTensor& plb,
                         WnFusedW* o) {
  const int C = kWnC;
  auto rows_of = [](const HostTensor& w) {  // [rows][K] (k = 1) as double
    const int rows = (int)w.shape[0], K = (int)(w.data.size() / rows);
    std::vector<std::vector<double>> r(rows, std::vector<double>(K));
    for (int n = 0; n < rows; ++n)
      for (int k = 0; k < K; ++k) r[n][k] = w.data[(size_t)n * K + k];
    return r;
  };
  for (int i = 0; i < 4; ++i) {
    HostTensor w, wr;
    STTS_TRY(get_weight(c, q + "enc.in_layers." + std::to_string(i), &w));
    STTS_GET(b, q + "enc.in_layers." + std::to_string(i) + ".bias");
    STTS_CHECK(w.shape[0] == 2 * C && w.shape[1] == C && w.shape[2] == 5 && (int)b->data.size() == 2 * C, "wn_fused: in_layers.%d has an unexpected shape", i);
    if (c->prec != PREC_F32) {
      // direct form, K-major rows [tap][cin] so that k-step s = tap * 4 + (cin / 32) covers K offsets [32 s, 32 s + 32)
      std::vector<std::vector<float>> kr((size_t)2 * C, std::vector<float>(5 * C));
      for (int n = 0; n < 2 * C; ++n)
        for (int ci = 0; ci < C; ++ci)
          for (int k = 0; k < 5; ++k) kr[n][(size_t)k * C + ci] = w.data[((size_t)n * C + ci) * 5 + k];
      // wave w, tile (half h, c): output rows h * 128 + 32 w + 16 c + col
      const std::vector<unsigned short> f16v = pack_fragments16(c->prec, kWnWaves, 5 * C / 32, 4, [&](int wv, int t, int col) {
        return kr[(size_t)(t >> 1) * C + 32 * wv + 16 * (t & 1) + col].data();
      }, f32_to_bf16, f32_to_f16);
      STTS_TRY(dev_upload(c, f16v, &o->H1[i]));
    }
    const bool x3pack = c->prec == PREC_F32 && c->allow_x3 && c->pack_x3 && x3_enabled();
    if (x3pack) {
      std::vector<std::vector<float>> kr((size_t)2 * C, std::vector<float>(5 * C));
      for (int n = 0; n < 2 * C; ++n)
        for (int ci = 0; ci < C; ++ci)
          for (int k = 0; k < 5; ++k) kr[n][(size_t)k * C + ci] = w.data[((size_t)n * C + ci) * 5 + k];
      const std::vector<unsigned short> fx = pack_fragments_x3(kWnWaves, 5 * C / 32, 4, [&](int wv, int t, int col) {
        return kr[(size_t)(t >> 1) * C + 32 * wv + 16 * (t & 1) + col].data();
      }, split3_host);
      STTS_TRY(dev_upload(c, fx, &o->X1[i]));
      o->xp1 = (long)(fx.size() / 3 / 8);
    }
    for (int v = 0; v < 3 && c->prec == PREC_F32; ++v) {
      WnFusedMats mt;
      const int vm = v == 0 ? 2 : (v == 1 ? 4 : 1);
      STTS_CHECK(wn_fused_matrices(vm, &mt), "wn_fused: F(%d,5) matrices failed their self-check", vm);
      const int nc = mt.n;
      std::vector<std::vector<double>> plane((size_t)nc * 2 * C, std::vector<double>(C));  // [component][output row][cin]
      for (int j = 0; j < nc; ++j)
        for (int n = 0; n < 2 * C; ++n)
          for (int ci = 0; ci < C; ++ci) {
            double acc = 0;
            for (int k = 0; k < 5; ++k) acc += mt.G[j][k] * (double)w.data[((size_t)n * C + ci) * 5 + k];
            plane[(size_t)j * 2 * C + n][ci] = acc;
          }
      // wave w, tile ((component j, half h), c): output rows h * 128 + 32 w + 16 c + col  (tanh rows 0..127 | sigmoid rows 128..255)
      const std::vector<float> f = pack_fragments(kWnWaves, C / 16, nc * 4, [&](int wv, int t, int col) {
        return plane[(size_t)(t >> 2) * 2 * C + ((t >> 1) & 1) * C + 32 * wv + 16 * (t & 1) + col].data();
      });
      STTS_TRY(dev_upload(c, f, &o->W1[v][i]));
    }
    STTS_TRY(dev_upload(c, b->data, &o->b1[i]));
    STTS_TRY(get_weight(c, q + "enc.res_skip_layers." + std::to_string(i), &wr));
    STTS_GET(br, q + "enc.res_skip_layers." + std::to_string(i) + ".bias");
    const int n_rs = (int)wr.shape[0];
    STTS_CHECK((n_rs == 2 * C || n_rs == C) && wr.shape[1] == C && (n_rs == C) == (i == 3), "wn_fused: res_skip_layers.%d has an unexpected shape", i);
    const auto rr = rows_of(wr);
    const int nct = n_rs / 16 / kWnWaves;  // column tiles per wave
    if (c->prec == PREC_F32) {
      const std::vector<float> f2 = pack_fragments(kWnWaves, C / 16, nct, [&](int wv, int t, int col) { return rr[(size_t)16 * nct * wv + 16 * t + col].data(); });
      STTS_TRY(dev_upload(c, f2, &o->W2[i]));
      if (x3pack) {
        const std::vector<unsigned short> fx = pack_fragments_x3(kWnWaves, C / 32, nct, [&](int wv, int t, int col) {
          return wr.data.data() + ((size_t)16 * nct * wv + 16 * t + col) * C;
        }, split3_host);
        STTS_TRY(dev_upload(c, fx, &o->X2[i]));
        o->xp2[i] = (long)(fx.size() / 3 / 8);
        // wn_block_x3_kernel: wave w owns the res rows AND the skip rows [32 w, 32 w + 32) (tiles: res, res + 16, skip, skip + 16; layer 3: skip, skip + 16)
        const int nctb = n_rs == 2 * C ? 4 : 2;
        const std::vector<unsigned short> fxb = pack_fragments_x3(kWnWaves, C / 32, nctb, [&](int wv, int t, int col) {
          const int row = n_rs == 2 * C ? (t < 2 ? 32 * wv + 16 * t + col : C + 32 * wv + 16 * (t - 2) + col) : 32 * wv + 16 * t + col;
          return wr.data.data() + (size_t)row * C;
        }, split3_host);
        STTS_TRY(dev_upload(c, fxb, &o->X2b[i]));  // (same size as X2[i]: xp2[i] is its plane stride too)
      }
    } else {
      const std::vector<unsigned short> f2 = pack_fragments16(c->prec, kWnWaves, C / 32, nct, [&](int wv, int t, int col) {
        return wr.data.data() + ((size_t)16 * nct * wv + 16 * t + col) * C;
      }, f32_to_bf16, f32_to_f16);
      STTS_TRY(dev_upload(c, f2, &o->H2[i]));
      // wn_block16_kernel: wave w owns the res rows AND the skip rows [32 w, 32 w + 32) (tiles: res, res + 16, skip, skip + 16; layer 3: skip, skip + 16)
      const int nctb = n_rs == 2 * C ? 4 : 2;
      const std::vector<unsigned short> f2b = pack_fragments16(c->prec, kWnWaves, C / 32, nctb, [&](int wv, int t, int col) {
        const int row = n_rs == 2 * C ? (t < 2 ? 32 * wv + 16 * t + col : C + 32 * wv + 16 * (t - 2) + col) : 32 * wv + 16 * t + col;
        return wr.data.data() + (size_t)row * C;
      }, f32_to_bf16, f32_to_f16);
      STTS_TRY(dev_upload(c, f2b, &o->H2b[i]));
    }
    STTS_TRY(dev_upload(c, br->data, &o->b2[i]));
  }
  STTS_CHECK(pm.shape[0] == C / 2 && pm.shape[1] == C && pl.shape[0] == C / 2, "wn_fused: proj has an unexpected shape");
  const auto rm = rows_of(pm), rl = rows_of(pl);
  if (c->prec == PREC_F32) {
    const std::vector<float> f3 = pack_fragments(kWnWaves, C / 16, 2, [&](int wv, int t, int col) { return (t == 0 ? rm : rl)[(size_t)16 * wv + col].data(); });
    STTS_TRY(dev_upload(c, f3, &o->W3));
    if (c->allow_x3 && c->pack_x3 && x3_enabled()) {
      const std::vector<unsigned short> fx = pack_fragments_x3(kWnWaves, C / 32, 2, [&](int wv, int t, int col) {
        return (t == 0 ? pm : pl).data.data() + ((size_t)16 * wv + col) * C;
      }, split3_host);
      STTS_TRY(dev_upload(c, fx, &o->X3));
      o->xp3 = (long)(fx.size() / 3 / 8);
    }
  } else {
    const std::vector<unsigned short> f3 = pack_fragments16(c->prec, kWnWaves, C / 32, 2, [&](int wv, int t, int col) {
      return (t == 0 ? pm : pl).data.data() + ((size_t)16 * wv + col) * C;
    }, f32_to_bf16, f32_to_f16);
    STTS_TRY(dev_upload(c, f3, &o->H3));
  }
  STTS_TRY(dev_upload(c, pmb.data, &o->b3m));
  STTS_TRY(dev_upload(c, plb.data, &o->b3s));
  HostTensor wp;
  STTS_TRY(get_weight(c, q + "pre", &wp));
  STTS_GET(bp, q + "pre.bias");
  STTS_CHECK(wp.shape[0] == C && wp.shape[1] == C / 2, "wn_fused: pre has an unexpected shape");
  const auto rp = rows_of(wp);
  if (c->prec == PREC_F32) {
    const std::vector<float> f4 = pack_fragments(kWnWaves, C / 32, 2, [&](int wv, int t, int col) { return rp[(size_t)32 * wv + 16 * t + col].data(); });
    STTS_TRY(dev_upload(c, f4, &o->W4));
    if (c->allow_x3 && c->pack_x3 && x3_enabled()) {
      const std::vector<unsigned short> fx = pack_fragments_x3(kWnWaves, C / 64, 2, [&](int wv, int t, int col) {
        return wp.data.data() + ((size_t)32 * wv + 16 * t + col) * (C / 2);
      }, split3_host);
      STTS_TRY(dev_upload(c, fx, &o->X4));
      o->xp4 = (long)(fx.size() / 3 / 8);
    }
  } else {
    const std::vector<unsigned short> f4 = pack_fragments16(c->prec, kWnWaves, C / 64, 2, [&](int wv, int t, int col) {
      return wp.data.data() + ((size_t)32 * wv + 16 * t + col) * (C / 2);
    }, f32_to_bf16, f32_to_f16);
    STTS_TRY(dev_upload(c, f4, &o->H4));
  }
  STTS_TRY(dev_upload(c, bp->data, &o->b4));
  o->ready = c->prec == PREC_F32;
  o->ready16 = c->prec != PREC_F32;
  o->ready_x3 = c->prec == PREC_F32 && c->allow_x3 && c->pack_x3 && x3_enabled();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// finalize: frame-rate path
// ------------------------------------------------------------------------------------------------
inline int finalize_frame(stts_ctx* c, int which) {
  const stts_model_dims& d = c->d;
  STTS_CHECK(d.n_fft == kNfft && d.win_length == kWin && d.hop_length / 4 == kHop && d.sample_rate == 24000,
             "this build is specialised for n_fft 2048 / win 1200 / hop 300 / 24 kHz (model.yml defaults)");
  // channel sizes come from the model config (lib/config_loader.py:369-414); what the kernels need: 16-byte rows and column
  // offsets (multiples of 32 for the concatenated widths) and the generator reading the decoder's width
  STTS_CHECK(d.style_dim > 0 && d.style_dim % 4 == 0 && d.inter_dim > 0 && d.inter_dim % 4 == 0, "style_dim / inter_dim must be multiples of 4");
  STTS_CHECK(d.dec_hidden > 0 && d.dec_hidden % 32 == 0 && d.dec_residual >= 0 && d.dec_residual % 4 == 0,
             "decoder.hidden_dim must be a multiple of 32 (the flow runs on hidden_dim / 4 channels split in two halves), residual_dim a multiple of 4 (16-byte columns)");
  STTS_CHECK(d.gen_input == d.dec_hidden, "generator.input_dim (%d) must equal decoder.hidden_dim (%d): post_flow feeds the generator", d.gen_input, d.dec_hidden);
  STTS_CHECK(d.gen_hidden > 0 && d.gen_hidden % 32 == 0 && d.gen_inter > 0 && d.gen_inter % 32 == 0,
             "generator.hidden_dim / conv_intermediate_dim must be multiples of 32");
  const std::string sp = "speech_predictor.";
  // 16-bit operand modes: input channels padded to 64 (the K tile of conv_gemm16_kernel); the stages size their rows from the packed kc
  struct AlignReset {
    stts_ctx* c;
    ~AlignReset() { c->kc_align = 32; }
  } align_reset{c};
  c->kc_align = c->prec != PREC_F32 ? 64 : 32;
  // tables
  c->cur_tag = 0;
  if (!c->hann) {
    std::vector<float> h(kWin);
    for (int i = 0; i < kWin; ++i) h[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / kWin));
    STTS_TRY(dev_upload(c, h, &c->hann));
    std::vector<float2> tw(kNfft / 2);
    for (int i = 0; i < kNfft / 2; ++i) tw[i] = make_float2((float)cos(2.0 * M_PI * i / kNfft), (float)-sin(2.0 * M_PI * i / kNfft));
    STTS_TRY(dev_upload(c, tw, &c->twiddle));
    std::vector<double2> tw64(kNfft / 2);
    for (int i = 0; i < kNfft / 2; ++i) tw64[i] = make_double2(cos(2.0 * M_PI * i / kNfft), -sin(2.0 * M_PI * i / kNfft));
    STTS_TRY(dev_upload(c, tw64, &c->twiddle64));
  }
  // decoder (models/decoder.py:6-45)
  if (which & STTS_W_DECODER) {
    c->cur_tag = STTS_W_DECODER;
    c->dec_style = StyleTable();
    c->dec_style.K = d.style_dim;
    HostTensor wf, wn;
    STTS_TRY(get_weight(c, sp + "decoder.F0_conv", &wf));
    STTS_TRY(get_weight(c, sp + "decoder.N_conv", &wn));
    STTS_GET(bf, sp + "decoder.F0_conv.bias");
    STTS_GET(bn, sp + "decoder.N_conv.bias");
    for (int i = 0; i < 3; ++i) {
      c->front_wf[i] = wf.data[i];
      c->front_wn[i] = wn.data[i];
    }
    c->front_bf = bf->data[0];
    c->front_bn = bn->data[0];
    STTS_TRY(pack_plain(c, sp + "decoder.asr_res.0", true, 0, d.inter_dim, &c->asr_res));
    STTS_TRY(pack_adain_block(c, sp + "decoder.encode", d.inter_dim + 2, d.dec_hidden, &c->dec_style, &c->dec[0]));
    for (int i = 0; i < 4; ++i)
      STTS_TRY(pack_adain_block(c, sp + "decoder.decode." + std::to_string(i), d.dec_hidden + 2 + d.dec_residual, d.dec_hidden,
                                &c->dec_style, &c->dec[i + 1]));
    STTS_TRY(upload_table(c, &c->dec_style));
  }
  // prior + flow + post_flow (models/flow.py, models/speech_predictor.py:36-62)
  if (which & STTS_W_FLOW) {
    c->cur_tag = STTS_W_FLOW;
    c->flow_style = StyleTable();
    c->flow_style.K = d.style_dim;
    const int fh = d.dec_hidden / 4, half = fh / 2;
    HostTensor wm, wl;
    STTS_TRY(get_weight(c, sp + "prior_encoder.proj_mean", &wm));
    STTS_TRY(get_weight(c, sp + "prior_encoder.proj_logstd", &wl));
    STTS_GET(bm, sp + "prior_encoder.proj_mean.bias");
    STTS_GET(bl, sp + "prior_encoder.proj_logstd.bias");
    HostTensor wcat = wm, bcat = *bm;
    wcat.data.insert(wcat.data.end(), wl.data.begin(), wl.data.end());
    wcat.shape[0] = 2 * fh;
    bcat.data.insert(bcat.data.end(), bl->data.begin(), bl->data.end());
    STTS_TRY(pack_rows(c, wcat, &bcat, paired_rows(fh, 0, fh), 0, d.dec_hidden, d.dec_hidden, fh, &c->prior));
    for (int f = 0; f < 8; ++f) {
      const std::string q = sp + "flow.flows." + std::to_string(2 * f) + ".";
      FlowLayerW& L = c->flow[f];
      STTS_TRY(pack_plain(c, q + "pre", true, 0, half, &L.pre));
      for (int i = 0; i < 4; ++i) {
        HostTensor w;
        STTS_TRY(get_weight(c, q + "enc.in_layers." + std::to_string(i), &w));
        STTS_GET(b, q + "enc.in_layers." + std::to_string(i) + ".bias");
        STTS_TRY(pack_rows(c, w, b, paired_rows(fh, 0, fh), 0, fh, fh, fh, &L.in[i]));
        STTS_TRY(pack_plain(c, q + "enc.res_skip_layers." + std::to_string(i), true, 0, fh, &L.rs[i]));
      }
      HostTensor pm, pl;
      STTS_TRY(get_weight(c, q + "proj_mean", &pm));
      STTS_TRY(get_weight(c, q + "proj_logstd", &pl));
      STTS_GET(pmb, q + "proj_mean.bias");
      STTS_GET(plb, q + "proj_logstd.bias");
      HostTensor pc = pm, pcb = *pmb;
      pc.data.insert(pc.data.end(), pl.data.begin(), pl.data.end());
      pc.shape[0] = 2 * half;
      pcb.data.insert(pcb.data.end(), plb->data.begin(), plb->data.end());
      STTS_TRY(pack_rows(c, pc, &pcb, paired_rows(half, 0, half), 0, fh, fh, half, &L.proj));
      L.fused = WnFusedW();
      if (fh == kWnC && !getenv("STTS_NO_WN_FUSED")) STTS_TRY(pack_wn_fused(c, q, pm, pl, *pmb, *plb, &L.fused));
      HostTensor cw;
      STTS_TRY(get_weight(c, q + "enc.cond_layer", &cw));
      STTS_GET(cb, q + "enc.cond_layer.bias");
      L.cond_col0 = c->flow_style.add(cw, *cb);
    }
    STTS_TRY(upload_table(c, &c->flow_style));
    STTS_TRY(pack_plain(c, sp + "post_flow", true, 0, fh, &c->post_flow));
  }
  // generator (models/generator.py:340-438)
  if (which & STTS_W_GENERATOR) {
    c->cur_tag = STTS_W_GENERATOR;
    c->gen_style = StyleTable();
    c->gen_style.K = d.style_dim;
    const std::string g = sp + "generator.";
    const int h = d.gen_hidden, hp = h / 2;
    STTS_TRY(pack_plain(c, g + "amp_prior_conv", true, 0, kBins, &c->amp_prior));
    STTS_TRY(pack_plain(c, g + "phase_prior_conv", true, 0, kBins, &c->phase_prior));
    for (int q = 0; q < 2; ++q) c->wino_prior[q] = c->wino_out[q] = WinoConv();
    if (c->prec == PREC_F32) {
      for (int q = 0; q < 2; ++q) {
        const std::string nm = g + (q == 0 ? "amp_prior_conv" : "phase_prior_conv");
        HostTensor wq;
        STTS_TRY(get_weight(c, nm, &wq));
        if (wq.shape[2] == 7) STTS_TRY(pack_winograd(c, wq, find(c, nm + ".bias"), 0, kBins, hp, &c->wino_prior[q]));
      }
    }
    STTS_TRY(pack_plain(c, g + "projector", true, 0, d.gen_input, &c->proj_mel));
    STTS_TRY(pack_plain(c, g + "projector", false, d.gen_input, hp, &c->proj_la));
    STTS_TRY(pack_plain(c, g + "projector", false, d.gen_input + hp, hp, &c->proj_ph));
    // output convs: n_fft/2 + 1 = 8*128 + 1 channels -> GEMM for the first 1024, a dot-product kernel for the last
    for (int which = 0; which < 2; ++which) {
      const std::string nm = g + (which == 0 ? "amp_output_conv" : "phase_output_conv");
      HostTensor w;
      STTS_TRY(get_weight(c, nm, &w));
      STTS_GET(b, nm + ".bias");
      const int nmain = kBins - 1, ci = (int)w.shape[1], kk = (int)w.shape[2];
      STTS_CHECK((int)w.shape[0] == kBins && ci == h + hp, "%s: unexpected shape", nm.c_str());
      STTS_TRY(pack_rows(c, w, b, plain_rows(nmain), 0, ci, round_up(ci, 32), nmain, which == 0 ? &c->amp_out : &c->phase_out));
      if (c->prec == PREC_F32 && kk == 7) STTS_TRY(pack_winograd(c, w, b, 0, ci, nmain, &c->wino_out[which]));
      std::vector<float> last((size_t)kk * ci);
      for (int t = 0; t < kk; ++t)
        for (int q = 0; q < ci; ++q) last[(size_t)t * ci + q] = w.data[((size_t)nmain * ci + q) * kk + t];
      STTS_TRY(dev_upload(c, last, &c->nyq_w[which]));
      c->nyq_b[which] = b->data[nmain];
    }
    const int ks[4] = {31, 15, 7, 3};
    for (int i = 0; i < 4; ++i) {
      const std::string q = g + "convnext." + std::to_string(i) + ".";
      ConvNextW& B = c->cnx[i];
      STTS_GET(dw, q + "dwconv.weight");
      STTS_GET(db, q + "dwconv.bias");
      STTS_CHECK(dw->shape[2] == ks[i], "convnext.%d dwconv kernel %lld != %d", i, (long long)dw->shape[2], ks[i]);
      B.K = ks[i];
      std::vector<float> wt((size_t)B.K * h);
      for (int ch = 0; ch < h; ++ch)
        for (int k = 0; k < B.K; ++k) wt[(size_t)k * h + ch] = dw->data[(size_t)ch * B.K + k];
      STTS_TRY(dev_upload(c, wt, &B.dw_wt));
      STTS_TRY(dev_upload(c, db->data, &B.dw_b));
      STTS_TRY(add_style(c, &c->gen_style, q + "norm", h, &B.norm));
      STTS_TRY(pack_plain(c, q + "pwconv1", true, 0, h, &B.pw1));
      // pwconv2 with GRN's beta folded into the bias: W2 (U*s + beta) + b2 = (W2*s) U + (W2 beta + b2)
      HostTensor w2;
      STTS_TRY(get_weight(c, q + "pwconv2", &w2));
      STTS_GET(b2, q + "pwconv2.bias");
      STTS_GET(gg, q + "grn.gamma");
      STTS_GET(gb, q + "grn.beta");
      HostTensor b2f = *b2;
      const int ci = (int)w2.shape[1];
      for (int r = 0; r < (int)w2.shape[0]; ++r) {
        double s = 0;
        for (int k = 0; k < ci; ++k) s += (double)w2.data[(size_t)r * ci + k] * gb->data[k];
        b2f.data[r] = (float)((double)b2->data[r] + s);
      }
      STTS_TRY(pack_rows(c, w2, &b2f, plain_rows((int)w2.shape[0]), 0, ci, round_up(ci, 32), (int)w2.shape[0], &B.pw2));
      STTS_TRY(dev_upload(c, gg->data, &B.grn_gamma));
    }
    STTS_TRY(add_style(c, &c->gen_style, g + "amp_final_layer_norm", h, &c->head_amp));
    STTS_TRY(add_style(c, &c->gen_style, g + "phase_final_layer_norm", h, &c->head_phase));
    STTS_TRY(upload_table(c, &c->gen_style));
  }
  c->cur_tag = 0;
  c->ready |= which & (STTS_W_DECODER | STTS_W_FLOW | STTS_W_GENERATOR);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// workspace bump allocator (caller-owned memory)
// ------------------------------------------------------------------------------------------------
// Workspace sizing by dry run (stts_frame_workspace_bytes): the stages carve their buffers from an Arena over a fake address
// range, stop before their first launch and report how far they got - so the bound is what the stages really request,
// for any model dims, instead of a hand-derived closed form.
struct DryRun {
  bool on = false;
  size_t peak = 0;
};
inline DryRun& dry_run() {
  static thread_local DryRun d;
  return d;
}
#define STTS_DRY_RETURN(arena)                                                   \
  do {                                                                            \
    if (stts::dry_run().on) {                                                     \
      stts::dry_run().peak = std::max(stts::dry_run().peak, (arena).offset0 + (arena).used); \
      return 0;                                                                   \
    }                                                                             \
  } while (0)

struct Arena {
  char* base;
  size_t cap, used = 0;
  size_t offset0 = 0;  // where this arena starts inside the caller's workspace (dry-run accounting)
  bool ok = true;
  Arena(void* p, size_t n, size_t off0 = 0) : base((char*)p), cap(n), offset0(off0) {}
  template <typename T>
  T* get(size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
    if (used + bytes > cap) {
      ok = false;
      return nullptr;
    }
    T* p = (T*)(base + used);
    used += bytes;
    return p;
  }
};

struct Seg {
  int n_utt;
  const int* host;
  const int* dev;
  // capacity segments: `host` holds UPPER BOUNDS (cumulative capacities; every buffer and grid is sized by them), `dev` the real
  // offsets, known on the device only (frame_offsets_kernel).  The rows of utterance u are [dev[u], dev[u+1]) - packed, so the
  // rows in [dev[n_utt], host[n_utt]) are unused.  dev[u] <= host[u] for every u.
  bool cap = false;
  const int* rows_dev() const { return cap ? dev + n_utt : nullptr; }  // the real row count, for kernels that walk all rows
  int rows() const { return host[n_utt]; }
  int max_len() const {
    int m = 0;
    for (int i = 0; i < n_utt; ++i) m = std::max(m, host[i + 1] - host[i]);
    return m;
  }
};

inline GemmArgs gemm_args(const Seg& s) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.seg_off = s.dev;
  a.seg_host = s.host;
  a.n_utt = s.n_utt;
  a.rows_total = s.rows();
  a.capacity = s.cap;
  if (!s.cap && s.host && s.n_utt > 0) {  // equal lengths (known on the host): the kernels' tile lookup is arithmetic
    const int len = s.host[1] - s.host[0];
    bool same = len > 0;
    for (int u = 1; u < s.n_utt && same; ++u) same = s.host[u + 1] - s.host[u] == len;
    if (same) { a.uniform_len = len; a.uniform_lo0 = s.host[0]; }
  }
  a.alpha = 1.0f;
  a.xaff_slope = 0.2f;
  a.zeros = zero_page();
  return a;
}
inline void set_seg(GemmArgs& a, int i, const float* X, int ldx, int xcol0, const PackedConv& w, int pad = -1, int dil = 1) {
  GemmSeg& g = a.seg[i];
  g.X = X;
  g.W = w.W;
  g.W16 = w.W16;
  g.w16_plane = w.w16_plane;
  if (i == 0) a.prec = w.prec;
  g.w_utt_stride = 0;
  g.ldx = ldx;
  g.xcol0 = xcol0;
  g.kc = w.kc;
  g.ntaps = w.ntaps;
  g.dil = dil;
  g.pad = pad >= 0 ? pad : (w.ntaps - 1) / 2;
  g.kreal = w.cin_real;
  if (i == 0) a.wrows = w.rows_real;
  if (a.nseg < i + 1) a.nseg = i + 1;
}

// Y = conv(X) through the Winograd form (winograd.hip.h).  Scratch: wino_scratch_floats(s, wc) floats; its head holds the
// group offsets of the batch (wino_setup_kernel), computed once per scratch and shared by every conv that uses it.
struct WinoScratch {
  float* p = nullptr;
  bool setup = false;
  explicit operator bool() const { return p != nullptr; }
};
inline size_t wino_scratch_floats(const Seg& s, const WinoConv& wc) {
  const long pr = wino_plane_rows(s.rows(), s.n_utt);
  // (input planes: fp32, or the three bf16 planes of the split form - 6 bytes per element - when the contraction runs as split fp32)
  return (size_t)wc.mats.n * pr * (wc.planes.kc * 3 / 2 + round_up(wc.planes.N, 32)) + round_up(s.n_utt + 1 + 16, 32) + 32;
}
template <int N>
inline int run_winograd_n(hipStream_t st, const Seg& s, const float* X, int ldx, const WinoConv& wc, float* Y, int ldy, int act, const float* R, int ldr,
                          float alpha, WinoScratch& scratch, const float* aff = nullptr, int ld_aff = 0, float* stat = nullptr, int ld_stat = 0) {
  const int n = wc.mats.n, kc = wc.planes.kc, ldm = round_up(wc.planes.N, 32), ml = s.max_len();
  long pr = 0;  // rows of a component plane: the groups of 4 output rows of all utterances, packed
  for (int u = 0; u < s.n_utt; ++u) pr += ceil_div(s.host[u + 1] - s.host[u], kWinoM);
  const long pr_cap = wino_plane_rows(s.rows(), s.n_utt);
  int* segp = reinterpret_cast<int*>(scratch.p);  // [kWinoMaxN + 1] plane offsets j * plane rows: the contraction's "utterances"
  int* goff = segp + 16;                          // [n_utt + 1] first group of every utterance
  float* Xp = scratch.p + round_up(s.n_utt + 1 + 16, 32);
  // split fp32: the input transform writes the three bf16 planes of every component plane (the contraction stages them as they are)
  // (built, parity-tested, NOT selected: with the planes pre-split the contraction's K loop has no split arithmetic and, on tiles 27 / 28, no register
  //  staging at all - and runs exactly as fast, 78.1 vs 78.7 us on the decoder's conv2 at B = 8, while this transform writes 1.5 x the bytes and the
  //  launch loses its remainder-round plan: cfg2 2 037 vs 2 145 utt/s.  STTS_X3_PRESPLIT=1 switches it on for experiments.)
  static const bool presplit_env = getenv("STTS_X3_PRESPLIT") && atoi(getenv("STTS_X3_PRESPLIT")) != 0;
  const bool presplit = presplit_env && x3_enabled() && wc.planes.w16_plane > 0 && wc.planes.prec == PREC_F32 && kc % 8 == 0;
  const long xplane = (long)n * pr_cap * kc;  // elements between two split planes
  float* Mp = Xp + (size_t)n * pr_cap * kc * 3 / 2;
  WinoIn ti;
  WinoOut to;
  memcpy(ti.Bt, wc.mats.Bt, sizeof(ti.Bt));
  memcpy(to.At, wc.mats.At, sizeof(to.At));
  const int groups = ceil_div(ml, kWinoM);
  // measurement (bench.py roofline leg): the conv is timed as a whole (both transforms + contraction, stream markers) and
  // credited with the ALGORITHMIC flops of the direct convolution, 2 * rows * cout * cin * taps
  GemmProfiler& prof = gemm_profiler();
  const bool timed = prof.on;
  if (timed) {
    (void)hipEventRecord(prof.next(), st);
    prof.on = false;
  }
  if (!scratch.setup) {
    hipLaunchKernelGGL(wino_setup_kernel, dim3(1), dim3(64), 0, st, s.dev, s.n_utt, kWinoMaxN, goff, segp);
    scratch.setup = true;
  }
  if (presplit)
    hipLaunchKernelGGL((winograd_input_kernel<N, true>), dim3(ceil_div(groups, 4), ceil_div(kc / 4, 64), s.n_utt), dim3(64, 4), 0, st, X, ldx,
                       wc.planes.cin_real, s.dev, wc.pad, ti, Xp, kc, goff, aff, ld_aff, xplane);
  else
  hipLaunchKernelGGL((winograd_input_kernel<N>), dim3(ceil_div(groups, 4), ceil_div(kc / 4, 64), s.n_utt), dim3(64, 4), 0, st, X, ldx,
                     wc.planes.cin_real, s.dev, wc.pad, ti, Xp, kc, goff, aff, ld_aff, 0L);
  std::vector<int> seg_h(n + 1);
  for (int j = 0; j <= n; ++j) seg_h[j] = (int)(j * pr);
  Seg sp{n, seg_h.data(), segp};
  sp.cap = s.cap;  // capacity segments: the planes' host offsets (j * pr) are upper bounds of the device ones (j * real groups) too
  GemmArgs a = gemm_args(sp);
  set_seg(a, 0, Xp, kc, 0, wc.planes, 0);
  a.seg[0].w_utt_stride = (long)wc.planes.npad * kc;
  if (presplit) {
    a.x16 = 1;
    a.seg[0].x_plane = xplane;
  }
  a.N = wc.planes.N; a.bias = nullptr; a.Y = Mp; a.ldy = ldm;
  {
    const int rc = launch_conv_gemm(st, a, EPI_STORE, wc.planes.npad, n, (int)pr);
    if (rc) {
      prof.on = timed;
      return rc;
    }
  }
  // stat: the output transform also leaves the AdaIN statistics of Y per chunk of kWinoStatChunk rows (wino_stat_chunks(s) chunks per utterance)
  if (stat)
    hipLaunchKernelGGL((winograd_output_kernel<N, true>), dim3(ceil_div(groups, 4), ceil_div(ceil_div(wc.planes.N, 4), 64), s.n_utt), dim3(64, 4), 0, st, Mp, ldm, goff,
                       s.dev, to, wc.planes.bias, act, R, ldr, alpha, Y, ldy, wc.planes.N, stat, ld_stat, ceil_div(groups, 4));
  else
    hipLaunchKernelGGL((winograd_output_kernel<N, false>), dim3(ceil_div(groups, 4), ceil_div(ceil_div(wc.planes.N, 4), 64), s.n_utt), dim3(64, 4), 0, st, Mp, ldm, goff,
                       s.dev, to, wc.planes.bias, act, R, ldr, alpha, Y, ldy, wc.planes.N, nullptr, 0, 0);
  if (timed) {
    prof.on = true;
    (void)hipEventRecord(prof.next(), st);
    prof.add((x3_enabled() && wc.planes.w16_plane > 0) ? "winograd_conv_x3" : "winograd_conv", 0, 2.0 * (double)s.rows() * wc.planes.rows_real * wc.planes.cin_real * wc.mats.r,
             2.0 * (double)n * (double)pr * wc.planes.rows_real * wc.planes.cin_real, 0.0);
  }
  STTS_HIP(hipGetLastError());
  return 0;
}
inline int wino_stat_chunks(const Seg& s) { return ceil_div(ceil_div(s.max_len(), kWinoM), 4); }
inline int run_winograd(hipStream_t st, const Seg& s, const float* X, int ldx, const WinoConv& wc, float* Y, int ldy, int act, const float* R, int ldr,
                        float alpha, WinoScratch& scratch, const float* aff = nullptr, int ld_aff = 0, float* stat = nullptr, int ld_stat = 0) {
  STTS_CHECK(wc.ready && (wc.mats.n == 8 || wc.mats.n == 12), "winograd conv not packed");
  STTS_CHECK(ldx % 4 == 0 && ldy % 4 == 0 && (!R || ldr % 4 == 0), "winograd conv: leading dimensions must be multiples of 4");
  STTS_CHECK(!stat || (ld_stat % 4 == 0 && ld_stat >= round_up(wc.planes.N, 4)), "winograd conv: statistics rows of %d floats do not cover %d channels", ld_stat, wc.planes.N);
  return wc.mats.n == 8 ? run_winograd_n<8>(st, s, X, ldx, wc, Y, ldy, act, R, ldr, alpha, scratch, aff, ld_aff, stat, ld_stat)
                        : run_winograd_n<12>(st, s, X, ldx, wc, Y, ldy, act, R, ldr, alpha, scratch, aff, ld_aff, stat, ld_stat);
}

inline int run_style(hipStream_t st, const StyleTable& t, const float* style, int n_utt, float* out) {
  if (n_utt >= 16 && t.K % 4 == 0 && t.K <= 128) {  // lane = utterance: no per-utterance reduction (same sums, other order: ~1e-7)
    STTS_LAUNCH_PROF("style_fc_batch_kernel", (size_t)t.J * (t.K + n_utt) * 4, style_fc_batch_kernel, dim3(ceil_div(t.J, 4)), dim3(256), st, t.W, t.b, style, out,
                     t.J, t.K, n_utt, t.K, t.ld());
    STTS_HIP(hipGetLastError());
    return 0;
  }
  STTS_LAUNCH_PROF("style_fc_kernel", (size_t)t.J * (t.K + n_utt) * 4, style_fc_kernel, dim3(ceil_div(t.J, 4)), dim3(256), st, t.W, t.b, style, out, t.J, t.K, n_utt, t.K, t.ld());
  STTS_HIP(hipGetLastError());
  return 0;
}

inline dim3 rows_grid(const Seg& s, int per_row_work) {
  const long total = (long)s.max_len() * per_row_work;
  return dim3((unsigned)std::min<long>(std::max<long>(1, ceil_div((int)std::min<long>(total, 1 << 30), 256)), 512), s.n_utt);
}

// rows from which the 16-bit operand modes keep contraction inputs as 16-bit rows in HBM (B = 4 x 3 s; measured bf16, 3-s utterances:
// B = 1: 827 vs 739 utt/s without / with 16-bit rows, B = 2: 1 424 vs 1 313, B = 4: 2 272 vs 2 304, B = 8: 3 442 vs 3 522)
inline long rows16_threshold() {
  static const long v = getenv("STTS_ROWS16") ? atol(getenv("STTS_ROWS16")) : 3840;
  return v;
}

// AdaIN + activation: Y[:, :ldy] = act((1+gamma) * InstanceNorm(X[:, :C]) + beta), zeros in the pad columns.
// part: scratch of adain_part_floats(s, C) floats.
inline size_t adain_part_floats(const Seg& s, int C, int chunk_rows = kStatChunk) { return (size_t)s.n_utt * ceil_div(s.max_len(), chunk_rows) * 2 * round_up(C, 32); }
// out16: PREC_BF16 / PREC_F16 = Y is a 16-bit row buffer (ldy in elements), the input of a contraction in that operand mode
inline int run_adain(hipStream_t st, const Seg& s, const float* X, int ldx, int C, float* Y, int ldy, const float* style_out, int ld_style,
                     int gcol0, int act, const float* alpha, float* part, int out16 = 0, bool have_stats = false) {
  // have_stats: `part` already holds the chunk statistics of X (written by the producing contraction's epilogue, GemmArgs::stat_part)
  const int nchunk = ceil_div(s.max_len(), kStatChunk), ldp = round_up(C, 32);
  if (!have_stats)
    STTS_LAUNCH_PROF("adain_partial_kernel", (size_t)s.rows() * C * 4, adain_partial_kernel, dim3(ceil_div(C, 32), nchunk, s.n_utt), dim3(256), st, X, ldx, C, s.dev, part, ldp, nchunk, kStatChunk);
  const int rb = (long)ceil_div(ldy, 64) * ceil_div(s.rows(), 64) >= 4096 ? 256 : 64;  // rows per block
  STTS_LAUNCH_PROF("adain_apply_kernel", (size_t)s.rows() * (C * 4 + ldy * (out16 ? 2 : 4)), adain_apply_kernel, dim3(ceil_div(ldy, 64), ceil_div(s.max_len(), rb), s.n_utt), dim3(256), st, X, ldx, Y, ldy, C, s.dev,
                     part, ldp, nchunk, style_out, ld_style, gcol0, 1e-5f, act, alpha, out16, rb);
  STTS_HIP(hipGetLastError());
  return 0;
}

// A contraction whose split-K reduce pass is left to the consumer of its output (small batches: the next AdaIN's statistics kernel finishes it).
struct PendingReduce {
  SplitSrc src{};   // src.partial == nullptr: nothing pending
  float* Y = nullptr;
  int ldy = 0;
};
inline void pending_from(PendingReduce* p, const SplitInfo& info, const GemmArgs& a) {
  p->src = SplitSrc{};
  if (info.ksplit > 1) {
    p->src = SplitSrc{info.partial, info.ksplit, info.slice_rows, info.ld_part, a.N, a.bias, a.act, a.R, a.ldr, a.rcol0, a.alpha};
    p->Y = a.Y + a.ycol0;
    p->ldy = a.ldy;
  }
}
// finishes a pending reduce on its own (no consumer took it)
inline int pending_finish(hipStream_t st, PendingReduce* p) {
  if (!p || !p->src.partial) return 0;
  const SplitSrc& r = p->src;
  const long work = (long)r.slice_rows * ((r.N + 3) / 4);
  STTS_LAUNCH_PROF("splitk_reduce_kernel", (size_t)work * 4 * 4 * (r.ksplit + 1), splitk_reduce_kernel, dim3((unsigned)std::min<long>(2048, (work + 255) / 256)), dim3(256), st, r.partial,
                   r.ksplit, r.slice_rows, 0, r.ld_part, r.N, r.bias, r.act, r.R, r.ldr, r.rcol0, r.alpha, p->Y, p->ldy, 0);
  p->src = SplitSrc{};
  STTS_HIP(hipGetLastError());
  return 0;
}
// batches up to this many rows fold AdaIN into the consuming contraction's staging (run_adain_block); above it the decoder convs run in Winograd form
inline long fold_rows() {
  static const long v = getenv("STTS_FOLD_ROWS") ? atol(getenv("STTS_FOLD_ROWS")) : 2500;
  return v;
}
// A second stream for a contraction that is independent of the launches around it (run_adain_block: the learned 1x1 shortcut next to conv1), with the
// fork / join events that order it against the caller's stream and an output buffer of its own ([rows, cout]).
struct SideWork {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  float* out = nullptr;
};
struct AdainStats {
  float* in = nullptr;   // statistics of x for norm1 (layout of adain_partial_kernel, round_up(cin, 32) columns)
  bool in_ready = false;
  float* mid = nullptr;  // scratch: statistics of conv1's output for norm2
  float* out = nullptr;  // where conv2 writes the statistics of y
  int out_ld = 0;
  bool out_ready = false;
  int chunk_rows = kStatChunk;  // kWinoStatChunk: fp32 Winograd branch, the statistics come out of the convs' output transforms (winograd_output_kernel<N, true>)
};
// AdaptiveDecoderBlock (models/ada_norm.py:166-182).  x [rows, ldx] (cols >= cin may hold anything when
// kcin == round_up(cin) because the packed weights are zero there, but AdaIN writes zeros anyway).
// scratch: act1 [rows, kcin], h [rows, cout], act2 [rows, cout], ss [adain_part_floats(s, max(kcin, cout))]
inline int run_adain_block(hipStream_t st, const Seg& s, const AdainBlockW& B, const float* style_out, int ld_style, const float* x, int ldx,
                           float* y, int ldy, float* act1, float* hbuf, float* act2, float* ss, int force_tile = 0, WinoScratch* wino = nullptr,
                           unsigned short* xs16 = nullptr, bool xs16_ready = false, unsigned short* y16 = nullptr, int ldy16 = 0,
                           AdainStats* stats = nullptr, PendingReduce* pend_in = nullptr, PendingReduce* pend_out = nullptr, const SideWork* side = nullptr) {
  // side (fp32 Winograd branch, learned shortcut): the 1x1 shortcut contraction - it reads the block's input only - runs on side->stream next to conv1's three
  // launches (whose plane contraction fills the chip 1.25 times and whose transforms leave the matrix cores idle) instead of between conv1 and conv2.
  // pend_in / pend_out (small batches, fold path): x's first columns may still be the split-K partial sums of the contraction that produced them
  // (pend_in: finished by norm1's statistics launch, which also writes them to x); conv2's own reduce pass is left to the caller's next consumer
  // (pend_out).  Without them every contraction finishes its own output.
  // stats (16-bit modes, large batches; decoder_forward): statistics written by the producing contractions' epilogues (conv_gemm16_kernel,
  // GemmArgs::stat_part) instead of a pass of adain_partial_kernel per norm - norm1's come from the previous block's conv2 (`in`, if `in_ready`),
  // norm2's from this block's conv1 (`mid`), and conv2 leaves those of y for the next block's norm1 in `out` (`out_ld` columns per row, the
  // next block's padded input width; its constant columns are filled once by the caller); `out_ready` reports whether it did.
  // xs16: [rows, ldx] 16-bit scratch for the rounded copy of x a learned shortcut reads (16-bit modes, large batches);
  // xs16_ready: it already holds that copy (the previous block's conv2 wrote it).  y16: also write y rounded, as [rows, ldy16].
  const int ml = s.max_len();
  // Small batches (launch-latency bound): AdaIN -> LeakyReLU is folded into the staging of the contraction that consumes
  // it (conv_gemm_f32<..., XAFF>): the statistics pass stays, a 64-thread kernel turns them into per-(utterance, channel)
  // scale / shift tables (kept in act1 / act2), and the normalised tensor is never written: B = 1 frame path -4 %.
  // Large batches keep the separate apply pass: the affine in the K loop costs the 512-channel contractions ~14 % at
  // B = 8 (142 vs 125 us), more than the 11 us pass it removes.  fp32: from 2 500 rows on both convs run in Winograd form instead (AdaIN rides in
  // their input transforms): 3-s utterances, same box: B = 3: 2.78 -> 2.63 ms, B = 4: 2.88 -> 2.83; B = 2: 2.16 -> 2.26 and B = 1: 1.77 -> 1.92 (fold stays).
  const bool fold = s.rows() <= (B.conv1.prec == PREC_F32 ? fold_rows() : 4096);  // (16-bit modes: measured at 4 096 only)
  STTS_CHECK(ldx >= B.kcin, "adain block: input leading dimension %d < padded channels %d", ldx, B.kcin);
  static const bool affine_lanes = getenv("STTS_AFFINE_SERIAL") == nullptr;  // experiments: the one-thread-per-channel merge (adain_affine_kernel)
  auto affine = [&](const float* X, int ld, int C, int ld_aff, int gcol0, float* aff, PendingReduce* pend = nullptr) {
    const int nchunk = ceil_div(ml, kStatChunk), ldp = round_up(C, 32);
    if (pend && pend->src.partial) {  // the producer's reduce pass rides in the statistics launch (X's first columns are written here)
      STTS_LAUNCH_PROF("adain_partial_reduce_kernel", (size_t)s.rows() * C * 4 * (1 + pend->src.ksplit), adain_partial_reduce_kernel, dim3(ceil_div(C, 32), nchunk, s.n_utt), dim3(256), st,
                       const_cast<float*>(X), ld, C, s.dev, ss, ldp, nchunk, pend->src);
      pend->src = SplitSrc{};
    } else
    STTS_LAUNCH_PROF("adain_partial_kernel", (size_t)s.rows() * C * 4, adain_partial_kernel, dim3(ceil_div(C, 32), nchunk, s.n_utt), dim3(256), st, X, ld, C, s.dev, ss, ldp, nchunk, kStatChunk);
    if (affine_lanes)
      STTS_LAUNCH_PROF("adain_affine_lanes_kernel", (size_t)s.n_utt * nchunk * 2 * ldp * 4, adain_affine_lanes_kernel, dim3(ceil_div(ld_aff, 16), s.n_utt), dim3(256), st, ss, ldp, nchunk, s.dev, style_out, ld_style,
                       gcol0, C, 1e-5f, aff, ld_aff, kStatChunk);
    else
    STTS_LAUNCH_PROF("adain_affine_kernel", (size_t)s.n_utt * nchunk * 2 * ldp * 4, adain_affine_kernel, dim3(ceil_div(ld_aff, 64), s.n_utt), dim3(64), st, ss, ldp, nchunk, s.dev, style_out, ld_style, gcol0, C,
                       1e-5f, aff, ld_aff, kStatChunk);
  };
  // norm1 -> LeakyReLU -> conv1
  GemmArgs a = gemm_args(s);
  PendingReduce pend_mid;
  if (pend_in && pend_in->src.partial) STTS_CHECK(pend_in->Y == x && pend_in->ldy == ldx, "adain block: the pending reduce does not belong to this input");
  if (!fold) STTS_TRY(pending_finish(st, pend_in));
  if (fold) {
    affine(x, ldx, B.cin, B.kcin, B.n1.col0, act1, pend_in);
    set_seg(a, 0, x, ldx, 0, B.conv1);
    a.xaff = act1;
    a.ld_xaff = B.kcin;
  }
  const bool wino1 = !fold && wino && *wino && B.w1.ready && force_tile == 0;
  // (learned shortcut: conv2 in Winograd form + the 1x1 shortcut as a contraction of its own, added as the Winograd conv's residual)
  static const bool wino2_sc = getenv("STTS_NO_WINO_CONV2SC") == nullptr;
  const bool wino2 = !fold && wino && *wino && B.w2.ready && (!B.sc.W || wino2_sc) && force_tile == 0;
  // fp32 Winograd branch: every normalised tensor is the output of a Winograd conv (conv1 -> norm2, conv2 -> the next block's norm1), whose output
  // transform leaves the chunk statistics behind: the norms keep only their 64-thread affine launch (no adain_partial_kernel pass)
  const bool wstats = stats && stats->chunk_rows == kWinoStatChunk && wino1 && wino2;
  auto affine_from = [&](const float* part, int C, int ld_aff, int gcol0, float* aff) {
    const int wch = wino_stat_chunks(s), ldp = round_up(C, 32);
    STTS_LAUNCH_PROF("adain_affine_lanes_kernel", (size_t)s.n_utt * wch * 2 * ldp * 4, adain_affine_lanes_kernel, dim3(ceil_div(ld_aff, 16), s.n_utt), dim3(256), st, part, ldp, wch, s.dev, style_out, ld_style,
                     gcol0, C, 1e-5f, aff, ld_aff, kWinoStatChunk);
  };
  // 16-bit operand modes, large batches: the normalised activations are WRITTEN as 16-bit rows (act1 / act2 reinterpreted),
  // so the contractions stage half the bytes and convert nothing; a learned shortcut reads a rounded copy of x (xs16)
  const int h16 = (s.rows() >= rows16_threshold() && B.conv1.prec != PREC_F32 && force_tile == 0 && (!B.sc.W || xs16) && ldx % 8 == 0) ? B.conv1.prec : 0;
  const int nchunk = ceil_div(ml, kStatChunk);
  const bool fuse_stats = stats && h16 && !fold && !wino1 && !wino2;
  if (!fold && !wino1) {
    const bool ready = fuse_stats && stats->in_ready;
    STTS_TRY(run_adain(st, s, x, ldx, B.cin, act1, B.kcin, style_out, ld_style, B.n1.col0, ACT_LRELU, nullptr, ready ? stats->in : ss, h16, ready));
    set_seg(a, 0, act1, B.kcin, 0, B.conv1);
    a.x16 = h16 != 0;
  }
  if (stats) stats->out_ready = false;
  a.N = B.cout;
  a.bias = B.conv1.bias;
  a.Y = hbuf;
  a.ldy = B.cout;
  auto shortcut = [&](hipStream_t sst, float* out) -> int {
    GemmArgs g = gemm_args(s);
    set_seg(g, 0, x, ldx, 0, B.sc);
    g.N = B.cout; g.bias = nullptr; g.Y = out; g.ldy = B.cout;
    return launch_conv_gemm(sst, g, EPI_STORE, B.sc.npad, s.n_utt, ml, 0);
  };
  const bool sc_side = side && side->stream && side->out && wino1 && wino2 && B.sc.W && !gemm_profiler().on;
  if (sc_side) {  // x is complete in st's order here; side->out was last read by the previous block's conv2, which st has queued before this event
    STTS_HIP(hipEventRecord(side->fork, st));
    STTS_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
    const int rc = shortcut(side->stream, side->out);
    STTS_HIP(hipEventRecord(side->join, side->stream));  // (recorded whatever rc is: st joins below or at the caller's error path via stream order)
    if (rc) { (void)hipStreamWaitEvent(st, side->join, 0); return rc; }
  }
  if (wino1) {
    // large batches: conv1 (k = 3) in Winograd F(6,3) form, 8 instead of 18 multiplies per 6 outputs (winograd.hip.h); AdaIN + LeakyReLU ride
    // in its input transform (an elementwise, bandwidth-bound kernel: there the affine is free, unlike in the K loop)
    if (wstats && stats->in_ready) affine_from(stats->in, B.cin, B.kcin, B.n1.col0, act1);
    else affine(x, ldx, B.cin, B.kcin, B.n1.col0, act1);
    STTS_TRY(run_winograd(st, s, x, ldx, B.w1, hbuf, B.cout, ACT_NONE, nullptr, 0, 1.0f, *wino, act1, B.kcin, wstats ? stats->mid : nullptr, round_up(B.cout, 32)));
  } else {
    if (fuse_stats && gemm16_will_run(a, EPI_STORE, B.conv1.npad, s.n_utt)) {
      a.stat_part = stats->mid; a.ld_stat = round_up(B.cout, 32); a.stat_nchunk = nchunk;
    }
    SplitInfo info1{};
    if (fold) a.defer = &info1;  // norm2's statistics launch finishes a split-K conv1
    STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, B.conv1.npad, s.n_utt, ml, force_tile));
    if (fold) pending_from(&pend_mid, info1, a);
  }
  const bool mid_ready = a.stat_part != nullptr;
  // norm2 -> LeakyReLU -> conv2 (+ learned 1x1 shortcut as a second K segment | + identity residual), / sqrt(2)
  GemmArgs b = gemm_args(s);
  if (fold) {
    affine(hbuf, B.cout, B.cout, B.cout, B.n2.col0, act2, &pend_mid);
    set_seg(b, 0, hbuf, B.cout, 0, B.conv2);
    b.xaff = act2;
    b.ld_xaff = B.cout;
  } else if (wino2) {
    if (wstats) affine_from(stats->mid, B.cout, B.cout, B.n2.col0, act2);
    else affine(hbuf, B.cout, B.cout, B.cout, B.n2.col0, act2);
  } else {
    STTS_TRY(run_adain(st, s, hbuf, B.cout, B.cout, act2, B.cout, style_out, ld_style, B.n2.col0, ACT_LRELU, nullptr, mid_ready ? stats->mid : ss, h16, mid_ready));
    set_seg(b, 0, act2, B.cout, 0, B.conv2);
    b.x16 = h16 != 0;
  }
  if (B.sc.W) {
    if (h16) {
      if (!xs16_ready) launch_cast_rows(st, h16, x, ldx, B.cin, xs16, ldx, s.rows());
      set_seg(b, 1, reinterpret_cast<const float*>(xs16), ldx, 0, B.sc);
    } else {
      set_seg(b, 1, x, ldx, 0, B.sc);
    }
  } else {
    b.R = x;
    b.ldr = ldx;
  }
  b.N = B.cout;
  b.bias = B.conv2.bias;
  b.Y = y;
  b.ldy = ldy;
  b.alpha = 0.70710678118654752440f;
  const bool y16_epi = y16 && h16 && !wino2;  // the 16-bit contraction's epilogue rounds y for the next block's shortcut
  if (y16_epi) {
    b.Y16 = y16;
    b.ldy16 = ldy16;
  }
  if (wino2) {
    const float* res = x;
    int ld_res = ldx;
    if (B.sc.W && sc_side) {
      STTS_HIP(hipStreamWaitEvent(st, side->join, 0));
      res = side->out;
      ld_res = B.cout;
    } else if (B.sc.W) {  // act1 is free again (conv1 has consumed it): the shortcut's output lives there
      STTS_TRY(shortcut(st, act1));
      res = act1;
      ld_res = B.cout;
    }
    const bool leave = wstats && stats->out;  // the statistics of y for the next block's norm1 (its hidden columns)
    STTS_TRY(run_winograd(st, s, hbuf, B.cout, B.w2, y, ldy, ACT_NONE, res, ld_res, b.alpha, *wino, act2, B.cout, leave ? stats->out : nullptr, leave ? stats->out_ld : 0));
    if (leave) stats->out_ready = true;
  } else {
    if (fuse_stats && stats->out && gemm16_will_run(b, EPI_STORE, B.conv2.npad, s.n_utt)) {
      b.stat_part = stats->out; b.ld_stat = stats->out_ld; b.stat_nchunk = nchunk;
      stats->out_ready = true;
    }
    SplitInfo info2{};
    if (fold && pend_out && !y16) b.defer = &info2;  // (a rounded copy of y is cast from y right below: y must exist then)
    STTS_TRY(launch_conv_gemm(st, b, EPI_STORE, B.conv2.npad, s.n_utt, ml, force_tile));
    if (fold && pend_out && !y16) pending_from(pend_out, info2, b);
  }
  if (y16 && !y16_epi) launch_cast_rows(st, B.conv1.prec, y, ldy, B.cout, y16, ldy16, s.rows(), B.cout);  // (cout % 8 == 0: caller)
  STTS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// stage: Decoder.forward (models/decoder.py:47-60)
// ------------------------------------------------------------------------------------------------
inline int decoder_forward(stts_ctx* c, hipStream_t st, const Seg& s, const float* asr, int ld_asr, const float* pitch, const float* energy,
                           const float* style, float* x_out, int ld_x, Arena& ws, const SideWork* side_lane = nullptr) {
  const stts_model_dims& d = c->d;
  const long R = s.rows();
  const int ccat = d.dec_hidden + d.dec_residual + 2, ldcat = c->dec[1].kcin;  // 578 -> 608 (640 in the 16-bit modes): the packed input width
  const int cenc = d.inter_dim + 2, ldenc = c->dec[0].kcin;                    // 130 -> 160 (192)
  STTS_CHECK(ldcat >= ccat && ldenc >= cenc, "decoder_forward: packed widths %d / %d do not cover %d / %d channels", ldcat, ldenc, ccat, cenc);
  // (dec[0] runs on the encoder-input width, the other blocks on the concat width: scratch is sized by the wider of the two - a config
  //  with inter_dim > hidden_dim + residual_dim is legal)
  const int ldwide = std::max(ldenc, ldcat);
  float* enc_in = ws.get<float>(R * ldenc);
  float* xa = ws.get<float>(R * ldcat);
  float* xb = ws.get<float>(R * ldcat);
  float* act1 = ws.get<float>(R * ldwide);
  float* hbuf = ws.get<float>(R * d.dec_hidden);
  float* act2 = ws.get<float>(R * d.dec_hidden);
  float* ss = ws.get<float>(adain_part_floats(s, ldwide));
  float* ss_in = ws.get<float>(adain_part_floats(s, ldwide, kWinoStatChunk));   // statistics from the producers' epilogues (AdainStats): 16-bit contractions, or
  float* ss_mid = ws.get<float>(adain_part_floats(s, ldwide, kWinoStatChunk));  // the fp32 Winograd output transforms (chunks of 24 rows: the larger footprint)
  float* sty = ws.get<float>((size_t)s.n_utt * c->dec_style.ld());
  // conv1 of every block in Winograd form once the batch is large enough to be throughput-bound (B = 1: 35 vs 30 us)
  WinoScratch wino;
  if (R > fold_rows() && c->dec[1].w1.ready) wino.p = ws.get<float>(wino_scratch_floats(s, c->dec[1].w1));
  // the learned shortcuts' outputs when they run on the side lane (run_adain_block; carved whenever the Winograd branch may run, so the size does not depend on the caller's streams)
  float* sc_out = wino.p && c->dec[1].sc.W ? ws.get<float>(R * d.dec_hidden) : nullptr;
  // 16-bit modes, large batches: rounded copies of the blocks' inputs for the learned shortcuts, ping-pong like xa / xb: a block's
  // conv2 epilogue writes the hidden columns of the next block's copy, the constant columns (asr_res, F0, N) are rounded once
  const bool x16 = R >= rows16_threshold() && c->prec != PREC_F32 && d.dec_hidden % 8 == 0;
  unsigned short* xs16a = x16 ? ws.get<unsigned short>(R * ldwide) : nullptr;
  unsigned short* xs16b = x16 ? ws.get<unsigned short>(R * ldwide) : nullptr;
  STTS_CHECK(ws.ok, "decoder_forward: workspace too small");
  STTS_DRY_RETURN(ws);
  STTS_TRY(run_style(st, c->dec_style, style, s.n_utt, sty));
  FrontArgs fa;
  fa.asr = asr; fa.ld_asr = ld_asr; fa.pitch = pitch; fa.energy = energy;
  for (int i = 0; i < 3; ++i) { fa.wf[i] = c->front_wf[i]; fa.wn[i] = c->front_wn[i]; }
  fa.bf = c->front_bf; fa.bn = c->front_bn;
  fa.enc_in = enc_in; fa.ld_enc = ldenc; fa.xa = xa; fa.xb = xb; fa.ld_x = ldcat;
  fa.c_asr = d.inter_dim; fa.c_hidden = d.dec_hidden; fa.c_res = d.dec_residual;
  STTS_LAUNCH_PROF("decoder_front_kernel", (size_t)R * (d.inter_dim + 2 + ldenc + 2 * 4) * 4, decoder_front_kernel, rows_grid(s, ldenc / 4), dim3(256), st, fa, s.dev);
  // asr_res = wn-conv1x1(asr) into the residual columns of both concat buffers (decoder.py:54)
  for (float* dst : {xa, xb}) {
    GemmArgs a = gemm_args(s);
    set_seg(a, 0, asr, ld_asr, 0, c->asr_res);
    a.N = d.dec_residual; a.bias = c->asr_res.bias; a.Y = dst; a.ldy = ldcat; a.ycol0 = d.dec_hidden;
    STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, c->asr_res.npad, s.n_utt, s.max_len()));
  }
  const int lds = c->dec_style.ld();
  const int ctail = d.dec_hidden & ~7, ntail = ldcat - ctail;  // constant columns (from the 8-aligned column at or below dec_hidden)
  auto cast_tail = [&](const float* src, unsigned short* dst) {
    launch_cast_rows(st, c->prec, src + ctail, ldcat, ccat - ctail, dst + ctail, ldcat, R, ntail);
  };
  // dec[0] rounds its own (narrow) input into xs16b as scratch, so xs16b's constant columns are filled after it
  // 16-bit modes, large batches: the statistics of a block's input come out of the previous block's conv2 epilogue (its 512 hidden columns) and,
  // for the constant columns [asr_res | F0 | N] that every decode block sees, from ONE statistics pass here
  static const bool no_stat_fuse = getenv("STTS_NO_STAT_FUSE") != nullptr;  // experiments: every norm runs its own statistics pass
  static const bool no_wino_stats = getenv("STTS_NO_WINO_STATS") != nullptr;  // experiments: the fp32 Winograd branch runs a statistics pass per norm
  const bool wstat_mode = c->prec == PREC_F32 && wino && !no_wino_stats;
  const bool use_stats = (x16 && !no_stat_fuse) || wstat_mode;
  AdainStats stats;
  if (wstat_mode) stats.chunk_rows = kWinoStatChunk;
  stats.mid = ss_mid;
  stats.out = ss_in;
  stats.out_ld = round_up(ccat, 32);
  // small batches: a block's conv2 leaves its split-K reduce pass to the next block's first statistics launch (PendingReduce)
  PendingReduce pend;
  STTS_TRY(run_adain_block(st, s, c->dec[0], sty, lds, enc_in, ldenc, xa, ldcat, act1, hbuf, act2, ss, 0, &wino, xs16b, false, xs16a, ldcat, use_stats ? &stats : nullptr,
                           nullptr, &pend));
  if (x16) {
    cast_tail(xa, xs16a);
    cast_tail(xb, xs16b);
  }
  // (once, as soon as a block has left statistics of its output: whichever block that is, the constant columns must be there before the next norm1 merges them)
  bool const_ready = false;
  auto const_columns = [&]() -> int {
    if (!stats.out_ready || const_ready) return 0;
    const int nchunk = ceil_div(s.max_len(), stats.chunk_rows), cc0 = d.dec_hidden, ncst = ccat - cc0;
    STTS_CHECK(cc0 % 32 == 0, "decoder_forward: hidden_dim must be a multiple of 32");
    STTS_LAUNCH_PROF("adain_partial_kernel", (size_t)R * ncst * 4, adain_partial_kernel, dim3(ceil_div(ncst, 32), nchunk, s.n_utt), dim3(256), st, xa + cc0, ldcat, ncst, s.dev,
                     ss_in + cc0, stats.out_ld, nchunk, stats.chunk_rows);
    const_ready = true;
    return 0;
  };
  STTS_TRY(const_columns());
  float* cur = xa;
  float* nxt = xb;
  unsigned short* cur16 = xs16a;
  unsigned short* nxt16 = xs16b;
  // Measured (3-s utterances, same box, alternated): B = 16: 5.82 -> 5.74 ms per step (+1.4 %); B = 8: 3.505 -> 3.52 (-0.5 %), B = 4: 2.39 -> 2.42 (-1.3 %): below
  // ~12 000 rows the two cross-queue waits per block cost what the overlap gains, and the shortcuts stay on the caller's stream (STTS_SC_SIDE_MIN_ROWS).
  static const long sc_side_min_rows = getenv("STTS_SC_SIDE_MIN_ROWS") ? atol(getenv("STTS_SC_SIDE_MIN_ROWS")) : 12000;
  SideWork side;
  if (side_lane && sc_out && R >= sc_side_min_rows) {
    side = *side_lane;
    side.out = sc_out;
  }
  auto blocks = [&]() -> int {
    for (int i = 1; i <= 4; ++i) {
      float* dst = i == 4 ? x_out : nxt;
      const int ldd = i == 4 ? ld_x : ldcat;
      STTS_TRY(const_columns());
      stats.in = ss_in;
      stats.in_ready = stats.out_ready;  // (the previous block's conv2 - conv_gemm16_kernel's epilogue or the fp32 Winograd output transform - left the statistics of its output)
      stats.out = i == 4 ? nullptr : ss_in;
      STTS_TRY(run_adain_block(st, s, c->dec[i], sty, lds, cur, ldcat, dst, ldd, act1, hbuf, act2, ss, 0, &wino, cur16, x16, i == 4 ? nullptr : nxt16, ldcat,
                               use_stats ? &stats : nullptr, &pend, &pend, side.stream ? &side : nullptr));
      std::swap(cur, nxt);
      std::swap(cur16, nxt16);
    }
    return 0;
  };
  const int rc = blocks();
  if (rc && side.stream) {  // an error path may have left a shortcut running on the side lane: the caller's stream must not outrun it (it reads x and writes the workspace)
    (void)hipEventRecord(side.join, side.stream);
    (void)hipStreamWaitEvent(st, side.join, 0);
  }
  if (rc) return rc;
  return pending_finish(st, &pend);  // the last block's output has no AdaIN behind it
}

#ifdef STTS_WN_TRACE
// diagnostics build only: per-wave phase stamps of the 32 fused WaveNet launches of one flow pass, averaged to stderr
struct WnTrace {
  long long* dev = nullptr;
  long blocks[32] = {};
  static constexpr long kMax = 1024;
};
inline WnTrace& wn_trace() {
  static WnTrace t;
  return t;
}
inline long long* wn_trace_buffer(int launch, long blocks) {
  WnTrace& t = wn_trace();
  if (!t.dev) (void)hipMalloc(&t.dev, 32 * WnTrace::kMax * 64 * sizeof(long long));
  if (blocks > WnTrace::kMax) return nullptr;
  t.blocks[launch] = blocks;
  (void)hipMemsetAsync(t.dev + launch * WnTrace::kMax * 64, 0, blocks * 64 * sizeof(long long), 0);
  return t.dev + launch * WnTrace::kMax * 64;
}
inline void wn_trace_report(hipStream_t st) {
  WnTrace& t = wn_trace();
  (void)hipStreamSynchronize(st);
  static int calls = 0;
  if (++calls % 8 != 0) return;
  std::vector<long long> h(WnTrace::kMax * 64);
  for (int kind = 0; kind < 2; ++kind) {  // plain layers, last layers
    double ph[8][6] = {}, wall = 0, clk = 0;
    long n = 0;
    for (int l = 0; l < 32; ++l) {
      if ((l % 4 == 3) != (kind == 1) || !t.blocks[l]) continue;
      (void)hipMemcpy(h.data(), t.dev + l * WnTrace::kMax * 64, t.blocks[l] * 64 * sizeof(long long), hipMemcpyDeviceToHost);
      for (long b = 0; b < t.blocks[l]; ++b) {
        const long long* r0 = &h[64 * b];
        if (!r0[5]) continue;
        for (int wv = 0; wv < kWnWaves; ++wv) {
          const long long* r = r0 + 8 * wv;
          for (int i = 0; i < 6; ++i) ph[wv][i] += (double)(r[i] - r0[0]);  // cycles since wave 0 started
        }
        wall += (double)(r0[7] - r0[6]) * 10.0;  // ns
        clk += (double)(r0[5] - r0[0]);
        ++n;
      }
    }
    if (!n) continue;
    const double ghz = clk / wall;
    fprintf(stderr, "[wn trace] %s layers: %ld blocks, wave 0: %.2f us per block at %.2f GHz; per wave, us since the block started: start | prologue end | phase-1 end | gate barrier | phase-2 end | end\n",
            kind ? "last" : "plain", n, wall / n * 1e-3, ghz);
    for (int wv = 0; wv < kWnWaves; ++wv) {
      fprintf(stderr, "    wave %d:", wv);
      for (int i = 0; i < 6; ++i) fprintf(stderr, " %6.2f", ph[wv][i] / n / ghz * 1e-3);
      fprintf(stderr, "\n");
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// stage: PriorEncoder + reverse flow + post_flow (models/flow.py:311-315, :132-151, :196-218, :63-88)
// ------------------------------------------------------------------------------------------------
inline int prior_flow_forward(stts_ctx* c, hipStream_t st, const Seg& s, const float* x, int ld_x, const float* style, const float* noise,
                              float* mel, int ld_mel, float* z_prior_out, float* z_flow_out, Arena& ws, unsigned short* mel16 = nullptr,
                              int ld_mel16 = 0) {
  // mel16: also write mel rounded to the operand precision (the vocoder's projector reads it; 16-bit modes, large batches)
  const stts_model_dims& d = c->d;
  const long R = s.rows();
  const int fh = d.dec_hidden / 4, half = fh / 2, ml = s.max_len();
  // (kWnRowPad rows of slack: wn_fused_kernel reads whole 16-row tiles, also past the last utterance; never stored)
  float* z = ws.get<float>((R + kWnRowPad) * fh);
  float* hf = ws.get<float>((R + kWnRowPad) * fh);
  float* hf2 = ws.get<float>((R + kWnRowPad) * fh);
  float* outf = ws.get<float>((R + kWnRowPad) * fh);
  // the fused WaveNet kernels are built for 128 flow channels (decoder.hidden_dim 512); other widths run every layer as
  // two contractions (gate epilogue, then res/skip split-accumulate) and need the gated activations in memory
  const bool generic = fh != kWnC;
  float* actsg = generic ? ws.get<float>(R * fh) : nullptr;
  float* cond = ws.get<float>((size_t)s.n_utt * c->flow_style.ld());
  STTS_CHECK(ws.ok, "prior_flow_forward: workspace too small");
  STTS_DRY_RETURN(ws);
  STTS_TRY(run_style(st, c->flow_style, style, s.n_utt, cond));
  {
    GemmArgs a = gemm_args(s);
    set_seg(a, 0, x, ld_x, 0, c->prior);
    a.N = fh; a.bias = c->prior.bias; a.Z = z; a.ldz = fh; a.noise = noise; a.ldnoise = fh;
    STTS_TRY(launch_conv_gemm(st, a, EPI_PRIOR, c->prior.npad, s.n_utt, ml));
  }
  if (z_prior_out) STTS_HIP(hipMemcpyAsync(z_prior_out, z, R * fh * sizeof(float), hipMemcpyDeviceToDevice, st));
  // reversed(flows) = Flip, layer 7, Flip, layer 6, ..., Flip, layer 0: after k flips the roles of the halves swap,
  // so layer f reads half p = (f odd) and updates the other half in place; after layer 0 the order is natural.
  // 32-row or 16-row blocks for the fused WaveNet layer (fp32): one block per CU is resident, so a launch takes
  // ceil(blocks / 256) rounds of ~38 us (32 rows) or ~21 us (16 rows: half the MFMA chain, but the weight staging per
  // block is the same).  B = 8: 240 x 38 us beats 480 blocks = 2 x 21; B = 12: 720 blocks = 3 x 21 beats 360 = 2 x 38.
  bool rows16 = false;
  int fused_m = 0;  // fp32: wn_fused_kernel with F(2,5) (32-row blocks) or F(4,5) (64-row blocks); 0 = the staged kernels
  if (c->prec == PREC_F32) {
    long b64 = 0, b32 = 0, b16 = 0;
    for (int u = 0; u < s.n_utt; ++u) {
      const int len = s.host[u + 1] - s.host[u];
      b64 += ceil_div(len, 64);
      b32 += ceil_div(len, 32);
      b16 += ceil_div(len, 16);
    }
    rows16 = ceil_div((int)b16, 256) * 21 < ceil_div((int)b32, 256) * 38;
    if (c->flow[0].fused.ready) {
      // wn_fused_kernel: one block per CU is resident, so a launch takes ceil(blocks / 256) rounds of ~kT2 us (F(2,5), 32-row
      // blocks) or ~kT4 us (F(4,5), 64-row blocks); the staged 16-row kernel (one round = ~kT16 us) only wins while its
      // blocks fit one round (B <= 4 at 3 s).  Measured (MI355X, tools/flow_bench.py): B = 8 28 us, B = 16 41 us (F(4,5)) vs
      // 55 (F(2,5)), B = 64 150 us vs 209 per WaveNet layer; staged kernels 40 / - / 287.
      const int force = getenv("STTS_WN_M") ? atoi(getenv("STTS_WN_M")) : 0;  // tests / tools: force a block shape
      constexpr int kT16 = 22, kT2 = 28, kT4 = 41;
      const int t16 = ceil_div((int)b16, 256) * kT16, t2 = ceil_div((int)b32, 256) * kT2, t4 = ceil_div((int)b64, 256) * kT4;
      // the same kernel in its direct form on 16-row blocks (M = 1) takes the place of the staged 16-row kernel for the smallest
      // batches (B = 1: 19.9 vs 21.1 us per layer, B = 4: 22.0 vs 23.2; F(2,5) there: 25.3 / 27.4)
      constexpr int kT1 = 21;
      const int t1 = ceil_div((int)b16, 256) * kT1;
      (void)t16;
      fused_m = t4 < t2 ? 4 : 2;
      if (t1 <= std::min(t2, t4)) fused_m = 1;
      rows16 = false;
      if (force == 1 || force == 2 || force == 4) fused_m = force;
      if (force == 16) {  // the staged 16-row kernel (kept for comparison)
        fused_m = 0;
        rows16 = true;
      }
    }
  }
  // 16-bit operand modes: wn_fused16_kernel on 64- or 128-row blocks (the taller block halves the weight stream per row; it
  // needs ~1.5 chip rounds of blocks to pay)
  int fused16_rt = 0;
  if (c->prec != PREC_F32 && !generic && c->flow[0].fused.ready16) {
    long b128 = 0;
    for (int u = 0; u < s.n_utt; ++u) b128 += ceil_div(s.host[u + 1] - s.host[u], 128);
    fused16_rt = b128 >= 384 ? 8 : 4;
    // at least ~0.75 chip rounds of 128-row blocks: one launch per coupling layer with h and `out` on chip (wn_block16.hip.h)
    if (b128 >= 192) fused16_rt = 16;
    const int force = getenv("STTS_WN_RT") ? atoi(getenv("STTS_WN_RT")) : 0;  // tests / tools
    if (force == 4 || force == 8 || force == 16) fused16_rt = force;
    if (force == -1) fused16_rt = 0;  // the staged kernel
  }
  // split fp32 (fp32 mode): wn_fused_x3_kernel on 32-row blocks, 64-row blocks once those fill the chip more than once (the taller block halves the
  // weight stream per row); small batches (16-row blocks win there) stay on the f32 kernel's direct form
  int fusedx3_rt = 0;
  if (c->prec == PREC_F32 && fused_m != 0 && fused_m != 1 && c->flow[0].fused.ready_x3) {
    long b64 = 0, b32 = 0;
    for (int u = 0; u < s.n_utt; ++u) {
      b64 += ceil_div(s.host[u + 1] - s.host[u], 64);
      b32 += ceil_div(s.host[u + 1] - s.host[u], 32);
    }
    // one block per CU is resident: a launch takes ceil(blocks / 256) rounds.  Fitted to 3-s batches of 8 .. 32 (us per launch): 32-row blocks 4.5 + 17.5 per
    // round, 64-row blocks (twice the matrix work per block for one weight stream) 1.5 + 29.5 per round.  B = 8: 22 vs 31; B = 10 .. 16: 40 vs 31;
    // B = 20 / 24: 57 vs 60; B = 32: 74 vs 60 - the measured optimum at every one of them
    fusedx3_rt = 1.5 + 29.5 * ceil_div((int)b64, 256) < 4.5 + 17.5 * ceil_div((int)b32, 256) ? 4 : 2;
    const int force = getenv("STTS_WN_X3") ? atoi(getenv("STTS_WN_X3")) : 0;  // tests / tools: 2 / 4 = block shape, -1 = the f32 kernel
    if (force == 1 || force == 2 || force == 4) fusedx3_rt = force;
    if (force == -1) fusedx3_rt = 0;
  }
  // small batches (the f32 kernel would run its direct form on 16-row blocks, M = 1): the split kernel on 16-row blocks too
  if (c->prec == PREC_F32 && fused_m == 1 && c->flow[0].fused.ready_x3) {
    fusedx3_rt = 1;
    const int force = getenv("STTS_WN_X3") ? atoi(getenv("STTS_WN_X3")) : 0;
    if (force == 1 || force == 2 || force == 4) fusedx3_rt = force;
    if (force == -1) fusedx3_rt = 0;
  }
  // ... and one launch per COUPLING layer (wn_block_x3_kernel: four WaveNet layers + post + coupling + next pre, h and `out` on chip) with 32 output
  // rows per block (48 computed), 48 (64 computed: what fits the LDS next to the fp32 residual stream) once the 32-row blocks exceed one chip round
  int blockx3_rt = 0;
  if (fusedx3_rt) {
    long b32 = 0;
    for (int u = 0; u < s.n_utt; ++u) b32 += ceil_div(s.host[u + 1] - s.host[u], 32);
    (void)b32;
    // (built, parity-tested, NOT selected: 8 launches of 98 us against 32 of 22.6 at B = 8 - 0.79 vs 0.72 ms per step - and 1.80 vs 1.30 ms at B = 16: the
    //  2 x 8 halo rows are 1.5 x the matrix work of a 32-row block and the four layers of a block run back to back on one wave per SIMD, which costs
    //  more than the 24 launch ramps + prologues it saves.  STTS_WN_X3B=3 | 4 selects it for experiments.)
    const int force = getenv("STTS_WN_X3B") ? atoi(getenv("STTS_WN_X3B")) : 0;  // tests / tools: 3 / 4 = tile shape
    if (force == 3 || force == 4) blockx3_rt = force;
    if (force == -1) blockx3_rt = 0;
  }
  auto wptr = [&](const PackedConv& pc) -> const void* { return c->prec != PREC_F32 ? (const void*)pc.W16 : (const void*)pc.W; };
  float* blk_in = hf;  // wn_block16_kernel: the coupling layer's h_0 (ping-pongs between hf and hf2)
  for (int f = 7; f >= 0; --f) {
    const FlowLayerW& L = c->flow[f];
    const int p = f & 1;
    if (generic) {
      // ResidualCouplingLayer.forward(reverse) as plain contractions (flow.py:196-218, WN :63-88): pre -> 4 x {conv k5 with
      // the gate in its epilogue, res/skip with the h / out split in its epilogue} -> post with the coupling in its epilogue
      GemmArgs a = gemm_args(s);
      set_seg(a, 0, z, fh, p * half, L.pre);
      a.N = fh; a.bias = L.pre.bias; a.Y = hf; a.ldy = fh;
      STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, L.pre.npad, s.n_utt, ml));
      for (int i = 0; i < 4; ++i) {
        GemmArgs g = gemm_args(s);
        set_seg(g, 0, hf, fh, 0, L.in[i]);
        g.N = fh; g.bias = L.in[i].bias; g.Y = actsg; g.ldy = fh;
        g.gate = cond; g.ld_gate = c->flow_style.ld(); g.gcol0 = L.cond_col0 + i * 2 * fh; g.gC = fh;
        STTS_TRY(launch_conv_gemm(st, g, EPI_GATE, L.in[i].npad, s.n_utt, ml));
        GemmArgs r = gemm_args(s);
        set_seg(r, 0, actsg, fh, 0, L.rs[i]);
        r.N = L.rs[i].N; r.bias = L.rs[i].bias;
        r.D0 = hf; r.ldd0 = fh; r.acc0 = 1;                 // h += rs[:fh]   (every row tile reads only its own rows of `acts`)
        r.D1 = outf; r.ldd1 = fh; r.acc1 = i > 0;           // out (+)= rs[fh:] ; last layer: all of rs
        r.nsplit = L.rs[i].N == 2 * fh ? fh : 0;
        STTS_TRY(launch_conv_gemm(st, r, EPI_SPLIT_ACC, L.rs[i].npad, s.n_utt, ml));
      }
      GemmArgs q = gemm_args(s);
      set_seg(q, 0, outf, fh, 0, L.proj);
      q.N = half; q.bias = L.proj.bias; q.Z = z; q.ldz = fh; q.zcol0 = (1 - p) * half;
      STTS_TRY(launch_conv_gemm(st, q, EPI_COUPLE, L.proj.npad, s.n_utt, ml));
      continue;
    }
    if (f == 7) {  // later blocks get their `pre` from the tail of the previous block's last WaveNet launch
      GemmArgs a = gemm_args(s);
      set_seg(a, 0, z, fh, p * half, L.pre);
      a.N = fh; a.bias = L.pre.bias; a.Y = hf; a.ldy = fh;
      STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, L.pre.npad, s.n_utt, ml));
    }
    if (blockx3_rt) {
      // ---- split fp32: one launch for the whole coupling layer; reads h_0 = pre(z0) from `blk_in`, writes the next coupling layer's h_0 to the other buffer
      WnBlockX3Args ba;
      memset(&ba, 0, sizeof(ba));
      ba.Hin = blk_in; ba.seg_off = s.dev; ba.gate = cond; ba.ld_gate = c->flow_style.ld();
      double flops = 0;
      for (int i = 0; i < 4; ++i) {
        ba.W1[i] = L.fused.X1[i]; ba.b1[i] = L.fused.b1[i]; ba.W2[i] = L.fused.X2b[i]; ba.b2[i] = L.fused.b2[i]; ba.p2[i] = L.fused.xp2[i];
        ba.gcol0[i] = L.cond_col0 + i * 2 * fh;
        flops += 2.0 * (double)R * 2 * fh * 5 * fh + 2.0 * (double)R * (double)L.rs[i].N * fh;
      }
      ba.p1 = L.fused.xp1; ba.p3 = L.fused.xp3;
      ba.tail = f > 0 ? 2 : 1;
      ba.W3 = L.fused.X3; ba.b3m = L.fused.b3m; ba.b3s = L.fused.b3s; ba.Z = z; ba.ldz = fh; ba.zcol0 = (1 - p) * half;
      flops += 2.0 * (double)R * fh * fh;
      float* blk_out = blk_in == hf ? hf2 : hf;
      if (f > 0) {
        ba.W4 = c->flow[f - 1].fused.X4; ba.p4 = c->flow[f - 1].fused.xp4; ba.b4 = c->flow[f - 1].fused.b4; ba.Hpre = blk_out;
        flops += 2.0 * (double)R * fh * half;
      }
      GemmProfiler& prof = gemm_profiler();
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (prof.on) {
        e0 = prof.next();
        e1 = prof.next();
        // executed = the algorithmic flops x the halo recompute (16 RT computed rows per 16 RT - 16 output rows)
        prof.add("wn_block_kernel_x3", 0, flops, flops * (16.0 * blockx3_rt) / (16.0 * blockx3_rt - 16.0), 0.0);
      }
      const int out_rows = 16 * blockx3_rt - 2 * kWnBlockX3Halo;
      const dim3 bgrid(ceil_div(ml, out_rows), s.n_utt);
      if (blockx3_rt == 3) STTS_LAUNCH_TIMED((wn_block_x3_kernel<3>), bgrid, dim3(64 * kWnWaves), st, e0, e1, ba);
      else STTS_LAUNCH_TIMED((wn_block_x3_kernel<4>), bgrid, dim3(64 * kWnWaves), st, e0, e1, ba);
      blk_in = blk_out;
      STTS_HIP(hipGetLastError());
      continue;
    }
    if (fused16_rt == 16) {
      // ---- one launch for the whole coupling layer (16-bit modes, large batches): reads h_0 = pre(z0) from `blk_in`, writes the next
      // coupling layer's h_0 to the other buffer (neighbouring blocks still read their halo rows of `blk_in`)
      WnBlock16Args ba;
      memset(&ba, 0, sizeof(ba));
      ba.Hin = blk_in; ba.seg_off = s.dev; ba.gate = cond; ba.ld_gate = c->flow_style.ld();
      double flops = 0;
      for (int i = 0; i < 4; ++i) {
        ba.W1[i] = L.fused.H1[i]; ba.b1[i] = L.fused.b1[i]; ba.W2[i] = L.fused.H2b[i]; ba.b2[i] = L.fused.b2[i];
        ba.gcol0[i] = L.cond_col0 + i * 2 * fh;
        flops += 2.0 * (double)R * 2 * fh * 5 * fh + 2.0 * (double)R * (double)L.rs[i].N * fh;
      }
      ba.tail = f > 0 ? 2 : 1;
      ba.W3 = L.fused.H3; ba.b3m = L.fused.b3m; ba.b3s = L.fused.b3s; ba.Z = z; ba.ldz = fh; ba.zcol0 = (1 - p) * half;
      flops += 2.0 * (double)R * fh * fh;
      float* blk_out = blk_in == hf ? hf2 : hf;
      if (f > 0) {
        ba.W4 = c->flow[f - 1].fused.H4; ba.b4 = c->flow[f - 1].fused.b4; ba.Hpre = blk_out;
        flops += 2.0 * (double)R * fh * half;
      }
      GemmProfiler& prof = gemm_profiler();
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (prof.on) {
        e0 = prof.next();
        e1 = prof.next();
        prof.add("wn_block16_kernel", 0, flops, flops, 0.0);
      }
      const dim3 bgrid(ceil_div(ml, kWnBlockRows), s.n_utt);
      if (c->prec == PREC_BF16) STTS_LAUNCH_TIMED((wn_block16_kernel<PREC_BF16>), bgrid, dim3(64 * kWnWaves), st, e0, e1, ba);
      else STTS_LAUNCH_TIMED((wn_block16_kernel<PREC_F16>), bgrid, dim3(64 * kWnWaves), st, e0, e1, ba);
      blk_in = blk_out;
      STTS_HIP(hipGetLastError());
      continue;
    }
    float* hcur = hf;
    float* hnext = hf2;
    for (int i = 0; i < 4; ++i) {
      // one launch per WaveNet layer: conv k5 + gate + res/skip + h/out update (wn_layer.hip.h); the last one also
      // applies the block's post projection + reverse coupling and the next block's pre projection to its rows
      WnArgs w;
      w.Hin = hcur; w.Hout = i < 3 ? hnext : nullptr; w.Out = outf; w.seg_off = s.dev;
      w.Win = wptr(L.in[i]); w.bin = L.in[i].bias; w.Wrs = wptr(L.rs[i]); w.brs = L.rs[i].bias;
      w.gate = cond; w.ld_gate = c->flow_style.ld(); w.gcol0 = L.cond_col0 + i * 2 * fh;
      w.n_rs = L.rs[i].N; w.out_acc = i > 0;
      w.tail = 0; w.Wproj = w.Wpre = nullptr; w.bproj = w.bpre = nullptr; w.Z = w.Hpre = nullptr; w.ldz = w.zcol0 = 0;
      double extra = 0;
      if (i == 3) {
        w.tail = f > 0 ? 2 : 1;
        w.Wproj = wptr(L.proj); w.bproj = L.proj.bias; w.Z = z; w.ldz = fh; w.zcol0 = (1 - p) * half;
        extra = 2.0 * (double)R * fh * fh;
        if (f > 0) {
          const FlowLayerW& nx = c->flow[f - 1];
          w.Wpre = wptr(nx.pre); w.bpre = nx.pre.bias; w.Hpre = hf;  // layer 3 reads hf2; hf is free and is the next block's h
          extra += 2.0 * (double)R * fh * half;
        }
      }
      GemmProfiler& prof = gemm_profiler();
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (prof.on) {
        e0 = prof.next();
        e1 = prof.next();
        const double rest = 2.0 * (double)R * (double)L.rs[i].N * fh + extra, conv = 2.0 * (double)R * 2 * fh * 5 * fh;
        // the fused kernel executes F(M,5): M + 4 instead of 5 M products per channel and group of M rows
        prof.add(fusedx3_rt ? "wn_fused_kernel_x3" : fused_m ? "wn_fused_kernel" : (fused16_rt ? "wn_fused16_kernel" : "wn_layer_kernel"), 0, conv + rest,
                 ((fused_m && !fusedx3_rt) ? conv * (fused_m + 4) / (5.0 * fused_m) : conv) + rest, 0.0);
      }
      const dim3 wgrid(ceil_div(ml, 32), s.n_utt);
      if (fusedx3_rt) {
        WnFusedX3Args xa;
        memset(&xa, 0, sizeof(xa));
        WnFused16Args& fa = xa.b;
        fa.Hin = hcur; fa.Hout = w.Hout; fa.Out = outf; fa.seg_off = s.dev;
        fa.W1 = L.fused.X1[i]; fa.b1 = L.fused.b1[i]; fa.W2 = L.fused.X2[i]; fa.b2 = L.fused.b2[i];
        xa.p1 = L.fused.xp1; xa.p2 = L.fused.xp2[i]; xa.p3 = L.fused.xp3;
        fa.gate = cond; fa.ld_gate = w.ld_gate; fa.gcol0 = w.gcol0; fa.out_acc = w.out_acc; fa.tail = w.tail;
        fa.W3 = L.fused.X3; fa.b3m = L.fused.b3m; fa.b3s = L.fused.b3s; fa.Z = w.Z; fa.ldz = w.ldz; fa.zcol0 = w.zcol0;
        if (w.tail > 1) { fa.W4 = c->flow[f - 1].fused.X4; xa.p4 = c->flow[f - 1].fused.xp4; fa.b4 = c->flow[f - 1].fused.b4; fa.Hpre = w.Hpre; }
        if (s.n_utt <= kWnSegInline && !s.cap) {
          fa.n_inline = s.n_utt;
          memcpy(fa.seg_inline, s.host, (s.n_utt + 1) * sizeof(int));
        }
        const dim3 fgrid(ceil_div(ml, 16 * fusedx3_rt), s.n_utt);
#ifdef STTS_WN_TRACE
        xa.dbg = wn_trace_buffer((f * 4 + i), (long)fgrid.x * fgrid.y);
#endif
        static const int x3_waves = getenv("STTS_WN_X3_WAVES") ? atoi(getenv("STTS_WN_X3_WAVES")) : 8;  // experiments: 4 waves per block (default: eight, wn_fused_x3.hip.h)
        const bool eight = x3_waves != 4;
#define STTS_WNX3(RT_)                                                                                                                  \
  do {                                                                                                                                   \
    if (eight) {                                                                                                                         \
      if (i == 3) STTS_LAUNCH_TIMED((wn_fused_x3_kernel<RT_, true, 2 * kWnWaves>), fgrid, dim3(128 * kWnWaves), st, e0, e1, xa);        \
      else STTS_LAUNCH_TIMED((wn_fused_x3_kernel<RT_, false, 2 * kWnWaves>), fgrid, dim3(128 * kWnWaves), st, e0, e1, xa);             \
    } else {                                                                                                                             \
      if (i == 3) STTS_LAUNCH_TIMED((wn_fused_x3_kernel<RT_, true>), fgrid, dim3(64 * kWnWaves), st, e0, e1, xa);                       \
      else STTS_LAUNCH_TIMED((wn_fused_x3_kernel<RT_, false>), fgrid, dim3(64 * kWnWaves), st, e0, e1, xa);                            \
    }                                                                                                                                    \
  } while (0)
        if (fusedx3_rt == 4) STTS_WNX3(4);
        else if (fusedx3_rt == 1) STTS_WNX3(1);
        else STTS_WNX3(2);
#undef STTS_WNX3
      }
      else if (fused_m) {
        auto launch = [&](auto mtag) {
          constexpr int M = decltype(mtag)::value;
          WnFusedArgs<M> fa;
          memset(&fa, 0, sizeof(fa));
          fa.Hin = hcur; fa.Hout = w.Hout; fa.Out = outf; fa.seg_off = s.dev;
          fa.W1 = L.fused.W1[M == 2 ? 0 : (M == 4 ? 1 : 2)][i]; fa.b1 = L.fused.b1[i]; fa.W2 = L.fused.W2[i]; fa.b2 = L.fused.b2[i];
          fa.gate = cond; fa.ld_gate = w.ld_gate; fa.gcol0 = w.gcol0; fa.out_acc = w.out_acc; fa.tail = w.tail;
          fa.W3 = L.fused.W3; fa.b3m = L.fused.b3m; fa.b3s = L.fused.b3s; fa.Z = w.Z; fa.ldz = w.ldz; fa.zcol0 = w.zcol0;
          if (w.tail > 1) { fa.W4 = c->flow[f - 1].fused.W4; fa.b4 = c->flow[f - 1].fused.b4; fa.Hpre = w.Hpre; }
          if (s.n_utt <= kWnSegInline && !s.cap) {  // (the inlined offsets are the host's: not with capacity segments)
            fa.n_inline = s.n_utt;
            memcpy(fa.seg_inline, s.host, (s.n_utt + 1) * sizeof(int));
          }
          const dim3 fgrid(ceil_div(ml, 16 * M), s.n_utt);
#ifdef STTS_WN_TRACE
          fa.dbg = wn_trace_buffer((f * 4 + i), (long)fgrid.x * fgrid.y);
#endif
          if (i == 3) STTS_LAUNCH_TIMED((wn_fused_kernel<M, true>), fgrid, dim3(64 * kWnWaves), st, e0, e1, fa);
          else STTS_LAUNCH_TIMED((wn_fused_kernel<M, false>), fgrid, dim3(64 * kWnWaves), st, e0, e1, fa);
        };
        if (fused_m == 1) launch(std::integral_constant<int, 1>{});
        else if (fused_m == 4) launch(std::integral_constant<int, 4>{});
        else launch(std::integral_constant<int, 2>{});
      }
      else if (fused16_rt) {
        WnFused16Args fa;
        memset(&fa, 0, sizeof(fa));
        fa.Hin = hcur; fa.Hout = w.Hout; fa.Out = outf; fa.seg_off = s.dev;
        fa.W1 = L.fused.H1[i]; fa.b1 = L.fused.b1[i]; fa.W2 = L.fused.H2[i]; fa.b2 = L.fused.b2[i];
        fa.gate = cond; fa.ld_gate = w.ld_gate; fa.gcol0 = w.gcol0; fa.out_acc = w.out_acc; fa.tail = w.tail;
        fa.W3 = L.fused.H3; fa.b3m = L.fused.b3m; fa.b3s = L.fused.b3s; fa.Z = w.Z; fa.ldz = w.ldz; fa.zcol0 = w.zcol0;
        if (w.tail > 1) { fa.W4 = c->flow[f - 1].fused.H4; fa.b4 = c->flow[f - 1].fused.b4; fa.Hpre = w.Hpre; }
        if (s.n_utt <= kWnSegInline && !s.cap) {
          fa.n_inline = s.n_utt;
          memcpy(fa.seg_inline, s.host, (s.n_utt + 1) * sizeof(int));
        }
        auto launch16 = [&](auto ptag, auto rtag) {
          constexpr int P = decltype(ptag)::value, RTv = decltype(rtag)::value;
          const dim3 fgrid(ceil_div(ml, 16 * RTv), s.n_utt);
          if (i == 3) STTS_LAUNCH_TIMED((wn_fused16_kernel<P, RTv, true>), fgrid, dim3(64 * kWnWaves), st, e0, e1, fa);
          else STTS_LAUNCH_TIMED((wn_fused16_kernel<P, RTv, false>), fgrid, dim3(64 * kWnWaves), st, e0, e1, fa);
        };
        using I1 = std::integral_constant<int, PREC_BF16>;
        using I2 = std::integral_constant<int, PREC_F16>;
        using R4 = std::integral_constant<int, 4>;
        using R8 = std::integral_constant<int, 8>;
        if (c->prec == PREC_BF16) { if (fused16_rt == 8) launch16(I1{}, R8{}); else launch16(I1{}, R4{}); }
        else { if (fused16_rt == 8) launch16(I2{}, R8{}); else launch16(I2{}, R4{}); }
      }
      // small batches (fp32): 16-row blocks, twice the workgroups at half the chain length (wn_layer_small.hip.h)
      else if (rows16) STTS_LAUNCH_TIMED(wn_layer_rows16_kernel, dim3(ceil_div(ml, 16), s.n_utt), dim3(1024), st, e0, e1, w);
      else if (c->prec == PREC_BF16) STTS_LAUNCH_TIMED(wn_layer_kernel<PREC_BF16>, wgrid, dim3(1024), st, e0, e1, w);
      else if (c->prec == PREC_F16) STTS_LAUNCH_TIMED(wn_layer_kernel<PREC_F16>, wgrid, dim3(1024), st, e0, e1, w);
      else STTS_LAUNCH_TIMED(wn_layer_kernel<PREC_F32>, wgrid, dim3(1024), st, e0, e1, w);
      if (const char* dbgs = getenv("STTS_WN_DEBUG")) {  // diagnostics: stop after launch #n and hand back h (n > 0) or out (n < 0)
        const int n = atoi(dbgs), k = (7 - f) * 4 + i + 1;
        if (z_flow_out && (n == k || n == -k)) {
          STTS_HIP(hipMemcpyAsync(z_flow_out, n > 0 ? (i < 3 ? hnext : hf) : outf, R * fh * sizeof(float), hipMemcpyDeviceToDevice, st));
          return 0;
        }
      }
      std::swap(hcur, hnext);
    }
    STTS_HIP(hipGetLastError());
  }
#ifdef STTS_WN_TRACE
  if (fused_m || fusedx3_rt) wn_trace_report(st);
#endif
  if (z_flow_out) STTS_HIP(hipMemcpyAsync(z_flow_out, z, R * fh * sizeof(float), hipMemcpyDeviceToDevice, st));
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, z, fh, 0, c->post_flow);
  a.N = d.dec_hidden; a.bias = c->post_flow.bias; a.Y = mel; a.ldy = ld_mel;
  if (mel16 && c->post_flow.prec != PREC_F32) {
    a.Y16 = mel16;
    a.ldy16 = ld_mel16;
  }
  STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, c->post_flow.npad, s.n_utt, ml));
  if (mel16 && c->post_flow.prec == PREC_F32) launch_cast_rows(st, c->prec, mel, ld_mel, d.dec_hidden, mel16, ld_mel16, R);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// stage: harmonic source + STFT (models/generator.py:247-315, :32-44, :406-410)
// ------------------------------------------------------------------------------------------------
inline int harmonic_stft(stts_ctx* c, hipStream_t st, const Seg& s, const float* pitch, const float* noise, const float* init_phase,
                         int batch_scope, float* prior_out, float* har_spec, float* har_phase, int ld, Arena& ws, int out16 = 0) {
  const long R = s.rows();
  double* prefix = ws.get<double>(R);
  float* stats = ws.get<float>(2 * s.n_utt);
  float* sig = prior_out ? prior_out : ws.get<float>(R * kHop);
  STTS_CHECK(ws.ok, "harmonic_stft: workspace too small");
  STTS_CHECK(ld >= kBins && ld <= 64 * ((kBins + 63) / 64), "harmonic_stft: row stride %d outside [%d, %d]", ld, kBins, 64 * ((kBins + 63) / 64));
  STTS_DRY_RETURN(ws);
  if (!s.cap)  // (capacity segments: pcph_kernel checks the real lengths and raises the device error word 4)
    for (int u = 0; u < s.n_utt; ++u)
      STTS_CHECK((long)(s.host[u + 1] - s.host[u]) * kHop > kNfft / 2, "utterance %d too short for reflect padding (%d frames; need > %d samples)", u,
                 s.host[u + 1] - s.host[u], kNfft / 2);
  STTS_LAUNCH_PROF("pcph_prep_kernel", (size_t)R * 12, pcph_prep_kernel, dim3(s.n_utt), dim3(256), st, pitch, s.dev, prefix, stats);
  STTS_LAUNCH_PROF("pcph_kernel", (size_t)R * kHop * 8, pcph_kernel, dim3(std::min(1024, ceil_div(s.max_len() * kHop, 256)), s.n_utt), dim3(256), st, pitch, s.dev, s.n_utt,
                     prefix, stats, noise, init_phase, batch_scope, sig, c->d_err);
  STTS_LAUNCH_PROF("stft_kernel", (size_t)R * (kHop + 2 * kBins) * 4, stft_kernel, dim3((s.max_len() + kFftWaves - 1) / kFftWaves, s.n_utt), dim3(64 * kFftWaves), st, sig, s.dev, c->hann, c->twiddle64, har_spec, har_phase, ld, out16);
  STTS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// stage: vocoder body + iSTFT (models/generator.py:412-433)
// ------------------------------------------------------------------------------------------------
inline int ln_launch(hipStream_t st, const float* X, int ldx, int C, long n_rows, const int* row_utt, float eps, int adaptive, int nout,
                     const LnOut& o0, const LnOut& o1, int act, const LnIn& in = LnIn{}, const int* n_rows_dev = nullptr) {
  STTS_LAUNCH_PROF("row_layernorm_kernel", (size_t)n_rows * C * (1 + nout) * 4, row_layernorm_kernel, dim3((unsigned)ceil_div((int)n_rows, 4)), dim3(256), st, X, ldx, C, (int)n_rows, row_utt, eps,
                     adaptive, nout, o0, o1, act, in, n_rows_dev);
  STTS_HIP(hipGetLastError());
  return 0;
}

// prior convs (generator.py:412-413) write straight into the concat slots [h, h+hp) of the two head inputs.  The two
// convs are independent (and independent of decoder/flow), so the caller may put them on different streams.
inline bool vocoder_rows16(const stts_ctx* c, long rows) { return c->prec != PREC_F32 && rows >= rows16_threshold(); }

// 16-bit operand modes, large batches: the head inputs are 16-bit row buffers (`head` reinterpreted, [rows, hc] elements) and
// the conv reads a rounded copy of har (har16: scratch of rows * ld_har elements).
inline int prior_conv(stts_ctx* c, hipStream_t st, const Seg& s, int which, const float* har, int ld_har, float* head, WinoScratch* wino = nullptr,
                      unsigned short* har16 = nullptr, bool har_is16 = false) {
  const int h = c->d.gen_hidden, hp = h / 2, hc = h + hp;
  const PackedConv& w = which == 0 ? c->amp_prior : c->phase_prior;
  if (wino && *wino && c->wino_prior[which].ready)  // k = 7 in Winograd F(6,7) form: 12 instead of 42 multiplies per 6 outputs
    return run_winograd(st, s, har, ld_har, c->wino_prior[which], head + h, hc, ACT_NONE, nullptr, 0, 1.0f, *wino);
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, har, ld_har, 0, w);
  a.N = hp; a.bias = w.bias;
  if (vocoder_rows16(c, s.rows())) {
    STTS_CHECK((har16 || har_is16) && ld_har % 8 == 0, "prior_conv: 16-bit mode needs the rounded-copy scratch");
    if (!har_is16) launch_cast_rows(st, c->prec, har, ld_har, kBins, har16, ld_har, s.rows());
    a.seg[0].X = har_is16 ? har : reinterpret_cast<const float*>(har16);
    a.x16 = 1;
    a.Y = nullptr; a.Y16 = reinterpret_cast<unsigned short*>(head); a.ldy16 = hc; a.ycol16 = h;
  } else {
    a.Y = head; a.ldy = hc; a.ycol0 = h;
  }
  return launch_conv_gemm(st, a, EPI_STORE, w.npad, s.n_utt, s.max_len());
}

// everything after the prior convs: projector, ConvNeXt blocks, heads, output convs, iSTFT (generator.py:414-433).
// headA / headP [rows, 768]: columns [512, 768) already hold logamp_prior / phase_prior.
inline int vocoder_body(stts_ctx* c, hipStream_t st, const Seg& s, const float* mel, int ld_mel, const float* style, float* headA, float* headP,
                        float* audio, float* logamp_out, float* phase_out, int ld_lp, Arena& ws, const unsigned short* mel16_in = nullptr) {
  // mel16_in: [rows, round_up(gen_input, 32)] mel already rounded to the operand precision by its producer (frame_path)
  const stts_model_dims& d = c->d;
  const long R = s.rows();
  const int h = d.gen_hidden, hp = h / 2, hc = h + hp, inter = d.gen_inter, ml = s.max_len();
  const int ldlp = round_up(kBins, 32);
  float* xa = ws.get<float>(R * h);
  float* xb = ws.get<float>(R * h);
  float* dw = ws.get<float>(R * h);
  float* nrm = ws.get<float>(R * h);
  float* U = ws.get<float>(R * inter);
  const int ss_stride = ceil_div(ml, 128) * 4;
  float* part = ws.get<float>((size_t)s.n_utt * ss_stride * inter);
  float* gscale = ws.get<float>((size_t)s.n_utt * inter);
  float* w2u = ws.get<float>((size_t)s.n_utt * c->cnx[0].pw2.npad * inter);
  float* sty = ws.get<float>((size_t)s.n_utt * c->gen_style.ld());
  int* row_utt = ws.get<int>(R);
  float* la = logamp_out ? logamp_out : ws.get<float>(R * ldlp);
  float* ph = phase_out ? phase_out : ws.get<float>(R * ldlp);
  const int ldl = logamp_out ? ld_lp : ldlp;
  // 16-bit operand modes: every contraction input of the body is a 16-bit row buffer written by its producer (LayerNorm
  // outputs, the SiLU output of pwconv1, the head inputs) - nrm / U / headA / headP are reinterpreted; mel gets a rounded copy
  // (from rows16_threshold() rows on: below that the contractions are latency-bound one-round launches and the extra copies cost more
  //  than the staging they save)
  const int p16 = (c->prec != PREC_F32 && vocoder_rows16(c, R)) ? c->prec : 0;
  unsigned short* mel16 = (p16 && !mel16_in) ? ws.get<unsigned short>(R * round_up(d.gen_input, 32)) : nullptr;
  float* yw = ws.get<float>((R + s.n_utt) * kWin);
  WinoScratch wino;
  if (c->wino_out[0].ready && c->wino_out[1].ready) wino.p = ws.get<float>(wino_scratch_floats(s, c->wino_out[0]));
  STTS_CHECK(ws.ok, "vocoder: workspace too small");
  STTS_DRY_RETURN(ws);
  STTS_CHECK(!logamp_out == !phase_out, "logamp_out and phase_out must be given together");
  const int lds = c->gen_style.ld();
  STTS_TRY(run_style(st, c->gen_style, style, s.n_utt, sty));
  STTS_LAUNCH_PROF("row_utt_kernel", (size_t)R * 4, row_utt_kernel, dim3(ceil_div(ml, 256), s.n_utt), dim3(256), st, s.dev, s.n_utt, row_utt);
  // projector over cat[mel, logamp_prior, phase_prior] as three K segments (generator.py:414)
  {
    GemmArgs a = gemm_args(s);
    if (p16) {
      const int ldm16 = round_up(d.gen_input, 32);
      if (!mel16_in) launch_cast_rows(st, p16, mel, ld_mel, d.gen_input, mel16, ldm16, R);
      set_seg(a, 0, reinterpret_cast<const float*>(mel16_in ? mel16_in : mel16), ldm16, 0, c->proj_mel);
      a.x16 = 1;
    } else {
      set_seg(a, 0, mel, ld_mel, 0, c->proj_mel);
    }
    set_seg(a, 1, headA, hc, h, c->proj_la);
    set_seg(a, 2, headP, hc, h, c->proj_ph);
    a.N = h; a.bias = c->proj_mel.bias; a.Y = xa; a.ldy = h;
    STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, c->proj_mel.npad, s.n_utt, ml));
  }
  float* cur = xa;
  float* nxt = xb;
  for (int i = 0; i < 4; ++i) {
    const ConvNextW& B = c->cnx[i];
    if (h <= kDwLnMaxC && h % 4 == 0 && (B.K == 3 || B.K == 7 || B.K == 15 || B.K == 31)) {
      // depthwise conv + adaptive LayerNorm in one launch (the [rows, h] intermediate never reaches HBM)
      // (32-row blocks for the long kernels - half the halo re-reads, two blocks per CU - measured no faster at B = 64: 64 vs 58 us)
      const dim3 fg(ceil_div(ml, 16), s.n_utt);
      const size_t fb = (size_t)R * h * (4 + (p16 ? 2 : 4));
#define STTS_DWLN(KK) STTS_LAUNCH_PROF("dwconv_ln_kernel", fb, (dwconv_ln_kernel<KK, 16>), fg, dim3(256), st, cur, h, h, s.dev, B.dw_wt, B.dw_b, 1e-6f, sty, lds, B.norm.col0, nrm, h, p16)
      if (B.K == 3) STTS_DWLN(3);
      else if (B.K == 7) STTS_DWLN(7);
      else if (B.K == 15) STTS_DWLN(15);
      else STTS_DWLN(31);
#undef STTS_DWLN
    } else {
      STTS_LAUNCH_PROF("dwconv_kernel", (size_t)R * h * 2 * 4, (dwconv_kernel<31>), dim3(ceil_div(h, 64), ceil_div(ml, 64), s.n_utt), dim3(256), st, cur, h, dw, h, h, s.dev, B.dw_wt,
                         B.dw_b, B.K, (int)ACT_NONE);
      LnOut o0{nrm, h, 0, sty, nullptr, lds, B.norm.col0, p16}, o1{};
      STTS_TRY(ln_launch(st, dw, h, h, R, row_utt, 1e-6f, 1, 1, o0, o1, ACT_NONE, LnIn{}, s.rows_dev()));
    }
    GemmArgs a = gemm_args(s);
    set_seg(a, 0, nrm, h, 0, B.pw1);
    a.N = inter; a.bias = B.pw1.bias; a.act = ACT_SILU;
    a.x16 = p16 != 0;
    if (p16) { a.Y = nullptr; a.Y16 = reinterpret_cast<unsigned short*>(U); a.ldy16 = inter; }  // GRN's sums of squares come from the fp32 values
    else { a.Y = U; a.ldy = inter; }
    a.sumsq_part = part; a.ld_ss = inter; a.ss_stride = ss_stride;
    STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, B.pw1.npad, s.n_utt, ml));
    STTS_LAUNCH_PROF("grn_gx_kernel", (size_t)s.n_utt * ss_stride * inter * 4, grn_gx_kernel, dim3(ceil_div(inter, 32), s.n_utt), dim3(256), st, part, inter, ss_stride, s.dev, inter, gscale, inter);
    GemmArgs b = gemm_args(s);
    set_seg(b, 0, U, inter, 0, B.pw2);
    if (B.pw2.prec == PREC_F32 && B.pw2.w16_plane > 0 && x3_enabled()) {
      // split fp32: GRN's per-(utterance, channel) factor scales the ACTIVATION while pwconv2's tile is staged (the contraction's input affine with
      // slope 1) instead of a per-utterance copy of the weight: the shared split planes of W stay valid and the scale_weight pass is gone
      hipLaunchKernelGGL(grn_xaff_kernel, dim3(s.n_utt), dim3(256), 0, st, gscale, inter, B.grn_gamma, w2u, B.pw2.kc, inter);
      b.xaff = w2u;
      b.ld_xaff = B.pw2.kc;
      b.xaff_slope = 1.0f;
      static const bool no_scale_only = getenv("STTS_XAFF_GENERAL") != nullptr;  // experiments: the general scale / shift / slope form
      b.xaff_scale_only = B.pw2.kc == inter && !no_scale_only;  // (no pad column: every staged value is a written one)
    } else {
    launch_scale_weight(st, dim3(128, s.n_utt), B.pw2.prec, B.pw2.W, gscale, inter, B.grn_gamma, w2u, B.pw2.npad, B.pw2.kc);
    b.seg[0].W = w2u;
    b.seg[0].W16 = reinterpret_cast<const unsigned short*>(w2u);
    b.seg[0].w16_plane = 0;
    b.seg[0].w_utt_stride = (long)B.pw2.npad * B.pw2.kc;
    }
    b.x16 = p16 != 0;
    b.N = h; b.bias = B.pw2.bias; b.Y = nxt; b.ldy = h; b.R = cur; b.ldr = h;
    STTS_TRY(launch_conv_gemm(st, b, EPI_STORE, B.pw2.npad, s.n_utt, ml));
    std::swap(cur, nxt);
  }
  // two AdaLN heads (eps 1e-5) into columns [0,512) of the head inputs (generator.py:417-423)
  {
    LnOut o0{headA, hc, 0, sty, nullptr, lds, c->head_amp.col0, p16}, o1{headP, hc, 0, sty, nullptr, lds, c->head_phase.col0, p16};
    STTS_TRY(ln_launch(st, cur, h, h, R, row_utt, 1e-5f, 1, 2, o0, o1, ACT_NONE, LnIn{}, s.rows_dev()));
  }
  {
    GemmArgs a = gemm_args(s);
    set_seg(a, 0, headA, hc, 0, c->amp_out);
    a.x16 = p16 != 0;
    a.N = kBins - 1; a.bias = c->amp_out.bias; a.Y = la; a.ldy = ldl;
    if (wino) STTS_TRY(run_winograd(st, s, headA, hc, c->wino_out[0], la, ldl, ACT_NONE, nullptr, 0, 1.0f, wino));
    else STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, c->amp_out.npad, s.n_utt, ml));
    GemmArgs b = gemm_args(s);
    set_seg(b, 0, headP, hc, 0, c->phase_out);
    b.x16 = p16 != 0;
    b.N = kBins - 1; b.bias = c->phase_out.bias; b.Y = ph; b.ldy = ldl;
    if (wino) STTS_TRY(run_winograd(st, s, headP, hc, c->wino_out[1], ph, ldl, ACT_NONE, nullptr, 0, 1.0f, wino));
    else STTS_TRY(launch_conv_gemm(st, b, EPI_STORE, c->phase_out.npad, s.n_utt, ml));
    const int kk = c->amp_out.ntaps;
    STTS_CHECK(kk <= kChanTaps, "output conv kernel size %d > %d", kk, kChanTaps);
    const ChanConvSet sa{headA, c->nyq_w[0], c->nyq_b[0], la}, sp{headP, c->nyq_w[1], c->nyq_b[1], ph};
    const dim3 cg((unsigned)ceil_div(ml, 4 * kChanRows), 2, s.n_utt);
    const size_t cb = (size_t)2 * R * (hc * (p16 ? 2 : 4) + 4);
    if (p16 == PREC_BF16) STTS_LAUNCH_PROF("single_channel_conv_kernel", cb, single_channel_conv_kernel<PREC_BF16>, cg, dim3(256), st, sa, sp, hc, hc, s.dev, kk, ldl, kBins - 1);
    else if (p16 == PREC_F16) STTS_LAUNCH_PROF("single_channel_conv_kernel", cb, single_channel_conv_kernel<PREC_F16>, cg, dim3(256), st, sa, sp, hc, hc, s.dev, kk, ldl, kBins - 1);
    else STTS_LAUNCH_PROF("single_channel_conv_kernel", cb, single_channel_conv_kernel<0>, cg, dim3(256), st, sa, sp, hc, hc, s.dev, kk, ldl, kBins - 1);
  }
  STTS_LAUNCH_PROF("istft_frames_kernel", (size_t)R * 2 * kBins * 4, istft_frames_kernel, dim3((ml + 1 + kFftWaves - 1) / kFftWaves, s.n_utt), dim3(64 * kFftWaves), st, la, ph, ldl, s.dev, c->hann, c->twiddle64, yw);
  STTS_LAUNCH_PROF("istft_ola_kernel", (size_t)R * kHop * 4, istft_ola_kernel, dim3(std::min(1024, ceil_div(ml * kHop, 256)), s.n_utt), dim3(256), st, yw, s.dev, c->hann, audio);
  STTS_HIP(hipGetLastError());
  return 0;
}

inline int vocoder_forward(stts_ctx* c, hipStream_t st, const Seg& s, const float* mel, int ld_mel, const float* style, const float* har_spec,
                           const float* har_phase, int ld_har, float* audio, float* logamp_out, float* phase_out, int ld_lp, Arena& ws) {
  const long R = s.rows();
  const int hc = c->d.gen_hidden + c->d.gen_hidden / 2;
  float* headA = ws.get<float>(R * hc);
  float* headP = ws.get<float>(R * hc);
  STTS_CHECK(ws.ok, "vocoder_forward: workspace too small");
  {
    Arena tmp(ws.base + ws.used, ws.cap - ws.used, ws.offset0 + ws.used);  // released again before vocoder_body carves its own buffers
    WinoScratch wino;
    if (c->wino_prior[0].ready) wino.p = tmp.get<float>(wino_scratch_floats(s, c->wino_prior[0]));
    unsigned short* har16 = c->prec != PREC_F32 ? tmp.get<unsigned short>(R * ld_har) : nullptr;
    STTS_CHECK(tmp.ok, "vocoder_forward: workspace too small");
    if (dry_run().on) {
      dry_run().peak = std::max(dry_run().peak, tmp.offset0 + tmp.used);
    } else {
      STTS_TRY(prior_conv(c, st, s, 0, har_spec, ld_har, headA, &wino, har16));
      STTS_TRY(prior_conv(c, st, s, 1, har_phase, ld_har, headP, &wino, har16));
      }
  }
  return vocoder_body(c, st, s, mel, ld_mel, style, headA, headP, audio, logamp_out, phase_out, ld_lp, ws);
}

// row stride of the harmonic spectra: the prior convs' packed input width (1056; 1088 in the 16-bit modes)
inline int har_ld(const stts_ctx* c) { return std::max(round_up(kBins, 32), c->amp_prior.kc); }

inline int frame_path(stts_ctx* c, hipStream_t st, const Seg& s, const float* asr, int ld_asr, const float* pitch, const float* energy,
                      const float* style, const float* prior_noise, const float* src_noise, const float* init_phase, int batch_scope,
                      float* audio, void* wsp, size_t ws_bytes) {
  const long R = s.rows();
  Arena top(wsp, ws_bytes);
  const int ldh = har_ld(c), hc = c->d.gen_hidden + c->d.gen_hidden / 2, dh = c->d.dec_hidden;
  float* x = top.get<float>(R * dh);
  float* mel = top.get<float>(R * dh);
  float* hs = top.get<float>(R * ldh);
  float* hp = top.get<float>(R * ldh);
  float* headA = top.get<float>(R * hc);
  float* headP = top.get<float>(R * hc);
  const int ldm16 = round_up(c->d.gen_input, 32);
  unsigned short* mel16 = (c->prec != PREC_F32 && vocoder_rows16(c, R) && c->d.gen_input == dh) ? top.get<unsigned short>(R * ldm16) : nullptr;
  const size_t side_bytes = (size_t)R * (sizeof(double) + kHop * sizeof(float)) + 4096 + 8 * s.n_utt;
  const size_t side_off = top.used;
  char* side_ws = top.get<char>(side_bytes);
  // (side stream, below: the prior convs then need scratch of their own - the decoder uses the stage region at the same time)
  static const long side_min_rows = getenv("STTS_SIDE_MIN_ROWS") ? atol(getenv("STTS_SIDE_MIN_ROWS")) : 3000;
  const bool side_cfg = c->prec == PREC_F32 && R > side_min_rows && c->wino_prior[0].ready;
  float* side_scratch = side_cfg ? top.get<float>(wino_scratch_floats(s, c->wino_prior[0])) : nullptr;
  STTS_CHECK(top.ok, "frame_path: workspace too small");
  const size_t mark = top.used;
  auto stage = [&]() {
    Arena a((char*)wsp + mark, ws_bytes - mark, mark);
    return a;
  };
  // Measured: running the (independent) source -> STFT -> prior-conv chain on side streams next to decoder/flow gains
  // < 1 % at B = 8 (7.51 vs 7.57 ms/step): the decoder GEMMs already fill the chip, so the stages stay on one stream.
  // 16-bit operand modes, large batches: the spectra are only read by the prior convs, so the STFT writes them as 16-bit rows
  const int h16 = vocoder_rows16(c, R) ? c->prec : 0;
  // fp32, large batches: the source -> STFT -> prior-conv chain (independent of decoder and flow until the vocoder) runs on a SIDE STREAM of the caller's
  // stream (fork / join events; the prior convs' scratch is its own region, carved above): it fills the chip while the decoder runs its small
  // bandwidth-bound kernels.  Same kernels, same results; same-box A/B (3-s utterances): B = 8: 4.43 -> 4.38 ms per step (+1.2 %), B = 4: 2.91 -> 2.86
  // (+1.6 %), B = 2: +0.8 %, B = 1: 1.76 -> 1.78 (the two cross-queue waits cost more than the overlap gains: off below 3 000 rows; STTS_SIDE_MIN_ROWS).
  // (Round 1, when the step took 7.5 ms and the prior convs were direct contractions, the same experiment gained < 1 %.)  Not while the per-launch profiler is on (its events belong to one stream),
  // not in the 16-bit modes (the persistent contraction kernel of the decoder and that of the prior convs would fight for the same CUs).
  const bool side_off_env = getenv("STTS_NO_SIDE_STREAM") != nullptr;  // experiments / tests (read per call)
  // (not while the caller's stream is being captured into a graph: the lane's stream and events are created on first use, which a capture may not allow)
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (side_scratch && !side_off_env) (void)hipStreamIsCapturing(st, &capturing);
  if (side_scratch && !dry_run().on && !gemm_profiler().on && !side_off_env && capturing == hipStreamCaptureStatusNone) {
    stts_ctx::SideLane lane;
    {
      std::lock_guard<std::mutex> lock(c->side_mu);
      stts_ctx::SideLane& l = c->side_lanes[st];
      if (!l.stream) {
        STTS_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        STTS_HIP(hipEventCreateWithFlags(&l.fork, hipEventDisableTiming));
        STTS_HIP(hipEventCreateWithFlags(&l.join, hipEventDisableTiming));
        STTS_HIP(hipStreamCreateWithFlags(&l.stream2, hipStreamNonBlocking));
        STTS_HIP(hipEventCreateWithFlags(&l.fork2, hipEventDisableTiming));
        STTS_HIP(hipEventCreateWithFlags(&l.join2, hipEventDisableTiming));
      }
      lane = l;
    }
    STTS_HIP(hipEventRecord(lane.fork, st));
    STTS_HIP(hipStreamWaitEvent(lane.stream, lane.fork, 0));
    // Whatever fails from here on, the caller's stream joins the side stream before this call returns: work that is still running there reads and
    // writes the caller's workspace, which the caller is free to reuse (in stream order) as soon as it has the status.
    auto forked = [&]() -> int {
      { Arena a(side_ws, side_bytes, side_off); STTS_TRY(harmonic_stft(c, lane.stream, s, pitch, src_noise, init_phase, batch_scope, nullptr, hs, hp, ldh, a, h16)); }
      WinoScratch wino;
      wino.p = side_scratch;
      STTS_TRY(prior_conv(c, lane.stream, s, 0, hs, ldh, headA, &wino, nullptr, h16 != 0));
      STTS_TRY(prior_conv(c, lane.stream, s, 1, hp, ldh, headP, &wino, nullptr, h16 != 0));
      return 0;
    };
    int rc = forked();
    STTS_HIP(hipEventRecord(lane.join, lane.stream));
    if (rc == 0) {
      auto main_part = [&]() -> int {
        SideWork sw;
        sw.stream = lane.stream2; sw.fork = lane.fork2; sw.join = lane.join2;
        { Arena a = stage(); STTS_TRY(decoder_forward(c, st, s, asr, ld_asr, pitch, energy, style, x, dh, a, &sw)); }
        { Arena a = stage(); STTS_TRY(prior_flow_forward(c, st, s, x, dh, style, prior_noise, mel, dh, nullptr, nullptr, a, mel16, ldm16)); }
        return 0;
      };
      rc = main_part();
    }
    STTS_HIP(hipStreamWaitEvent(st, lane.join, 0));
    if (rc) return rc;
    { Arena a = stage(); STTS_TRY(vocoder_body(c, st, s, mel, dh, style, headA, headP, audio, nullptr, nullptr, 0, a, mel16)); }
    return 0;
  }
  { Arena a(side_ws, side_bytes, side_off); STTS_TRY(harmonic_stft(c, st, s, pitch, src_noise, init_phase, batch_scope, nullptr, hs, hp, ldh, a, h16)); }
  {
    Arena a = stage();
    WinoScratch wino;
    if (c->wino_prior[0].ready) wino.p = a.get<float>(wino_scratch_floats(s, c->wino_prior[0]));
    STTS_CHECK(a.ok, "frame_path: workspace too small");
    if (dry_run().on) {
      dry_run().peak = std::max(dry_run().peak, a.offset0 + a.used);
    } else {
      STTS_TRY(prior_conv(c, st, s, 0, hs, ldh, headA, &wino, nullptr, h16 != 0));
      STTS_TRY(prior_conv(c, st, s, 1, hp, ldh, headP, &wino, nullptr, h16 != 0));
      }
  }
  { Arena a = stage(); STTS_TRY(decoder_forward(c, st, s, asr, ld_asr, pitch, energy, style, x, dh, a)); }
  { Arena a = stage(); STTS_TRY(prior_flow_forward(c, st, s, x, dh, style, prior_noise, mel, dh, nullptr, nullptr, a, mel16, ldm16)); }
  { Arena a = stage(); STTS_TRY(vocoder_body(c, st, s, mel, dh, style, headA, headP, audio, nullptr, nullptr, 0, a, mel16)); }
  return 0;
}

// Bytes of caller-owned workspace that every frame-rate entry point accepts for a batch of `R` rows in `n_utt` utterances,
// the longest `max_len` rows: a DRY RUN of the stages themselves over a fake address range (each carves its buffers, records
// how far it got and returns before its first launch), so the bound follows the model dims and whatever the stages really
// request.  Buffer sizes depend on the batch only through (R, n_utt, max_len); weights must be finalized (which convs have a
// Winograd form decides their scratch).
inline size_t frame_workspace_bytes(const stts_ctx* cc, int64_t R, int n_utt, int max_len) {
  stts_ctx* c = const_cast<stts_ctx*>(cc);
  if (R <= 0 || n_utt <= 0 || max_len <= 0) return 0;
  max_len = (int)std::min<int64_t>(max_len, R);
  std::vector<int> off(n_utt + 1, 0);  // any lengths with this total, count and maximum
  long rest = R - max_len;
  off[1] = max_len;
  for (int u = 1; u < n_utt; ++u) {
    const long len = std::max<long>(1, std::min<long>(max_len, rest - (n_utt - 1 - u)));
    off[u + 1] = off[u] + (int)len;
    rest -= len;
  }
  Seg s{n_utt, off.data(), nullptr};
  DryRun& d = dry_run();
  d.on = true;
  d.peak = 0;
  char* fake = reinterpret_cast<char*>((uintptr_t)1 << 20);
  const size_t cap = (size_t)1 << 46;
  (void)frame_path(c, nullptr, s, nullptr, c->d.inter_dim, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, fake, cap);
  {  // the staged entry points start their carving at offset 0; vocoder_forward with its own head buffers is the largest of them
    Arena a(fake, cap);
    (void)vocoder_forward(c, nullptr, s, nullptr, c->d.dec_hidden, nullptr, nullptr, nullptr, har_ld(c), nullptr, nullptr, nullptr, 0, a);
    Arena b(fake, cap);
    (void)decoder_forward(c, nullptr, s, nullptr, c->d.inter_dim, nullptr, nullptr, nullptr, nullptr, c->d.dec_hidden, b);
  }
  d.on = false;
  return d.peak + 4096;
}

}  // namespace stts

#include "phoneme_model.hip.h"
