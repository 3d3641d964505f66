// Phoneme-rate predictors: weight packing and stage orchestration.
//   TextEncoder          models/text_encoder.py:397-462   (x3: duration_predictor.text_encoder, speech_predictor.text_encoder, pe_text_encoder)
//   TextStyleEncoder     models/text_style_encoder.py:6-26 (x3)
//   ProsodyEncoder       models/prosody_encoder.py:10-81   (x2)
//   DurationPredictor    models/duration_predictor.py:8-36 ; DurationProcessor train/utils.py:385-494
//   PitchEnergyPredictor models/pitch_energy_predictor.py:11-121
// Included by model.hip.h after the frame-rate definitions (shares PackedConv, StyleTable, run_adain_block ...).
#pragma once

namespace stts {

struct TextEncW {
  bool ready = false;
  float* emb = nullptr;
  int C = 0, inter = 0, heads = 0, n_layers = 0, ffk = 0, filter = 0, tokens = 0;
  PackedConv pre[3], pre_proj, proj_m;
  float* pre_g[3];
  float* pre_b[3];
  struct Layer {
    PackedConv qkv, o, f1, f2;
    float *g1, *b1, *g2, *b2;
  } layer[16];
};

struct StyleEncW {
  bool ready = false;
  int inter = 0, sd = 0, n = 0;
  PackedConv conv_in;
  struct Blk {
    float *dw_wt, *dw_b, *ln_g, *ln_b, *grn_gamma;
    PackedConv pw1, pw2;
  } blk[8];
};

struct ProsodyW {
  int d = 0, C = 0, n_layers = 0, heads = 2;
  struct L {
    PackedConv qkv, o, f1, f2, proj;
    StyleSlot n1, n2;
  } l[8];
};

struct DurationW {
  bool ready = false;
  ProsodyW pros;
  StyleTable table;
  PackedConv proj;
};

struct PitchEnergyW {
  bool ready = false;
  ProsodyW pros;
  StyleTable table;
  StyleSlot qn, kn;
  PackedConv q, kv, o, cp2;
  float *cp_dw_wt = nullptr, *cp_dw_b = nullptr;
  AdainBlockW f0[3], n[3];
  float *f0_w = nullptr, *n_w = nullptr;
  float f0_b = 0.f, n_b = 0.f;
  int C = 0;
};

struct PhonemeModel {
  TextEncW te[3];    // 0 duration_predictor.text_encoder, 1 speech_predictor.text_encoder, 2 pe_text_encoder
  StyleEncW se[3];   // same order
  DurationW dur;
  PitchEnergyW pe;
};

static const char* kTextEncPrefix[3] = {"duration_predictor.text_encoder.", "speech_predictor.text_encoder.", "pe_text_encoder."};
static const char* kStyleEncPrefix[3] = {"duration_predictor.style_encoder.", "speech_predictor.style_encoder.", "pe_text_style_encoder."};

inline int upload_vec(stts_ctx* c, const std::string& name, float** out) {
  STTS_GET(t, name);
  return dev_upload(c, t->data, out);
}

// stack several [cout_i][cin][k] weights (and biases) along cout
inline int pack_stacked(stts_ctx* c, const std::vector<std::string>& names, PackedConv* out) {
  HostTensor w, b;
  for (size_t i = 0; i < names.size(); ++i) {
    HostTensor wi;
    STTS_TRY(get_weight(c, names[i], &wi));
    STTS_GET(bi, names[i] + ".bias");
    if (i == 0) {
      w = wi;
      b = *bi;
    } else {
      w.data.insert(w.data.end(), wi.data.begin(), wi.data.end());
      w.shape[0] += wi.shape[0];
      b.data.insert(b.data.end(), bi->data.begin(), bi->data.end());
    }
  }
  const int cin = (int)w.shape[1];
  return pack_rows(c, w, &b, plain_rows((int)w.shape[0]), 0, cin, round_up(cin, 32), (int)w.shape[0], out);
}

inline int pack_text_encoder(stts_ctx* c, const std::string& p, int inter, TextEncW* W) {
  const stts_model_dims& d = c->d;
  W->C = d.te_hidden; W->inter = inter; W->heads = d.te_heads; W->n_layers = d.te_layers; W->ffk = d.te_kernel; W->filter = d.te_filter;
  W->tokens = d.tokens;
  STTS_CHECK(W->C % 32 == 0 && W->n_layers <= 16, "text encoder: hidden_dim must be a multiple of 32, layers <= 16");
  STTS_TRY(upload_vec(c, p + "emb.weight", &W->emb));
  for (int i = 0; i < 3; ++i) {
    const std::string si = std::to_string(i);
    STTS_TRY(pack_plain(c, p + "prenet.conv_layers." + si, true, 0, W->C, &W->pre[i]));
    STTS_TRY(upload_vec(c, p + "prenet.norm_layers." + si + ".gamma", &W->pre_g[i]));
    STTS_TRY(upload_vec(c, p + "prenet.norm_layers." + si + ".beta", &W->pre_b[i]));
  }
  STTS_TRY(pack_plain(c, p + "prenet.proj", true, 0, W->C, &W->pre_proj));
  for (int i = 0; i < W->n_layers; ++i) {
    const std::string a = p + "encoder.attn_layers." + std::to_string(i) + ".";
    const std::string f = p + "encoder.ffn_layers." + std::to_string(i) + ".";
    TextEncW::Layer& L = W->layer[i];
    STTS_TRY(pack_stacked(c, {a + "conv_q", a + "conv_k", a + "conv_v"}, &L.qkv));
    STTS_TRY(pack_plain(c, a + "conv_o", true, 0, W->C, &L.o));
    STTS_TRY(pack_plain(c, f + "conv_1", true, 0, W->C, &L.f1));
    STTS_TRY(pack_plain(c, f + "conv_2", true, 0, W->filter, &L.f2));
    STTS_TRY(upload_vec(c, p + "encoder.norm_layers_1." + std::to_string(i) + ".gamma", &L.g1));
    STTS_TRY(upload_vec(c, p + "encoder.norm_layers_1." + std::to_string(i) + ".beta", &L.b1));
    STTS_TRY(upload_vec(c, p + "encoder.norm_layers_2." + std::to_string(i) + ".gamma", &L.g2));
    STTS_TRY(upload_vec(c, p + "encoder.norm_layers_2." + std::to_string(i) + ".beta", &L.b2));
  }
  STTS_TRY(pack_plain(c, p + "proj_m", true, 0, W->C, &W->proj_m));
  W->ready = true;
  return 0;
}

inline int pack_style_encoder(stts_ctx* c, const std::string& p, int inter, StyleEncW* W) {
  const stts_model_dims& d = c->d;
  W->inter = inter; W->sd = d.style_dim; W->n = d.style_layers;
  STTS_CHECK(W->n <= 8, "style encoder: at most 8 blocks");
  STTS_TRY(pack_plain(c, p + "conv_in", true, 0, inter, &W->conv_in));
  for (int i = 0; i < W->n; ++i) {
    const std::string q = p + "blocks." + std::to_string(i) + ".";
    StyleEncW::Blk& B = W->blk[i];
    STTS_GET(dw, q + "dwconv.weight");
    const int K = (int)dw->shape[2], C = (int)dw->shape[0];
    STTS_CHECK(K == 7 && C == W->sd, "style encoder dwconv shape");
    std::vector<float> wt((size_t)K * C);
    for (int ch = 0; ch < C; ++ch)
      for (int k = 0; k < K; ++k) wt[(size_t)k * C + ch] = dw->data[(size_t)ch * K + k];
    STTS_TRY(dev_upload(c, wt, &B.dw_wt));
    STTS_TRY(upload_vec(c, q + "dwconv.bias", &B.dw_b));
    STTS_TRY(upload_vec(c, q + "norm.weight", &B.ln_g));
    STTS_TRY(upload_vec(c, q + "norm.bias", &B.ln_b));
    STTS_TRY(pack_plain(c, q + "pwconv1", true, 0, W->sd, &B.pw1));
    HostTensor w2;
    STTS_TRY(get_weight(c, q + "pwconv2", &w2));
    STTS_GET(b2, q + "pwconv2.bias");
    STTS_GET(gb, q + "grn.beta");
    HostTensor b2f = *b2;
    const int ci = (int)w2.shape[1];
    for (int r = 0; r < (int)w2.shape[0]; ++r) {
      double s = 0;
      for (int k = 0; k < ci; ++k) s += (double)w2.data[(size_t)r * ci + k] * gb->data[k];
      b2f.data[r] = (float)((double)b2->data[r] + s);
    }
    STTS_TRY(pack_rows(c, w2, &b2f, plain_rows((int)w2.shape[0]), 0, ci, round_up(ci, 32), (int)w2.shape[0], &B.pw2));
    STTS_TRY(upload_vec(c, q + "grn.gamma", &B.grn_gamma));
  }
  W->ready = true;
  return 0;
}

inline int pack_prosody(stts_ctx* c, const std::string& p, int d_model, int n_layers, StyleTable* table, ProsodyW* W) {
  W->d = d_model; W->C = d_model + c->d.style_dim; W->n_layers = n_layers;
  STTS_CHECK(W->C % 32 == 0 && n_layers <= 8, "prosody encoder: channels must be a multiple of 32");
  for (int i = 0; i < n_layers; ++i) {
    const std::string si = std::to_string(i);
    const std::string a = p + "attn_layers." + si + ".";
    ProsodyW::L& L = W->l[i];
    STTS_TRY(pack_stacked(c, {a + "conv_q", a + "conv_k", a + "conv_v"}, &L.qkv));
    STTS_TRY(pack_plain(c, a + "conv_o", true, 0, W->C, &L.o));
    STTS_TRY(pack_plain(c, p + "ffn_layers." + si + ".conv_1", true, 0, W->C, &L.f1));
    STTS_TRY(pack_plain(c, p + "ffn_layers." + si + ".conv_2", true, 0, 2 * W->C, &L.f2));
    STTS_TRY(pack_plain(c, p + "proj_layers." + si, true, 0, W->C, &L.proj));
    STTS_TRY(add_style(c, table, p + "norm_layers_1." + si, W->C, &L.n1));
    STTS_TRY(add_style(c, table, p + "norm_layers_2." + si, W->C, &L.n2));
  }
  return 0;
}

inline int finalize_phoneme(stts_ctx* c, PhonemeModel* M, int which) {
  const stts_model_dims& d = c->d;
  const int inter[3] = {d.inter_dim, d.inter_dim, d.pe_inter};
  const int te_bit[3] = {STTS_W_DURATION, STTS_W_SPEECH_TEXT, STTS_W_PE_TEXT};
  const int se_bit[3] = {STTS_W_DURATION, STTS_W_SPEECH_TEXT, STTS_W_PE_STYLE};
  for (int i = 0; i < 3; ++i) {
    c->cur_tag = te_bit[i];
    if (which & te_bit[i]) STTS_TRY(pack_text_encoder(c, kTextEncPrefix[i], inter[i], &M->te[i]));
    c->cur_tag = se_bit[i];
    if (which & se_bit[i]) STTS_TRY(pack_style_encoder(c, kStyleEncPrefix[i], inter[i], &M->se[i]));
  }
  c->cur_tag = 0;
  if (which & STTS_W_DURATION) {
    c->cur_tag = STTS_W_DURATION;
    DurationW& D = M->dur;
    D = DurationW();
    STTS_TRY(pack_prosody(c, "duration_predictor.prosody_encoder.", d.inter_dim, d.dur_layers, &D.table, &D.pros));
    STTS_TRY(upload_table(c, &D.table));
    STTS_TRY(pack_plain(c, "duration_predictor.duration_proj.linear_layer", true, 0, D.pros.C, &D.proj));
    STTS_CHECK(d.dur_classes == 16, "duration decoding is specialised for the reference's 16-class table (train/utils.py:391-393)");
    D.ready = true;
  }
  if (which & STTS_W_PITCH_ENERGY) {
    c->cur_tag = STTS_W_PITCH_ENERGY;
    PitchEnergyW& P = M->pe;
    P = PitchEnergyW();
    const std::string p = "pitch_energy_predictor.";
    STTS_TRY(pack_prosody(c, p + "prosody_encoder.", d.pe_inter, 3, &P.table, &P.pros));
    P.C = P.pros.C;
    STTS_TRY(add_style(c, &P.table, p + "query_norm", P.C, &P.qn));
    STTS_TRY(add_style(c, &P.table, p + "key_norm", P.C, &P.kn));
    STTS_TRY(pack_plain(c, p + "cross_attention.conv_q", true, 0, P.C, &P.q));
    STTS_TRY(pack_stacked(c, {p + "cross_attention.conv_k", p + "cross_attention.conv_v"}, &P.kv));
    STTS_TRY(pack_plain(c, p + "cross_attention.conv_o", true, 0, P.C, &P.o));
    {
      HostTensor dw;
      STTS_TRY(get_weight(c, p + "cross_post.0", &dw));  // depthwise, weight-normed [C,1,5]
      const int K = (int)dw.shape[2];
      std::vector<float> wt((size_t)K * P.C);
      for (int ch = 0; ch < P.C; ++ch)
        for (int k = 0; k < K; ++k) wt[(size_t)k * P.C + ch] = dw.data[(size_t)ch * K + k];
      STTS_CHECK(K == 5, "cross_post.0 kernel");
      STTS_TRY(dev_upload(c, wt, &P.cp_dw_wt));
      STTS_TRY(upload_vec(c, p + "cross_post.0.bias", &P.cp_dw_b));
    }
    STTS_TRY(pack_plain(c, p + "cross_post.2", true, 0, P.C, &P.cp2));
    for (int i = 0; i < 3; ++i) {
      STTS_TRY(pack_adain_block(c, p + "F0." + std::to_string(i), P.C, P.C, &P.table, &P.f0[i]));
      STTS_TRY(pack_adain_block(c, p + "N." + std::to_string(i), P.C, P.C, &P.table, &P.n[i]));
    }
    STTS_TRY(upload_table(c, &P.table));
    STTS_TRY(upload_vec(c, p + "F0_proj.weight", &P.f0_w));
    STTS_TRY(upload_vec(c, p + "N_proj.weight", &P.n_w));
    STTS_GET(fb, p + "F0_proj.bias");
    STTS_GET(nb, p + "N_proj.bias");
    P.f0_b = fb->data[0];
    P.n_b = nb->data[0];
    P.ready = true;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ helpers
inline int gemm_store(hipStream_t st, const Seg& s, const float* X, int ldx, int xcol0, const PackedConv& w, float* Y, int ldy, int ycol0,
                      int act = ACT_NONE, const float* R = nullptr, int ldr = 0, float alpha = 1.0f) {
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, X, ldx, xcol0, w);
  a.N = w.N; a.bias = w.bias; a.Y = Y; a.ldy = ldy; a.ycol0 = ycol0; a.act = act; a.R = R; a.ldr = ldr; a.alpha = alpha;
  return launch_conv_gemm(st, a, EPI_STORE, w.npad, s.n_utt, s.max_len());
}

inline int static_ln(hipStream_t st, const float* X, int ldx, int C, long rows, float eps, const float* g, const float* b, float* Y, int ldy, int act) {
  LnOut o0{Y, ldy, 0, g, b, 0, 0}, o1{};
  return ln_launch(st, X, ldx, C, rows, nullptr, eps, 0, 1, o0, o1, act);
}
inline int adaptive_ln(hipStream_t st, const float* X, int ldx, int C, long rows, const int* row_utt, float eps, const float* sty, int ld_sty,
                       int gcol0, float* Y, int ldy, const int* rows_dev = nullptr) {
  LnOut o0{Y, ldy, 0, sty, nullptr, ld_sty, gcol0}, o1{};
  return ln_launch(st, X, ldx, C, rows, row_utt, eps, 1, 1, o0, o1, ACT_NONE, LnIn{}, rows_dev);
}

// Conv/Linear (+ residual) followed by a LayerNorm over the channels.  When the launcher cut K over several blocks the
// LayerNorm kernel finishes the partial sums itself (one launch less); otherwise it reads the contraction's output t.
inline int gemm_ln(hipStream_t st, const Seg& s, const float* X, int ldx, const PackedConv& w, float* t, int ldt, const float* R, int ldr, int C,
                   float eps, int adaptive, const int* row_utt, const LnOut& o0, int act_ln) {
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, X, ldx, 0, w);
  a.N = w.N; a.bias = w.bias; a.Y = t; a.ldy = ldt; a.R = R; a.ldr = ldr;
  SplitInfo info{};
  a.defer = &info;
  STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, w.npad, s.n_utt, s.max_len()));
  LnIn in{};
  if (info.ksplit > 1) in = LnIn{info.partial, info.ksplit, info.slice_rows, info.ld_part, a.bias, a.act, a.R, a.ldr, a.rcol0, a.alpha};
  LnOut o1{};
  return ln_launch(st, t, ldt, C, s.rows(), row_utt, eps, adaptive, 1, o0, o1, act_ln, in);
}

inline int run_attention(hipStream_t st, const Seg& sq, const Seg& sk, const float* Q, int ldq, int qcol0, const float* K, int ldk, int kcol0,
                         const float* V, int ldv, int vcol0, float* O, int ldo, int heads, int kc, const int* band_centre, int window, int force_kernel = 0) {
  STTS_CHECK(kc <= kAttnMaxKc && kc % 4 == 0, "attention: head size %d unsupported", kc);
  // heads of 16 / 32 / 40 / 64 / 96 / 128 / 160 channels: the matrix-core kernel, whatever the lengths (any number of keys) - the choice must not depend on the
  // batch, so that an utterance's result is the same alone and packed with others; other head sizes: one wave per 4 queries
  static const int mfma_min = getenv("STTS_ATTN_MFMA_MIN") ? atoi(getenv("STTS_ATTN_MFMA_MIN")) : 0;  // (experiments: shortest sequence for the matrix-core kernel)
  const bool aligned = ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && qcol0 % 4 == 0 && kcol0 % 4 == 0 && vcol0 % 4 == 0;
  STTS_CHECK((force_kernel != 2 && force_kernel != 3) || (attn_mfma_kc(kc) && aligned), "attention: the matrix-core kernel needs heads of 16 / 32 / 40 / 64 / 96 / 128 / 160 channels and 16-byte aligned rows");
  if (force_kernel == 2 || force_kernel == 3 || (force_kernel == 0 && attn_mfma_kc(kc) && aligned && sq.max_len() >= mfma_min && sk.max_len() >= mfma_min)) {
    const dim3 grid(ceil_div(sq.max_len(), kAttnMfmaQ), heads, sq.n_utt);
    const float scale = 1.0f / sqrtf((float)kc);
    // long sequences with heads of 64 (the flow-matching decoder): the keys split over two wave groups per query tile (the result depends on the split,
    // at fp32 rounding level: chosen by the LONGEST key sequence of the call, so an utterance alone and in a batch of shorter ones agree)
    static const int split_min = getenv("STTS_ATTN_SPLIT_MIN") ? atoi(getenv("STTS_ATTN_SPLIT_MIN")) : 128;
    if (kc == 64 && force_kernel != 3 && sk.max_len() >= split_min) {
      hipLaunchKernelGGL((attention_mfma_kernel<64, 2>), grid, dim3(512), 0, st, Q, ldq, qcol0, K, ldk, kcol0, V, ldv, vcol0, O, ldo, sq.dev, sk.dev, band_centre, window, scale);
      STTS_HIP(hipGetLastError());
      return 0;
    }
#define STTS_ATTN(KCV) hipLaunchKernelGGL(attention_mfma_kernel<KCV>, grid, dim3(256), 0, st, Q, ldq, qcol0, K, ldk, kcol0, V, ldv, vcol0, O, ldo, sq.dev, sk.dev, band_centre, window, scale)
    if (kc == 16) STTS_ATTN(16);
    else if (kc == 32) STTS_ATTN(32);
    else if (kc == 40) STTS_ATTN(40);
    else if (kc == 64) STTS_ATTN(64);
    else if (kc == 96) STTS_ATTN(96);
    else if (kc == 128) STTS_ATTN(128);
    else STTS_ATTN(160);
#undef STTS_ATTN
    STTS_HIP(hipGetLastError());
    return 0;
  }
  STTS_CHECK(sk.max_len() <= kAttnMaxKeys, "attention: more than %d keys (%d)", kAttnMaxKeys, sk.max_len());
  hipLaunchKernelGGL(attention_kernel, dim3(ceil_div(sq.max_len(), 4 * kAttnQ), heads, sq.n_utt), dim3(256), 0, st, Q, ldq, qcol0, K, ldk, kcol0, V, ldv, vcol0,
                     O, ldo, heads, kc, sq.dev, sk.dev, band_centre, window, 1.0f / sqrtf((float)kc));
  STTS_HIP(hipGetLastError());
  return 0;
}
inline void run_rope(hipStream_t st, const Seg& s, float* X, int ldx, int col0, int heads, int kc, int col1 = -1) {
  const int d = (int)(kc * 0.5);
  hipLaunchKernelGGL(rope_kernel, dim3(std::max(1, ceil_div(s.max_len() * heads * (d / 2) * (col1 >= 0 ? 2 : 1), 256)), s.n_utt), dim3(256), 0, st, X,
                     ldx, col0, col1, heads, kc, d, s.dev);
}

// ------------------------------------------------------------------------------------------------ TextEncoder.forward
// tokens [rows] int64 -> mu [rows, inter] (proj_m output), optional x [rows, C] (last hidden layer).
inline int text_encoder_forward(stts_ctx* c, hipStream_t st, const TextEncW& W, const Seg& s, const long* tokens, float* mu, int ld_mu, float* x_out,
                                Arena& ws) {
  const long R = s.rows();
  const int C = W.C;
  float* x = ws.get<float>(R * C);
  float* h = ws.get<float>(R * C);
  float* t = ws.get<float>(R * C);
  float* qkv = ws.get<float>(R * 3 * C);
  float* att = ws.get<float>(R * C);
  float* ff = ws.get<float>(R * W.filter);
  STTS_CHECK(ws.ok, "text_encoder_forward: workspace too small");
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)std::min<long>(1024, ceil_div((int)(R * C / 4), 256))), dim3(256), 0, st, tokens, W.emb, C, W.tokens,
                     sqrtf((float)C), x, C, (int)R, c->d_err);
  // ConvReluNorm prenet (text_encoder.py:79-86): 3 x (conv k5 -> channel LayerNorm eps 1e-4 -> ReLU), + 1x1 proj residual
  const float* cur = x;
  for (int i = 0; i < 3; ++i) {
    STTS_TRY(gemm_ln(st, s, cur, C, W.pre[i], t, C, nullptr, 0, C, 1e-4f, 0, nullptr, LnOut{h, C, 0, W.pre_g[i], W.pre_b[i], 0, 0}, ACT_RELU));
    cur = h;
  }
  STTS_TRY(gemm_store(st, s, h, C, 0, W.pre_proj, t, C, 0, ACT_NONE, x, C));
  std::swap(x, t);  // x = x_org + proj(h)
  const int kc = C / W.heads;
  for (int i = 0; i < W.n_layers; ++i) {
    const TextEncW::Layer& L = W.layer[i];
    STTS_TRY(gemm_store(st, s, x, C, 0, L.qkv, qkv, 3 * C, 0));
    run_rope(st, s, qkv, 3 * C, 0, W.heads, kc, C);  // q and k blocks of the fused q|k|v buffer
    STTS_TRY(run_attention(st, s, s, qkv, 3 * C, 0, qkv, 3 * C, C, qkv, 3 * C, 2 * C, att, C, W.heads, kc, nullptr, 0));
    STTS_TRY(gemm_ln(st, s, att, C, L.o, t, C, x, C, C, 1e-4f, 0, nullptr, LnOut{x, C, 0, L.g1, L.b1, 0, 0}, ACT_NONE));
    STTS_TRY(gemm_store(st, s, x, C, 0, L.f1, ff, W.filter, 0, ACT_RELU));
    STTS_TRY(gemm_ln(st, s, ff, W.filter, L.f2, t, C, x, C, C, 1e-4f, 0, nullptr, LnOut{x, C, 0, L.g2, L.b2, 0, 0}, ACT_NONE));
  }
  if (x_out) STTS_HIP(hipMemcpyAsync(x_out, x, R * C * sizeof(float), hipMemcpyDeviceToDevice, st));
  STTS_TRY(gemm_store(st, s, x, C, 0, W.proj_m, mu, ld_mu, 0));
  return 0;
}

// ------------------------------------------------------------------------------------------------ TextStyleEncoder.forward
// x [rows, >= inter] -> style [n_utt, 64].  Statistics run per utterance over its own tokens (the reference at B = 1).
inline int text_style_forward(stts_ctx* c, hipStream_t st, const StyleEncW& W, const Seg& s, const float* x, int ldx, float* style, int ld_style,
                              Arena& ws) {
  (void)c;
  const long R = s.rows();
  const int sd = W.sd, inter4 = 4 * sd, ml = s.max_len();
  float* h = ws.get<float>(R * sd);
  float* h2 = ws.get<float>(R * sd);
  float* dw = ws.get<float>(R * sd);
  float* nrm = ws.get<float>(R * sd);
  float* U = ws.get<float>(R * inter4);
  const int ss_stride = ceil_div(ml, 128) * 4;
  float* part = ws.get<float>((size_t)s.n_utt * ss_stride * inter4);
  float* gx = ws.get<float>((size_t)s.n_utt * inter4);
  float* w2u = ws.get<float>((size_t)s.n_utt * W.blk[0].pw2.npad * inter4);
  STTS_CHECK(ws.ok, "text_style_forward: workspace too small");
  STTS_TRY(gemm_store(st, s, x, ldx, 0, W.conv_in, h, sd, 0));
  for (int i = 0; i < W.n; ++i) {
    const StyleEncW::Blk& B = W.blk[i];
    hipLaunchKernelGGL((dwconv_kernel<31>), dim3(ceil_div(sd, 64), ceil_div(ml, 64), s.n_utt), dim3(256), 0, st, h, sd, dw, sd, sd, s.dev, B.dw_wt, B.dw_b,
                       7, (int)ACT_NONE);
    STTS_TRY(static_ln(st, dw, sd, sd, R, 1e-6f, B.ln_g, B.ln_b, nrm, sd, ACT_NONE));
    GemmArgs a = gemm_args(s);
    set_seg(a, 0, nrm, sd, 0, B.pw1);
    a.N = inter4; a.bias = B.pw1.bias; a.Y = U; a.ldy = inter4; a.act = ACT_GELU;
    a.sumsq_part = part; a.ld_ss = inter4; a.ss_stride = ss_stride;
    STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, B.pw1.npad, s.n_utt, ml));
    hipLaunchKernelGGL(grn_gx_kernel, dim3(ceil_div(inter4, 32), s.n_utt), dim3(256), 0, st, part, inter4, ss_stride, s.dev, inter4, gx, inter4);
    launch_scale_weight(st, dim3(8, s.n_utt), B.pw2.prec, B.pw2.W, gx, inter4, B.grn_gamma, w2u, B.pw2.npad, B.pw2.kc);
    GemmArgs b = gemm_args(s);
    set_seg(b, 0, U, inter4, 0, B.pw2);
    b.seg[0].W = w2u;
    b.seg[0].W16 = reinterpret_cast<const unsigned short*>(w2u);
    b.seg[0].w16_plane = 0;  // (per-utterance fp32 copies: no split planes, this contraction stays on the f32 matrix cores)
    b.seg[0].w_utt_stride = (long)B.pw2.npad * B.pw2.kc;
    b.N = sd; b.bias = B.pw2.bias; b.Y = h2; b.ldy = sd; b.R = h; b.ldr = sd;
    STTS_TRY(launch_conv_gemm(st, b, EPI_STORE, B.pw2.npad, s.n_utt, ml));
    std::swap(h, h2);
  }
  hipLaunchKernelGGL(mean_rows_kernel, dim3(s.n_utt), dim3(256), 0, st, h, sd, sd, s.dev, style, ld_style);
  STTS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ ProsodyEncoder.forward
// enc [rows, >= d], style [n_utt, 64], sty = style table output of the owning module -> out [rows, C = d + 64]
inline int prosody_forward(hipStream_t st, const ProsodyW& W, const Seg& s, const float* enc, int ld_enc, const float* style, const float* sty,
                           int ld_sty, const int* row_utt, float* out, Arena& ws) {
  const long R = s.rows();
  const int C = W.C, d = W.d, sd = C - d, kc = C / W.heads;
  float* x = out;  // [rows, C] : cols [0,d) features, [d,C) style
  float* t = ws.get<float>(R * C);
  float* y = ws.get<float>(R * C);
  float* qkv = ws.get<float>(R * 3 * C);
  float* att = ws.get<float>(R * C);
  float* ff = ws.get<float>(R * 2 * C);
  STTS_CHECK(ws.ok, "prosody_forward: workspace too small");
  STTS_HIP(hipMemcpy2DAsync(x, C * sizeof(float), enc, ld_enc * sizeof(float), d * sizeof(float), R, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(broadcast_style_kernel, dim3(std::max(1, ceil_div(s.max_len() * sd, 256)), s.n_utt), dim3(256), 0, st, style, sd, sd, x, C, d, s.dev);
  for (int i = 0; i < W.n_layers; ++i) {
    const ProsodyW::L& L = W.l[i];
    STTS_TRY(gemm_store(st, s, x, C, 0, L.qkv, qkv, 3 * C, 0));
    run_rope(st, s, qkv, 3 * C, 0, W.heads, kc, C);  // q and k blocks of the fused q|k|v buffer
    STTS_TRY(run_attention(st, s, s, qkv, 3 * C, 0, qkv, 3 * C, C, qkv, 3 * C, 2 * C, att, C, W.heads, kc, nullptr, 0));
    STTS_TRY(gemm_ln(st, s, att, C, L.o, t, C, x, C, C, 1e-5f, 1, row_utt, LnOut{y, C, 0, sty, nullptr, ld_sty, L.n1.col0}, ACT_NONE));
    STTS_TRY(gemm_store(st, s, y, C, 0, L.f1, ff, 2 * C, 0, ACT_RELU));
    STTS_TRY(gemm_ln(st, s, ff, 2 * C, L.f2, t, C, y, C, C, 1e-5f, 1, row_utt, LnOut{y, C, 0, sty, nullptr, ld_sty, L.n2.col0}, ACT_NONE));
    STTS_TRY(gemm_store(st, s, y, C, 0, L.proj, x, C, 0));  // columns [0,d); the style columns of x persist
  }
  return 0;
}

inline size_t phoneme_workspace_bytes(const stts_ctx* c, int64_t n_tok, int64_t n_frames, int n_utt) {
  const stts_model_dims& d = c->d;
  const size_t C = d.te_hidden, Cp = d.pe_inter + d.style_dim;
  const size_t tok = (size_t)n_tok * (C * 8 + d.te_filter + Cp * 12 + d.style_dim * 12 + 64) * sizeof(float);
  const size_t frm = (size_t)n_frames * (Cp * 12 + 16) * sizeof(float) + ((size_t)n_frames / kWinoM + n_utt + 1) * 8 * 2 * Cp * sizeof(float);  // + Winograd scratch
  const size_t per = (size_t)n_utt * ((size_t)(n_tok / std::max(1, n_utt) / 32 + 16) * 4 * d.style_dim * 8 + 128 * 256 + 8192) * sizeof(float);
  return tok + frm + per + ((size_t)4 << 20);
}

// ------------------------------------------------------------------------------------------------ DurationPredictor.forward
inline int duration_forward(stts_ctx* c, PhonemeModel& M, hipStream_t st, const Seg& s, const long* tokens, float* logits, int ld_logits, int* dur_out,
                            float* mu_out, float* style_out, float* prosody_out, Arena& ws) {
  const long R = s.rows();
  const DurationW& D = M.dur;
  const int d = c->d.inter_dim, C = D.pros.C;
  float* mu = mu_out ? mu_out : ws.get<float>(R * d);
  float* style = style_out ? style_out : ws.get<float>((size_t)s.n_utt * c->d.style_dim);
  float* pros = prosody_out ? prosody_out : ws.get<float>(R * C);
  float* sty = ws.get<float>((size_t)s.n_utt * D.table.ld());
  int* row_utt = ws.get<int>(R);
  STTS_CHECK(ws.ok, "duration_forward: workspace too small");
  hipLaunchKernelGGL(row_utt_kernel, dim3(ceil_div(s.max_len(), 256), s.n_utt), dim3(256), 0, st, s.dev, s.n_utt, row_utt);
  { Arena a(ws.base + ws.used, ws.cap - ws.used); STTS_TRY(text_encoder_forward(c, st, M.te[0], s, tokens, mu, d, nullptr, a)); }
  { Arena a(ws.base + ws.used, ws.cap - ws.used); STTS_TRY(text_style_forward(c, st, M.se[0], s, mu, d, style, c->d.style_dim, a)); }
  STTS_TRY(run_style(st, D.table, style, s.n_utt, sty));
  { Arena a(ws.base + ws.used, ws.cap - ws.used); STTS_TRY(prosody_forward(st, D.pros, s, mu, d, style, sty, D.table.ld(), row_utt, pros, a)); }
  STTS_TRY(gemm_store(st, s, pros, C, 0, D.proj, logits, ld_logits, 0));
  if (dur_out) hipLaunchKernelGGL(duration_decode_kernel, dim3(ceil_div((int)R, 256)), dim3(256), 0, st, logits, ld_logits, 16, (int)R, dur_out);
  STTS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ PitchEnergyPredictor.forward
// sp: tokens, sf: mel-rate frames.  dur [n_tok] int32 (device).  pe_enc [n_tok, >= 256], pe_style [n_utt, 64] -> f0, n [n_frames]
inline int pitch_energy_forward(stts_ctx* c, PhonemeModel& M, hipStream_t st, const Seg& sp, const Seg& sf, const int* dur, const float* pe_enc, int ld_enc,
                                const float* pe_style, float* f0, float* nrg, float* prosody_out, float* cross_out, Arena& ws) {
  const PitchEnergyW& P = M.pe;
  const long Rp = sp.rows(), Rf = sf.rows();
  const int C = P.C, heads = 8, kc = C / heads;
  float* pros = prosody_out ? prosody_out : ws.get<float>(Rp * C);
  float* sty = ws.get<float>((size_t)sp.n_utt * P.table.ld());
  int* row_utt_p = ws.get<int>(Rp);
  int* row_utt_f = ws.get<int>(Rf);
  int* src_row = ws.get<int>(Rf);
  int* centre = ws.get<int>(Rf);
  float* base = ws.get<float>(Rf * C);
  float* qn = ws.get<float>(Rf * C);
  float* kn = ws.get<float>(Rp * C);
  float* q = ws.get<float>(Rf * C);
  float* kv = ws.get<float>(Rp * 2 * C);
  float* att = ws.get<float>(Rf * C);
  float* t1 = ws.get<float>(Rf * C);
  float* t2 = ws.get<float>(Rf * C);
  float* x = cross_out ? cross_out : ws.get<float>(Rf * C);
  float* act1 = ws.get<float>(Rf * C);
  float* hb = ws.get<float>(Rf * C);
  float* act2 = ws.get<float>(Rf * C);
  float* ss = ws.get<float>(adain_part_floats(sf, C));
  WinoScratch wino;  // large batches: Winograd convs
  if (Rf > fold_rows() && P.f0[0].w1.ready) wino.p = ws.get<float>(wino_scratch_floats(sf, P.f0[0].w1));
  // ... whose output transforms leave the AdaIN statistics of what they write (decoder_forward's scheme: AdainStats with 24-row chunks)
  static const bool no_wino_stats = getenv("STTS_NO_WINO_STATS") != nullptr;
  const bool wstat_mode = wino && !no_wino_stats;
  float* ss_in = wstat_mode ? ws.get<float>(adain_part_floats(sf, C, kWinoStatChunk)) : nullptr;
  float* ss_mid = wstat_mode ? ws.get<float>(adain_part_floats(sf, C, kWinoStatChunk)) : nullptr;
  float* ss_x = wstat_mode ? ws.get<float>(adain_part_floats(sf, C, kWinoStatChunk)) : nullptr;  // statistics of x: one pass for both branches' first norm
  STTS_CHECK(ws.ok, "pitch_energy_forward: workspace too small");
  hipLaunchKernelGGL(row_utt_kernel, dim3(ceil_div(sp.max_len(), 256), sp.n_utt), dim3(256), 0, st, sp.dev, sp.n_utt, row_utt_p);
  hipLaunchKernelGGL(row_utt_kernel, dim3(ceil_div(sf.max_len(), 256), sf.n_utt), dim3(256), 0, st, sf.dev, sf.n_utt, row_utt_f);
  STTS_TRY(run_style(st, P.table, pe_style, sp.n_utt, sty));
  const int lds = P.table.ld();
  { Arena a(ws.base + ws.used, ws.cap - ws.used); STTS_TRY(prosody_forward(st, P.pros, sp, pe_enc, ld_enc, pe_style, sty, lds, row_utt_p, pros, a)); }
  // compute_cross (pitch_energy_predictor.py:83-102): base = prosody^T @ alignment == gather by the frame->token map
  hipLaunchKernelGGL(frame_token_map_kernel, dim3(sp.n_utt), dim3(256), 0, st, dur, sp.dev, sf.dev, 1, src_row);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<long>(2048, std::max<long>(1, ceil_div((int)(Rf * C / 4), 256)))), dim3(256), 0, st, pros, C,
                     src_row, base, C, 0, C, (int)Rf, sf.rows_dev());
  hipLaunchKernelGGL(local_token_kernel, dim3(ceil_div(sf.max_len(), 256), sf.n_utt), dim3(256), 0, st, src_row, sf.dev, sp.dev, centre);
  STTS_TRY(adaptive_ln(st, base, C, C, Rf, row_utt_f, 1e-5f, sty, lds, P.qn.col0, qn, C, sf.rows_dev()));
  STTS_TRY(adaptive_ln(st, pros, C, C, Rp, row_utt_p, 1e-5f, sty, lds, P.kn.col0, kn, C));
  STTS_TRY(gemm_store(st, sf, qn, C, 0, P.q, q, C, 0));
  STTS_TRY(gemm_store(st, sp, kn, C, 0, P.kv, kv, 2 * C, 0));
  run_rope(st, sf, q, C, 0, heads, kc);
  run_rope(st, sp, kv, 2 * C, 0, heads, kc);
  STTS_TRY(run_attention(st, sf, sp, q, C, 0, kv, 2 * C, 0, kv, 2 * C, C, att, C, heads, kc, centre, 5));
  STTS_TRY(gemm_store(st, sf, att, C, 0, P.o, t1, C, 0));
  hipLaunchKernelGGL((dwconv_kernel<31>), dim3(ceil_div(C, 64), ceil_div(sf.max_len(), 64), sf.n_utt), dim3(256), 0, st, t1, C, t2, C, C, sf.dev, P.cp_dw_wt,
                     P.cp_dw_b, 5, (int)ACT_SILU);
  STTS_TRY(gemm_store(st, sf, t2, C, 0, P.cp2, x, C, 0, ACT_NONE, base, C, 0.70710678118654752440f));
  // F0 and N branches: 3 AdaptiveDecoderBlocks each + 1x1 projection to one channel
  for (int br = 0; br < 2; ++br) {
    const AdainBlockW* blocks = br == 0 ? P.f0 : P.n;
    const float* cur = x;
    float* bufs[2] = {t1, t2};
    AdainStats stats;
    stats.chunk_rows = kWinoStatChunk;
    stats.mid = ss_mid;
    const bool x_stats = wstat_mode && blocks[0].cin == C && C % 32 == 0;
    if (x_stats && br == 0) {
      const int nchunk = wino_stat_chunks(sf);
      STTS_LAUNCH_PROF("adain_partial_kernel", (size_t)Rf * C * 4, adain_partial_kernel, dim3(ceil_div(C, 32), nchunk, sf.n_utt), dim3(256), st, x, C, C, sf.dev, ss_x, C, nchunk, kWinoStatChunk);
    }
    for (int i = 0; i < 3; ++i) {
      // norm1 of blocks 1, 2: statistics left by the previous block's conv2; norm2: by this block's conv1 (both Winograd output transforms)
      stats.in = i == 0 ? ss_x : ss_in;
      stats.in_ready = i == 0 ? x_stats : (stats.out_ready && blocks[i].cin == blocks[i - 1].cout);
      stats.out = (i < 2 && blocks[i].cout <= C) ? ss_in : nullptr;
      stats.out_ld = round_up(blocks[i].cout, 32);
      STTS_TRY(run_adain_block(st, sf, blocks[i], sty, lds, cur, C, bufs[i & 1], C, act1, hb, act2, ss, 0, &wino, nullptr, false, nullptr, 0, wstat_mode ? &stats : nullptr));
      cur = bufs[i & 1];
    }
    const ChanConvSet cs{cur, br == 0 ? P.f0_w : P.n_w, br == 0 ? P.f0_b : P.n_b, br == 0 ? f0 : nrg};
    hipLaunchKernelGGL(single_channel_conv_kernel<0>, dim3((unsigned)ceil_div(sf.max_len(), 4 * kChanRows), 1, sf.n_utt), dim3(256), 0, st, cs, cs, C, C, sf.dev, 1,
                       1, 0);
  }
  STTS_HIP(hipGetLastError());
  return 0;
}

}  // namespace stts
