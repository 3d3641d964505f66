// extern "C" entry points declared in include/stylish_hip.h.  Single translation unit: hipcc compiles this file
// (which includes every kernel header) into libstylish_hip.so for gfx950.
#include "model.hip.h"
#include "cfm.hip.h"

using namespace stts;

#define API_BEGIN try {
#define API_END                                                         \
  }                                                                     \
  catch (const std::exception& e) { return stts::fail("exception: %s", e.what()); } \
  catch (...) { return stts::fail("unknown exception"); }

extern "C" {

const char* stts_last_error(void) { return stts::last_error().c_str(); }
int stts_version(void) { return 1; }

int stts_ctx_create(const stts_model_dims* dims, int device, stts_ctx** out) {
  API_BEGIN
  STTS_CHECK(dims && out, "null argument");
  int n = 0;
  STTS_HIP(hipGetDeviceCount(&n));
  STTS_CHECK(device >= 0 && device < n, "device %d not available (%d visible)", device, n);
  STTS_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  STTS_HIP(hipGetDeviceProperties(&prop, device));
  STTS_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  stts_ctx* c = new stts_ctx();
  c->d = *dims;
  c->device = device;
  void* p = nullptr;
  STTS_HIP(hipMalloc(&p, 256));
  STTS_HIP(hipMemset(p, 0, 256));
  c->d_err = (int*)p;
  c->allocs.push_back(p);
  c->alloc_tag.push_back(0);
  *out = c;
  return 0;
  API_END
}

void stts_ctx_destroy(stts_ctx* c) {
  if (!c) return;
  for (void* p : c->allocs) (void)hipFree(p);
  for (auto& kv : c->side_lanes) {
    if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
    if (kv.second.join) (void)hipEventDestroy(kv.second.join);
    if (kv.second.stream) (void)hipStreamDestroy(kv.second.stream);
    if (kv.second.fork2) (void)hipEventDestroy(kv.second.fork2);
    if (kv.second.join2) (void)hipEventDestroy(kv.second.join2);
    if (kv.second.stream2) (void)hipStreamDestroy(kv.second.stream2);
  }
  delete c;
}

int stts_set_precision(stts_ctx* c, int precision) {
  API_BEGIN
  STTS_CHECK(c, "bad argument");
  STTS_CHECK(precision >= 0 && precision <= 3, "precision must be STTS_PREC_F32, _BF16, _F16 or _F32_NATIVE");
  const int prec = precision == STTS_PREC_F32_NATIVE ? (int)PREC_F32 : precision;
  STTS_CHECK(c->ready == 0 || (prec == c->prec && (precision != STTS_PREC_F32_NATIVE) == c->allow_x3), "precision must be chosen before weights are finalized");
  c->prec = prec;
  c->allow_x3 = precision != STTS_PREC_F32_NATIVE;
  return 0;
  API_END
}

int stts_load_weight(stts_ctx* c, const char* name, const float* data, const int64_t* shape, int ndim) {
  API_BEGIN
  STTS_CHECK(c && name && data && shape && ndim >= 1 && ndim <= 4, "bad argument");
  HostTensor t;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    t.shape.push_back(shape[i]);
    n *= shape[i];
  }
  t.data.assign(data, data + n);
  c->host[name] = std::move(t);
  return 0;
  API_END
}

// every exit path of a finalize leaves the allocation tag at "context lifetime"
struct TagReset {
  stts_ctx* c;
  ~TagReset() { c->cur_tag = 0; }
};

int stts_finalize_weights(stts_ctx* c, int which) {
  API_BEGIN
  STTS_CHECK(c, "null ctx");
  STTS_HIP(hipSetDevice(c->device));
  TagReset tag_reset{c};
  // re-finalizing: the previous packing of these components goes away, and so do their `ready` bits - they come back only for
  // the components that finalize successfully below (a failed re-finalize must not leave a stage runnable on freed buffers)
  c->ready &= ~which;
  free_component_allocs(c, which);
  if (which & (STTS_W_DECODER | STTS_W_FLOW | STTS_W_GENERATOR)) STTS_TRY(finalize_frame(c, which));
  const int ph = which & (STTS_W_SPEECH_TEXT | STTS_W_DURATION | STTS_W_PE_TEXT | STTS_W_PE_STYLE | STTS_W_PITCH_ENERGY);
  if (ph) {
    if (!c->phoneme) c->phoneme = std::make_shared<PhonemeModel>();
    // the phoneme-rate predictors always run in fp32 (include/stylish_hip.h, stts_set_precision): durations are integers and must
    // equal the fp32 reference's bit for bit, and these stages are latency-bound (nothing to win from 16-bit operands)
    // (and on the f32 matrix cores, not the split-fp32 form: latency-bound launches, and per-utterance GRN weights in the style encoder)
    const int saved_prec = c->prec;
    c->prec = PREC_F32;
    static const bool phoneme_x3 = getenv("STTS_PHONEME_X3") && atoi(getenv("STTS_PHONEME_X3")) != 0;  // experiment: the split form for the phoneme-rate contractions too
    c->pack_x3 = phoneme_x3;
    const int rc = finalize_phoneme(c, static_cast<PhonemeModel*>(c->phoneme.get()), ph);
    c->pack_x3 = true;
    c->prec = saved_prec;
    STTS_TRY(rc);
    c->ready |= ph;
  }
  STTS_HIP(hipDeviceSynchronize());
  return 0;
  API_END
}

int stts_check_status(stts_ctx* c, void* stream) {
  API_BEGIN
  int e = 0;
  STTS_HIP(hipStreamSynchronize((hipStream_t)stream));
  STTS_HIP(hipMemcpy(&e, c->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (e) {
    // bit flags (the kernels set them with atomicOr): every pending condition is reported, the most specific text first
    STTS_HIP(hipMemset(c->d_err, 0, sizeof(int)));
    std::string msg;
    if (e & 2) msg += "text encoder: token id outside [0, tokens); ";
    if (e & 4) msg += "harmonic source: an utterance is too short for the STFT's reflect padding (needs more than " + std::to_string(kNfft / 2) + " samples); ";
    if (e & 1) msg += "harmonic source: a frame is voiced (f0 > 10 Hz) but no f0 exceeds 20 Hz (reference raises: models/generator.py:285); ";
    if (e & ~7) msg += "unknown device error bits " + std::to_string(e & ~7) + "; ";
    msg.resize(msg.size() - 2);
    return stts::fail("%s", msg.c_str());
  }
  return 0;
  API_END
}

int stts_har_ld(const stts_ctx* c) {
  if (!c) return 0;
  if (c->amp_prior.kc) return har_ld(c);  // the generator is packed: the prior convs' input width
  return round_up(kBins, c->prec != PREC_F32 ? 64 : 32);  // before: what finalize_frame will pad the bins to in this precision (kc_align)
}

size_t stts_frame_workspace_bytes(const stts_ctx* c, int64_t rows, int n_utt, int max_len) { return frame_workspace_bytes(c, rows, n_utt, max_len); }

#define SEG_CHECK(mask)                                                              \
  STTS_CHECK(c && (c->ready & (mask)) == (mask), "weights for this stage are not finalized (need components 0x%x, have 0x%x)", (mask), c ? c->ready : 0); \
  STTS_CHECK(n_utt > 0 && seg_off_host && seg_off_dev && seg_off_host[0] == 0, "bad utterance offsets"); \
  for (int _u = 0; _u < n_utt; ++_u) STTS_CHECK(seg_off_host[_u + 1] > seg_off_host[_u], "utterance %d is empty", _u); \
  STTS_HIP(hipSetDevice(c->device));                                                 \
  Seg s{n_utt, seg_off_host, seg_off_dev};                                           \
  hipStream_t st = (hipStream_t)stream

int stts_decoder_forward(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* asr,
                         int ld_asr, const float* pitch, const float* energy, const float* style, float* x_out, int ld_x, void* ws,
                         size_t ws_bytes) {
  API_BEGIN
  SEG_CHECK(STTS_W_DECODER);
  STTS_CHECK(ld_asr >= c->d.inter_dim && ld_asr % 4 == 0 && ld_x >= c->d.dec_hidden, "bad leading dimension");
  Arena a(ws, ws_bytes);
  return decoder_forward(c, st, s, asr, ld_asr, pitch, energy, style, x_out, ld_x, a);
  API_END
}

int stts_prior_flow_forward(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* x,
                            int ld_x, const float* style, const float* prior_noise, float* mel_out, int ld_mel, float* z_prior_out,
                            float* z_flow_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  SEG_CHECK(STTS_W_FLOW);
  STTS_CHECK(ld_x % 4 == 0 && ld_x >= c->d.dec_hidden && ld_mel >= c->d.dec_hidden, "bad leading dimension");
  Arena a(ws, ws_bytes);
  return prior_flow_forward(c, st, s, x, ld_x, style, prior_noise, mel_out, ld_mel, z_prior_out, z_flow_out, a);
  API_END
}

int stts_harmonic_stft(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* pitch,
                       const float* src_noise, const float* init_phase, int batch_scope, float* prior_signal_out, float* har_spec,
                       float* har_phase, int ld_har, void* ws, size_t ws_bytes) {
  API_BEGIN
  SEG_CHECK(STTS_W_GENERATOR);
  STTS_CHECK(ld_har >= kBins, "ld_har %d < %d", ld_har, kBins);
  Arena a(ws, ws_bytes);
  return harmonic_stft(c, st, s, pitch, src_noise, init_phase, batch_scope, prior_signal_out, har_spec, har_phase, ld_har, a);
  API_END
}

int stts_vocoder_forward(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* mel,
                         int ld_mel, const float* style, const float* har_spec, const float* har_phase, int ld_har, float* audio_out,
                         float* logamp_out, float* phase_out, int ld_lp, void* ws, size_t ws_bytes) {
  API_BEGIN
  SEG_CHECK(STTS_W_GENERATOR);
  STTS_CHECK(ld_mel % 4 == 0 && ld_har % 32 == 0 && ld_har >= har_ld(c), "har/mel leading dimension: ld_har must be a multiple of 32 covering %d columns (the prior convs' packed input width)", har_ld(c));
  STTS_CHECK(!logamp_out || ld_lp >= kBins, "ld_lp too small");
  Arena a(ws, ws_bytes);
  return vocoder_forward(c, st, s, mel, ld_mel, style, har_spec, har_phase, ld_har, audio_out, logamp_out, phase_out, ld_lp, a);
  API_END
}

int stts_frame_path(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* asr, int ld_asr,
                    const float* pitch, const float* energy, const float* style, const float* prior_noise, const float* src_noise,
                    const float* init_phase, int batch_scope, float* audio_out, void* ws, size_t ws_bytes, int seg_flags) {
  API_BEGIN
  SEG_CHECK(STTS_W_DECODER | STTS_W_FLOW | STTS_W_GENERATOR);
  s.cap = (seg_flags & STTS_SEG_CAPACITY) != 0;
  STTS_CHECK(ld_asr >= c->d.inter_dim && ld_asr % 4 == 0, "bad ld_asr");
  return frame_path(c, st, s, asr, ld_asr, pitch, energy, style, prior_noise, src_noise, init_phase, batch_scope, audio_out, ws, ws_bytes);
  API_END
}

int stts_length_regulate(stts_ctx* c, void* stream, int n_utt, const int32_t* dur, const int32_t* tok_off, const int32_t* frm_off,
                         int64_t n_frames, int rep, const float* enc, int ld_enc, int C, float* out, int ld_out, int32_t* src_row_ws) {
  API_BEGIN
  if (c) STTS_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  STTS_CHECK(dur && tok_off && frm_off && enc && out && src_row_ws && n_utt > 0 && n_frames >= 0 && rep >= 1, "length_regulate: bad argument");
  STTS_CHECK(C % 4 == 0 && ld_enc % 4 == 0 && ld_out % 4 == 0, "length_regulate: channel counts must be multiples of 4");
  hipLaunchKernelGGL(frame_token_map_kernel, dim3(n_utt), dim3(256), 0, st, dur, tok_off, frm_off, rep, src_row_ws);
  const long work = n_frames * (C / 4);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<long>(2048, std::max<long>(1, (work + 255) / 256))), dim3(256), 0, st, enc, ld_enc,
                     src_row_ws, out, ld_out, 0, C, (int)n_frames, frm_off + n_utt);  // n_frames may be a capacity: the real count is frm_off[n_utt]
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

int stts_frame_offsets(stts_ctx* c, void* stream, int n_utt, const int32_t* tok_off_dev, const int32_t* dur, const int32_t* cap_off_dev,
                       int32_t* off_T_dev, int32_t* off_T4_dev, int32_t* need_dev) {
  API_BEGIN
  STTS_CHECK(c && n_utt > 0 && tok_off_dev && dur && cap_off_dev && off_T_dev && off_T4_dev && need_dev, "frame_offsets: bad argument");
  STTS_HIP(hipSetDevice(c->device));
  hipLaunchKernelGGL(frame_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dur, tok_off_dev, n_utt, cap_off_dev, off_T_dev, off_T4_dev, need_dev);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

int stts_upsample4(stts_ctx* c, void* stream, int n_utt, const int32_t* off_T_host, const int32_t* off_T, const int32_t* off_T4, const float* x,
                   float* y) {
  API_BEGIN
  if (c) STTS_HIP(hipSetDevice(c->device));
  int ml = 0;
  for (int u = 0; u < n_utt; ++u) ml = std::max(ml, off_T_host[u + 1] - off_T_host[u]);
  hipLaunchKernelGGL(upsample4_kernel, dim3(ceil_div(4 * ml, 256), n_utt), dim3(256), 0, (hipStream_t)stream, x, off_T, off_T4, y);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

int stts_euler_step(void* stream, float* x, const float* v, float dt, int64_t n) {
  API_BEGIN
  STTS_CHECK(x && v && n >= 0, "bad argument");
  if (n) hipLaunchKernelGGL(euler_step_kernel, dim3((unsigned)std::min<int64_t>(4096, (n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, v, dt, (long)n);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

// ------------------------------------------------------------------------------------------------ CfmMelDecoder estimator (cfm.hip.h)
int stts_cfm_finalize(stts_ctx* c, const stts_cfm_dims* dims) {
  API_BEGIN
  STTS_CHECK(c && dims, "null argument");
  STTS_HIP(hipSetDevice(c->device));
  TagReset tag_reset{c};
  c->ready &= ~STTS_W_CFM;
  free_component_allocs(c, STTS_W_CFM);
  CfmDims d;
  d.feat = dims->feat_dim; d.asr = dims->asr_dim; d.spk = dims->spk_dim; d.hidden = dims->hidden_dim; d.emb = dims->emb_dim; d.depth = dims->depth;
  d.enc_blocks = dims->enc_blocks; d.dec_blocks = dims->dec_blocks; d.prev_depth = dims->prev_depth; d.post_depth = dims->post_depth; d.head_dim = dims->head_dim;
  auto m = std::make_shared<CfmModel>();
  c->cur_tag = STTS_W_CFM;
  c->pack_x3 = false;  // latency-bound estimator: f32 matrix cores
  const int rc = finalize_cfm(c, d, m.get());
  c->pack_x3 = true;
  c->cur_tag = 0;
  if (rc) return rc;
  c->cfm = m;
  c->ready |= STTS_W_CFM;
  STTS_HIP(hipDeviceSynchronize());
  return 0;
  API_END
}

size_t stts_cfm_workspace_bytes(const stts_ctx* c, int64_t rows, int n_utt) {
  if (!c || !c->cfm) return 0;
  return cfm_workspace_bytes(const_cast<stts_ctx*>(c), *static_cast<const CfmModel*>(c->cfm.get()), rows, n_utt);
}

int stts_cfm_estimator(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* x, int ld_x,
                       const float* asr, int ld_asr, const float* f0, const float* n_curve, const int32_t* curve_off_host, const int32_t* curve_off_dev,
                       const float* spk_emb, const float* t, const float* sine_noise, float* out, int ld_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  STTS_CHECK(c && c->cfm && (c->ready & STTS_W_CFM), "the CfmMelDecoder weights are not finalized (stts_cfm_finalize)");
  STTS_CHECK(n_utt > 0 && seg_off_host && seg_off_dev && seg_off_host[0] == 0, "bad utterance offsets");
  STTS_CHECK(curve_off_host && curve_off_dev && curve_off_host[0] == 0, "bad curve offsets");
  for (int u = 0; u < n_utt; ++u) {
    STTS_CHECK(seg_off_host[u + 1] > seg_off_host[u], "utterance %d is empty", u);
    STTS_CHECK(curve_off_host[u + 1] > curve_off_host[u], "utterance %d has an empty F0 / N curve", u);
  }
  STTS_CHECK(x && asr && f0 && n_curve && spk_emb && t && sine_noise && out && ws, "null tensor");
  STTS_HIP(hipSetDevice(c->device));
  Seg s{n_utt, seg_off_host, seg_off_dev};
  Arena a(ws, ws_bytes);
  return cfm_estimator(c, *static_cast<const CfmModel*>(c->cfm.get()), (hipStream_t)stream, s, x, ld_x, asr, ld_asr, f0, n_curve, curve_off_dev, spk_emb, t,
                       sine_noise, out, ld_out, a);
  API_END
}

int stts_to_time_major(void* stream, const float* x, int B, int C, int T, float* y, int ldy) {
  API_BEGIN
  STTS_CHECK(ldy >= C, "ldy < C");
  hipLaunchKernelGGL(to_time_major_kernel, dim3(ceil_div(T, 32), ceil_div(ldy, 32), B), dim3(256), 0, (hipStream_t)stream, x, B, C, T, y, ldy, 0, ldy);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

int stts_to_channel_major(void* stream, const float* x, int ldx, int B, int C, int T, float* y) {
  API_BEGIN
  hipLaunchKernelGGL(to_channel_major_kernel, dim3(ceil_div(T, 32), ceil_div(C, 32), B), dim3(256), 0, (hipStream_t)stream, x, ldx, 0, B, C, T, y);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

// ------------------------------------------------------------------------------------------------ phoneme-rate stages
#define PH_CHECK(mask)                                                               \
  STTS_CHECK(c && c->phoneme && (c->ready & (mask)) == (mask), "weights for this stage are not finalized (need components 0x%x, have 0x%x)", (mask), c ? c->ready : 0); \
  STTS_HIP(hipSetDevice(c->device));                                                 \
  PhonemeModel& M = *static_cast<PhonemeModel*>(c->phoneme.get());                   \
  hipStream_t st = (hipStream_t)stream

static int seg_ok(int n_utt, const int32_t* h, const int32_t* d) {
  STTS_CHECK(n_utt > 0 && h && d && h[0] == 0, "bad utterance offsets");
  for (int u = 0; u < n_utt; ++u) STTS_CHECK(h[u + 1] > h[u], "utterance %d is empty", u);
  return 0;
}

size_t stts_phoneme_workspace_bytes(const stts_ctx* c, int64_t n_tokens, int64_t n_frames, int n_utt) {
  return phoneme_workspace_bytes(c, n_tokens, n_frames, n_utt);
}

int stts_text_encoder_forward(stts_ctx* c, void* stream, int which, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev,
                              const int64_t* tokens, float* mu_out, int ld_mu, float* x_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  const int te_mask[3] = {STTS_W_DURATION, STTS_W_SPEECH_TEXT, STTS_W_PE_TEXT};
  STTS_CHECK(which >= 0 && which < 3, "which must be 0 (duration), 1 (speech) or 2 (pitch/energy)");
  PH_CHECK(te_mask[which]);
  STTS_TRY(seg_ok(n_utt, tok_off_host, tok_off_dev));
  STTS_CHECK(ld_mu >= M.te[which].inter, "ld_mu too small");
  Seg s{n_utt, tok_off_host, tok_off_dev};
  Arena a(ws, ws_bytes);
  return text_encoder_forward(c, st, M.te[which], s, (const long*)tokens, mu_out, ld_mu, x_out, a);
  API_END
}

int stts_text_style_forward(stts_ctx* c, void* stream, int which, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev, const float* x,
                            int ldx, float* style_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  const int se_mask[3] = {STTS_W_DURATION, STTS_W_SPEECH_TEXT, STTS_W_PE_STYLE};
  STTS_CHECK(which >= 0 && which < 3, "which must be 0, 1 or 2");
  PH_CHECK(se_mask[which]);
  STTS_TRY(seg_ok(n_utt, tok_off_host, tok_off_dev));
  STTS_CHECK(ldx % 32 == 0 && ldx >= M.se[which].inter, "style encoder input: ld must be a multiple of 32 covering inter_dim");
  Seg s{n_utt, tok_off_host, tok_off_dev};
  Arena a(ws, ws_bytes);
  return text_style_forward(c, st, M.se[which], s, x, ldx, style_out, c->d.style_dim, a);
  API_END
}

int stts_duration_forward(stts_ctx* c, void* stream, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev, const int64_t* tokens,
                          float* logits_out, int32_t* dur_out, float* mu_out, float* style_out, float* prosody_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  PH_CHECK(STTS_W_DURATION);
  STTS_TRY(seg_ok(n_utt, tok_off_host, tok_off_dev));
  Seg s{n_utt, tok_off_host, tok_off_dev};
  Arena a(ws, ws_bytes);
  return duration_forward(c, M, st, s, (const long*)tokens, logits_out, 16, dur_out, mu_out, style_out, prosody_out, a);
  API_END
}

int stts_pitch_energy_forward(stts_ctx* c, void* stream, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev,
                              const int32_t* frm_off_host, const int32_t* frm_off_dev, const int32_t* dur, const float* pe_enc, int ld_enc,
                              const float* pe_style, float* f0_out, float* energy_out, float* prosody_out, float* cross_out, void* ws,
                              size_t ws_bytes, int seg_flags) {
  API_BEGIN
  PH_CHECK(STTS_W_PITCH_ENERGY);
  STTS_TRY(seg_ok(n_utt, tok_off_host, tok_off_dev));
  STTS_TRY(seg_ok(n_utt, frm_off_host, frm_off_dev));
  STTS_CHECK(ld_enc >= c->d.pe_inter && ld_enc % 4 == 0, "bad ld_enc");
  Seg sp{n_utt, tok_off_host, tok_off_dev}, sf{n_utt, frm_off_host, frm_off_dev};
  sf.cap = (seg_flags & STTS_SEG_CAPACITY) != 0;
  Arena a(ws, ws_bytes);
  return pitch_energy_forward(c, M, st, sp, sf, dur, pe_enc, ld_enc, pe_style, f0_out, energy_out, prosody_out, cross_out, a);
  API_END
}

int stts_duration_decode(void* stream, const float* logits, int ld, int n_rows, int32_t* dur_out) {
  API_BEGIN
  STTS_CHECK(logits && dur_out && ld >= 16 && n_rows > 0, "bad argument");
  hipLaunchKernelGGL(duration_decode_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, logits, ld, 16, n_rows, dur_out);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

int stts_duration_to_alignment(void* stream, const int32_t* dur, int n_tokens, int n_frames, float* alignment_out) {
  API_BEGIN
  STTS_CHECK(dur && alignment_out && n_tokens > 0 && n_tokens <= 1024 && n_frames > 0, "bad argument (at most 1024 tokens)");
  const long total = (long)n_tokens * n_frames;
  hipLaunchKernelGGL(alignment_matrix_kernel, dim3((unsigned)std::min<long>(1024, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dur, n_tokens,
                     n_frames, alignment_out);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

// ------------------------------------------------------------------------------------------------ conv-form STFT (ONNX export)
static int conv_stft_tables(stts_ctx* c) {
  if (c->hann) return 0;
  STTS_HIP(hipSetDevice(c->device));
  c->cur_tag = 0;  // context-lifetime tables, whatever was finalized last
  std::vector<float> h(kWin);
  for (int i = 0; i < kWin; ++i) h[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / kWin));
  STTS_TRY(dev_upload(c, h, &c->hann));
  std::vector<float2> tw(kNfft / 2);
  std::vector<double2> tw64(kNfft / 2);
  for (int i = 0; i < kNfft / 2; ++i) {
    tw64[i] = make_double2(cos(2.0 * M_PI * i / kNfft), -sin(2.0 * M_PI * i / kNfft));
    tw[i] = make_float2((float)tw64[i].x, (float)tw64[i].y);
  }
  STTS_TRY(dev_upload(c, tw, &c->twiddle));
  STTS_TRY(dev_upload(c, tw64, &c->twiddle64));
  return 0;
}

int stts_conv_stft_transform(stts_ctx* c, void* stream, int n_utt, const int32_t* frame_off_host, const int32_t* frame_off_dev, const float* wave,
                             int hop, float* mag, float* x, float* y, int ld) {
  API_BEGIN
  STTS_CHECK(c && wave && mag && x && y && hop > 0 && ld >= kBins, "bad argument");
  STTS_CHECK(c->d.n_fft == kNfft && c->d.win_length == kWin, "conv STFT: built for n_fft 2048 / win 1200 (model.yml)");
  STTS_TRY(seg_ok(n_utt, frame_off_host, frame_off_dev));
  STTS_TRY(conv_stft_tables(c));
  int mf = 0;
  for (int u = 0; u < n_utt; ++u) {
    STTS_CHECK(frame_off_host[u + 1] - frame_off_host[u] >= 2, "conv STFT: utterance %d needs at least 2 frames (hop samples)", u);
    mf = std::max(mf, frame_off_host[u + 1] - frame_off_host[u]);
  }
  hipLaunchKernelGGL(conv_stft_kernel, dim3(mf, n_utt), dim3(256), 0, (hipStream_t)stream, wave, frame_off_dev, hop, c->hann, c->twiddle64, mag, x, y, ld);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

int stts_conv_stft_inverse(stts_ctx* c, void* stream, int n_utt, const int32_t* frame_off_host, const int32_t* frame_off_dev, const float* mag,
                           const float* x, const float* y, int ld, int hop, float* wave_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  STTS_CHECK(c && wave_out && mag && x && y && hop > 0 && ld >= kBins, "bad argument");
  STTS_CHECK(c->d.n_fft == kNfft && c->d.win_length == kWin, "conv STFT: built for n_fft 2048 / win 1200 (model.yml)");
  STTS_TRY(seg_ok(n_utt, frame_off_host, frame_off_dev));
  STTS_TRY(conv_stft_tables(c));
  const long frames = frame_off_host[n_utt];
  STTS_CHECK(ws && ws_bytes >= (size_t)frames * kWin * sizeof(float), "conv iSTFT: workspace needs frames * 1200 floats");
  int mf = 0;
  for (int u = 0; u < n_utt; ++u) {
    STTS_CHECK(frame_off_host[u + 1] - frame_off_host[u] >= 2, "conv iSTFT: utterance %d needs at least 2 frames", u);
    mf = std::max(mf, frame_off_host[u + 1] - frame_off_host[u]);
  }
  float* yw = (float*)ws;
  hipLaunchKernelGGL(conv_istft_frames_kernel, dim3(mf, n_utt), dim3(256), 0, (hipStream_t)stream, mag, x, y, ld, frame_off_dev, c->hann, c->twiddle, yw);
  hipLaunchKernelGGL(conv_istft_ola_kernel, dim3(std::min(1024, ceil_div((mf - 1) * hop, 256)), n_utt), dim3(256), 0, (hipStream_t)stream, yw, frame_off_dev,
                     hop, wave_out);
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

// ------------------------------------------------------------------------------------------------ profiling
int stts_profile_begin(void) {
  gemm_profiler().begin();
  return 0;
}

int stts_profile_end(void* stream, int* launches, double* total_ms, double* total_flops) {
  API_BEGIN
  GemmProfiler& p = gemm_profiler();
  p.on = false;
  STTS_HIP(hipStreamSynchronize((hipStream_t)stream));
  double ms = 0, fl = 0;
  int n = 0;
  for (size_t i = 0; i < p.recs.size() && 2 * i + 1 < p.used; ++i) {
    if (p.recs[i].kind != 0) continue;  // the contraction kernels only (the other kernels: stts_profile_report)
    float t = 0;
    STTS_HIP(hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]));
    ms += t;
    fl += p.recs[i].flops;
    ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return 0;
  API_END
}

int stts_profile_report(void* stream, char* json, size_t cap) {
  API_BEGIN
  STTS_CHECK(json && cap > 2, "bad argument");
  GemmProfiler& p = gemm_profiler();
  p.on = false;
  STTS_HIP(hipStreamSynchronize((hipStream_t)stream));
  struct Agg {
    int kind = 0, n = 0;
    double ms = 0, flops = 0, exec = 0, bytes = 0;
  };
  std::map<std::string, Agg> agg;
  std::vector<std::string> order;
  for (size_t i = 0; i < p.recs.size() && 2 * i + 1 < p.used; ++i) {
    float t = 0;
    STTS_HIP(hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]));
    const ProfRec& r = p.recs[i];
    if (getenv("STTS_PROF_DUMP"))  // diagnostics: every launch in issue order
      fprintf(stderr, "[prof] %4zu %-28s %9.2f us %10.3f GFLOP (%.3f executed) %9.3f MB\n", i, r.name, 1e3 * t, r.flops * 1e-9, r.exec_flops * 1e-9, r.bytes * 1e-6);
    if (!agg.count(r.name)) order.push_back(r.name);
    Agg& g = agg[r.name];
    g.kind = r.kind;
    ++g.n;
    g.ms += t;
    g.flops += r.flops;
    g.exec += r.exec_flops;
    g.bytes += r.bytes;
  }
  std::string out = "[";
  for (size_t k = 0; k < order.size(); ++k) {
    const Agg& g = agg[order[k]];
    char buf[512];
    snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"kind\": \"%s\", \"launches\": %d, \"ms\": %.6f, \"gflop\": %.4f, \"executed_gflop\": %.4f, \"mbytes\": %.4f}",
             k ? ", " : "", order[k].c_str(), g.kind == 0 ? "contraction" : "other", g.n, g.ms, g.flops * 1e-9, g.exec * 1e-9, g.bytes * 1e-6);
    out += buf;
  }
  out += "]";
  STTS_CHECK(out.size() + 1 <= cap, "profile report needs %zu bytes", out.size() + 1);
  memcpy(json, out.c_str(), out.size() + 1);
  return 0;
  API_END
}

}  // extern "C"

#include "mrf_block.hip.h"
#ifdef STTS_TEST_OPS  // single-layer test operators + the contraction micro-benchmark (not part of the product surface)
#include "test_ops.hip.h"
#endif
