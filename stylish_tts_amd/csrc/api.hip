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
  delete c;
}

int stts_set_precision(stts_ctx* c, int precision) {
  API_BEGIN
  STTS_CHECK(c, "bad argument");
  STTS_CHECK(precision >= 0 && precision <= 2, "precision must be STTS_PREC_F32, _BF16 or _F16");
  STTS_CHECK(c->ready == 0 || precision == c->prec, "precision must be chosen before weights are finalized");
  c->prec = precision;
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

int stts_finalize_weights(stts_ctx* c, int which) {
  API_BEGIN
  STTS_CHECK(c, "null ctx");
  STTS_HIP(hipSetDevice(c->device));
  free_component_allocs(c, which);  // re-finalizing: the previous packing of these components goes away
  if (which & (STTS_W_DECODER | STTS_W_FLOW | STTS_W_GENERATOR)) STTS_TRY(finalize_frame(c, which));
  const int ph = which & (STTS_W_SPEECH_TEXT | STTS_W_DURATION | STTS_W_PE_TEXT | STTS_W_PE_STYLE | STTS_W_PITCH_ENERGY);
  if (ph) {
    if (!c->phoneme) c->phoneme = std::make_shared<PhonemeModel>();
    STTS_TRY(finalize_phoneme(c, static_cast<PhonemeModel*>(c->phoneme.get()), ph));
    c->ready |= ph;
    c->cur_tag = 0;
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
    STTS_HIP(hipMemset(c->d_err, 0, sizeof(int)));
    if (e == 2) return stts::fail("text encoder: token id outside [0, tokens)");
    return stts::fail("harmonic source: a frame is voiced (f0 > 10 Hz) but no f0 exceeds 20 Hz (reference raises: models/generator.py:285)");
  }
  return 0;
  API_END
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
  STTS_CHECK(ld_mel % 4 == 0 && ld_har % 32 == 0 && ld_har >= round_up(kBins, 32), "har/mel leading dimension must cover 1056 columns, multiple of 32");
  STTS_CHECK(!logamp_out || ld_lp >= kBins, "ld_lp too small");
  Arena a(ws, ws_bytes);
  return vocoder_forward(c, st, s, mel, ld_mel, style, har_spec, har_phase, ld_har, audio_out, logamp_out, phase_out, ld_lp, a);
  API_END
}

int stts_frame_path(stts_ctx* c, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* asr, int ld_asr,
                    const float* pitch, const float* energy, const float* style, const float* prior_noise, const float* src_noise,
                    const float* init_phase, int batch_scope, float* audio_out, void* ws, size_t ws_bytes) {
  API_BEGIN
  SEG_CHECK(STTS_W_DECODER | STTS_W_FLOW | STTS_W_GENERATOR);
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
                     src_row_ws, out, ld_out, 0, C, (int)n_frames);
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
  free_component_allocs(c, STTS_W_CFM);
  c->ready &= ~STTS_W_CFM;
  CfmDims d;
  d.feat = dims->feat_dim; d.asr = dims->asr_dim; d.spk = dims->spk_dim; d.hidden = dims->hidden_dim; d.emb = dims->emb_dim; d.depth = dims->depth;
  d.enc_blocks = dims->enc_blocks; d.dec_blocks = dims->dec_blocks; d.prev_depth = dims->prev_depth; d.post_depth = dims->post_depth; d.head_dim = dims->head_dim;
  auto m = std::make_shared<CfmModel>();
  c->cur_tag = STTS_W_CFM;
  const int rc = finalize_cfm(c, d, m.get());
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
                              size_t ws_bytes) {
  API_BEGIN
  PH_CHECK(STTS_W_PITCH_ENERGY);
  STTS_TRY(seg_ok(n_utt, tok_off_host, tok_off_dev));
  STTS_TRY(seg_ok(n_utt, frm_off_host, frm_off_dev));
  STTS_CHECK(ld_enc >= c->d.pe_inter && ld_enc % 4 == 0, "bad ld_enc");
  Seg sp{n_utt, tok_off_host, tok_off_dev}, sf{n_utt, frm_off_host, frm_off_dev};
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

// ------------------------------------------------------------------------------------------------ test operators
int stts_op_conv1d(void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* x, int ldx, int cin,
                   const float* w_host, const float* bias_host, int cout, int k, int dil, int act, float* y, int ldy, int force_tile, int precision) {
  API_BEGIN
  hipStream_t st = (hipStream_t)stream;
  STTS_CHECK(ldx % 32 == 0 && ldx >= cin, "op_conv1d: ldx must be a multiple of 32 covering cin");
  STTS_CHECK(precision >= 0 && precision <= 2, "precision must be STTS_PREC_F32, _BF16 or _F16");
  stts_ctx tmp;  // only for allocation bookkeeping
  tmp.prec = precision;
  struct FreeAll {  // every exit path (including the early STTS_TRY / STTS_CHECK returns) waits for the stream and frees the temporaries
    stts_ctx& t;
    hipStream_t st;
    ~FreeAll() {
      (void)hipStreamSynchronize(st);
      for (void* p : t.allocs) (void)hipFree(p);
    }
  } free_all{tmp, st};
  HostTensor w;
  w.shape = {cout, cin, k};
  w.data.assign(w_host, w_host + (size_t)cout * cin * k);
  HostTensor b;
  b.shape = {cout};
  if (bias_host) b.data.assign(bias_host, bias_host + cout);
  Seg s{n_utt, seg_off_host, seg_off_dev};
  if (force_tile == -4) {  // the Winograd F(6, k) form (k = 3 or 7, dilation 1): winograd.hip.h
    STTS_CHECK(dil == 1 && precision == 0, "op_conv1d: the Winograd form is fp32, dilation 1");
    WinoConv wc;
    STTS_TRY(pack_winograd(&tmp, w, bias_host ? &b : nullptr, 0, cin, cout, &wc));
    float* scratch = nullptr;
    STTS_HIP(hipMalloc(&scratch, wino_scratch_floats(s, wc) * sizeof(float)));
    tmp.allocs.push_back(scratch);
    WinoScratch wz;
    wz.p = scratch;
    STTS_TRY(run_winograd(st, s, x, ldx, wc, y, ldy, act, nullptr, 0, 1.0f, wz));
    STTS_HIP(hipStreamSynchronize(st));
    return 0;
  }
  PackedConv pc;
  STTS_TRY(pack_rows(&tmp, w, bias_host ? &b : nullptr, plain_rows(cout), 0, cin, round_up(cin, 32), cout, &pc));
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, x, ldx, 0, pc, (k - 1) / 2, dil);
  a.N = cout; a.bias = pc.bias; a.Y = y; a.ldy = ldy; a.act = act;
  if (force_tile >= 100) {  // tests: the contraction reads 16-bit activation rows (rounded copy of x), tile = force_tile - 100
    STTS_CHECK(precision != 0, "op_conv1d: 16-bit activation rows need a 16-bit operand mode");
    unsigned short* x16 = nullptr;
    STTS_HIP(hipMalloc(&x16, (size_t)s.rows() * ldx * sizeof(unsigned short)));
    tmp.allocs.push_back(x16);
    launch_cast_rows(st, precision, x, ldx, ldx, x16, ldx, s.rows());
    a.seg[0].X = reinterpret_cast<const float*>(x16);
    a.x16 = 1;
    force_tile -= 100;
  }
  STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, pc.npad, n_utt, s.max_len(), force_tile));
  STTS_HIP(hipStreamSynchronize(st));
  return 0;
  API_END
}

static int op_scratch(Arena& a, const Seg& s, int kc, int cout, float** act1, float** h, float** act2, float** ss, float** sty, int ld_sty) {
  const long R = s.rows();
  const int n_utt = s.n_utt;
  *act1 = a.get<float>(R * kc);
  *h = a.get<float>(R * cout);
  *act2 = a.get<float>(R * cout);
  *ss = a.get<float>(adain_part_floats(s, std::max(kc, cout)));
  *sty = a.get<float>((size_t)n_utt * ld_sty);
  STTS_CHECK(a.ok, "op: workspace too small");
  return 0;
}

int stts_op_adain_block(stts_ctx* c, void* stream, const char* prefix, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                        const float* x, int ldx, int cin, int cout, const float* style, float* y, int ldy, void* ws, size_t ws_bytes) {
  API_BEGIN
  STTS_CHECK(c && prefix, "null argument");
  hipStream_t st = (hipStream_t)stream;
  std::string key = prefix;
  if (!c->op_blocks.count(key)) {
    auto blk = std::make_unique<AdainBlockW>();
    auto tab = std::make_unique<StyleTable>();
    STTS_TRY(pack_adain_block(c, key, cin, cout, tab.get(), blk.get()));
    STTS_TRY(upload_table(c, tab.get()));
    c->op_blocks[key] = std::move(blk);
    c->op_tables[key] = std::move(tab);
  }
  const AdainBlockW& B = *c->op_blocks[key];
  const StyleTable& T = *c->op_tables[key];
  STTS_CHECK(ldx == B.kcin, "op_adain_block: ldx must equal cin padded to 32 (%d)", B.kcin);
  Seg s{n_utt, seg_off_host, seg_off_dev};
  Arena a(ws, ws_bytes);
  float *act1, *h, *act2, *ss, *sty;
  STTS_TRY(op_scratch(a, s, B.kcin, B.cout, &act1, &h, &act2, &ss, &sty, T.ld()));
  STTS_TRY(run_style(st, T, style, n_utt, sty));
  return run_adain_block(st, s, B, sty, T.ld(), x, ldx, y, ldy, act1, h, act2, ss);
  API_END
}

int stts_op_mrf_block(stts_ctx* c, void* stream, const char* prefix, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                      const float* x, int ldx, int channels, int kernel, const float* style, float* y, int ldy, void* ws, size_t ws_bytes) {
  API_BEGIN
  STTS_CHECK(c && prefix, "null argument");
  STTS_CHECK(channels % 32 == 0 && ldx == channels && ldy == channels, "op_mrf_block: channels must be a multiple of 32 and ld == channels");
  hipStream_t st = (hipStream_t)stream;
  std::string key = prefix;
  if (!c->op_mrf.count(key)) {
    auto m = std::make_unique<MrfW>();
    m->channels = channels;
    m->kernel = kernel;
    for (int i = 0; i < 3; ++i) {
      const std::string si = std::to_string(i);
      STTS_TRY(pack_plain(c, key + "convs1." + si, true, 0, channels, &m->c1[i]));
      STTS_TRY(pack_plain(c, key + "convs2." + si, true, 0, channels, &m->c2[i]));
      STTS_TRY(add_style(c, &m->table, key + "adain1." + si, channels, &m->a1[i]));
      STTS_TRY(add_style(c, &m->table, key + "adain2." + si, channels, &m->a2[i]));
      STTS_GET(al1, key + "alpha1." + si);
      STTS_GET(al2, key + "alpha2." + si);
      STTS_TRY(dev_upload(c, al1->data, &m->alpha1[i]));
      STTS_TRY(dev_upload(c, al2->data, &m->alpha2[i]));
    }
    STTS_TRY(upload_table(c, &m->table));
    c->op_mrf[key] = std::move(m);
  }
  const MrfW& M = *c->op_mrf[key];
  Seg s{n_utt, seg_off_host, seg_off_dev};
  const long R = s.rows();
  Arena a(ws, ws_bytes);
  float* cur = a.get<float>(R * channels);
  float* t1 = a.get<float>(R * channels);
  float* t2 = a.get<float>(R * channels);
  float* ss = a.get<float>(adain_part_floats(s, channels));
  float* sty = a.get<float>((size_t)n_utt * M.table.ld());
  STTS_CHECK(a.ok, "op_mrf_block: workspace too small");
  STTS_TRY(run_style(st, M.table, style, n_utt, sty));
  STTS_HIP(hipMemcpyAsync(cur, x, R * channels * sizeof(float), hipMemcpyDeviceToDevice, st));
  const int ml = s.max_len(), lds = M.table.ld();
  // 3 x { AdaIN -> Snake -> dilated conv -> AdaIN -> Snake -> conv -> + x }  (models/ada_norm.py:109-120)
  for (int i = 0; i < 3; ++i) {
    STTS_TRY(run_adain(st, s, cur, channels, channels, t1, channels, sty, lds, M.a1[i].col0, ACT_NONE, M.alpha1[i], ss));
    GemmArgs g1 = gemm_args(s);
    set_seg(g1, 0, t1, channels, 0, M.c1[i], (kernel - 1) / 2, M.dil[i]);
    g1.N = channels; g1.bias = M.c1[i].bias; g1.Y = t2; g1.ldy = channels;
    STTS_TRY(launch_conv_gemm(st, g1, EPI_STORE, M.c1[i].npad, n_utt, ml));
    STTS_TRY(run_adain(st, s, t2, channels, channels, t1, channels, sty, lds, M.a2[i].col0, ACT_NONE, M.alpha2[i], ss));
    GemmArgs g2 = gemm_args(s);
    set_seg(g2, 0, t1, channels, 0, M.c2[i]);
    g2.N = channels; g2.bias = M.c2[i].bias; g2.R = cur; g2.ldr = channels;
    float* dst = i == 2 ? y : t2;
    g2.Y = dst; g2.ldy = channels;
    STTS_TRY(launch_conv_gemm(st, g2, EPI_STORE, M.c2[i].npad, n_utt, ml));
    if (i < 2) std::swap(cur, t2);
  }
  STTS_HIP(hipGetLastError());
  return 0;
  API_END
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ kernel microbench
// Times `iters` back-to-back launches of conv_gemm_f32 on synthetic data (tuning aid for tools/gemm_bench.py).
extern "C" int stts_bench_gemm(void* stream, int n_utt, int rows_per_utt, int cin, int cout, int k, int tile, int iters, double* avg_ms, int tune) {
  API_BEGIN
  hipStream_t st = (hipStream_t)stream;
  const long R = (long)n_utt * rows_per_utt;
  const int kc = round_up(cin, 32), npad = round_up(cout, 128), ldy = round_up(cout, 32);
  float *X, *W, *Y, *B;
  int* so;
  STTS_HIP(hipMalloc(&X, R * kc * sizeof(float)));
  STTS_HIP(hipMalloc(&W, (size_t)npad * k * kc * sizeof(float)));
  STTS_HIP(hipMalloc(&Y, R * ldy * sizeof(float)));
  STTS_HIP(hipMalloc(&B, npad * sizeof(float)));
  STTS_HIP(hipMalloc(&so, (n_utt + 1) * sizeof(int)));
  std::vector<int> h(n_utt + 1);
  for (int i = 0; i <= n_utt; ++i) h[i] = i * rows_per_utt;
  STTS_HIP(hipMemcpy(so, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  // pseudo-random fill (zeros would flatter the clock: guide §5.4 rule 25)
  {
    std::vector<float> t((size_t)std::max<long>(R * kc, (long)npad * k * kc));
    uint32_t s = 12345;
    for (auto& v : t) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    STTS_HIP(hipMemcpy(X, t.data(), R * kc * sizeof(float), hipMemcpyHostToDevice));
    STTS_HIP(hipMemcpy(W, t.data(), (size_t)npad * k * kc * sizeof(float), hipMemcpyHostToDevice));
    STTS_HIP(hipMemset(B, 0, npad * sizeof(float)));
  }
  PackedConv pc;
  pc.W = W; pc.bias = B; pc.npad = npad; pc.N = cout; pc.kc = kc; pc.ntaps = k; pc.cin_real = cin; pc.rows_real = cout;
  unsigned short* W16 = nullptr;
  if (tune & 384) {  // bit 7: bf16 operands, bit 8: fp16 operands
    std::vector<unsigned short> h16((size_t)npad * k * kc);
    uint32_t s2 = 777;
    for (auto& v : h16) { s2 = s2 * 1664525u + 1013904223u; v = f32_to_bf16(((s2 >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    STTS_HIP(hipMalloc(&W16, h16.size() * 2));
    STTS_HIP(hipMemcpy(W16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
    pc.W16 = W16;
    pc.prec = (tune & 128) ? PREC_BF16 : PREC_F16;
  }
  Seg s{n_utt, h.data(), so};
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, X, kc, 0, pc);
  a.N = cout; a.bias = B; a.Y = Y; a.ldy = ldy; a.tune = tune & 63;
  unsigned short* X16 = nullptr;
  if ((tune & 1024) && (tune & 384)) {  // bit 10: 16-bit activation rows (X rounded once, outside the timed launches)
    STTS_HIP(hipMalloc(&X16, R * kc * sizeof(unsigned short)));
    launch_cast_rows(st, pc.prec, X, kc, kc, X16, kc, R);
    a.seg[0].X = reinterpret_cast<const float*>(X16);
    a.x16 = 1;
  }
  long long* dbg = nullptr;
  const size_t dbg_n = 8 * 16384;
  if (tune & 64) { STTS_HIP(hipMalloc(&dbg, dbg_n * 8)); STTS_HIP(hipMemset(dbg, 0, dbg_n * 8)); }
  a.dbg = dbg;
  hipEvent_t e0, e1;
  STTS_HIP(hipEventCreate(&e0));
  STTS_HIP(hipEventCreate(&e1));
  if (tune & 512) {  // the Winograd F(6, k) form of the same conv, transforms included (k = 3 or 7)
    stts_ctx tmp;
    HostTensor hw;
    hw.shape = {cout, cin, k};
    hw.data.resize((size_t)cout * cin * k);
    uint32_t s3 = 4242;
    for (auto& v : hw.data) { s3 = s3 * 1664525u + 1013904223u; v = (((s3 >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.05f; }
    WinoConv wc;
    STTS_TRY(pack_winograd(&tmp, hw, nullptr, 0, cin, cout, &wc));
    float* scratch = nullptr;
    STTS_HIP(hipMalloc(&scratch, wino_scratch_floats(s, wc) * sizeof(float)));
    WinoScratch wz;
    wz.p = scratch;
    for (int i = 0; i < 2; ++i) STTS_TRY(run_winograd(st, s, X, kc, wc, Y, ldy, 0, nullptr, 0, 1.0f, wz));
    STTS_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) STTS_TRY(run_winograd(st, s, X, kc, wc, Y, ldy, 0, nullptr, 0, 1.0f, wz));
    STTS_HIP(hipEventRecord(e1, st));
    STTS_HIP(hipEventSynchronize(e1));
    float msw = 0;
    STTS_HIP(hipEventElapsedTime(&msw, e0, e1));
    *avg_ms = msw / iters;
    (void)hipFree(scratch);
    for (void* p : tmp.allocs) (void)hipFree(p);
    (void)hipFree(X); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(B); (void)hipFree(so);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
  }
  for (int i = 0; i < 2; ++i) STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, npad, n_utt, rows_per_utt, tile));
  STTS_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, npad, n_utt, rows_per_utt, tile));
  STTS_HIP(hipEventRecord(e1, st));
  STTS_HIP(hipEventSynchronize(e1));
  float ms = 0;
  STTS_HIP(hipEventElapsedTime(&ms, e0, e1));
  *avg_ms = ms / iters;
  if (dbg) {
    std::vector<long long> hdb(dbg_n);
    STTS_HIP(hipMemcpy(hdb.data(), dbg, dbg_n * 8, hipMemcpyDeviceToHost));
    // per-block records of the LAST launch: [t0, t1, t2, t3, iters, hw_id, xcc_id, -]
    long long tmin = -1, tmax = 0;
    std::map<long long, std::vector<std::pair<long long, long long>>> cu;
    int nb = 0;
    double pro = 0, loop = 0, epi = 0;
    for (size_t b = 0; b < dbg_n / 8; ++b) {
      const long long* r = &hdb[8 * b];
      if (!r[0]) continue;
      ++nb;
      if (tmin < 0 || r[0] < tmin) tmin = r[0];
      if (r[3] > tmax) tmax = r[3];
      pro += r[1] - r[0]; loop += r[2] - r[1]; epi += r[3] - r[2];
      const long long hw = r[5], key = ((r[6] & 15) << 16) | (hw & 0xFF00);  // xcc | se, sh, cu
      cu[key].push_back({r[0], r[3]});
    }
    int hist[8] = {0};
    long long worst = 0;
    for (auto& kv : cu) {
      hist[std::min<size_t>(kv.second.size(), 7)]++;
      long long e = 0;
      for (auto& p : kv.second) e = std::max(e, p.second);
      worst = std::max(worst, e - tmin);
    }
    fprintf(stderr, "  blocks %d on %zu CUs; blocks/CU histogram 1:%d 2:%d 3:%d 4:%d 5+:%d; span %.1f us; avg prologue %.1f loop %.1f epilogue %.1f us\n", nb,
            cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], (tmax - tmin) * 0.01, pro / nb * 0.01, loop / nb * 0.01,
            epi / nb * 0.01);
    // start-time spread and per-block duration spread
    long long smax = 0, dmin = 1LL << 60, dmax = 0;
    for (size_t b = 0; b < dbg_n / 8; ++b) {
      const long long* r = &hdb[8 * b];
      if (!r[0]) continue;
      smax = std::max(smax, r[0] - tmin);
      dmin = std::min(dmin, r[3] - r[0]);
      dmax = std::max(dmax, r[3] - r[0]);
    }
    fprintf(stderr, "  latest start +%.1f us; block duration min %.1f max %.1f us\n", smax * 0.01, dmin * 0.01, dmax * 0.01);
    double xd[8] = {0}, xc[8] = {0}; int xn[8] = {0};
    for (size_t b = 0; b < dbg_n / 8; ++b) {
      const long long* r = &hdb[8 * b];
      if (!r[0]) continue;
      const int x = r[6] & 7;
      xd[x] += (r[3] - r[0]) * 0.01; xc[x] += (double)r[7] / ((r[3] - r[0]) * 0.01); xn[x]++;
    }
    {
      std::vector<double> du;
      for (size_t b = 0; b < dbg_n / 8; ++b) if (hdb[8 * b]) du.push_back((hdb[8 * b + 3] - hdb[8 * b]) * 0.01);
      std::sort(du.begin(), du.end());
      fprintf(stderr, "  duration percentiles us: p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f\n", du[du.size() / 20], du[du.size() / 4], du[du.size() / 2],
              du[du.size() * 3 / 4], du[du.size() * 19 / 20], du.back());
      // by original linear block id modulo 64 (8 XCDs x 8): shows placement patterns
      const int gx = npad / 128;
      double byd[16] = {0}; int byn[16] = {0};
      for (size_t b = 0; b < dbg_n / 8; ++b) {
        if (!hdb[8 * b]) continue;
        const int cuid = (hdb[8 * b + 5] >> 8) & 15;
        byd[cuid] += (hdb[8 * b + 3] - hdb[8 * b]) * 0.01; byn[cuid]++;
      }
      (void)gx;
      fprintf(stderr, "  avg by cu_id:");
      for (int i = 0; i < 16; ++i) if (byn[i]) fprintf(stderr, " %d:%.0f(%d)", i, byd[i] / byn[i], byn[i]);
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "  per XCC avg block us / shader MHz:");
    for (int x = 0; x < 8; ++x) if (xn[x]) fprintf(stderr, " %d:%.0f/%.0f", x, xd[x] / xn[x], xc[x] / xn[x]);
    fprintf(stderr, "\n");
    (void)hipFree(dbg);
  }
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(B); (void)hipFree(so);
  if (W16) (void)hipFree(W16);
  if (X16) (void)hipFree(X16);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
  API_END
}
