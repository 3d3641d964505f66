/* stylish_hip.h — C-ABI of the MI355X (gfx950) implementation of the Stylish-TTS inference hot path.
 *
 * The reference (Fannovel16/stylish-tts) is 100 % Python: it has no FFI, plugin or operator interface, only
 * nn.Module.forward() signatures.  This library sits UNDERNEATH those modules: the Python
 * shims in stylish_tts_amd/modules.py keep the reference's constructor arguments, forward() signatures and
 * state_dict keys and call the entry points below through ctypes.  Each entry point names the reference
 * function it replaces (paths relative to /root/reference/src/stylish_tts/).
 *
 * Conventions
 *  - plain C: pointers, sizes, int status (0 = ok; stts_last_error() gives the text).  Never throws.
 *  - `stream` is a hipStream_t passed as void*.  All work is enqueued on it; nothing synchronises unless stated.
 *  - device tensors are fp32, TIME-MAJOR packed rows: X[utterance offset + frame][channel], row stride `ld`
 *    floats (a multiple of 4).  Utterances are described by seg_off[n_utt + 1] (int32 row offsets), passed both
 *    as a host array (grid sizing) and as a device array (kernels).
 *  - the caller owns inputs, outputs and the workspace; the library owns the context and the packed weights.
 *  - noise is an explicit input (the reference draws it from the global torch generator:
 *    models/flow.py:314, models/generator.py:272,306).
 */
#ifndef STYLISH_HIP_H_
#define STYLISH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stts_ctx stts_ctx;

/* Model shape: the hot-path keys of ModelConfig (lib/config_loader.py:369-414, train/config/model.yml). */
typedef struct stts_model_dims {
  int32_t n_fft, win_length, hop_length, sample_rate; /* 2048, 1200, 300, 24000 (vocoder hop = hop_length/4) */
  int32_t style_dim, inter_dim;                      /* 64, 128 */
  int32_t dec_hidden, dec_residual;                  /* decoder.hidden_dim 512, residual_dim 64 */
  int32_t gen_input, gen_hidden, gen_inter, gen_io_kernel; /* generator.* : 512, 512, 1536, 7 */
  int32_t tokens, te_hidden, te_filter, te_heads, te_layers, te_kernel; /* text_encoder.* */
  int32_t style_layers;                              /* style_encoder.layers */
  int32_t dur_layers, dur_classes, dur_max;          /* duration_predictor.n_layer, duration_classes, max_duration */
  int32_t pe_inter;                                  /* pitch_energy_predictor.inter_dim */
} stts_model_dims;

const char* stts_last_error(void);
int stts_version(void);

/* Context: device selection + weights.  Weight tensors are handed over by their reference state_dict name,
 * prefixed with the module name used by build_model (models/models.py:79-101), e.g.
 * "speech_predictor.decoder.encode.conv1.parametrizations.weight.original1".  Data is copied (host fp32). */
int stts_ctx_create(const stts_model_dims* dims, int device, stts_ctx** out);
void stts_ctx_destroy(stts_ctx* ctx);
int stts_load_weight(stts_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim);
/* Fold weight-norm (w = g*v/||v||), re-lay out every conv/linear as W[cout][tap][cin] padded for the MFMA
 * tiles, upload.  `which` is a bit mask of the components whose weights have been loaded (one per reference
 * sub-module, so that a single module can be dropped in on its own): */
enum {
  STTS_W_DECODER = 1,        /* speech_predictor.decoder                                   models/decoder.py */
  STTS_W_FLOW = 2,           /* speech_predictor.{prior_encoder, flow, post_flow}          models/flow.py */
  STTS_W_GENERATOR = 4,      /* speech_predictor.generator                                 models/generator.py */
  STTS_W_SPEECH_TEXT = 8,    /* speech_predictor.{text_encoder, style_encoder}             models/speech_predictor.py:17-25 */
  STTS_W_DURATION = 16,      /* duration_predictor.*                                       models/duration_predictor.py */
  STTS_W_PE_TEXT = 32,       /* pe_text_encoder                                            models/models.py:49-52 */
  STTS_W_PE_STYLE = 64,      /* pe_text_style_encoder                                      models/models.py:53-57 */
  STTS_W_PITCH_ENERGY = 128, /* pitch_energy_predictor.*                                   models/pitch_energy_predictor.py */
  STTS_W_FRAME_PATH = 7,
  STTS_W_ALL = 255,
  STTS_W_CFM = 256           /* cfm_mel_decoder.* (finalized by stts_cfm_finalize, not part of STTS_W_ALL)  models/cfm/cfm_mel_decoder.py */
};
int stts_finalize_weights(stts_ctx* ctx, int which);
/* Operand precision of the FRAME-RATE Conv1d / Linear contractions (call before the first stts_finalize_weights).
 * F32 is the reference's arithmetic (BASELINE cfg2).  BF16 / F16 (BASELINE cfg3 / cfg5): the operands of every matrix-core
 * contraction of the decoder, the flow and the vocoder are rounded to nearest-even exactly once - weights when they are packed,
 * activations either when a tile is staged for the matrix cores (calls of < 3 840 rows: they stay fp32 in HBM) or by the kernel
 * that produces them (larger calls: contraction inputs are 16-bit rows in HBM; the same arithmetic) - and products accumulate
 * in fp32; norms, gates, style projections, FFTs and every non-contraction kernel stay fp32.
 * The PHONEME-RATE predictors (text encoders, style encoders, duration and pitch / energy predictors) always run in fp32:
 * durations are integers (bit-exact against the fp32 reference in every mode) and those stages are latency-bound.
 * F32, how the fp32 products are formed (round 4): every fp32 operand is the EXACT sum of three bf16 numbers (8 + 8 + 8 significand
 * bits), so x * w is nine bf16 x bf16 products, each exact in fp32; the frame-rate contractions run the six largest on the bf16
 * matrix cores with fp32 accumulation (the three dropped terms are <= 2^-23 |x w|, 2^-27 |x w| rms, zero-mean: below the
 * rounding of the fp32 accumulation that both forms share).  Nothing is rounded to 16 bits: results agree with the
 * f32 matrix cores' to fp32 accumulation noise (tests/test_hip_split_fp32.py measures both against float64).
 * F32_NATIVE keeps every fp32 contraction on v_mfma_f32_32x32x2_f32 (rounds 1-3; also process-wide with STTS_NO_X3=1). */
enum { STTS_PREC_F32 = 0, STTS_PREC_BF16 = 1, STTS_PREC_F16 = 2, STTS_PREC_F32_NATIVE = 3 };
int stts_set_precision(stts_ctx* ctx, int precision);
/* Reads the device-side error word (sets last_error): 1 = a voiced frame exists but no f0 > 20 Hz (the reference raises there,
 * models/generator.py:285), 2 = a token id outside the embedding table, 4 = an utterance
 * too short for the STFT's reflect padding.  Synchronises the stream. */
int stts_check_status(stts_ctx* ctx, void* stream);

/* Workspace the frame-rate stages need for `rows` vocoder frames (sum of T4) in `n_utt` utterances,
 * the longest having `max_len` frames. */
size_t stts_frame_workspace_bytes(const stts_ctx* ctx, int64_t rows, int n_utt, int max_len);
/* Row stride (floats) of the harmonic spectra `har_spec` / `har_phase` that stts_harmonic_stft writes and stts_vocoder_forward reads: the
 * n_fft/2 + 1 bins padded to the packed input width of the prior convs (32 in fp32, 64 in the 16-bit operand modes).  Valid after
 * stts_set_precision; callers size their buffers with it instead of mirroring the padding rule. */
int stts_har_ld(const stts_ctx* ctx);

/* Decoder.forward (models/decoder.py:47-60) incl. every AdaptiveDecoderBlock (models/ada_norm.py:166-182).
 * asr [rows, ld_asr>=128], pitch/energy [rows] (already at the hop/4 rate), style [n_utt, 64] -> x [rows, 512]. */
int stts_decoder_forward(stts_ctx* ctx, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                         const float* asr, int ld_asr, const float* pitch, const float* energy, const float* style,
                         float* x_out, int ld_x, void* ws, size_t ws_bytes);

/* PriorEncoder.forward (models/flow.py:311-315) + ResidualCouplingBlock.forward(reverse=True)
 * (models/flow.py:132-151) + post_flow (models/speech_predictor.py:111).
 * x [rows,512], prior_noise [rows,128] ~ N(0,1) -> mel [rows,512]; optional z_prior / z_flow [rows,128]. */
int stts_prior_flow_forward(stts_ctx* ctx, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                            const float* x, int ld_x, const float* style, const float* prior_noise, float* mel_out, int ld_mel,
                            float* z_prior_out, float* z_flow_out, void* ws, size_t ws_bytes);

/* generate_pcph (models/generator.py:247-315) + TorchSTFT.transform + atan2 (models/generator.py:32-44,406-410).
 * pitch [rows], src_noise [75*rows] ~ N(0,1), init_phase: device pointer to ONE float in [0,1) shared by the call.
 * batch_scope != 0: harmonic count from the min f0 of the whole call (reference semantics of a batched call);
 * 0: per utterance (the reference called per utterance).  Outputs har_spec / har_phase [rows, ld>=1025],
 * optional prior_signal [75*rows].
 * PARITY NOTE (conditional): har_phase = atan2(Im, Re) is discontinuous and feeds phase_prior_conv linearly
 * (models/generator.py:408-413).  With center=True reflect padding frame 0 is even-symmetric, so its spectrum is real up to FFT
 * rounding and every negative-real bin is +pi or -pi by the sign of rounding noise - in torch.stft too; ~0-magnitude bins have
 * arbitrary phase.  This entry computes the transform in fp64 (the signs of the exact transform); it does NOT reproduce torch's
 * coin flips.  Consequence: waveforms match the reference within 1e-3 everywhere only after the reference's value is adopted at
 * those ill-conditioned bins (~0.09 % of the bins; tests do so through oracle.align_branch, which accepts a bin only if the two
 * angles agree mod 2 pi within 5e-3 or the magnitude is < 2e-4).  Un-adopted (what stts_frame_path computes) the first ~40 frames of
 * an utterance can differ from a given torch run by up to ~0.2, the rest by <= 1.5e-2; tests/test_hip_benchmarked_path.py pins
 * that every such difference lies inside the receptive field of an adopted bin. */
int stts_harmonic_stft(stts_ctx* ctx, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                       const float* pitch, const float* src_noise, const float* init_phase, int batch_scope,
                       float* prior_signal_out, float* har_spec, float* har_phase, int ld_har, void* ws, size_t ws_bytes);

/* Generator.forward body + TorchSTFT.inverse + tanh (models/generator.py:412-433, ConvNeXtBlock :468-485,
 * GRN :496-499, AdaptiveLayerNorm models/ada_norm.py:193-201).
 * mel [rows,512], style [n_utt,64], har_spec/har_phase [rows, ld_har] -> audio [75*rows];
 * optional logamp / phase [rows, ld_lp>=1025] (row T4 of the reference's replicate pad equals row T4-1). */
int stts_vocoder_forward(stts_ctx* ctx, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                         const float* mel, int ld_mel, const float* style, const float* har_spec, const float* har_phase, int ld_har,
                         float* audio_out, float* logamp_out, float* phase_out, int ld_lp, void* ws, size_t ws_bytes);

/* Utterance offsets of a frame-rate call may be CAPACITY SEGMENTS (seg_flags & STTS_SEG_CAPACITY): the host array then holds
 * upper bounds (cumulative capacities - every buffer, workspace and grid is sized by them), the device array the real offsets,
 * which only exist on the device (stts_frame_offsets).  Rows of utterance u are [dev[u], dev[u+1]) - packed, dev[u] <= host[u] -
 * so outputs are packed by the REAL lengths and the caller reads the device offsets once, together with the output.  This is
 * what removes the host round trip between the duration predictor and everything frame-rate (train/test_onnx.py:65-66). */
enum { STTS_SEG_CAPACITY = 1 };

/* The frame-rate hot path in one call: decoder -> prior -> reverse flow -> post_flow -> harmonic source ->
 * STFT -> vocoder -> iSTFT (models/speech_predictor.py:92-118).  This is the benchmarked unit.  seg_flags: 0 or STTS_SEG_CAPACITY. */
int stts_frame_path(stts_ctx* ctx, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev,
                    const float* asr, int ld_asr, const float* pitch, const float* energy, const float* style,
                    const float* prior_noise, const float* src_noise, const float* init_phase, int batch_scope,
                    float* audio_out, void* ws, size_t ws_bytes, int seg_flags);

/* ---- phoneme-rate predictors.  Sequences are packed: tok_off[n_utt+1] token offsets, tokens int64 [n_tok]. ---- */
size_t stts_phoneme_workspace_bytes(const stts_ctx* ctx, int64_t n_tokens, int64_t n_frames, int n_utt);

/* TextEncoder.forward (models/text_encoder.py:433-462).  which: 0 duration_predictor.text_encoder,
 * 1 speech_predictor.text_encoder, 2 pe_text_encoder.  -> mu [n_tok, ld_mu >= inter_dim] (proj_m), optional x [n_tok, 128]. */
int stts_text_encoder_forward(stts_ctx* ctx, void* stream, int which, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev,
                              const int64_t* tokens, float* mu_out, int ld_mu, float* x_out, void* ws, size_t ws_bytes);
/* TextStyleEncoder.forward (models/text_style_encoder.py:20-26; BasicConvNeXtBlock models/conv_next.py:38-51).
 * which as above (0 duration_predictor.style_encoder, 1 speech_predictor.style_encoder, 2 pe_text_style_encoder).
 * x [n_tok, ldx] -> style [n_utt, 64].  Statistics are per utterance over its own tokens (the reference at B = 1). */
int stts_text_style_forward(stts_ctx* ctx, void* stream, int which, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev,
                            const float* x, int ldx, float* style_out, void* ws, size_t ws_bytes);
/* DurationPredictor.forward (models/duration_predictor.py:30-36) -> logits [n_tok, 16]; optional
 * DurationProcessor.prediction_to_duration (train/utils.py:468-474) -> dur int32 [n_tok]; optional taps
 * text_encoder mu [n_tok,128], style [n_utt,64], prosody [n_tok,192]. */
int stts_duration_forward(stts_ctx* ctx, void* stream, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev,
                          const int64_t* tokens, float* logits_out, int32_t* dur_out, float* mu_out, float* style_out, float* prosody_out,
                          void* ws, size_t ws_bytes);
/* PitchEnergyPredictor.forward (models/pitch_energy_predictor.py:104-121) incl. compute_cross (:83-102) with the
 * reference's inverted band mask (:194-212 vs models/text_encoder.py:255-262).  The alignment is given as integer
 * durations (train/utils.py:476-489).  pe_enc [n_tok, ld_enc >= 256], pe_style [n_utt,64] -> f0, energy [n_frames]
 * at the mel-frame rate; optional taps prosody [n_tok,320], cross [n_frames,320]. */
int stts_pitch_energy_forward(stts_ctx* ctx, void* stream, int n_utt, const int32_t* tok_off_host, const int32_t* tok_off_dev,
                              const int32_t* frm_off_host, const int32_t* frm_off_dev, const int32_t* dur, const float* pe_enc, int ld_enc,
                              const float* pe_style, float* f0_out, float* energy_out, float* prosody_out, float* cross_out, void* ws,
                              size_t ws_bytes, int seg_flags /* 0 or STTS_SEG_CAPACITY: applies to the frame offsets */);

/* The host side of DurationProcessor between the two models (train/test_onnx.py:65-66 reads the durations on the host to size the
 * alignment), on the device: from the integer durations of a packed batch, off_T [n_utt+1] = cumulative mel frames per utterance,
 * off_T4 = 4 x (vocoder frames), need [n_utt] = frames of each utterance.  cap_off [n_utt+1] (device): the capacity layout the caller
 * sized its buffers by (mel frames).  An utterance that exceeds its capacity is truncated to it - every later stage stays in
 * bounds, its output is invalid - and need[u] > capacity tells the caller, who reads `need` once together with the output, to repeat
 * the call with capacities >= need (no shared error state: calls may be in flight on several streams).  All pointers are device
 * pointers; nothing synchronises. */
int stts_frame_offsets(stts_ctx* ctx, void* stream, int n_utt, const int32_t* tok_off_dev, const int32_t* dur, const int32_t* cap_off_dev,
                       int32_t* off_T_dev, int32_t* off_T4_dev, int32_t* need_dev);

/* DurationProcessor.prediction_to_duration (train/utils.py:468-474): logits [n_rows, ld >= 16] -> int32 durations. */
int stts_duration_decode(void* stream, const float* logits, int ld, int n_rows, int32_t* dur_out);
/* DurationProcessor.duration_to_alignment (train/utils.py:476-489): durations [P] -> 0/1 matrix [P, T = sum(dur)]. */
int stts_duration_to_alignment(void* stream, const int32_t* dur, int n_tokens, int n_frames, float* alignment_out);

/* Length regulator (train/utils.py:476-489 + models/speech_predictor.py:88-93): integer durations per token ->
 * time-major gather of the phoneme encoding at rate rep (1: mel frames, 4: vocoder frames).
 * dur [n_tok] int32 (device), tok_off [n_utt+1], frm_off [n_utt+1] (= rep * cumulative durations), both device.
 * enc [n_tok, ld_enc] -> out [frames, ld_out] columns [0, C).  Any number of tokens per utterance (the durations are
 * scanned in chunks); limits elsewhere: attention with heads other than 16 / 32 / 40 / 64 / 96 / 128 / 160 channels takes at most 1024 keys per utterance (the
 * reference's own limit is 510 tokens, train/dataloader.py:106-109; those head sizes run on the matrix-core kernel, which streams
 * the keys and has no limit) and stts_duration_to_alignment at most 1024 tokens. */
int stts_length_regulate(stts_ctx* ctx, void* stream, int n_utt, const int32_t* dur, const int32_t* tok_off, const int32_t* frm_off,
                         int64_t n_frames, int rep, const float* enc, int ld_enc, int C, float* out, int ld_out, int32_t* src_row_ws);
/* nn.Upsample(scale_factor=4, mode="linear") of pitch / energy (models/speech_predictor.py:64,89-90). */
int stts_upsample4(stts_ctx* ctx, void* stream, int n_utt, const int32_t* off_T_host, const int32_t* off_T, const int32_t* off_T4,
                   const float* x, float* y);

/* One explicit-Euler update of the flow-matching sampler, x += dt * v over n floats
 * (models/cfm/cfm.py:65-84, CfmSampler.solve_euler: `x = x + dt * dphi_dt`). */
int stts_euler_step(void* stream, float* x, const float* v, float dt, int64_t n);

/* The estimator of that sampler in the reference: CfmMelDecoder._forward (models/cfm/cfm_mel_decoder.py:318-398; XUT transformer,
 * models/xut/), inference mode.  Its dimensions are constructor keywords in the reference (:190-206), hence this struct.
 * Weights: stts_load_weight(ctx, "cfm_mel_decoder.<state_dict key>", ...) then stts_cfm_finalize.
 * One evaluation on a packed batch: x [rows, ld_x] time-major (feat_dim columns), asr [rows, ld_asr] (ld_asr a multiple of 32, pad
 * columns finite), f0 / n_curve: per-utterance curves back to back with offsets curve_off (resampled to the utterance's frames by
 * F.interpolate's nearest rule, :322-323), spk_emb [n_utt, spk_dim], t [n_utt], sine_noise [rows] = the one RNG draw inside the
 * estimator (SineGenerator's additive noise, :99; the caller draws it), out [rows, ld_out] = dphi/dt.  All tensors on the device.
 * Limits: head_dim 16 / 32 / 40 / 64 (the class default) / 96 / 128 / 160 for utterances beyond 1024 frames - other head sizes keep an utterance's
 * attention scores in LDS (<= 1024 keys); SineGenerator without overtones (the reference's configuration). */
typedef struct stts_cfm_dims {
  int32_t feat_dim, asr_dim, spk_dim, hidden_dim, emb_dim, depth, enc_blocks, dec_blocks, prev_depth, post_depth, head_dim;
} stts_cfm_dims;
int stts_cfm_finalize(stts_ctx* ctx, const stts_cfm_dims* dims);
size_t stts_cfm_workspace_bytes(const stts_ctx* ctx, int64_t rows, int n_utt);
int stts_cfm_estimator(stts_ctx* ctx, void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* x, int ld_x,
                       const float* asr, int ld_asr, const float* f0, const float* n_curve, const int32_t* curve_off_host,
                       const int32_t* curve_off_dev, const float* spk_emb, const float* t, const float* sine_noise, float* out, int ld_out,
                       void* workspace, size_t workspace_bytes);

/* Layout bridge for the nn.Module shims: reference [B, C, T] (equal T) <-> time-major rows. */
int stts_to_time_major(void* stream, const float* x_bct, int B, int C, int T, float* y, int ldy);
int stts_to_channel_major(void* stream, const float* x, int ldx, int B, int C, int T, float* y_bct);

/* STFT.transform / STFT.inverse of models/stft.py:98-187: the conv1d / conv_transpose1d DFT-matrix STFT that the
 * reference's ONNX export swaps into the generator (train/convert_to_onnx.py:31-36).  Not torch.stft: replicate padding, the
 * Hann window at the start of the frame, magnitude sqrt(re^2 + im^2 + 1e-14) with re/mag and im/mag as outputs; the inverse
 * sums the bins one-sided, scales by 1/n_fft and overlap-adds without window-envelope normalisation.  n_fft 2048 / window
 * 1200 (model.yml), any hop.  Utterance u: F_u = frame_off[u+1] - frame_off[u] >= 2 frames <-> (F_u - 1) * hop samples,
 * packed at sample offset hop * (frame_off[u] - u).  mag / x / y: time-major [frames, ld >= 1025] (pad columns zeroed by
 * the transform).  The inverse needs frames * 1200 floats of workspace.
 * (With the export's own arguments - hop 300 against the generator's hop 75 - the reference's generator raises at
 * models/generator.py:414, tests/golden/onnx_stft_wiring_evidence.json, so these are pinned as a standalone module.) */
int stts_conv_stft_transform(stts_ctx* ctx, void* stream, int n_utt, const int32_t* frame_off_host, const int32_t* frame_off_dev,
                             const float* wave, int hop, float* mag, float* x, float* y, int ld);
int stts_conv_stft_inverse(stts_ctx* ctx, void* stream, int n_utt, const int32_t* frame_off_host, const int32_t* frame_off_dev,
                           const float* mag, const float* x, const float* y, int ld, int hop, float* wave_out, void* ws, size_t ws_bytes);

/* Measurement hook (bench.py roofline leg): between begin and end every conv_gemm_f32 / wn_layer_kernel launch carries a
 * HIP start/stop event pair on its own stream.  end() synchronises and returns the launch count, the summed kernel time and
 * the summed ALGORITHMIC flops (2 * rows * cout * cin * taps, un-padded sizes). */
int stts_profile_begin(void);
int stts_profile_end(void* stream, int* launches, double* total_ms, double* total_flops);
/* The same measurement per kernel, for every launch of the frame path (the bandwidth-bound kernels included): a JSON array
 * [{"kernel", "kind": "contraction"|"other", "launches", "ms", "gflop" (algorithmic), "executed_gflop" (what the matrix
 * cores execute: less for the Winograd forms), "mbytes" (algorithmic HBM bytes)}] summed over the launches
 * since stts_profile_begin.  Ends the measurement like stts_profile_end. */
int stts_profile_report(void* stream, char* json, size_t json_capacity);

/* AdaptiveGeneratorBlock.forward (HiFi-GAN MRF + Snake, models/ada_norm.py:109-120); standalone block only: the enclosing
 * UpsampleGenerator cannot be instantiated in the reference. */
int stts_op_mrf_block(stts_ctx* ctx, void* stream, const char* prefix, int n_utt, const int32_t* seg_off_host,
                      const int32_t* seg_off_dev, const float* x, int ldx, int channels, int kernel, const float* style, float* y,
                      int ldy, void* ws, size_t ws_bytes);

#ifdef STTS_TEST_OPS
/* Test surface, only in a library built with -DSTTS_TEST_OPS (tests/ and tools/; not part of the product ABI).
 * Single operators for parity tests (the same kernels the stages use): */
/* F.conv1d(stride 1, zero pad (k-1)/2*dil) on time-major rows; w is the reference layout [cout, cin, k] on the HOST. */
int stts_op_conv1d(void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* x, int ldx, int cin,
                   const float* w_host, const float* bias_host, int cout, int k, int dil, int act, float* y, int ldy, int force_tile, int precision);
/* AdaptiveDecoderBlock.forward (models/ada_norm.py:166-182) with weights named `prefix` + reference keys. */
int stts_op_adain_block(stts_ctx* ctx, void* stream, const char* prefix, int n_utt, const int32_t* seg_off_host,
                        const int32_t* seg_off_dev, const float* x, int ldx, int cin, int cout, const float* style, float* y, int ldy,
                        void* ws, size_t ws_bytes);
/* Multi-head scaled-dot-product attention on packed sequences (MultiHeadAttention.attention, models/text_encoder.py:233-277;
 * models/xut/attention.py): q [q rows, heads * kc], k / v [k rows, heads * kc], o [q rows, heads * kc]; utterance u's queries see its
 * own keys only.  band_centre (optional, [q rows] int32) + window: the pitch/energy predictor's inverted band mask.
 * kernel: 0 = the stage's own choice, 1 = one wave per four queries (attention_kernel), 2 = matrix cores (attention_mfma_kernel,
 * kc 16 / 32 / 40 / 64 / 96 / 128 / 160; heads of 64 split the keys over two wave groups from 128 keys on), 3 = matrix cores, never split. */
int stts_op_attention(void* stream, int n_utt, const int32_t* q_off_host, const int32_t* q_off_dev, const int32_t* k_off_host,
                      const int32_t* k_off_dev, const float* q, const float* k, const float* v, float* o, int heads, int kc,
                      const int32_t* band_centre, int window, int kernel);
/* Tuning aid (tools/gemm_bench.py): average time of `iters` back-to-back contraction launches on synthetic data.
 * tile: 0 = the launcher's own choice; tune bits: 128 bf16 operands, 256 fp16 operands; only in a library built with
 * -DSTTS_GEMM_TRACE: 2/4/8/16 K-loop ablations (results invalid, timing only), 64 block-timeline trace. */
int stts_bench_gemm(void* stream, int n_utt, int rows_per_utt, int cin, int cout, int k, int tile, int iters, double* avg_ms, int tune);
#endif /* STTS_TEST_OPS */

#ifdef __cplusplus
}
#endif
#endif /* STYLISH_HIP_H_ */
