// Sanitizer driver (CPU): loads a weight dump, finalizes every component and walks the stage entry points over the shapes of
// BASELINE's configs with the HIP runtime stubbed out (hip_stub.cpp).  Exercised under -fsanitize=address,undefined:
// weight-norm folding, row / Winograd / fragment packing, style tables, the workspace closed forms against what the stages
// really carve (a stage returns "workspace too small" if the bound is wrong), and the launch planning of every contraction.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/stylish_hip.h"

extern "C" long stts_stub_launch_count();

#define CK(expr)                                                                  \
  do {                                                                            \
    if ((expr) != 0) {                                                            \
      fprintf(stderr, "FAILED %s:%d %s: %s\n", __FILE__, __LINE__, #expr, stts_last_error()); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static std::vector<float> buf(size_t n) { return std::vector<float>(n ? n : 1, 0.5f); }

static int frame_case(stts_ctx* c, const std::vector<int>& lens, const char* what) {
  const int n = (int)lens.size();
  std::vector<int32_t> off(n + 1, 0);
  int ml = 0;
  for (int i = 0; i < n; ++i) {
    off[i + 1] = off[i] + lens[i];
    ml = lens[i] > ml ? lens[i] : ml;
  }
  const long R = off[n];
  const size_t wsb = stts_frame_workspace_bytes(c, R, n, ml);
  struct Ws {  // untouched (tens of GB for the large configs: the host code only computes pointers into it)
    char* p;
    explicit Ws(size_t n) : p((char*)malloc(n)) {}
    ~Ws() { free(p); }
    char* data() { return p; }
  } ws(wsb);
  if (!ws.p) {
    fprintf(stderr, "cannot reserve %zu bytes of address space\n", wsb);
    return 1;
  }
  auto asr = buf(R * 128), pitch = buf(R), energy = buf(R), style = buf((size_t)n * 64), pn = buf(R * 128), sn = buf(R * 75), ph = buf(1), audio = buf(R * 75);
  auto x = buf(R * 512), mel = buf(R * 512), hs = buf(R * 1088), hp = buf(R * 1088);  // 1088: the spectrum row stride every operand mode accepts (fp32 needs >= 1056, the 16-bit modes 1088)
  const long before = stts_stub_launch_count();
  CK(stts_frame_path(c, nullptr, n, off.data(), off.data(), asr.data(), 128, pitch.data(), energy.data(), style.data(), pn.data(), sn.data(), ph.data(), 0,
                     audio.data(), ws.data(), wsb, 0));
  const long fused = stts_stub_launch_count() - before;
  // the same call with capacity segments (STTS_SEG_CAPACITY: the host offsets are upper bounds, the device ones - the same array here - real)
  CK(stts_frame_path(c, nullptr, n, off.data(), off.data(), asr.data(), 128, pitch.data(), energy.data(), style.data(), pn.data(), sn.data(), ph.data(), 0,
                     audio.data(), ws.data(), wsb, STTS_SEG_CAPACITY));
  // the staged entry points carve the same workspace stage by stage
  CK(stts_decoder_forward(c, nullptr, n, off.data(), off.data(), asr.data(), 128, pitch.data(), energy.data(), style.data(), x.data(), 512, ws.data(), wsb));
  CK(stts_prior_flow_forward(c, nullptr, n, off.data(), off.data(), x.data(), 512, style.data(), pn.data(), mel.data(), 512, nullptr, nullptr, ws.data(), wsb));
  CK(stts_harmonic_stft(c, nullptr, n, off.data(), off.data(), pitch.data(), sn.data(), ph.data(), 1, nullptr, hs.data(), hp.data(), 1088, ws.data(), wsb));
  CK(stts_vocoder_forward(c, nullptr, n, off.data(), off.data(), mel.data(), 512, style.data(), hs.data(), hp.data(), 1088, audio.data(), nullptr, nullptr, 0,
                          ws.data(), wsb));
  printf("  %-44s rows %8ld  workspace %8.1f MB  %ld launches per frame-path call\n", what, R, wsb / 1048576.0, fused);
  return 0;
}

static int phoneme_case(stts_ctx* c, const std::vector<int>& toks, const std::vector<int>& frames, const char* what) {
  const int n = (int)toks.size();
  std::vector<int32_t> to(n + 1, 0), fo(n + 1, 0), fo4(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    to[i + 1] = to[i] + toks[i];
    fo[i + 1] = fo[i] + frames[i];
    fo4[i + 1] = fo4[i] + 4 * frames[i];
  }
  const long P = to[n], T = fo[n];
  const size_t wsb = stts_phoneme_workspace_bytes(c, P, T, n);
  std::vector<char> ws(wsb);
  std::vector<int64_t> tokens(P, 3);
  std::vector<int32_t> dur(P, 1);
  auto mu = buf(P * 256), xh = buf(P * 128), sty = buf((size_t)n * 64), logits = buf(P * 16), f0 = buf(T), en = buf(T), enc4 = buf(4 * T * 256), up = buf(4 * T);
  std::vector<int32_t> dur_out(P), idx(4 * T + 1);
  for (int which = 0; which < 3; ++which) {
    CK(stts_text_encoder_forward(c, nullptr, which, n, to.data(), to.data(), tokens.data(), mu.data(), 256, xh.data(), ws.data(), wsb));
    CK(stts_text_style_forward(c, nullptr, which, n, to.data(), to.data(), mu.data(), 256, sty.data(), ws.data(), wsb));
  }
  CK(stts_duration_forward(c, nullptr, n, to.data(), to.data(), tokens.data(), logits.data(), dur_out.data(), nullptr, nullptr, nullptr, ws.data(), wsb));
  CK(stts_pitch_energy_forward(c, nullptr, n, to.data(), to.data(), fo.data(), fo.data(), dur.data(), mu.data(), 256, sty.data(), f0.data(), en.data(), nullptr,
                               nullptr, ws.data(), wsb, 0));
  CK(stts_pitch_energy_forward(c, nullptr, n, to.data(), to.data(), fo.data(), fo.data(), dur.data(), mu.data(), 256, sty.data(), f0.data(), en.data(), nullptr,
                               nullptr, ws.data(), wsb, STTS_SEG_CAPACITY));
  {
    std::vector<int32_t> offT(n + 1), offT4(n + 1), need(n);
    CK(stts_frame_offsets(c, nullptr, n, to.data(), dur.data(), fo.data(), offT.data(), offT4.data(), need.data()));
  }
  CK(stts_length_regulate(c, nullptr, n, dur.data(), to.data(), fo4.data(), 4 * T, 4, mu.data(), 256, 128, enc4.data(), 128, idx.data()));
  CK(stts_upsample4(c, nullptr, n, fo.data(), fo.data(), fo4.data(), f0.data(), up.data()));
  printf("  %-44s tokens %6ld frames %7ld  workspace %8.1f MB\n", what, P, T, wsb / 1048576.0);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: asan_driver <weights.bin>\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  stts_model_dims d;
  if (fread(&d, sizeof(d), 1, f) != 1) return 2;
  for (int prec = 0; prec < 3; prec += 2) {  // fp32 and fp16 operand modes (16-bit weight copies are packed too)
    stts_ctx* c = nullptr;
    CK(stts_ctx_create(&d, 0, &c));
    CK(stts_set_precision(c, prec));
    fseek(f, sizeof(d), SEEK_SET);
    int n_tensors = 0;
    for (;;) {
      int32_t name_len = 0, ndim = 0;
      if (fread(&name_len, 4, 1, f) != 1) break;
      std::string name(name_len, '\0');
      if (fread(&name[0], 1, name_len, f) != (size_t)name_len || fread(&ndim, 4, 1, f) != 1) return 2;
      int64_t shape[4] = {1, 1, 1, 1}, count = 1;
      for (int i = 0; i < ndim; ++i) {
        if (fread(&shape[i], 8, 1, f) != 1) return 2;
        count *= shape[i];
      }
      std::vector<float> v(count);
      if (fread(v.data(), 4, count, f) != (size_t)count) return 2;
      CK(stts_load_weight(c, name.c_str(), v.data(), shape, ndim));
      ++n_tensors;
    }
    CK(stts_finalize_weights(c, 255));
    printf("precision %d: %d tensors loaded and packed\n", prec, n_tensors);
    if (frame_case(c, std::vector<int>(8, 960), "cfg2: 8 x 3 s")) return 1;
    if (frame_case(c, {960}, "B = 1 x 3 s")) return 1;
    if (frame_case(c, {40, 131, 76, 14, 15, 16, 17, 33}, "short ragged utterances")) return 1;
    if (prec == 0) {  // (ASan shadow-poisons the 28 GB workspace reservation: once is enough)
      std::vector<int> lens;  // cfg4: 256 utterances of 0.25 - 10 s (same generator as tests/test_hip_full_size.py would give a similar spread)
      unsigned s = 4;
      for (int i = 0; i < 256; ++i) {
        s = s * 1664525u + 1013904223u;
        lens.push_back(4 * (20 + (int)((s >> 8) % 781)));
      }
      if (frame_case(c, lens, "cfg4: 256 utterances of 0.25-10 s")) return 1;
    }
    if (frame_case(c, std::vector<int>(64, 3200), "cfg5 per GPU: 64 x 10 s")) return 1;
    if (phoneme_case(c, std::vector<int>(64, 50), std::vector<int>(64, 240), "cfg3: 64 x 50 tokens")) return 1;
    if (phoneme_case(c, {510, 2, 160}, {1020, 4, 800}, "token-count extremes")) return 1;
    stts_ctx_destroy(c);
  }
  fclose(f);
  printf("asan driver: all cases ran, %ld stubbed launches\n", stts_stub_launch_count());
  return 0;
}
