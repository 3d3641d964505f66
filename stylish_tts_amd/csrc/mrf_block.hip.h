// AdaptiveGeneratorBlock.forward (HiFi-GAN MRF + Snake, models/ada_norm.py:11-120) as a standalone
// operator: the enclosing UpsampleGenerator cannot be instantiated in the reference, the block can.
// Included at the end of api.hip (it uses that file's API_BEGIN helpers).
#pragma once

extern "C" {

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
