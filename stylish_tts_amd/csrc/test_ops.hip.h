// Test operators (single layers behind the C-ABI, used by tests/ to compare one kernel at a time with the oracle) and the
// contraction micro-benchmark of tools/gemm_bench.py.  None of this is on the product path: api.hip includes this file only
// when built with -DSTTS_TEST_OPS (the default of __graft_entry__.compile, because tests/ need the operators;
// STTS_PRODUCT_ONLY=1 builds the library without them).  Uses api.hip's API_BEGIN / SEG_CHECK helpers.
#pragma once

extern "C" {

// ------------------------------------------------------------------------------------------------ test operators
int stts_op_conv1d(void* stream, int n_utt, const int32_t* seg_off_host, const int32_t* seg_off_dev, const float* x, int ldx, int cin,
                   const float* w_host, const float* bias_host, int cout, int k, int dil, int act, float* y, int ldy, int force_tile, int precision) {
  API_BEGIN
  hipStream_t st = (hipStream_t)stream;
  STTS_CHECK(ldx % 32 == 0 && ldx >= cin, "op_conv1d: ldx must be a multiple of 32 covering cin");
  // precision 0: fp32, split-fp32 contraction (PREC_X3, the default form); 3: fp32 on the f32 matrix cores (STTS_PREC_F32_NATIVE)
  STTS_CHECK(precision >= 0 && precision <= 3, "precision must be STTS_PREC_F32, _BF16, _F16 or _F32_NATIVE");
  stts_ctx tmp;  // only for allocation bookkeeping
  tmp.prec = precision == 3 ? 0 : precision;
  tmp.allow_x3 = precision != 3;
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
    STTS_CHECK(dil == 1 && (precision == 0 || precision == 3), "op_conv1d: the Winograd form is fp32, dilation 1");
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
    STTS_CHECK(precision == 1 || precision == 2, "op_conv1d: 16-bit activation rows need a 16-bit operand mode");
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

int stts_op_attention(void* stream, int n_utt, const int32_t* q_off_host, const int32_t* q_off_dev, const int32_t* k_off_host,
                      const int32_t* k_off_dev, const float* q, const float* k, const float* v, float* o, int heads, int kc,
                      const int32_t* band_centre, int window, int kernel) {
  API_BEGIN
  STTS_CHECK(n_utt > 0 && q_off_host && q_off_dev && k_off_host && k_off_dev && q && k && v && o && heads > 0, "null / empty argument");
  Seg sq{n_utt, q_off_host, q_off_dev}, sk{n_utt, k_off_host, k_off_dev};
  const int ld = heads * kc;
  return run_attention((hipStream_t)stream, sq, sk, q, ld, 0, k, ld, 0, v, ld, 0, o, ld, heads, kc, band_centre, window, kernel);
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
  std::vector<float> hw((size_t)npad * k * kc);
  {
    std::vector<float> t((size_t)std::max<long>(R * kc, (long)npad * k * kc));
    uint32_t s = 12345;
    for (auto& v : t) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    STTS_HIP(hipMemcpy(X, t.data(), R * kc * sizeof(float), hipMemcpyHostToDevice));
    STTS_HIP(hipMemcpy(W, t.data(), (size_t)npad * k * kc * sizeof(float), hipMemcpyHostToDevice));
    std::copy(t.begin(), t.begin() + hw.size(), hw.begin());
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
  if (tune & 2048) {  // bit 11: split-fp32 contraction (the three bf16 planes of W)
    std::vector<unsigned short> h3(hw.size() * 3);
    for (size_t i = 0; i < hw.size(); ++i) split3_host(hw[i], &h3[i], &h3[hw.size() + i], &h3[2 * hw.size() + i]);
    STTS_HIP(hipMalloc(&W16, h3.size() * 2));
    STTS_HIP(hipMemcpy(W16, h3.data(), h3.size() * 2, hipMemcpyHostToDevice));
    pc.W16 = W16;
    pc.w16_plane = (long)hw.size();
  }
  Seg s{n_utt, h.data(), so};
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, X, kc, 0, pc);
  a.N = cout; a.bias = B; a.Y = Y; a.ldy = ldy; a.tune = tune & 63;
  if ((tune & 2048) && !x3_enabled()) return stts::fail("split-fp32 contractions are switched off (STTS_NO_X3)");
  unsigned short* X16 = nullptr;
  if ((tune & 1024) && (tune & 384)) {  // bit 10: 16-bit activation rows (X rounded once, outside the timed launches)
    STTS_HIP(hipMalloc(&X16, R * kc * sizeof(unsigned short)));
    launch_cast_rows(st, pc.prec, X, kc, kc, X16, kc, R);
    a.seg[0].X = reinterpret_cast<const float*>(X16);
    a.x16 = 1;
  }
  if ((tune & 4096) && (tune & 2048)) {  // bit 12: pre-split activation planes (split once, outside the timed launches)
    STTS_HIP(hipMalloc(&X16, 3 * R * kc * sizeof(unsigned short)));
    launch_split_rows(st, X, kc, kc, X16, kc, R * kc, R);
    a.seg[0].X = reinterpret_cast<const float*>(X16);
    a.seg[0].x_plane = R * kc;
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

