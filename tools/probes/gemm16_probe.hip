// Probe / unit harness for conv_gemm16_kernel (csrc/gemm16.hip.h): correctness against a naive device reference (double
// accumulation over the same 16-bit operands) and throughput, (the register-staged 256 x 256 tile it replaces is timed by tools/gemm_bench.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/bin/gemm16_probe tools/probes/gemm16_probe.hip   (~25 s)
//   gemm16_probe [n_utt rows_per_utt cin cout k [check]]     (no arguments: the layer shapes of the 16-bit frame path at B = 64)
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define STTS_GEMM_NO_LAUNCHER
#include "../../stylish_tts_amd/csrc/gemm16.hip.h"

using namespace stts;

static unsigned short h_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__global__ void ref_kernel(const unsigned short* X, int ldx, const unsigned short* W, int kc, int ntaps, int pad, const int* seg_off, int n_utt,
                           const float* bias, const float* R, int ldr, float alpha, int N, float* Y, int ldy, long rows, const unsigned short* X2, int kc2,
                           const unsigned short* W2) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * N) return;
  const int row = (int)(idx / N), n = (int)(idx % N);
  int u = 0;
  while (u + 1 < n_utt && row >= seg_off[u + 1]) ++u;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  double s = 0;
  for (int t = 0; t < ntaps; ++t) {
    const int r = row + t - pad;
    if (r < lo || r >= hi) continue;
    for (int c = 0; c < kc; ++c) {
      const float x = __uint_as_float((unsigned)X[(long)r * ldx + c] << 16), w = __uint_as_float((unsigned)W[((long)n * ntaps + t) * kc + c] << 16);
      s += (double)x * w;
    }
  }
  if (X2)  // second K segment: a 1 x 1 conv of another input (the learned shortcut of a decoder block)
    for (int c = 0; c < kc2; ++c) s += (double)__uint_as_float((unsigned)X2[(long)row * kc2 + c] << 16) * __uint_as_float((unsigned)W2[(long)n * kc2 + c] << 16);
  float v = (float)s + bias[n];
  if (R) v += R[(long)row * ldr + n];
  Y[(long)row * ldy + n] = v * alpha;
}

// epi bits: 1 fp32 Y, 2 16-bit Y, 4 residual, 8 GRN sums of squares, 16 SiLU
static int run(int n_utt, int rows_per_utt, int cin, int cout, int k, bool check, bool ragged, int epi = 5, const char* name = "", int cin2 = 0) {
  const int kc = round_up(cin, 64), npad = round_up(cout, 256), ldy = round_up(cout, 32);
  std::vector<int> h(n_utt + 1, 0);
  for (int i = 0; i < n_utt; ++i) h[i + 1] = h[i] + (ragged ? std::max(1, rows_per_utt - 37 * (i % 5) - (i == 1 ? rows_per_utt / 2 : 0)) : rows_per_utt);
  const long R = h[n_utt];
  unsigned short *X, *W;
  float *Y, *Yref, *B, *Res;
  int* so;
  STTS_HIP(hipMalloc(&X, (R * kc + 64) * 2));
  STTS_HIP(hipMalloc(&W, (size_t)npad * k * kc * 2));
  STTS_HIP(hipMalloc(&Y, R * ldy * 4));
  STTS_HIP(hipMalloc(&Yref, R * ldy * 4));
  STTS_HIP(hipMalloc(&Res, R * ldy * 4));
  STTS_HIP(hipMalloc(&B, npad * 4));
  STTS_HIP(hipMalloc(&so, (n_utt + 1) * 4));
  STTS_HIP(hipMemcpy(so, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  {
    std::vector<unsigned short> t((size_t)std::max<long>(R * kc, (long)npad * k * kc));
    uint32_t s = 12345;
    for (auto& v : t) { s = s * 1664525u + 1013904223u; v = h_bf16(((s >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    STTS_HIP(hipMemcpy(X, t.data(), R * kc * 2, hipMemcpyHostToDevice));
    for (auto& v : t) { s = s * 1664525u + 1013904223u; v = h_bf16((((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.05f); }
    // rows >= cout and channels >= cin of the packed weight are zero (as pack_rows leaves them)
    for (int n = 0; n < npad; ++n)
      for (int tt = 0; tt < k; ++tt)
        for (int c = 0; c < kc; ++c)
          if (n >= cout || c >= cin) t[((size_t)n * k + tt) * kc + c] = 0;
    STTS_HIP(hipMemcpy(W, t.data(), (size_t)npad * k * kc * 2, hipMemcpyHostToDevice));
    std::vector<float> b(npad), rr((size_t)R * ldy);
    for (auto& v : b) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (auto& v : rr) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    STTS_HIP(hipMemcpy(B, b.data(), npad * 4, hipMemcpyHostToDevice));
    STTS_HIP(hipMemcpy(Res, rr.data(), rr.size() * 4, hipMemcpyHostToDevice));
  }
  unsigned short *X2 = nullptr, *W2 = nullptr;
  const int kc2 = round_up(std::max(cin2, 1), 64);
  if (cin2) {
    STTS_HIP(hipMalloc(&X2, (R * kc2 + 64) * 2));
    STTS_HIP(hipMalloc(&W2, (size_t)npad * kc2 * 2));
    std::vector<unsigned short> t((size_t)std::max<long>(R * kc2, (long)npad * kc2));
    uint32_t s2 = 999;
    for (auto& v : t) { s2 = s2 * 1664525u + 1013904223u; v = h_bf16(((s2 >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    STTS_HIP(hipMemcpy(X2, t.data(), R * kc2 * 2, hipMemcpyHostToDevice));
    for (auto& v : t) { s2 = s2 * 1664525u + 1013904223u; v = h_bf16((((s2 >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.05f); }
    for (int n = 0; n < npad; ++n)
      for (int c = 0; c < kc2; ++c)
        if (n >= cout || c >= cin2) t[(size_t)n * kc2 + c] = 0;
    STTS_HIP(hipMemcpy(W2, t.data(), (size_t)npad * kc2 * 2, hipMemcpyHostToDevice));
  }
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.seg_off = so; a.seg_host = h.data(); a.n_utt = n_utt; a.rows_total = (int)R; a.zeros = zero_page();
  a.nseg = 1; a.prec = PREC_BF16; a.wrows = cout;
  a.seg[0].X = reinterpret_cast<const float*>(X); a.seg[0].W = reinterpret_cast<const float*>(W); a.seg[0].W16 = W; a.seg[0].ldx = kc; a.seg[0].kc = kc;
  a.seg[0].ntaps = k; a.seg[0].dil = 1; a.seg[0].pad = (k - 1) / 2; a.seg[0].kreal = cin;
  if (cin2) {
    a.nseg = 2;
    a.seg[1].X = reinterpret_cast<const float*>(X2); a.seg[1].W = reinterpret_cast<const float*>(W2); a.seg[1].W16 = W2; a.seg[1].ldx = kc2; a.seg[1].kc = kc2;
    a.seg[1].ntaps = 1; a.seg[1].dil = 1; a.seg[1].pad = 0; a.seg[1].kreal = cin2;
  }
  a.x16 = 1;
#ifdef STTS_GEMM_TRACE
  long long* dbg = nullptr;
  STTS_HIP(hipMalloc(&dbg, 8 * 8 * 256));
  STTS_HIP(hipMemset(dbg, 0, 8 * 8 * 256));
  a.dbg = dbg;
#endif
  a.N = cout; a.bias = B; a.alpha = 0.5f;
  unsigned short* Y16 = nullptr;
  float* part = nullptr;
  if (epi & 1) { a.Y = Y; a.ldy = ldy; }
  if (epi & 2) { STTS_HIP(hipMalloc(&Y16, R * ldy * 2)); a.Y16 = Y16; a.ldy16 = ldy; }
  if (epi & 4) { a.R = Res; a.ldr = ldy; }
  if (epi & 8) { a.ss_stride = ceil_div(rows_per_utt, 128) * 4; a.ld_ss = ldy; STTS_HIP(hipMalloc(&part, (size_t)n_utt * a.ss_stride * ldy * 4)); a.sumsq_part = part; }
  if (epi & 16) a.act = ACT_SILU;
  hipStream_t st = nullptr;
  int rc = 0;
  if (check) {
    STTS_HIP(hipMemset(Y, 0xff, R * ldy * 4));
    STTS_TRY(launch_conv_gemm16<0>(st, a, npad, n_utt));
    const long work = R * cout;
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, X, kc, W, kc, k, (k - 1) / 2, so, n_utt, B, (epi & 4) ? Res : nullptr, ldy, 0.5f, cout, Yref, ldy, R, X2, kc2, W2);
    STTS_HIP(hipDeviceSynchronize());
    std::vector<float> y((size_t)R * ldy), yr((size_t)R * ldy);
    STTS_HIP(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
    STTS_HIP(hipMemcpy(yr.data(), Yref, yr.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, mx = 0;
    long bad = 0;
    for (long r = 0; r < R; ++r)
      for (int n = 0; n < cout; ++n) {
        const double d = fabs((double)y[r * ldy + n] - yr[r * ldy + n]);
        if (!(d == d) || d > 1e-3) { if (bad < 5) fprintf(stderr, "  mismatch row %ld col %d: %g vs %g\n", r, n, y[r * ldy + n], yr[r * ldy + n]); ++bad; }
        err = std::max(err, d);
        mx = std::max(mx, fabs((double)yr[r * ldy + n]));
      }
    printf("check n_utt=%d rows=%ld cin=%d cout=%d k=%d%s: max abs err %.3e (max |ref| %.3f), %ld bad\n", n_utt, R, cin, cout, k, ragged ? " ragged" : "", err, mx, bad);
    rc = bad != 0;
  } else {
    hipEvent_t e0, e1;
    STTS_HIP(hipEventCreate(&e0));
    STTS_HIP(hipEventCreate(&e1));
    const double flops = 2.0 * R * cout * ((double)cin * k + cin2);
    const int abl[] = {0, 1, 2, 4, 8, 16, 3, 7, 32};
    const char* abl_name[] = {"as built", "no DMA", "no ds_read", "no MFMA", "no stores", "no barriers", "no DMA+reads", "epilogue only", "no setprio"};
    static const int n_abl = getenv("ABL") ? (atoi(getenv("ABL")) == 2 ? 9 : 2) : 1;  // ABL=1: as built vs no setprio; ABL=2: all of them
    auto go = [&](int which) {
      switch (which) {
        case 1: return launch_conv_gemm16<1>(st, a, npad, n_utt);
        case 2: return launch_conv_gemm16<2>(st, a, npad, n_utt);
        case 4: return launch_conv_gemm16<4>(st, a, npad, n_utt);
        case 8: return launch_conv_gemm16<8>(st, a, npad, n_utt);
        case 16: return launch_conv_gemm16<16>(st, a, npad, n_utt);
        case 3: return launch_conv_gemm16<3>(st, a, npad, n_utt);
        case 7: return launch_conv_gemm16<7>(st, a, npad, n_utt);
        case 32: return launch_conv_gemm16<32>(st, a, npad, n_utt);
        case 64: return launch_conv_gemm16<64>(st, a, npad, n_utt);
        default: return launch_conv_gemm16<0>(st, a, npad, n_utt);
      }
    };
    // interleaved rounds in one process (guide 5.4 rule 24): per variant the minimum and the median of the rounds
    const int rounds = 5, iters = 8;
    std::vector<std::vector<float>> t(n_abl);
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < n_abl; ++v) {
        const int which = n_abl == 2 ? (v == 0 ? 0 : 64) : abl[v];
        STTS_TRY(go(which));
        STTS_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) STTS_TRY(go(which));
        STTS_HIP(hipEventRecord(e1, st));
        STTS_HIP(hipEventSynchronize(e1));
        float ms = 0;
        STTS_HIP(hipEventElapsedTime(&ms, e0, e1));
        t[v].push_back(ms / iters);
      }
#ifdef STTS_GEMM_TRACE
    {
      STTS_TRY(go(0));
      STTS_HIP(hipStreamSynchronize(st));
      std::vector<long long> h(8 * 256);
      STTS_HIP(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      double s[6] = {};
      for (int b = 0; b < 256; ++b)
        for (int i = 0; i < 6; ++i) s[i] += (double)h[8 * b + i] / 256.0;
      printf("%-11s block timeline (averages over 256 blocks, us per block): tiles %.2f | lookup + prologue issue %.1f | prologue wait %.1f | K loop %.1f | loop-end drain %.1f | epilogue %.1f\n", name,
             s[0], s[1] * 0.01, s[2] * 0.01, s[3] * 0.01, s[4] * 0.01, s[5] * 0.01);
    }
#endif
    for (int v = 0; v < n_abl; ++v) {
      std::sort(t[v].begin(), t[v].end());
      const float mn = t[v][0], md = t[v][rounds / 2];
      printf("%-11s %-13s rows=%ld cin=%d cout=%d k=%d epi=%d: min %7.1f us %7.1f TFLOP/s | median %7.1f us %7.1f TFLOP/s\n", name, (n_abl == 2 && v == 1) ? "nt stores" : abl_name[v], R, cin, cout, k,
             epi, 1e3 * mn, flops / (mn * 1e-3) * 1e-12, 1e3 * md, flops / (md * 1e-3) * 1e-12);
    }
  }
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(Yref); (void)hipFree(Res); (void)hipFree(B); (void)hipFree(so);
  if (Y16) (void)hipFree(Y16);
  if (X2) (void)hipFree(X2);
  if (W2) (void)hipFree(W2);
  if (part) (void)hipFree(part);
  return rc;
}

int main(int argc, char** argv) {
  if (argc >= 6) {
    const int rc = run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 && atoi(argv[6]) != 0, argc > 7 && atoi(argv[7]) != 0);
    if (rc) fprintf(stderr, "error: %s\n", last_error().c_str());
    return rc;
  }
  int bad = 0;
  // correctness: small and ragged shapes, k = 1 / 3 / 7, one and several K tiles
  bad += run(1, 256, 64, 256, 1, true, false);
  bad += run(2, 300, 128, 256, 3, true, false);
  bad += run(5, 700, 192, 512, 7, true, true);
  bad += run(3, 960, 512, 1000, 1, true, true);
  bad += run(2, 500, 128, 512, 3, true, true, 1);
  bad += run(3, 400, 128, 256, 3, true, true, 1, "", 192);   // two K segments (conv k3 + 1x1 of another input)
  bad += run(7, 90, 64, 256, 7, true, true, 5);             // utterances shorter than a tile, k = 7 reaching across both ends
  if (bad) { fprintf(stderr, "FAILED: %s\n", last_error().c_str()); return 1; }
  // throughput: the layers of the 16-bit frame path at B = 64 x 3 s (61 440 rows)
  struct { const char* name; int cin, cout, k, epi, cin2; } shapes[] = {
      {"out conv", 768, 1024, 7, 1, 0}, {"prior conv", 1088, 256, 7, 2, 0}, {"pwconv1", 512, 1536, 1, 2 | 8 | 16, 0}, {"pwconv2", 1536, 512, 1, 1 | 4, 0},
      {"dec conv1", 640, 512, 3, 1, 0}, {"dec conv2+sc", 512, 512, 3, 1 | 2, 640}, {"projector", 1024, 512, 1, 1, 0}};
  const int nb = getenv("NB") ? atoi(getenv("NB")) : 64;
  for (auto& s : shapes) bad += run(nb, 960, s.cin, s.cout, s.k, false, false, s.epi, s.name, s.cin2);
  return bad;
}
