// 1-D Winograd / Toom-Cook fast convolution F(6, r) over the TIME axis for the wide 'same' convolutions of the frame path
// (stride 1, dilation 1, r = 3 or 7 taps).  Six consecutive outputs of a channel need 6 + r - 1 inputs and, computed as
//   y = A^T [ (G g) (.) (B^T d) ],
// only n = 6 + r - 1 products per input channel instead of 6 r:  r = 7 -> 12 instead of 42 (x 0.286), r = 3 -> 8 of 18.
// The sum over input channels of each of the n component products is an ordinary contraction, so the work is
//   1. winograd_input_kernel : X [rows, cin] -> X' [n][groups, cin]   (B^T d per group of 6 output rows; zeros outside
//                              the utterance = the conv's zero padding)
//   2. conv_gemm_f32         : n independent 1-tap contractions [groups, cin] x [cin, cout], one weight plane G_j g per
//                              component (the launcher's per-utterance-weights mode: component = "utterance")
//   3. winograd_output_kernel: M [n][groups, cout] -> Y [rows, cout] = A^T M (+ bias, activation, residual, scale)
// Matrices come from the evaluation points {0, +-1, +-2, +-1/2, (+-3/2, +-3/4,) inf} (Cook-Toom), built in double on the
// host and validated against direct correlation at start-up; the fp32 error of F(6,7) with these points is ~3e-6 of the
// output scale (direct: 3e-7; F(4,7) with {.., +-3} was 5e-6), far inside the path's 2e-4 / 1e-3 parity bars.
// Group g of utterance u lives at row goff[u] + g of every component plane, goff = prefix sum of ceil(len / 6)
// (wino_setup_kernel): the planes are packed exactly, so the contraction's row tiles carry no padding beyond the last one.
#pragma once
#include <array>
#include <cmath>
#include <vector>

#include "gemm.hip.h"

namespace stts {

constexpr int kWinoM = 6;       // outputs per group
constexpr int kWinoMaxN = 12;   // components of F(6,7)

struct WinoMats {
  int r = 0, n = 0;
  float Bt[kWinoMaxN][kWinoMaxN];  // input transform  [component][input row of the group]
  float At[kWinoM][kWinoMaxN];     // output transform [output row of the group][component]
  double G[kWinoMaxN][8];          // weight transform [component][tap] (host only)
};

// polynomial helpers (coefficients low -> high)
inline std::vector<double> wino_polymul(const std::vector<double>& a, const std::vector<double>& b) {
  std::vector<double> c(a.size() + b.size() - 1, 0.0);
  for (size_t i = 0; i < a.size(); ++i)
    for (size_t j = 0; j < b.size(); ++j) c[i + j] += a[i] * b[j];
  return c;
}

// F(6, r): returns false for an unsupported r or if the self-check against direct correlation fails
inline bool wino_matrices(int r, WinoMats* out) {
  std::vector<double> pts;
  if (r == 3) pts = {0, 1, -1, 2, -2, 0.5, -0.5};
  else if (r == 7) pts = {0, 1, -1, 2, -2, 0.5, -0.5, 1.5, -1.5, 0.75, -0.75};
  else return false;
  const int m = kWinoM, n = m + r - 1;
  std::vector<std::vector<double>> A(n, std::vector<double>(m, 0.0)), G(n, std::vector<double>(r, 0.0)), C(n, std::vector<double>(n, 0.0));
  for (int j = 0; j < n - 1; ++j) {
    for (int i = 0; i < m; ++i) A[j][i] = std::pow(pts[j], i);
    for (int k = 0; k < r; ++k) G[j][k] = std::pow(pts[j], k);
    std::vector<double> num{1.0};
    double den = 1.0;
    for (int l = 0; l < n - 1; ++l)
      if (l != j) {
        num = wino_polymul(num, {-pts[l], 1.0});
        den *= pts[j] - pts[l];
      }
    for (size_t q = 0; q < num.size(); ++q) C[q][j] = num[q] / den;  // Lagrange basis polynomial of point j
  }
  A[n - 1][m - 1] = 1.0;
  G[n - 1][r - 1] = 1.0;
  std::vector<double> M{1.0};
  for (int l = 0; l < n - 1; ++l) M = wino_polymul(M, {-pts[l], 1.0});
  for (int q = 0; q < n; ++q) C[q][n - 1] = M[q];
  out->r = r;
  out->n = n;
  for (int j = 0; j < kWinoMaxN; ++j)
    for (int q = 0; q < kWinoMaxN; ++q) out->Bt[j][q] = (j < n && q < n) ? (float)C[q][j] : 0.0f;
  for (int i = 0; i < kWinoM; ++i)
    for (int j = 0; j < kWinoMaxN; ++j) out->At[i][j] = j < n ? (float)A[j][i] : 0.0f;
  for (int j = 0; j < kWinoMaxN; ++j)
    for (int k = 0; k < 8; ++k) out->G[j][k] = (j < n && k < r) ? G[j][k] : 0.0;
  // self-check: y_i = sum_k g_k d_{i+k}
  std::vector<double> g(r), d(n);
  for (int k = 0; k < r; ++k) g[k] = std::sin(1.0 + k);
  for (int q = 0; q < n; ++q) d[q] = std::cos(0.3 + 1.7 * q);
  for (int i = 0; i < m; ++i) {
    double ref = 0, y = 0;
    for (int k = 0; k < r; ++k) ref += g[k] * d[i + k];
    for (int j = 0; j < n; ++j) {
      double gg = 0, dd = 0;
      for (int k = 0; k < r; ++k) gg += G[j][k] * g[k];
      for (int q = 0; q < n; ++q) dd += C[q][j] * d[q];
      y += A[j][i] * gg * dd;
    }
    if (std::fabs(y - ref) > 1e-9 * (1.0 + std::fabs(ref))) return false;
  }
  return true;
}

struct WinoIn {
  float Bt[kWinoMaxN][kWinoMaxN];
};
struct WinoOut {
  float At[kWinoM][kWinoMaxN];
};

// upper bound of the rows of a component plane for a batch of `rows` frames in n_utt utterances (scratch sizing)
inline long wino_plane_rows(long rows, int n_utt) { return rows / kWinoM + n_utt + 1; }

// one block: goff[u] = sum_{v<u} ceil(len_v / kWinoM) (n_utt + 1 entries) and the plane offsets segp[j] = j * goff[n_utt], j <= n
__global__ void __launch_bounds__(64) wino_setup_kernel(const int* __restrict__ seg_off, int n_utt, int n, int* __restrict__ goff, int* __restrict__ segp) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int u0 = 0; u0 < n_utt; u0 += 64) {
    const int u = u0 + lane;
    const int g = u < n_utt ? (seg_off[u + 1] - seg_off[u] + kWinoM - 1) / kWinoM : 0;
    int incl = g;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    if (u < n_utt) goff[u] = base + incl - g;
    base += __shfl(incl, 63, 64);
  }
  if (lane == 0) goff[n_utt] = base;
  for (int j = lane; j <= n; j += 64) segp[j] = j * base;
}

// grid (ceil(max groups / 4), ceil(C4 / 64), n_utt), block (64, 4): thread = 4 channels of one group
// SPLIT: the planes are written as the three bf16 planes of the exact fp32 split (gemm.hip.h, PREC_X3; `xplane` elements apart), so the contraction
// stages them without a conversion: 6 instead of 4 bytes per element here, no split arithmetic in the contraction's K loop
template <int N, bool SPLIT = false>
__global__ void __launch_bounds__(256) winograd_input_kernel(const float* __restrict__ X, int ldx, int C, const int* __restrict__ seg_off, int pad,
                                                             const WinoIn t, float* __restrict__ Xp, int ldp, const int* __restrict__ goff,
                                                             const float* __restrict__ aff, int ld_aff, long xplane = 0) {
  const int u = blockIdx.z;
  const int lo = seg_off[u], len = seg_off[u + 1] - lo;
  const int groups = (len + kWinoM - 1) / kWinoM;
  const int g = blockIdx.x * 4 + threadIdx.y;
  const int c4 = (blockIdx.y * 64 + threadIdx.x) * 4;
  if (g >= groups || c4 >= ldp) return;
  // aff (optional): the conv's input is lrelu_0.2(x * scale + shift) with per-(utterance, channel) scale / shift rows
  // (AdaIN folded into this transform: adain_affine_kernel).  Rows outside the utterance stay zero (the conv pads the
  // activated tensor), scale 0 marks a pad channel.
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (aff && c4 < C) {
    sc = *reinterpret_cast<const f32x4*>(aff + ((long)u * 2) * ld_aff + c4);
    sh = *reinterpret_cast<const f32x4*>(aff + ((long)u * 2 + 1) * ld_aff + c4);
  }
  f32x4 d[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int row = g * kWinoM - pad + q;
    const bool in = row >= 0 && row < len && c4 < C;
    d[q] = in ? *reinterpret_cast<const f32x4*>(X + (long)(lo + row) * ldx + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (aff && in) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = (sc[e] == 0.0f ? 0.0f : d[q][e] * sc[e]) + sh[e];
        d[q][e] = y >= 0.0f ? y : 0.2f * y;
      }
    }
    if (c4 + 3 >= C) {  // the channel count need not be a multiple of 4: pad columns of X may hold anything
#pragma unroll
      for (int e = 1; e < 4; ++e)
        if (c4 + e >= C) d[q][e] = 0.0f;
    }
  }
  const long prow = goff[u] + g, plane_rows = goff[gridDim.z];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < N; ++q) v += t.Bt[j][q] * d[q];
    if constexpr (SPLIT) {
      u32x2 p0, p1, p2;
      split3_bf16(v, p0, p1, p2);
      unsigned short* o = reinterpret_cast<unsigned short*>(Xp) + ((long)j * plane_rows + prow) * ldp + c4;
      *reinterpret_cast<u32x2*>(o) = p0;
      *reinterpret_cast<u32x2*>(o + xplane) = p1;
      *reinterpret_cast<u32x2*>(o + 2 * xplane) = p2;
    } else
    *reinterpret_cast<f32x4*>(Xp + ((long)j * plane_rows + prow) * ldp + c4) = v;
  }
}

// Y[row][n] = (act(sum_j At[i][j] M_j[group][n] + bias[n]) [+ R[row][n]]) * alpha for the rows of each group inside its utterance
// STATS: the block also leaves the AdaIN statistics of its 4 groups = kWinoStatChunk rows of Y (chunk blockIdx.x of the utterance) behind, in
// adain_partial_kernel's layout with chunk_rows = kWinoStatChunk: stat[((u * stat_nchunk + chunk) * 2 + {0: mean, 1: M2}) * ld_stat + n] - the
// consuming AdaIN (models/ada_norm.py:129-139) then needs no pass over Y of its own.
constexpr int kWinoStatChunk = 4 * kWinoM;
template <int N, bool STATS>
__global__ void __launch_bounds__(256) winograd_output_kernel(const float* __restrict__ Mp, int ldm, const int* __restrict__ goff, const int* __restrict__ seg_off,
                                                              const WinoOut t, const float* __restrict__ bias, int act, const float* __restrict__ R,
                                                              int ldr, float alpha, float* __restrict__ Y, int ldy, int Nout,
                                                              float* __restrict__ stat, int ld_stat, int stat_nchunk) {
  const int u = blockIdx.z;
  const int lo = seg_off[u], len = seg_off[u + 1] - lo;
  const int groups = (len + kWinoM - 1) / kWinoM;
  const int g = blockIdx.x * 4 + threadIdx.y;
  const int n4 = (blockIdx.y * 64 + threadIdx.x) * 4;
  const bool live = g < groups && n4 < Nout;
  if (!STATS && !live) return;
  if (STATS && (int)blockIdx.x * kWinoStatChunk >= len) return;  // (uniform: no row of this chunk exists)
  f32x4 o[kWinoM];
#pragma unroll
  for (int i = 0; i < kWinoM; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const long prow = goff[u] + g, plane_rows = goff[gridDim.z];
    f32x4 m[N];
#pragma unroll
    for (int j = 0; j < N; ++j) m[j] = *reinterpret_cast<const f32x4*>(Mp + ((long)j * plane_rows + prow) * ldm + n4);
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = *reinterpret_cast<const f32x4*>(bias + n4);
#pragma unroll
    for (int i = 0; i < kWinoM; ++i) {
      const int row = g * kWinoM + i;
      if (row < len) {
        f32x4 v = b;
#pragma unroll
        for (int j = 0; j < N; ++j) v += t.At[i][j] * m[j];
        float* y = Y + (long)(lo + row) * ldy + n4;
        if (n4 + 4 <= Nout) {  // whole quad (every layer of the model: cout % 4 == 0; leading dimensions are multiples of 4, checked by run_winograd): 16-byte residual load and store
          f32x4 e;
#pragma unroll
          for (int c = 0; c < 4; ++c) e[c] = act_apply(v[c], act);
          if (R) e += *reinterpret_cast<const f32x4*>(R + (long)(lo + row) * ldr + n4);
          e *= alpha;
          *reinterpret_cast<f32x4*>(y) = e;
          o[i] = e;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (n4 + c < Nout) {
              float e = act_apply(v[c], act);
              if (R) e += R[(long)(lo + row) * ldr + n4 + c];
              e *= alpha;
              y[c] = e;
              o[i][c] = e;
            }
          }
        }
      }
    }
  }
  if (STATS) {
    // a wave = one group: every thread reduces its own (up to six) rows first - mean and centred squares of the group, no exchange - and the block's
    // four groups are merged by Chan's update in group order through LDS: one barrier, nothing kept in registers across it
    __shared__ f32x4 red[2][4][64];
    const int gn = min(kWinoM, max(0, len - g * kWinoM));  // rows of this wave's group inside the utterance
    f32x4 gmean = {0.f, 0.f, 0.f, 0.f}, gm2 = {0.f, 0.f, 0.f, 0.f};
    if (gn > 0) {
#pragma unroll
      for (int i = 0; i < kWinoM; ++i) gmean += o[i];  // rows beyond the utterance hold zeros
      gmean *= 1.0f / (float)gn;
#pragma unroll
      for (int i = 0; i < kWinoM; ++i) {
        if (i < gn) {
          const f32x4 dv = o[i] - gmean;
          gm2 += dv * dv;
        }
      }
    }
    red[0][threadIdx.y][threadIdx.x] = gmean;
    red[1][threadIdx.y][threadIdx.x] = gm2;
    __syncthreads();
    if (threadIdx.y == 0 && n4 < Nout) {
      const int base = (int)blockIdx.x * kWinoStatChunk;
      const int nrows = min(kWinoStatChunk, len - base);
      f32x4 mean = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) mean += (float)min(kWinoM, max(0, len - base - w * kWinoM)) * red[0][w][threadIdx.x];
      mean *= 1.0f / (float)nrows;
      f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const f32x4 dv = red[0][w][threadIdx.x] - mean;
        m2 += red[1][w][threadIdx.x] + (float)min(kWinoM, max(0, len - base - w * kWinoM)) * dv * dv;
      }
      float* p = stat + ((long)(u * stat_nchunk + blockIdx.x) * 2) * ld_stat + n4;
      *reinterpret_cast<f32x4*>(p) = mean;
      *reinterpret_cast<f32x4*>(p + ld_stat) = m2;
    }
  }
}

}  // namespace stts
