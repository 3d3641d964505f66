// Phoneme-rate kernels: embedding, rotary position embedding, multi-head attention,
// style broadcast / masked mean, duration decoding.  Sequences are short (P <= 512 tokens, T <= ~1000 frames),
// so these kernels are written for exactness and low launch count, not for bandwidth: each is a tiny fraction of
// the step (the phoneme-rate part is ~3 % of the FLOPs).
#pragma once
#include "common.h"

namespace stts {

// x[row][:] = emb[token[row]][:] * sqrt(C)          (models/text_encoder.py:451)
__global__ void __launch_bounds__(256) embed_kernel(const long* __restrict__ tokens, const float* __restrict__ emb, int C, int n_tokens_vocab,
                                                    float scale, float* __restrict__ Y, int ldy, int n_rows, int* __restrict__ err) {
  const int nv = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n_rows * nv; i += (long)gridDim.x * 256) {
    const int r = (int)(i / nv), c4 = (int)(i % nv) * 4;
    long t = tokens[r];
    if (t < 0 || t >= n_tokens_vocab) {
      atomicOr(err, 2);
      t = 0;
    }
    const float4 e = *reinterpret_cast<const float4*>(emb + t * C + c4);
    *reinterpret_cast<float4*>(Y + (long)r * ldy + c4) = make_float4(e.x * scale, e.y * scale, e.z * scale, e.w * scale);
  }
}

// Rotary position embedding in place on the first d features of every head (RotaryPositionalEmbeddings.forward,
// models/text_encoder.py:146-168; d = int(k_channels * 0.5), text_encoder.py:194-195):
//   j <  d/2: x_j' = x_j cos(p th_j) - x_{j+d/2} sin(p th_j)
//   j >= d/2: x_j' = x_j cos(p th_{j-d/2}) + x_{j-d/2} sin(p th_{j-d/2}),   th_i = 10000^(-2i/d), p = position in the utterance
// grid (row chunks, n_utt); thread per (row, head, pair).
// col1 >= 0: the same rotation also on the column block at col1 (queries and keys of a fused q|k|v buffer in one launch).
__global__ void __launch_bounds__(256) rope_kernel(float* __restrict__ X, int ldx, int col0, int col1, int n_heads, int kc, int d,
                                                   const int* __restrict__ seg_off) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  const int half = d / 2;
  const long per = (long)n * n_heads * half;
  const long total = col1 >= 0 ? 2 * per : per;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += (long)gridDim.x * 256) {
    const long i = i0 >= per ? i0 - per : i0;
    const int cbase = i0 >= per ? col1 : col0;
    const int j = (int)(i % half);
    const int h = (int)((i / half) % n_heads);
    const int p = (int)(i / ((long)half * n_heads));
    const float theta = 1.0f / powf(10000.0f, (float)(2 * j) / (float)d);
    const float ang = (float)p * theta;
    const float cs = cosf(ang), sn = sinf(ang);
    float* x = X + (long)(lo + p) * ldx + cbase + h * kc;
    const float a = x[j], b = x[j + half];
    x[j] = a * cs - b * sn;
    x[j + half] = b * cs + a * sn;
  }
}

// Multi-head scaled-dot-product attention (MultiHeadAttention.attention, models/text_encoder.py:233-277) on packed
// sequences: queries of utterance u attend to the keys of utterance u only (the reference's padding mask fills -1e4,
// whose softmax weight underflows to exactly 0 in fp32, so dropping padded keys is identical).
// One wave per (query row, head).  Phase 1: lanes <-> keys, scores into LDS; phase 2: lanes <-> channels.
// band (cross-attention of the pitch/energy predictor): the reference builds "True = NOT allowed" but the attention fills
// -1e4 where its mask is FALSE (pitch_energy_predictor.py:194-212 vs text_encoder.py:255-262), so scores are lowered by
// 1e4 INSIDE |key - centre[query]| <= window and untouched outside.  Reproduced as is.
constexpr int kAttnMaxKeys = 1024, kAttnMaxKc = 192, kAttnQ = 4;
// A wave owns kAttnQ consecutive queries of one head: a key row (phase 1) or a value element (phase 2) is fetched once and used
// for all of them - a quarter of the K / V traffic of one query per wave, which is what bound this kernel on the 240-800-frame
// sequences of the flow-matching decoder.  Per query the arithmetic and its order are unchanged.
__global__ void __launch_bounds__(256) attention_kernel(const float* __restrict__ Q, int ldq, int qcol0, const float* __restrict__ K, int ldk,
                                                        int kcol0, const float* __restrict__ V, int ldv, int vcol0, float* __restrict__ O, int ldo,
                                                        int n_heads, int kc, const int* __restrict__ q_off, const int* __restrict__ k_off,
                                                        const int* __restrict__ band_centre, int window, float scale) {
  __shared__ float sq[4][kAttnQ][kAttnMaxKc];
  __shared__ float sp[4][kAttnQ][kAttnMaxKeys];
  const int u = blockIdx.z, h = blockIdx.y;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qlo = q_off[u], nq = q_off[u + 1] - qlo;
  const int klo = k_off[u], nk = k_off[u + 1] - klo;
  const int q0 = (blockIdx.x * 4 + w) * kAttnQ;
  if (q0 >= nq) return;  // whole wave exits together; no block-level barrier is used below
  const int nqw = min(kAttnQ, nq - q0);
  int centre[kAttnQ];
#pragma unroll
  for (int qq = 0; qq < kAttnQ; ++qq) {
    const int qi = q0 + min(qq, nqw - 1);  // (a short last group repeats its last query: computed, not stored)
    const float* q = Q + (long)(qlo + qi) * ldq + qcol0 + h * kc;
    for (int c = lane; c < kc; c += 64) sq[w][qq][c] = q[c];
    centre[qq] = band_centre ? band_centre[qlo + qi] : 0;
  }
  __builtin_amdgcn_wave_barrier();
  float mx[kAttnQ];
#pragma unroll
  for (int qq = 0; qq < kAttnQ; ++qq) mx[qq] = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const float* kr = K + (long)(klo + j) * ldk + kcol0 + h * kc;
    float s[kAttnQ];
#pragma unroll
    for (int qq = 0; qq < kAttnQ; ++qq) s[qq] = 0.f;
    for (int c = 0; c < kc; c += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + c);
#pragma unroll
      for (int qq = 0; qq < kAttnQ; ++qq) s[qq] += sq[w][qq][c] * kv.x + sq[w][qq][c + 1] * kv.y + sq[w][qq][c + 2] * kv.z + sq[w][qq][c + 3] * kv.w;
    }
#pragma unroll
    for (int qq = 0; qq < kAttnQ; ++qq) {
      float v = s[qq] * scale;
      if (band_centre && j >= centre[qq] - window && j <= centre[qq] + window) v += -1e4f;
      sp[w][qq][j] = v;
      mx[qq] = fmaxf(mx[qq], v);
    }
  }
  float inv[kAttnQ];
#pragma unroll
  for (int qq = 0; qq < kAttnQ; ++qq) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx[qq] = fmaxf(mx[qq], __shfl_xor(mx[qq], o, 64));
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) {
      const float e = expf(sp[w][qq][j] - mx[qq]);
      sp[w][qq][j] = e;
      sum += e;
    }
    inv[qq] = 1.0f / wave_sum(sum);
  }
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < kc; c += 64) {
    const float* vp = V + (long)klo * ldv + vcol0 + h * kc + c;
    float acc[kAttnQ];
#pragma unroll
    for (int qq = 0; qq < kAttnQ; ++qq) acc[qq] = 0.f;
#pragma unroll 8
    for (int j = 0; j < nk; ++j) {
      const float v = vp[(long)j * ldv];
#pragma unroll
      for (int qq = 0; qq < kAttnQ; ++qq) acc[qq] += sp[w][qq][j] * v;
    }
#pragma unroll
    for (int qq = 0; qq < kAttnQ; ++qq)
      if (qq < nqw) O[(long)(qlo + q0 + qq) * ldo + h * kc + c] = acc[qq] * inv[qq];
  }
}

// ---------------------------------------------------------------------------------------------
// The same attention on the matrix cores, for heads of KC = 16 / 32 / 40 / 64 / 96 / 128 / 160 channels (the text encoders: 8 heads x 16; the predictors' prosody
// encoders: 2 x 96 and 2 x 160; the pitch / energy cross-attention: 8 x 40; the flow-matching decoder's XUT blocks: heads of 64, 240-800 frames, models/xut/attention.py) - flash-style: a block = 128 queries of one
// (utterance, head), one wave per 32 queries, the keys / values stream through LDS in blocks of 32 (double buffered, one barrier per
// block), softmax is kept as a running maximum and sum per query (the exact softmax of attention_kernel up to fp32 rounding: ~1e-7
// relative), any number of keys.  v_mfma_f32_32x32x2_f32, fp32 throughout.  Both products are taken TRANSPOSED so that a lane owns one query:
//   S^T[key][q] = K[key][:] . Q[q][:]   A = K block from LDS, B = the wave's Q tile held in KC / 2 registers (lane: query l & 31, channels 2 s + (l >> 5))
//   O^T[ch][q] += V^T[ch][key] P^T[key][q]   A = V block from LDS, B = exp(S^T - max) - the accumulator registers of the first product, as they are:
//     register i of a lane holds key kappa(i, h) = 8 (i >> 2) + 4 h + (i & 3), and a sum over keys may visit them in any order, so k-step i
//     of the second product simply uses key kappa(i, h) for both operands.
// Per-query max / sum / rescale are per-lane scalars (+ one exchange with lane ^ 32, which holds the other 16 keys of the block).
// LDS: K rows have a stride of KC + 2 floats (a lane's read K[l & 31][2 s + h] then hits 64 distinct banks: (KC + 2) mod 64 is 2 or 34),
// V rows KC + 8 (the two half-waves read keys 4 rows apart: 4 (KC + 8) = 32 mod 64, opposite bank halves).
// ---------------------------------------------------------------------------------------------
constexpr int kAttnMfmaQ = 128;
inline bool attn_mfma_kc(int kc) { return kc == 16 || kc == 32 || kc == 40 || kc == 64 || kc == 96 || kc == 128 || kc == 160; }
// SPLIT = 2 (long sequences): two wave groups per query tile take the even / odd key blocks, each with its own running maximum / sum / accumulator,
// merged through LDS at the end - a problem of 8 x 800 frames x 4 heads is 896 wave tiles for 1 024 SIMDs, i.e. less than one wave per SIMD, and a wave
// alone on its SIMD alternates dependent MFMA chains with the softmax's vector work; with the keys split the SIMDs hold two waves that interleave.
template <int KC, int SPLIT = 1>
__global__ void __launch_bounds__(256 * SPLIT) attention_mfma_kernel(const float* __restrict__ Q, int ldq, int qcol0, const float* __restrict__ K, int ldk,
                                                             int kcol0, const float* __restrict__ V, int ldv, int vcol0, float* __restrict__ O, int ldo,
                                                             const int* __restrict__ q_off, const int* __restrict__ k_off,
                                                             const int* __restrict__ band_centre, int window, float scale) {
  static_assert(KC % 8 == 0 && KC <= 160, "head size");
  // KCP: the head size padded to whole 32-channel output tiles (the text encoders' heads of 16 and the predictors' heads of 40 / 160 channels: the
  // pad columns of the V stage are zeros, written once); row strides (floats): K rows KC + 2, V rows KCP + 8; G float4 groups per staged row
  constexpr int KCP = (KC + 31) / 32 * 32, NT = KCP / 32, KS = KC + 2, VS = KCP + 8, G = KC / 4, NE = (32 * G + 255) / 256;
  static_assert(SPLIT == 1 || SPLIT == 2, "key split");
  __shared__ float Ks[2][SPLIT][32 * KS];
  __shared__ f32x4 Vs[2][SPLIT][32 * VS / 4];
  const int u = blockIdx.z, h = blockIdx.y;
  const int qlo = q_off[u], nq = q_off[u + 1] - qlo;
  const int klo = k_off[u], nk = k_off[u + 1] - klo;
  if ((int)blockIdx.x * kAttnMfmaQ >= nq) return;
  const int grp = threadIdx.x >> 8;                // key group of this wave (SPLIT = 2: even / odd key blocks)
  const int tid = threadIdx.x & 255, w = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int q0 = blockIdx.x * kAttnMfmaQ + 32 * w;  // this wave's queries (waves beyond the utterance's end compute on its last query and store nothing)
  const int ql = min(q0 + l31, nq - 1);
  float qf[KC / 2];
  {
    const float* qp = Q + (long)(qlo + ql) * ldq + qcol0 + h * KC + lh;
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) qf[s] = qp[2 * s];
  }
  if constexpr (KCP > KC) {  // zero pad columns of both V stages (the staging below writes the real columns only)
    for (int i = tid; i < 2 * 32 * (KCP - KC) / 4; i += 256) {
      const int b = i / (32 * (KCP - KC) / 4), j = i % (32 * (KCP - KC) / 4), r = j / ((KCP - KC) / 4), g = j % ((KCP - KC) / 4);
      Vs[b][grp][r * (VS / 4) + G + g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int centre = band_centre ? band_centre[qlo + ql] : 0;
  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const int nkb = (nk + 31) / 32;
  // staging: thread -> (key r, float4 group g) of the 32 x KC block, NE of each matrix per thread
  f32x4 kreg[NE], vreg[NE];
  auto gfetch = [&](int kb) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = tid + 256 * e, r = min(idx / G, 31), g = idx % G;
      const int j = kb * 32 + r;
      const bool ok = j < nk;
      const long row = klo + min(j, nk - 1);
      const f32x4 kv = *reinterpret_cast<const f32x4*>(K + row * ldk + kcol0 + h * KC + 4 * g);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(V + row * ldv + vcol0 + h * KC + 4 * g);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      kreg[e] = ok ? kv : z;
      vreg[e] = ok ? vv : z;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = tid + 256 * e, r = idx / G, g = idx % G;
      if (32 * G % 256 != 0 && idx >= 32 * G) continue;  // (heads whose 32 x KC block is not a multiple of 256 float4s)
      float2* kd = reinterpret_cast<float2*>(&Ks[buf][grp][r * KS + 4 * g]);  // (rows are 8-byte aligned: KS is even)
      kd[0] = make_float2(kreg[e].x, kreg[e].y);
      kd[1] = make_float2(kreg[e].z, kreg[e].w);
      Vs[buf][grp][r * (VS / 4) + g] = vreg[e];
    }
  };
  // iteration it: group grp works on key block kb = SPLIT it + grp (a group past the last block stages zeros and skips the arithmetic)
  const int nit = (nkb + SPLIT - 1) / SPLIT;
  gfetch(min(grp, nkb - 1));
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1, kb = SPLIT * it + grp;
    lstore(buf);
    if (it + 1 < nit) gfetch(min(kb + SPLIT, nkb - 1));
    __syncthreads();
    if (kb >= nkb) continue;
    const float* kl = Ks[buf][grp] + l31 * KS + lh;
    const float* vl = reinterpret_cast<const float*>(Vs[buf][grp]) + l31;
    // ---- S^T = K Q^T
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kl[2 * s], qf[s], sacc, 0, 0, 0);
    // ---- scores of query l31 against keys kappa(i, lh): scale, band, validity; running softmax
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = kb * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
      float v = sacc[i] * scale;
      if (band_centre && j >= centre - window && j <= centre + window) v += -1e4f;
      if (j >= nk) v = -INFINITY;
      sacc[i] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(m, mx);
    const float resc = expf(m - mnew);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sacc[i] = expf(sacc[i] - mnew);
      psum += sacc[i];
    }
    psum += __shfl_xor(psum, 32, 64);
    lsum = lsum * resc + psum;
    m = mnew;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t][i] *= resc;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = 8 * (i >> 2) + 4 * lh + (i & 3);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[key * VS + 32 * t], sacc[i], o[t], 0, 0, 0);
      }
  }
  if constexpr (SPLIT == 2) {
    // merge the two key groups: group 1 parks (o, m, l) in the stage memory, group 0 combines (softmax over the union of the keys)
    __syncthreads();  // every wave is done with the stages
    float* xch = reinterpret_cast<float*>(Vs) + (size_t)w * (NT * 16 + 2) * 64 + lane;  // [wave][NT * 16 + 2][lane]
    static_assert(sizeof(Vs) >= 4 * (NT * 16 + 2) * 64 * sizeof(float), "exchange area");
    if (grp == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) xch[(t * 16 + i) * 64] = o[t][i];
      xch[(NT * 16) * 64] = m;
      xch[(NT * 16 + 1) * 64] = lsum;
    }
    __syncthreads();
    if (grp == 1) return;
    const float m2 = xch[(NT * 16) * 64], l2 = xch[(NT * 16 + 1) * 64];
    const float mnew = fmaxf(m, m2);
    const float s1 = expf(m - mnew), s2 = expf(m2 - mnew);  // (group 1 without a key block: m2 = -inf, s2 = 0, its o and l are zeros)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t][i] = o[t][i] * s1 + xch[(t * 16 + i) * 64] * s2;
    lsum = lsum * s1 + l2 * s2;
  }
  if (q0 + l31 < nq) {
    const float inv = 1.0f / lsum;
    float* op = O + (long)(qlo + q0 + l31) * ldo + h * KC + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        if (32 * t + 8 * i4 + 4 * lh < KC) {  // (KC is a multiple of 8: a lane's four channels are all real or all pad)
          const f32x4 v = {o[t][4 * i4] * inv, o[t][4 * i4 + 1] * inv, o[t][4 * i4 + 2] * inv, o[t][4 * i4 + 3] * inv};
          *reinterpret_cast<f32x4*>(op + 32 * t + 8 * i4) = v;  // channels 32 t + 8 i4 + 4 lh + (0..3)
        }
      }
  }
}

// Y[row][col0 + c] = style[u][c] for every row of utterance u (ProsodyEncoder concat, models/prosody_encoder.py:67-69,80)
__global__ void __launch_bounds__(256) broadcast_style_kernel(const float* __restrict__ style, int ld_style, int C, float* __restrict__ Y, int ldy,
                                                              int col0, const int* __restrict__ seg_off) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n * C; i += (long)gridDim.x * 256) {
    const int r = (int)(i / C), c = (int)(i % C);
    Y[(long)(lo + r) * ldy + col0 + c] = style[(long)u * ld_style + c];
  }
}

// style[u][c] = mean over the utterance's rows of X[:, c]     (TextStyleEncoder masked mean, text_style_encoder.py:24-26)
__global__ void __launch_bounds__(256) mean_rows_kernel(const float* __restrict__ X, int ldx, int C, const int* __restrict__ seg_off,
                                                        float* __restrict__ out, int ld_out) {
  const int u = blockIdx.x;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int r = lo; r < hi; ++r) s += X[(long)r * ldx + c];
    out[(long)u * ld_out + c] = s / (float)(hi - lo);
  }
}

// DurationProcessor.prediction_to_duration (train/utils.py:468-474): softmax . class table summed, round (half to
// even), clamp >= 1; argmax -> table; hard if hard < 7 else soft.  One thread per token; 16 classes.
__constant__ float kClassToDur[16] = {1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 18, 22, 27, 32, 38, 46};
__global__ void __launch_bounds__(256) duration_decode_kernel(const float* __restrict__ logits, int ld, int n_classes, int n_rows,
                                                              int* __restrict__ dur) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const float* l = logits + (long)r * ld;
  float mx = l[0];
  int am = 0;
  for (int k = 1; k < n_classes; ++k)
    if (l[k] > mx) {
      mx = l[k];
      am = k;
    }
  float den = 0.f;
  float e[16];
  for (int k = 0; k < n_classes; ++k) {
    e[k] = expf(l[k] - mx);
    den += e[k];
  }
  float soft = 0.f;
  for (int k = 0; k < n_classes; ++k) soft += (e[k] / den) * kClassToDur[k];
  soft = fmaxf(rintf(soft), 1.0f);
  const float hard = kClassToDur[am];
  dur[r] = (int)(hard < 7.0f ? hard : soft);
}

// DurationProcessor.duration_to_alignment (train/utils.py:476-489): 0/1 matrix [P, T], T = sum(dur); one thread per frame
__global__ void __launch_bounds__(256) alignment_matrix_kernel(const int* __restrict__ dur, int P, int T, float* __restrict__ out) {
  __shared__ int cum[1025];
  if (threadIdx.x == 0) {
    int a = 0;
    for (int i = 0; i < P; ++i) {
      a += dur[i];
      cum[i] = a;
    }
  }
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)P * T; i += (long)gridDim.x * 256) {
    const int p = (int)(i / T), t = (int)(i % T);
    const int lo = p == 0 ? 0 : cum[p - 1];
    out[i] = (t >= lo && t < cum[p]) ? 1.0f : 0.0f;
  }
}

// token-local index of the token each frame belongs to (build_monotonic_band_mask's tau = alignment.argmax(dim=1),
// pitch_energy_predictor.py:201)
__global__ void __launch_bounds__(256) local_token_kernel(const int* __restrict__ src_row, const int* __restrict__ frm_off,
                                                          const int* __restrict__ tok_off, int* __restrict__ centre) {
  const int u = blockIdx.y;
  const int lo = frm_off[u], hi = frm_off[u + 1];
  for (int f = lo + blockIdx.x * 256 + threadIdx.x; f < hi; f += gridDim.x * 256) centre[f] = src_row[f] - tok_off[u];
}

}  // namespace stts
