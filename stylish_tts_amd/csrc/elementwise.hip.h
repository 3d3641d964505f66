// HBM-bound kernels of the frame-rate path: style projections, instance-norm statistics and AdaIN apply,
// row LayerNorm / AdaLN, depthwise conv, GRN finalisation, length regulator, layout transposes.
// All activations are time-major packed rows (see gemm.hip.h); every kernel reads/writes float4 along the
// channel axis so a wave touches 1 KiB contiguous per instruction.
#pragma once
#include "common.h"
#include "gemm.hip.h"

namespace stts {

// ---------------------------------------------------------------------------------------------
// style projections: out[u][j] = b[j] + sum_k W[j][k] * s[u][k]   (K = style_dim = 64)
// Every AdaptiveInstance.fc / AdaptiveLayerNorm.fc (models/ada_norm.py:133,191) and every WN.cond_layer
// (models/flow.py:38-40,67-68) of a stage is concatenated into one [J, 64] table: one launch per stage.
// One wave per output row j; lanes = k.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) style_fc_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                        const float* __restrict__ s, float* __restrict__ out, int J, int K,
                                                        int n_utt, int lds_s, int ld_out) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (j >= J) return;
  float w0 = lane < K ? W[(long)j * K + lane] : 0.f;
  float w1 = (lane + 64) < K ? W[(long)j * K + lane + 64] : 0.f;
  const float bj = b[j];
  for (int u = 0; u < n_utt; ++u) {
    float v = lane < K ? w0 * s[(long)u * lds_s + lane] : 0.f;
    if (lane + 64 < K) v += w1 * s[(long)u * lds_s + lane + 64];
    v = wave_sum(v);
    if (lane == 0) out[(long)u * ld_out + j] = v + bj;
  }
}

// The same projection for larger batches: lane = utterance (64 per pass), every lane runs the K-long dot product of its own
// style row with the wave's weight row (broadcast from LDS) - no cross-lane reduction per utterance (the kernel above spends a
// 6-step shuffle reduction per (row, utterance): 51 us at 64 utterances, this one ~10).
__global__ void __launch_bounds__(256) style_fc_batch_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ s,
                                                              float* __restrict__ out, int J, int K, int n_utt, int lds_s, int ld_out) {
  __shared__ float wrow[4][128];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wv;
  if (j < J)
    for (int k = lane; k < K; k += 64) wrow[wv][k] = W[(long)j * K + k];
  __syncthreads();
  if (j >= J) return;
  const float bj = b[j];
  for (int u0 = 0; u0 < n_utt; u0 += 64) {
    const int u = u0 + lane;
    if (u >= n_utt) break;
    const float* sp = s + (long)u * lds_s;
    float acc = 0.f;
    for (int k = 0; k < K; k += 4) {
      const float4 x = *reinterpret_cast<const float4*>(sp + k);
      acc += x.x * wrow[wv][k] + x.y * wrow[wv][k + 1] + x.z * wrow[wv][k + 2] + x.w * wrow[wv][k + 3];
    }
    out[(long)u * ld_out + j] = acc + bj;
  }
}

// ---------------------------------------------------------------------------------------------
// AdaIN = (1+gamma) * InstanceNorm(x) + beta, per (utterance, channel) over time, eps 1e-5, biased variance
// (models/ada_norm.py:129-139).  Two launches:
//  1. adain_partial_kernel: per (utterance, 128-row chunk, 32-channel group) the chunk mean and the centred sum of
//     squares, from registers (each row read once) - grid (C/32, chunks, n_utt) fills the chip even at B = 1.
//  2. adain_apply_kernel: each thread owns 4 channels; it merges the chunk statistics (Chan's parallel-variance
//     update, exact and order-fixed => deterministic), folds them with the style affine into scale/shift and
//     streams its rows:  y = act(x*scale + shift).   Pad columns C..ldy are written as zeros.
// part layout: [(u * nchunk + chunk) * 2 + {0: mean, 1: M2}][ldp]
// ---------------------------------------------------------------------------------------------
constexpr int kStatChunk = 128;

// Four consecutive outputs of a row, either fp32 or rounded (nearest even) to the operand type of the 16-bit modes: the
// producers of contraction inputs write 16-bit rows directly (conv_gemm_f32<..., X16>); idx counts ELEMENTS of the row buffer.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4(float* Y, long idx, float a, float b, float c, float d, int prec16) {
  if (prec16 == PREC_BF16) {
    const bf16x4_t o = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    *reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(Y) + idx) = o;
  } else if (prec16 == PREC_F16) {
    const f16x4_t o = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    *reinterpret_cast<f16x4_t*>(reinterpret_cast<unsigned short*>(Y) + idx) = o;
  } else {
    *reinterpret_cast<float4*>(Y + idx) = make_float4(a, b, c, d);
  }
}
// four consecutive 16-bit operands -> fp32
__device__ __forceinline__ float4 load4_16(const unsigned short* X, long idx, int prec16) {
  if (prec16 == PREC_BF16) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(X + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  const f16x4_t v = *reinterpret_cast<const f16x4_t*>(X + idx);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// chunk_rows (<= kStatChunk): rows per chunk - 128, or 24 where the statistics have to line up with those a Winograd output transform leaves
// (winograd_output_kernel<N, true>: four groups of six rows per block)
__global__ void __launch_bounds__(256) adain_partial_kernel(const float* __restrict__ X, int ldx, int C, const int* __restrict__ seg_off,
                                                            float* __restrict__ part, int ldp, int nchunk, int chunk_rows) {
  __shared__ float red[8][33];
  const int u = blockIdx.z, ch = blockIdx.y;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), cl = threadIdx.x & 31;
  const int rl = threadIdx.x >> 5;
  const int lo = seg_off[u] + ch * chunk_rows, hi = min(seg_off[u + 1], lo + chunk_rows);
  if (lo >= hi) return;
  const bool ok = c < C;
  float v[kStatChunk / 8];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < kStatChunk / 8; ++i) {
    const int r = lo + rl + 8 * i;
    v[i] = (ok && r < hi) ? X[(long)r * ldx + c] : 0.f;
    acc += v[i];
  }
  red[rl][cl] = acc;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) mean += red[i][cl];
  mean /= (float)(hi - lo);
  __syncthreads();
  acc = 0.f;
#pragma unroll
  for (int i = 0; i < kStatChunk / 8; ++i) {
    const int r = lo + rl + 8 * i;
    if (r < hi) {
      const float d = v[i] - mean;
      acc += d * d;
    }
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && ok) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m2 += red[i][cl];
    float* p = part + ((long)(u * nchunk + ch) * 2) * ldp;
    p[c] = mean;
    p[ldp + c] = m2;
  }
}

// A split-K contraction's reduce pass, handed to the consumer (GemmArgs::defer): Y[row][c] = (act(sum_k partial[k][row][c] + bias[c]) [+ R]) * alpha
// for c < N - splitk_reduce_kernel's arithmetic and order.
struct SplitSrc {
  const float* partial;  // null: nothing pending, the rows are read from X
  int ksplit, slice_rows, ld_part, N;
  const float* bias;
  int act;
  const float* R;
  int ldr, rcol0;
  float alpha;
};
// adain_partial_kernel whose input rows are (for channels < src.N) still the partial sums of a split-K contraction: the block finishes them, WRITES
// them to X and takes the statistics from the values in flight - the reduce pass and the statistics pass of a small-batch AdaIN in one launch.
__global__ void __launch_bounds__(256) adain_partial_reduce_kernel(float* __restrict__ X, int ldx, int C, const int* __restrict__ seg_off,
                                                                   float* __restrict__ part, int ldp, int nchunk, const SplitSrc src) {
  __shared__ float red[8][33];
  const int u = blockIdx.z, ch = blockIdx.y;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), cl = threadIdx.x & 31;
  const int rl = threadIdx.x >> 5;
  const int lo = seg_off[u] + ch * kStatChunk, hi = min(seg_off[u + 1], lo + kStatChunk);
  if (lo >= hi) return;
  const bool ok = c < C, pend = ok && c < src.N;
  const float bv = (pend && src.bias) ? src.bias[c] : 0.f;
  float v[kStatChunk / 8];
#pragma unroll
  for (int i = 0; i < kStatChunk / 8; ++i) v[i] = 0.f;
  if (pend) {
    // slice by slice, the 16 rows of a slice as one batch of independent loads (a per-row loop over the slices is 16 x ksplit dependent round trips)
    for (int k = 0; k < src.ksplit; ++k) {
      const float* pk = src.partial + ((long)k * src.slice_rows) * src.ld_part + c;
#pragma unroll
      for (int i = 0; i < kStatChunk / 8; ++i) {
        const int r = lo + rl + 8 * i;
        v[i] += pk[(long)min(r, hi - 1) * src.ld_part];  // (0 + p0 = p0: the sums are splitk_reduce_kernel's, slice 0 first)
      }
    }
#pragma unroll
    for (int i = 0; i < kStatChunk / 8; ++i) {
      const int r = lo + rl + 8 * i;
      float x = act_apply(v[i] + bv, src.act);
      if (src.R) x += src.R[(long)min(r, hi - 1) * src.ldr + src.rcol0 + c];
      x *= src.alpha;
      if (r < hi) X[(long)r * ldx + c] = x;
      v[i] = r < hi ? x : 0.f;
    }
  } else if (ok) {
#pragma unroll
    for (int i = 0; i < kStatChunk / 8; ++i) {
      const int r = lo + rl + 8 * i;
      const float x = X[(long)min(r, hi - 1) * ldx + c];
      v[i] = r < hi ? x : 0.f;
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < kStatChunk / 8; ++i) acc += v[i];
  red[rl][cl] = acc;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) mean += red[i][cl];
  mean /= (float)(hi - lo);
  __syncthreads();
  acc = 0.f;
#pragma unroll
  for (int i = 0; i < kStatChunk / 8; ++i) {
    const int r = lo + rl + 8 * i;
    if (r < hi) {
      const float d = v[i] - mean;
      acc += d * d;
    }
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && ok) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m2 += red[i][cl];
    float* p = part + ((long)(u * nchunk + ch) * 2) * ldp;
    p[c] = mean;
    p[ldp + c] = m2;
  }
}

// merge of the chunk statistics of channel c of utterance u (len rows) with the style affine:
//   (1 + gamma) * (x - mean) * rstd + beta = x * scale + shift
__device__ __forceinline__ void adain_scale_shift(const float* __restrict__ part, int ldp, int nchunk, int u, int c, int len,
                                                  const float* __restrict__ gb, int ld_gb, int gcol0, int C, float eps, float* scale, float* shift,
                                                  int chunk_rows = kStatChunk) {
  const float n = (float)len;
  const int nch = (len + chunk_rows - 1) / chunk_rows;
  float mean = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    const float nc = (float)min(chunk_rows, len - ch * chunk_rows);
    mean += nc * part[((long)(u * nchunk + ch) * 2) * ldp + c];
  }
  mean /= n;
  float m2 = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    const float nc = (float)min(chunk_rows, len - ch * chunk_rows);
    const float* p = part + ((long)(u * nchunk + ch) * 2) * ldp;
    const float d = p[c] - mean;
    m2 += p[ldp + c] + nc * d * d;
  }
  const float rstd = rsqrtf(m2 / n + eps);
  const float g = gb[(long)u * ld_gb + gcol0 + c], be = gb[(long)u * ld_gb + gcol0 + C + c];
  *scale = rstd * (1.0f + g);
  *shift = be - mean * (*scale);
}

// AdaIN as an input affine of the following contraction (conv_gemm_f32<..., XAFF>): aff[u][0][c] = scale, aff[u][1][c] =
// shift for c < C, zeros in the pad columns (so pad columns of X contribute nothing whatever they hold).
// grid (ceil(ld_aff / 64), n_utt), block 64.
__global__ void __launch_bounds__(64) adain_affine_kernel(const float* __restrict__ part, int ldp, int nchunk, const int* __restrict__ seg_off,
                                                          const float* __restrict__ gb, int ld_gb, int gcol0, int C, float eps,
                                                          float* __restrict__ aff, int ld_aff, int chunk_rows) {
  const int u = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= ld_aff) return;
  float sc = 0.f, sh = 0.f;
  if (c < C) adain_scale_shift(part, ldp, nchunk, u, c, seg_off[u + 1] - seg_off[u], gb, ld_gb, gcol0, C, eps, &sc, &sh, chunk_rows);
  aff[((long)u * 2) * ld_aff + c] = sc;
  aff[((long)u * 2 + 1) * ld_aff + c] = sh;
}

// The same table with the merge spread over 16 chunk lanes per channel (block 256 = 16 channels x 16 lanes, grid (ceil(ld_aff / 16), n_utt)): a lane
// owns chunks lane, lane + 16, ...; its loads of a phase are independent, so a phase costs a few L2 round trips instead of one per chunk (the
// single-thread merge walks 2 x nchunk dependent-latency loads: 7.5 us at 8 chunks, and the Winograd-path statistics have 40 chunks of 24 rows at 3 s).
// Chan's update in a fixed order (lane partial sums in lane order): deterministic.
__global__ void __launch_bounds__(256) adain_affine_lanes_kernel(const float* __restrict__ part, int ldp, int nchunk, const int* __restrict__ seg_off,
                                                                 const float* __restrict__ gb, int ld_gb, int gcol0, int C, float eps,
                                                                 float* __restrict__ aff, int ld_aff, int chunk_rows) {
  __shared__ float red[16][17];
  const int u = blockIdx.y, cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int len = seg_off[u + 1] - seg_off[u];
  const int nch = (len + chunk_rows - 1) / chunk_rows;
  const bool ok = c < C;
  const float* p0 = part + ((long)u * nchunk * 2) * ldp + (ok ? c : 0);
  float acc = 0.f;
#pragma unroll 4
  for (int ch = lane; ch < nch; ch += 16) {
    const float nc = (float)min(chunk_rows, len - ch * chunk_rows);
    acc += nc * p0[(long)ch * 2 * ldp];
  }
  red[lane][cl] = acc;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mean += red[i][cl];
  mean /= (float)len;
  __syncthreads();
  acc = 0.f;
#pragma unroll 4
  for (int ch = lane; ch < nch; ch += 16) {
    const float nc = (float)min(chunk_rows, len - ch * chunk_rows);
    const float* p = p0 + (long)ch * 2 * ldp;
    const float d = p[0] - mean;
    acc += p[ldp] + nc * d * d;
  }
  red[lane][cl] = acc;
  __syncthreads();
  if (lane != 0 || c >= ld_aff) return;
  float sc = 0.f, sh = 0.f;
  if (ok && len > 0) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m2 += red[i][cl];
    const float rstd = rsqrtf(m2 / (float)len + eps);
    const float g = gb[(long)u * ld_gb + gcol0 + c], be = gb[(long)u * ld_gb + gcol0 + C + c];
    sc = rstd * (1.0f + g);
    sh = be - mean * sc;
  }
  aff[((long)u * 2) * ld_aff + c] = sc;
  aff[((long)u * 2 + 1) * ld_aff + c] = sh;
}

// grid (ceil(ldy/64), row blocks of rb, n_utt); block 256 = 16 float4 columns x 16 row lanes.  rb = 64, or 256 for large batches (the
// per-block merge of the chunk statistics is then paid once per 64 KB instead of once per 16 KB of rows).
// snake: y = v + sin^2(alpha*v)/alpha (AdaptiveGeneratorBlock, models/ada_norm.py:114,117) when alpha != null.
__global__ void __launch_bounds__(256) adain_apply_kernel(const float* __restrict__ X, int ldx, float* __restrict__ Y, int ldy, int C,
                                                          const int* __restrict__ seg_off, const float* __restrict__ part, int ldp, int nchunk,
                                                          const float* __restrict__ gb, int ld_gb, int gcol0, float eps, int act,
                                                          const float* __restrict__ alpha, int out16, int rb) {
  // out16: 0 = fp32 rows; PREC_BF16 / PREC_F16 = Y is a 16-bit row buffer (ldy in elements) read by the next contraction
  const int u = blockIdx.z;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  const int r0 = lo + blockIdx.y * rb;
  if (r0 >= hi) return;
  // the block's 64 channels: four chunk lanes per channel merge the chunk statistics (Chan's update, lane partial sums in lane order: deterministic;
  // a lane's loads of a phase are independent - one thread per channel walked 2 x nchunk dependent loads before the block's first row moved:
  // 25 chunks at 10 s), result shared through LDS
  __shared__ float s_sc[64], s_sh[64], s_al[64], s_red[4][64];
  {
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int len = hi - lo;
    const int nch = (len + kStatChunk - 1) / kStatChunk;
    const bool ok = c < C;
    const float* p0 = part + ((long)u * nchunk * 2) * ldp + (ok ? c : 0);
    float acc = 0.f;
#pragma unroll 4
    for (int ch = lane; ch < nch; ch += 4) acc += (float)min(kStatChunk, len - ch * kStatChunk) * p0[(long)ch * 2 * ldp];
    s_red[lane][cl] = acc;
    __syncthreads();
    const float mean = (s_red[0][cl] + s_red[1][cl] + s_red[2][cl] + s_red[3][cl]) / (float)len;
    __syncthreads();
    acc = 0.f;
#pragma unroll 4
    for (int ch = lane; ch < nch; ch += 4) {
      const float* p = p0 + (long)ch * 2 * ldp;
      const float d = p[0] - mean;
      acc += p[ldp] + (float)min(kStatChunk, len - ch * kStatChunk) * d * d;
    }
    s_red[lane][cl] = acc;
    __syncthreads();
    if (lane == 0) {
      float scv = 0.f, shv = 0.f, alv = 1.f;
      if (ok) {
        const float m2 = s_red[0][cl] + s_red[1][cl] + s_red[2][cl] + s_red[3][cl];
        const float rstd = rsqrtf(m2 / (float)len + eps);
        const float g = gb[(long)u * ld_gb + gcol0 + c], be = gb[(long)u * ld_gb + gcol0 + C + c];
        scv = rstd * (1.0f + g);
        shv = be - mean * scv;
        if (alpha) alv = alpha[c];
      }
      s_sc[cl] = scv;
      s_sh[cl] = shv;
      s_al[cl] = alv;
    }
  }
  __syncthreads();
  const int cl = (threadIdx.x & 15) * 4;
  const int c4 = blockIdx.x * 64 + cl;
  if (c4 >= ldy) return;
  float sc[4], sh[4], al[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sc[k] = s_sc[cl + k];
    sh[k] = s_sh[cl + k];
    al[k] = s_al[cl + k];
  }
  const int rend = min(hi, r0 + rb);
  for (int r = r0 + (threadIdx.x >> 4); r < rend; r += 16) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C) {
      const float4 x = *reinterpret_cast<const float4*>(X + (long)r * ldx + c4);
      float xv[4] = {x.x, x.y, x.z, x.w}, ov[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = 0.f;
        if (c4 + k < C) {
          v = xv[k] * sc[k] + sh[k];
          if (alpha) {
            const float sn = sinf(al[k] * v);
            v = v + sn * sn / al[k];
          } else {
            v = act_apply(v, act);
          }
        }
        ov[k] = v;
      }
      o = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
    store4(Y, (long)r * ldy + c4, o.x, o.y, o.z, o.w, out16);
  }
}

// ---------------------------------------------------------------------------------------------
// Row LayerNorm over channels (one wave per row), optionally style-adaptive, up to two outputs:
//   yk = LN(x)*(g) + b ; adaptive: g = 1+gamma_u[c], b = beta_u[c]  (AdaptiveLayerNorm, ada_norm.py:193-201)
//   static: g = gamma[c], b = beta[c]                               (text_encoder.LayerNorm :24-33, nn.LayerNorm)
// Optional post ops: ReLU (prenet), row mask multiply.  C <= 2048, C % 4 == 0.
// ---------------------------------------------------------------------------------------------
struct LnOut {
  float* Y;
  int ldy, ycol0;
  const float* g;  // adaptive: style table (row u), gamma at gcol0 + c, beta at gcol0 + C + c ; static: gamma[c]
  const float* b;  // static beta (ignored when adaptive)
  int ld_g, gcol0;
  int prec16;      // 0: Y holds fp32 rows; PREC_BF16 / PREC_F16: Y is a 16-bit row buffer (ldy / ycol0 in elements)
};
// LnIn (optional): the row is not read from X but finished from the partial sums of a split-K contraction,
//   x = (act(sum_k partial[k][row] + bias) [+ R[row]]) * alpha  in slice order (what splitk_reduce_kernel computes),
// so a contraction followed by a LayerNorm needs no separate reduce pass.
struct LnIn {
  const float* partial;  // null: read X
  int ksplit, slice_rows, ld_part;
  const float* bias;
  int act;
  const float* R;
  int ldr, rcol0;
  float alpha;
};
__global__ void __launch_bounds__(256) row_layernorm_kernel(const float* __restrict__ X, int ldx, int C, int n_rows,
                                                            const int* __restrict__ row_utt, float eps, int adaptive, int nout,
                                                            LnOut o0, LnOut o1, int act, LnIn in, const int* __restrict__ n_rows_dev) {
  // n_rows_dev (optional): the batch's real row count on the device - n_rows is then an upper bound (capacity segments) and the rows
  // beyond it, whose row_utt entries nobody wrote, are skipped
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows || (n_rows_dev && row >= *n_rows_dev)) return;
  const float* x = X + (long)row * ldx;
  float4 v[8];
  const int nv = C / 4;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nv) {
      if (in.partial) {
        f32x4 t = *reinterpret_cast<const f32x4*>(in.partial + (long)row * in.ld_part + q * 4);
        for (int k = 1; k < in.ksplit; ++k) t += *reinterpret_cast<const f32x4*>(in.partial + ((long)k * in.slice_rows + row) * in.ld_part + q * 4);
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          e[c] = act_apply(t[c] + (in.bias ? in.bias[q * 4 + c] : 0.0f), in.act);
          if (in.R) e[c] += in.R[(long)row * in.ldr + in.rcol0 + q * 4 + c];
          e[c] *= in.alpha;
        }
        v[i] = make_float4(e[0], e[1], e[2], e[3]);
      } else {
        v[i] = *reinterpret_cast<const float4*>(x + q * 4);
      }
    }
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = lane + i * 64;
    if (q < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
  const int u = adaptive ? row_utt[row] : 0;
  for (int k = 0; k < nout; ++k) {
    const LnOut& o = k == 0 ? o0 : o1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = lane + i * 64;
      if (q < nv) {
        float4 g, b;
        if (adaptive) {
          const float* gp = o.g + (long)u * o.ld_g + o.gcol0 + q * 4;
          g = *reinterpret_cast<const float4*>(gp);
          b = *reinterpret_cast<const float4*>(gp + C);
          g.x += 1.f; g.y += 1.f; g.z += 1.f; g.w += 1.f;
        } else {
          g = *reinterpret_cast<const float4*>(o.g + q * 4);
          b = *reinterpret_cast<const float4*>(o.b + q * 4);
        }
        float4 y;
        y.x = act_apply((v[i].x - mean) * rstd * g.x + b.x, act);
        y.y = act_apply((v[i].y - mean) * rstd * g.y + b.y, act);
        y.z = act_apply((v[i].z - mean) * rstd * g.z + b.z, act);
        y.w = act_apply((v[i].w - mean) * rstd * g.w + b.w, act);
        store4(o.Y, (long)row * o.ldy + o.ycol0 + q * 4, y.x, y.y, y.z, y.w, o.prec16);
      }
    }
  }
}

// row -> utterance id table (for kernels that walk rows flat)
__global__ void row_utt_kernel(const int* __restrict__ seg_off, int n_utt, int* __restrict__ row_utt) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  for (int r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) row_utt[r] = u;
}

// ---------------------------------------------------------------------------------------------
// depthwise Conv1d along time (groups = C): y[r][c] = b[c] + sum_k w[c][k] x[r + k - pad][c], zero outside the
// utterance (ConvNeXtBlock.dwconv, models/generator.py:449-455; conv_next.py:25-27; cross_post.0).
// Block = 64 rows x 64 channels staged in LDS with halo; thread = (channel, 16-row strip).
// Wt is tap-major [K][C] so a wave reads 64 consecutive channels.  Optional fused SiLU.
// ---------------------------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(256) dwconv_kernel(const float* __restrict__ X, int ldx, float* __restrict__ Y, int ldy, int C,
                                                     const int* __restrict__ seg_off, const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, int K, int act) {
  constexpr int RB = 64;
  __shared__ float tile[(RB + KMAX - 1) * 64];
  const int u = blockIdx.z;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  const int r0 = lo + blockIdx.y * RB;
  if (r0 >= hi) return;
  const int c0 = blockIdx.x * 64;
  const int pad = (K - 1) / 2;
  const int nrows = RB + K - 1;
  for (int i = threadIdx.x; i < nrows * 16; i += 256) {
    const int rr = i >> 4, c4 = (i & 15) * 4;
    const int g = r0 - pad + rr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g >= lo && g < hi && c0 + c4 < C) v = *reinterpret_cast<const float4*>(X + (long)g * ldx + c0 + c4);
    *reinterpret_cast<float4*>(&tile[rr * 64 + c4]) = v;
  }
  __syncthreads();
  const int c = threadIdx.x & 63, strip = threadIdx.x >> 6;  // 4 strips of 16 rows
  if (c0 + c >= C) return;
  float acc[16];
  const float bv = bias[c0 + c];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = bv;
  for (int k = 0; k < K; ++k) {
    const float w = Wt[(long)k * C + c0 + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += w * tile[(strip * 16 + i + k) * 64 + c];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + strip * 16 + i;
    if (r < hi) Y[(long)r * ldy + c0 + c] = act_apply(acc[i], act);
  }
}

// ---------------------------------------------------------------------------------------------
// ConvNeXt front half in one launch (models/generator.py:449-462): depthwise conv along time + adaptive LayerNorm over channels,
//   y[r] = LN_C(dw(x)[r]) * (1 + gamma_u) + beta_u,
// for C <= 512 and the kernel sizes the generator uses (template K).  A block owns ROWS rows x all channels.  Every thread pulls
// the ROWS + K - 1 inputs of its channel straight from global memory into a register window (one coalesced 256-byte row segment
// per wave and row, all of them in flight at once; the halo rows of neighbouring blocks come out of L2), the conv results go to
// LDS (ROWS x C floats: 32 KB at 16 rows, so five blocks share a CU - staging the INPUT rows + halo there, as round 2 did,
// took 94 KB at K = 31 = one block of four waves per CU, and the kernel ran at 0.19 of the HBM rate), and each wave normalises
// its share of the rows - the intermediate [rows, C] tensor (one write + one read, and one of two launches) never reaches HBM.
// ROWS = 32 for the long kernels halves their halo re-reads.  Per element the arithmetic and its order are those of
// dwconv_kernel followed by row_layernorm_kernel (taps in order; lane-strided float4 sums, then the wave reduction).
// ---------------------------------------------------------------------------------------------
constexpr int kDwLnMaxC = 512;
template <int K, int ROWS>
__global__ void __launch_bounds__(256) dwconv_ln_kernel(const float* __restrict__ X, int ldx, int C, const int* __restrict__ seg_off,
                                                        const float* __restrict__ Wt, const float* __restrict__ bias, float eps,
                                                        const float* __restrict__ style, int ld_style, int gcol0, float* __restrict__ Y, int ldy,
                                                        int prec16) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-descriptor type exists on the device side only)
  constexpr int NR = ROWS + K - 1;
  __shared__ float tile[ROWS * kDwLnMaxC];
  const int u = blockIdx.y;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  const int r0 = lo + blockIdx.x * ROWS;
  if (r0 >= hi) return;
  constexpr int pad = (K - 1) / 2;
  const int c4n = C / 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (long)lo * ldx), 0, ((hi - lo - 1) * ldx + C) * 4, 0x00020000);
  // depthwise conv: thread = channels tid, tid + 256 (C <= 512), all ROWS rows, inputs held in a register window
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = threadIdx.x + 256 * h;
    if (c < C) {
      float win[NR];
      // rows outside the utterance (the conv's zero padding) are out-of-range offsets of the utterance's buffer descriptor: the hardware
      // returns zeros, the NR loads carry no clamp, select or branch and are all in flight together
#pragma unroll
      for (int i = 0; i < NR; ++i) win[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, ((r0 - lo - pad + i) * ldx + c) * 4, 0, 0));
      const float bv = bias[c];
      float acc[ROWS];
#pragma unroll
      for (int i = 0; i < ROWS; ++i) acc[i] = bv;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float w = Wt[(long)k * C + c];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) acc[i] += w * win[i + k];
      }
#pragma unroll
      for (int i = 0; i < ROWS; ++i) tile[i * C + c] = acc[i];
    }
  }
  __syncthreads();
  // LayerNorm: wave w takes rows w, w + 4, ...
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* gp0 = style + (long)u * ld_style + gcol0;
  float4 gg[2], bb[2];
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {
    const int q = lane + 64 * q2;
    gg[q2] = bb[q2] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < c4n) {
      gg[q2] = *reinterpret_cast<const float4*>(gp0 + q * 4);
      bb[q2] = *reinterpret_cast<const float4*>(gp0 + C + q * 4);
      gg[q2].x += 1.f; gg[q2].y += 1.f; gg[q2].z += 1.f; gg[q2].w += 1.f;
    }
  }
  for (int i = wv; i < ROWS; i += 4) {
    const int r = r0 + i;
    if (r >= hi) break;
    float4 v[2];
    float sum = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int q = lane + 64 * q2;
      v[q2] = q < c4n ? *reinterpret_cast<const float4*>(&tile[i * C + q * 4]) : make_float4(0.f, 0.f, 0.f, 0.f);
      sum += v[q2].x + v[q2].y + v[q2].z + v[q2].w;
    }
    const float mean = wave_sum(sum) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
      if (lane + 64 * q2 < c4n) {
        const float a = v[q2].x - mean, b = v[q2].y - mean, c = v[q2].z - mean, d = v[q2].w - mean;
        ss += a * a + b * b + c * c + d * d;
      }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int q = lane + 64 * q2;
      if (q < c4n) {
        const float4 g = gg[q2], b = bb[q2];
        store4(Y, (long)r * ldy + q * 4, (v[q2].x - mean) * rstd * g.x + b.x, (v[q2].y - mean) * rstd * g.y + b.y, (v[q2].z - mean) * rstd * g.z + b.z,
               (v[q2].w - mean) * rstd * g.w + b.w, prec16);
      }
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// GRN (models/generator.py:496-499): Gx[u][c] = ||U[:, c]||_2 over the utterance's rows (from the per-tile
// partial sums the pwconv1 GEMM epilogue wrote), Nx = Gx / (mean_c Gx + 1e-6).  GRN(U) = U*(gamma*Nx + 1) + beta
// is folded into pwconv2:  W2_u[co][c] = W2[co][c] * (gamma[c]*Nx[u][c] + 1)   (beta goes into the bias at load).
// ---------------------------------------------------------------------------------------------
// gx[u][c] = sqrt(sum over the utterance's 32-row sub-tiles of the partial sums of squares); grid (ceil(C/256), n_utt)
__global__ void __launch_bounds__(256) grn_gx_kernel(const float* __restrict__ part, int ld_ss, int ss_stride, const int* __restrict__ seg_off,
                                                     int C, float* __restrict__ gx, int ld_gx) {
  // block = 32 channels x 8 slot lanes (one thread per channel walking all slots: 100 dependent-latency loads at 10-s utterances, 31 us);
  // lane sl sums slots sl, sl + 8, ... and the eight partial sums are added in lane order (deterministic)
  __shared__ float red[8][33];
  const int u = blockIdx.y, cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int nsub = (seg_off[u + 1] - seg_off[u] + 31) / 32;
  float s = 0.f;
  if (c < C)
    for (int t = sl; t < nsub; t += 8) s += part[((long)u * ss_stride + t) * ld_ss + c];
  red[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i][cl];
    gx[(long)u * ld_gx + c] = sqrtf(tot);
  }
}

// Wu[u][row][k] = W[row][k] * (gamma[k] * gx[u][k] / (mean_k gx[u][:] + 1e-6) + 1)   (W packed [Npad][kc], one tap)
// every block recomputes the channel mean of its utterance (kc values, L2 resident) - no extra launch.
// PREC != 0: the scaled copy is written rounded to bf16 / fp16 (the contraction's 16-bit operand mode).
template <int PREC>
__global__ void __launch_bounds__(256) scale_weight_kernel(const float* __restrict__ W, const float* __restrict__ gx, int ld_gx,
                                                           const float* __restrict__ gamma, void* __restrict__ Wu_, int npad, int kc) {
  __shared__ float red[4];
  const int u = blockIdx.y;
  float part = 0.f;
  for (int k = threadIdx.x; k < kc; k += 256) part += gx[(long)u * ld_gx + k];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1] + red[2] + red[3]) / (float)kc + 1e-6f);
  const long total4 = (long)npad * kc / 4;
  const int k4n = kc / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int k4 = (int)(i % k4n) * 4;
    const float4 w = reinterpret_cast<const float4*>(W)[i];
    const float4 g = *reinterpret_cast<const float4*>(gx + (long)u * ld_gx + k4);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + k4);
    const f32x4 v = {w.x * (ga.x * g.x * inv + 1.f), w.y * (ga.y * g.y * inv + 1.f), w.z * (ga.z * g.z * inv + 1.f), w.w * (ga.w * g.w * inv + 1.f)};
    if constexpr (PREC == PREC_F32) reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Wu_) + (long)u * npad * kc)[i] = v;
    else reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(Wu_) + (long)u * npad * kc)[i] = pack4_16<PREC>(v);
  }
}

// GRN as an input affine of pwconv2 (split-fp32 contractions): xaff[u][0][k] = gamma[k] * gx[u][k] / (mean_k gx[u] + 1e-6) + 1, xaff[u][1][k] = 0;
// pad columns k >= C get scale 0 (the contraction's staging treats scale 0 as "pad: contributes nothing").  models/generator.py:488-499
__global__ void __launch_bounds__(256) grn_xaff_kernel(const float* __restrict__ gx, int ld_gx, const float* __restrict__ gamma, float* __restrict__ xaff, int ld, int C) {
  __shared__ float red[4];
  const int u = blockIdx.x;
  float part = 0.f;
  for (int k = threadIdx.x; k < C; k += 256) part += gx[(long)u * ld_gx + k];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1] + red[2] + red[3]) / (float)C + 1e-6f);
  for (int k = threadIdx.x; k < ld; k += 256) {
    xaff[((long)u * 2) * ld + k] = k < C ? gamma[k] * gx[(long)u * ld_gx + k] * inv + 1.f : 0.f;
    xaff[((long)u * 2 + 1) * ld + k] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Length regulator part 2 + decoder front end (models/speech_predictor.py:88-97, models/decoder.py:48-51):
//   frame t4 of utterance u takes token tok[t4>>2]           (alignment.repeat_interleave(4) then enc @ alignment)
//   pitch/energy: nn.Upsample(scale 4, linear, align_corners=False) from the T-rate curves
// Here: gather of the phoneme encoding into time-major rows.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ enc, int ld_enc, const int* __restrict__ src_row,
                                                          float* __restrict__ Y, int ldy, int ycol0, int C, int n_rows, const int* __restrict__ n_rows_dev) {
  if (n_rows_dev) n_rows = min(n_rows, *n_rows_dev);  // capacity segments: src_row is filled for the real rows only
  const int nv = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n_rows * nv; i += (long)gridDim.x * 256) {
    const int r = (int)(i / nv), c4 = (int)(i % nv) * 4;
    *reinterpret_cast<float4*>(Y + (long)r * ldy + ycol0 + c4) = *reinterpret_cast<const float4*>(enc + (long)src_row[r] * ld_enc + c4);
  }
}

// Frame offsets of a batch from the predicted durations, on the device (the reference reads the durations on the host to size the
// alignment matrix: train/test_onnx.py:65-66, train/utils.py:476-489): T_u = sum of utterance u's durations, off_T = cumulative
// sum, off_T4 = 4 x.  cap_off [n_utt + 1]: the caller's CAPACITY layout (upper bounds the host sized every buffer and grid by);
// need [n_utt] receives T_u itself.  An utterance that does not fit its capacity is truncated to it, so everything downstream
// stays in bounds; the caller, who reads `need` together with the output, sees need[u] > capacity and repeats the call.
// (No shared error word: several calls may be in flight on different streams.)  One block; a wave per utterance sums its
// durations, one thread scans.
__global__ void __launch_bounds__(1024) frame_offsets_kernel(const int* __restrict__ dur, const int* __restrict__ tok_off, int n_utt,
                                                             const int* __restrict__ cap_off, int* __restrict__ off_T, int* __restrict__ off_T4,
                                                             int* __restrict__ need) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int u = wave; u < n_utt; u += 16) {
    int s = 0;
    for (int i = tok_off[u] + lane; i < tok_off[u + 1]; i += 64) s += max(dur[i], 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) need[u] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    off_T[0] = 0;
    off_T4[0] = 0;
    for (int u = 0; u < n_utt; ++u) {
      run += min(need[u], cap_off[u + 1] - cap_off[u]);
      off_T[u + 1] = run;
      off_T4[u + 1] = 4 * run;
    }
  }
}

// durations [sum P] (int) per utterance -> frame->token row map at rate `rep` (rep=1: T frames, rep=4: T4 frames).
// One block per utterance, any number of tokens: the durations are scanned 256 at a time (LDS scan + running carry) and
// every token writes its own rep * dur frames (at most 4 * 46), so there is no per-utterance token limit.
// tok_off: token offsets [n_utt+1]; frm_off: frame offsets at this rate [n_utt+1] (= rep * sum of durations); frames
// past rep * sum(dur) (inconsistent offsets) take the last token, nothing is written outside [frm_off[u], frm_off[u+1]).
__global__ void __launch_bounds__(256) frame_token_map_kernel(const int* __restrict__ dur, const int* __restrict__ tok_off,
                                                              const int* __restrict__ frm_off, int rep, int* __restrict__ src_row) {
  __shared__ int part[256];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int t0 = tok_off[u], P = tok_off[u + 1] - t0;
  const int f0 = frm_off[u], nf = frm_off[u + 1] - f0;
  int carry = 0;
  for (int base = 0; base < P; base += 256) {
    const int i = base + tid;
    const int d = i < P ? max(dur[t0 + i], 0) : 0;
    part[tid] = d;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    const int end = carry + part[tid];
    const int lo = rep * (end - d), hi = min(rep * end, nf);
    for (int f = lo; f < hi; ++f) src_row[f0 + f] = t0 + i;
    carry += part[255];
    __syncthreads();
  }
  for (int f = rep * carry + tid; f < nf; f += 256) src_row[f0 + f] = t0 + max(P - 1, 0);
}

// Linear x4 upsample (align_corners=False) of a per-frame curve, per utterance.
__global__ void __launch_bounds__(256) upsample4_kernel(const float* __restrict__ x, const int* __restrict__ off_T, const int* __restrict__ off_T4,
                                                        float* __restrict__ y) {
  const int u = blockIdx.y;
  const int a0 = off_T[u], T = off_T[u + 1] - a0, b0 = off_T4[u];
  for (int o = blockIdx.x * 256 + threadIdx.x; o < 4 * T; o += gridDim.x * 256) {
    float src = fmaxf(((float)o + 0.5f) * 0.25f - 0.5f, 0.0f);
    const int i0 = (int)floorf(src);
    const int i1 = min(i0 + 1, T - 1);
    const float lam = src - (float)i0;
    y[b0 + o] = (1.0f - lam) * x[a0 + i0] + lam * x[a0 + i1];
  }
}

// Decoder front end: F0 = wn-conv1d(1->1,k3)(pitch), N likewise (decoder.py:48-49), written with the phoneme
// encoding into the encode block's input [asr | F0 | N | 0..] and into the constant columns of both ping-pong
// concat buffers [x(512) | asr_res(64) | F0 | N | 0..] (decoder.py:51,56).
struct FrontArgs {
  const float* asr; int ld_asr;      // [rows, 128]
  const float* pitch; const float* energy;  // [rows]
  float wf[3], bf, wn[3], bn;        // folded 3-tap filters
  float* enc_in; int ld_enc;         // [rows, 160]
  float* xa; float* xb; int ld_x;    // [rows, 608]
  int c_asr, c_hidden, c_res;        // 128, 512, 64
};
__global__ void __launch_bounds__(256) decoder_front_kernel(FrontArgs a, const int* __restrict__ seg_off) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  const int nv = a.ld_enc / 4;
  const long total = (long)(hi - lo) * nv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = lo + (int)(i / nv), c4 = (int)(i % nv) * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < a.c_asr) {
      o = *reinterpret_cast<const float4*>(a.asr + (long)r * a.ld_asr + c4);
    } else {
      // the quads behind the phoneme encoding: quad 0 carries (F0, N, 0, 0); the same quads fill the tail [F0 | N | 0..] of both concat
      // buffers with 16-byte stores (one lane writing the ~30 zeros of a row one by one had made this kernel 10x slower than its traffic)
      const int k = (c4 - a.c_asr) / 4, nk = (a.ld_enc - a.c_asr) / 4;
      if (k == 0) {
        const float pm = r > lo ? a.pitch[r - 1] : 0.f, pp = r + 1 < hi ? a.pitch[r + 1] : 0.f;
        const float em = r > lo ? a.energy[r - 1] : 0.f, ep = r + 1 < hi ? a.energy[r + 1] : 0.f;
        o.x = a.wf[0] * pm + a.wf[1] * a.pitch[r] + a.wf[2] * pp + a.bf;
        o.y = a.wn[0] * em + a.wn[1] * a.energy[r] + a.wn[2] * ep + a.bn;
      }
      const int cc = a.c_hidden + a.c_res;  // 576: F0, N columns of the concat buffers; zero the tail
      for (int kk = k; cc + 4 * kk < a.ld_x; kk += nk) {
        const float4 t = kk == 0 ? o : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(a.xa + (long)r * a.ld_x + cc + 4 * kk) = t;
        *reinterpret_cast<float4*>(a.xb + (long)r * a.ld_x + cc + 4 * kk) = t;
      }
    }
    *reinterpret_cast<float4*>(a.enc_in + (long)r * a.ld_enc + c4) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// layout transposes at the module boundary: reference [B, C, T] (channel-major, equal T) <-> time-major rows
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) to_time_major_kernel(const float* __restrict__ X, int B, int C, int T, float* __restrict__ Y, int ldy,
                                                            int ycol0, int zero_to) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < T) ? X[((long)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < zero_to) Y[((long)b * T + t) * ldy + ycol0 + c] = tile[tx][i];
  }
}
__global__ void __launch_bounds__(256) to_channel_major_kernel(const float* __restrict__ X, int ldx, int xcol0, int B, int C, int T,
                                                               float* __restrict__ Y) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && t < T) ? X[((long)b * T + t) * ldx + xcol0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (c < C && t < T) Y[((long)b * C + c) * T + t] = tile[tx][i];
  }
}

// One extra output channel of a Conv1d as a dot product per row.  Used for the Nyquist bin of the vocoder's output convs:
// 1025 = 8 x 128 + 1 output channels, so the GEMM computes 1024 of them with full tiles and this kernel the last one
// (generator.py:344-358).  w is tap-major [ntaps][C].  A wave owns kRows consecutive rows and loads each of the kRows + ntaps - 1
// input rows it needs ONCE (a row-per-wave form re-read every row ntaps times: 0.27 ms at 61 440 rows); grid.y selects one
// of up to two (input, weights, output) sets so both heads go in one launch.
// x16: 0 = X holds fp32 rows; PREC_BF16 / PREC_F16 = 16-bit rows (ldx in elements); the weights stay fp32 either way.
struct ChanConvSet {
  const float* X;
  const float* w;
  float bias;
  float* Y;
};
constexpr int kChanRows = 8, kChanTaps = 7;
template <int X16>
__global__ void __launch_bounds__(256) single_channel_conv_kernel(ChanConvSet s0, ChanConvSet s1, int ldx, int C, const int* __restrict__ seg_off, int ntaps,
                                                                  int ldy, int ycol) {
  constexpr int x16 = X16;  // (compile time: with a run-time switch inside the unrolled row loop hipcc branched around every load and waited for each)
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-descriptor type exists on the device side only)
  const ChanConvSet& S = blockIdx.y == 0 ? s0 : s1;
  const int u = blockIdx.z;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  const int lane = threadIdx.x & 63;
  const int r0 = lo + (blockIdx.x * 4 + (threadIdx.x >> 6)) * kChanRows;
  if (r0 >= hi) return;
  const int pad = (ntaps - 1) / 2, nv = C / 4;
  // (rows a wave with fewer taps than kChanTaps does not need are loaded too: their weights are zero)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(S.X) + (long)lo * ldx * (x16 ? 2 : 4)), 0, ((hi - lo - 1) * ldx + C) * (x16 ? 2 : 4), 0x00020000);
  float acc[kChanRows];
#pragma unroll
  for (int i = 0; i < kChanRows; ++i) acc[i] = 0.f;
  for (int q = lane; q < nv; q += 64) {  // this lane's 4 channels: taps in registers, rows streamed once
    float4 wt[kChanTaps];
#pragma unroll
    for (int t = 0; t < kChanTaps; ++t) wt[t] = t < ntaps ? *reinterpret_cast<const float4*>(S.w + (long)t * C + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    // all input rows of this channel group are requested before the first is used; rows outside the utterance (the conv's zero
    // padding) are out-of-range offsets of the utterance's buffer descriptor and come back as zeros (no clamp, select or branch)
    float4 a[kChanRows + kChanTaps - 1];
#pragma unroll
    for (int j = 0; j < kChanRows + kChanTaps - 1; ++j) {
      const int off = (r0 - lo - pad + j) * ldx + q * 4;  // elements
      if constexpr (X16 != 0) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xr, off * 2, 0, 0);
        if constexpr (X16 == PREC_BF16) a[j] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
        else {
          const f16x4_t h = __builtin_bit_cast(f16x4_t, v);
          a[j] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
      } else {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, off * 4, 0, 0);
        a[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
#pragma unroll
    for (int j = 0; j < kChanRows + kChanTaps - 1; ++j) {
#pragma unroll
      for (int t = 0; t < kChanTaps; ++t) {
        const int i = j - t;  // output row r0 + i takes input row r0 + i + t - pad = g
        if (i >= 0 && i < kChanRows) acc[i] += a[j].x * wt[t].x + a[j].y * wt[t].y + a[j].z * wt[t].z + a[j].w * wt[t].w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kChanRows; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0 && r0 + i < hi) S.Y[(long)(r0 + i) * ldy + ycol] = v + S.bias;
  }
#endif
}

__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// x += dt * v (CfmSampler.solve_euler, models/cfm/cfm.py:79); separate multiply and add like the reference's
// `x + dt * dphi_dt` (no fused multiply-add, so the result is bit-identical to torch's two roundings)
__global__ void __launch_bounds__(256) euler_step_kernel(float* __restrict__ x, const float* __restrict__ v, float dt, long n) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float p = dt * v[i];
    x[i] = x[i] + p;
  }
}

// out[i] = i * step, i < n  (segment offsets of equally sized planes)
__global__ void seg_linear_kernel(int* __restrict__ out, int n, int step) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = i * step;
}

// fp32 rows -> 16-bit rows rounded to nearest even (the operand precision of the 16-bit modes): Y16[r][c] for c < width, zeros
// for c >= cols.  For contraction inputs whose producer is not one of the kernels that write 16-bit rows themselves.
template <int PREC>
__global__ void __launch_bounds__(256) cast_rows_kernel(const float* __restrict__ X, int ldx, int cols, unsigned short* __restrict__ Y, int ldy, int width,
                                                        long rows) {
  const int q = width / 8;  // 16-byte groups of 8 output elements per row
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * q; i += (long)gridDim.x * 256) {
    const long r = i / q;
    const int c0 = (int)(i % q) * 8;
    f32x4 out;
    if constexpr (PREC == PREC_BF16) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)(c0 + e < cols ? X[r * ldx + c0 + e] : 0.0f);  // round to nearest even
      out = __builtin_bit_cast(f32x4, o);
    } else {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)(c0 + e < cols ? X[r * ldx + c0 + e] : 0.0f);
      out = __builtin_bit_cast(f32x4, o);
    }
    *reinterpret_cast<f32x4*>(Y + r * ldy + c0) = out;
  }
}
// fp32 rows -> the three bf16 planes of the exact split (gemm.hip.h, PREC_X3): Y[p][r][c], planes `plane` elements apart, zeros beyond `cols`
__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ X, int ldx, int cols, unsigned short* __restrict__ Y, int ldy, long plane, int width, long rows) {
  const int q = width / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * q; i += (long)gridDim.x * 256) {
    const long r = i / q;
    const int c0 = (int)(i % q) * 4;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c0 + e < cols ? X[r * ldx + c0 + e] : 0.0f;
    u32x2 p0, p1, p2;
    split3_bf16(v, p0, p1, p2);
    unsigned short* o = Y + r * ldy + c0;
    *reinterpret_cast<u32x2*>(o) = p0;
    *reinterpret_cast<u32x2*>(o + plane) = p1;
    *reinterpret_cast<u32x2*>(o + 2 * plane) = p2;
  }
}
inline void launch_split_rows(hipStream_t st, const float* X, int ldx, int cols, unsigned short* Y, int ldy, long plane, long rows, int width = 0) {
  if (width == 0) width = ldy;
  const dim3 grid((unsigned)std::min<long>(4096, std::max<long>(1, (rows * (width / 4) + 255) / 256)));
  STTS_LAUNCH_PROF("split_rows_kernel", (size_t)rows * (cols * 4 + width * 6), split_rows_kernel, grid, dim3(256), st, X, ldx, cols, Y, ldy, plane, width, rows);
}
// width: columns written per row (a multiple of 8; 0 = the whole row stride ldy)
inline void launch_cast_rows(hipStream_t st, int prec, const float* X, int ldx, int cols, unsigned short* Y, int ldy, long rows, int width = 0) {
  if (width == 0) width = ldy;
  const dim3 grid((unsigned)std::min<long>(4096, std::max<long>(1, (rows * (width / 8) + 255) / 256)));
  const size_t bytes = (size_t)rows * (cols * 4 + width * 2);
  if (prec == PREC_BF16) STTS_LAUNCH_PROF("cast_rows_kernel", bytes, cast_rows_kernel<PREC_BF16>, grid, dim3(256), st, X, ldx, cols, Y, ldy, width, rows);
  else STTS_LAUNCH_PROF("cast_rows_kernel", bytes, cast_rows_kernel<PREC_F16>, grid, dim3(256), st, X, ldx, cols, Y, ldy, width, rows);
}

inline void launch_scale_weight(hipStream_t st, dim3 grid, int prec, const float* W, const float* gx, int ld_gx, const float* gamma, void* Wu, int npad,
                                int kc) {
  const size_t bytes = (size_t)grid.y * npad * kc * (prec == PREC_F32 ? 4 : 2) + (size_t)npad * kc * 4;  // per-utterance copies out, W in
  if (prec == PREC_BF16) STTS_LAUNCH_PROF("scale_weight_kernel", bytes, scale_weight_kernel<PREC_BF16>, grid, dim3(256), st, W, gx, ld_gx, gamma, Wu, npad, kc);
  else if (prec == PREC_F16) STTS_LAUNCH_PROF("scale_weight_kernel", bytes, scale_weight_kernel<PREC_F16>, grid, dim3(256), st, W, gx, ld_gx, gamma, Wu, npad, kc);
  else STTS_LAUNCH_PROF("scale_weight_kernel", bytes, scale_weight_kernel<PREC_F32>, grid, dim3(256), st, W, gx, ld_gx, gamma, Wu, npad, kc);
}

}  // namespace stts
