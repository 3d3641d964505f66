// wn_fused16_kernel<PREC, RT, LAST>: the fused WaveNet layer of the reverse flow (models/flow.py:63-88; see wn_fused.hip.h for
// the fp32 form) for the 16-bit operand modes: bf16 / fp16 operands on v_mfma_f32_16x16x32, fp32 accumulate, the k = 5 conv in
// its DIRECT form (five row-shifted reads of one LDS tile: rounding operands in a Winograd domain would not be the reference's
// arithmetic rounded at the contraction inputs, which is what the rounded oracle pins).
// Same structure as the fp32 kernel: a block owns 16 RT rows (RT = 4: 64, RT = 8: 128) and ALL 256 gate channels, 4 waves (one
// per SIMD), wave w owns the tanh and the sigmoid tiles of channels [32 w, 32 w + 32) for every row tile; weights are read
// straight from global memory in MFMA-fragment order (wn_fused16_pack: [wave][k-step of 32][tile][lane][8 x 16 bit], 1 KB per
// wave load), one tap (4 k-steps) ahead; products are oriented D^T = W x A^T so a lane holds four consecutive channels.
// Rounding points = those of wn_layer_kernel<PREC>: h when it enters LDS, the gated activations, the finished `out` tile before
// `post`, the coupled half of z before `pre`; gate, coupling and the h / out / z updates are fp32.
// At 16-bit rates the matrix work is ~2 us per 64 rows; the launch is bound by the weight stream (0.38 MB per block from L2,
// 3.3 us at the measured 117 GB/s per CU) and by its fixed parts, so blocks are as tall as the batch allows.
#pragma once
#include <functional>
#include <vector>

#include "wn_fused.hip.h"

namespace stts {

template <int PREC>
__device__ __forceinline__ f32x4 mfma16x16(const f32x4 a, const f32x4 b, const f32x4 c) {
  if constexpr (PREC == PREC_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// four fp32 -> four 16-bit operands (round to nearest even), as two dwords
template <int PREC>
__device__ __forceinline__ u32x2 round4(const f32x4 v) {
  if constexpr (PREC == PREC_BF16) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    const b4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return __builtin_bit_cast(u32x2, o);
  } else {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    return __builtin_bit_cast(u32x2, o);
  }
}

struct WnFused16Args {
  const float* Hin;
  float* Hout;
  float* Out;
  const int* seg_off;
  const unsigned short* W1;  // in_layers, fragments [4 waves][20 k-steps: tap-major, 4 x 32 channels][4 tiles: (tanh, sigmoid) x 2][64][8]
  const float* b1;           // [256] natural order
  const unsigned short* W2;  // res_skip, fragments [4][4 k-steps][4 (LAST: 2) tiles][64][8]
  const float* b2;
  const float* gate;
  int ld_gate, gcol0;
  int out_acc, tail;
  const unsigned short* W3;  // post (mean | logstd), fragments [4][4][2][64][8]
  const float* b3m;
  const float* b3s;
  float* Z;
  int ldz, zcol0;
  const unsigned short* W4;  // next block's pre, fragments [4][2 k-steps][2][64][8]
  const float* b4;
  float* Hpre;
  int n_inline;
  int seg_inline[kWnSegInline + 1];
};

template <int PREC, int RT, bool LAST>
__global__ void __launch_bounds__(256) wn_fused16_kernel(const WnFused16Args a) {
  constexpr int ROWS = 16 * RT, C = kWnC, NW = kWnWaves, CT = C / NW / 16, NCT = LAST ? CT : 2 * CT, TAPS = 5, PAD = 2;
  constexpr int KS = C / 32;  // 32-channel k-steps of a 128-channel contraction
  // 16-bit row tiles, 256 bytes per row = 16 slots of 8 channels, slot index XORed with (row & 15): the 16 lanes of an operand
  // read (same logical slot, 16 consecutive rows) then hit 16 different slots
  __shared__ f32x4 Hs[(ROWS + 2 * PAD) * 16];  // h rows [row0 - 2, row0 + ROWS + 2); later the coupled half of z
  __shared__ f32x4 As[ROWS * 16];              // gated activations; later the finished `out` tile

  const int utt = blockIdx.y;
  const int lo = a.n_inline ? a.seg_inline[utt] : a.seg_off[utt], hi = a.n_inline ? a.seg_inline[utt + 1] : a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * ROWS;
  if (row0 >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int nvalid = hi - row0;

  // ---- phase-1 weight stream: one tap (4 k-steps x 4 tiles = 16 KB per wave) ahead
  constexpr int T1 = KS * 2 * CT;  // fragments per tap: 4 k-steps x (tanh, sigmoid) x CT
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.W1) + (size_t)w * TAPS * (T1 * 64) + lane;
  f32x4 bq0[T1], bq1[T1];
  auto load1 = [&](f32x4(&dst)[T1], int tap) {
#pragma unroll
    for (int j = 0; j < T1; ++j) dst[j] = w1[(tap * T1 + j) * 64];
  };
  load1(bq0, 0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: h rows -> 16-bit LDS tile (rows outside the utterance are the conv's zero padding)
  for (int idx = tid; idx < (ROWS + 2 * PAD) * 16; idx += 256) {
    const int r = idx >> 4, sl = idx & 15;
    const int row = row0 + r - PAD;
    const bool ok = row >= lo && row < hi;
    const float* src = a.Hin + (long)min(max(row, lo), hi - 1) * C + sl * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
    // (eight element-wise conversions into one 8 x 16-bit vector: taking the dwords of a converted 4-vector apart again
    //  came out with the first dword twice under this compiler)
    f32x4 o;
    if constexpr (PREC == PREC_BF16) {
      const bf16x8 h = {(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w, (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
      o = __builtin_bit_cast(f32x4, h);
    } else {
      const f16x8 h = {(_Float16)v0.x, (_Float16)v0.y, (_Float16)v0.z, (_Float16)v0.w, (_Float16)v1.x, (_Float16)v1.y, (_Float16)v1.z, (_Float16)v1.w};
      o = __builtin_bit_cast(f32x4, h);
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    Hs[r * 16 + (sl ^ (r & 15))] = ok ? o : z;
  }
  // gate operands of this lane's channels 32 w + 16 c + 4 lq + (0..3)
  f32x4 ba[CT], bb[CT], ga[CT], gb[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int ch = (C / NW) * w + 16 * c + 4 * lq;
    ba[c] = *reinterpret_cast<const f32x4*>(a.b1 + ch);
    bb[c] = *reinterpret_cast<const f32x4*>(a.b1 + C + ch);
    ga[c] = *reinterpret_cast<const f32x4*>(a.gate + (long)utt * a.ld_gate + a.gcol0 + ch);
    gb[c] = *reinterpret_cast<const f32x4*>(a.gate + (long)utt * a.ld_gate + a.gcol0 + C + ch);
  }
  __syncthreads();

  // B operand of the 16x16x32 MFMA from a row tile: lane (row l15, k-group lq) reads 8 consecutive channels of its row
  auto rows_frag = [&](const f32x4* tile, int row, int kstep) { return tile[row * 16 + ((4 * kstep + lq) ^ (row & 15))]; };

  // ---- phase 1: conv k5, K = 5 taps x 128 channels; acc[half][c][rt]: channels 32 w + 16 c + 4 lq + i, row 16 rt + l15
  f32x4 acc[2][CT][RT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[h][c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto tap1 = [&](int tap, const f32x4(&cur)[T1]) {
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      f32x4 av[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av[rt] = rows_frag(Hs, 16 * rt + l15 + tap, t);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[h][c][rt] = mfma16x16<PREC>(cur[(t * 2 + h) * CT + c], av[rt], acc[h][c][rt]);
    }
  };
#pragma unroll 1
  for (int tap = 0; tap < TAPS - 1; tap += 2) {
    load1(bq1, tap + 1);
    __builtin_amdgcn_sched_barrier(0);
    tap1(tap, bq0);
    load1(bq0, tap + 2);
    __builtin_amdgcn_sched_barrier(0);
    tap1(tap + 1, bq1);
  }
  tap1(TAPS - 1, bq0);

  // ---- phase-2 operands: all res/skip weights, bias, the h / out values the epilogue updates
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.W2) + (size_t)w * KS * (NCT * 64) + lane;
  f32x4 cq2[KS][NCT];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int c = 0; c < NCT; ++c) cq2[t][c] = w2[(t * NCT + c) * 64];
  f32x4 bv[NCT], old[RT][NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int n = 16 * NCT * w + 16 * c + 4 * lq;
    bv[c] = *reinterpret_cast<const f32x4*>(a.b2 + n);
    const bool to_h = !LAST && n < C;
    const int col = (!LAST && n >= C) ? n - C : n;
    const float* src = to_h ? a.Hin : a.Out;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)(row0 + 16 * rt + l15) * C + col);  // (kWnRowPad rows of slack)
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      old[rt][c] = (to_h || a.out_acc) ? v : z;
    }
  }

  // ---- gate -> 16-bit activations in LDS (4 consecutive channels = 8 bytes of a row's slot)
  auto put4 = [&](f32x4* tile, int row, int ch, const f32x4 v) {
    u32x2* p = reinterpret_cast<u32x2*>(tile + row * 16 + ((ch >> 3) ^ (row & 15)));
    p[(ch >> 2) & 1] = round4<PREC>(v);
  };
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 va = acc[0][c][rt] + ba[c] + ga[c], vb = acc[1][c][rt] + bb[c] + gb[c];
      f32x4 act;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390082f * va[i]) + 1.0f);
        act[i] = th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * vb[i]));
      }
      put4(As, 16 * rt + l15, (C / NW) * w + 16 * c + 4 * lq, act);
    }
  __syncthreads();

  // ---- phase 2: res/skip, K = 128 from LDS; wave w owns columns [16 NCT w, 16 NCT (w + 1))
  f32x4 acc2[RT][NCT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc2[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    f32x4 av[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) av[rt] = rows_frag(As, 16 * rt + l15, t);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc2[rt][c] = mfma16x16<PREC>(cq2[t][c], av[rt], acc2[rt][c]);
  }

  if constexpr (!LAST) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int n = 16 * NCT * w + 16 * c + 4 * lq;
      const bool to_h = n < C;
      const int col = to_h ? n : n - C;
      float* dst = to_h ? a.Hout : a.Out;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(dst + (long)(row0 + 16 * rt + l15) * C + col) = old[rt][c] + (acc2[rt][c] + bv[c]);
    }
    return;
  } else {
    // ---- tail: post + reverse coupling (+ the next block's pre); wave w: mean / log-std tiles of channels [16 w, 16 w + 16)
    const f32x4* w3 = reinterpret_cast<const f32x4*>(a.W3) + (size_t)w * KS * (2 * 64) + lane;
    f32x4 pq[KS][2];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      pq[t][0] = w3[(t * 2 + 0) * 64];
      pq[t][1] = w3[(t * 2 + 1) * 64];
    }
    const int cc = 16 * w + 4 * lq;
    const f32x4 pm = *reinterpret_cast<const f32x4*>(a.b3m + cc), ps = *reinterpret_cast<const f32x4*>(a.b3s + cc);
    f32x4 zold[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) zold[rt] = *reinterpret_cast<const f32x4*>(a.Z + (long)(row0 + 16 * rt + l15) * a.ldz + a.zcol0 + cc);
    __syncthreads();  // every wave has finished reading the gated activations
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) put4(As, 16 * rt + l15, 16 * NCT * w + 16 * c + 4 * lq, old[rt][c] + (acc2[rt][c] + bv[c]));
    __syncthreads();
    f32x4 acc3[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc3[rt][0] = acc3[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KS; ++t) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 av = rows_frag(As, 16 * rt + l15, t);
        acc3[rt][0] = mfma16x16<PREC>(pq[t][0], av, acc3[rt][0]);
        acc3[rt][1] = mfma16x16<PREC>(pq[t][1], av, acc3[rt][1]);
      }
    }
    constexpr int KS4 = KS / 2;  // K = 64
    const f32x4* w4 = reinterpret_cast<const f32x4*>(a.W4) + (size_t)w * KS4 * (2 * 64) + lane;
    f32x4 rq[KS4][2];
    f32x4 hb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (a.tail > 1) {
#pragma unroll
      for (int t = 0; t < KS4; ++t) {
        rq[t][0] = w4[(t * 2 + 0) * 64];
        rq[t][1] = w4[(t * 2 + 1) * 64];
      }
      hb[0] = *reinterpret_cast<const f32x4*>(a.b4 + 32 * w + 4 * lq);
      hb[1] = *reinterpret_cast<const f32x4*>(a.b4 + 32 * w + 16 + 4 * lq);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 mm = acc3[rt][0] + pm, ls = acc3[rt][1] + ps;
      f32x4 z1;
#pragma unroll
      for (int i = 0; i < 4; ++i) z1[i] = (zold[rt][i] - mm[i]) * __expf(-ls[i]);  // flow.py:209
      if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(a.Z + (long)(row0 + 16 * rt + l15) * a.ldz + a.zcol0 + cc) = z1;
      put4(Hs, 16 * rt + l15, cc, z1);  // the conv tile is long dead: rows [0, ROWS) x channels [0, 64) of it now hold z1
    }
    if (a.tail < 2) return;
    __syncthreads();
    f32x4 acc4[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc4[rt][0] = acc4[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KS4; ++t) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 av = rows_frag(Hs, 16 * rt + l15, t);
        acc4[rt][0] = mfma16x16<PREC>(rq[t][0], av, acc4[rt][0]);
        acc4[rt][1] = mfma16x16<PREC>(rq[t][1], av, acc4[rt][1]);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(a.Hpre + (long)(row0 + 16 * rt + l15) * C + 32 * w + 16 * c + 4 * lq) = acc4[rt][c] + hb[c];
  }
}

// MFMA-fragment order for v_mfma_f32_16x16x32_{bf16,f16} A operands (weights = the MFMA's row operand) read straight from global:
//   out[((w * ksteps + s) * tiles + j) * 64 + lane][e] = round16(row(w, j, lane & 15)[32 s + 8 (lane >> 4) + e]),  e = 0..7
inline std::vector<unsigned short> pack_fragments16(int prec, int waves, int ksteps, int tiles, const std::function<const float*(int, int, int)>& row,
                                                    unsigned short (*to_bf16)(float), unsigned short (*to_f16)(float)) {
  std::vector<unsigned short> out((size_t)waves * ksteps * tiles * 64 * 8, 0);
  for (int w = 0; w < waves; ++w)
    for (int j = 0; j < tiles; ++j)
      for (int c = 0; c < 16; ++c) {
        const float* src = row(w, j, c);
        if (!src) continue;
        for (int s = 0; s < ksteps; ++s)
          for (int kq = 0; kq < 4; ++kq)
            for (int e = 0; e < 8; ++e) {
              const float v = src[32 * s + 8 * kq + e];
              out[((((size_t)w * ksteps + s) * tiles + j) * 64 + kq * 16 + c) * 8 + e] = prec == PREC_BF16 ? to_bf16(v) : to_f16(v);
            }
      }
  return out;
}

}  // namespace stts
