// wn_fused_x3_kernel<RT, LAST>: the fused WaveNet layer of the reverse flow (models/flow.py:63-88, 196-218; wn_fused.hip.h is the form on
// the f32 matrix cores) as a SPLIT-FP32 contraction (gemm.hip.h, PREC_X3): every fp32 operand - h, the gated activations, the finished
// `out` tile, the coupled half of z, and every weight - is the exact sum of three bf16 terms, and a product is formed from its six largest
// bf16 x bf16 cross terms on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Nothing is rounded to 16 bits.
//
// Structure = wn_fused16_kernel's (wn_fused16.hip.h): a block owns 16 RT rows and all 256 gate channels; the fragment arrays are packed for 4 waves, packed wave w
// owning the tanh and sigmoid tiles of channels [32 w, 32 w + 32).  The kernel runs with EIGHT waves (two per SIMD; template parameter NW): two waves share a packed
// wave's slice, one tile of 16 channels each - same arrays, same LDS tiles, same bits out, each fragment still fetched by exactly one wave; the second wave per SIMD
// hides part of the weight stream's and the gate's latency: 23.6 -> 22.6 us per launch at B = 8, flow 0.756 -> 0.723 ms, -4..5 % at B = 16 .. 64 (STTS_WN_X3_WAVES=4|8); the k = 5 conv in its DIRECT form (five row-shifted reads of one LDS tile;
// the f32 kernel's F(2,5) Toom-Cook form saves 40 % of the multiplies, the split form 62 % of the matrix-pipe cycles of a direct f32 conv);
// weights come straight from global memory in MFMA-fragment order, three planes per matrix, HALF A TAP (2 k-steps x 4 tiles x 3 planes =
// 24 KB per wave) ahead of the MFMAs that consume them; products are oriented D^T = W x A^T so a lane holds four consecutive channels.
// Per layer and block of 32 rows (RT = 2): 5 taps x 4 k-steps x 8 tiles x 6 = 960 MFMAs of 16 cycles per wave = 7.4 us at 2.07 GHz against
// 13.5 us for the F(2,5) form on v_mfma_f32_16x16x4_f32; the weight stream is 1.5 x the f32 kernel's bytes (6 instead of 4 per weight).
#pragma once
#include "wn_fused16.hip.h"

namespace stts {

struct WnFusedX3Args {
  WnFused16Args b;        // W1 .. W4 point to plane 0 of the three-plane fragment arrays
  long p1, p2, p3, p4;    // f32x4 units between two planes of W1 / W2 / W3 / W4
  long long* dbg;         // per-block phase stamps, written only by a -DSTTS_WN_TRACE build (diagnostics; wn_fused.hip.h's layout)
};

// B operand planes of four fp32 values -> their slots in the three 16-bit row tiles
__device__ __forceinline__ void put4_x3(f32x4* tile, int plane, int row, int ch, const f32x4 v) {
  u32x2 p0, p1, p2;
  split3_bf16(v, p0, p1, p2);
  u32x2* q = reinterpret_cast<u32x2*>(tile + row * 16 + ((ch >> 3) ^ (row & 15))) + ((ch >> 2) & 1);
  q[0] = p0;
  q[plane * 2] = p1;
  q[plane * 4] = p2;
}

typedef unsigned u32x4_x3 __attribute__((ext_vector_type(4)));

// NW = 4 (one wave per SIMD) or 8: two waves share the slice of the fragment arrays that was packed for one of four - wave w = 2 pw + sub owns tile(s) `sub` of packed wave pw's
// tiles in every phase - so the arrays, the LDS tiles and the results are the same; the weight stream per block too (each fragment is fetched by one wave).
template <int RT, bool LAST, int NW = kWnWaves>
__global__ void __launch_bounds__(64 * NW) wn_fused_x3_kernel(const WnFusedX3Args ax) {
  const WnFused16Args& a = ax.b;
  constexpr int ROWS = 16 * RT, C = kWnC, PW = kWnWaves, SUB = NW / PW, CTP = C / PW / 16, CT = CTP / SUB, NCTP = LAST ? CTP : 2 * CTP, NCT = NCTP / SUB, TAPS = 5, PAD = 2;
  static_assert(NW == PW || NW == 2 * PW, "4 or 8 waves");
  constexpr int KS = C / 32;                   // 32-channel k-steps of a 128-channel contraction
  constexpr int HROWS = ROWS + 2 * PAD;
  constexpr int HPL = HROWS * 16, APL = ROWS * 16;  // f32x4 slots of one plane
  // three 16-bit planes per tile, 256 bytes per row = 16 slots of 8 channels, slot index XORed with (row & 15) (wn_fused16.hip.h)
  __shared__ f32x4 Hs[3 * HPL];  // h rows [row0 - 2, row0 + ROWS + 2); later the coupled half of z
  __shared__ f32x4 As[3 * APL];  // gated activations; later the finished `out` tile

  const int utt = blockIdx.y;
  const int lo = a.n_inline ? a.seg_inline[utt] : a.seg_off[utt], hi = a.n_inline ? a.seg_inline[utt + 1] : a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * ROWS;
  if (row0 >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pw = w / SUB, sub = w % SUB;  // packed wave whose slice this wave shares, and which part of it
  const int l15 = lane & 15, lq = lane >> 4;
  const int nvalid = hi - row0;
#ifdef STTS_WN_TRACE
  long long* const dbg_rec = ax.dbg ? ax.dbg + 64 * (long)(blockIdx.x + gridDim.x * blockIdx.y) + 8 * w : nullptr;
  auto stamp = [&](int i) { if (dbg_rec && lane == 0) dbg_rec[i] = i == 6 || i == 7 ? wall_clock64() : clock64(); };
  stamp(6);
#else
  auto stamp = [](int) {};
#endif
  stamp(0);

  // ---- phase-1 weight stream: half a tap (2 k-steps x (tanh, sigmoid) x CT tiles, three planes) ahead
  constexpr int T1 = KS * 2 * CTP;  // packed fragments per tap and plane (of one packed wave)
  constexpr int H1P = T1 / 2;       // ... per half tap
  constexpr int H1 = H1P / SUB;     // this wave's share: index (tt * 2 + h) * CT + c  <->  packed (tt * 2 + h) * CTP + sub * CT + c
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.W1) + (size_t)pw * TAPS * (T1 * 64) + lane;
  f32x4 bq0[3][H1], bq1[3][H1];
  auto load1 = [&](f32x4(&dst)[3][H1], int half) {  // half = 2 tap + (k-steps 2, 3)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < H1; ++j) dst[p][j] = w1[p * ax.p1 + (size_t)(half * H1P + (j / CT) * CTP + sub * CT + j % CT) * 64];
  };
  load1(bq0, 0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: h rows -> the three planes of the LDS tile (rows outside the utterance are the conv's zero padding)
  for (int idx = tid; idx < HROWS * 16; idx += 64 * NW) {
    const int r = idx >> 4, sl = idx & 15;
    const int row = row0 + r - PAD;
    const bool ok = row >= lo && row < hi;
    const float* src = a.Hin + (long)min(max(row, lo), hi - 1) * C + sl * 8;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);  // (clamped address: the load is unconditional, the select follows)
    v0 = ok ? v0 : z;
    v1 = ok ? v1 : z;
    u32x2 a0, a1, a2, b0, b1, b2;
    split3_bf16(v0, a0, a1, a2);
    split3_bf16(v1, b0, b1, b2);
    const int o = r * 16 + (sl ^ (r & 15));
    Hs[o] = __builtin_bit_cast(f32x4, (u32x4_x3){a0.x, a0.y, b0.x, b0.y});
    Hs[o + HPL] = __builtin_bit_cast(f32x4, (u32x4_x3){a1.x, a1.y, b1.x, b1.y});
    Hs[o + 2 * HPL] = __builtin_bit_cast(f32x4, (u32x4_x3){a2.x, a2.y, b2.x, b2.y});
  }
  // gate operands of this lane's channels 32 w + 16 c + 4 lq + (0..3)
  f32x4 ba[CT], bb[CT], ga[CT], gb[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int ch = (C / PW) * pw + 16 * (sub * CT + c) + 4 * lq;
    ba[c] = *reinterpret_cast<const f32x4*>(a.b1 + ch);
    bb[c] = *reinterpret_cast<const f32x4*>(a.b1 + C + ch);
    ga[c] = *reinterpret_cast<const f32x4*>(a.gate + (long)utt * a.ld_gate + a.gcol0 + ch);
    gb[c] = *reinterpret_cast<const f32x4*>(a.gate + (long)utt * a.ld_gate + a.gcol0 + C + ch);
  }
  __syncthreads();
  stamp(1);

  // B operand of the 16x16x32 MFMA from a row tile: lane (row l15, k-group lq) reads 8 consecutive channels of its row
  auto rows_frag = [&](const f32x4* tile, int row, int kstep) { return tile[row * 16 + ((4 * kstep + lq) ^ (row & 15))]; };
  // six products of one (weight fragment, row fragment) pair: w0 x0 + w0 x1 + w1 x0 + w0 x2 + w1 x1 + w2 x0, smallest first
  auto mma6 = [&](const f32x4 w0, const f32x4 w1_, const f32x4 w2_, const f32x4 (&x)[3], f32x4 acc) {
    acc = mfma16x16<PREC_BF16>(w2_, x[0], acc);
    acc = mfma16x16<PREC_BF16>(w0, x[2], acc);
    acc = mfma16x16<PREC_BF16>(w1_, x[1], acc);
    acc = mfma16x16<PREC_BF16>(w1_, x[0], acc);
    acc = mfma16x16<PREC_BF16>(w0, x[1], acc);
    acc = mfma16x16<PREC_BF16>(w0, x[0], acc);
    return acc;
  };

  // ---- phase 1: conv k5, K = 5 taps x 128 channels; acc[half][c][rt]: channels 32 w + 16 c + 4 lq + i, row 16 rt + l15
  f32x4 acc[2][CT][RT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[h][c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the B fragments of a step (RT row tiles x three planes of the conv tile at one tap / k-step) are fetched from LDS one step AHEAD of the
  // 6 x 4 x RT MFMAs that consume them: the wave is alone on its SIMD, nothing else hides an LDS round trip
  f32x4 avc[RT][3];
  auto frags = [&](f32x4(&dst)[RT][3], int step) {  // step = 2 half + tt = 4 tap + k-step
    const int tap = step >> 2, t = step & 3;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[rt][p] = rows_frag(Hs + p * HPL, 16 * rt + l15 + tap, t);
  };
  frags(avc, 0);
  auto half1 = [&](int half, const f32x4(&cur)[3][H1]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int step = 2 * half + tt;
      f32x4 avn[RT][3];
      frags(avn, min(step + 1, 4 * TAPS - 1));  // (the last step re-reads itself: no branch in the loop body)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int j = (tt * 2 + h) * CT + c;
            acc[h][c][rt] = mma6(cur[0][j], cur[1][j], cur[2][j], avc[rt], acc[h][c][rt]);
          }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int p = 0; p < 3; ++p) avc[rt][p] = avn[rt][p];
    }
  };
#pragma unroll 1
  for (int half = 0; half < 2 * TAPS - 2; half += 2) {
    load1(bq1, half + 1);
    __builtin_amdgcn_sched_barrier(0);
    half1(half, bq0);
    load1(bq0, half + 2);
    __builtin_amdgcn_sched_barrier(0);
    half1(half + 1, bq1);
  }
  load1(bq1, 2 * TAPS - 1);
  __builtin_amdgcn_sched_barrier(0);
  half1(2 * TAPS - 2, bq0);
  half1(2 * TAPS - 1, bq1);
  stamp(2);

  // ---- phase-2 operands: the res/skip weights of the first two k-steps, bias, the h / out values the epilogue updates
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.W2) + (size_t)pw * KS * (NCTP * 64) + lane;
  auto load2 = [&](f32x4(&dst)[3][2 * NCT], int half) {  // own index tt * NCT + c  <->  packed tt * NCTP + sub * NCT + c
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2 * NCT; ++j) dst[p][j] = w2[p * ax.p2 + (size_t)(half * 2 * NCTP + (j / NCT) * NCTP + sub * NCT + j % NCT) * 64];
  };
  f32x4 cq0[3][2 * NCT], cq1[3][2 * NCT];
  load2(cq0, 0);
  load2(cq1, 1);
  f32x4 bv[NCT], old[RT][NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int n = 16 * NCTP * pw + 16 * (sub * NCT + c) + 4 * lq;
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

  // ---- gate -> the three planes of the activation tile (4 consecutive channels = 8 bytes of a row's slot)
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
      put4_x3(As, APL, 16 * rt + l15, (C / PW) * pw + 16 * (sub * CT + c) + 4 * lq, act);
    }
  __syncthreads();
  stamp(3);

  // ---- phase 2: res/skip, K = 128 from LDS; wave w owns columns [16 NCT w, 16 NCT (w + 1))
  f32x4 acc2[RT][NCT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc2[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto half2 = [&](int half, const f32x4(&cur)[3][2 * NCT]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      f32x4 av[RT][3];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int p = 0; p < 3; ++p) av[rt][p] = rows_frag(As + p * APL, 16 * rt + l15, 2 * half + tt);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc2[rt][c] = mma6(cur[0][tt * NCT + c], cur[1][tt * NCT + c], cur[2][tt * NCT + c], av[rt], acc2[rt][c]);
    }
  };
  half2(0, cq0);
  half2(1, cq1);
  stamp(4);

  if constexpr (!LAST) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int n = 16 * NCTP * pw + 16 * (sub * NCT + c) + 4 * lq;
      const bool to_h = n < C;
      const int col = to_h ? n : n - C;
      float* dst = to_h ? a.Hout : a.Out;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(dst + (long)(row0 + 16 * rt + l15) * C + col) = old[rt][c] + (acc2[rt][c] + bv[c]);
    }
    stamp(5);
    stamp(7);
    return;
  } else {
    // ---- tail: post + reverse coupling (+ the next block's pre); packed wave pw: mean / log-std tiles of channels [16 pw, 16 pw + 16).  With eight waves the first
    // wave of each pair runs it (a 16-channel tile pair per packed wave: nothing to split), the other one only keeps the barriers and writes its part of `out`.
    const bool tw = sub == 0;
    constexpr int KS4 = KS / 2;  // K = 64
    f32x4 pq[3][KS][2], rq[3][KS4][2];
    f32x4 pm = {0.f, 0.f, 0.f, 0.f}, ps = pm, zold[RT];
    f32x4 hb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int cc = 16 * pw + 4 * lq;
    if (tw) {
      const f32x4* w3 = reinterpret_cast<const f32x4*>(a.W3) + (size_t)pw * KS * (2 * 64) + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          pq[p][t][0] = w3[p * ax.p3 + (t * 2 + 0) * 64];
          pq[p][t][1] = w3[p * ax.p3 + (t * 2 + 1) * 64];
        }
      pm = *reinterpret_cast<const f32x4*>(a.b3m + cc);
      ps = *reinterpret_cast<const f32x4*>(a.b3s + cc);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) zold[rt] = *reinterpret_cast<const f32x4*>(a.Z + (long)(row0 + 16 * rt + l15) * a.ldz + a.zcol0 + cc);
    }
    __syncthreads();  // every wave has finished reading the gated activations
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) put4_x3(As, APL, 16 * rt + l15, 16 * NCTP * pw + 16 * (sub * NCT + c) + 4 * lq, old[rt][c] + (acc2[rt][c] + bv[c]));
    __syncthreads();
    if (tw) {
      f32x4 acc3[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc3[rt][0] = acc3[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KS; ++t) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          f32x4 av[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) av[p] = rows_frag(As + p * APL, 16 * rt + l15, t);
          acc3[rt][0] = mma6(pq[0][t][0], pq[1][t][0], pq[2][t][0], av, acc3[rt][0]);
          acc3[rt][1] = mma6(pq[0][t][1], pq[1][t][1], pq[2][t][1], av, acc3[rt][1]);
        }
      }
      if (a.tail > 1) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(a.W4) + (size_t)pw * KS4 * (2 * 64) + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int t = 0; t < KS4; ++t) {
            rq[p][t][0] = w4[p * ax.p4 + (t * 2 + 0) * 64];
            rq[p][t][1] = w4[p * ax.p4 + (t * 2 + 1) * 64];
          }
        hb[0] = *reinterpret_cast<const f32x4*>(a.b4 + 32 * pw + 4 * lq);
        hb[1] = *reinterpret_cast<const f32x4*>(a.b4 + 32 * pw + 16 + 4 * lq);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 mm = acc3[rt][0] + pm, ls = acc3[rt][1] + ps;
        f32x4 z1;
#pragma unroll
        for (int i = 0; i < 4; ++i) z1[i] = (zold[rt][i] - mm[i]) * __expf(-ls[i]);  // flow.py:209
        if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(a.Z + (long)(row0 + 16 * rt + l15) * a.ldz + a.zcol0 + cc) = z1;
        put4_x3(Hs, HPL, 16 * rt + l15, cc, z1);  // the conv tile is long dead: rows [0, ROWS) x channels [0, 64) of it now hold z1
      }
    }
    if (a.tail < 2) {
      stamp(5);
      stamp(7);
      return;
    }
    __syncthreads();
    if (tw) {
      f32x4 acc4[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc4[rt][0] = acc4[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KS4; ++t) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          f32x4 av[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) av[p] = rows_frag(Hs + p * HPL, 16 * rt + l15, t);
          acc4[rt][0] = mma6(rq[0][t][0], rq[1][t][0], rq[2][t][0], av, acc4[rt][0]);
          acc4[rt][1] = mma6(rq[0][t][1], rq[1][t][1], rq[2][t][1], av, acc4[rt][1]);
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(a.Hpre + (long)(row0 + 16 * rt + l15) * C + 32 * pw + 16 * c + 4 * lq) = acc4[rt][c] + hb[c];
    }
    stamp(5);
    stamp(7);
  }
}

// three planes of pack_fragments16's layout: out[p][((w * ksteps + s) * tiles + j) * 64 + lane][e] = term p of the exact bf16 split of row(w, j, lane & 15)[32 s + 8 (lane >> 4) + e]
inline std::vector<unsigned short> pack_fragments_x3(int waves, int ksteps, int tiles, const std::function<const float*(int, int, int)>& row,
                                                     void (*split3)(float, unsigned short*, unsigned short*, unsigned short*)) {
  const size_t plane = (size_t)waves * ksteps * tiles * 64 * 8;
  std::vector<unsigned short> out(3 * plane, 0);
  for (int w = 0; w < waves; ++w)
    for (int j = 0; j < tiles; ++j)
      for (int c = 0; c < 16; ++c) {
        const float* src = row(w, j, c);
        if (!src) continue;
        for (int s = 0; s < ksteps; ++s)
          for (int kq = 0; kq < 4; ++kq)
            for (int e = 0; e < 8; ++e) {
              const size_t i = ((((size_t)w * ksteps + s) * tiles + j) * 64 + kq * 16 + c) * 8 + e;
              split3(src[32 * s + 8 * kq + e], &out[i], &out[plane + i], &out[2 * plane + i]);
            }
      }
  return out;
}

}  // namespace stts
