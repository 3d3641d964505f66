// wn_block_x3_kernel<RT>: ONE launch per coupling layer of the reverse flow (models/flow.py:196-218) in fp32, as split-fp32 contractions
// (gemm.hip.h PREC_X3; wn_fused_x3.hip.h is the one-launch-per-WaveNet-layer form): the four WaveNet layers (flow.py:63-88: conv k5 -> gate ->
// res / skip), `post`, the reverse coupling and the next coupling layer's `pre`, with the residual stream h and the skip accumulator `out`
// ON CHIP for the whole block.  Structure = wn_block16_kernel's (wn_block16.hip.h), operands = three bf16 planes each, six MFMAs per product.
//
// Why: at B = 8 a WaveNet-layer launch is 22 us of which ~9 us are matrix work; launch ramp, the prologue that stages h, the stores of h and `out`
// and the start of the weight stream are paid 32 times per step.  Here they are paid 8 times, h_0 is read once and only z and the next h_0 are written.
//
// A block owns OUT = 16 RT - 16 output rows of one utterance and computes on 16 RT rows: each of the four k = 5 convs reaches 2 rows further, so the
// outermost rows of the tile go stale layer by layer and exactly the inner OUT rows are right after the fourth - the 2 x 8 halo rows inside the
// utterance are recomputed (bit-identically) by the neighbouring blocks, rows outside the utterance are the convs' zero padding at EVERY layer.
// RT = 3: 32 output rows (B = 8 x 3 s: 240 blocks, 1.5 x the matrix work of the per-layer form); RT = 4: 48 output rows (1.33 x; 134 KB of LDS).
// 4 waves, one per SIMD: wave w owns gate channels [32 w, 32 w + 32) in phase 1 and, in phase 2, the res channels AND the skip channels
// [32 w, 32 w + 32): `out` is a register accumulator across the layers, h (fp32) sits in LDS in the lane order of its accumulator tiles (each lane
// re-reads only what it wrote) and passes through the registers as the MFMAs' C operand (h_{i+1} = h_i + res_i).
#pragma once
#include "wn_fused_x3.hip.h"

namespace stts {

constexpr int kWnBlockX3Halo = 8;  // 4 layers x 2 rows

struct WnBlockX3Args {
  const float* Hin;  // h_0 = pre(z0) of this coupling layer, [rows, 128] fp32
  const int* seg_off;
  const unsigned short* W1[4];  // in_layers fragments, plane 0 of three: [4 waves][20 k-steps, tap-major][4 tiles: (tanh, sigmoid) x 2][64][8]
  const float* b1[4];           // [256] natural order
  const unsigned short* W2[4];  // res_skip fragments in BLOCK order: [4 waves][4 k-steps][4 tiles: res 32w, res 32w+16, skip 32w, skip 32w+16][64][8]; layer 3: the 2 skip tiles
  const float* b2[4];           // [256] natural order (res | skip); layer 3: [128] (skip)
  long p1, p2[4], p3, p4;       // f32x4 units between two planes
  const float* gate;            // style projections [n_utt][ld_gate]
  int ld_gate, gcol0[4];
  int tail;                     // 1: post + coupling; 2: + the next coupling layer's pre
  const unsigned short* W3;     // post (mean | logstd) fragments [4][4][2][64][8]
  const float* b3m;
  const float* b3s;
  float* Z;
  int ldz, zcol0;
  const unsigned short* W4;     // next pre, fragments [4][2][2][64][8]
  const float* b4;
  float* Hpre;                  // next coupling layer's h_0 (a different buffer than Hin: neighbouring blocks still read their halo from Hin)
};

template <int RT>
__global__ void __launch_bounds__(256) wn_block_x3_kernel(const WnBlockX3Args a) {
  constexpr int TROWS = 16 * RT, OUT = TROWS - 2 * kWnBlockX3Halo, C = kWnC, NW = kWnWaves, CT = C / NW / 16, TAPS = 5, PAD = 2, KS = C / 32;
  static_assert(CT == 2 && NW == 4 && OUT > 0, "tile geometry");
  constexpr int HPL = (TROWS + 2 * PAD) * 16, APL = TROWS * 16;  // f32x4 slots of one plane
  __shared__ f32x4 Hs[3 * HPL];            // h rows [-2, TROWS + 2) of the tile, three planes; later the coupled half of z
  __shared__ f32x4 As[3 * APL];            // gated activations; later the finished `out`
  __shared__ f32x4 Hf[NW * RT * CT * 64];  // the residual stream in fp32: [wave][row tile][channel tile][lane]

  const int utt = blockIdx.y;
  const int lo = a.seg_off[utt], hi = a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * OUT;
  if (row0 >= hi) return;
  const int rbase = row0 - kWnBlockX3Halo;  // global row of tile row 0
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int ch0 = (C / NW) * w + 4 * lq;  // this lane's first channel of tile c: ch0 + 16 c

  // rows of this lane: tile row 16 rt + l15; inside the utterance?  (outside = the convs' zero padding at every layer)
  unsigned in_mask = 0, out_mask = 0;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = 16 * rt + l15, g = rbase + r;
    if (g >= lo && g < hi) in_mask |= 1u << rt;
    if (r >= kWnBlockX3Halo && r < kWnBlockX3Halo + OUT && g < hi) out_mask |= 1u << rt;
  }

  // the two pad rows on either side of the conv tile stay zero (all three planes)
  if (tid < 4 * 16) {
    const int r = tid >> 4, sl = tid & 15;
    const int o = (r < 2 ? r : TROWS + r) * 16 + sl;
    Hs[o] = Hs[o + HPL] = Hs[o + 2 * HPL] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- residual stream (fp32, LDS) and skip accumulator (registers) of this wave's 32 + 32 channels, all rows of the tile
  f32x4* const hf = Hf + (size_t)w * (RT * CT * 64) + lane;  // tile (rt, c) at hf[(rt * CT + c) * 64]
  f32x4 oacc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const long g = min(max(rbase + 16 * rt + l15, lo), hi - 1);
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.Hin + g * C + ch0 + 16 * c);
      hf[(rt * CT + c) * 64] = ((in_mask >> rt) & 1) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      oacc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  auto rows_frag = [&](const f32x4* tile, int row, int kstep) { return tile[row * 16 + ((4 * kstep + lq) ^ (row & 15))]; };
  auto mma6 = [&](const f32x4 w0, const f32x4 w1_, const f32x4 w2_, const f32x4 (&x)[3], f32x4 acc) {
    acc = mfma16x16<PREC_BF16>(w2_, x[0], acc);
    acc = mfma16x16<PREC_BF16>(w0, x[2], acc);
    acc = mfma16x16<PREC_BF16>(w1_, x[1], acc);
    acc = mfma16x16<PREC_BF16>(w1_, x[0], acc);
    acc = mfma16x16<PREC_BF16>(w0, x[1], acc);
    acc = mfma16x16<PREC_BF16>(w0, x[0], acc);
    return acc;
  };

  // tail operands (post weights / bias, the coupled half of z): requested inside the LAST layer, when its conv accumulators are dead
  f32x4 pq[3][KS][2], pm, ps, zold[RT];
  const int cc = 16 * w + 4 * lq;
  auto load_tail = [&]() {
    const f32x4* w3 = reinterpret_cast<const f32x4*>(a.W3) + (size_t)w * KS * (2 * 64) + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        pq[p][t][0] = w3[p * a.p3 + (t * 2 + 0) * 64];
        pq[p][t][1] = w3[p * a.p3 + (t * 2 + 1) * 64];
      }
    pm = *reinterpret_cast<const f32x4*>(a.b3m + cc);
    ps = *reinterpret_cast<const f32x4*>(a.b3s + cc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const long g = min(max(rbase + 16 * rt + l15, lo), hi - 1);
      zold[rt] = *reinterpret_cast<const f32x4*>(a.Z + g * a.ldz + a.zcol0 + cc);
    }
  };
  constexpr int T1 = KS * 2 * CT;  // in_layers fragments per tap and plane: 4 k-steps x (tanh, sigmoid) x CT = 16
  constexpr int TH = T1 / 2;       // ... per half tap (2 k-steps): the unit the weight stream runs ahead by

  // one WaveNet layer; LAST (layer 3): res_skip has the skip half only
  auto layer = [&](const int l, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const f32x4* w1 = reinterpret_cast<const f32x4*>(a.W1[l]) + (size_t)w * TAPS * (T1 * 64) + lane;
    // the weight stream runs ONE half tap (6 x 24 x RT MFMAs: >= 1 us at RT = 3) ahead of the multiplies in two register buffers of three planes
    f32x4 bq0[3][TH], bq1[3][TH];
    auto load1 = [&](f32x4(&dst)[3][TH], int half) {  // half = 2 tap + (0 | 1)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < TH; ++j) dst[p][j] = w1[p * a.p1 + (size_t)(half * TH + j) * 64];
    };
    load1(bq0, 0);
    // gate operands of this lane's channels: requested together with the first weights, BEFORE the conv tile is written
    f32x4 ba[CT], bb[CT], ga[CT], gb[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int ch = ch0 + 16 * c;
      ba[c] = *reinterpret_cast<const f32x4*>(a.b1[l] + ch);
      ga[c] = *reinterpret_cast<const f32x4*>(a.gate + (long)utt * a.ld_gate + a.gcol0[l] + ch);
      bb[c] = *reinterpret_cast<const f32x4*>(a.b1[l] + C + ch);
      gb[c] = *reinterpret_cast<const f32x4*>(a.gate + (long)utt * a.ld_gate + a.gcol0[l] + C + ch);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- h (three planes) -> conv tile; rows outside the utterance are zero
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < CT; ++c)
        put4_x3(Hs, HPL, PAD + 16 * rt + l15, ch0 + 16 * c, ((in_mask >> rt) & 1) ? hf[(rt * CT + c) * 64] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      ba[c] += ga[c];
      bb[c] += gb[c];
    }
    __syncthreads();

    // ---- phase 1: conv k5, K = 5 taps x 128 channels; acc[half][c][rt]: channels 32 w + 16 c + 4 lq + i, tile row 16 rt + l15
    f32x4 acc[2][CT][RT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[h][c][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto half_tap = [&](int half, const f32x4(&cur)[3][TH]) {
      const int tap = half >> 1, t0 = (half & 1) * 2;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        f32x4 av[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int p = 0; p < 3; ++p) av[rt][p] = rows_frag(Hs + p * HPL, 16 * rt + l15 + tap, t0 + tt);  // tile row r + tap - 2 = conv-tile row r + tap
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              const int j = (tt * 2 + h) * CT + c;
              acc[h][c][rt] = mma6(cur[0][j], cur[1][j], cur[2][j], av[rt], acc[h][c][rt]);
            }
      }
    };
#pragma unroll 1
    for (int half = 0; half < 2 * TAPS - 2; half += 2) {  // (rolled: fully unrolled, the ten half taps' loads and fragments spill)
      load1(bq1, half + 1);
      __builtin_amdgcn_sched_barrier(0);
      half_tap(half, bq0);
      load1(bq0, half + 2);
      __builtin_amdgcn_sched_barrier(0);
      half_tap(half + 1, bq1);
    }
    load1(bq1, 2 * TAPS - 1);
    __builtin_amdgcn_sched_barrier(0);
    half_tap(2 * TAPS - 2, bq0);
    half_tap(2 * TAPS - 1, bq1);

    // ---- phase-2 weights of this layer (res | skip in block order; layer 3: skip only) + their bias
    constexpr int NCT = LAST ? CT : 2 * CT;
    const f32x4* w2 = reinterpret_cast<const f32x4*>(a.W2[l]) + (size_t)w * KS * (NCT * 64) + lane;
    f32x4 cq2[3][KS][NCT];
    f32x4 bh[CT], bo[CT];
    auto load2 = [&]() {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
          for (int c = 0; c < NCT; ++c) cq2[p][t][c] = w2[p * a.p2[l] + (t * NCT + c) * 64];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        bh[c] = LAST ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.b2[l] + ch0 + 16 * c);
        bo[c] = *reinterpret_cast<const f32x4*>(a.b2[l] + (LAST ? 0 : C) + ch0 + 16 * c);
      }
    };
    // ---- gate -> the three planes of the activation tile
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 va = acc[0][c][rt] + ba[c], vb = acc[1][c][rt] + bb[c];
        f32x4 act;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (the per-layer kernels' form of the gate, so that both forms of the flow round alike)
          const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390082f * va[i]) + 1.0f);
          act[i] = th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * vb[i]));
        }
        put4_x3(As, APL, 16 * rt + l15, ch0 + 16 * c, act);
      }
    __builtin_amdgcn_sched_barrier(0);  // (the phase-2 operands are fetched once the conv accumulators are dead, not before)
    load2();
    if constexpr (LAST) load_tail();
    __syncthreads();

    // ---- phase 2: res / skip, K = 128 from LDS.  Skip: onto the register accumulator.  Res: the tile of h passes through the registers
    // as the MFMA chain's C operand, one row tile at a time (LDS -> h + bias -> 4 k-steps -> LDS).
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 av[KS][3];
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) av[t][p] = rows_frag(As + p * APL, 16 * rt + l15, t);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        f32x4 o = oacc[rt][c] + bo[c];
#pragma unroll
        for (int t = 0; t < KS; ++t) o = mma6(cq2[0][t][LAST ? c : CT + c], cq2[1][t][LAST ? c : CT + c], cq2[2][t][LAST ? c : CT + c], av[t], o);
        oacc[rt][c] = o;
        if constexpr (!LAST) {
          f32x4 h = hf[(rt * CT + c) * 64] + bh[c];
#pragma unroll
          for (int t = 0; t < KS; ++t) h = mma6(cq2[0][t][c], cq2[1][t][c], cq2[2][t][c], av[t], h);
          hf[(rt * CT + c) * 64] = h;
        }
      }
    }
    // (the next layer's conv tile is written after this point; every wave passed the barrier above, so nobody still reads Hs.  Its
    //  gate writes As only after the barrier that follows the Hs writes, which every wave reaches after finishing this phase 2.)
  };
  // (layers 0-2 as a rolled loop: unrolled, the scheduler overlaps the layers' live ranges and spills)
#pragma unroll 1
  for (int l = 0; l < 3; ++l) layer(l, std::false_type{});
  layer(3, std::true_type{});

  // ---- tail: post + reverse coupling (+ the next coupling layer's pre); wave w: mean / log-std tiles of channels [16 w, 16 w + 16)
  __syncthreads();  // every wave has finished reading the gated activations of the last layer
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) put4_x3(As, APL, 16 * rt + l15, ch0 + 16 * c, oacc[rt][c]);  // the finished `out`
  __syncthreads();
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
  constexpr int KS4 = KS / 2;  // K = 64
  const f32x4* w4 = reinterpret_cast<const f32x4*>(a.W4) + (size_t)w * KS4 * (2 * 64) + lane;
  f32x4 rq[3][KS4][2];
  f32x4 hb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (a.tail > 1) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < KS4; ++t) {
        rq[p][t][0] = w4[p * a.p4 + (t * 2 + 0) * 64];
        rq[p][t][1] = w4[p * a.p4 + (t * 2 + 1) * 64];
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
    if ((out_mask >> rt) & 1) *reinterpret_cast<f32x4*>(a.Z + (long)(rbase + 16 * rt + l15) * a.ldz + a.zcol0 + cc) = z1;
    put4_x3(Hs, HPL, 16 * rt + l15, cc, z1);  // the conv tile is dead: rows [0, TROWS) x channels [0, 64) of it now hold z1
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
      if ((out_mask >> rt) & 1) *reinterpret_cast<f32x4*>(a.Hpre + (long)(rbase + 16 * rt + l15) * C + 32 * w + 16 * c + 4 * lq) = acc4[rt][c] + hb[c];
}

}  // namespace stts
