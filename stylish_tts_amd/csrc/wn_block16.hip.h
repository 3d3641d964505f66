// wn_block16_kernel<PREC>: ONE launch per coupling layer of the reverse flow (models/flow.py:196-218) for the 16-bit operand modes at
// large batches: the four WaveNet layers (flow.py:63-88: conv k5 -> gate -> res / skip), `post`, the reverse coupling and the next
// coupling layer's `pre`, with the residual stream h and the skip accumulator `out` ON CHIP for the whole block.
//
// Why: with one launch per WaveNet layer (wn_fused16_kernel) every layer reads h, writes h, reads `out` and writes `out` - 126 MB of
// fp32 rows per layer at B = 64 x 3 s against 24 GFLOP, i.e. the flow stage ran at HBM speed (48 us per layer, 1.5 ms per step).  Here
// h_0 is read once (+ 12.5 % halo), `out` never leaves the registers, and only z and the next h_0 are written: ~5 x less traffic.
//
// A block owns 128 output rows of one utterance and computes on 144 = 128 + 2 x 8 rows: each of the four k = 5 convs reaches 2 rows
// further, so the outermost rows of the tile go stale layer by layer and exactly the inner 128 are right after the fourth - the
// halo rows inside the utterance are recomputed (bit-identically) by the neighbouring blocks, rows outside the utterance are the
// convs' zero padding at EVERY layer (masked whenever h enters the conv tile).
// 4 waves, one per SIMD (~330 registers): wave w owns gate channels [32 w, 32 w + 32) in phase 1 and, in phase 2, the res channels
// AND the skip channels [32 w, 32 w + 32): the skip sum `out` is a register accumulator that simply keeps accumulating across the
// layers, the residual stream h (fp32) sits in LDS in the lane order of its accumulator tiles (each lane re-reads only what it wrote:
// no barrier) and passes through the registers as the MFMAs' C operand (h_{i+1} = h_i + res_i).  Weights in MFMA-fragment order straight from global memory (L2) into registers, as in
// wn_fused16_kernel; the res / skip matrices are packed in this kernel's own tile order (pack_wn_fused: H2b).
// Rounding points = those of wn_fused16_kernel / wn_layer_kernel<PREC> (what the rounded oracle pins): h when it enters the conv
// tile, the gated activations, the finished `out` before `post`, the coupled half of z before `pre`; everything else fp32.
#pragma once
#include "wn_fused16.hip.h"

namespace stts {

constexpr int kWnBlockRows = 128;  // output rows per block
constexpr int kWnBlockHalo = 8;    // 4 layers x 2 rows

struct WnBlock16Args {
  const float* Hin;  // h_0 = pre(z0) of this coupling layer, [rows, 128] fp32
  const int* seg_off;
  const unsigned short* W1[4];  // in_layers fragments, as wn_fused16_kernel: [4 waves][20 k-steps, tap-major][4 tiles: (tanh, sigmoid) x 2][64][8]
  const float* b1[4];           // [256] natural order
  const unsigned short* W2[4];  // res_skip fragments in BLOCK order: [4 waves][4 k-steps][4 tiles: res 32w, res 32w+16, skip 32w, skip 32w+16][64][8]; layer 3: the 2 skip tiles
  const float* b2[4];           // [256] natural order (res | skip); layer 3: [128] (skip)
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

template <int PREC>
__global__ void __launch_bounds__(256) wn_block16_kernel(const WnBlock16Args a) {
  constexpr int RT = 9, TROWS = 16 * RT, C = kWnC, NW = kWnWaves, CT = C / NW / 16, TAPS = 5, PAD = 2, KS = C / 32;
  static_assert(TROWS == kWnBlockRows + 2 * kWnBlockHalo && CT == 2 && NW == 4, "tile geometry");
  // 16-bit row tiles, 256 bytes per row = 16 slots of 8 channels, slot index XORed with (row & 15) (wn_fused16_kernel's layout)
  __shared__ f32x4 Hs[(TROWS + 2 * PAD) * 16];  // h rows [-2, TROWS + 2) of the tile; later the coupled half of z
  __shared__ f32x4 As[TROWS * 16];              // gated activations; later the finished `out`
  __shared__ f32x4 Hf[NW * RT * CT * 64];       // the residual stream in fp32: [wave][row tile][channel tile][lane], 72 KB

  const int utt = blockIdx.y;
  const int lo = a.seg_off[utt], hi = a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * kWnBlockRows;
  if (row0 >= hi) return;
  const int rbase = row0 - kWnBlockHalo;  // global row of tile row 0
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int ch0 = (C / NW) * w + 4 * lq;  // this lane's first channel of tile c: ch0 + 16 c

  // rows of this lane: tile row 16 rt + l15; inside the utterance?  (outside = the convs' zero padding at every layer)
  unsigned in_mask = 0, out_mask = 0;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = 16 * rt + l15, g = rbase + r;
    if (g >= lo && g < hi) in_mask |= 1u << rt;
    if (r >= kWnBlockHalo && r < kWnBlockHalo + kWnBlockRows && g < hi) out_mask |= 1u << rt;
  }

  // the two pad rows on either side of the conv tile stay zero
  if (tid < 4 * 16) {
    const int r = tid >> 4, sl = tid & 15;
    Hs[(r < 2 ? r : TROWS + r) * 16 + sl] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- residual stream (fp32, LDS) and skip accumulator (registers) of this wave's 32 + 32 channels, all 144 rows
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

  auto put4 = [&](f32x4* tile, int row, int ch, const f32x4 v) {  // 4 consecutive channels = 8 bytes of a row's slot
    u32x2* p = reinterpret_cast<u32x2*>(tile + row * 16 + ((ch >> 3) ^ (row & 15)));
    p[(ch >> 2) & 1] = round4<PREC>(v);
  };
  // B operand of the 16x16x32 MFMA from a row tile: lane (row l15, k-group lq) reads 8 consecutive channels of its row
  auto rows_frag = [&](const f32x4* tile, int row, int kstep) { return tile[row * 16 + ((4 * kstep + lq) ^ (row & 15))]; };

  // tail operands (post weights / bias, the coupled half of z): requested inside the LAST layer, when its conv accumulators are dead, so that they
  // arrive while its res / skip phase runs (requested at the tail they were ~4 us of exposed round trips per block)
  f32x4 pq[KS][2], pm, ps, zold[RT];
  const int cc = 16 * w + 4 * lq;
  auto load_tail = [&]() {
    const f32x4* w3 = reinterpret_cast<const f32x4*>(a.W3) + (size_t)w * KS * (2 * 64) + lane;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      pq[t][0] = w3[(t * 2 + 0) * 64];
      pq[t][1] = w3[(t * 2 + 1) * 64];
    }
    pm = *reinterpret_cast<const f32x4*>(a.b3m + cc);
    ps = *reinterpret_cast<const f32x4*>(a.b3s + cc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const long g = min(max(rbase + 16 * rt + l15, lo), hi - 1);
      zold[rt] = *reinterpret_cast<const f32x4*>(a.Z + g * a.ldz + a.zcol0 + cc);
    }
  };
  constexpr int T1 = KS * 2 * CT;  // in_layers fragments per tap: 4 k-steps x (tanh, sigmoid) x CT = 16
  constexpr int TH = T1 / 2;       // ... per half tap (2 k-steps): the unit the weight stream runs ahead by

  // one WaveNet layer; LAST (layer 3): res_skip has the skip half only
  auto layer = [&](const int l, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const f32x4* w1 = reinterpret_cast<const f32x4*>(a.W1[l]) + (size_t)w * TAPS * (T1 * 64) + lane;
    // the weight stream runs TWO half taps (~1 us of MFMAs) ahead of the multiplies in three register buffers: with one half tap ahead
    // (0.48 us) every half tap waited for L2 (this wave is alone on its SIMD: nothing else hides the latency)
    // (three half taps ahead in four buffers, and the phase-2 weights requested before the gate, measured: 166 and 143 us per launch against 143 - spills / no gain)
    f32x4 bq[3][TH];
    auto load1 = [&](f32x4(&dst)[TH], int half) {  // half = 2 tap + (0 | 1)
#pragma unroll
      for (int j = 0; j < TH; ++j) dst[j] = w1[(half * TH + j) * 64];
    };
    load1(bq[0], 0);
    load1(bq[1], 1);
    // gate operands of this lane's channels: requested together with the first weights, BEFORE the conv tile is written (every global round trip of
    // the layer's start is then in flight at once; requested after the LDS pass they cost ~1.5 us per layer, block timeline of round 3)
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
    // ---- h (rounded) -> conv tile; rows outside the utterance are zero
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < CT; ++c)
        put4(Hs, PAD + 16 * rt + l15, ch0 + 16 * c, ((in_mask >> rt) & 1) ? hf[(rt * CT + c) * 64] : f32x4{0.f, 0.f, 0.f, 0.f});
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
    // the B fragments of a step (9 row tiles of the conv tile at one tap / k-step) are fetched one step ahead of the 36 MFMAs that consume them
    f32x4 avc[RT];
    auto frags = [&](f32x4(&dst)[RT], int step) {  // step = 2 half + tt = 4 tap + k-step
      const int tap = step >> 2, t = step & 3;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) dst[rt] = rows_frag(Hs, 16 * rt + l15 + tap, t);  // tile row r + tap - 2 = conv-tile row r + tap
    };
    frags(avc, 0);
    auto half_tap = [&](int half, const f32x4(&cur)[TH]) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int step = 2 * half + tt;
        f32x4 avn[RT];
        frags(avn, min(step + 1, 4 * TAPS - 1));  // (the last step re-reads itself: no branch in the loop body)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[h][c][rt] = mfma16x16<PREC>(cur[(tt * 2 + h) * CT + c], avc[rt], acc[h][c][rt]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) avc[rt] = avn[rt];
      }
    };
    static_assert(2 * TAPS == 10, "the rolled loop below covers nine half taps in threes + one");
#pragma unroll 1
    for (int h3 = 0; h3 < 9; h3 += 3) {  // (rolled: fully unrolled, the ten half taps' loads and fragments spill)
      load1(bq[2], h3 + 2);
      __builtin_amdgcn_sched_barrier(0);
      half_tap(h3, bq[0]);
      load1(bq[0], h3 + 3);
      __builtin_amdgcn_sched_barrier(0);
      half_tap(h3 + 1, bq[1]);
      if (h3 + 4 < 2 * TAPS) load1(bq[1], h3 + 4);
      __builtin_amdgcn_sched_barrier(0);
      half_tap(h3 + 2, bq[2]);
    }
    half_tap(9, bq[0]);

    // ---- phase-2 weights of this layer (res | skip in block order; layer 3: skip only) + their bias
    constexpr int NCT = LAST ? CT : 2 * CT;
    const f32x4* w2 = reinterpret_cast<const f32x4*>(a.W2[l]) + (size_t)w * KS * (NCT * 64) + lane;
    f32x4 cq2[KS][NCT];
    f32x4 bh[CT], bo[CT];
    auto load2 = [&]() {
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) cq2[t][c] = w2[(t * NCT + c) * 64];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        bh[c] = LAST ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.b2[l] + ch0 + 16 * c);
        bo[c] = *reinterpret_cast<const f32x4*>(a.b2[l] + (LAST ? 0 : C) + ch0 + 16 * c);
      }
    };
    // ---- gate -> 16-bit activations in LDS
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 va = acc[0][c][rt] + ba[c], vb = acc[1][c][rt] + bb[c];
        f32x4 act;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // tanh(a) sigmoid(b) = (E - 1) / ((E + 1) (1 + F)), E = e^{2a}, F = e^{-b}: two exp2 and ONE rcp (quarter-rate instructions are what
          // this phase is made of); a is clamped where tanh is 1 in fp32 so that E stays finite
          const float E = __builtin_amdgcn_exp2f(2.885390082f * fminf(va[i], 15.0f));
          const float F = __builtin_amdgcn_exp2f(-1.442695041f * vb[i]);
          act[i] = (E - 1.0f) * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
        }
        put4(As, 16 * rt + l15, ch0 + 16 * c, act);
      }
    __builtin_amdgcn_sched_barrier(0);  // (the phase-2 operands are fetched once the conv accumulators are dead, not before)
    load2();
    if constexpr (LAST) load_tail();
    __syncthreads();

    // ---- phase 2: res / skip, K = 128 from LDS.  Skip: onto the register accumulator.  Res: the tile of h passes through the registers
    // as the MFMA chain's C operand, one row tile at a time (LDS -> h + bias -> 4 k-steps -> LDS).
    f32x4 av[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int t = 0; t < KS; ++t) av[rt][t] = rows_frag(As, 16 * rt + l15, t);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        f32x4 o = oacc[rt][c] + bo[c];
#pragma unroll
        for (int t = 0; t < KS; ++t) o = mfma16x16<PREC>(cq2[t][LAST ? c : CT + c], av[rt][t], o);
        oacc[rt][c] = o;
        if constexpr (!LAST) {
          f32x4 h = hf[(rt * CT + c) * 64] + bh[c];
#pragma unroll
          for (int t = 0; t < KS; ++t) h = mfma16x16<PREC>(cq2[t][c], av[rt][t], h);
          hf[(rt * CT + c) * 64] = h;
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
    for (int rt = 0; rt < RT; ++rt) put4(As, 16 * rt + l15, ch0 + 16 * c, oacc[rt][c]);  // the finished `out`, rounded
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
    if ((out_mask >> rt) & 1) *reinterpret_cast<f32x4*>(a.Z + (long)(rbase + 16 * rt + l15) * a.ldz + a.zcol0 + cc) = z1;
    put4(Hs, 16 * rt + l15, cc, z1);  // the conv tile is dead: rows [0, TROWS) x channels [0, 64) of it now hold z1
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
      if ((out_mask >> rt) & 1) *reinterpret_cast<f32x4*>(a.Hpre + (long)(rbase + 16 * rt + l15) * C + 32 * w + 16 * c + 4 * lq) = acc4[rt][c] + hb[c];
}

}  // namespace stts
