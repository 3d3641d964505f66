// wn_fused_kernel<M, LAST>: one WaveNet layer of the reverse flow in one launch (models/flow.py:63-88), fp32, with the
// k = 5 convolution in Toom-Cook / Winograd form F(M, 5) over time and every weight streamed STRAIGHT INTO REGISTERS.
//
//   x_in = wn-conv1d k5 (h)                      128 -> 256        flow.py:72
//   acts = tanh(x_in[:128] + g[:128]) * sigmoid(x_in[128:] + g[128:])      fused_add_tanh_sigmoid_multiply, flow.py:7-14
//   rs   = wn-Linear(acts)                       128 -> 256 (last layer: 128)   flow.py:78
//   h   += rs[:128] ; out += rs[128:]            (last layer: out += rs)        flow.py:80-87
//   LAST: m, logs = post(out) ; z1 = (z1 - m) exp(-logs)  (flow.py:199-211) ; h' = pre(z1) of the next coupling layer (:188-190)
//
// Why this shape.  At the benchmark batch (8 x 3 s = 7 680 frames) a CU owns ~30 time rows, so every block needs ALL of a
// layer's weights (0.8 MB) for very few rows: the layer is a "skinny" contraction whose cost is set by (a) MFMA issue,
// (b) the L2 -> CU weight stream and (c) everything that is not the K loop.  The first fused kernel (wn_layer.hip.h)
// staged weights global -> VGPR -> LDS with a barrier per 32-channel chunk and split K over four wave groups, whose
// partial sums were exchanged through LDS twice per layer: 35.5 us per launch, 0.51 of the fp32 MFMA roof.  Here:
//   * a block is 16 GROUPS of M consecutive rows (M = 2: 32 rows, M = 4: 64 rows); the conv runs as F(M, 5): M outputs
//     from M + 4 inputs with M + 4 products per channel instead of 5 M  (M = 2: x 0.6, M = 4: x 0.4 of the multiplies).
//     The transformed input B^T d of the WHOLE block ([M + 4 components][16 groups][128 channels], 48 / 64 KB) is built
//     once in the prologue and kept in LDS in MFMA-fragment order, so the main loop has no barrier at all;
//   * v_mfma_f32_16x16x4_f32 tiles: rows = the 16 groups, cols = 16 output channels.  Wave w of 4 (one per SIMD, up to 512
//     registers each) owns the tanh tiles and the sigmoid tiles of channels [32 w, 32 w + 32) for all components: the output
//     transform A^T m, the gate and the res/skip split are register-local (no K split, no exchange);
//   * no two waves share a weight, so LDS would only be a detour: the host packs every matrix in fragment order
//     ([wave][16-channel block][tile][lane][4 floats], pack_fragments below) and a wave reads its B operands with
//     coalesced 16-byte global loads one block ahead, one load between every four MFMAs (1 KB per wave instruction,
//     L2-resident: all blocks stream the same 0.8 MB at the same time, measured 117 GB/s per CU = 0.8 MB in 6.7 us);
//   * res/skip (K = 128 from the gated activations in LDS) and, on the last layer of a coupling block, post + coupling +
//     the next block's pre run the same way on small 16x16 tiles.
// Numerics: F(2,5) / F(4,5) with the points {0, +-1, 2, -1/2 (, -2, 1/2), inf}: fp32 error ~7e-7 / 1.6e-6 of the conv's
// output scale (direct MFMA chain: 1.3e-7), matrices built in double and self-checked on the host (wn_fused_matrices).
// 16-bit operand modes keep wn_layer_kernel (rounded operands in the Winograd domain would not be the reference's
// arithmetic rounded at the contraction inputs, which is what the rounded oracle pins).
#pragma once
#include <cmath>
#include <functional>
#include <vector>

#include "gemm.hip.h"

namespace stts {

constexpr int kWnC = 128;    // flow hidden channels (dec_hidden / 4)
constexpr int kWnRowPad = 128;  // rows of slack the kernels may READ past the last utterance in Hin / Out / Z (a whole block)
constexpr int kWnWaves = 4;  // waves per block: one per SIMD (two per SIMD ran phase 1 at 62 % of the MFMA rate, one at 90 %: tools/probes/p1_probe.hip)

// F(M, 5) transform matrices for the points {0, 1, -1, 2, -1/2 (, -2, 1/2), inf}: exact rationals, compile-time constants so
// that the zeros and ones fold away in the kernel; the host regenerates them (wn_fused_matrices, which also yields the
// weight transform G) and refuses to pack weights if the two disagree.
template <int M>
struct WnConst;
template <>
struct WnConst<1> {  // "F(1,5)": the direct form - component j is the input row at offset j - 2, its plane is tap j (16-row blocks for the smallest batches)
  static constexpr double Bt[5][5] = {{1, 0, 0, 0, 0}, {0, 1, 0, 0, 0}, {0, 0, 1, 0, 0}, {0, 0, 0, 1, 0}, {0, 0, 0, 0, 1}};
  static constexpr double At[1][5] = {{1, 1, 1, 1, 1}};
};
template <>
struct WnConst<2> {
  static constexpr double Bt[6][6] = {{1.0, 1.5, -2.0, -1.5, 1.0, 0.0},
                                      {0.0, 1.0 / 3.0, 5.0 / 6.0, 1.0 / 6.0, -1.0 / 3.0, 0.0},
                                      {0.0, 1.0 / 3.0, 1.0 / 6.0, -5.0 / 6.0, 1.0 / 3.0, 0.0},
                                      {0.0, -1.0 / 30.0, -1.0 / 15.0, 1.0 / 30.0, 1.0 / 15.0, 0.0},
                                      {0.0, -32.0 / 15.0, 16.0 / 15.0, 32.0 / 15.0, -16.0 / 15.0, 0.0},
                                      {0.0, 1.0, 1.5, -2.0, -1.5, 1.0}};
  static constexpr double At[2][6] = {{1.0, 1.0, 1.0, 1.0, 1.0, 0.0}, {0.0, 1.0, -1.0, 2.0, -0.5, 1.0}};
};
template <>
struct WnConst<4> {
  static constexpr double Bt[8][8] = {{1.0, 0.0, -5.25, 0.0, 5.25, 0.0, -1.0, 0.0},
                                      {0.0, -2.0 / 9.0, -2.0 / 9.0, 17.0 / 18.0, 17.0 / 18.0, -2.0 / 9.0, -2.0 / 9.0, 0.0},
                                      {0.0, 2.0 / 9.0, -2.0 / 9.0, -17.0 / 18.0, 17.0 / 18.0, 2.0 / 9.0, -2.0 / 9.0, 0.0},
                                      {0.0, 1.0 / 180.0, 1.0 / 360.0, -1.0 / 36.0, -1.0 / 72.0, 1.0 / 45.0, 1.0 / 90.0, 0.0},
                                      {0.0, -64.0 / 45.0, 128.0 / 45.0, 16.0 / 9.0, -32.0 / 9.0, -16.0 / 45.0, 32.0 / 45.0, 0.0},
                                      {0.0, -1.0 / 180.0, 1.0 / 360.0, 1.0 / 36.0, -1.0 / 72.0, -1.0 / 45.0, 1.0 / 90.0, 0.0},
                                      {0.0, 64.0 / 45.0, 128.0 / 45.0, -16.0 / 9.0, -32.0 / 9.0, 16.0 / 45.0, 32.0 / 45.0, 0.0},
                                      {0.0, -1.0, 0.0, 5.25, 0.0, -5.25, 0.0, 1.0}};
  static constexpr double At[4][8] = {{1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0},
                                      {0.0, 1.0, -1.0, 2.0, -0.5, -2.0, 0.5, 0.0},
                                      {0.0, 1.0, 1.0, 4.0, 0.25, 4.0, 0.25, 0.0},
                                      {0.0, 1.0, -1.0, 8.0, -0.125, -8.0, 0.125, 1.0}};
};

constexpr int kWnSegInline = 64;  // utterance offsets travel in the kernel arguments up to this batch size (one dependent load less)

template <int M>
struct WnFusedArgs {
  static constexpr int NC = M + 4;
  const float* Hin;    // [rows, 128]
  float* Hout;         // [rows, 128] (unused when LAST)
  float* Out;          // [rows, 128] (read when out_acc; not written when LAST)
  const int* seg_off;
  const float* W1;     // F(M,5) planes of in_layers, fragments [4 waves][8 blocks][NC][2 halves][2 tiles][64 lanes][4]
  const float* b1;     // [256] natural order: tanh rows 0..127 | sigmoid rows 128..255
  const float* W2;     // res_skip, fragments [4][8][4 (LAST: 2)][64][4]
  const float* b2;     // [256] (LAST: [128])
  const float* gate;   // [n_utt][ld_gate]
  int ld_gate, gcol0;
  int out_acc;         // accumulate into Out (0 on the first layer of a coupling block)
  int tail;            // LAST only: 1 = post + coupling, 2 = additionally the next block's pre
  const float* W3;     // post (mean | logstd), fragments [4][8][2][64][4]
  const float* b3m;    // [64]
  const float* b3s;    // [64]
  float* Z;            // [rows, ldz]; columns [zcol0, zcol0 + 64) are updated in place
  int ldz, zcol0;
  const float* W4;     // next block's pre, fragments [4][4][2][64][4]
  const float* b4;     // [128]
  float* Hpre;         // [rows, 128]
  long long* dbg;      // per-block phase stamps, written only by a -DSTTS_WN_TRACE build (diagnostics)
  int n_inline;        // > 0: seg_inline holds the n_inline + 1 utterance offsets (seg_off is not read)
  int seg_inline[kWnSegInline + 1];
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) (register arrays indexed by I stay in registers)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int M, bool LAST>
__global__ void __launch_bounds__(256) wn_fused_kernel(const WnFusedArgs<M> a) {
  constexpr int NC = M + 4, ROWS = 16 * M, RT = M, C = kWnC, KB = C / 16;  // KB: 16-channel blocks of a 128-channel contraction
  constexpr int NW = kWnWaves, CT = C / NW / 16;                             // waves, 16-channel tiles of a wave's channel range
  constexpr int NCT = LAST ? CT : 2 * CT;                                    // res/skip column tiles per wave
  __shared__ f32x4 Af[NC * KB * 64];  // B^T d in fragment order [component][block][lane]; later the coupled half of z [RT][4][lane]
  __shared__ f32x4 A2[RT * KB * 64];  // gated activations, then the finished `out` tile: [row tile][block][lane]

  const int utt = blockIdx.y;
  const int lo = a.n_inline ? a.seg_inline[utt] : a.seg_off[utt], hi = a.n_inline ? a.seg_inline[utt + 1] : a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * ROWS;
  if (row0 >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int nvalid = hi - row0;
#ifdef STTS_WN_TRACE
  // record [start, prologue end, phase-1 end, gate end, phase-2 end, end] in shader cycles + the 100 MHz wall clock at both ends
  // (every wave records its own row of 8: [64 * block + 8 * wave + i])
  long long* const dbg_rec = a.dbg ? a.dbg + 64 * (long)(blockIdx.x + gridDim.x * blockIdx.y) + 8 * w : nullptr;
  auto stamp = [&](int i) { if (dbg_rec && lane == 0) dbg_rec[i] = i == 6 || i == 7 ? wall_clock64() : clock64(); };
  stamp(6);
#else
  auto stamp = [](int) {};
#endif
  stamp(0);

  // ---- phase-1 weight stream: wave w reads its own contiguous slice [8 blocks][NC][2 halves][CT][64 lanes] x 16 B one ring
  // step ahead of the MFMAs that consume it, one load between every four MFMAs.  A ring step covers CPB components of one
  // 16-channel block: all 6 for F(2,5) (24 KB per wave), 4 of the 8 for F(4,5) (16 KB; its 128 accumulator registers leave
  // no room for two whole blocks).
  constexpr int CPB = M <= 2 ? NC : NC / 2, SPB = NC / CPB, NS1 = KB * SPB;  // components per step, steps per block, steps
  constexpr int T1 = CPB * 2 * CT;                                            // fragments (1 KB each) per wave and step
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.W1) + (size_t)w * NS1 * (T1 * 64) + lane;
  f32x4 bq0[T1], bq1[T1];
  auto load1 = [&](f32x4(&dst)[T1], int u) {
#pragma unroll
    for (int j = 0; j < T1; ++j) dst[j] = w1[(u * T1 + j) * 64];
  };

  load1(bq0, 0);  // needs only the kernel arguments: in flight while the utterance offsets and the rows arrive
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: input transform of the block's rows [row0 - 2, row0 + ROWS + 2): thread = (group, 2 channel quads)
  {
    const int g = tid & 15;
    f32x4 d[2][NC];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int row = row0 + M * g - 2 + q, cq = (tid >> 4) + 16 * k;
        const bool ok = row >= lo && row < hi;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.Hin + (long)min(max(row, lo), hi - 1) * C + 4 * cq);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        d[k][q] = ok ? v : z;
      }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int cq = (tid >> 4) + 16 * k;
      static_for<0, NC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        static_for<0, NC>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr float bt = (float)WnConst<M>::Bt[j][q];
          if constexpr (bt == 1.0f) v += d[k][q];
          else if constexpr (bt == -1.0f) v -= d[k][q];
          else if constexpr (bt != 0.0f) v += bt * d[k][q];
        });
        Af[(j * KB + (cq >> 2)) * 64 + (cq & 3) * 16 + g] = v;
      });
    }
  }
  // Orientation of every product: D^T = W x A^T, i.e. the MFMA's A operand is the weight fragment (rows = 16 output
  // channels) and its B operand the activation fragment (cols = the 16 groups / time rows).  A lane then holds FOUR
  // CONSECUTIVE CHANNELS (registers i = 0..3 -> channel 4 lq + i of the tile) of one group / row (l15), so bias, gate,
  // h / out / z traffic and the LDS hand-offs between the phases are all 16-byte accesses.
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
  stamp(1);

  // ---- phase 1: NC components x (tanh, sigmoid) x CT tiles, K = 128 channels in 8 blocks of 16 (4 MFMA k-steps each).
  // A ROLLED loop of two ring steps per iteration (two named register buffers).
  f32x4 acc[NC][2][CT];
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[j][h][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto step1 = [&](auto pc, int it, const f32x4(&cur)[T1]) {
    constexpr int p = decltype(pc)::value, j0 = SPB == 1 ? 0 : p * CPB;  // step u = 2 it + p: block t, components [j0, j0 + CPB)
    const int t = SPB == 1 ? 2 * it + p : it;
    f32x4 av[CPB];
#pragma unroll
    for (int j = 0; j < CPB; ++j) av[j] = Af[((j0 + j) * KB + t) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < CPB; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[j0 + j][h][c] = mfma4(cur[(j * 2 + h) * CT + c][s], av[j][s], acc[j0 + j][h][c]);
  };
  auto interleave = [&]() {  // the T1 loads of the NEXT step go between this step's 4 T1 MFMAs, one per four
#pragma unroll
    for (int q = 0; q < T1; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll 1
  for (int it = 0; it < NS1 / 2; ++it) {
    load1(bq1, 2 * it + 1);
    step1(std::integral_constant<int, 0>{}, it, bq0);
    interleave();
    if (2 * it + 2 < NS1) load1(bq0, 2 * it + 2);
    step1(std::integral_constant<int, 1>{}, it, bq1);
    interleave();
  }

  stamp(2);
  // ---- phase-2 operands requested now: all of this wave's res/skip weights, bias, the h / out values the epilogue updates.
  // (rows past the utterance are read too - the buffers carry kWnRowPad rows of slack - and never stored)
  const f32x4* w2 = reinterpret_cast<const f32x4*>(a.W2) + (size_t)w * KB * (NCT * 64) + lane;
  f32x4 cq2[KB][NCT];
#pragma unroll
  for (int t = 0; t < KB; ++t)
#pragma unroll
    for (int c = 0; c < NCT; ++c) cq2[t][c] = w2[(t * NCT + c) * 64];
  f32x4 bv[NCT], old[RT][NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int n = 16 * NCT * w + 16 * c + 4 * lq;  // first of this lane's four res/skip output columns of tile c
    bv[c] = *reinterpret_cast<const f32x4*>(a.b2 + n);
    const bool to_h = !LAST && n < C;
    const int col = (!LAST && n >= C) ? n - C : n;
    const float* src = to_h ? a.Hin : a.Out;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)(row0 + 16 * rt + l15) * C + col);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      old[rt][c] = (to_h || a.out_acc) ? v : z;
    }
  }

  // ---- output transform + gate: rows M * group + m of this lane's channels, written to LDS as the B operand of res/skip
#pragma unroll
  for (int c = 0; c < CT; ++c)
    static_for<0, M>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      f32x4 ya = {0.f, 0.f, 0.f, 0.f}, yb = {0.f, 0.f, 0.f, 0.f};
      static_for<0, NC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr float at = (float)WnConst<M>::At[m][j];
        if constexpr (at == 1.0f) {
          ya += acc[j][0][c];
          yb += acc[j][1][c];
        } else if constexpr (at == -1.0f) {
          ya -= acc[j][0][c];
          yb -= acc[j][1][c];
        } else if constexpr (at != 0.0f) {
          ya += at * acc[j][0][c];
          yb += at * acc[j][1][c];
        }
      });
      const f32x4 va = ya + ba[c] + ga[c], vb = yb + bb[c] + gb[c];
      f32x4 act;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // tanh(va) * sigmoid(vb) with hardware exp2 / rcp (1 ulp each), as wn_layer_kernel
        const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390082f * va[i]) + 1.0f);
        act[i] = th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * vb[i]));
      }
      const int row = M * l15 + m;
      A2[((row >> 4) * KB + CT * w + c) * 64 + lq * 16 + (row & 15)] = act;
    });
  __syncthreads();
  stamp(3);

  // ---- phase 2: res/skip, K = 128 from LDS; wave w owns columns [16 NCT w, 16 NCT (w + 1)), all row tiles
  f32x4 acc2[RT][NCT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc2[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    f32x4 av[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) av[rt] = A2[(rt * KB + t) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc2[rt][c] = mfma4(cq2[t][c][s], av[rt][s], acc2[rt][c]);
  }

  stamp(4);
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
    stamp(5);
    stamp(7);
    return;
  } else {
    // ---- tail: post projection + reverse coupling (+ the next coupling block's pre) on this block's rows.
    // wave w: mean tile and log-std tile of channels [16 w, 16 w + 16) of the coupled half, all row tiles
    const f32x4* w3 = reinterpret_cast<const f32x4*>(a.W3) + (size_t)w * KB * (2 * 64) + lane;
    f32x4 pq[KB][2];  // all of this wave's post weights (16 KB), requested at once
#pragma unroll
    for (int t = 0; t < KB; ++t) {
      pq[t][0] = w3[(t * 2 + 0) * 64];
      pq[t][1] = w3[(t * 2 + 1) * 64];
    }
    const int cc = 16 * w + 4 * lq;  // first of this lane's four channels of the coupled half
    const f32x4 pm = *reinterpret_cast<const f32x4*>(a.b3m + cc), ps = *reinterpret_cast<const f32x4*>(a.b3s + cc);
    f32x4 zold[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) zold[rt] = *reinterpret_cast<const f32x4*>(a.Z + (long)(row0 + 16 * rt + l15) * a.ldz + a.zcol0 + cc);
    __syncthreads();  // every wave has finished reading the gated activations
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)  // finished `out` values of (row 16 rt + l15, channels 16 (NCT w + c) + 4 lq + (0..3))
        A2[(rt * KB + NCT * w + c) * 64 + lq * 16 + l15] = old[rt][c] + (acc2[rt][c] + bv[c]);
    __syncthreads();
    f32x4 acc3[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc3[rt][0] = acc3[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KB; ++t) {
      f32x4 av[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av[rt] = A2[(rt * KB + t) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc3[rt][0] = mfma4(pq[t][0][s], av[rt][s], acc3[rt][0]);
          acc3[rt][1] = mfma4(pq[t][1][s], av[rt][s], acc3[rt][1]);
        }
    }
    // next block's pre: weights of wave w's two column tiles, requested ahead of the coupling math
    constexpr int KB4 = KB / 2;  // K = 64
    const f32x4* w4 = reinterpret_cast<const f32x4*>(a.W4) + (size_t)w * KB4 * (2 * 64) + lane;
    f32x4 rq[KB4][2];
    f32x4 hb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (a.tail > 1) {
#pragma unroll
      for (int t = 0; t < KB4; ++t) {
        rq[t][0] = w4[(t * 2 + 0) * 64];
        rq[t][1] = w4[(t * 2 + 1) * 64];
      }
      hb[0] = *reinterpret_cast<const f32x4*>(a.b4 + 32 * w + 4 * lq);
      hb[1] = *reinterpret_cast<const f32x4*>(a.b4 + 32 * w + 16 + 4 * lq);
    }
    // [row tile][4 blocks][lane]: the coupled half as the B operand of `pre` (the phase-1 operand in Af is long dead)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 mm = acc3[rt][0] + pm, ls = acc3[rt][1] + ps;
      f32x4 z1;
#pragma unroll
      for (int i = 0; i < 4; ++i) z1[i] = (zold[rt][i] - mm[i]) * __expf(-ls[i]);  // x1 = (x1 - m) * exp(-logs), flow.py:209
      if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(a.Z + (long)(row0 + 16 * rt + l15) * a.ldz + a.zcol0 + cc) = z1;
      Af[(rt * KB4 + w) * 64 + lq * 16 + l15] = z1;
    }
    if (a.tail < 2) {
      stamp(5);
      stamp(7);
      return;
    }
    __syncthreads();
    f32x4 acc4[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc4[rt][0] = acc4[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KB4; ++t) {
      f32x4 av[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av[rt] = Af[(rt * KB4 + t) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc4[rt][0] = mfma4(rq[t][0][s], av[rt][s], acc4[rt][0]);
          acc4[rt][1] = mfma4(rq[t][1][s], av[rt][s], acc4[rt][1]);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (16 * rt + l15 < nvalid) *reinterpret_cast<f32x4*>(a.Hpre + (long)(row0 + 16 * rt + l15) * C + 32 * w + 16 * c + 4 * lq) = acc4[rt][c] + hb[c];
    stamp(5);
    stamp(7);
  }
}

// ------------------------------------------------------------------------------------------------ host side
// F(m, 5) matrices from the points {0, 1, -1, 2, -1/2 (, -2, 1/2), inf} (Cook-Toom), in double; false if the self-check
// against direct correlation fails.  y_i = sum_j At[i][j] * (G g)_j * (Bt d)_j,  d = the m + 4 inputs of the group.
struct WnFusedMats {
  int m = 0, n = 0;
  double Bt[8][8], At[4][8], G[8][5];
};
inline bool wn_fused_matrices(int m, WnFusedMats* out) {
  const int r = 5, n = m + r - 1;
  if (m == 1) {  // the direct form: identity transforms
    *out = WnFusedMats();
    out->m = 1;
    out->n = 5;
    for (int j = 0; j < 8; ++j) {
      for (int qq = 0; qq < 8; ++qq) out->Bt[j][qq] = (j < 5 && j == qq) ? 1.0 : 0.0;
      for (int k = 0; k < 5; ++k) out->G[j][k] = (j < 5 && j == k) ? 1.0 : 0.0;
      for (int i = 0; i < 4; ++i) out->At[i][j] = (i == 0 && j < 5) ? 1.0 : 0.0;
    }
    return true;
  }
  if (m != 2 && m != 4) return false;
  const double all[7] = {0, 1, -1, 2, -0.5, -2, 0.5};
  std::vector<double> pts(all, all + (n - 1));
  auto polymul = [](const std::vector<double>& x, const std::vector<double>& y) {
    std::vector<double> z(x.size() + y.size() - 1, 0.0);
    for (size_t i = 0; i < x.size(); ++i)
      for (size_t j = 0; j < y.size(); ++j) z[i + j] += x[i] * y[j];
    return z;
  };
  std::vector<std::vector<double>> A(n, std::vector<double>(m, 0.0)), G(n, std::vector<double>(r, 0.0)), Cm(n, std::vector<double>(n, 0.0));
  for (int j = 0; j < n - 1; ++j) {
    for (int i = 0; i < m; ++i) A[j][i] = std::pow(pts[j], i);
    for (int k = 0; k < r; ++k) G[j][k] = std::pow(pts[j], k);
    std::vector<double> num{1.0};
    double den = 1.0;
    for (int l = 0; l < n - 1; ++l)
      if (l != j) {
        num = polymul(num, {-pts[l], 1.0});
        den *= pts[j] - pts[l];
      }
    for (size_t qq = 0; qq < num.size(); ++qq) Cm[qq][j] = num[qq] / den;
  }
  A[n - 1][m - 1] = 1.0;
  G[n - 1][r - 1] = 1.0;
  std::vector<double> Mp{1.0};
  for (int l = 0; l < n - 1; ++l) Mp = polymul(Mp, {-pts[l], 1.0});
  for (int qq = 0; qq < n; ++qq) Cm[qq][n - 1] = Mp[qq];
  out->m = m;
  out->n = n;
  for (int j = 0; j < 8; ++j) {
    for (int qq = 0; qq < 8; ++qq) out->Bt[j][qq] = (j < n && qq < n) ? Cm[qq][j] : 0.0;
    for (int k = 0; k < 5; ++k) out->G[j][k] = j < n ? G[j][k] : 0.0;
    for (int i = 0; i < 4; ++i) out->At[i][j] = (j < n && i < m) ? A[j][i] : 0.0;
  }
  std::vector<double> g(r), d(n);
  for (int k = 0; k < r; ++k) g[k] = std::sin(1.0 + k);
  for (int qq = 0; qq < n; ++qq) d[qq] = std::cos(0.3 + 1.7 * qq);
  for (int i = 0; i < m; ++i) {
    double ref = 0, y = 0;
    for (int k = 0; k < r; ++k) ref += g[k] * d[i + k];
    for (int j = 0; j < n; ++j) {
      double gg = 0, dd = 0;
      for (int k = 0; k < r; ++k) gg += out->G[j][k] * g[k];
      for (int qq = 0; qq < n; ++qq) dd += out->Bt[j][qq] * d[qq];
      y += out->At[i][j] * gg * dd;
    }
    if (std::fabs(y - ref) > 1e-9 * (1.0 + std::fabs(ref))) return false;
  }
  // the kernel's compile-time copies must be these matrices
  for (int j = 0; j < n; ++j) {
    for (int qq = 0; qq < n; ++qq) {
      const double kc = m == 2 ? WnConst<2>::Bt[j][qq] : WnConst<4>::Bt[j][qq];
      if (std::fabs(kc - out->Bt[j][qq]) > 1e-12) return false;
    }
    for (int i = 0; i < m; ++i) {
      const double kc = m == 2 ? WnConst<2>::At[i][j] : WnConst<4>::At[i][j];
      if (std::fabs(kc - out->At[i][j]) > 1e-12) return false;
    }
  }
  return true;
}

// MFMA-fragment order for v_mfma_f32_16x16x4_f32 B operands read straight from global memory:
//   out[((w * kb + t) * tiles + j) * 64 + lane][s] = row(w, j, lane & 15)[16 t + 4 (lane >> 4) + s]
// i.e. k-step s of 16-channel block t uses channel 16 t + 4 (lane >> 4) + s on lane group lane >> 4; the LDS images of the
// A operands use the same channel order.  row(w, j, c) returns the K-long source row of output column c of tile j of wave
// w (nullptr = zeros).
inline std::vector<float> pack_fragments(int waves, int kb, int tiles, const std::function<const double*(int, int, int)>& row) {
  std::vector<float> out((size_t)waves * kb * tiles * 64 * 4, 0.f);
  for (int w = 0; w < waves; ++w)
    for (int j = 0; j < tiles; ++j)
      for (int c = 0; c < 16; ++c) {
        const double* src = row(w, j, c);
        if (!src) continue;
        for (int t = 0; t < kb; ++t)
          for (int kq = 0; kq < 4; ++kq)
            for (int s = 0; s < 4; ++s)
              out[((((size_t)w * kb + t) * tiles + j) * 64 + kq * 16 + c) * 4 + s] = (float)src[16 * t + 4 * kq + s];
      }
  return out;
}

}  // namespace stts
