// wn_layer_kernel: one WaveNet layer of the reverse flow in ONE launch (models/flow.py:70-87):
//   x_in = wn-conv1d k5 (h)                      128 -> 256
//   acts = tanh(x_in[:128] + g[:128]) * sigmoid(x_in[128:] + g[128:])        (fused_add_tanh_sigmoid_multiply, flow.py:7-14)
//   rs   = wn-Linear(acts)                       128 -> 256 (last layer: 128)
//   h   += rs[:128] ; out += rs[128:]            (last layer: out += rs)
// As two conv_gemm_f32 launches (gate epilogue, then split-accumulate epilogue) a B = 8 batch gives one wave per SIMD
// and two launches per layer for 3 GFLOP.  Here a block owns 32 time rows x all 256 gate columns, so the gated
// activations never leave the CU: phase 1 contracts K = 5 taps x 128 channels from HBM/L2 tiles, the gate epilogue
// writes acts [32 x 128] into LDS, phase 2 contracts K = 128 with the A operand read straight from LDS, and the
// epilogue updates h / out.  KG = 4: 16 waves per block = 4 column groups x 4 K-groups (each K-group takes a quarter of
// every 32-channel chunk; partial sums are exchanged through LDS in two halving steps), four waves per SIMD.
// Measured with in-kernel timestamps (B = 8, 35.5 us per launch): 24 K-iterations at 1.17 us = 28 us, i.e. ~80 % of the
// MFMA issue limit (2048 cycles per iteration per SIMD); prologue 2 us, exchange + gate 3 us, everything else 1.5 us.
// The operands of both epilogues (bias, gate, h / out values) are requested before the loops that precede them.
// h is double buffered (Hin -> Hout): neighbouring blocks still read this block's Hin rows as their conv halo.
#pragma once
#include <type_traits>

#include "gemm.hip.h"

namespace stts {

struct WnArgs {
  const float* Hin;    // [rows, 128]
  float* Hout;         // [rows, 128] (may be null on the last layer)
  float* Out;          // [rows, 128]
  const int* seg_off;
  const float* Win;    // packed paired [256][5][128]
  const float* bin;    // [256] packed order
  const float* Wrs;    // packed plain [n_rs padded to 128][1][128]
  const float* brs;    // [n_rs]
  const float* gate;   // [n_utt][ld_gate]
  int ld_gate, gcol0;
  int n_rs;            // 256 (layers 0..n-2: h-part | out-part) or 128 (last layer: out only)
  int out_acc;         // accumulate into Out (0 on the first layer: output = zeros + ...)
  // tail (last layer of a coupling block only): 0 = none; 1 = the block's `post` projection + reverse coupling
  // (flow.py:199-211) on this tile's rows; 2 = additionally the NEXT coupling block's `pre` projection (flow.py:188-190),
  // whose input is exactly the half just updated.  Both are 1x1 (no halo), so the 32-row tile is self-contained.
  int tail;
  const float* Wproj;  // packed paired [128][1][128]
  const float* bproj;  // [128] packed order
  float* Z;            // [rows, ldz]; columns [zcol0, zcol0 + 64) are updated in place
  int ldz, zcol0;
  const float* Wpre;   // packed plain [128][1][64]
  const float* bpre;   // [128]
  float* Hpre;         // [rows, 128]: h of the next coupling block
};

template <int KG>  // K-groups per block: 2 (8 waves) or 4 (16 waves, four per SIMD)
__global__ void __launch_bounds__(256 * KG) wn_layer_kernel(const WnArgs a) {
  constexpr int RT = 32, C = 128, NG = 256, TAPS = 5, PAD = 2;
  constexpr int NT = 256 * KG, WPT = 2048 / NT, KEEP = 16 / KG;  // threads, W f32x4 per thread per tile, accumulator rows a wave finishes
  constexpr int STG = (RT + NG) * 8;  // f32x4 per staging buffer
  __shared__ f32x4 stage[2 * STG];    // 73,728 B
  __shared__ f32x4 acts[RT * 32];     // 16 KB: [32-channel chunk][row][8 slots], same swizzle as the staging tiles

  const int utt = blockIdx.y;
  const int lo = a.seg_off[utt], hi = a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * RT;
  if (row0 >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kg = wid >> 2, wc = wid & 3;  // K-group, column group (64 packed columns)
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

  // ------------------------------------------------------------------ phase 1: gate GEMM, K = 5 taps x 4 chunks of 32
  // Staging pipeline: an iteration is only 16 MFMAs per wave (~1 us per SIMD), shorter than a loaded-memory round trip,
  // so tiles are fetched TWO iterations ahead into two register sets (set = tile parity) and moved to LDS one iteration
  // ahead: iteration `it` computes tile it from buffer it&1, stores tile it+1 (fetched during it-2) and fetches tile it+3.
  // two explicit register sets (named, not an array indexed by the set: that would be demoted to LDS/scratch)
  struct RegSet {
    f32x4 x, w[WPT];
    bool ok;
  };
  RegSet rs0, rs1;
  rs0.ok = rs1.ok = false;
  auto gload1 = [&](RegSet& rs, int t) {
    t = min(t, TAPS * 4 - 1);
    const int tap = t >> 2, chunk = t & 3;
    if (tid < 256) {
      const int r = tid >> 3, sl = tid & 7;
      const int grow = row0 + r + tap - PAD;
      const bool ok = grow >= lo && grow < hi;
      const int crow = min(max(grow, lo), hi - 1);
      rs.x = *reinterpret_cast<const f32x4*>(a.Hin + (long)crow * C + chunk * 32 + sl * 4);
      rs.ok = ok;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      rs.w[i] = *reinterpret_cast<const f32x4*>(a.Win + ((long)n * TAPS + tap) * C + chunk * 32 + sl * 4);
    }
  };
  auto lstore = [&](const RegSet& rs, int b, bool with_x) {
    f32x4* Xs = stage + b * STG;
    f32x4* Ws = Xs + RT * 8;
    if (with_x && tid < 256) {
      const int r = tid >> 3, sl = tid & 7;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      Xs[r * 8 + (sl ^ ((r >> 1) & 7))] = rs.ok ? rs.x : z;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      Ws[n * 8 + (sl ^ ((n >> 1) & 7))] = rs.w[i];
    }
  };
  auto mma = [&](const f32x4 xa, const f32x4* Ws, int kk) {
    const int slot = 2 * kk + lh;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cidx = wc * 64 + j * 32 + l31;
      const f32x4 wb = Ws[cidx * 8 + (slot ^ ((cidx >> 1) & 7))];
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, wb.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, wb.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, wb.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, wb.w, acc[j], 0, 0, 0);
    }
  };
  // exchange between the two K-groups: each group keeps the half of the 16 accumulator rows it will finish
  // (kg 0: r < 8, kg 1: r >= 8) and hands the other half over through LDS.
  float* red = reinterpret_cast<float*>(stage);  // [kg][wc][j][8][lane] : KG * 4 * 2 * 8 * 64 floats = 32 / 64 KB
  // Accumulator registers must only ever be indexed by compile-time constants (a run-time index demotes the vector
  // to memory, and the compiler even re-merges two static branches into one dynamic loop).  So K-group 1 first swaps
  // its register halves with conditional moves; afterwards BOTH groups finish registers 0..7 and hand over 8..15,
  // and only the logical row number (an ordinary integer) depends on the group: ro = r + 8*kg resp. r - 8*kg.
  auto exchange = [&]() {
    const int k0 = kg & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float lo8 = acc[j][r], hi8 = acc[j][r + 8];
        acc[j][r] = k0 ? hi8 : lo8;
        acc[j][r + 8] = k0 ? lo8 : hi8;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 8; r < 16; ++r) red[(((kg * 4 + wc) * 2 + j) * 8 + (r - 8)) * 64 + lane] = acc[j][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[j][r] += red[((((kg ^ 1) * 4 + wc) * 2 + j) * 8 + r) * 64 + lane];
    if constexpr (KG == 4) {  // second level: partner kg ^ 2, quarters of the 16 rows
      const int k1 = kg >> 1;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lo4 = acc[j][r], hi4 = acc[j][r + 4];
          acc[j][r] = k1 ? hi4 : lo4;
          acc[j][r + 4] = k1 ? lo4 : hi4;
        }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 4; r < 8; ++r) red[(((kg * 4 + wc) * 2 + j) * 8 + (r - 4)) * 64 + lane] = acc[j][r];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] += red[((((kg ^ 2) * 4 + wc) * 2 + j) * 8 + r) * 64 + lane];
    }
  };
  // logical accumulator row of register r (< KEEP) after the exchange
  auto logical_row = [&](int r) { return KG == 4 ? r + 4 * (kg >> 1) + 8 * (kg & 1) : r + 8 * kg; };

  gload1(rs0, 0);
  lstore(rs0, 0, true);
  gload1(rs1, 1);
  gload1(rs0, 2);
  // operands of the gate epilogue, fetched behind the first tiles so their latency is hidden by the loop
  const int ch = wc * 32 + l31;  // activation channel of this lane
  const float ba = a.bin[wc * 64 + l31], bb = a.bin[wc * 64 + 32 + l31];
  const float ga = a.gate[(long)utt * a.ld_gate + a.gcol0 + ch], gb = a.gate[(long)utt * a.ld_gate + a.gcol0 + C + ch];
  __syncthreads();
  auto iter1 = [&](int it, RegSet& nset) {  // nset: register set holding tile it+1
    const f32x4* Xs = stage + (it & 1) * STG;
    const f32x4* Ws = Xs + RT * 8;
    {
      const int kk = (4 / KG) * kg, slot = 2 * kk + lh;
      mma(Xs[l31 * 8 + (slot ^ ((l31 >> 1) & 7))], Ws, kk);
    }
    lstore(nset, (it + 1) & 1, true);
    if (it + 3 < TAPS * 4) gload1(nset, it + 3);  // no dummy fetches at the tail: the next phase reuses the registers
    if constexpr (KG == 2) {
      const int kk = 2 * kg + 1, slot = 2 * kk + lh;
      mma(Xs[l31 * 8 + (slot ^ ((l31 >> 1) & 7))], Ws, kk);
    }
    __syncthreads();
  };
  for (int it = 0; it < TAPS * 4; it += 2) {
    iter1(it, rs1);
    iter1(it + 1, rs0);
  }

  // phase-2 operands are requested now, before the exchange and the gate math: the first two res/skip weight tiles
  // and the h / out values the final epilogue updates.
  const bool col_active = wc * 64 < a.n_rs;  // last layer: only 128 output columns
  auto gload2 = [&](RegSet& rs, int chunk) {
    chunk = min(chunk, 3);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rs.w[i] = n < a.n_rs ? *reinterpret_cast<const f32x4*>(a.Wrs + (long)n * C + chunk * 32 + sl * 4) : z;
    }
  };
  gload2(rs0, 0);
  gload2(rs1, 1);
  const int nvalid = hi - row0;
  const bool wide = a.n_rs == 2 * C;  // 256-wide res/skip: first half goes to h, the rest to out
  float bv[2], old[2][KEEP];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = wc * 64 + j * 32 + l31;
    bv[j] = col_active ? a.brs[n] : 0.0f;
    const bool to_h = wide && n < C;
    const int col = (wide && n >= C) ? n - C : n;
    const float* src = to_h ? a.Hin : a.Out;
#pragma unroll
    for (int r = 0; r < KEEP; ++r) {
      const int ro = logical_row(r), row = (ro & 3) + 8 * (ro >> 2) + 4 * lh;
      old[j][r] = (col_active && row < nvalid && (to_h || a.out_acc)) ? src[(long)(row0 + row) * C + col] : 0.0f;
    }
  }

  exchange();
  {
    float* af = reinterpret_cast<float*>(acts);
    const int chunk = ch >> 5, slot = (ch >> 2) & 7;
    auto gate_one = [&](float xa, float xb, int r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float va = xa + ba + ga, vb = xb + bb + gb;
      // tanh(va) * sigmoid(vb) with hardware exp2 / rcp (1 ulp each): 16 waves share four VALUs here, and IEEE
      // division sequences made this epilogue cost more than an MFMA iteration
      const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390082f * va) + 1.0f);
      af[(chunk * (RT * 8) + row * 8 + (slot ^ ((row >> 1) & 7))) * 4 + (ch & 3)] = th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * vb));
    };
#pragma unroll
    for (int r = 0; r < KEEP; ++r) gate_one(acc[0][r], acc[1][r], logical_row(r));
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase 2: res/skip GEMM, K = 128 from LDS acts
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  lstore(rs0, 0, false);
  gload2(rs0, 2);
  __syncthreads();
  auto iter2 = [&](int it, RegSet& nset) {
    const f32x4* Ws = stage + (it & 1) * STG + RT * 8;
    if (col_active) {
      const int kk = (4 / KG) * kg, slot = 2 * kk + lh;
      mma(acts[it * (RT * 8) + l31 * 8 + (slot ^ ((l31 >> 1) & 7))], Ws, kk);
    }
    lstore(nset, (it + 1) & 1, false);
    if (it + 3 < 4) gload2(nset, it + 3);
    if constexpr (KG == 2) {
      if (col_active) {
        const int kk = 2 * kg + 1, slot = 2 * kk + lh;
        mma(acts[it * (RT * 8) + l31 * 8 + (slot ^ ((l31 >> 1) & 7))], Ws, kk);
      }
    }
    __syncthreads();
  };
  for (int it = 0; it < 4; it += 2) {
    iter2(it, rs1);
    iter2(it + 1, rs0);
  }
  exchange();
  if (!a.tail) {
    if (col_active) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = wc * 64 + j * 32 + l31;
        const bool to_h = wide && n < C;
        const int col = (wide && n >= C) ? n - C : n;
        float* dst = to_h ? a.Hout : a.Out;
#pragma unroll
        for (int r = 0; r < KEEP; ++r) {
          const int ro = logical_row(r), row = (ro & 3) + 8 * (ro >> 2) + 4 * lh;
          if (row < nvalid) dst[(long)(row0 + row) * C + col] = old[j][r] + (acc[j][r] + bv[j]);
        }
      }
    }
    return;
  }

  // ------------------------------------------------------------------ tail: post projection + coupling (+ next pre)
  // The finished `out` tile [32 x 128] goes to LDS (same chunked, swizzled layout as the gated activations) instead of
  // HBM and is the A operand of the coupling projection; the updated half of z then feeds the next block's `pre`.
  float* af = reinterpret_cast<float*>(acts);
  auto act_put = [&](int row, int ch, float v) {
    af[((ch >> 5) * (RT * 8) + row * 8 + (((ch >> 2) & 7) ^ ((row >> 1) & 7))) * 4 + (ch & 3)] = v;
  };
  auto gloadw = [&](RegSet& rs, const float* W, int kc, int chunk) {  // 128 packed columns x 32 channels
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rs.w[i] = n < C ? *reinterpret_cast<const f32x4*>(W + (long)n * kc + chunk * 32 + sl * 4) : z;
    }
  };
  // contraction of the LDS tile in `acts` (NCH chunks of 32 channels) with W [128][kc]; all waves stage, column
  // groups 0 and 1 multiply.  The caller has put a barrier between the last reader of `stage` and this call.
  auto lds_gemm = [&](auto nch, const float* W, int kc) {
    constexpr int NCH = decltype(nch)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    gloadw(rs0, W, kc, 0);
    gloadw(rs1, W, kc, 1);
    lstore(rs0, 0, false);
    if constexpr (NCH > 2) gloadw(rs0, W, kc, 2);
    __syncthreads();
    auto step = [&](int it, RegSet& nset) {
      const f32x4* Ws = stage + (it & 1) * STG + RT * 8;
      if (wc < 2) {
        const int kk = (4 / KG) * kg, slot = 2 * kk + lh;
        mma(acts[it * (RT * 8) + l31 * 8 + (slot ^ ((l31 >> 1) & 7))], Ws, kk);
        if constexpr (KG == 2) {
          const int kk2 = 2 * kg + 1, slot2 = 2 * kk2 + lh;
          mma(acts[it * (RT * 8) + l31 * 8 + (slot2 ^ ((l31 >> 1) & 7))], Ws, kk2);
        }
      }
      if (it + 1 < NCH) lstore(nset, (it + 1) & 1, false);
      if (it + 3 < NCH) gloadw(nset, W, kc, it + 3);
      __syncthreads();
    };
#pragma unroll
    for (int it = 0; it < NCH; it += 2) {
      step(it, rs1);
      step(it + 1, rs0);
    }
  };
  if (col_active) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < KEEP; ++r) {
        const int ro = logical_row(r), row = (ro & 3) + 8 * (ro >> 2) + 4 * lh;
        act_put(row, wc * 64 + j * 32 + l31, old[j][r] + (acc[j][r] + bv[j]));
      }
  }
  // coupling operands, requested ahead of the projection
  const int cc = (wc & 1) * 32 + l31;  // result channel of this lane (column groups 0, 1)
  float pa = 0.f, pb = 0.f, zold[KEEP];
  if (wc < 2) {
    pa = a.bproj[wc * 64 + l31];
    pb = a.bproj[wc * 64 + 32 + l31];
  }
#pragma unroll
  for (int r = 0; r < KEEP; ++r) {
    const int ro = logical_row(r), row = (ro & 3) + 8 * (ro >> 2) + 4 * lh;
    zold[r] = (wc < 2 && row < nvalid) ? a.Z[(long)(row0 + row) * a.ldz + a.zcol0 + cc] : 0.0f;
  }
  __syncthreads();  // `out` tile complete; every wave is past the exchange that used `stage`
  lds_gemm(std::integral_constant<int, 4>{}, a.Wproj, C);
  exchange();
  float hb[2] = {0.f, 0.f};
  if (a.tail > 1 && wc < 2) {
    hb[0] = a.bpre[wc * 64 + l31];
    hb[1] = a.bpre[wc * 64 + 32 + l31];
  }
  if (wc < 2) {
#pragma unroll
    for (int r = 0; r < KEEP; ++r) {
      const int ro = logical_row(r), row = (ro & 3) + 8 * (ro >> 2) + 4 * lh;
      const float z1 = (zold[r] - (acc[0][r] + pa)) * __expf(-(acc[1][r] + pb));  // x1 = (x1 - m) * exp(-logs)
      if (row < nvalid) a.Z[(long)(row0 + row) * a.ldz + a.zcol0 + cc] = z1;
      act_put(row, cc, z1);  // every wave finished reading `acts` before the last barrier of lds_gemm
    }
  }
  if (a.tail < 2) return;
  __syncthreads();
  lds_gemm(std::integral_constant<int, 2>{}, a.Wpre, C / 2);
  exchange();
  if (wc < 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < KEEP; ++r) {
        const int ro = logical_row(r), row = (ro & 3) + 8 * (ro >> 2) + 4 * lh;
        if (row < nvalid) a.Hpre[(long)(row0 + row) * C + wc * 64 + j * 32 + l31] = acc[j][r] + hb[j];
      }
  }
}

}  // namespace stts
