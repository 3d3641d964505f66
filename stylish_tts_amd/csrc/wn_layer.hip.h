// wn_layer_kernel: one WaveNet layer of the reverse flow in ONE launch (models/flow.py:70-87):
//   x_in = wn-conv1d k5 (h)                      128 -> 256
//   acts = tanh(x_in[:128] + g[:128]) * sigmoid(x_in[128:] + g[128:])        (fused_add_tanh_sigmoid_multiply, flow.py:7-14)
//   rs   = wn-Linear(acts)                       128 -> 256 (last layer: 128)
//   h   += rs[:128] ; out += rs[128:]            (last layer: out += rs)
// As two conv_gemm_f32 launches (gate epilogue, then split-accumulate epilogue) a B = 8 batch gives one wave per SIMD
// and two launches per layer for 3 GFLOP.  Here a block owns 32 time rows x all 256 gate columns, so the gated
// activations never leave the CU: phase 1 contracts K = 5 taps x 128 channels from HBM/L2 tiles, the gate epilogue
// writes acts [32 x 128] into LDS, phase 2 contracts K = 128 with the A operand read straight from LDS, and the
// epilogue updates h / out.  16 waves per block = 4 column groups x 4 K-groups (each K-group takes a quarter of every
// staged chunk; partial sums are exchanged through LDS in two halving steps), four waves per SIMD.
// Measured with in-kernel timestamps (fp32, B = 8, 35.5 us per launch): 24 K-iterations at 1.17 us = 28 us, i.e. ~80 %
// of the MFMA issue limit (2048 cycles per iteration per SIMD); prologue 2 us, exchange + gate 3 us, the rest 1.5 us.
// The operands of both epilogues (bias, gate, h / out values) are requested before the loops that precede them.
// h is double buffered (Hin -> Hout): neighbouring blocks still read this block's Hin rows as their conv halo.
//
// PREC (gemm.hip.h): fp32 operands stage 32 channels per iteration (128-byte tile rows, v_mfma_f32_32x32x2_f32 x 4 per
// K-group); bf16 / fp16 operands stage 64 channels per iteration in the same 128-byte rows (one v_mfma_f32_32x32x16 per
// K-group), activations are rounded when they enter LDS, weights are read pre-rounded.
#pragma once
#include <type_traits>

#include "gemm.hip.h"

namespace stts {

struct WnArgs {
  const float* Hin;    // [rows, 128]
  float* Hout;         // [rows, 128] (may be null on the last layer)
  float* Out;          // [rows, 128]
  const int* seg_off;
  const void* Win;     // packed paired [256][5][128]   (fp32, or 16-bit in the 16-bit operand modes; likewise below)
  const float* bin;    // [256] packed order
  const void* Wrs;     // packed plain [n_rs padded to 128][1][128]
  const float* brs;    // [n_rs]
  const float* gate;   // [n_utt][ld_gate]
  int ld_gate, gcol0;
  int n_rs;            // 256 (layers 0..n-2: h-part | out-part) or 128 (last layer: out only)
  int out_acc;         // accumulate into Out (0 on the first layer: output = zeros + ...)
  // tail (last layer of a coupling block only): 0 = none; 1 = the block's `post` projection + reverse coupling
  // (flow.py:199-211) on this tile's rows; 2 = additionally the NEXT coupling block's `pre` projection (flow.py:188-190),
  // whose input is exactly the half just updated.  Both are 1x1 (no halo), so the 32-row tile is self-contained.
  int tail;
  const void* Wproj;   // packed paired [128][1][128]
  const float* bproj;  // [128] packed order
  float* Z;            // [rows, ldz]; columns [zcol0, zcol0 + 64) are updated in place
  int ldz, zcol0;
  const void* Wpre;    // packed plain [128][1][64]
  const float* bpre;   // [128]
  float* Hpre;         // [rows, 128]: h of the next coupling block
};

template <int PREC>
__global__ void __launch_bounds__(1024) wn_layer_kernel(const WnArgs a) {
  constexpr bool B16 = PREC != PREC_F32;
  constexpr int RT = 32, C = 128, NG = 256, TAPS = 5, PAD = 2, KG = 4;
  constexpr int NT = 256 * KG, WPT = 2048 / NT, KEEP = 16 / KG;  // threads, 16-byte W loads per thread per tile, accumulator rows a wave finishes
  constexpr int ESZ = B16 ? 2 : 4;   // operand bytes
  constexpr int CHK = 128 / ESZ;     // channels per staged chunk: a tile row is always 128 bytes = 8 slots of 16
  constexpr int NCC = C / CHK;       // chunks per 128 channels (4 / 2)
  constexpr int XT = RT * CHK / 4;   // threads that fetch the activation tile (one fp32 x 4 each)
  constexpr int STG = (RT + NG) * 8;  // 16-byte slots per staging buffer
  __shared__ f32x4 stage[2 * STG];    // 73,728 B
  __shared__ f32x4 acts[NCC * RT * 8];  // [chunk][row][8 slots], same swizzle as the staging tiles (16 / 8 KB)

  const int utt = blockIdx.y;
  const int lo = a.seg_off[utt], hi = a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * RT;
  if (row0 >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kg = wid >> 2, wc = wid & 3;  // K-group, column group (64 packed columns)
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  };
  zero_acc();

  // Staging pipeline: an iteration is shorter than a loaded-memory round trip, so tiles are fetched TWO iterations ahead
  // into two register sets (set = tile parity) and moved to LDS one iteration ahead: iteration `it` computes tile it from
  // buffer it&1, stores tile it+1 (fetched during it-2) and fetches tile it+3.
  // two explicit register sets (named, not an array indexed by the set: that would be demoted to LDS/scratch)
  struct RegSet {
    f32x4 x, w[WPT];
    bool ok;
  };
  RegSet rs0, rs1;
  rs0.ok = rs1.ok = false;
  // weight tile: 256 (or 128) packed rows x one chunk; row n of a [rows][row_elems] operand, chunk at element offset e0
  auto wload = [&](RegSet& rs, const void* W, int nrows, int row_elems, int e0) {
    const char* base = reinterpret_cast<const char*>(W);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rs.w[i] = n < nrows ? *reinterpret_cast<const f32x4*>(base + ((long)n * row_elems + e0) * ESZ + sl * 16) : z;
    }
  };
  auto gload1 = [&](RegSet& rs, int t) {
    const int tap = t / NCC, chunk = t % NCC;
    if (tid < XT) {
      const int r = tid / (CHK / 4), sl = tid % (CHK / 4);
      const int grow = row0 + r + tap - PAD;
      const bool ok = grow >= lo && grow < hi;
      const int crow = min(max(grow, lo), hi - 1);
      rs.x = *reinterpret_cast<const f32x4*>(a.Hin + (long)crow * C + chunk * CHK + sl * 4);
      rs.ok = ok;
    }
    wload(rs, a.Win, NG, TAPS * C, tap * C + chunk * CHK);
  };
  auto lstore = [&](const RegSet& rs, int b, bool with_x) {
    f32x4* Xs = stage + b * STG;
    f32x4* Ws = Xs + RT * 8;
    if (with_x && tid < XT) {
      const int r = tid / (CHK / 4), sl = tid % (CHK / 4);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 v = rs.ok ? rs.x : z;
      if constexpr (B16) reinterpret_cast<u32x2*>(Xs)[(r * 8 + ((sl >> 1) ^ ((r >> 1) & 7))) * 2 + (sl & 1)] = pack4_16<PREC>(v);
      else Xs[r * 8 + (sl ^ ((r >> 1) & 7))] = v;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      Ws[n * 8 + (sl ^ ((n >> 1) & 7))] = rs.w[i];
    }
  };
  // this K-group's share of the staged chunk: slot pair 2*kg, 2*kg + 1 (lanes 0-31 / 32-63)
  auto mma = [&](const f32x4* As, const f32x4* Ws) {
    const int slot = 2 * kg + lh;
    const f32x4 xa = As[l31 * 8 + (slot ^ ((l31 >> 1) & 7))];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cidx = wc * 64 + j * 32 + l31;
      const f32x4 wb = Ws[cidx * 8 + (slot ^ ((cidx >> 1) & 7))];
      if constexpr (B16) {
        acc[j] = mfma16<PREC>(xa, wb, acc[j]);
      } else {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, wb.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, wb.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, wb.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, wb.w, acc[j], 0, 0, 0);
      }
    }
  };
  // one element of the [32 x 128] LDS activation tile (rounded to the operand type in the 16-bit modes)
  auto act_put = [&](int row, int ch, float v) {
    if constexpr (B16) {
      const int e = (((ch / CHK) * (RT * 8) + row * 8 + (((ch % CHK) >> 3) ^ ((row >> 1) & 7))) * 8 + (ch & 7));
      if constexpr (PREC == PREC_BF16) reinterpret_cast<__bf16*>(acts)[e] = (__bf16)v;
      else reinterpret_cast<_Float16*>(acts)[e] = (_Float16)v;
    } else {
      reinterpret_cast<float*>(acts)[((ch >> 5) * (RT * 8) + row * 8 + (((ch >> 2) & 7) ^ ((row >> 1) & 7))) * 4 + (ch & 3)] = v;
    }
  };
  // exchange between the four K-groups in two halving steps (partner kg ^ 1, then kg ^ 2): each group keeps the part
  // of the 16 accumulator rows it will finish and hands the rest over through LDS.
  // Accumulator registers must only ever be indexed by compile-time constants (a run-time index demotes the vector
  // to memory, and the compiler even re-merges two static branches into one dynamic loop).  So the odd partner first
  // swaps its register halves with conditional moves; afterwards BOTH finish the low registers and hand over the high
  // ones, and only the logical row number (an ordinary integer) depends on the group.
  float* red = reinterpret_cast<float*>(stage);  // [kg][wc][j][8][lane] : 4 * 4 * 2 * 8 * 64 floats = 64 KB
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
  };
  // tile row (0..31) of accumulator register r (< KEEP) after the exchange
  auto tile_row = [&](int r) {
    const int ro = r + 4 * (kg >> 1) + 8 * (kg & 1);
    return (ro & 3) + 8 * (ro >> 2) + 4 * lh;
  };

  // ------------------------------------------------------------------ phase 1: gate GEMM, K = 5 taps x 128 channels
  constexpr int N1 = TAPS * NCC;
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
    mma(Xs, Xs + RT * 8);
    if (it + 1 < N1) lstore(nset, (it + 1) & 1, true);
    if (it + 3 < N1) gload1(nset, it + 3);  // no dummy fetches at the tail: the next phase reuses the registers
    __syncthreads();
  };
  for (int it = 0; it < N1; it += 2) {
    iter1(it, rs1);
    iter1(it + 1, rs0);
  }

  // phase-2 operands are requested now, before the exchange and the gate math: the first two res/skip weight tiles
  // and the h / out values the final epilogue updates.
  const bool col_active = wc * 64 < a.n_rs;  // last layer: only 128 output columns
  constexpr int N2 = NCC;
  wload(rs0, a.Wrs, a.n_rs, C, 0);
  wload(rs1, a.Wrs, a.n_rs, C, CHK);
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
      const int row = tile_row(r);
      old[j][r] = (col_active && row < nvalid && (to_h || a.out_acc)) ? src[(long)(row0 + row) * C + col] : 0.0f;
    }
  }

  exchange();
#pragma unroll
  for (int r = 0; r < KEEP; ++r) {
    // tanh(va) * sigmoid(vb) with hardware exp2 / rcp (1 ulp each): 16 waves share four VALUs here, and IEEE
    // division sequences made this epilogue cost more than an MFMA iteration
    const float va = acc[0][r] + ba + ga, vb = acc[1][r] + bb + gb;
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390082f * va) + 1.0f);
    act_put(tile_row(r), ch, th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * vb)));
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase 2: res/skip GEMM, K = 128 from LDS acts
  zero_acc();
  lstore(rs0, 0, false);
  if (2 < N2) wload(rs0, a.Wrs, a.n_rs, C, 2 * CHK);
  __syncthreads();
  auto iter2 = [&](int it, RegSet& nset) {
    if (col_active) mma(acts + it * (RT * 8), stage + (it & 1) * STG + RT * 8);
    if (it + 1 < N2) lstore(nset, (it + 1) & 1, false);
    if (it + 3 < N2) wload(nset, a.Wrs, a.n_rs, C, (it + 3) * CHK);
    __syncthreads();
  };
  for (int it = 0; it < N2; it += 2) {
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
          const int row = tile_row(r);
          if (row < nvalid) dst[(long)(row0 + row) * C + col] = old[j][r] + (acc[j][r] + bv[j]);
        }
      }
    }
    return;
  }

  // ------------------------------------------------------------------ tail: post projection + coupling (+ next pre)
  // The finished `out` tile [32 x 128] goes to LDS (same chunked, swizzled layout as the gated activations) instead of
  // HBM and is the A operand of the coupling projection; the updated half of z then feeds the next block's `pre`.
  // contraction of the LDS tile in `acts` (NCH chunks) with W [128][row_elems]; all waves stage, column groups 0 and 1
  // multiply.  The caller has put a barrier between the last reader of `stage` and this call.
  auto lds_gemm = [&](auto nch, const void* W, int row_elems) {
    constexpr int NCH = decltype(nch)::value;
    zero_acc();
    wload(rs0, W, C, row_elems, 0);
    if constexpr (NCH > 1) wload(rs1, W, C, row_elems, CHK);
    lstore(rs0, 0, false);
    if constexpr (NCH > 2) wload(rs0, W, C, row_elems, 2 * CHK);
    __syncthreads();
    auto step = [&](int it, RegSet& nset) {
      if (wc < 2) mma(acts + it * (RT * 8), stage + (it & 1) * STG + RT * 8);
      if (it + 1 < NCH) lstore(nset, (it + 1) & 1, false);
      if (it + 3 < NCH) wload(nset, W, C, row_elems, (it + 3) * CHK);
      __syncthreads();
    };
#pragma unroll
    for (int it = 0; it < NCH; it += 2) {
      step(it, rs1);
      if (it + 1 < NCH) step(it + 1, rs0);
    }
  };
  if (col_active) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < KEEP; ++r) act_put(tile_row(r), wc * 64 + j * 32 + l31, old[j][r] + (acc[j][r] + bv[j]));
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
    const int row = tile_row(r);
    zold[r] = (wc < 2 && row < nvalid) ? a.Z[(long)(row0 + row) * a.ldz + a.zcol0 + cc] : 0.0f;
  }
  __syncthreads();  // `out` tile complete; every wave is past the exchange that used `stage`
  lds_gemm(std::integral_constant<int, NCC>{}, a.Wproj, C);
  exchange();
  float hb[2] = {0.f, 0.f};
  if (a.tail > 1 && wc < 2) {
    hb[0] = a.bpre[wc * 64 + l31];
    hb[1] = a.bpre[wc * 64 + 32 + l31];
  }
  if (wc < 2) {
#pragma unroll
    for (int r = 0; r < KEEP; ++r) {
      const int row = tile_row(r);
      const float z1 = (zold[r] - (acc[0][r] + pa)) * __expf(-(acc[1][r] + pb));  // x1 = (x1 - m) * exp(-logs)
      if (row < nvalid) a.Z[(long)(row0 + row) * a.ldz + a.zcol0 + cc] = z1;
      act_put(row, cc, z1);  // every wave finished reading `acts` before the last barrier of lds_gemm
    }
  }
  if (a.tail < 2) return;
  __syncthreads();
  lds_gemm(std::integral_constant<int, NCC / 2>{}, a.Wpre, C / 2);
  exchange();
  if (wc < 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < KEEP; ++r) {
        const int row = tile_row(r);
        if (row < nvalid) a.Hpre[(long)(row0 + row) * C + wc * 64 + j * 32 + l31] = acc[j][r] + hb[j];
      }
  }
}

}  // namespace stts
