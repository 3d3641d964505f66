// wn_layer_rows16_kernel: the fused WaveNet layer of wn_layer.hip.h for SMALL batches (fp32 operands only).
// A 32-row block gives one utterance of 3 s only 30 workgroups, each a 24-iteration MFMA chain of ~35 us: the launch is
// latency-bound on 30 of 256 CUs.  Here a block owns 16 time rows (v_mfma_f32_16x16x4_f32 tiles), so there are twice the
// workgroups and every iteration carries half the matrix work (~1 024 instead of 2 048 cycles per SIMD).  Per block the
// weight staging is unchanged, i.e. twice the L2 -> LDS bytes per frame: that is why large batches (already one block
// per CU, ~5.6 TB/s of tile traffic) keep the 32-row kernel.
//   16 waves = 8 channel groups (16 gate pairs: 16 'a' + 16 'b' packed columns) x 2 K-groups (16 channels of each
//   32-channel chunk); K-group 1 hands its accumulators to K-group 0 through LDS, which runs the epilogues.
// Operand fragments: lane l supplies row / column l % 16 and the four channels 4 * (l / 16) .. + 3 of its K-group's 16
// as ONE 16-byte LDS read; MFMA j of the four uses element j, so the k-order inside a 16-channel group is permuted
// identically for both operands (a dot product does not care).  D: lane l holds column l % 16, rows 4 * (l / 16) + i.
#pragma once
#include "wn_layer.hip.h"

namespace stts {

__global__ void __launch_bounds__(1024) wn_layer_rows16_kernel(const WnArgs a) {
  constexpr int RT = 16, C = 128, NG = 256, TAPS = 5, PAD = 2, NT = 1024, WPT = 2048 / NT;
  constexpr int STG = (RT + NG) * 8;    // 16-byte slots per staging buffer (a tile row = 32 channels = 128 bytes)
  __shared__ f32x4 stage[2 * STG];      // 69,632 B
  __shared__ f32x4 acts[4 * RT * 8];    // [32-channel chunk][row][8 slots], 8 KB

  const int utt = blockIdx.y;
  const int lo = a.seg_off[utt], hi = a.seg_off[utt + 1];
  const int row0 = lo + blockIdx.x * RT;
  if (row0 >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kg = wid >> 3, cg = wid & 7;   // K-group, channel group
  const int l15 = lane & 15, lg = lane >> 4;
  const int nvalid = hi - row0;

  f32x4 acc[2];
  auto zero_acc = [&]() {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  struct RegSet {
    f32x4 x, w[WPT];
    bool ok;
  };
  RegSet rs0, rs1;
  rs0.ok = rs1.ok = false;
  auto wload = [&](RegSet& rs, const void* W, int nrows, int row_elems, int e0) {
    const float* base = reinterpret_cast<const float*>(W);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rs.w[i] = n < nrows ? *reinterpret_cast<const f32x4*>(base + (long)n * row_elems + e0 + sl * 4) : z;
    }
  };
  auto gload1 = [&](RegSet& rs, int t) {
    const int tap = t >> 2, chunk = t & 3;
    if (tid < RT * 8) {
      const int r = tid >> 3, sl = tid & 7;
      const int grow = row0 + r + tap - PAD;
      rs.ok = grow >= lo && grow < hi;
      rs.x = *reinterpret_cast<const f32x4*>(a.Hin + (long)min(max(grow, lo), hi - 1) * C + chunk * 32 + sl * 4);
    }
    wload(rs, a.Win, NG, TAPS * C, tap * C + chunk * 32);
  };
  auto lstore = [&](const RegSet& rs, int b, bool with_x) {
    f32x4* Xs = stage + b * STG;
    f32x4* Ws = Xs + RT * 8;
    if (with_x && tid < RT * 8) {
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
  // tile j of this wave sits at packed column cbase + j * cstep + l15
  auto mma = [&](const f32x4* As, const f32x4* Ws, int cbase, int cstep) {
    const int slot = kg * 4 + lg;
    const f32x4 xa = As[l15 * 8 + (slot ^ ((l15 >> 1) & 7))];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cbase + j * cstep + l15;
      const f32x4 wb = Ws[c * 8 + (slot ^ ((c >> 1) & 7))];
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb.w, acc[j], 0, 0, 0);
    }
  };
  auto act_put = [&](int row, int ch, float v) {
    reinterpret_cast<float*>(acts)[((ch >> 5) * (RT * 8) + row * 8 + (((ch >> 2) & 7) ^ ((row >> 1) & 7))) * 4 + (ch & 3)] = v;
  };
  // K-group 1 parks its partial sums in LDS (the staging buffers are idle), K-group 0 adds them and carries on alone
  float* red = reinterpret_cast<float*>(stage);  // [cg][j][i][lane]
  auto exchange = [&]() {
    if (kg == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((cg * 2 + j) * 4 + i) * 64 + lane] = acc[j][i];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] += red[((cg * 2 + j) * 4 + i) * 64 + lane];
    }
  };
  const int pair_base = (cg >> 1) * 64 + (cg & 1) * 16;  // packed 'a' column of result channel cg*16 (the 'b' column is +32)
  const int plain_base = cg * 32;                        // plain packing: this wave's 32 output columns

  // ------------------------------------------------------------------ phase 1: gate GEMM
  constexpr int N1 = TAPS * 4;
  gload1(rs0, 0);
  lstore(rs0, 0, true);
  gload1(rs1, 1);
  gload1(rs0, 2);
  const int ch = cg * 16 + l15;
  const float ba = a.bin[pair_base + l15], bb = a.bin[pair_base + 32 + l15];
  const float ga = a.gate[(long)utt * a.ld_gate + a.gcol0 + ch], gb = a.gate[(long)utt * a.ld_gate + a.gcol0 + C + ch];
  __syncthreads();
  auto iter1 = [&](int it, RegSet& nset) {
    const f32x4* Xs = stage + (it & 1) * STG;
    mma(Xs, Xs + RT * 8, pair_base, 32);
    if (it + 1 < N1) lstore(nset, (it + 1) & 1, true);
    if (it + 3 < N1) gload1(nset, it + 3);
    __syncthreads();
  };
  for (int it = 0; it < N1; it += 2) {
    iter1(it, rs1);
    iter1(it + 1, rs0);
  }
  const bool col_active = plain_base < a.n_rs;
  wload(rs0, a.Wrs, a.n_rs, C, 0);
  wload(rs1, a.Wrs, a.n_rs, C, 32);
  const bool wide = a.n_rs == 2 * C;
  float bv[2], old[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = plain_base + j * 16 + l15;
    const bool mine = col_active && kg == 0;
    bv[j] = mine ? a.brs[n] : 0.0f;
    const bool to_h = wide && n < C;
    const int col = (wide && n >= C) ? n - C : n;
    const float* src = to_h ? a.Hin : a.Out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lg + i;
      old[j][i] = (mine && row < nvalid && (to_h || a.out_acc)) ? src[(long)(row0 + row) * C + col] : 0.0f;
    }
  }
  exchange();
  if (kg == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float va = acc[0][i] + ba + ga, vb = acc[1][i] + bb + gb;
      const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.885390082f * va) + 1.0f);
      act_put(4 * lg + i, ch, th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695041f * vb)));
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase 2: res/skip GEMM from the LDS tile
  zero_acc();
  lstore(rs0, 0, false);
  wload(rs0, a.Wrs, a.n_rs, C, 64);
  __syncthreads();
  auto iter2 = [&](int it, RegSet& nset) {
    if (col_active) mma(acts + it * (RT * 8), stage + (it & 1) * STG + RT * 8, plain_base, 16);
    if (it + 1 < 4) lstore(nset, (it + 1) & 1, false);
    if (it + 3 < 4) wload(nset, a.Wrs, a.n_rs, C, (it + 3) * 32);
    __syncthreads();
  };
  iter2(0, rs1);
  iter2(1, rs0);
  iter2(2, rs1);
  iter2(3, rs0);
  exchange();
  if (!a.tail) {
    if (col_active && kg == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = plain_base + j * 16 + l15;
        const bool to_h = wide && n < C;
        const int col = (wide && n >= C) ? n - C : n;
        float* dst = to_h ? a.Hout : a.Out;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 4 * lg + i;
          if (row < nvalid) dst[(long)(row0 + row) * C + col] = old[j][i] + (acc[j][i] + bv[j]);
        }
      }
    }
    return;
  }

  // ------------------------------------------------------------------ tail: post projection + coupling (+ next pre)
  auto lds_gemm = [&](auto nch, const void* W, int row_elems, int cbase, int cstep) {
    constexpr int NCH = decltype(nch)::value;
    zero_acc();
    wload(rs0, W, C, row_elems, 0);
    wload(rs1, W, C, row_elems, 32);
    lstore(rs0, 0, false);
    if constexpr (NCH > 2) wload(rs0, W, C, row_elems, 64);
    __syncthreads();
    auto step = [&](int it, RegSet& nset) {
      if (cg < 4) mma(acts + it * (RT * 8), stage + (it & 1) * STG + RT * 8, cbase, cstep);
      if (it + 1 < NCH) lstore(nset, (it + 1) & 1, false);
      if (it + 3 < NCH) wload(nset, W, C, row_elems, (it + 3) * 32);
      __syncthreads();
    };
#pragma unroll
    for (int it = 0; it < NCH; it += 2) {
      step(it, rs1);
      step(it + 1, rs0);
    }
  };
  if (col_active && kg == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) act_put(4 * lg + i, plain_base + j * 16 + l15, old[j][i] + (acc[j][i] + bv[j]));
  }
  const bool coupler = cg < 4 && kg == 0;  // result channel ch (< 64) of the coupled half
  float pa = 0.f, pb = 0.f, zold[4];
  if (coupler) {
    pa = a.bproj[pair_base + l15];
    pb = a.bproj[pair_base + 32 + l15];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) zold[i] = (coupler && 4 * lg + i < nvalid) ? a.Z[(long)(row0 + 4 * lg + i) * a.ldz + a.zcol0 + ch] : 0.0f;
  __syncthreads();
  lds_gemm(std::integral_constant<int, 4>{}, a.Wproj, C, pair_base, 32);
  exchange();
  float hb[2] = {0.f, 0.f};
  if (a.tail > 1 && coupler) {
    hb[0] = a.bpre[plain_base + l15];
    hb[1] = a.bpre[plain_base + 16 + l15];
  }
  if (coupler) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lg + i;
      const float z1 = (zold[i] - (acc[0][i] + pa)) * __expf(-(acc[1][i] + pb));
      if (row < nvalid) a.Z[(long)(row0 + row) * a.ldz + a.zcol0 + ch] = z1;
      act_put(row, ch, z1);
    }
  }
  if (a.tail < 2) return;
  __syncthreads();
  lds_gemm(std::integral_constant<int, 2>{}, a.Wpre, C / 2, plain_base, 16);
  exchange();
  if (coupler) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * lg + i;
        if (row < nvalid) a.Hpre[(long)(row0 + row) * C + plain_base + j * 16 + l15] = acc[j][i] + hb[j];
      }
  }
}

}  // namespace stts
