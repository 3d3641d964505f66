// conv_gemm16_kernel: the dense contraction of the hot path for the 16-bit operand modes at large batches
// (BASELINE cfg3 bf16 / cfg5 fp16): Conv1d k = 1..7 and Linear on v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate.
//
// Same contract as conv_gemm_f32<..., EPI_STORE, X16> (gemm.hip.h): activations are 16-bit TIME-MAJOR rows written by their
// producers, weights are packed [cout][tap][cin] 16-bit, up to 3 K segments, utterance boundaries are the conv's zero padding,
// epilogue = bias / activation / residual / alpha / fp32 and-or 16-bit store / GRN sums of squares.  What differs is the loop:
//
//   * block = 256 time rows x 256 output channels, K tile = 64 channels of one (segment, tap): 64 KB per stage, two stages;
//   * every operand byte goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write), in half tiles of
//     128 rows x 128 bytes; the 16-byte-slot swizzle slot ^= (row >> 1) & 7 is applied to the SOURCE slot a lane fetches (the LDS
//     destination of an LDS-DMA is lane-linear), and to the fragment reads (conflict-free ds_read_b128);
//   * 8 waves = 2 (time) x 4 (cout), wave tile 128 x 64 as 8 x 4 accumulators of 16 x 16; a K tile is FOUR phases of one
//     64 x 32 quadrant each (16 MFMAs), and each phase is  { fragment reads + one half tile of LDS-DMA ; barrier ; MFMAs ; barrier };
//   * the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so on every SIMD one wave's MFMA
//     segment runs beside the other wave's read / DMA segment: the matrix pipe always has a wave to issue from;
//   * waits are counted: the DMA of tile t+1's activations and tile t+2's weights stays in flight across the barriers, one
//     `s_waitcnt vmcnt(4)` per K tile retires what the NEXT tile reads (never vmcnt(0) in the loop);
//   * products are oriented D^T = W x X^T (weights as the MFMA's row operand): a lane then holds 4 CONSECUTIVE output channels of
//     one time row, so every epilogue access is 16 bytes (8 for 16-bit rows).
//
// Staging schedule (tile t in stage t & 1; X0 / X1 = activation rows 0-127 / 128-255, W0 / W1 = output channels likewise):
//   phase 0: reads X-quadrant 0 + W-quadrant 0 | DMA X0(t+1) -> other stage     phase 2: reads X-quadrant 1 | DMA W0(t+2) -> this stage
//   phase 1: reads W-quadrant 1                | DMA X1(t+1) -> other stage     phase 3: (no reads)         | DMA W1(t+2) -> this stage, vmcnt(4)
// Hazards: a region is re-filled only after every wave's reads of it have retired and a barrier has passed (the W reads of phase 1
// are waited for BEFORE that phase's barrier, because phase 2 re-fills W one barrier later); data is read one phase after the
// counted wait (+ barrier) that retires its DMA (guide: "Read a staged buffer one phase AFTER the wait that retires it").
#pragma once
#include "gemm.hip.h"

namespace stts {

constexpr int kG16Tile = 256;  // rows and output channels per block
constexpr int kG16K = 64;      // channels per K tile

template <int PREC>
__device__ __forceinline__ f32x4 g16_mfma(const f32x4 a, const f32x4 b, const f32x4 c) {
  if constexpr (PREC == PREC_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// workgroup barrier that neither the compiler nor the machine scheduler moves LDS reads, LDS-DMA or MFMAs across
#define STTS_G16_BARRIER()                    \
  do {                                        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)
#define STTS_G16_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)  // lgkmcnt(0) alone (gfx9 encoding: vmcnt 63, expcnt 7)
#define STTS_G16_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n)&15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))  // vmcnt(n) alone

// ABL (tools/probes/gemm16_probe.hip only; results invalid, timing only): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop,
// 4 no MFMAs, 8 no epilogue stores, 16 no barriers in the loop, 32 no s_setprio around the MFMA clusters, 64 nontemporal epilogue stores
template <int PREC, bool MSEG, int ABL = 0>
__global__ void __launch_bounds__(512, 2) conv_gemm16_kernel(const GemmArgs a) {
  static_assert(PREC == PREC_BF16 || PREC == PREC_F16, "16-bit operand modes only");
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-descriptor type and builtins exist on the device side only: the host pass sees an empty body)
  // ALL of the block's LDS is this one array (a second __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt before
  // every fragment read): 2 stages x [X0 | X1 | W0 | W1] x 128 rows x 8 slots of 16 bytes = 128 KB
  __shared__ f32x4 lds[2 * 4 * 1024 + 64];  // (+ 1 KB behind the stages: the block's tile directory, below)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wv >> 2, wc = wv & 3;  // time half / cout quarter of this wave; waves w and w + 4 share a SIMD

  // one LDS-DMA wave-instruction = 64 lanes x 16 B = 8 rows of 128 B; a half tile = 16 of them = 2 per wave.  Per-lane constants:
  // row of the half tile and the (swizzled) 16-byte slot this lane's LDS position holds, for the wave's two instructions
  int drow[2], dslot[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    drow[q] = (wv + 8 * q) * 8 + (lane >> 3);
    dslot[q] = ((lane & 7) ^ ((drow[q] >> 1) & 7)) * 16;  // bytes
  }
  // fragment addressing: lane l holds row (l & 15), k = 8 (l >> 4) .. + 7 of a 16 x 32 block: slot 4 ks + (l >> 4), swizzled
  const int frow = lane & 15, fsw = (lane >> 1) & 7;
  const int foff0 = frow * 8 + ((lane >> 4) ^ fsw);        // k sub-step 0
  const int foff1 = frow * 8 + ((4 + (lane >> 4)) ^ fsw);  // k sub-step 1
  const int xbase = wr * 1024;                              // this wave's activation half tile
  const int wbase = 2048 + (wc >> 1) * 1024 + (wc & 1) * 64 * 8;  // its 64 output channels inside their half tile

  // ---- persistent blocks: block b walks the tiles b, b + gridDim.x, ... of the virtual grid (cout tiles x row tiles), re-numbered
  // XCD-aware like conv_gemm_f32 (workgroups b and b + 8 share an XCD, so do the virtual blocks they walk; speed only).
  // The epilogue's stores of one tile drain while the next tile's K loop runs.
  const unsigned gx = a.gemm16_gx;
  unsigned nvirt = gx * (unsigned)a.tiles_y;
  if (a.capacity) {  // capacity segments: tiles_y counts the host's upper bounds; walk (and renumber over) the real row tiles only
    int tot = 0;
    for (int u0 = 0; u0 < a.n_utt; u0 += 64) {
      const int u = u0 + lane;
      tot += u < a.n_utt ? (a.seg_off[u + 1] - a.seg_off[u] + kG16Tile - 1) / kG16Tile : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    nvirt = gx * (unsigned)__builtin_amdgcn_readfirstlane(tot);
  }
  // Tile directory (batches of up to 64 utterances): lane u's exclusive / inclusive prefix of 256-row tiles, first row and length, computed ONCE per
  // block and parked in LDS - per tile the lookup is then a ballot and three LDS reads instead of two global loads, a 6-step scan and two dependent
  // scalar loads (~1.5 us per tile in the block timeline: as much as a K tile).
  const bool dir_ok = a.n_utt <= 64;
  int* const dir = reinterpret_cast<int*>(lds + 2 * 4 * 1024);  // [4][64]: excl, incl, lo, len
  if (dir_ok && wv == 0) {
    const int lo_u = lane < a.n_utt ? a.seg_off[lane] : 0, len_u = lane < a.n_utt ? a.seg_off[lane + 1] - lo_u : 0;
    const int tiles = (len_u + kG16Tile - 1) / kG16Tile;
    int incl = tiles;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int vv = __shfl_up(incl, d, 64);
      if (lane >= d) incl += vv;
    }
    dir[lane] = incl - tiles;
    dir[64 + lane] = incl;
    dir[128 + lane] = lo_u;
    dir[192 + lane] = len_u;
  }
  if (dir_ok) __syncthreads();
#ifdef STTS_GEMM_TRACE
  long long tr_acc[6] = {0, 0, 0, 0, 0, 0}, tr_t = wall_clock64();
  auto lap = [&](int i) { const long long n = wall_clock64(); tr_acc[i] += n - tr_t; tr_t = n; };
#else
  auto lap = [](int) {};
#endif
  for (unsigned v = blockIdx.x; v < nvirt; v += gridDim.x) {
    int bx, by;
    {
      const unsigned q = nvirt >> 3, r = nvirt & 7, xcd = v & 7;
      const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
      bx = id % gx;
      by = id / gx;
    }
    int utt = -1, local = 0, dir_lo = 0, dir_len = 0;
    if (dir_ok) {
      const int t = by + a.tile0;
      const int excl = dir[lane], incl = dir[64 + lane];
      const unsigned long long hit = __ballot(t >= excl && t < incl);
      if (hit) {
        const int src = __ffsll((long long)hit) - 1;
        utt = src;
        local = t - dir[src];
        dir_lo = dir[128 + src];
        dir_len = dir[192 + src];
      }
    } else {
      const int t = by + a.tile0;
      int base = 0;
      for (int u0 = 0; u0 < a.n_utt && utt < 0; u0 += 64) {
        const int u = u0 + lane;
        const int tiles = u < a.n_utt ? (a.seg_off[u + 1] - a.seg_off[u] + kG16Tile - 1) / kG16Tile : 0;
        int incl = tiles;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int vv = __shfl_up(incl, d, 64);
          if (lane >= d) incl += vv;
        }
        const int excl = incl - tiles;
        const unsigned long long hit = __ballot(t >= base + excl && t < base + incl);
        if (hit) {
          const int src = __ffsll((long long)hit) - 1;
          utt = u0 + src;
          local = t - base - __shfl(excl, src, 64);
        }
        base += __shfl(incl, 63, 64);
      }
    }
    if (utt < 0) continue;  // a row tile beyond the batch's last one (the host grid is an upper bound when the offsets live on the device)
    utt = __builtin_amdgcn_readfirstlane(utt);
    local = __builtin_amdgcn_readfirstlane(local);
    const int lo = dir_ok ? __builtin_amdgcn_readfirstlane(dir_lo) : a.seg_off[utt];
    const int len = dir_ok ? __builtin_amdgcn_readfirstlane(dir_len) : a.seg_off[utt + 1] - lo;
    const int hi = lo + len, rel0 = local * kG16Tile;
    const int row0 = lo + rel0;
    if (rel0 >= len) continue;
    const int m0 = bx * kG16Tile;

    // ---- staging cursors (scalar): X runs one tile ahead of the multiplies, W two.  Operands are fetched through buffer descriptors:
    // the activations' descriptor spans exactly the utterance's rows, so a conv tap that reaches before the first or past the last row
    // (zero padding) is an out-of-range offset and the hardware returns zeros - no per-lane select, no page of zeros.
    struct Cur {
      int s, tap, chunk;
      __amdgpu_buffer_rsrc_t xr, wr;
      int ldx2, kc, ntaps, dil, pad, wrow2;
    };
    const int nseg = MSEG ? a.nseg : 1;
    auto load_seg = [&](Cur& c, int s) {
      const GemmSeg& g = (!MSEG || s == 0) ? a.seg[0] : (s == 1 ? a.seg[1] : a.seg[2]);
      c.s = s;
      const unsigned short* X = reinterpret_cast<const unsigned short*>(g.X) + (long)lo * g.ldx + g.xcol0;
      const unsigned short* W = g.W16 + (long)utt * g.w_utt_stride + (long)m0 * g.ntaps * g.kc;
      c.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, ((len - 1) * g.ldx + g.kc) * 2, 0x00020000);
      c.wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, kG16Tile * g.ntaps * g.kc * 2, 0x00020000);
      c.ldx2 = g.ldx * 2; c.kc = g.kc; c.ntaps = g.ntaps; c.dil = g.dil; c.pad = g.pad; c.wrow2 = g.ntaps * g.kc * 2;
    };
    auto advance = [&](Cur& c) {  // tap is the inner index (the taps of one chunk re-read almost the same rows: L1 / L2 hits); clamps at the last tile
      if (c.tap + 1 < c.ntaps) { ++c.tap; return; }
      if ((c.chunk + 1) * kG16K < c.kc) { c.tap = 0; ++c.chunk; return; }
      if (MSEG && c.s + 1 < nseg) { load_seg(c, c.s + 1); c.tap = 0; c.chunk = 0; }
    };
    Cur cx, cw;
    load_seg(cx, 0); cx.tap = 0; cx.chunk = 0;
    cw = cx;
    int total = a.seg[0].ntaps * (a.seg[0].kc / kG16K);
    if (MSEG) {
      if (a.nseg > 1) total += a.seg[1].ntaps * (a.seg[1].kc / kG16K);
      if (a.nseg > 2) total += a.seg[2].ntaps * (a.seg[2].kc / kG16K);
    }
    auto issue_x = [&](const Cur& c, int h, int stage, bool loop = true) {
      if ((ABL & 1) && loop) return;
      const int srow = rel0 + h * 128 + (c.tap - c.pad) * c.dil;  // scalar: first row of the half tile, shifted by the tap
      const int scol = c.chunk * (kG16K * 2);                    // scalar: byte offset of the chunk
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int voff = (drow[q] + srow) * c.ldx2 + dslot[q] + scol;  // negative (rows before the utterance) = a huge unsigned offset: out of range
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (__attribute__((address_space(3))) void*)(lds + stage * 4096 + h * 1024 + (wv + 8 * q) * 64), 16, voff, 0, 0, 0);
      }
    };
    auto issue_w = [&](const Cur& c, int h, int stage, bool loop = true) {
      if ((ABL & 1) && loop) return;
      const int soff = h * 128 * c.wrow2 + (c.tap * c.kc + c.chunk * kG16K) * 2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int voff = drow[q] * c.wrow2 + dslot[q];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr, (__attribute__((address_space(3))) void*)(lds + stage * 4096 + 2048 + h * 1024 + (wv + 8 * q) * 64), 16, voff, soff, 0, 0);
      }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xf[4][2], wf[2][2][2];

    auto read_x = [&](int stage, int xs, bool first = false) {
      if ((ABL & 2) && !first) return;
      const f32x4* p = lds + stage * 4096 + xbase + xs * 64 * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i][0] = p[i * 128 + foff0];
        xf[i][1] = p[i * 128 + foff1];
      }
    };
    auto read_w = [&](int stage, int cs, bool first = false) {
      if ((ABL & 2) && !first) return;
      const f32x4* p = lds + stage * 4096 + wbase + cs * 32 * 8;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wf[cs][j][0] = p[j * 128 + foff0];
        wf[cs][j][1] = p[j * 128 + foff1];
      }
    };
    auto mma = [&](int xs, int cs) {  // quadrant (xs, cs): 4 x 2 tiles x 2 k sub-steps = 16 MFMAs
      if (ABL & 4) return;
      // priority for the MFMA cluster: with it the wave that multiplies wins the issue arbitration against its SIMD partner's read / DMA
      // segment (measured: output conv 699 -> 515 us; the partner's segment is a handful of instructions and still fits beside it)
      if (!(ABL & 32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[xs * 4 + i][cs * 2 + j] = g16_mfma<PREC>(wf[cs][j][ks], xf[i][ks], acc[xs * 4 + i][cs * 2 + j]);
      if (!(ABL & 32)) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: X(0), W(0), W(1)
    issue_x(cx, 0, 0, false);
    issue_x(cx, 1, 0, false);
    advance(cx);
    issue_w(cw, 0, 0, false);
    issue_w(cw, 1, 0, false);
    advance(cw);
    issue_w(cw, 0, 1, false);
    issue_w(cw, 1, 1, false);
    advance(cw);
    asm volatile("" ::: "memory");
    lap(1);
    STTS_G16_WAIT_VM(4);  // everything but W(1) has landed (this wave's share; the barrier covers the others') - and the previous tile's stores
    STTS_G16_BARRIER();
    lap(2);
    if (wr == 1) STTS_G16_BARRIER();  // the second wave group runs one barrier behind the first
    if (ABL & 2) {  // ablation: the fragments are read once, before the loop
      read_x(0, 0, true);
      read_w(0, 0, true);
      read_w(0, 1, true);
    }
#define STTS_G16_LOOP_BARRIER() do { if (!(ABL & 16)) STTS_G16_BARRIER(); } while (0)

#pragma unroll 1
    for (int t = 0; t < total; ++t) {
      const int st = t & 1;
      // phase 0
      read_x(st, 0);
      read_w(st, 0);
      issue_x(cx, 0, st ^ 1);
      STTS_G16_LOOP_BARRIER();
      STTS_G16_WAIT_LGKM0();
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 0);
      STTS_G16_LOOP_BARRIER();
      // phase 1
      read_w(st, 1);
      issue_x(cx, 1, st ^ 1);
      advance(cx);
      asm volatile("" ::: "memory");
      STTS_G16_WAIT_LGKM0();  // before the barrier: phase 2 re-fills the W half tiles
      STTS_G16_LOOP_BARRIER();
      mma(0, 1);
      STTS_G16_LOOP_BARRIER();
      // phase 2
      read_x(st, 1);
      issue_w(cw, 0, st);
      STTS_G16_LOOP_BARRIER();
      STTS_G16_WAIT_LGKM0();
      __builtin_amdgcn_sched_barrier(0);
      mma(1, 1);
      STTS_G16_LOOP_BARRIER();
      // phase 3
      issue_w(cw, 1, st);
      advance(cw);
      asm volatile("" ::: "memory");
      STTS_G16_WAIT_VM(4);  // all but W(t+2): X(t+1) (and W(t+1), older) have landed -> read from the next phase on
      STTS_G16_LOOP_BARRIER();
      mma(1, 0);
      STTS_G16_LOOP_BARRIER();
    }
    lap(3);
    if (wr == 0) STTS_G16_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead DMA of the clamped cursors has landed ...
    STTS_G16_BARRIER();                               // ... in every wave: the next tile's prologue may re-fill the stages

    lap(4);
    // ---- epilogue: acc[tm][tn][e] = output (time row wr*128 + tm*16 + (lane & 15), channel wc*64 + tn*16 + 4*(lane >> 4) + e)
    const int nvalid = len - rel0;
    const bool hasR = a.R != nullptr, hasY = a.Y != nullptr, hasY16 = a.Y16 != nullptr, hasSS = a.sumsq_part != nullptr;
    const int act = a.act;
    const float alpha = a.alpha;
    // (opaque per tile: the 96 per-lane store / residual offsets derived from these are invariant across the tiles a persistent block walks,
    //  and hoisted out of the tile loop they would live - spilled - through every K loop)
    int tl = wr * 128 + frow;                          // + tm * 16
    int cl = wc * 64 + 4 * (lane >> 4);                // + tn * 16
    asm volatile("" : "+v"(tl), "+v"(cl));
    f32x4 bv[4];
    bool nok[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = m0 + cl + tn * 16;
      bv[tn] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      nok[tn] = n < a.N;  // N is a multiple of 4 (launcher)
    }
    float* const Yb = hasY ? a.Y + (long)row0 * a.ldy + a.ycol0 + m0 + cl : nullptr;
    // residual rows through a buffer descriptor over this tile's valid rows (rows past them read as zeros and are not stored), fetched ONE ROW TILE
    // AHEAD of their use: with the load inside the guarded store block every one of the 32 accumulator tiles waited for its own round trip
    // (pwconv2: 29 us of epilogue per 256 x 256 tile, block timeline of round 3)
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(hasR ? a.R + (long)row0 * a.ldr + a.rcol0 + m0 : a.zeros), 0, hasR ? (unsigned)(((long)(nvalid < kG16Tile ? nvalid : kG16Tile) * a.ldr) * 4) : 0u, 0x00020000);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    f32x4 rnext[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto load_r = [&](int tm) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) rnext[tn] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rr, ((tl + tm * 16) * a.ldr + cl + tn * 16) * 4, 0, 0));
    };
    if (!MSEG && hasR) load_r(0);
    const float* const Rb = hasR ? a.R + (long)row0 * a.ldr + a.rcol0 + m0 + cl : nullptr;  // (multi-segment kernels: the residual is read where it is used)
    unsigned short* const Y16b = hasY16 ? a.Y16 + (long)row0 * a.ldy16 + a.ycol16 + m0 + cl : nullptr;
    const bool hasST = a.stat_part != nullptr;
    f32x4 ss[4], su[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) ss[tn] = su[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The unrolled 8 x 4 store loop, SPECIALISED at compile time on what this launch's epilogue does: with `act`, `R`, `Y` and `Y16` as run-time
    // values the loop body carried ~1 300 scalar branches per tile (the activation switch per element, the output / residual tests per accumulator
    // tile) - 10-14 us of epilogue per 256 x 256 tile, as much as a short K loop (block timeline of round 3).
    auto store_loop = [&](auto gen_c, auto silu_c, auto hr_c, auto hy_c, auto hy16_c) {
      constexpr bool GEN = decltype(gen_c)::value;  // generic body: everything decided at run time
      constexpr bool SILU = decltype(silu_c)::value;
      const bool HR = GEN ? hasR : decltype(hr_c)::value, HY = GEN ? hasY : decltype(hy_c)::value, HY16 = GEN ? hasY16 : decltype(hy16_c)::value;
#pragma unroll
      for (int tm = 0; tm < 8; ++tm) {
        const int tr = tl + tm * 16;
        const bool rok = tr < nvalid;
        f32x4 rcur[4];
        if constexpr (!MSEG) {
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) rcur[tn] = rnext[tn];
          if (HR && tm + 1 < 8) load_r(tm + 1);
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          f32x4 vv = acc[tm][tn] + bv[tn];
          if constexpr (GEN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = act_apply(vv[e], act);
          } else if constexpr (SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = act_apply(vv[e], ACT_SILU);
          }
          if (rok && nok[tn]) {
            if constexpr (MSEG) {
              if (HR) vv += *reinterpret_cast<const f32x4*>(Rb + (unsigned)(tr * a.ldr + tn * 16));
            } else {
              if (HR) vv += rcur[tn];
            }
            vv *= alpha;
            if (HY && !((ABL & 8) && vv[0] != 12345.f)) {
              if (ABL & 64) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(Yb + (unsigned)(tr * a.ldy + tn * 16)));
              else *reinterpret_cast<f32x4*>(Yb + (unsigned)(tr * a.ldy + tn * 16)) = vv;
            }
            if (HY16 && !((ABL & 8) && vv[0] != 12345.f)) {
              if (ABL & 64) __builtin_nontemporal_store(pack4_16<PREC>(vv), reinterpret_cast<u32x2*>(Y16b + (unsigned)(tr * a.ldy16 + tn * 16)));
              else *reinterpret_cast<u32x2*>(Y16b + (unsigned)(tr * a.ldy16 + tn * 16)) = pack4_16<PREC>(vv);
            }
            ss[tn] += vv * vv;
            su[tn] += vv;
          }
        }
      }
    };
    {
      using T = std::true_type;
      using F = std::false_type;
      bool done = false;
      // (the multi-segment kernel gets only its own two: with all five bodies it went from 230 to 256 registers + spills inside the K loop,
      //  conv2 + shortcut 165 -> 245 us)
      if (ABL == 0 && !(a.tune & 0x4000)) {  // (tune bit 0x4000, STTS_G16_GENERIC=1: A/B switch for the generic body)
        done = true;
        if constexpr (MSEG) {
          if (act == ACT_NONE && !hasR && hasY && hasY16) store_loop(F{}, F{}, F{}, T{}, T{});        // conv2 + shortcut: fp32 rows and their rounded copy
          else if (act == ACT_NONE && !hasR && hasY && !hasY16) store_loop(F{}, F{}, F{}, T{}, F{});  // projector
          else done = false;
        } else {
          if (act == ACT_SILU && !hasR && !hasY && hasY16) store_loop(F{}, T{}, F{}, F{}, T{});       // pwconv1: SiLU -> 16-bit rows (+ GRN sums)
          else if (act == ACT_NONE && !hasR && hasY && !hasY16) store_loop(F{}, F{}, F{}, T{}, F{});  // fp32 rows: conv1, output convs
          else if (act == ACT_NONE && hasR && hasY && !hasY16) store_loop(F{}, F{}, T{}, T{}, F{});   // pwconv2: + residual
          else if (act == ACT_NONE && !hasR && !hasY && hasY16) store_loop(F{}, F{}, F{}, F{}, T{});  // 16-bit rows only (prior convs)
          else done = false;
        }
      }
      if (!done) store_loop(T{}, F{}, F{}, F{}, F{});
    }
    if (hasST) {
      // AdaIN statistics of this wave's 128 rows = chunk 2 local + wr of the utterance (adain_partial_kernel's chunks are 128 rows from the
      // utterance's first): mean and sum of squared deviations per channel from the sums over the valid rows (fp32: n <= 128 values of O(1))
      const int chunk = local * 2 + wr, nc = min(128, len - chunk * 128);
      if (nc > 0) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          f32x4 s1 = su[tn], s2 = ss[tn];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s1[e] += __shfl_xor(s1[e], o, 64);
              s2[e] += __shfl_xor(s2[e], o, 64);
            }
          if (frow == 0 && nok[tn]) {
            const f32x4 mean = s1 * (1.0f / (float)nc);
            f32x4 m2 = s2 - mean * s1;
#pragma unroll
            for (int e = 0; e < 4; ++e) m2[e] = fmaxf(m2[e], 0.0f);
            float* p = a.stat_part + ((long)(utt * a.stat_nchunk + chunk) * 2) * a.ld_stat + m0 + cl + tn * 16;
            *reinterpret_cast<f32x4*>(p) = mean;
            *reinterpret_cast<f32x4*>(p + a.ld_stat) = m2;
          }
        }
      }
    }
    if (hasSS) {
      // GRN: per-channel sums of squares over this wave's 128 rows, written as the first of the four 32-row slots the consumer sums
      // (grn_gx_kernel reads ceil(len / 32) slots per utterance); the other three slots of the group get zeros
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f32x4 sq = ss[tn];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) sq[e] += __shfl_xor(sq[e], o, 64);
        if (frow < 4 && nok[tn]) {
          const int sub = local * 8 + wr * 4 + frow;  // 32-row slot of the utterance
          if (sub * 32 < len) {
            const long slot = (long)utt * a.ss_stride + sub;
            *reinterpret_cast<f32x4*>(a.sumsq_part + slot * a.ld_ss + m0 + cl + tn * 16) = frow == 0 ? sq : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
    lap(5);
#ifdef STTS_GEMM_TRACE
    ++tr_acc[0];
#endif
  }
#ifdef STTS_GEMM_TRACE
  if (a.dbg && tid == 0)
    for (int i = 0; i < 6; ++i) a.dbg[8 * (long)blockIdx.x + i] = tr_acc[i];
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

// Can this contraction run on conv_gemm16_kernel?  16-bit activation rows, store epilogue, every segment's channels a multiple of 64,
// cout padded to 256, N a multiple of 4, host offsets known (compact grid).
inline bool gemm16_eligible(const GemmArgs& a, int epi, int npad) {
  if (epi != EPI_STORE || a.prec == PREC_F32 || !a.x16 || !a.seg_host || a.xaff || npad % kG16Tile != 0 || a.N % 4 != 0) return false;
  if (a.ldy % 4 || a.ycol0 % 4 || a.ldr % 4 || a.rcol0 % 4 || a.ldy16 % 4 || a.ycol16 % 4 || a.ld_ss % 4 || a.ld_stat % 4) return false;
  for (int i = 0; i < a.nseg; ++i)
    if (a.seg[i].kc % kG16K != 0 || a.seg[i].ldx % 8 != 0 || a.seg[i].xcol0 % 8 != 0 || !a.seg[i].W16 || a.seg[i].ldx - a.seg[i].xcol0 < a.seg[i].kc) return false;
  return true;
}

inline bool gemm16_will_run(const GemmArgs& a, int epi, int npad, int n_utt) {
  static const long min_tiles = getenv("STTS_GEMM16_MIN_TILES") ? atol(getenv("STTS_GEMM16_MIN_TILES")) : 192;
  return gemm16_eligible(a, epi, npad) && gemm16_tiles(a, npad, n_utt) >= min_tiles;
}

// 256 x 256 tiles of the launch: exact from the host offsets (an upper bound when they are capacities)
inline long gemm16_tiles(const GemmArgs& a, int npad, int n_utt) {
  long rt = 0;
  for (int u = 0; u < n_utt; ++u) rt += ceil_div(a.seg_host[u + 1] - a.seg_host[u], kG16Tile);
  return rt * (npad / kG16Tile);
}

template <int ABL>
inline int launch_conv_gemm16(hipStream_t st, const GemmArgs& a, int npad, int n_utt) {
  STTS_CHECK(gemm16_eligible(a, EPI_STORE, npad), "conv_gemm16: contraction not eligible");
  const long rt = gemm16_tiles(a, npad, n_utt) / (npad / kG16Tile);
  GemmArgs as = a;
  as.n_utt = n_utt;
  as.compact = 1;
  as.tile0 = 0;
  as.tiles_y = (int)rt;
  as.ksplit = 1;
  as.gemm16_gx = npad / kG16Tile;
  static const bool force_generic = getenv("STTS_G16_GENERIC") != nullptr;
  if (force_generic) as.tune |= 0x4000;
  GemmProfiler& prof = gemm_profiler();
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof.on) {
    e0 = prof.next();
    e1 = prof.next();
    const double fl = gemm_algorithmic_flops(a);
    prof.add("conv_gemm16_kernel", 0, fl, fl, 0.0);
  }
  // persistent: one block per CU (128 KB of LDS each) walks the virtual grid of (npad / 256) x rt tiles
  const long nvirt = (long)(npad / kG16Tile) * rt;
  const dim3 grid((unsigned)std::min<long>(nvirt, 256)), block(512);
  if (a.prec == PREC_BF16) {
    if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm16_kernel<PREC_BF16, false, ABL>), grid, block, st, e0, e1, as);
    else STTS_LAUNCH_TIMED((conv_gemm16_kernel<PREC_BF16, true, ABL>), grid, block, st, e0, e1, as);
  } else if constexpr (ABL == 0) {
    if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm16_kernel<PREC_F16, false>), grid, block, st, e0, e1, as);
    else STTS_LAUNCH_TIMED((conv_gemm16_kernel<PREC_F16, true>), grid, block, st, e0, e1, as);
  }
  STTS_HIP(hipGetLastError());
  return 0;
}

#ifdef STTS_GEMM16_TU
int launch_conv_gemm16_main(hipStream_t st, const GemmArgs& a, int npad, int n_utt) { return launch_conv_gemm16<0>(st, a, npad, n_utt); }
#endif

}  // namespace stts
