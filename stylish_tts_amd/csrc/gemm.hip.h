// conv_gemm_f32: the dense contraction of the hot path (Conv1d k=1..7 and Linear) on the gfx950
// f32-input matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 accumulate chain).
//
// Replaces every F.conv1d / nn.Linear on the reference's frame-rate path
// (models/ada_norm.py:158-163 conv1/conv2/conv1x1, models/flow.py:39-60,182-192 WN + projections,
// models/generator.py:344-386 prior/projector/output convs, :462-467 ConvNeXt pwconv1/2).
//
// Data layout (HBM): activations are TIME-MAJOR packed rows  X[row = utterance offset + frame][channel],
// leading dimension a multiple of 32 floats (128 B rows, zero padded).  Weights are packed
// W[cout_padded][tap][cin_padded] so that the contraction index (tap, cin) is contiguous for both MFMA
// operands.  A conv tap is a ROW shift of X, so every LDS/VGPR access stays 16-byte aligned; utterance
// boundaries provide the conv zero padding (rows outside [seg_off[u], seg_off[u+1]) read as 0).
//
// Tiling: block = BN time rows x BM output channels, WARPS_N x WARPS_M waves of 64 lanes; each wave owns
// (BN/WARPS_N) x (BM/WARPS_M) as 32x32 accumulator tiles (rows = time on the A operand, cols = cout on the
// B operand, so a store instruction writes 32 consecutive output channels = 128 B).  K advances 32
// channels per iteration per (segment, tap); tiles are staged global -> VGPR -> LDS (double buffered, one
// barrier per iteration) with a 16-byte-slot XOR swizzle (slot ^= (row>>1)&7) that makes both the
// ds_write_b128 and the ds_read_b128 fragment reads conflict-free (guide §LDS, 64-bank b128 groups).
// Up to 3 input segments accumulate into one output (conv + 1x1 shortcut; concat inputs), and the
// epilogue fuses bias / activation / residual / gates so no extra pass over the output is needed.
#pragma once
#include <type_traits>
#include <mutex>
#include <map>
#include <hip/hip_ext.h>
#include <cmath>
#include <algorithm>
#include <vector>

#include "common.h"

namespace stts {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;  // native vector: stays in VGPRs (HIP's float4 struct copies via memcpy)

enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_RELU = 3, ACT_LRELU = 4 };
enum Epi {
  EPI_STORE = 0,      // Y = (act(acc + bias) [+ R]) * alpha ; optional per-tile sum of squares
  EPI_GATE = 1,       // paired: Y = tanh(a + g[utt][c]) * sigmoid(b + g[utt][C + c])      (flow.py:7-14)
  EPI_SPLIT_ACC = 2,  // cols < nsplit: D0 (+)= v ; cols >= nsplit: D1 (+)= v                 (flow.py:80-87)
  EPI_COUPLE = 3,     // paired: Z1 = (Z1 - a) * exp(-b)                                      (flow.py:209)
  EPI_PRIOR = 4,      // paired: Z = a + noise * exp(b)                                       (flow.py:314)
};

struct GemmSeg {
  const float* X;     // time-major activations
  const float* W;     // packed [Npad][ntaps][kc]
  const unsigned short* W16;  // the same tensor rounded to bf16 / fp16 (16-bit operand modes); PREC_X3: the three bf16 planes of the exact split
  long w16_plane;     // PREC_X3: elements between two planes of W16 (0: this weight has no split form)
  long x_plane;       // PREC_X3 with pre-split activations (GemmArgs::x16): X points to the first of three bf16 planes [rows, ldx], x_plane elements apart
  long w_utt_stride;  // floats between per-utterance copies of W (0: shared)
  int ldx, xcol0, kc, ntaps, dil, pad;
  int kreal;  // un-padded input channels (host side: algorithmic FLOP accounting only)
};

// Filled by the launcher when the caller takes over the split-K reduction (GemmArgs::defer): ksplit == 1 means the
// contraction wrote Y itself.
struct SplitInfo {
  const float* partial;
  int ksplit, slice_rows, ld_part;
};

struct GemmArgs {
  GemmSeg seg[3];
  int nseg;
  const int* seg_off;  // device [n_utt + 1], rows
  const int* seg_host;  // the same offsets on the host (launcher only): exact tile counts for mixed-length batches
  int n_utt;
  int compact;          // grid.y enumerates only the row tiles that exist (sum over utterances), grid.z = split-K slice
  int uniform_len, uniform_lo0;  // > 0: every utterance has this many rows, the first starts at row uniform_lo0 (host side knows: the tile lookup is then
                        // a division - no offset loads, no scan; equal-length batches, B = 1, the planes of a Winograd-form conv)
  int tile0, tiles_y;   // compact: this launch covers global row tiles [tile0, tile0 + tiles_y)
  int gemm16_gx;        // conv_gemm16_kernel (persistent blocks): cout tiles of the virtual grid
  int capacity;         // seg_host holds UPPER BOUNDS of the utterance lengths (the real offsets live only on the device): grids are sized from
                        // them and blocks beyond the device-side tile count exit; plans that need exact host offsets are off
  int rows_total, wrows;  // host side: rows of the call, un-padded weight rows (FLOP accounting only)
  int tune;               // experiment switches (tools/gemm_bench.py ablations)
  int prec;               // PREC_F32 / PREC_BF16 / PREC_F16 operands (every segment then carries W16)
  // input affine of segment 0 (AdaIN folded into the staging): x' = lrelu_0.2(x * xaff[u][0][c] + xaff[u][1][c]); EPI_STORE only
  const float* xaff;
  int ld_xaff;
  float xaff_slope;       // negative-side slope of that activation (0.2: LeakyReLU; 1: a pure per-channel scale / shift, e.g. GRN folded into pwconv2's staging)
  bool xaff_scale_only;   // the shift row is all zeros, the slope 1 and no input column is padding (kc == channels): x' = x * scale, one multiply per element
                          // (split-fp32 single-segment launches have an instantiation for it, XAFF = 2; elsewhere the general form runs - same result)
  long long* dbg;         // block-timeline records (only written when built with -DSTTS_GEMM_TRACE; tools/gemm_bench.py)
  const float* zeros;     // >= 16 bytes of zeros in global memory (source of out-of-utterance rows for the LDS-DMA path)
  int ksplit;             // > 1: grid.z = n_utt * ksplit, block (u, ks) contracts a 1/ksplit slice of K into partial[ks]
  float* partial;         // [ksplit][rows_total][ld_part] raw partial sums (EPI_STORE only; splitk_reduce_kernel finishes)
  int ld_part;
  SplitInfo* defer;       // host side: non-null = do not launch the reduce pass, report the partials instead (e.g. to a LayerNorm)
  int N;               // output channels actually stored (paired epilogues: channels of the result)
  const float* bias;   // [Npad] in packed row order, may be null
  // 16-bit operand modes: every segment's X points to 16-bit rows already rounded to `prec` (ldx / xcol0 in elements);
  // set by the callers whose producers write 16-bit activations (conv_gemm_f32<..., X16>)
  int x16;
  // EPI_STORE
  float* Y;            // may be null when only the 16-bit copy is wanted
  int ldy, ycol0;
  unsigned short* Y16; // optional: the same values rounded to `prec` (operand of the next contraction), [rows, ldy16]
  int ldy16, ycol16;
  const float* R;
  int ldr, rcol0;
  float alpha;
  int act;
  float* sumsq_part;  // optional: [utt * ss_stride + (32-row sub-tile index within the utterance)][ld_ss]
  int ld_ss, ss_stride;
  // optional (conv_gemm16_kernel only): AdaIN statistics of the OUTPUT, per utterance and 128-row chunk, in adain_partial_kernel's layout
  // stat_part[((utt * stat_nchunk + chunk) * 2 + {0: mean, 1: sum of squared deviations}) * ld_stat + channel] - the consumer's statistics pass disappears
  float* stat_part;
  int ld_stat, stat_nchunk;
  // EPI_GATE
  const float* gate;  // [n_utt][ld_gate]; a-part at gcol0 + c, b-part at gcol0 + gC + c
  int ld_gate, gcol0, gC;
  // EPI_SPLIT_ACC
  float* D0;
  float* D1;
  int ldd0, ldd1, nsplit, acc0, acc1;
  // EPI_COUPLE / EPI_PRIOR
  float* Z;
  int ldz, zcol0;
  const float* noise;
  int ldnoise;
};

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    // (v_rcp_f32, 1 ulp, instead of the IEEE division sequence: at 94 M outputs the ~12 extra instructions per element were most of pwconv1's epilogue)
    case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_LRELU: return v >= 0.0f ? v : 0.2f * v;
    // (keep this switch small: it is inlined per element into every contraction epilogue and into the LayerNorm kernel - a Mish case
    //  here (tanh, log1p, exp) cost the bf16 B = 64 frame path 7 %; cfm.hip.h applies Mish in its own pass)
    default: return v;
  }
}

// 16-bit operand modes (PREC 1 = bf16, 2 = fp16): activations stay fp32 in HBM and are rounded (RNE) when a tile is
// staged into LDS, weights are stored pre-rounded ([cout][tap][cin] 16-bit), products accumulate in fp32 on
// v_mfma_f32_32x32x16_{bf16,f16}.  A 32-channel K chunk is then 64 bytes per tile row (4 slots of 16 bytes, swizzled by
// (row >> 2) & 3) and two MFMAs instead of sixteen.
//
// PREC_X3 ("split fp32"): the operands ARE the fp32 values.  Every fp32 number is the exact sum of three bf16 numbers
// (x = x0 + x1 + x2, 3 x 8 significand bits, each term the round-to-nearest bf16 of what the previous ones left), so
// x * w = sum of nine bf16 x bf16 products, each exact in fp32.  The kernel runs the six largest of them
// (x0 w0, x0 w1, x1 w0, x0 w2, x1 w1, x2 w0) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped terms are
// at most 2^-23 |x w| (one fp32 ulp of the product), 2^-27 |x w| rms and zero-mean (tests/test_split_fp32_cpu.py) - a quarter of
// the rms error of rounding the product to fp32 once, and far below the rounding of the fp32 accumulation that both forms share
// (tests/test_hip_split_fp32.py measures both forms against float64).  Weights are split once at pack time (three planes in W16,
// w16_plane apart), activations stay fp32 in HBM and are split when a tile is staged into LDS.  Six bf16 MFMAs do the work of
// sixteen v_mfma_f32_32x32x2_f32 in 3/8 of the matrix-pipe cycles.
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2, PREC_X3 = 3 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int PREC>
__device__ __forceinline__ u32x2 pack4_16(const f32x4 v) {
  u32x2 r;
  if constexpr (PREC == PREC_BF16) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, b2));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, b2));
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, h2));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, h2));
  }
  return r;
}
// exact three-term bf16 split of four fp32 values: v = p0 + p1 + p2 (each u32x2 = four bf16 in channel order)
__device__ __forceinline__ void split3_bf16(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  auto pk = [](float a, float b) { const f2 t = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(t, b2)); };  // v_cvt_pk_bf16_f32 (RNE)
  auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
  auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
  p0.x = pk(v.x, v.y); p0.y = pk(v.z, v.w);
  const float r0 = v.x - lo(p0.x), r1 = v.y - hi(p0.x), r2 = v.z - lo(p0.y), r3 = v.w - hi(p0.y);  // exact
  p1.x = pk(r0, r1); p1.y = pk(r2, r3);
  p2.x = pk(r0 - lo(p1.x), r1 - hi(p1.x)); p2.y = pk(r2 - lo(p1.y), r3 - hi(p1.y));                 // exact remainders, representable in bf16
}
template <int PREC>
__device__ __forceinline__ f32x16 mfma16(const f32x4 a, const f32x4 b, const f32x16 c) {
  if constexpr (PREC == PREC_BF16 || PREC == PREC_X3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// KSPLIT = 2: two wave groups share every staged tile and split its 32-channel chunk in halves (kk 0,1 / kk 2,3); their
// partial accumulators are summed through LDS before the epilogue.  Doubles the waves per SIMD for launches that only
// have ~one 128x128 tile per CU (B = 8: every 512-channel layer), which is where the matrix pipe otherwise idles.
template <int BM, int BN, int WARPS_M, int WARPS_N, int EPI, int KSPLIT = 1, bool GLDS = false, int PREC = PREC_F32, int XAFF = 0, bool MSEG = true,
          bool X16 = false>
__global__ void __launch_bounds__(WARPS_M* WARPS_N * 64 * KSPLIT) conv_gemm_f32(const GemmArgs a) {
  static_assert(!(XAFF && GLDS), "the input affine lives on the register staging path");
  static_assert(!X16 || (PREC != PREC_F32 && !XAFF && EPI == EPI_STORE), "16-bit activation rows: 16-bit operand modes, store epilogue");
  constexpr int NT = WARPS_M * WARPS_N * 64 * KSPLIT;
  constexpr bool B16 = PREC != PREC_F32;
  constexpr bool X3 = PREC == PREC_X3;  // split fp32: three bf16 planes per operand, six MFMAs per (tile, 16 channels)
  constexpr int NPL = X3 ? 3 : 1;
  static_assert(!X3 || !GLDS || X16, "split fp32: fp32 activation rows are split on the register staging path; LDS-DMA needs pre-split planes");
  static_assert(!(B16 && GLDS) || X16, "LDS-DMA staging of 16-bit operands needs 16-bit activation rows (nothing converts on the way)");
  static_assert(KSPLIT == 1 || KSPLIT == 2, "KSPLIT");
  constexpr int WR = BN / WARPS_N, WC = BM / WARPS_M;
  constexpr int TR = WR / 32, TC = WC / 32;
  constexpr int XL = X16 ? (BN * 4 * NPL + NT - 1) / NT : BN * 8 / NT;  // 16-byte X loads per thread per tile (16-bit rows: 4 per 32 channels and plane)
  constexpr int WL = B16 ? (BM * 4 * NPL + NT - 1) / NT : BM * 8 / NT;  // 16-byte W loads per thread per tile
  static_assert(WR % 32 == 0 && WC % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert((X16 || (BN * 8) % NT == 0) && (B16 || (BM * 8) % NT == 0), "tile loads must divide over the block");
  static_assert(EPI == EPI_STORE || EPI == EPI_SPLIT_ACC || TC % 2 == 0, "paired epilogues need an even TC");
  // two stages of [X rows | W rows] x 32 channels (16-bit operands: half of it; the K-group reduction of KSPLIT = 2 needs the full
  // size); the 16-bit LDS-DMA path keeps three stages so that tiles are requested two iterations ahead
  // (three for the 256-row tiles; EIGHT for the 128 x 128 tile of small batches, whose K iteration - four MFMAs per wave - is far
  //  shorter than a memory round trip: with tiles requested two iterations ahead that loop ran at one iteration per ~0.6 us)
  constexpr int NSTAGE = (B16 && GLDS) ? (X3 ? 3 : ((BM + BN) <= 256 ? 8 : 3)) : 2;
  // 16-byte slots per tile row and stage: fp32 8 (32 channels); 16-bit 4 per plane; split fp32: planes [p][X rows | W rows][4]
  constexpr int SLOTS = X3 ? 12 : ((B16 && KSPLIT == 1) ? 4 : 8);
  constexpr int PLANE = (BN + BM) * 4;      // 16-bit layouts: f32x4 slots of one plane of one stage
  constexpr int STG16 = PLANE * NPL;        // ... of one stage
  __shared__ f32x4 lds[NSTAGE * (BN + BM) * SLOTS];

  // XCD-aware block -> tile map (guide T1; speed only, any placement is correct): workgroups are dealt round-robin
  // over the 8 XCDs, each with its own L2.  Re-number them so that one XCD works through a CONTIGUOUS range of
  // tiles with the cout tile as the fastest index: all cout tiles of a row tile then share their X rows in one L2.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    unsigned gy = gridDim.y;
    if (a.capacity) {
      // Capacity segments: grid.y counts the row tiles of the host's UPPER BOUNDS; the real count comes from the device offsets.
      // Blocks beyond it leave at once, and the XCD-aware renumbering below runs over the real tiles only - over the padded
      // grid the empty tail would fall to the last XCDs and leave them idle (measured: +32 % at 1.4 x capacity).
      const int lane_ = threadIdx.x & 63;
      int tot = 0;
      for (int u0 = 0; u0 < a.n_utt; u0 += 64) {
        const int u = u0 + lane_;
        tot += u < a.n_utt ? (a.seg_off[u + 1] - a.seg_off[u] + BN - 1) / BN : 0;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
      // (every lane holds the same sum, but a cross-lane read is divergent to the compiler: without the readfirstlane the tile indices, the K-loop
      //  cursor and the segment fields all lived in VGPRs, and every cursor advance was a v_cmp + s_and_saveexec branch that cut the K loop into basic blocks)
      gy = (unsigned)__builtin_amdgcn_readfirstlane(tot);
      if ((unsigned)by >= gy) return;
    }
    const unsigned gx = gridDim.x, nwg = gx * gy * gridDim.z;
    const unsigned orig = bx + gx * (by + gy * bz);
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    bx = id % gx;
    by = (id / gx) % gy;
    bz = id / (gx * gy);
  }
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
  int utt, ks;
  if (a.compact) {
    // Mixed-length batches: a (row tile, utterance) grid is mostly empty blocks and, worse, the contiguous per-XCD ranges
    // above would give whole utterances to one XCD (long ones = hot XCDs).  grid.y instead counts the row tiles that
    // exist; every wave finds its utterance with a prefix sum of tiles-per-utterance over its lanes.
    ks = bz;
    const int t = by + a.tile0, lane_ = threadIdx.x & 63;
    utt = a.n_utt - 1;
    int base = 0, local = 0;
    bool done = false;
    if (a.uniform_len > 0) {
      const int tpu = (a.uniform_len + BN - 1) / BN;
      utt = t / tpu;
      local = t - utt * tpu;
      done = utt < a.n_utt;
    }
    for (int u0 = 0; u0 < a.n_utt && !done && a.uniform_len <= 0; u0 += 64) {
      const int u = u0 + lane_;
      const int tiles = u < a.n_utt ? (a.seg_off[u + 1] - a.seg_off[u] + BN - 1) / BN : 0;
      int incl = tiles;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane_ >= d) incl += v;
      }
      const int excl = incl - tiles;
      const unsigned long long hit = __ballot(t >= base + excl && t < base + incl);
      if (hit) {
        const int src = __ffsll((long long)hit) - 1;
        utt = u0 + src;
        local = t - base - __shfl(excl, src, 64);
        done = true;
      }
      base += __shfl(incl, 63, 64);
    }
    if (!done) return;  // a row tile beyond the batch's last one: the host grid is an upper bound when the offsets live on the device
    utt = __builtin_amdgcn_readfirstlane(utt);  // (wave-uniform values behind cross-lane reads: scalar registers from here on, see above)
    by = __builtin_amdgcn_readfirstlane(local);
  } else {
    utt = bz / ksplit;
    ks = bz % ksplit;
  }
  const int lo = a.uniform_len > 0 ? a.uniform_lo0 + utt * a.uniform_len : a.seg_off[utt];
  const int hi = a.uniform_len > 0 ? lo + a.uniform_len : a.seg_off[utt + 1];
  const int row0 = lo + by * BN;
  if (row0 >= hi) return;
  const int m0 = bx * BM;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef STTS_GEMM_TRACE
  // per-block record [t_start, t_prologue_end, t_loop_end, t_end (10 ns ticks), iterations, HW_ID, XCC_ID, shader cycles]
  const unsigned dbg_blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  auto stamp = [&](int i) { if (a.dbg && tid == 0) a.dbg[8 * (long)dbg_blk + i] = wall_clock64(); };
  const long long dbg_c0 = clock64();
  if (a.dbg && tid == 0) { a.dbg[8 * (long)dbg_blk + 5] = __builtin_amdgcn_s_getreg(0xF804); a.dbg[8 * (long)dbg_blk + 6] = __builtin_amdgcn_s_getreg(0xF814); }
  auto stamp_end = [&]() { stamp(3); if (a.dbg && tid == 0) a.dbg[8 * (long)dbg_blk + 7] = clock64() - dbg_c0; };
  auto ablate = [&](int bit) { return (a.tune & bit) != 0; };  // timing-only ablations of the K loop (tools/gemm_bench.py)
#else
  auto stamp = [&](int) {};
  auto stamp_end = [&]() {};
  auto ablate = [](int) { return false; };
#endif
  stamp(0);
  const int kg = (tid >> 6) / (WARPS_M * WARPS_N), wid = (tid >> 6) % (WARPS_M * WARPS_N);  // K-group, wave position
  const int wn = wid / WARPS_M, wm = wid % WARPS_M;
  (void)kg;

  f32x16 acc[TR][TC];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- K-loop cursor.  The loop body is ONE basic block (no data-dependent branches): loads are unconditional from a
  // clamped row and zeroed by select, the cursor advances with scalar selects, and the current segment's fields sit in
  // scalar registers.  That lets the scheduler sink the address arithmetic, the global loads and the ds_writes into the
  // shadows of the MFMAs (one 32x32x2 MFMA occupies the matrix pipe for 64 cycles but issues in ~8).
  int s = 0, tap = 0, chunk = 0;
  // (X16: X is a 16-bit buffer behind the float pointer type; xcol0 / ldx count its elements)
  auto xbase = [&](const GemmSeg& g) {
    return X16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.X) + g.xcol0) : g.X + g.xcol0;
  };
  const float* gX = xbase(a.seg[0]);
  const float* gW = a.seg[0].W + (long)utt * a.seg[0].w_utt_stride + (long)m0 * a.seg[0].ntaps * a.seg[0].kc;
  const unsigned short* gW16 = B16 ? a.seg[0].W16 + (long)utt * a.seg[0].w_utt_stride + (long)m0 * a.seg[0].ntaps * a.seg[0].kc : nullptr;
  int g_ldx = a.seg[0].ldx, g_kc = a.seg[0].kc, g_ntaps = a.seg[0].ntaps, g_dil = a.seg[0].dil, g_pad = a.seg[0].pad;
  const int nseg = a.nseg;

  // Two named register sets: tiles are fetched TWO iterations ahead (an iteration is ~2 us of matrix-pipe time per
  // SIMD, about one loaded-memory round trip) and moved to LDS one iteration ahead.
  struct RegSet {
    f32x4 x[XL], w[WL];
    bool ok[XL];  // rows outside the utterance are zeroed when the registers are consumed (lstore), so the load stays in flight
    f32x4 sc, sh;  // XAFF: scale / shift of this thread's 4 channels (segment 0), identity elsewhere
    bool aff;
  };
  RegSet rsA, rsB;
  // split fp32: a K iteration is 3/8 of the fp32 one in matrix-pipe time (~0.7 us per 128 x 128 x 32), shorter than a loaded-memory round trip, so tiles
  // are fetched FOUR iterations ahead into four register sets
  // (measured, B = 8: no gain on the 128 x 128 tile - pwconv1 95 -> 98 us, decoder conv2 79 -> 78 - and a loss on the 128 x 64 tile, whose 128
  //  registers become 162 and halve its waves per SIMD: 84 -> 96 us; the loop is not waiting for its loads.  Kept behind a build flag.)
#ifdef STTS_X3_DEEP
  constexpr bool DEEP = X3 && KSPLIT == 1 && BN <= 128;
#else
  constexpr bool DEEP = false;
#endif
  RegSet rsC, rsD;
  // Addressing: uniform (scalar) base pointers + 32-bit per-thread byte offsets, so the loads use the saddr form and the
  // per-iteration vector arithmetic is a clamp, a multiply and an add per X row (the W offsets are loop invariant).
  // Offsets are relative to the utterance / the weight tile, hence always < 2^31 bytes.
  unsigned woff[WL];
  const int len = hi - lo, rel0 = row0 - lo;
  long g_wplane = X3 ? a.seg[0].w16_plane : 0, g_xplane = (X3 && X16) ? a.seg[0].x_plane : 0;
  auto seg_offsets = [&]() {
    const int wrow = g_ntaps * g_kc;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = tid + i * NT;
      if constexpr (X3) {  // plane-major: load idx covers (plane, row, 8 channels); the plane offset rides in the 32-bit byte offset (host-checked)
        const int pl = min(idx / (BM * 4), 2), rem = idx % (BM * 4);
        woff[i] = (unsigned)(((long)pl * g_wplane + (rem >> 2) * wrow + (rem & 3) * 8) * 2);
      } else if constexpr (B16) woff[i] = (unsigned)((min(idx >> 2, BM - 1) * wrow + (idx & 3) * 8) * 2);  // 8 x 16-bit per load
      else woff[i] = (unsigned)(((idx >> 3) * wrow + (idx & 7) * 4) * 4);
    }
  };
  seg_offsets();
  auto gload = [&](RegSet& rs) {
    const int shift = (tap - g_pad) * g_dil;
#ifdef STTS_GEMM_TRACE
    const bool hot = a.tune & 16;  // ablation only: every tile read hits the same few KB (L1/L2 resident) -> wrong results
#else
    constexpr bool hot = false;
#endif
    const char* xb = X16 ? reinterpret_cast<const char*>(reinterpret_cast<const unsigned short*>(gX) + (long)lo * g_ldx + chunk * 32)
                         : reinterpret_cast<const char*>(gX + (long)lo * g_ldx + (hot ? 0 : chunk * 32));
    const char* wb = B16 ? reinterpret_cast<const char*>(gW16 + tap * g_kc + chunk * 32)
                         : reinterpret_cast<const char*>((hot ? a.seg[0].W : gW) + (hot ? 0 : tap * g_kc + chunk * 32));
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = tid + i * NT;
      const int pl = (X16 && X3) ? min(idx / (BN * 4), 2) : 0, idp = (X16 && X3) ? idx % (BN * 4) : idx;  // pre-split planes: (plane, row, 8 channels)
      const int r = X16 ? idp >> 2 : idx >> 3, sl = X16 ? idp & 3 : idx & 7;
      const int rel = (hot ? 0 : rel0) + r + shift;
      const int crel = min(max(rel, 0), len - 1);
      // (24-bit multiply: full rate, v_mul_lo_u32 is quarter rate; rows and row strides are far below 2^24, byte offsets below 2^31 - host-checked)
      const unsigned rowoff = __umul24((unsigned)crel, (unsigned)g_ldx);
      const unsigned off = X16 ? (rowoff + sl * 8) * 2 : (rowoff + sl * 4) * 4;
      if constexpr (X16 && X3) rs.x[i] = *reinterpret_cast<const f32x4*>(xb + (long)pl * g_xplane * 2 + off);
      else
      rs.x[i] = *reinterpret_cast<const f32x4*>(xb + off);
      rs.ok[i] = rel >= 0 && rel < len;
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) rs.w[i] = *reinterpret_cast<const f32x4*>(wb + woff[i]);
    if constexpr (XAFF) {
      rs.aff = s == 0;
      const float* ap = a.xaff + (long)utt * 2 * a.ld_xaff + (rs.aff ? chunk * 32 + (tid & 7) * 4 : 0);
      rs.sc = *reinterpret_cast<const f32x4*>(ap);
      if constexpr (XAFF == 1) rs.sh = *reinterpret_cast<const f32x4*>(ap + a.ld_xaff);
    }
    // advance: TAP is the inner index, so the taps of one 32-channel chunk re-read (almost) the same rows of X
    // back to back and hit L1/L2; channel-inner order re-streamed the whole X tile per tap from the fabric (rocprof
    // FETCH_SIZE was ~10x the compulsory bytes).  Scalar selects only; the last tile is re-loaded when the cursor
    // would run off the end.
    // (bitwise, not short-circuit: the compiler turned `&&` into a scalar branch around the chunk update, one more block boundary in the loop)
    const int w1 = tap + 1 >= g_ntaps, wk = (chunk + 1) * 32 >= g_kc;
    const bool wrapt1 = w1 != 0;
    const bool wrapt = (w1 & wk) != 0;  // segment finished
    const bool last = ((w1 & wk) & (MSEG ? (int)(s + 1 >= nseg) : 1)) != 0;
    tap = last ? tap : (wrapt1 ? 0 : tap + 1);
    chunk = last ? chunk : (wrapt ? 0 : chunk + w1);
    if constexpr (MSEG) {  // (single-segment launches are compiled without this branch: the K loop is one basic block)
      if (wrapt && !last) {  // uniform, taken at most twice per kernel
        ++s;
        const GemmSeg& n = s == 1 ? a.seg[1] : a.seg[2];
        gX = xbase(n);
        gW = n.W + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
        if constexpr (B16) gW16 = n.W16 + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
        g_ldx = n.ldx; g_kc = n.kc; g_ntaps = n.ntaps; g_dil = n.dil; g_pad = n.pad;
        if constexpr (X3) { g_wplane = n.w16_plane; g_xplane = n.x_plane; }
        seg_offsets();
      }
    }
  };
  auto xin = [&](const RegSet& rs, int i) {  // the value of X load i that enters LDS
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v = rs.x[i];
    if constexpr (XAFF == 2) {  // scale only (GRN in front of pwconv2): single segment, so every tile takes it
      static_assert(XAFF != 2 || !MSEG, "the scale-only input affine is instantiated for single-segment launches");
      v = v * rs.sc;
    } else if constexpr (XAFF == 1) {
      f32x4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // scale 0 marks a pad column: its X value may be uninitialised memory (NaN * 0 would poison the tile)
        const float y = (rs.sc[q] == 0.0f ? 0.0f : v[q] * rs.sc[q]) + rs.sh[q];
        t[q] = y >= 0.0f ? y : a.xaff_slope * y;
      }
      v = rs.aff ? t : v;
    }
    return rs.ok[i] ? v : z;
  };
  auto lstore = [&](const RegSet& rs, int b) {
    if constexpr (B16) {
      f32x4* Xs = lds + b * STG16;  // 4 x 16-byte slots (64 bytes) per row (split fp32: per plane)
      f32x4* Ws = Xs + BN * 4;
      u32x2* Xh = reinterpret_cast<u32x2*>(Xs);
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int idx = tid + i * NT;
        if constexpr (X3 && X16) {  // pre-split planes: 8 channels of one plane go straight into their slot
          const int pl = idx / (BN * 4), idp = idx % (BN * 4), r = idp >> 2, sl = idp & 3;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          if (BN * 12 % NT == 0 || idx < BN * 12) Xs[pl * PLANE + r * 4 + (sl ^ ((r >> 2) & 3))] = rs.ok[i] ? rs.x[i] : z;
        } else if constexpr (X3) {  // four fp32 channels -> their place in each of the three bf16 planes
          const int r = idx >> 3, sl = idx & 7;
          u32x2 p0, p1, p2;
          split3_bf16(xin(rs, i), p0, p1, p2);
          const int o = (r * 4 + ((sl >> 1) ^ ((r >> 2) & 3))) * 2 + (sl & 1);
          Xh[o] = p0;
          Xh[o + PLANE * 2] = p1;
          Xh[o + PLANE * 4] = p2;
        } else if constexpr (X16) {  // rows arrive rounded: 16 bytes = 8 channels go straight into their slot
          const int r = idx >> 2, sl = idx & 3;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          if (BN * 4 % NT == 0 || idx < BN * 4) Xs[r * 4 + (sl ^ ((r >> 2) & 3))] = rs.ok[i] ? rs.x[i] : z;
        } else {
          const int r = idx >> 3, sl = idx & 7;
          Xh[(r * 4 + ((sl >> 1) ^ ((r >> 2) & 3))) * 2 + (sl & 1)] = pack4_16<PREC>(xin(rs, i));
        }
      }
#pragma unroll
      for (int i = 0; i < WL; ++i) {
        const int idx = tid + i * NT;
        if constexpr (X3) {
          const int pl = idx / (BM * 4), rem = idx % (BM * 4), n = rem >> 2, c = rem & 3;
          if (BM * 12 % NT == 0 || idx < BM * 12) Ws[pl * PLANE + n * 4 + (c ^ ((n >> 2) & 3))] = rs.w[i];
        } else {
        const int n = idx >> 2, c = idx & 3;
        if (BM * 4 % NT == 0 || idx < BM * 4) Ws[n * 4 + (c ^ ((n >> 2) & 3))] = rs.w[i];
        }
      }
    } else {
    f32x4* Xs = lds + b * (BN + BM) * 8;
    f32x4* Ws = Xs + BN * 8;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = tid + i * NT;
      const int r = idx >> 3, sl = idx & 7;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      Xs[r * 8 + (sl ^ ((r >> 1) & 7))] = xin(rs, i);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = tid + i * NT;
      const int n = idx >> 3, sl = idx & 7;
      Ws[n * 8 + (sl ^ ((n >> 1) & 7))] = rs.w[i];
    }
    }
  };

  int total = a.seg[0].ntaps * (a.seg[0].kc / 32);
  if (a.nseg > 1) total += a.seg[1].ntaps * (a.seg[1].kc / 32);
  if (a.nseg > 2) total += a.seg[2].ntaps * (a.seg[2].kc / 32);
  if (ksplit > 1) {  // block-level split-K: this block's slice [it0, it1) of the iterations; position the cursor at it0
    int it0 = (int)((long)total * ks / ksplit);
    const int it1 = (int)((long)total * (ks + 1) / ksplit);
    total = it1 - it0;
    for (int q = 0; q < 2; ++q) {  // skip whole segments (at most two)
      const int n_it = g_ntaps * (g_kc / 32);
      if (it0 >= n_it && s + 1 < nseg) {
        it0 -= n_it;
        ++s;
        const GemmSeg& n = s == 1 ? a.seg[1] : a.seg[2];
        gX = xbase(n);
        gW = n.W + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
        if constexpr (B16) gW16 = n.W16 + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
        g_ldx = n.ldx; g_kc = n.kc; g_ntaps = n.ntaps; g_dil = n.dil; g_pad = n.pad;
        if constexpr (X3) { g_wplane = n.w16_plane; g_xplane = n.x_plane; }
        seg_offsets();
      }
    }
    chunk = it0 / g_ntaps;
    tap = it0 % g_ntaps;
  }

  const int l31 = lane & 31, lh = lane >> 5;
  auto mma_step = [&](const f32x4* Xs, const f32x4* Ws, int kk) {
    const int slot = 2 * kk + lh;
    f32x4 xa[TR], wb[TC];
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int r = wn * WR + i * 32 + l31;
      xa[i] = Xs[r * 8 + (slot ^ ((r >> 1) & 7))];
    }
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = wm * WC + j * 32 + l31;
      wb[j] = Ws[c * 8 + (slot ^ ((c >> 1) & 7))];
    }
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].x, wb[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].y, wb[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].z, wb[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].w, wb[j].w, acc[i][j], 0, 0, 0);
      }
  };

  // 16-bit operands: kk = 0, 1 are the two 16-channel halves of the chunk; lanes 0-31 / 32-63 supply k 0-7 / 8-15
  auto mma_step16 = [&](int b, int kk) {
    const f32x4* Xs = lds + b * STG16;
    const f32x4* Ws = Xs + BN * 4;
    const int slot = 2 * kk + lh;
    if constexpr (X3) {
      f32x4 xa[TR][3], wb[TC][3];
      const bool nord = ablate(32);  // timing-only ablation: operands are not read from LDS
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int r = wn * WR + i * 32 + l31;
        const int o = r * 4 + (slot ^ ((r >> 2) & 3));
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (!nord) xa[i][p] = Xs[o + p * PLANE];
          else xa[i][p] = f32x4{(float)r, (float)kk, (float)p, 0.f};
        }
      }
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int c = wm * WC + j * 32 + l31;
        const int o = c * 4 + (slot ^ ((c >> 2) & 3));
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (!nord) wb[j][p] = Ws[o + p * PLANE];
          else wb[j][p] = f32x4{(float)c, (float)kk, (float)p, 0.f};
        }
      }
      if (ablate(1)) return;  // timing-only ablation: no MFMAs
      // the six products with p + q <= 2, smallest first (the accumulator takes the 2^-16 terms before the 2^-8 and the leading ones)
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          acc[i][j] = mfma16<PREC>(xa[i][2], wb[j][0], acc[i][j]);
          acc[i][j] = mfma16<PREC>(xa[i][0], wb[j][2], acc[i][j]);
          acc[i][j] = mfma16<PREC>(xa[i][1], wb[j][1], acc[i][j]);
          acc[i][j] = mfma16<PREC>(xa[i][1], wb[j][0], acc[i][j]);
          acc[i][j] = mfma16<PREC>(xa[i][0], wb[j][1], acc[i][j]);
          acc[i][j] = mfma16<PREC>(xa[i][0], wb[j][0], acc[i][j]);
        }
      return;
    }
    f32x4 xa[TR], wb[TC];
    const bool nord = ablate(32);  // timing-only ablation: operands are not read from LDS
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int r = wn * WR + i * 32 + l31;
      if (!nord) xa[i] = Xs[r * 4 + (slot ^ ((r >> 2) & 3))];
      else xa[i] = f32x4{(float)r, (float)kk, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = wm * WC + j * 32 + l31;
      if (!nord) wb[j] = Ws[c * 4 + (slot ^ ((c >> 2) & 3))];
      else wb[j] = f32x4{(float)c, (float)kk, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) acc[i][j] = mfma16<PREC>(xa[i], wb[j], acc[i][j]);
  };

  if constexpr (GLDS) {
    // ---- LDS-DMA staging: global_load_lds writes 64 lanes x 16 B = 8 tile rows straight into LDS (no VGPR round trip,
    // no ds_write, no mid-loop waits).  The destination is lane-linear, so the 16-byte-slot swizzle is applied to the
    // SOURCE slot each lane fetches (guide 5.4 rule 21); rows outside the utterance fetch from a page of zeros.
    static_assert(KSPLIT == 1, "GLDS path is single K-group");
    if constexpr (B16) {
      // 16-bit rows (64 bytes per tile row): one wave-instruction = 16 tile rows; three stages, tile it+2 is requested while tile it is
      // multiplied, and the wait before the barrier leaves that newest tile's requests in flight (vmcnt retires in order)
      constexpr int NW = NT / 64, NI1 = (BN + BM) / 16, NI = NI1 * NPL, PER = NI / NW;  // (split fp32: the three planes of the tile, one after the other)
      static_assert((NI % NW == 0 || X3) && BN % 16 == 0, "tile rows must divide over the waves");
      constexpr int PERC = (NI + NW - 1) / NW;  // (split fp32, 36 instructions over 8 waves: the last wave-instructions are issued twice - same bytes to the same place - so every wave counts the same vmcnt)
      const int wv = tid >> 6;
      auto issue = [&](int b) {
        const int shift = (tap - g_pad) * g_dil;
        const char* xb = reinterpret_cast<const char*>(reinterpret_cast<const unsigned short*>(gX) + (long)lo * g_ldx + chunk * 32);
        const char* wb = reinterpret_cast<const char*>(gW16 + tap * g_kc + chunk * 32);
        const int wrow = g_ntaps * g_kc;
#pragma unroll
        for (int q = 0; q < PERC; ++q) {
          const int instq = min(wv + q * NW, NI - 1);  // 16-row group of the combined [X rows | W rows] tile of plane pl
          const int pl = instq / NI1, inst = instq % NI1;
          const int trow = inst * 16 + (lane >> 2);  // row in the combined tile
          const char* src;
          if (inst < BN / 16) {                      // wave-uniform
            const int r = trow;
            const int sslot = (lane & 3) ^ ((r >> 2) & 3);
            const int rel = rel0 + r + shift;
            const bool ok = rel >= 0 && rel < len;
            src = ok ? xb + (long)pl * g_xplane * 2 + (unsigned)((rel * g_ldx + sslot * 8) * 2) : reinterpret_cast<const char*>(a.zeros);
          } else {
            const int n = trow - BN;
            const int sslot = (lane & 3) ^ ((n >> 2) & 3);
            src = wb + (long)pl * g_wplane * 2 + (unsigned)((n * wrow + sslot * 8) * 2);
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(lds + b * STG16 + pl * PLANE + inst * 64), 16, 0, 0);
        }
        const int w1 = tap + 1 >= g_ntaps, wk = (chunk + 1) * 32 >= g_kc;  // (bitwise: see the register path's cursor)
        const bool wrapt1 = w1 != 0;
        const bool wrapt = (w1 & wk) != 0;
        const bool last = ((w1 & wk) & (MSEG ? (int)(s + 1 >= nseg) : 1)) != 0;
        tap = last ? tap : (wrapt1 ? 0 : tap + 1);
        chunk = last ? chunk : (wrapt ? 0 : chunk + w1);
        if constexpr (MSEG) {
          if (wrapt && !last) {
            ++s;
            const GemmSeg& n = s == 1 ? a.seg[1] : a.seg[2];
            gX = xbase(n);
            gW16 = n.W16 + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
            g_ldx = n.ldx; g_kc = n.kc; g_ntaps = n.ntaps; g_dil = n.dil; g_pad = n.pad;
            if constexpr (X3) { g_wplane = n.w16_plane; g_xplane = n.x_plane; }
          }
        }
      };
      // wait until only the NSTAGE - 2 newest tiles' requests of this wave are outstanding: the next tile to multiply has landed
      (void)PER;
      constexpr int INFLIGHT = (NSTAGE - 2) * PERC;
      static_assert(INFLIGHT >= 1 && INFLIGHT <= 63, "vmcnt immediate");
      auto wait_next_tile = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt((INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14));  // vmcnt(INFLIGHT), gfx9 encoding
        asm volatile("" ::: "memory");
      };
#pragma unroll
      for (int q = 0; q < NSTAGE - 1; ++q) issue(q);  // (the cursor re-issues the last tile once it runs off the end: harmless, nobody multiplies it)
      wait_next_tile();
      __syncthreads();
      stamp(1);
      int cur = 0, nxt = NSTAGE - 1;
      for (int it = 0; it < total; ++it) {
        if (!ablate(8)) issue(nxt);  // tile it + NSTAGE - 1 -> the stage tile it - 1 was read from (every wave has passed the barrier since)
        mma_step16(cur, 0);
        mma_step16(cur, 1);
        wait_next_tile();            // tile it + 1 has landed (this wave's share; the barrier covers the others')
        if (!ablate(2)) __syncthreads();
        cur = cur == NSTAGE - 1 ? 0 : cur + 1;
        nxt = nxt == NSTAGE - 1 ? 0 : nxt + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the look-ahead requests before the stages are reused / the block ends
      __syncthreads();
      stamp(2);
    } else {
        static_assert(KSPLIT == 1, "GLDS path is single K-group");
      constexpr int NW = NT / 64, NI = (BN + BM) / 8;  // wave-instructions per tile
      static_assert(NI % NW == 0, "tile rows must divide over the waves");
      const int wv = tid >> 6;
      auto issue = [&](int b) {
        const int shift = (tap - g_pad) * g_dil;
        const char* xb = reinterpret_cast<const char*>(gX + (long)lo * g_ldx + chunk * 32);
        const char* wb = reinterpret_cast<const char*>(gW + tap * g_kc + chunk * 32);
        const int wrow = g_ntaps * g_kc;
  #pragma unroll
        for (int q = 0; q < NI / NW; ++q) {
          const int inst = wv + q * NW;             // 8-row group of the combined [X rows | W rows] tile
          const int trow = inst * 8 + (lane >> 3);  // row in the combined tile
          const char* src;
          if (inst < BN / 8) {                      // wave-uniform
            const int r = trow;
            const int sslot = (lane & 7) ^ ((r >> 1) & 7);
            const int rel = rel0 + r + shift;
            const bool ok = rel >= 0 && rel < len;
            src = ok ? xb + (unsigned)((rel * g_ldx + sslot * 4) * 4) : reinterpret_cast<const char*>(a.zeros);
          } else {
            const int n = trow - BN;
            const int sslot = (lane & 7) ^ ((n >> 1) & 7);
            src = wb + (unsigned)((n * wrow + sslot * 4) * 4);
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(lds + b * (BN + BM) * 8 + inst * 64), 16, 0, 0);
        }
        // advance the cursor (same order as the register path)
        const int w1 = tap + 1 >= g_ntaps, wk = (chunk + 1) * 32 >= g_kc;
        const bool wrapt1 = w1 != 0;
        const bool wrapt = (w1 & wk) != 0;
        const bool last = ((w1 & wk) & (int)(s + 1 >= nseg)) != 0;
        tap = last ? tap : (wrapt1 ? 0 : tap + 1);
        chunk = last ? chunk : (wrapt ? 0 : chunk + w1);
        if (wrapt && !last) {
          ++s;
          const GemmSeg& n = s == 1 ? a.seg[1] : a.seg[2];
          gX = xbase(n);
          gW = n.W + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
          if constexpr (B16) gW16 = n.W16 + (long)utt * n.w_utt_stride + (long)m0 * n.ntaps * n.kc;
          g_ldx = n.ldx; g_kc = n.kc; g_ntaps = n.ntaps; g_dil = n.dil; g_pad = n.pad;
        }
      };
      issue(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int it = 0; it < total; ++it) {
        issue((it + 1) & 1);  // tile it+1 (the cursor re-issues the last tile at the end: harmless, nobody reads it)
        const f32x4* Xs = lds + (it & 1) * (BN + BM) * 8;
        const f32x4* Ws = Xs + BN * 8;
        mma_step(Xs, Ws, 0);
        mma_step(Xs, Ws, 1);
        mma_step(Xs, Ws, 2);
        mma_step(Xs, Ws, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  } else {
  gload(rsA);  // tile 0
  lstore(rsA, 0);
  gload(rsB);  // tile 1
  if constexpr (DEEP) {
    gload(rsC);  // tile 2
    gload(rsD);  // tile 3
  }
  gload(rsA);  // tile 2 (DEEP: 4)  (the cursor clamps at the last tile, extra fetches are harmless re-loads)
  __syncthreads();
  stamp(1);

  // iteration `it` computes tile it from buffer it&1; tile it+1 (fetched during iteration it-2) moves registers -> the
  // other buffer (legal: that buffer was last read in iteration it-1 and a barrier has passed); then tile it+3 is fetched
  // into the freed register set.
  int return_guard = 0;
  (void)return_guard;
  // split fp32 (one K-group): the MFMA operands of a 16-channel half are read from LDS one half AHEAD of the 6 x TR x TC MFMAs that consume
  // them, into two named fragment sets - the reads of (tile it, half 1) fly under the MFMAs of half 0, those of (tile it + 1, half 0) are
  // issued right after the barrier that publishes that tile and fly under the MFMAs of half 1.  With the reads in front of their own MFMAs the
  // two waves of a SIMD, which the barrier keeps in lockstep, both sat out every LDS round trip with the matrix pipe idle.
  // (measured, B = 8: 128 x 128 tile 78.7 -> 75.6 us on the decoder's conv2; the 128 x 64 tile loses its second block per CU to the 30 extra
  //  registers, 82.6 -> 86.2 us, and the 256-row tile spills: only the 8-wave 128 x 128 tile is pipelined this way)
  constexpr bool FPIPE = X3 && KSPLIT == 1 && BN == 128 && BM == 128 && WARPS_M * WARPS_N == 8 && !GLDS;
  struct Frag {
    f32x4 xa[TR][3], wb[TC][3];
  };
  Frag fA, fB;
  auto frag_read = [&](int b, int kk, Frag& f) {
    const f32x4* Xs = lds + b * STG16;
    const f32x4* Ws = Xs + BN * 4;
    const int slot = 2 * kk + lh;
    const bool nord = ablate(32);
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int r = wn * WR + i * 32 + l31;
      const int o = r * 4 + (slot ^ ((r >> 2) & 3));
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (!nord) f.xa[i][p] = Xs[o + p * PLANE];
        else f.xa[i][p] = f32x4{(float)r, (float)kk, (float)p, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = wm * WC + j * 32 + l31;
      const int o = c * 4 + (slot ^ ((c >> 2) & 3));
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (!nord) f.wb[j][p] = Ws[o + p * PLANE];
        else f.wb[j][p] = f32x4{(float)c, (float)kk, (float)p, 0.f};
      }
    }
  };
  auto frag_mma = [&](const Frag& f) {
    if (ablate(1)) return;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        acc[i][j] = mfma16<PREC>(f.xa[i][2], f.wb[j][0], acc[i][j]);
        acc[i][j] = mfma16<PREC>(f.xa[i][0], f.wb[j][2], acc[i][j]);
        acc[i][j] = mfma16<PREC>(f.xa[i][1], f.wb[j][1], acc[i][j]);
        acc[i][j] = mfma16<PREC>(f.xa[i][1], f.wb[j][0], acc[i][j]);
        acc[i][j] = mfma16<PREC>(f.xa[i][0], f.wb[j][1], acc[i][j]);
        acc[i][j] = mfma16<PREC>(f.xa[i][0], f.wb[j][0], acc[i][j]);
      }
  };
  if constexpr (FPIPE) frag_read(0, 0, fA);
  auto iter = [&](int it, RegSet& nset) {
    const f32x4* Xs = lds + (it & 1) * (BN + BM) * 8;
    const f32x4* Ws = Xs + BN * 8;
    if constexpr (FPIPE) {
      frag_read(it & 1, 1, fB);
      frag_mma(fA);
      if (!ablate(4)) lstore(nset, (it + 1) & 1);
      if (!ablate(8)) gload(nset);  // (issued behind the barrier instead, under the second half's MFMAs: 3-4 % slower on every shape, 73.0 -> 76.4 us on the decoder's conv2)
#ifndef STTS_X3_NO_SGB
      // Issue pattern for the half in front of the barrier (6 TR TC MFMAs, the split of the next tile, its LDS writes, the fetch after it): one MFMA, then
      // seven vector instructions, twelve times over - the scheduler's own order left the MFMAs in bunches of three to nine.  Measured (B = 8, per launch):
      // decoder conv2 73.1 -> 72.1 us, output conv 537 -> 519, Winograd planes 81.0 -> 79.9; with LDS / memory slots in the pattern as well: slower (75.5);
      // 5 or 8 instead of 7: no gain.  (The same pattern placed behind the barrier - where the second half has no vector work to spread - cost 4-5 %.)
#pragma unroll
      for (int g = 0; g < 6 * TR * TC; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);  // VALU
      }
#endif
      if (!ablate(2)) __syncthreads();  // (waits for this wave's LDS traffic first: its reads of this tile have landed before anybody overwrites the stage)
      frag_read((it + 1) & 1, 0, fA);  // (after the last tile: a stale stage, read and dropped)
      frag_mma(fB);
      return;
    }
    if constexpr (B16) {
      if constexpr (KSPLIT == 1) {
        mma_step16(it & 1, 0);
        if (!ablate(4)) lstore(nset, (it + 1) & 1);
        if (!ablate(8)) gload(nset);
        mma_step16(it & 1, 1);
#ifndef STTS_X3_NO_SGB
        if constexpr (X3) {
          // split fp32, tiles without the fragment pipeline: one MFMA, then this tile's share of the iteration's vector instructions (~25 of addressing + ~30 per X
          // load of split arithmetic, over 12 TR TC MFMAs) + 1.  128 x 256 tile: decoder conv2 110.3 -> 102.6 us, Winograd planes 71.5 -> 69.4, output conv
          // 470 -> 446; 128 x 64 tile 77.4 -> 76.9; one or two more per group: back to the unpatterned times
          constexpr int NM = 12 * TR * TC, NV = (25 + 30 * XL + NM - 1) / NM + 1;
#pragma unroll
          for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
          }
        }
#endif
      } else {
        mma_step16(it & 1, kg);
        lstore(nset, (it + 1) & 1);
        gload(nset);
      }
    } else if constexpr (KSPLIT == 1) {
      mma_step(Xs, Ws, 0);
      if (!ablate(4)) lstore(nset, (it + 1) & 1);
      mma_step(Xs, Ws, 1);
      if (!ablate(8)) gload(nset);
      mma_step(Xs, Ws, 2);
      mma_step(Xs, Ws, 3);
    } else {
      mma_step(Xs, Ws, 2 * kg);
      lstore(nset, (it + 1) & 1);
      gload(nset);
      mma_step(Xs, Ws, 2 * kg + 1);
    }
    if (!ablate(2)) __syncthreads();
  };
  if constexpr (DEEP) {
    // iteration `it` moves tile it + 1 (requested during iteration it - 4) into LDS and requests tile it + 5 into the freed set
    int it = 0;
    for (; it + 3 < total; it += 4) {
      iter(it, rsB);
      iter(it + 1, rsC);
      iter(it + 2, rsD);
      iter(it + 3, rsA);
    }
    if (it < total) iter(it, rsB);
    if (it + 1 < total) iter(it + 1, rsC);
    if (it + 2 < total) iter(it + 2, rsD);
  } else {
    int it = 0;
    for (; it + 1 < total; it += 2) {  // branch-free body: two iterations, one per register set
      iter(it, rsB);
      iter(it + 1, rsA);
    }
    if (it < total) iter(it, rsB);
  }
  stamp(2);
#ifdef STTS_GEMM_TRACE
  if (a.dbg && tid == 0) a.dbg[8 * (long)dbg_blk + 4] = total;
#endif
  }

  if constexpr (KSPLIT == 2) {
    // sum the two K-groups: group 1 parks its accumulators in LDS (the staging buffers are free now), group 0 adds them
    // and runs the epilogue alone.
    static_assert(WARPS_M * WARPS_N * TR * TC * 16 * 64 * 4 <= NSTAGE * (BN + BM) * SLOTS * 16, "reduction buffer must fit in the staging LDS");
    float* red = reinterpret_cast<float*>(lds);
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((wid * TR + i) * TC + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(((wid * TR + i) * TC + j) * 16 + r) * 64 + lane];
  }
  // ------------------------------------------------------------------ epilogue
  // acc[i][j][r]: row = wn*WR + i*32 + (r&3) + 8*(r>>2) + 4*lh ; col = wm*WC + j*32 + l31
  const int nvalid = hi - row0;  // rows of this tile inside the utterance
  if constexpr (EPI == EPI_STORE) {
    if (ksplit > 1) {
      float* P = a.partial + (long)ks * a.rows_total * a.ld_part;
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int n = m0 + wm * WC + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wn * WR + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (rl < nvalid) P[(long)(row0 + rl) * a.ld_part + n] = acc[i][j][r];
          }
      }
      stamp_end();
      return;
    }
    // Uniform (scalar) base pointers at the tile's first row and column + 32-bit per-lane element offsets: a store is one vector
    // add and a saddr-form instruction, and the epilogue keeps few enough registers live next to the accumulators (the 256 x 256
    // tile spilled 290 of them with per-element 64-bit addresses: its epilogue took 42 us of a 228 us block).
    const bool hasR = a.R != nullptr, hasY = a.Y != nullptr, hasSS = a.sumsq_part != nullptr;
    bool hasY16 = false;
    if constexpr (B16) hasY16 = a.Y16 != nullptr;
    const int act = a.act;
    const float alpha = a.alpha;
    float* const Yb = hasY ? a.Y + (long)row0 * a.ldy + a.ycol0 + m0 : nullptr;
    const float* const Rb = hasR ? a.R + (long)row0 * a.ldr + a.rcol0 + m0 : nullptr;
    unsigned short* const Y16b = hasY16 ? a.Y16 + (long)row0 * a.ldy16 + a.ycol16 + m0 : nullptr;
    const int rlane = wn * WR + 4 * lh, clane = wm * WC + l31;  // + i * 32 + (r & 3) + 8 * (r >> 2) ; + j * 32
    // running element offsets of this lane's current row; consecutive accumulator elements are 1 row apart, or 5 (r = 3 -> 4, 7 -> 8,
    // 11 -> 12, and 15 -> 0 of the next 32-row tile: 27 -> 32), so the whole wave tile is walked with two uniform steps per buffer
    unsigned oy = (unsigned)(rlane * a.ldy + clane), orr = (unsigned)(rlane * a.ldr + clane), o16 = (unsigned)(rlane * a.ldy16 + clane);
    const unsigned sy1 = (unsigned)a.ldy, sy5 = 5u * sy1, sr1 = (unsigned)a.ldr, sr5 = 5u * sr1, s161 = (unsigned)a.ldy16, s165 = 5u * s161;
    float bv[TC];
    bool nok[TC];
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int n = m0 + clane + j * 32;
      bv[j] = a.bias ? a.bias[n] : 0.0f;
      nok[j] = n < a.N;
    }
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      float ss[TC];
#pragma unroll
      for (int j = 0; j < TC; ++j) ss[j] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = i * 32 + (r & 3) + 8 * (r >> 2);  // compile-time row of this element inside the wave tile
        const bool rok = rlane + dr < nvalid;
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          float v = act_apply(acc[i][j][r] + bv[j], act);
          if (rok && nok[j]) {
            if (hasR) v += Rb[orr + j * 32];
            v *= alpha;
            if (hasY) Yb[oy + j * 32] = v;
            if constexpr (B16) {
              if (hasY16) {
                if constexpr (PREC == PREC_BF16) reinterpret_cast<__bf16*>(Y16b)[o16 + j * 32] = (__bf16)v;
                else reinterpret_cast<_Float16*>(Y16b)[o16 + j * 32] = (_Float16)v;
              }
            }
            ss[j] += v * v;
          }
        }
        const bool big = (r & 3) == 3;
        oy += big ? sy5 : sy1;
        orr += big ? sr5 : sr1;
        o16 += big ? s165 : s161;
      }
      if (hasSS) {
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          float t2 = ss[j] + __shfl_xor(ss[j], 32, 64);
          const int sub = by * (BN / 32) + (wn * TR + i);  // 32-row sub-tile of the utterance; the consumer reads ceil(len / 32) of them
          if (lh == 0 && nok[j] && sub * 32 < hi - lo) {
            const long t = (long)utt * a.ss_stride + sub;
            a.sumsq_part[t * a.ld_ss + m0 + clane + j * 32] = t2;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // one 32-row band at a time: keeps the temporaries of 8 tiles from piling up
    }
  } else if constexpr (EPI == EPI_SPLIT_ACC) {
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int n = m0 + wm * WC + j * 32 + l31;
      const float bv = a.bias ? a.bias[n] : 0.0f;
      if (n < a.N) {
        float* D;
        int ld, col, accum;
        if (n < a.nsplit) {
          D = a.D0; ld = a.ldd0; col = n; accum = a.acc0;
        } else {
          D = a.D1; ld = a.ldd1; col = n - a.nsplit; accum = a.acc1;
        }
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wn * WR + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (rl < nvalid) {
              float* p = D + (long)(row0 + rl) * ld + col;
              const float v = acc[i][j][r] + bv;
              *p = accum ? *p + v : v;
            }
          }
      }
    }
  } else {
    // paired: within each 64-column group of the packed weight, tile 2p holds the 'a' rows and tile 2p+1
    // the matching 'b' rows, so both halves of a gate sit in the same lane.
#pragma unroll
    for (int jp = 0; jp < TC / 2; ++jp) {
      const int npk = m0 + wm * WC + jp * 64 + l31;         // packed index of the 'a' row
      const int c = (npk >> 6) * 32 + (npk & 31);           // result channel
      const float ba = a.bias ? a.bias[npk] : 0.0f;
      const float bb = a.bias ? a.bias[npk + 32] : 0.0f;
      if (c < a.N) {
        float ga = 0.f, gb = 0.f;
        if constexpr (EPI == EPI_GATE) {
          ga = a.gate[(long)utt * a.ld_gate + a.gcol0 + c];
          gb = a.gate[(long)utt * a.ld_gate + a.gcol0 + a.gC + c];
        }
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wn * WR + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (rl < nvalid) {
              const long grow = row0 + rl;
              const float va = acc[i][2 * jp][r] + ba, vb = acc[i][2 * jp + 1][r] + bb;
              if constexpr (EPI == EPI_GATE) {
                const float t = tanhf(va + ga);
                const float sg = 1.0f / (1.0f + __expf(-(vb + gb)));
                a.Y[grow * a.ldy + a.ycol0 + c] = t * sg;
              } else if constexpr (EPI == EPI_COUPLE) {
                float* p = a.Z + grow * a.ldz + a.zcol0 + c;
                *p = (*p - va) * __expf(-vb);
              } else {  // EPI_PRIOR
                a.Z[grow * a.ldz + a.zcol0 + c] = va + a.noise[grow * a.ldnoise + c] * __expf(vb);
              }
            }
          }
      }
    }
  }
  stamp_end();
}

// Finishes a block-level split-K contraction: Y = (act(sum_ks partial[ks] + bias) [+ R]) * alpha, partials summed in a
// fixed order (deterministic).  One thread per (row, 4 columns).
static __global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, int rows, int row_first, int ld_part, int N,
                                                            const float* __restrict__ bias, int act, const float* __restrict__ R, int ldr,
                                                            int rcol0, float alpha, float* __restrict__ Y, int ldy, int ycol0) {
  const int n4 = (N + 3) / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)(rows - row_first) * n4; i += (long)gridDim.x * 256) {
    const int row = row_first + (int)(i / n4), n = (int)(i % n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(partial + (long)row * ld_part + n);
    for (int k = 1; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(partial + ((long)k * rows + row) * ld_part + n);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (n + c < N) {
        float x = act_apply(v[c] + (bias ? bias[n + c] : 0.0f), act);
        if (R) x += R[(long)row * ldr + rcol0 + n + c];
        Y[(long)row * ldy + ycol0 + n + c] = x * alpha;
      }
    }
  }
}

// grow-only scratch for split-K partial sums, one buffer per (device, launch stream): contractions issued on different
// streams may run concurrently and must not share partials, and the null stream exists on every device.
// Growing allocates a NEW buffer and retires the old one (freed later, never while kernels that were given it may still be
// queued), so a larger batch after small ones costs one hipMalloc, not a device-wide synchronisation.  Retired buffers are kept
// PER DEVICE and reclaimed only from a call on that device, after synchronising it.
// Stream capture: growth (hipMalloc, possibly hipDeviceSynchronize) is illegal inside a capture; callers that capture a stage into
// a HIP graph run it once eagerly first (cfm_decoder.py does), which sizes this scratch, and shapes must not grow afterwards.
inline float* splitk_scratch(hipStream_t st, size_t bytes) {
  struct Buf {
    float* p = nullptr;
    size_t cap = 0;
  };
  static std::map<std::pair<int, hipStream_t>, Buf> bufs;
  static std::map<int, std::vector<void*>> retired;  // per device
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  Buf& b = bufs[{dev, st}];
  if (bytes > b.cap) {
    std::vector<void*>& old = retired[dev];
    if (b.p) old.push_back(b.p);
    if (old.size() > 8) {  // bounded: reclaim this device's retired buffers once everything queued on it so far has drained
      (void)hipDeviceSynchronize();
      for (void* q : old) (void)hipFree(q);
      old.clear();
    }
    b.cap = std::max(bytes + bytes / 2, (size_t)(8u << 20));
    if (hipMalloc(&b.p, b.cap) != hipSuccess) {
      b.p = nullptr;
      b.cap = 0;
    }
  }
  return b.p;
}

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
// Optional per-launch timing of the contraction kernel with HIP events on the launch stream (bench.py's
// roofline leg).  Off by default: when off the launcher records nothing.
// One record per timed launch: kind 0 = a dense contraction (conv_gemm_f32, the fused WaveNet kernels, a whole Winograd-form
// conv), kind 1 = everything else (bandwidth- / latency-bound kernels).  flops = ALGORITHMIC work (direct-conv flops,
// un-padded), exec_flops = what the matrix cores execute (smaller for the Winograd forms), bytes = algorithmic HBM bytes.
struct ProfRec {
  const char* name;
  int kind;
  double flops, exec_flops, bytes;
};
struct GemmProfiler {
  bool on = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<ProfRec> recs;   // recs[i] belongs to the pair (ev[2 i], ev[2 i + 1])
  size_t used = 0;
  void begin() {
    on = true;
    used = 0;
    recs.clear();
  }
  hipEvent_t next() {
    if (used == ev.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      ev.push_back(e);
    }
    return ev[used++];
  }
  void add(const char* name, int kind, double flops, double exec_flops, double bytes) { recs.push_back(ProfRec{name, kind, flops, exec_flops, bytes}); }
};
inline GemmProfiler& gemm_profiler() {
  static GemmProfiler p;
  return p;
}
// 256 zero bytes on the CURRENT device (out-of-range operand fetches of the contraction kernels read them): one page per
// device, created on first use, also under concurrent first calls from several host threads.
inline const float* zero_page() {
  constexpr int kMaxDev = 64;
  static float* pages[kMaxDev] = {};
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDev) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!pages[dev]) {
    float* p = nullptr;
    (void)hipMalloc(&p, 256);
    (void)hipMemset(p, 0, 256);
    (void)hipDeviceSynchronize();
    pages[dev] = p;
  }
  return pages[dev];
}
inline double gemm_algorithmic_flops(const GemmArgs& a) {
  double k = 0;
  for (int i = 0; i < a.nseg; ++i) k += (double)a.seg[i].kreal * a.seg[i].ntaps;
  return 2.0 * (double)a.rows_total * (double)a.wrows * k;
}

// With a profiler event pair the kernel is launched through hipExtLaunchKernelGGL, whose events take the dispatch's own
// begin / end timestamps (what rocprofv3 reports), instead of bracketing the launch with stream markers.
#define STTS_LAUNCH_TIMED(kernel, grid, block, st, e0, e1, ...)                                   \
  do {                                                                                            \
    if (e0) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, __VA_ARGS__);            \
    else hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                             \
  } while (0)

// A bandwidth- / latency-bound kernel: timed like the contractions while the profiler is on (bench.py's `hbm_kernels` list),
// a plain launch otherwise.  bytes = algorithmic HBM bytes of this launch.
#define STTS_LAUNCH_PROF(name, bytes, kernel, grid, block, st, ...)                                \
  do {                                                                                            \
    stts::GemmProfiler& _p = stts::gemm_profiler();                                               \
    if (_p.on) {                                                                                  \
      hipEvent_t _e0 = _p.next(), _e1 = _p.next();                                                \
      _p.add(name, 1, 0.0, 0.0, (double)(bytes));                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, st, _e0, _e1, 0, __VA_ARGS__);                \
    } else {                                                                                      \
      hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                \
    }                                                                                             \
  } while (0)

// conv_gemm16_kernel (gemm16.hip.h): the 16-bit-row store contractions of large batches
inline bool gemm16_eligible(const GemmArgs& a, int epi, int npad);
inline long gemm16_tiles(const GemmArgs& a, int npad, int n_utt);
inline bool gemm16_will_run(const GemmArgs& a, int epi, int npad, int n_utt);  // would launch_conv_gemm pick conv_gemm16_kernel for this call?
template <int ABL>
inline int launch_conv_gemm16(hipStream_t st, const GemmArgs& a, int npad, int n_utt);

// The contraction kernels are instantiated in one translation unit per operand form (csrc/gemm_tu_*.hip define STTS_GEMM_TU_FORM and
// include this header; __graft_entry__.compile builds them in parallel); every other unit sees these declarations only.
void gemm_dispatch_f32(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1);
void gemm_dispatch_bf16(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1);
void gemm_dispatch_f16(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1);
void gemm_dispatch_x3(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1);
int launch_conv_gemm16_main(hipStream_t st, const GemmArgs& a, int npad, int n_utt);  // = launch_conv_gemm16<0> (gemm_tu_g16.hip)

#ifndef STTS_GEMM_NO_LAUNCHER  // (probes that only need the types and conv_gemm16_kernel skip the ~50 instantiations below)
template <int BM, int BN, int WM, int WN, int KS = 1, bool GL = false, int PR = PREC_F32>
inline void launch_cfg(hipStream_t st, const GemmArgs& a, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  dim3 grid(npad / BM, ceil_div(max_rows, BN), n_utt * (a.ksplit > 1 ? a.ksplit : 1)), block(WM * WN * 64 * KS);
  if (a.compact) grid = dim3(npad / BM, a.tiles_y, a.ksplit > 1 ? a.ksplit : 1);
  if constexpr (GL && PR != PREC_F32) {  // 16-bit LDS-DMA tiles: 16-bit activation rows + store epilogue only (the dispatcher guarantees it)
    if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, true, PR, false, false, true>), grid, block, st, e0, e1, a);
    else STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, true, PR, false, true, true>), grid, block, st, e0, e1, a);
    return;
  } else
  switch (epi) {
    case EPI_STORE:
      if constexpr (!GL && PR != PREC_F32 && KS == 1) {
        if (a.x16) {  // 16-bit activation rows (written by the producing kernels): no conversion, half the staging bytes
          if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, GL, PR, false, false, true>), grid, block, st, e0, e1, a);
          else STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, GL, PR, false, true, true>), grid, block, st, e0, e1, a);
          break;
        }
      }
      if constexpr (!GL) {
        if (a.xaff) {
          if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, GL, PR, 1, false>), grid, block, st, e0, e1, a);
          else STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, GL, PR, 1, true>), grid, block, st, e0, e1, a);
          break;
        }
        if (a.nseg == 1) {
          STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, GL, PR, false, false>), grid, block, st, e0, e1, a);
          break;
        }
      }
      STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, GL, PR>), grid, block, st, e0, e1, a);
      break;
    case EPI_SPLIT_ACC: STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_SPLIT_ACC, KS, GL, PR>), grid, block, st, e0, e1, a); break;
    default:
      if constexpr (BM / WM >= 64) {
        if (epi == EPI_GATE) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_GATE, KS, GL, PR>), grid, block, st, e0, e1, a);
        else if (epi == EPI_COUPLE) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_COUPLE, KS, GL, PR>), grid, block, st, e0, e1, a);
        else STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_PRIOR, KS, GL, PR>), grid, block, st, e0, e1, a);
      }
      break;
  }
}

// split fp32 (PREC_X3): store and prior epilogues only; the flow's generic gate / couple / split-accumulate launches stay on the f32 matrix cores
template <int BM, int BN, int WM, int WN, int KS = 1, int X16MODE = 0>  // X16MODE 1: pre-split activation planes, register staging; 2: LDS-DMA staging
inline void launch_cfg_x3(hipStream_t st, const GemmArgs& a, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  dim3 grid(npad / BM, ceil_div(max_rows, BN), n_utt * (a.ksplit > 1 ? a.ksplit : 1)), block(WM * WN * 64 * KS);
  if (a.compact) grid = dim3(npad / BM, a.tiles_y, a.ksplit > 1 ? a.ksplit : 1);
  if constexpr (X16MODE != 0) {
    if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, X16MODE == 2, PREC_X3, false, false, true>), grid, block, st, e0, e1, a);
    else STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, X16MODE == 2, PREC_X3, false, true, true>), grid, block, st, e0, e1, a);
    return;
  } else
  if (epi == EPI_PRIOR) {
    if constexpr (BM / WM >= 64) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_PRIOR, KS, false, PREC_X3>), grid, block, st, e0, e1, a);
  } else if (a.xaff) {
    if (a.nseg == 1 && a.xaff_scale_only) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, false, PREC_X3, 2, false>), grid, block, st, e0, e1, a);
    else if (a.nseg == 1) STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, false, PREC_X3, 1, false>), grid, block, st, e0, e1, a);
    else STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, false, PREC_X3, 1, true>), grid, block, st, e0, e1, a);
  } else if (a.nseg == 1) {
    STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, false, PREC_X3, false, false>), grid, block, st, e0, e1, a);
  } else {
    STTS_LAUNCH_TIMED((conv_gemm_f32<BM, BN, WM, WN, EPI_STORE, KS, false, PREC_X3>), grid, block, st, e0, e1, a);
  }
}
// tile -> instantiation, per operand form.  Defined (= every kernel instantiated) only in the translation unit of that form.
template <int PR>
inline void gemm_dispatch_tile(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1) {
  switch (tile) {
    case 2: launch_cfg<128, 64, 2, 2, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
    case 4: launch_cfg<128, 32, 4, 1, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;   // 32-row tile, 4 waves of one 32x32 tile each
    case 5: launch_cfg<128, 128, 4, 2, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;  // 8 waves per block
    case 6: launch_cfg<128, 64, 4, 2, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;   // 8 waves, 64-row tiles
    case 8: launch_cfg<128, 128, 4, 2, 2, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;  // 16 waves: 8 positions x 2 K-groups
    case 14:  // 256 cout x 256 rows, 8 waves of 64 x 128 (8 accumulator tiles): 16-bit operands at large batches, where the
              // 128x128 loop is bound by L2 -> LDS staging (47 B/clk/CU needed); this tile needs 31
      if constexpr (PR != PREC_F32) launch_cfg<256, 256, 4, 2, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1);
      break;
    case 15:
      if constexpr (PR != PREC_F32) launch_cfg<128, 256, 4, 2, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1);
      break;
    case 16:  // tiles 14 / 15 with LDS-DMA staging (global_load_lds_dwordx4, three stages): 16-bit activation rows only
      if constexpr (PR != PREC_F32) launch_cfg<256, 256, 4, 2, 1, true, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1);
      break;
    case 17:
      if constexpr (PR != PREC_F32) launch_cfg<128, 256, 4, 2, 1, true, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1);
      break;
    case 18:  // 128 x 128, 8 waves, LDS-DMA with eight stages: one-round launches of small batches in the 16-bit modes
      if constexpr (PR != PREC_F32) launch_cfg<128, 128, 4, 2, 1, true, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1);
      break;
    case 11:
      if constexpr (PR == PREC_F32) launch_cfg<128, 128, 4, 2, 1, true>(st, as, epi, npad, n_utt, max_rows, e0, e1);  // LDS-DMA staging, 8 waves
      break;
    case 13:
      if constexpr (PR == PREC_F32) launch_cfg<128, 64, 4, 2, 1, true>(st, as, epi, npad, n_utt, max_rows, e0, e1);  // LDS-DMA staging, 64-row tile
      break;
    default: launch_cfg<128, 32, 2, 1, 1, false, PR>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
  }
}
inline void gemm_dispatch_tile_x3(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1) {
  switch (tile) {
    case 2: launch_cfg_x3<128, 64, 2, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
    case 4: launch_cfg_x3<128, 32, 4, 1>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
    case 5: launch_cfg_x3<128, 128, 4, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
    case 6: launch_cfg_x3<128, 64, 4, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
    case 8: launch_cfg_x3<128, 128, 4, 2, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
    case 20: launch_cfg_x3<128, 128, 2, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;     // 4 waves of 64 x 64
    case 21: launch_cfg_x3<128, 128, 2, 2, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;  // 8 waves: 64 x 64 x two K-groups
    case 22: launch_cfg_x3<128, 256, 2, 4>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;     // 8 waves of 64 x 64, 256 rows
    // pre-split activation planes (three bf16 planes written by the producer: no split, no conversion in the loop)
    case 25: launch_cfg_x3<128, 128, 4, 2, 1, 1>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;  // tile 5, register staging
    case 26: launch_cfg_x3<128, 64, 4, 2, 1, 1>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;   // tile 6
    case 27: launch_cfg_x3<128, 128, 4, 2, 1, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;  // tile 5, LDS-DMA (three stages)
    case 28: launch_cfg_x3<128, 64, 4, 2, 1, 2>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;   // tile 6, LDS-DMA
    default: launch_cfg_x3<128, 32, 2, 1>(st, as, epi, npad, n_utt, max_rows, e0, e1); break;
  }
}
// Process-wide switch of the split-fp32 contractions (STTS_NO_X3=1: every fp32 contraction on v_mfma_f32_32x32x2_f32, the path of rounds 1-3)
inline bool x3_enabled() {
  static const bool on = !(getenv("STTS_NO_X3") && atoi(getenv("STTS_NO_X3")) != 0);
  return on;
}

// npad: padded cout of the packed weight (multiple of 128).  max_rows: longest utterance (rows).
// Tile choice: 128x128 when that already fills the chip, else smaller row tiles for more workgroups.
inline int launch_conv_gemm(hipStream_t st, const GemmArgs& a, int epi, int npad, int n_utt, int max_rows, int force_tile = 0) {
  STTS_CHECK(npad % 128 == 0, "conv_gemm: padded cout %d not a multiple of 128", npad);
  for (int i = 0; i < a.nseg; ++i) {
    STTS_CHECK(a.seg[i].kc % 32 == 0 && a.seg[i].ldx % 4 == 0 && a.seg[i].xcol0 % 4 == 0, "conv_gemm: segment %d misaligned (kc %d ldx %d xcol0 %d)", i,
               a.seg[i].kc, a.seg[i].ldx, a.seg[i].xcol0);
  }
  // (a segment may read its last padded channels from the NEXT row - a column slice whose width is not a multiple of 32, e.g. the
  //  halves of a 96-channel flow: the packed weights are zero there and the data finite; conv_gemm16_kernel's descriptors do not allow it)
  if (a.x16 && a.prec != PREC_F32)
    for (int i = 0; i < a.nseg; ++i)
      STTS_CHECK(a.seg[i].ldx % 8 == 0 && a.seg[i].xcol0 % 8 == 0, "conv_gemm: 16-bit activation rows need ldx / xcol0 multiples of 8 (segment %d)", i);
  // 16-bit activation rows, store epilogue, at least ~one 256 x 256 tile per CU: the persistent LDS-DMA kernel (gemm16.hip.h).
  // force_tile 19 selects it whatever the size (tests), any other forced tile keeps the launch on conv_gemm_f32.
  if (force_tile == 19 || (force_tile == 0 && gemm16_will_run(a, epi, npad, n_utt))) {
    if (gemm16_eligible(a, epi, npad)) return launch_conv_gemm16_main(st, a, npad, n_utt);
  }
  STTS_CHECK(!a.stat_part, "conv_gemm: output statistics (stat_part) exist only in conv_gemm16_kernel's epilogue: ask gemm16_will_run first");
  STTS_CHECK(force_tile != 19, "conv_gemm: tile 19 (conv_gemm16_kernel) needs 16-bit activation rows, a store epilogue, channels in multiples of 64 and cout padded to 256");
  constexpr int kCUs = 256;
  const int mt = npad / 128;
  auto row_tiles = [&](int bn) -> long {  // exact when the host offsets are known (mixed lengths)
    if (!a.seg_host) return (long)n_utt * ceil_div(max_rows, bn);
    long t = 0;
    for (int u = 0; u < n_utt; ++u) t += ceil_div(a.seg_host[u + 1] - a.seg_host[u], bn);
    return t;
  };
  int iters = 0;
  for (int i = 0; i < a.nseg; ++i) iters += a.seg[i].ntaps * (a.seg[i].kc / 32);
  // (the split-K reduce pass writes fp32 Y only: launches that want the 16-bit copy, or no fp32 output at all, stay whole)
  const bool splittable = force_tile == 0 && epi == EPI_STORE && !a.sumsq_part && a.Y && !a.Y16;
  // A launch takes about ceil(blocks / 256 CUs) block-times however many blocks are co-resident: a CU's matrix pipes are
  // the shared resource (block-timeline trace, profiles/).  When the last round would be mostly empty (288 tiles = 1.125
  // rounds for a 3.5 s batch of 8), the whole rounds run as they are and the REMAINDER row tiles run as a second launch
  // with K cut over up to 8 blocks (+ reduce pass over those rows only): 1 + ~1/8 rounds instead of 2.
  bool x3 = false;  // split fp32 (decided below, before any plan is made)
  struct Plan {
    long full_rt = 0, rem_rt = 0;  // row tiles in the plain launch / in the split-K remainder launch
    int rem_ksp = 1;
    double cost = 0;               // in 128-row block-times
  };
  auto plan_for = [&](int bn, double penalty) {
    Plan p;
    const long rt = row_tiles(bn), blocks = rt * mt;
    const long whole = blocks / kCUs;
    p.full_rt = rt;
    p.cost = std::ceil((double)blocks / kCUs);
    // (split fp32: whole launches.  Its blocks are 1.5 x shorter, and a remainder launch + its reduce pass then cost more than the partly empty last round:
    //  cfg2 3.58 -> 3.475 ms per step without the 12 remainder launches and 11 reduce passes of the Winograd plane contractions; STTS_X3_REM=1 brings them back)
    static const bool x3_rem = getenv("STTS_X3_REM") && atoi(getenv("STTS_X3_REM")) != 0;
    if ((!x3 || x3_rem) && splittable && a.seg_host && !a.capacity && whole >= 1 && blocks % kCUs != 0) {  // (the remainder launch needs exact host offsets)
      const long full_rt = whole * kCUs / mt, rem_blocks = (rt - full_rt) * mt;
      const int ksp = (int)std::min<long>(8, std::min<long>(iters / 4, kCUs / std::max<long>(rem_blocks, 1)));
      if (ksp >= 2 && full_rt > 0) {
        const double hybrid = (double)(full_rt * mt) / kCUs + std::max((double)rem_blocks / kCUs, 1.0 / ksp) * 1.15 + 0.1;
        if (hybrid < p.cost) {
          p.full_rt = full_rt;
          p.rem_rt = rt - full_rt;
          p.rem_ksp = ksp;
          p.cost = hybrid;
        }
      }
    }
    p.cost *= bn * penalty;
    return p;
  };
  const long blocks128 = mt * row_tiles(128);
  // split fp32: fp32 call, every segment carries the three bf16 planes of its weight, epilogue with a split instantiation
  x3 = a.prec == PREC_F32 && x3_enabled() && (epi == EPI_STORE || epi == EPI_PRIOR);
  for (int i = 0; i < a.nseg; ++i) x3 = x3 && a.seg[i].W16 != nullptr && a.seg[i].w16_plane > 0 && 6 * a.seg[i].w16_plane + 2L * 128 * a.seg[i].ntaps * a.seg[i].kc < (1L << 32);
  // pre-split activation planes (x16 on an fp32 call): store epilogue, no input affine, no block split-K (the callers know: run_winograd)
  if (a.x16 && a.prec == PREC_F32) {
    STTS_CHECK(x3 && epi == EPI_STORE && !a.xaff && !a.sumsq_part, "conv_gemm: pre-split activation planes need the split-fp32 store contraction");
    for (int i = 0; i < a.nseg; ++i) STTS_CHECK(a.seg[i].x_plane > 0 && a.seg[i].ldx % 8 == 0 && a.seg[i].xcol0 % 8 == 0, "conv_gemm: pre-split activation planes: segment %d misaligned", i);
  }
  if (force_tile >= 100) {  // tests / tools: 100 + t = tile t on the f32 matrix cores whatever the switch says
    x3 = false;
    force_tile -= 100;
  }
  if (force_tile != 0 && !((force_tile >= 2 && force_tile <= 6) || force_tile == 8 || (force_tile >= 20 && force_tile <= 22) || (a.x16 && force_tile >= 25 && force_tile <= 28))) x3 = false;  // a forced tile without a split form (LDS-DMA tiles)
  int tile = force_tile;
  const bool paired = epi != EPI_STORE && epi != EPI_SPLIT_ACC;  // paired epilogues need 64-column wave tiles
  Plan plan;
  if (tile == 0) {
    // small launches: 32-row tiles; unpaired epilogues spread the 128 output channels over four waves (a wave's MFMA
    // chain per iteration is then 16 instead of 32 instructions: these launches are latency-bound on that chain)
    if (blocks128 < 24) tile = paired ? 3 : 4;
    else if (paired) tile = 2;
    else {
      // 128x128 vs 128x64 tiles by that cost (576 blocks of 128x64 cost three half-sized rounds)
      const Plan p5 = plan_for(128, 1.0), p6 = plan_for(64, 1.03);
      tile = p5.cost <= p6.cost ? 5 : 6;
      // split fp32, short K (at most 24 iterations: the 1 x 1 convs over 512-768 channels): the prologue and epilogue of a block are a fifth of its life, and
      // two co-resident 128 x 64 blocks hide them behind each other's K loop (B = 8, per launch inside the step: pwconv1 94.2 -> 87.3 us, the small
      // 1 x 1 convs 19.0 -> 16.9 / 18.7 -> 15.7; deep K keeps the 128 x 128 tile: pwconv2 85.7 vs 87.4)
      if (x3 && iters <= 24 && !a.xaff) tile = 6;
      plan = tile == 5 ? p5 : p6;
      // one 128x128 tile per CU (B = 8: every 512-channel layer): two K-groups of 8 waves share each staged tile, which
      // keeps the matrix pipes busier than 8 waves do (118 vs 125.5 us) and beats cutting K over two blocks plus the
      // reduce pass (131 us)
      // (fp32 only: with 16-bit operands the loop is staging-bound and 8 waves are faster, 34 vs 40 us)
      // (split fp32: the 8-wave tile is the faster one there too, and the 16-wave tile's 128-register budget spills with the input affine)
      if (tile == 5 && blocks128 <= kCUs && a.prec == PREC_F32 && !x3) tile = 8;
    }
  }
  if (force_tile == 0 && a.prec != PREC_F32 && a.x16 && epi == EPI_STORE) {
    // 16-bit activation rows: 256-row tiles once they fill the chip at least ~1.5 times.  128 x 256 (two blocks per CU, so
    // one block's prologue / epilogue hides behind the other's K loop) unless K is deep (the k = 7 convs: >= 128 iterations),
    // where the 256 x 256 tile's lower staging rate wins (B = 64: out conv 875 vs 896 us, prior conv 287 vs 302; but
    // pwconv1 403 vs 213, decoder convs 200 vs 137: tools/gemm_bench.py TUNE=1152)
    const long rt256 = row_tiles(256);
    if (npad % 256 == 0 && iters >= 128 && rt256 * (npad / 256) >= 3 * kCUs / 2) tile = 14;
    else if (rt256 * (npad / 128) >= 3 * kCUs / 2) tile = 15;
    if (tile == 14 || tile == 15) plan = Plan();
    if (const char* e = getenv("STTS_TILE16")) {  // experiment switch: tile for every 16-bit-row store launch
      tile = atoi(e);
      plan = Plan();
    }
  }
  STTS_CHECK(!(a.x16 && (tile == 8 || tile == 11 || tile == 13 || a.xaff)), "conv_gemm: 16-bit activation rows need a plain register-staged tile");
  STTS_CHECK(!((tile == 14 || tile == 15) && (a.prec == PREC_F32 || epi != EPI_STORE)), "conv_gemm: tiles 14 / 15 are for 16-bit operand store launches");
  STTS_CHECK(!((tile >= 16 && tile <= 18) && (a.prec == PREC_F32 || epi != EPI_STORE || !a.x16)), "conv_gemm: tiles 16 - 18 are for 16-bit activation rows, store epilogue");
  STTS_CHECK((tile != 14 && tile != 16) || npad % 256 == 0, "conv_gemm: tiles 14 / 16 need cout padded to 256");
  if (x3 && !a.x16 && force_tile == 0 && (tile == 5 || tile == 6)) {
    static const int x3_tile = getenv("STTS_X3_TILE") ? atoi(getenv("STTS_X3_TILE")) : 0;  // experiments: 5 / 6 / 22 for every large split-fp32 launch
    if (x3_tile == 5 || x3_tile == 6 || x3_tile == 22) {
      tile = x3_tile;
      plan = Plan();
    }
  }
  // (... or a launch of at least 440 such blocks that fills its last chip round to 80 %: the output convs' Winograd planes at B = 8 are 480 blocks = 1.9
  //  rounds, 212.6 -> 193.0 and 206.5 -> 184.9 us per conv; 360 blocks = 1.4 rounds lose, pwconv1 94 -> 99)
  const long blocks22 = row_tiles(256) * mt;
  const bool fills22 = blocks22 >= 640 || (blocks22 >= 440 && (blocks22 % kCUs == 0 || blocks22 % kCUs >= kCUs * 4 / 5));
  if (x3 && !a.x16 && force_tile == 0 && (tile == 5 || tile == 6) && fills22 && !getenv("STTS_X3_TILE")) {
    // split fp32, launches of at least 2.5 chip rounds of 256-row tiles: 8 waves of 64 x 64 (half the weight staging per row, 12 instead of 18 fragment
    // reads per 24 MFMAs).  B = 64 x 3 s: every layer 8-12 % faster than the 128 x 128 tile (decoder conv2 648 -> 595 us, output conv 3 637 -> 3 342);
    // B = 24: the 1536- and 1024-wide layers (1 080 / 720 blocks) gain, the 512-wide ones (360 blocks = 1.4 rounds) would lose and keep the 128-row tile
    tile = 22;
    plan = Plan();
  }
  STTS_CHECK(!(tile >= 20 && tile <= 28) || x3, "conv_gemm: tiles 20 - 28 exist for the split-fp32 contractions only");
  if (x3 && a.x16) {  // pre-split activation planes: the 128 x 128 or the 128 x 64 tile, whole launches (no block split-K, no remainder launch)
    static const int gl_env = getenv("STTS_X3P_GLDS") ? atoi(getenv("STTS_X3P_GLDS")) : 1;  // experiments: 0 = register staging, 1 = LDS-DMA
    if (tile < 25) tile = (tile == 5 || tile == 8 || tile == 20 || tile == 21 || tile == 22) ? (gl_env ? 27 : 25) : (gl_env ? 28 : 26);
    plan = Plan();
  }
  STTS_CHECK(!(tile >= 25 && tile <= 28) || (x3 && a.x16), "conv_gemm: tiles 25 - 28 read pre-split activation planes");
  const int bn = ((tile >= 14 && tile <= 17) || tile == 22) ? 256 : (tile == 5 || tile == 8 || tile == 11 || tile == 18 || tile == 20 || tile == 21 || tile == 25 || tile == 27) ? 128 : ((tile == 3 || tile == 4) ? 32 : 64);
  if (plan.full_rt == 0 && plan.rem_rt == 0) plan.full_rt = row_tiles(bn);
  if (tile == 8 || tile == 21) {
    plan.full_rt = row_tiles(bn);
    plan.rem_rt = 0;
  }
  // (intra-block K-split, tiles 8-10, and 2-wave tiles measured no better than these at any layer shape: every
  //  configuration plateaus at ~80 % matrix-pipe occupancy)
  GemmArgs as = a;
  as.n_utt = n_utt;
  if (!as.zeros) as.zeros = zero_page();
  as.compact = a.seg_host != nullptr;  // grid.y = the row tiles that exist; needed for tile ranges and for balanced XCDs
  // Block-level split-K for launches that cannot fill the chip (phoneme-rate layers, B = 1): one wave's MFMA chain over
  // the whole K (~1 us per 32 channels x taps) is then the critical path, so K is cut over up to 8 blocks per tile.
  int main_ksp = 1;
  if (splittable && tile != 8 && tile != 21 && tile < 25 && plan.rem_rt == 0) {
    const long blocks = plan.full_rt * mt;
    // (16-bit operands: a contraction that already has one tile per CU is shorter than the reduce pass it would add)
    // K iterations a slice must keep: 4; 8 once the launch has half a chip of blocks anyway (a 16-iteration contraction over 128-256 blocks cut in
    // two gained less than its reduce pass costs: CFM estimator 8 x 800 frames 7.00 -> 6.82 ms; launches with fewer blocks still gain from the cut)
    static const int min_it_env = getenv("STTS_SPLITK_MIN_ITERS") ? std::max(1, atoi(getenv("STTS_SPLITK_MIN_ITERS"))) : 0;  // experiments
    const int min_it = min_it_env ? min_it_env : (blocks >= 128 ? 8 : 4);
    // (fp32 on the f32 matrix cores: launches of 128-256 blocks take the 16-wave tile above, so 512 never cuts them; split fp32: like the 16-bit forms)
    main_ksp = (int)std::min<long>(8, std::min<long>(iters / min_it, ((a.prec == PREC_F32 && !x3) ? 512 : 255) / std::max<long>(blocks, 1)));
    if (main_ksp < 2) main_ksp = 1;
  }
  float* part = nullptr;
  if (main_ksp > 1 || plan.rem_rt > 0) {
    part = splitk_scratch(st, (size_t)std::max(main_ksp, plan.rem_ksp) * a.rows_total * npad * sizeof(float));
    if (!part) {  // no scratch: one plain launch
      main_ksp = 1;
      plan.full_rt += plan.rem_rt;
      plan.rem_rt = 0;
    }
  }
  if (a.prec != PREC_F32) {
    for (int i = 0; i < a.nseg; ++i) STTS_CHECK(a.seg[i].W16 != nullptr, "conv_gemm: 16-bit operand mode without 16-bit weights (segment %d)", i);
    STTS_CHECK(tile != 11 && tile != 13, "conv_gemm: LDS-DMA tiles are fp32 only");
  }
  GemmProfiler& prof = gemm_profiler();
  const double flops = gemm_algorithmic_flops(a);
  const long all_rt = plan.full_rt + plan.rem_rt;
  auto launch_range = [&](long tile0, long ntiles, int ksp) {
    as.tile0 = (int)tile0;
    as.tiles_y = (int)ntiles;
    as.ksplit = ksp;
    as.partial = ksp > 1 ? part : nullptr;
    as.ld_part = npad;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof.on) {
      e0 = prof.next();
      e1 = prof.next();
      // (bench.py's roofline: "_x3" = the split-fp32 form, six bf16 MFMAs per 16 channels on the bf16 matrix cores; no suffix = the f32 matrix cores)
      prof.add(x3 ? "conv_gemm_f32_x3" : "conv_gemm_f32", 0, flops * (double)ntiles / (double)all_rt, flops * (double)ntiles / (double)all_rt, 0.0);
    }
    // the kernels live in one translation unit per operand form (csrc/gemm_tu_*.hip, compiled in parallel)
    if (x3) gemm_dispatch_x3(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
    else if (a.prec == PREC_BF16) gemm_dispatch_bf16(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
    else if (a.prec == PREC_F16) gemm_dispatch_f16(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
    else gemm_dispatch_f32(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
    if (ksp > 1 && a.defer && tile0 == 0 && ntiles == all_rt) {
      *a.defer = SplitInfo{part, ksp, a.rows_total, npad};
    } else if (ksp > 1) {
      // first row of global row tile `tile0` (row tiles are numbered utterance by utterance, i.e. in row order)
      int row_first = 0;
      if (tile0 > 0) {
        long t = tile0;
        for (int u = 0; u < n_utt; ++u) {
          const long tu = ceil_div(a.seg_host[u + 1] - a.seg_host[u], bn);
          if (t < tu) {
            row_first = a.seg_host[u] + (int)t * bn;
            break;
          }
          t -= tu;
        }
      }
      const long work = (long)(a.rows_total - row_first) * ((a.N + 3) / 4);
      STTS_LAUNCH_PROF("splitk_reduce_kernel", (size_t)work * 4 * 4 * (ksp + 1), splitk_reduce_kernel, dim3((unsigned)std::min<long>(2048, (work + 255) / 256)), dim3(256), st, part, ksp, a.rows_total,
                         row_first, npad, a.N, a.bias, a.act, a.R, a.ldr, a.rcol0, a.alpha, a.Y, a.ldy, a.ycol0);
    }
  };
  if (a.defer) *a.defer = SplitInfo{nullptr, 1, 0, 0};
  if (plan.full_rt > 0) launch_range(0, plan.full_rt, main_ksp);
  if (plan.rem_rt > 0) launch_range(plan.full_rt, plan.rem_rt, plan.rem_ksp);
  STTS_HIP(hipGetLastError());
  return 0;
}
#ifdef STTS_GEMM_TU_FORM
#if STTS_GEMM_TU_FORM == 0
void gemm_dispatch_f32(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1) {
  gemm_dispatch_tile<PREC_F32>(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
}
#elif STTS_GEMM_TU_FORM == 1
void gemm_dispatch_bf16(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1) {
  gemm_dispatch_tile<PREC_BF16>(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
}
#elif STTS_GEMM_TU_FORM == 2
void gemm_dispatch_f16(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1) {
  gemm_dispatch_tile<PREC_F16>(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
}
#elif STTS_GEMM_TU_FORM == 3
void gemm_dispatch_x3(hipStream_t st, const GemmArgs& as, int tile, int epi, int npad, int n_utt, int max_rows, hipEvent_t e0, hipEvent_t e1) {
  gemm_dispatch_tile_x3(st, as, tile, epi, npad, n_utt, max_rows, e0, e1);
}
#endif
#endif  // STTS_GEMM_TU_FORM
#endif  // STTS_GEMM_NO_LAUNCHER

}  // namespace stts
