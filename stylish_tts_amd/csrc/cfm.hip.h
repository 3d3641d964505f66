// CfmMelDecoder._forward on the GPU: the XUT estimator of the reference's flow-matching mel decoder (SURVEY.md §8f rank 4,
// §8a row 19; models/cfm/cfm_mel_decoder.py:190-398, models/xut/{xut,transformer,attention,axial_rope,adaln,layers,norm,time_emb}.py).
// Inference mode (no TREAD token dropout, cfm_mel_decoder.py:349).  Sequences are packed ([rows, C] time-major, utterance offsets),
// every Linear runs on conv_gemm_f32, attention on attention_kernel (phoneme.hip.h); what is new here are the small kernels around them
// (sine source, RMSNorm + shared AdaLN, axial RoPE, SwiGLU gate, gated residual, per-utterance MLPs) and the weight inventory.
// The workload is launch-bound (~250 launches of 4-10 us per evaluation at small batches): the kernels are written for exactness.
#pragma once
#include "model.hip.h"
#include "phoneme.hip.h"

namespace stts {

struct CfmDims {
  int feat = 80, asr = 768, spk = 1024, hidden = 256, emb = 256, depth = 4, enc_blocks = 1, dec_blocks = 2, prev_depth = 1, post_depth = 3, head_dim = 64;
};

struct XutBlockW {
  bool cross = false;
  PackedConv qkv, out, w12, w3, xq, xkv, xout;
  float* rope = nullptr;   // exp(log-frequencies) [heads][head_dim / 2]
  float* xrope = nullptr;
  float* n_attn = nullptr;  // RMSNorm weights
  float* n_mlp = nullptr;
  float* n_xattn = nullptr;
};

struct SmallLinear {  // a Linear applied to one row per utterance (dense fp32 weights [N][K])
  float* W = nullptr;
  float* b = nullptr;
  int N = 0, K = 0;
};

struct CfmModel {
  CfmDims d;
  PackedConv asr1, asr3, prior, in_x, in_asr, in_spk, out_proj;
  SmallLinear spk0, spk2, time_proj, ad1[3], ad3[3];
  float* ad_g[3] = {nullptr, nullptr, nullptr};
  float* ad_b[3] = {nullptr, nullptr, nullptr};
  float* time_freqs = nullptr;
  float merge_w = 1.0f;
  std::vector<XutBlockW> blocks;  // execution order: prev, enc (depth x enc_blocks), dec (depth x dec_blocks), post
  std::vector<int> role;          // per block: 0 plain, 1 = last block of an encoder level (its output is remembered), 2 = first decoder block (cross)
};

// ---------------------------------------------------------------------------------------------------------------- kernels
__device__ __forceinline__ float mishf(float v) {
  const float sp = v > 20.0f ? v : log1pf(expf(v));  // F.softplus, threshold 20
  return v * tanhf(sp);
}

// Y[u][j] = act(W[j] . X[u] + b[j]); one wave per output.  act: 0 none, 1 Mish.
__global__ void __launch_bounds__(256) small_linear_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, const float* __restrict__ b,
                                                           int N, int K, int act, float* __restrict__ Y, int ldy, int n_rows) {
  const int lane = threadIdx.x & 63;
  const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long)n_rows * N) return;
  const int u = (int)(o / N), j = (int)(o % N);
  const float* x = X + (long)u * ldx;
  const float* w = W + (long)j * K;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += x[k] * w[k];
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + (b ? b[j] : 0.f);
    if (act == 1) v = mishf(v);
    Y[(long)u * ldy + j] = v;
  }
}

// TimestepEmbedding (xut/time_emb.py:24-31): emb[u] = [cos(1000 t f_i) | sin(1000 t f_i)]
__global__ void __launch_bounds__(256) cfm_time_embed_kernel(const float* __restrict__ t, const float* __restrict__ freqs, int half, float* __restrict__ E,
                                                             int lde, int n_utt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_utt * half) return;
  const int u = i / half, k = i % half;
  const float a = (1000.0f * t[u]) * freqs[k];
  E[(long)u * lde + k] = cosf(a);
  E[(long)u * lde + half + k] = sinf(a);
}

// LayerNorm over C of one row per utterance (nn.LayerNorm, eps 1e-5, affine); one wave per row.
__global__ void __launch_bounds__(64) small_layernorm_kernel(const float* __restrict__ X, int ldx, int C, const float* __restrict__ g, const float* __restrict__ b,
                                                             float* __restrict__ Y, int ldy) {
  const int u = blockIdx.x, lane = threadIdx.x;
  const float* x = X + (long)u * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = x[c] - mean;
    q += d * d;
  }
  const float inv = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
  for (int c = lane; c < C; c += 64) Y[(long)u * ldy + c] = (x[c] - mean) * inv * g[c] + b[c];
}

// Source features of one utterance (cfm_mel_decoder.py:321-330): F0 / N resampled to n frames by F.interpolate's 'nearest' rule,
// SineGenerator with one component (:54-104; the random initial phase of the fundamental is zeroed, :68; torch's CPU cumsum of
// fp32 accumulates in double and rounds every partial sum to fp32 where it is used, reproduced here), merge = tanh(w * s), then
// har[row] = (source, N, t, 0 ...) in a 32-column buffer.  One block of 256 threads per utterance.  The two running sums are PREFIX
// SUMS IN DOUBLE of fp32 terms (rad in [0, 1), rad + shift in (-1, 1), at most a few thousand of them): every partial sum is exact in
// 53 bits, so a blocked scan gives bit for bit what the sequential loop gives (round 2 walked the frames with one thread: 0.43 ms at
// 8 x 800 frames, 6 % of an estimator evaluation).
__global__ void __launch_bounds__(256) cfm_source_kernel(const float* __restrict__ f0, const float* __restrict__ ncurve, const int* __restrict__ curve_off,
                                                         const int* __restrict__ seg_off, const float* __restrict__ t, const float* __restrict__ noise,
                                                         float merge_w, float* __restrict__ har, int ldh) {
  __shared__ double part[256];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  const int clo = curve_off[u], L = curve_off[u + 1] - clo;
  const float scale = (float)L / (float)n;
  auto src_idx = [&](int i) { return L == n ? i : min((int)floorf((float)i * scale), L - 1); };
  for (int i = tid; i < n; i += 256) {
    float* h = har + (long)(lo + i) * ldh;
    h[1] = ncurve[clo + src_idx(i)];
    h[2] = t[u];
    for (int c = 3; c < ldh; ++c) h[c] = 0.f;
  }
  // thread tid owns frames [a, b): exclusive prefix over the threads through LDS, then the running sum inside the chunk
  const int per = (n + 255) / 256, a = min(n, tid * per), b = min(n, a + per);
  auto rad_of = [&](int i) { return fmodf(f0[clo + src_idx(i)] / 24000.0f, 1.0f); };
  auto block_exclusive = [&](double mine) {  // sum of `mine` over the threads before this one (fixed order; exact, see above)
    __syncthreads();
    part[tid] = mine;
    __syncthreads();
    double run = 0.0;
    for (int k = 0; k < tid; ++k) run += part[k];
    return run;
  };
  double s1 = 0.0;
  for (int i = a; i < b; ++i) s1 += (double)rad_of(i);
  double c1 = block_exclusive(s1);  // cumsum(rad) before frame a
  // shift[i] needs fmod(cumsum(rad)[i]) and the same of frame i - 1
  float prev = a > 0 ? fmodf((float)c1, 1.0f) : 0.f;
  double s2 = 0.0;
  for (int i = a; i < b; ++i) {
    const float rad = rad_of(i);
    c1 += (double)rad;
    const float tmp = fmodf((float)c1, 1.0f);
    const float shift = (i > 0 && tmp - prev < 0.f) ? -1.0f : 0.0f;
    prev = tmp;
    s2 += (double)(rad + shift);
  }
  double c2 = block_exclusive(s2);  // cumsum(rad + shift) before frame a
  c1 -= s1;                         // back to the prefix before frame a
  prev = a > 0 ? fmodf((float)c1, 1.0f) : 0.f;
  for (int i = a; i < b; ++i) {
    const float f = f0[clo + src_idx(i)];
    const float rad = fmodf(f / 24000.0f, 1.0f);
    c1 += (double)rad;
    const float tmp = fmodf((float)c1, 1.0f);
    const float shift = (i > 0 && tmp - prev < 0.f) ? -1.0f : 0.0f;
    prev = tmp;
    c2 += (double)(rad + shift);
    const float sine = sinf(((float)c2 * 2.0f) * 3.14159265358979323846f);
    const float uv = f > 0.f ? 1.0f : 0.0f;
    const float namp = uv * 0.003f + (1.0f - uv) * 0.1f / 3.0f;
    const float sw = (sine * 0.1f) * uv + namp * noise[lo + i];
    har[(long)(lo + i) * ldh] = tanhf(sw * merge_w);
  }
}

// RMSNorm (eps 1e-6, xut/norm.py:25-40) + shared AdaLN modulation (xut/adaln.py:19-27): y = x / rms(x) * w * (scale_u + 1) + shift_u.
// ada: [n_utt][3 C] = scale | shift | gate.  One wave per row; grid (row groups of 4, n_utt).
// With T != nullptr the row first receives the previous branch's gated residual, x = X + T * (gate_prev_u + 1) (transformer.py:67-79:
// X is the NORMALISED tensor that branch was fed with), so a branch's residual add and the next branch's norm are one launch;
// Xout (optional) keeps x itself (the value a later cross-attention or the output projection reads).  Y may alias X.
__global__ void __launch_bounds__(256) rms_adaln_kernel(const float* __restrict__ X, int ldx, int C, const float* __restrict__ w, const float* __restrict__ ada,
                                                        const int* __restrict__ seg_off, float* __restrict__ Y, int ldy, const float* __restrict__ T,
                                                        const float* __restrict__ ada_prev, float* __restrict__ Xout) {
  const int u = blockIdx.y, lane = threadIdx.x & 63;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float* x = X + (long)(lo + r) * ldx;
  const float* t = T ? T + (long)(lo + r) * C : nullptr;
  const float* gp = T ? ada_prev + (long)u * 3 * C + 2 * C : nullptr;
  constexpr int kMaxPer = 16;  // C <= 1024
  float v[kMaxPer];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPer; ++i) {
    const int c = lane + 64 * i;
    float xv = 0.f;
    if (c < C) {
      xv = x[c];
      if (t) xv += t[c] * (gp[c] + 1.0f);
    }
    v[i] = xv;
    q += xv * xv;
  }
  const float inv = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-6f);
  const float* a = ada + (long)u * 3 * C;
  float* y = Y + (long)(lo + r) * ldy;
  float* xo = Xout ? Xout + (long)(lo + r) * C : nullptr;
#pragma unroll
  for (int i = 0; i < kMaxPer; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      if (xo) xo[c] = v[i];
      y[c] = (v[i] * inv * w[c]) * (a[c] + 1.0f) + a[C + c];
    }
  }
}

// Axial RoPE with one position axis (xut/axial_rope.py:10-29,122-149) in place on a column block of H heads x d features:
// position of row p of an n-row utterance = torch.linspace(-1, 1, n)[p]; pair i of head h turns by pos * freq[h][i]:
//   (x0, x1) -> (x0 cos - x1 sin, x1 cos + x0 sin).
// col1 >= 0: the same rotation also on the column block at col1 (queries and keys in one launch).
__global__ void __launch_bounds__(256) axial_rope_kernel(float* __restrict__ X, int ldx, int col0, int col1, int heads, int d, const float* __restrict__ freq,
                                                         const int* __restrict__ seg_off) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  const int half = d / 2;
  const long total = (long)n * heads * half;
  const float step = n > 1 ? 2.0f / (float)(n - 1) : 0.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % half), h = (int)((i / half) % heads), p = (int)(i / ((long)half * heads));
    const float pos = n == 1 ? -1.0f : (p < n / 2 ? -1.0f + (float)p * step : 1.0f - (float)(n - 1 - p) * step);
    const float ang = pos * freq[h * half + k];
    const float cs = cosf(ang), sn = sinf(ang);
    float* x = X + (long)(lo + p) * ldx + col0 + h * d + 2 * k;
    const float a = x[0], b = x[1];
    x[0] = a * cs - b * sn;
    x[1] = b * cs + a * sn;
    if (col1 >= 0) {
      float* y = X + (long)(lo + p) * ldx + col1 + h * d + 2 * k;
      const float a2 = y[0], b2 = y[1];
      y[0] = a2 * cs - b2 * sn;
      y[1] = b2 * cs + a2 * sn;
    }
  }
}

// Y = H + T * (gate_u + 1)   (transformer.py:67-79: the residual is taken from the NORMALISED tensor H the branch was fed with)
__global__ void __launch_bounds__(256) gated_residual_kernel(const float* __restrict__ H, const float* __restrict__ T, int C, const float* __restrict__ ada,
                                                             const int* __restrict__ seg_off, float* __restrict__ Y) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  const float* g = ada + (long)u * 3 * C + 2 * C;
  const int nv = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n * nv; i += (long)gridDim.x * 256) {
    const int r = (int)(i / nv), c = (int)(i % nv) * 4;
    const long o = (long)(lo + r) * C + c;
    const float4 h = *reinterpret_cast<const float4*>(H + o), t = *reinterpret_cast<const float4*>(T + o), gg = *reinterpret_cast<const float4*>(g + c);
    *reinterpret_cast<float4*>(Y + o) = make_float4(h.x + t.x * (gg.x + 1.0f), h.y + t.y * (gg.y + 1.0f), h.z + t.z * (gg.z + 1.0f), h.w + t.w * (gg.w + 1.0f));
  }
}

// in place: X = mish(X) over a dense [rows, C] buffer
__global__ void __launch_bounds__(256) mish_kernel(float* __restrict__ X, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = reinterpret_cast<float4*>(X)[i];
    v.x = mishf(v.x); v.y = mishf(v.y); v.z = mishf(v.z); v.w = mishf(v.w);
    reinterpret_cast<float4*>(X)[i] = v;
  }
}

// SwiGLU gate (xut/layers.py:23-29): U = silu(A[:, :M]) * A[:, M:]
__global__ void __launch_bounds__(256) swiglu_kernel(const float* __restrict__ A, int M, float* __restrict__ U, long rows) {
  const int nv = M / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * nv; i += (long)gridDim.x * 256) {
    const long r = i / nv;
    const int c = (int)(i % nv) * 4;
    const float4 a = *reinterpret_cast<const float4*>(A + r * 2 * M + c), b = *reinterpret_cast<const float4*>(A + r * 2 * M + M + c);
    auto f = [](float x, float y) { return x / (1.0f + expf(-x)) * y; };
    *reinterpret_cast<float4*>(U + r * M + c) = make_float4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w));
  }
}

// ---------------------------------------------------------------------------------------------------------------- weights
inline int pack_small(stts_ctx* c, const std::string& p, SmallLinear* o) {
  STTS_GET(w, p + ".weight");
  STTS_GET(b, p + ".bias");
  STTS_CHECK(w->shape.size() == 2 && b->data.size() == (size_t)w->shape[0], "'%s': expected a Linear", p.c_str());
  o->N = (int)w->shape[0];
  o->K = (int)w->shape[1];
  STTS_TRY(dev_upload(c, w->data, &o->W));
  STTS_TRY(dev_upload(c, b->data, &o->b));
  return 0;
}
inline int upload_vec(stts_ctx* c, const std::string& name, size_t n, float** out, bool exp_it = false) {
  STTS_GET(v, name);
  STTS_CHECK(v->data.size() == n, "'%s': %zu elements, expected %zu", name.c_str(), v->data.size(), n);
  std::vector<float> h = v->data;
  if (exp_it)
    for (auto& x : h) x = expf(x);
  return dev_upload(c, h, out);
}
inline int pack_xut_block(stts_ctx* c, const std::string& p, const CfmDims& d, bool cross, XutBlockW* o) {
  const int dim = d.hidden, heads = dim / d.head_dim;
  o->cross = cross;
  STTS_TRY(pack_plain(c, p + ".attn.qkv", false, 0, -1, &o->qkv));
  STTS_TRY(pack_plain(c, p + ".attn.out", true, 0, -1, &o->out));
  STTS_TRY(upload_vec(c, p + ".attn.rope.freqs", (size_t)heads * d.head_dim / 2, &o->rope, true));
  STTS_TRY(pack_plain(c, p + ".mlp.w12", true, 0, -1, &o->w12));
  STTS_TRY(pack_plain(c, p + ".mlp.w3", true, 0, -1, &o->w3));
  STTS_TRY(upload_vec(c, p + ".attn_pre_norm.norm.weight", dim, &o->n_attn));
  STTS_TRY(upload_vec(c, p + ".mlp_pre_norm.norm.weight", dim, &o->n_mlp));
  STTS_CHECK(o->qkv.N == 3 * dim && o->out.N == dim && o->w12.N == 8 * dim && o->w3.N == dim, "'%s': layer sizes do not match hidden_dim %d", p.c_str(), dim);
  if (cross) {
    STTS_TRY(pack_plain(c, p + ".xattn.q", false, 0, -1, &o->xq));
    STTS_TRY(pack_plain(c, p + ".xattn.kv", false, 0, -1, &o->xkv));
    STTS_TRY(pack_plain(c, p + ".xattn.out", true, 0, -1, &o->xout));
    STTS_TRY(upload_vec(c, p + ".xattn.rope.freqs", (size_t)heads * d.head_dim / 2, &o->xrope, true));
    STTS_TRY(upload_vec(c, p + ".xattn_pre_norm.norm.weight", dim, &o->n_xattn));
  }
  return 0;
}

inline int finalize_cfm(stts_ctx* c, const CfmDims& d, CfmModel* M) {
  const std::string P = "cfm_mel_decoder.";
  STTS_CHECK(d.hidden % d.head_dim == 0 && d.head_dim % 4 == 0 && d.head_dim <= kAttnMaxKc, "cfm: hidden_dim %d / head_dim %d unsupported", d.hidden, d.head_dim);
  STTS_CHECK(d.hidden % 32 == 0 && d.emb % 32 == 0 && d.feat >= 1 && d.depth >= 1 && d.enc_blocks >= 1 && d.dec_blocks >= 1 && d.prev_depth >= 0 && d.post_depth >= 0,
             "cfm: bad dimensions (hidden_dim and emb_dim must be multiples of 32)");
  M->d = d;
  STTS_TRY(pack_plain(c, P + "asr_emb.1", true, 0, -1, &M->asr1));
  STTS_TRY(pack_plain(c, P + "asr_emb.3", true, 0, -1, &M->asr3));
  STTS_TRY(pack_small(c, P + "spk_emb.0", &M->spk0));
  STTS_TRY(pack_small(c, P + "spk_emb.2", &M->spk2));
  STTS_TRY(pack_small(c, P + "time_emb.proj.0", &M->time_proj));
  STTS_TRY(upload_vec(c, P + "time_emb.freqs", d.hidden / 2, &M->time_freqs));
  {
    STTS_GET(mw, P + "m_source.1.merge.0.weight");
    STTS_CHECK(mw->data.size() == 1, "cfm: SineGenerator with harmonics is not supported (merge weight has %zu elements)", mw->data.size());
    M->merge_w = mw->data[0];
  }
  STTS_TRY(pack_plain(c, P + "prior_generator.1", true, 0, -1, &M->prior));
  STTS_TRY(pack_plain(c, P + "in_proj", true, 0, d.feat, &M->in_x));
  STTS_TRY(pack_plain(c, P + "in_proj", false, d.feat, d.emb, &M->in_asr));
  STTS_TRY(pack_plain(c, P + "in_proj", false, d.feat + d.emb, d.emb, &M->in_spk));
  STTS_TRY(pack_plain(c, P + "out_proj.0", true, 0, -1, &M->out_proj));
  STTS_CHECK(M->asr1.cin_real == d.asr && M->asr3.N == d.emb && M->spk0.K == d.spk && M->spk2.N == d.emb && M->prior.N == d.feat && M->out_proj.N == d.feat &&
                 M->in_x.N == d.hidden && M->time_proj.N == d.hidden,
             "cfm: tensor shapes do not match the given dimensions");
  const char* ad[3] = {"shared_adaln_attn", "shared_adaln_xattn", "shared_adaln_ffw"};
  for (int k = 0; k < 3; ++k) {
    STTS_TRY(upload_vec(c, P + ad[k] + ".0.weight", d.hidden, &M->ad_g[k]));
    STTS_TRY(upload_vec(c, P + ad[k] + ".0.bias", d.hidden, &M->ad_b[k]));
    STTS_TRY(pack_small(c, P + ad[k] + ".1", &M->ad1[k]));
    STTS_TRY(pack_small(c, P + ad[k] + ".3", &M->ad3[k]));
    STTS_CHECK(M->ad3[k].N == 3 * d.hidden, "cfm: %s must produce scale | shift | gate", ad[k]);
  }
  M->blocks.clear();
  M->role.clear();
  auto add = [&](const std::string& p, bool cross, int role) -> int {
    M->blocks.emplace_back();
    M->role.push_back(role);
    return pack_xut_block(c, P + p, d, cross, &M->blocks.back());
  };
  for (int i = 0; i < d.prev_depth; ++i) STTS_TRY(add("prev_tread_trns.blocks." + std::to_string(i), false, 0));
  for (int i = 0; i < d.depth; ++i)
    for (int j = 0; j < d.enc_blocks; ++j) STTS_TRY(add("backbone.enc_blocks." + std::to_string(i) + "." + std::to_string(j), false, j == d.enc_blocks - 1 ? 1 : 0));
  for (int i = 0; i < d.depth; ++i)
    for (int j = 0; j < d.dec_blocks; ++j) STTS_TRY(add("backbone.dec_blocks." + std::to_string(i) + "." + std::to_string(j), j == 0, j == 0 ? 2 : 0));
  for (int i = 0; i < d.post_depth; ++i) STTS_TRY(add("post_tread_trns.blocks." + std::to_string(i), false, 0));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------- forward
inline int run_small(hipStream_t st, const SmallLinear& w, const float* X, int ldx, int act, float* Y, int ldy, int n_rows) {
  hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)ceil_div((long)n_rows * w.N, 4L)), dim3(256), 0, st, X, ldx, w.W, w.b, w.N, w.K, act, Y, ldy, n_rows);
  STTS_HIP(hipGetLastError());
  return 0;
}
inline int cfm_linear(hipStream_t st, const Seg& s, const float* X, int ldx, const PackedConv& w, int act, float* Y, int ldy, const float* R = nullptr, int ldr = 0) {
  GemmArgs a = gemm_args(s);
  set_seg(a, 0, X, ldx, 0, w);
  a.N = w.N; a.bias = w.bias; a.act = act; a.Y = Y; a.ldy = ldy; a.R = R; a.ldr = ldr;
  return launch_conv_gemm(st, a, EPI_STORE, w.npad, s.n_utt, s.max_len());
}

// One evaluation of the estimator.  x [rows, ld_x] (feat columns), asr [rows, ld_asr], f0 / ncurve packed per utterance with curve_off
// (device ints, n_utt + 1), spk [n_utt, spk], t [n_utt], noise [rows] (the SineGenerator's randn draw), out [rows, ld_out] = dphi/dt.
inline int cfm_estimator(stts_ctx* c, const CfmModel& M, hipStream_t st, const Seg& s, const float* x, int ld_x, const float* asr, int ld_asr, const float* f0,
                         const float* ncurve, const int* curve_off_dev, const float* spk, const float* t, const float* noise, float* out, int ld_out, Arena& ws) {
  const CfmDims& d = M.d;
  const long R = s.rows();
  const int dim = d.hidden, mlp = 4 * dim, heads = dim / d.head_dim, U = s.n_utt;
  const int ldf = round_up(d.feat, 32);
  float* har = ws.get<float>(R * 32);
  float* xp = ws.get<float>(R * ldf);
  float* a1 = ws.get<float>(R * 4 * d.emb);
  float* ae = ws.get<float>(R * d.emb);
  float* sb = ws.get<float>(R * d.emb);
  float* h = ws.get<float>(R * dim);
  float* hn = ws.get<float>(R * dim);
  float* tmp = ws.get<float>(R * dim);
  float* att = ws.get<float>(R * dim);
  float* ctx = ws.get<float>(R * dim);
  float* qkv = ws.get<float>(R * 3 * dim);
  float* a12 = ws.get<float>(R * 2 * mlp);
  float* um = ws.get<float>(R * mlp);
  float* s1 = ws.get<float>((size_t)U * 4 * d.emb);
  float* se = ws.get<float>((size_t)U * d.emb);
  float* te0 = ws.get<float>((size_t)U * dim);
  float* te = ws.get<float>((size_t)U * dim);
  float* tl = ws.get<float>((size_t)U * dim);
  float* tm = ws.get<float>((size_t)U * 4 * dim);
  float* ada = ws.get<float>((size_t)3 * U * 3 * dim);
  STTS_CHECK(ws.ok, "cfm_estimator: workspace too small");
  STTS_DRY_RETURN(ws);
  STTS_CHECK(ld_x >= d.feat && ld_asr % 32 == 0 && ld_asr >= d.asr && ld_out >= d.feat, "cfm_estimator: leading dimensions too small (asr rows must be padded to 32 columns)");
  STTS_CHECK(attn_mfma_kc(d.head_dim) || s.max_len() <= kAttnMaxKeys, "cfm_estimator: with head_dim %d utterances of more than %d frames are not supported (%d)", d.head_dim,
             kAttnMaxKeys, s.max_len());
  // conditioning that does not depend on x
  STTS_TRY(cfm_linear(st, s, asr, ld_asr, M.asr1, ACT_NONE, a1, 4 * d.emb));
  hipLaunchKernelGGL(mish_kernel, dim3((unsigned)std::min<long>(2048, ceil_div(R * d.emb, 256L))), dim3(256), 0, st, a1, R * d.emb);  // R * 4 emb / 4 float4s
  STTS_TRY(cfm_linear(st, s, a1, 4 * d.emb, M.asr3, ACT_NONE, ae, d.emb));
  STTS_TRY(run_small(st, M.spk0, spk, d.spk, 1, s1, 4 * d.emb, U));
  STTS_TRY(run_small(st, M.spk2, s1, 4 * d.emb, 0, se, d.emb, U));
  hipLaunchKernelGGL(broadcast_style_kernel, dim3(std::max(1, ceil_div(s.max_len() * d.emb, 256)), U), dim3(256), 0, st, se, d.emb, d.emb, sb, d.emb, 0, s.dev);
  hipLaunchKernelGGL(cfm_source_kernel, dim3(U), dim3(256), 0, st, f0, ncurve, curve_off_dev, s.dev, t, noise, M.merge_w, har, 32);
  // x + prior_generator(har) (k = 7 conv over the three source features), then in_proj over [x | asr_emb | spk_emb] as three K segments
  // (xp's pad columns meet zero weights in in_proj, but the contraction reads them: they must be finite)
  if (ldf > d.feat) STTS_HIP(hipMemsetAsync(xp, 0, (size_t)R * ldf * sizeof(float), st));
  STTS_TRY(cfm_linear(st, s, har, 32, M.prior, ACT_NONE, xp, ldf, x, ld_x));
  {
    GemmArgs a = gemm_args(s);
    set_seg(a, 0, xp, ldf, 0, M.in_x);
    set_seg(a, 1, ae, d.emb, 0, M.in_asr);
    set_seg(a, 2, sb, d.emb, 0, M.in_spk);
    a.N = dim; a.bias = M.in_x.bias; a.Y = h; a.ldy = dim;
    STTS_TRY(launch_conv_gemm(st, a, EPI_STORE, M.in_x.npad, U, s.max_len()));
  }
  // time embedding and the three shared AdaLN states (scale | shift | gate per utterance)
  hipLaunchKernelGGL(cfm_time_embed_kernel, dim3(ceil_div(U * (dim / 2), 256)), dim3(256), 0, st, t, M.time_freqs, dim / 2, te0, dim, U);
  STTS_TRY(run_small(st, M.time_proj, te0, dim, 1, te, dim, U));
  for (int k = 0; k < 3; ++k) {
    hipLaunchKernelGGL(small_layernorm_kernel, dim3(U), dim3(64), 0, st, te, dim, dim, M.ad_g[k], M.ad_b[k], tl, dim);
    STTS_TRY(run_small(st, M.ad1[k], tl, dim, 1, tm, 4 * dim, U));
    STTS_TRY(run_small(st, M.ad3[k], tm, 4 * dim, 0, ada + (size_t)k * U * 3 * dim, 3 * dim, U));
  }
  const dim3 rgrid(ceil_div(s.max_len(), 4), U), egrid(std::max(1, std::min(1024, ceil_div(s.max_len() * (dim / 4), 256))), U);
  STTS_CHECK(dim <= 1024, "cfm_estimator: hidden_dim %d > 1024", dim);
  auto rope2 = [&](float* X, int ldx, int col0, int col1, const float* fr) {  // queries and keys of a q | k | v buffer
    hipLaunchKernelGGL(axial_rope_kernel, dim3(std::max(1, std::min(1024, ceil_div(s.max_len() * heads * (d.head_dim / 2), 256))), U), dim3(256), 0, st, X, ldx, col0,
                       col1, heads, d.head_dim, fr, s.dev);
  };
  // norm(x) for the next branch, fused with the gated residual of the branch that just finished (t != nullptr): hn <- adaln(hn + t * gate)
  auto norm = [&](const float* x, const float* w, const float* a_next, const float* t, const float* a_prev, float* xout) {
    hipLaunchKernelGGL(rms_adaln_kernel, rgrid, dim3(256), 0, st, x, dim, dim, w, a_next, s.dev, hn, dim, t, a_prev, xout);
  };
  const float* a_attn = ada;
  const float* a_x = ada + (size_t)1 * U * 3 * dim;
  const float* a_mlp = ada + (size_t)2 * U * 3 * dim;
  float* const other = tmp;  // the finished branch's projection; its gated residual is applied by the next norm launch
  for (size_t b = 0; b < M.blocks.size(); ++b) {
    const XutBlockW& B = M.blocks[b];
    // self-attention branch (its input norm also closes the previous block's SwiGLU branch)
    if (b == 0) norm(h, B.n_attn, a_attn, nullptr, nullptr, nullptr);
    else norm(hn, B.n_attn, a_attn, other, a_mlp, M.role[b - 1] == 1 ? ctx : nullptr);  // an encoder level's output is remembered (the newest wins)
    STTS_TRY(cfm_linear(st, s, hn, dim, B.qkv, ACT_NONE, qkv, 3 * dim));
    rope2(qkv, 3 * dim, 0, dim, B.rope);
    STTS_TRY(run_attention(st, s, s, qkv, 3 * dim, 0, qkv, 3 * dim, dim, qkv, 3 * dim, 2 * dim, att, dim, heads, d.head_dim, nullptr, 0));
    STTS_TRY(cfm_linear(st, s, att, dim, B.out, ACT_NONE, other, dim));
    const float* a_prev = a_attn;
    if (B.cross) {  // cross-attention to the last encoder level's output (xut.py:199-203: self_ctx[-1] for every decoder level)
      norm(hn, B.n_xattn, a_x, other, a_prev, nullptr);
      STTS_TRY(cfm_linear(st, s, hn, dim, B.xq, ACT_NONE, qkv, 3 * dim));            // q -> columns [0, dim)
      STTS_TRY(cfm_linear(st, s, ctx, dim, B.xkv, ACT_NONE, qkv + dim, 3 * dim));     // k | v -> columns [dim, 3 dim)
      rope2(qkv, 3 * dim, 0, dim, B.xrope);
      STTS_TRY(run_attention(st, s, s, qkv, 3 * dim, 0, qkv, 3 * dim, dim, qkv, 3 * dim, 2 * dim, att, dim, heads, d.head_dim, nullptr, 0));
      STTS_TRY(cfm_linear(st, s, att, dim, B.xout, ACT_NONE, other, dim));
      a_prev = a_x;
    }
    // SwiGLU branch
    norm(hn, B.n_mlp, a_mlp, other, a_prev, nullptr);
    STTS_TRY(cfm_linear(st, s, hn, dim, B.w12, ACT_NONE, a12, 2 * mlp));
    hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)std::min<long>(2048, ceil_div(R * (mlp / 4), 256L))), dim3(256), 0, st, a12, mlp, um, R);
    STTS_TRY(cfm_linear(st, s, um, mlp, B.w3, ACT_NONE, other, dim));
  }
  // the last block's SwiGLU residual: x = hn + other * (gate + 1)
  float* cur = h;
  hipLaunchKernelGGL(gated_residual_kernel, egrid, dim3(256), 0, st, hn, other, dim, a_mlp, s.dev, cur);
  STTS_TRY(cfm_linear(st, s, cur, dim, M.out_proj, ACT_NONE, out, ld_out));
  STTS_HIP(hipGetLastError());
  return 0;
}

// Workspace bytes of one estimator evaluation over `R` rows in `n_utt` utterances (a dry run of the carving above).
inline size_t cfm_workspace_bytes(stts_ctx* c, const CfmModel& M, int64_t R, int n_utt) {
  if (R <= 0 || n_utt <= 0) return 0;
  std::vector<int> off(n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) off[u + 1] = (int)(R * (u + 1) / n_utt);
  Seg s{n_utt, off.data(), nullptr};
  DryRun& dr = dry_run();
  dr.on = true;
  dr.peak = 0;
  Arena a(reinterpret_cast<char*>((uintptr_t)1 << 20), (size_t)1 << 46);
  (void)cfm_estimator(c, M, nullptr, s, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, a);
  dr.on = false;
  return dr.peak + 4096;
}

}  // namespace stts
