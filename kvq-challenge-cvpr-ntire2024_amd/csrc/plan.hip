// Host side of the trunk: shape plan (integer index maps) + the launch sequence of
// SwinTransformer3D.forward (swin_backbone.py:1044-1080) over the kernels in this library.
//
// The plan replaces the reference's lru_cached helper tensors:
//   compute_mask            (swin_backbone.py:559-586)  -> per-token region id (1 byte)
//   global_position_index   (:21-50)                    -> per-token fragment ids (2 bytes)
//   relative_position_index (:213-235)                  -> per-token linear position code
//   pad + roll + window_partition / window_reverse + roll + crop (:418-488) -> one gather map
//   PatchMerging's strided slices + pad (:542-550)      -> one 4-neighbour map
// A (nW,N,N,3) int64 tensor (472 MB at stage 0) becomes 8 bytes per token.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.hpp"
#include "tail.hpp"

// KVQ_SKIP (diagnostic, tools/skip_ablation.sh): bit mask of launch families the forward leaves out — the scores are garbage, the point is the
// marginal cost of a family in the multi-lane bench line (what the step would gain if the family were free), which the per-launch times
// of a one-stream step do not tell.  1 attention, 2 fused tails C <= 192, 4 fused tails C >= 256, 8 qkv GEMMs, 16 LayerNorm launches,
// 32 every launch of the last stage, 64 patch embedding, 128 patch merging (all forms).
// Diagnostic BUILDS only (-DKVQ_DIAG: `KVQ_BUILD_TAG=diag KVQ_EXTRA_HIPCC_FLAGS=-DKVQ_DIAG`): the product library ignores the variable, so an
// inherited environment can never make kvq_swin3d_forward return KVQ_OK with garbage scores; a diagnostic build says so once on stderr.
#ifdef KVQ_DIAG
static int skip_mask() {
  static const int m = [] {
    const int v = getenv("KVQ_SKIP") ? atoi(getenv("KVQ_SKIP")) : 0;
    if (v) fprintf(stderr, "libkvq_hip (KVQ_DIAG build): KVQ_SKIP=%d leaves launch families out - scores are GARBAGE, timing only\n", v);
    return v;
  }();
  return m;
}
#else
static constexpr int skip_mask() { return 0; }
#endif

namespace kvq {

struct StageGeom {
  int D, H, W, C, nH, depth;
  int L;
  // index [0] = un-shifted blocks, [1] = shifted blocks
  int ws[3], ss[3];
  int Dp, Hp, Wp, nW, N, Lp;
  bool shifted_any;
  int32_t* d_src[2];  // [Lp] source token or -1
  int32_t* d_pad[2];  // [Lp - L] the padding rows of the partition (window order), nullptr when Lp == L
  int32_t* d_skip[2]; // [nW] bit t: rows 16t..16t+15 of the window are padding only (attention passes such q-tiles over), or nullptr
  int32_t* d_padmask[2];  // [nW][13] bit r & 31 of word r >> 5: window row r is a padding row (attention32 writes its k | v itself), or nullptr
  int32_t* d_dst[2];  // [L] inverse of d_src (token -> window row).  A bijection — the next block's norm1 rows can be EMITTED through it —
                      // only when Lp == L (no padding)
  int32_t* d_tok[2];  // [nW*N][2]
  int32_t* d_merge;   // [L_next][4] or nullptr
  int Dn, Hn, Wn;     // dims after the merge
};

struct ProfEvent {
  int kind, variant;
  double flops, bytes;   // ALGORITHMIC work of the launch
  hipEvent_t a, b;
};

}  // namespace kvq

struct KvqSwinPlan {
  KvqSwinCfg cfg;
  int B, T, H, W, dtype;
  int D0, H0, W0, K0;
  std::vector<kvq::StageGeom> st;
  std::vector<void*> owned;      // device allocations to free
  size_t ws_bytes;
  size_t off_x0, off_x1, off_ln, off_big, off_o, off_sk, sk_bytes;   // off_sk: split-K scratch of the un-fused GEMMs
  unsigned char* run_ws;         // workspace of the forward in progress (for the GEMM helper)
  int table_len, center;
  std::vector<float*> taps;      // feats[i] destinations (kvq_swin3d_set_taps), empty = none
  // profiling
  bool profile;
  std::vector<kvq::ProfEvent> events;
  size_t ev_used;
};

namespace kvq {

// ATen legacy 'nearest' source index (UpSampleKernel nearest_idx): float32 scale, floorf, clamp.
static int nearest_src(int dst, int in_size, int out_size) {
  if (out_size == in_size) return dst;
  if (out_size == 2 * in_size) return dst >> 1;
  const float scale = (float)in_size / (float)out_size;
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

static int axis_region(int s, int P, int w, int sft) {
  if (sft == 0) return 2;          // the last slice(-0, None) repaints the whole axis
  if (s >= P - sft) return 2;
  if (s >= P - w) return 1;
  return 0;
}

static int upload(KvqSwinPlan* pl, const std::vector<int32_t>& h, int32_t** out) {
  void* d = nullptr;
  KVQ_CHECK_HIP(hipMalloc(&d, h.size() * sizeof(int32_t)));
  pl->owned.push_back(d);
  KVQ_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  *out = (int32_t*)d;
  return KVQ_OK;
}

static int build_stage_maps(KvqSwinPlan* pl, StageGeom& g, int par) {
  const KvqSwinCfg& cfg = pl->cfg;
  const int dims[3] = {g.D, g.H, g.W};
  int ws[3], ss[3];
  // forward(adaptive_window_size=True): the resized window partitions, the block's shift stays that of the configured one
  const bool adaptive = cfg.adaptive_window[0] > 0;
  for (int a = 0; a < 3; ++a) {
    const int wa = adaptive ? cfg.adaptive_window[a] : cfg.window[a];
    const bool clamp = dims[a] <= wa;
    ws[a] = clamp ? dims[a] : wa;
    ss[a] = (clamp || par == 0) ? 0 : cfg.window[a] / 2;
  }
  if (par == 0) {
    memcpy(g.ws, ws, sizeof(ws));
    g.Dp = round_up(g.D, ws[0]); g.Hp = round_up(g.H, ws[1]); g.Wp = round_up(g.W, ws[2]);
    g.N = ws[0] * ws[1] * ws[2];
    g.nW = (g.Dp / ws[0]) * (g.Hp / ws[1]) * (g.Wp / ws[2]);
    g.Lp = g.nW * g.N;
  } else {
    memcpy(g.ss, ss, sizeof(ss));
    g.shifted_any = ss[0] > 0 || ss[1] > 0 || ss[2] > 0;
  }
  const int nd = g.Dp / ws[0], nh = g.Hp / ws[1], nw = g.Wp / ws[2];
  const int Wh = cfg.window[1], Ww = cfg.window[2];
  std::vector<int32_t> src((size_t)g.Lp), tok((size_t)g.Lp * 2);
  size_t r = 0;
  for (int wd = 0; wd < nd; ++wd)
    for (int wh = 0; wh < nh; ++wh)
      for (int ww = 0; ww < nw; ++ww) {
        int n = 0;
        for (int ld = 0; ld < ws[0]; ++ld)
          for (int lh = 0; lh < ws[1]; ++lh)
            for (int lw = 0; lw < ws[2]; ++lw, ++n, ++r) {
              const int sd = wd * ws[0] + ld, sh = wh * ws[1] + lh, sw = ww * ws[2] + lw;
              const int ud = (sd + ss[0]) % g.Dp, uh = (sh + ss[1]) % g.Hp, uw = (sw + ss[2]) % g.Wp;
              const bool valid = ud < g.D && uh < g.H && uw < g.W;
              src[r] = valid ? (ud * g.H + uh) * g.W + uw : -1;
              // bias code: raster coordinate of index n in the CONFIGURED window (reference slices
              // relative_position_index[:N,:N], swin_backbone.py:263-264)
              // adaptive windows index the table by the token's own coordinate in the resized window (:266-271)
              const int cd = adaptive ? ld : n / (Wh * Ww), ch = adaptive ? lh : (n / Ww) % Wh, cw = adaptive ? lw : n % Ww;
              const int code = cd * (2 * Wh - 1) * (2 * Ww - 1) + ch * (2 * Ww - 1) + cw;
              const int fh = nearest_src(uh, ws[1], g.Hp), fw = nearest_src(uw, ws[2], g.Wp);
              const int region = axis_region(sd, g.Dp, ws[0], ss[0]) * 9 + axis_region(sh, g.Hp, ws[1], ss[1]) * 3 +
                                 axis_region(sw, g.Wp, ws[2], ss[2]);
              tok[2 * r] = code;
              tok[2 * r + 1] = (fh & 0xff) | ((fw & 0xff) << 8) | ((region & 0xff) << 16);
            }
      }
  int rc = upload(pl, src, &g.d_src[par]);
  if (rc) return rc;
  g.d_dst[par] = nullptr;
  g.d_pad[par] = nullptr;
  {
    std::vector<int32_t> dst((size_t)g.L, 0), pad;
    for (int i = 0; i < g.Lp; ++i) {
      if (src[i] >= 0) dst[src[i]] = i;
      else pad.push_back(i);
    }
    rc = upload(pl, dst, &g.d_dst[par]);
    if (rc) return rc;
    g.d_skip[par] = nullptr;
    g.d_padmask[par] = nullptr;
    if (!pad.empty()) {
      rc = upload(pl, pad, &g.d_pad[par]);
      if (rc) return rc;
      if (g.N <= 13 * 32) {
        std::vector<int32_t> pm((size_t)g.nW * 13, 0);
        for (int i : pad) pm[(size_t)(i / g.N) * 13 + ((i % g.N) >> 5)] |= (int32_t)(1u << ((i % g.N) & 31));
        rc = upload(pl, pm, &g.d_padmask[par]);
        if (rc) return rc;
      }
      std::vector<int32_t> skip((size_t)g.nW, 0);
      bool any = false;
      for (int wv = 0; wv < g.nW; ++wv)
        for (int t = 0; t * 16 < g.N && t < 32; ++t) {
          bool all_pad = true;
          for (int q = t * 16; q < g.N && q < t * 16 + 16; ++q) all_pad = all_pad && src[(size_t)wv * g.N + q] < 0;
          if (all_pad) { skip[wv] |= (int32_t)(1u << t); any = true; }
        }
      if (any) {
        rc = upload(pl, skip, &g.d_skip[par]);
        if (rc) return rc;
      }
    }
  }
  return upload(pl, tok, &g.d_tok[par]);
}

static int build_merge_map(KvqSwinPlan* pl, StageGeom& g) {
  g.Dn = g.D; g.Hn = (g.H + 1) / 2; g.Wn = (g.W + 1) / 2;
  std::vector<int32_t> m((size_t)g.Dn * g.Hn * g.Wn * 4);
  size_t r = 0;
  for (int d = 0; d < g.D; ++d)
    for (int h2 = 0; h2 < g.Hn; ++h2)
      for (int w2 = 0; w2 < g.Wn; ++w2, ++r) {
        // concat order x0(0,0) x1(+h) x2(+w) x3(+h,+w)  (swin_backbone.py:546-550)
        const int hh[4] = {2 * h2, 2 * h2 + 1, 2 * h2, 2 * h2 + 1};
        const int wv[4] = {2 * w2, 2 * w2, 2 * w2 + 1, 2 * w2 + 1};
        for (int p = 0; p < 4; ++p)
          m[4 * r + p] = (hh[p] < g.H && wv[p] < g.W) ? (d * g.H + hh[p]) * g.W + wv[p] : -1;
      }
  return upload(pl, m, &g.d_merge);
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace kvq

extern "C" int kvq_swin3d_plan_create(const KvqSwinCfg* cfg, int B, int T, int H, int W, int dtype,
                                      KvqSwinPlan** out) {
  using namespace kvq;
  KVQ_REQUIRE(cfg && out, KVQ_ERR_NULL, "kvq_swin3d_plan_create: NULL pointer");
  KVQ_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0, KVQ_ERR_SHAPE, "kvq_swin3d_plan_create: bad input shape");
  KVQ_REQUIRE(cfg->num_stages >= 1 && cfg->num_stages <= KVQ_MAX_STAGES, KVQ_ERR_UNSUPPORTED, "num_stages %d",
              cfg->num_stages);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "unknown dtype %d", dtype);
  KVQ_REQUIRE(cfg->embed_dim % 32 == 0 && (cfg->in_chans * cfg->patch[0] * cfg->patch[1] * cfg->patch[2]) % 32 == 0,
              KVQ_ERR_UNSUPPORTED, "embed_dim and in_chans*prod(patch) must be multiples of 32");
  KVQ_REQUIRE(cfg->window[0] * cfg->window[1] * cfg->window[2] <= 400, KVQ_ERR_UNSUPPORTED,
              "window of more than 400 tokens unsupported");
  if (cfg->adaptive_window[0] || cfg->adaptive_window[1] || cfg->adaptive_window[2])
    for (int a = 0; a < 3; ++a)       // the reference's relative_position_index[:d,:h,:w,...] slice needs d <= Wd etc. (:266-271)
      KVQ_REQUIRE(cfg->adaptive_window[a] >= 1 && cfg->adaptive_window[a] <= cfg->window[a], KVQ_ERR_SHAPE,
                  "adaptive window (%d,%d,%d) must lie inside the configured one", cfg->adaptive_window[0], cfg->adaptive_window[1],
                  cfg->adaptive_window[2]);
  for (int i = 0; i < cfg->num_stages; ++i)
    KVQ_REQUIRE(cfg->num_heads[i] * 32 == (cfg->embed_dim << i), KVQ_ERR_UNSUPPORTED,
                "stage %d: head_dim must be 32 (C=%d, heads=%d)", i, cfg->embed_dim << i, cfg->num_heads[i]);
  KvqSwinPlan* pl = new KvqSwinPlan();
  pl->cfg = *cfg; pl->B = B; pl->T = T; pl->H = H; pl->W = W; pl->dtype = dtype;
  pl->profile = false; pl->ev_used = 0;
  pl->D0 = ceil_div(T, cfg->patch[0]); pl->H0 = ceil_div(H, cfg->patch[1]); pl->W0 = ceil_div(W, cfg->patch[2]);
  pl->K0 = cfg->in_chans * cfg->patch[0] * cfg->patch[1] * cfg->patch[2];
  const int Wd = cfg->window[0], Wh = cfg->window[1], Ww = cfg->window[2];
  pl->table_len = (2 * Wd - 1) * (2 * Wh - 1) * (2 * Ww - 1);
  pl->center = (Wd - 1) * (2 * Wh - 1) * (2 * Ww - 1) + (Wh - 1) * (2 * Ww - 1) + (Ww - 1);
  int D = pl->D0, Hh = pl->H0, Wv = pl->W0;
  size_t max_x = 0, max_ln = 0, max_big = (size_t)B * D * Hh * Wv * pl->K0, max_o = 0, max_sk = 0;
  for (int i = 0; i < cfg->num_stages; ++i) {
    StageGeom g{};
    g.D = D; g.H = Hh; g.W = Wv; g.C = cfg->embed_dim << i; g.nH = cfg->num_heads[i]; g.depth = cfg->depths[i];
    g.L = D * Hh * Wv;
    int rc = build_stage_maps(pl, g, 0);
    if (!rc) rc = build_stage_maps(pl, g, 1);
    if (!rc && i < cfg->num_stages - 1) rc = build_merge_map(pl, g);
    if (rc) { kvq_swin3d_plan_destroy(pl); return rc; }
    const size_t BL = (size_t)B * g.L, BLp = (size_t)B * g.Lp;
    max_x = std::max(max_x, BL * g.C);
    max_ln = std::max(max_ln, std::max(BLp * g.C, BL * g.C));
    max_big = std::max(max_big, std::max(BLp * 3 * g.C, BL * (size_t)cfg->mlp_ratio * g.C));
    max_o = std::max(max_o, BLp * g.C);
    {   // GEMM shapes of this stage that may run split-K (long K, few tiles: stage 3 of the trunk)
      const int Mw = (int)BLp, Mt = (int)BL, C = g.C, Hd = cfg->mlp_ratio * g.C;
      max_sk = std::max(max_sk, std::max(kvq_gemm_splitk_bytes(Mw, C, C), std::max(kvq_gemm_splitk_bytes(Mt, Hd, C),
                                                                                    kvq_gemm_splitk_bytes(Mt, C, Hd))));
      if (i < cfg->num_stages - 1)
        max_sk = std::max(max_sk, kvq_gemm_splitk_bytes(B * g.Dn * g.Hn * g.Wn, 2 * C, 4 * C));
    }
    pl->st.push_back(g);
    if (i < cfg->num_stages - 1) { Hh = g.Hn; Wv = g.Wn; }
  }
  size_t off = 0;
  pl->off_x0 = off; off += align_up(max_x * 4);
  pl->off_x1 = off; off += align_up(max_x * 4);
  pl->off_ln = off; off += align_up(max_ln * 2);
  pl->off_big = off; off += align_up(max_big * 2);
  pl->off_o = off; off += align_up(max_o * 2);
  pl->off_sk = off; off += align_up(max_sk);
  pl->sk_bytes = max_sk;
  pl->run_ws = nullptr;
  pl->ws_bytes = off;
  *out = pl;
  return KVQ_OK;
}

extern "C" void kvq_swin3d_plan_destroy(KvqSwinPlan* pl) {
  if (!pl) return;
  for (void* d : pl->owned) (void)hipFree(d);
  for (auto& e : pl->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  delete pl;
}

extern "C" size_t kvq_swin3d_workspace_bytes(const KvqSwinPlan* pl) { return pl ? pl->ws_bytes : 0; }

extern "C" int kvq_swin3d_out_dims(const KvqSwinPlan* pl, int32_t out4[4]) {
  using namespace kvq;
  KVQ_REQUIRE(pl && out4, KVQ_ERR_NULL, "kvq_swin3d_out_dims: NULL");
  const StageGeom& g = pl->st.back();
  out4[0] = g.C; out4[1] = g.D; out4[2] = g.H; out4[3] = g.W;
  return KVQ_OK;
}

namespace kvq {
// stage / parity of weights->blocks[block]
static bool locate_block(const KvqSwinPlan* pl, int block, int* stage, int* par) {
  int k = 0;
  for (int i = 0; i < pl->cfg.num_stages; ++i) {
    const StageGeom& g = pl->st[i];
    if (block < k + g.depth) {
      *stage = i;
      *par = ((block - k) & 1) && g.shifted_any ? 1 : 0;
      return block >= k;
    }
    k += g.depth;
  }
  return false;
}
}  // namespace kvq

namespace kvq {
// distinct attention biases of a block: un-shifted windows that differ only in their depth index share one (the
// position codes, fragment ids and — absent — mask regions of their tokens are identical); windows are ordered
// depth-major, so window w has type w % n_types
static int bias_types(const StageGeom& g, int par) { return par == 0 ? g.nW / (g.Dp / g.ws[0]) : g.nW; }
constexpr float kQScale = 0.17677669529663687f;              // head_dim^-0.5 = 32^-0.5 (swin_backbone.py:208)
constexpr float kQScaleLog2 = 0.17677669529663687f * 1.4426950408889634f;      // the streaming kernel keeps scores in log2 units
}  // namespace kvq

extern "C" size_t kvq_swin3d_bias_dense_bytes(const KvqSwinPlan* pl, int block) {
  int i = 0, par = 0;
  if (!pl || !kvq::locate_block(pl, block, &i, &par)) return 0;
  const kvq::StageGeom& g = pl->st[i];
  return kvq_attn_bias32_bytes(kvq::bias_types(g, par), g.N, g.nH);        // 0 past 400 tokens per window: such blocks keep the gather path
}

extern "C" int kvq_swin3d_bias_dense_build(const KvqSwinPlan* pl, int block, const float* rpb, const float* fpb, void* out,
                                           float* max_abs, void* stream) {
  using namespace kvq;
  int i = 0, par = 0;
  KVQ_REQUIRE(pl && rpb && out, KVQ_ERR_NULL, "kvq_swin3d_bias_dense_build: NULL pointer");
  KVQ_REQUIRE(locate_block(pl, block, &i, &par), KVQ_ERR_SHAPE, "kvq_swin3d_bias_dense_build: no block %d", block);
  const StageGeom& g = pl->st[i];
  return kvq_attn_bias32_build(g.d_tok[par], rpb, pl->cfg.frag_bias[i] ? fpb : nullptr, pl->table_len, pl->center,
                               bias_types(g, par), g.N, g.nH, par, out, max_abs, stream);
}

extern "C" int kvq_swin3d_profile(KvqSwinPlan* pl, int enable) {
  using namespace kvq;
  KVQ_REQUIRE(pl, KVQ_ERR_NULL, "kvq_swin3d_profile: NULL");
  pl->profile = enable != 0;
  pl->ev_used = 0;
  return KVQ_OK;
}

extern "C" int kvq_swin3d_profile_read(KvqSwinPlan* pl, KvqProfRecord* out, int max_records, int* n_records) {
  using namespace kvq;
  KVQ_REQUIRE(pl && out && n_records, KVQ_ERR_NULL, "kvq_swin3d_profile_read: NULL");
  int n = 0;
  for (size_t i = 0; i < pl->ev_used && n < max_records; ++i, ++n) {
    ProfEvent& e = pl->events[i];
    KVQ_CHECK_HIP(hipEventSynchronize(e.b));
    float t = 0.f;
    KVQ_CHECK_HIP(hipEventElapsedTime(&t, e.a, e.b));
    out[n].kind = e.kind; out[n].variant = e.variant; out[n].ms = t; out[n].flops = e.flops; out[n].bytes = e.bytes;
  }
  *n_records = n;
  pl->ev_used = 0;
  return KVQ_OK;
}

namespace kvq {

// RAII-less bracket: records start/stop events on the launch stream when profiling is on.
struct Bracket {
  KvqSwinPlan* pl;
  hipStream_t st;
  ProfEvent* e;
  Bracket(KvqSwinPlan* p, hipStream_t s, int kind, int variant, double flops, double bytes)
      : pl(p), st(s), e(nullptr) {
    if (!pl->profile) return;
    if (pl->ev_used == pl->events.size()) {
      ProfEvent n{kind, variant, 0.0, 0.0, nullptr, nullptr};
      if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
      pl->events.push_back(n);
    }
    e = &pl->events[pl->ev_used++];
    e->kind = kind; e->variant = variant; e->flops = flops; e->bytes = bytes;
    (void)hipEventRecord(e->a, st);
  }
  ~Bracket() {
    if (e) (void)hipEventRecord(e->b, st);
  }
};

static int gemm(KvqSwinPlan* pl, hipStream_t st, int kind, const uint16_t* A, const uint16_t* Wt, const float* bias,
                int M, int N, int K, int epi, uint16_t* obf, float* of32, int nH = 0, float qs = 1.f,
                const int32_t* map = nullptr, int map_rows = 0, int out_rows = 0) {
  KvqGemmArgs a{};
  a.A = A; a.W = Wt; a.bias = bias; a.M = M; a.N = N; a.K = K; a.epilogue = epi; a.out_bf16 = obf; a.out_f32 = of32;
  a.num_heads = nH; a.q_scale = qs; a.scatter_map = map; a.map_rows = map_rows; a.out_rows = out_rows;
  a.dtype = pl->dtype;
  // Split-K stays OFF in the trunk (splitk_ws = NULL): whether a GEMM splits depends on its row count, i.e. on the batch — a clip's
  // score would differ in the last bits with what else is in the batch (tests/test_gpu_e2e.py::test_batch_invariance_and_determinism),
  // and the trunk's only long-K / few-tile shape (fc2 of stage 3) gains nothing from it (43.6 vs 43.8 us).
  // algorithmic bytes: A + W once, output once (fp32 residual epilogues read-modify-write)
  const double out_b = (epi == KVQ_EPI_RESID_F32) ? 8.0 : (epi == KVQ_EPI_STORE_F32 ? 4.0 : 2.0);
  Bracket br(pl, st, kind, (gemm8p_wanted(M, N, K) ? 4464 : gemm_variant(M, N, K)) * 10 + epi, 2.0 * M * N * K,
             2.0 * ((double)M * K + (double)N * K) + out_b * M * N);
  return kvq_gemm_bf16(&a, st);
}

static int ln(KvqSwinPlan* pl, hipStream_t st, const float* x, const int32_t* map, int nparts, int rows_in,
              int rows_out, int Cin, const float* g, const float* b, uint16_t* obf, float* of32, bool x16 = false) {
  const double elems = (double)pl->B * rows_out * nparts * Cin;
  if (skip_mask() & 16) return KVQ_OK;
  Bracket br(pl, st, KVQ_K_LAYERNORM, of32 ? 1 : 0, 0.0, elems * ((x16 ? 2.0 : 4.0) + (of32 ? 4.0 : 2.0)));
  return layernorm_rows_stream(x, x16 ? 1 : 0, map, nparts, pl->B, rows_in, rows_out, Cin, g, b, 1e-5f, obf, pl->dtype, of32, st);
}

}  // namespace kvq

#define KVQ_TRY(expr)      \
  do {                     \
    int _rc = (expr);      \
    if (_rc) return _rc;   \
  } while (0)

#define KVQ_TRY_UNLESS(bits, expr) \
  do {                             \
    if (!(skip_mask() & (bits))) KVQ_TRY(expr); \
  } while (0)

extern "C" int kvq_swin3d_set_taps(KvqSwinPlan* pl, float* const* taps) {
  KVQ_REQUIRE(pl, KVQ_ERR_NULL, "kvq_swin3d_set_taps: NULL plan");
  pl->taps.clear();
  if (taps) pl->taps.assign(taps, taps + pl->cfg.num_stages + 1);
  return KVQ_OK;
}

extern "C" int kvq_swin3d_tap_dims(const KvqSwinPlan* pl, int index, int32_t out4[4]) {
  KVQ_REQUIRE(pl && out4, KVQ_ERR_NULL, "kvq_swin3d_tap_dims: NULL");
  KVQ_REQUIRE(index >= 0 && index <= pl->cfg.num_stages, KVQ_ERR_SHAPE, "kvq_swin3d_tap_dims: index %d", index);
  if (index == 0) {
    out4[0] = pl->cfg.embed_dim; out4[1] = pl->D0; out4[2] = pl->H0; out4[3] = pl->W0;
    return KVQ_OK;
  }
  const kvq::StageGeom& g = pl->st[index - 1];
  const bool merged = index - 1 < pl->cfg.num_stages - 1;
  out4[0] = merged ? 2 * g.C : g.C; out4[1] = merged ? g.Dn : g.D; out4[2] = merged ? g.Hn : g.H; out4[3] = merged ? g.Wn : g.W;
  return KVQ_OK;
}

// stages stage_lo .. stage_hi of the trunk.  stage_lo == 0: starts from the clip x (patch embedding first); otherwise from
// the residual stream `io` (fp32 channels-last (B, D, H_lo, W_lo, C_lo), copied into the workspace).  Afterwards `io`, when
// given, receives the residual stream behind stage_hi (incl. its PatchMerging), and — stage_hi being the last stage — `feat`,
// when given, the final LayerNorm of it.
static int swin_run(const KvqSwinPlan* cpl, const KvqSwinWeights* w, const float* x, const KvqFragmentSource* frag, int stage_lo, int stage_hi, float* io,
                    float* feat, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(cpl && w && workspace, KVQ_ERR_NULL, "kvq_swin3d_forward: NULL pointer");
  KVQ_REQUIRE(w->blocks && w->embed_w && w->embed_b && w->norm_w && w->norm_b, KVQ_ERR_NULL,
              "kvq_swin3d_forward: incomplete weights");
  KvqSwinPlan* pl = const_cast<KvqSwinPlan*>(cpl);   // profiling state only
  KVQ_REQUIRE(stage_lo >= 0 && stage_lo <= stage_hi && stage_hi < pl->cfg.num_stages, KVQ_ERR_SHAPE,
              "kvq_swin3d_forward: stages %d..%d of %d", stage_lo, stage_hi, pl->cfg.num_stages);
  KVQ_REQUIRE(stage_lo == 0 ? (x != nullptr || frag != nullptr) : io != nullptr, KVQ_ERR_NULL, "kvq_swin3d_forward: no input for stage %d", stage_lo);
  KVQ_REQUIRE(io || (feat && stage_hi == pl->cfg.num_stages - 1), KVQ_ERR_NULL, "kvq_swin3d_forward: no output buffer");
  KVQ_REQUIRE(workspace_bytes >= pl->ws_bytes, KVQ_ERR_WORKSPACE, "kvq_swin3d_forward: workspace %zu < %zu bytes",
              workspace_bytes, pl->ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  const KvqSwinCfg& cfg = pl->cfg;
  const int B = pl->B;
  unsigned char* ws = (unsigned char*)workspace;
  pl->run_ws = ws;
  float* xa = (float*)(ws + pl->off_x0);
  float* xb = (float*)(ws + pl->off_x1);
  uint16_t* bln = (uint16_t*)(ws + pl->off_ln);
  uint16_t* bbig = (uint16_t*)(ws + pl->off_big);
  uint16_t* bo = (uint16_t*)(ws + pl->off_o);
  pl->ev_used = pl->profile ? pl->ev_used : 0;

  // ---- the residual stream of a stage in fp16 (round 6) --------------------------------------------------------------------------------
  // x is written once and read once per block (8 C bytes per token in fp32): 41 % of the step's HBM traffic.  Where EVERY producer and
  // consumer of a stage's stream is one of the token-per-lane launches (embedding / fused merge -> fused tails -> fused merge) the stream
  // lives in fp16 — 2.9e-6 on the score of a 32 x 224 x 224 clip in an fp32 emulation, two orders below the 16-bit MFMA operands' own
  // 3.6e-4 (tools/diag/resid16_probe.py): the stream carries 11 bits where every GEMM input is rounded to 8 or 11 anyway.  Decided by
  // geometry and weights only (never by the batch); forwards with feature taps keep fp32; a stage-split forward enters and leaves in fp32.
  // Consumers that take an fp16 stream: the fused tails of every width (padded partitions too), the fused merge, every LayerNorm launch
  // (a first block's norm1, the gather-LayerNorm of an un-fused merge);
  // the last stage keeps fp32 (its stream comes out of a GEMM epilogue and feeds the final LayerNorm and the fp32 feature output).
  // KVQ_RESID16=0: fp32 everywhere (rounds 1-5).
  static const bool resid16_on = !(getenv("KVQ_RESID16") && atoi(getenv("KVQ_RESID16")) == 0);
  static const int resid16_maxc = getenv("KVQ_RESID16_MAXC") ? atoi(getenv("KVQ_RESID16_MAXC")) : 1 << 30;      // 192: the token-per-lane stages only (A/B)
  static const int tail_maxc_x = getenv("KVQ_TAIL_MAXC") ? atoi(getenv("KVQ_TAIL_MAXC")) : 1 << 30;
  static const int merge_maxc_x = getenv("KVQ_MERGE_MAXC") ? atoi(getenv("KVQ_MERGE_MAXC")) : 192;
  bool any_tap = false;
  for (float* t : pl->taps) any_tap = any_tap || t != nullptr;
  auto fused_merge = [&](int i) -> bool {                      // stage i -> i + 1 is the one-launch merge (csrc/merge.hip)
    return i >= 0 && i < cfg.num_stages - 1 && w->merges[i].merge_pack && kvq_patch_merge_supported(pl->st[i].C) && pl->st[i].C <= merge_maxc_x;
  };
  const bool embed_fused = w->embed_pack && kvq_patch_embed_supported(cfg.in_chans, cfg.patch[0], cfg.patch[1], cfg.patch[2], cfg.embed_dim, pl->T, pl->H, pl->W);
  bool x16[KVQ_MAX_STAGES] = {false, false, false, false};
  if (resid16_on && !any_tap) {
    // a stage-split call (KSVQE: stages 0-1, modulation, stage 2, modulation, stage 3) takes its entry stream from `io` in fp32 and hands the
    // stream behind its last merge back in fp32: the stages strictly inside the call follow the same rule as in a whole forward
    int blk0 = 0;
    for (int i = 0; i < cfg.num_stages - 1 && i <= stage_hi; blk0 += pl->st[i].depth, ++i) {
      if (i < stage_lo) continue;
      const StageGeom& g = pl->st[i];
      // the producer writes fp16: the embedding, the fused merge, or — fp16 OPERANDS only: its 16-bit store is then the stream's type — the
      // reduction GEMM of an un-fused merge (Swin-B's stage 2, 18 of its 24 blocks, sits behind one)
      const bool produced16 = i == 0 ? (stage_lo == 0 && embed_fused) : (i > stage_lo && (fused_merge(i - 1) || pl->dtype == KVQ_DT_FP16));
      bool ok = g.C <= tail_maxc_x && g.C <= resid16_maxc && produced16;
      for (int b = 0; ok && b < g.depth; ++b) {
        const KvqSwinBlockW& bw = w->blocks[blk0 + b];
        const int par = (b & 1) && g.shifted_any ? 1 : 0;
        ok = bw.tail_pack && kvq_block_tail_supported(g.C, cfg.mlp_ratio * g.C) && bw.norm1_w && bw.norm1_b && g.d_dst[par];
      }
      x16[i] = ok;
    }
  }
  bool x16_cur = false;                                        // the stream `cur` holds right now
  auto f32_only = [&](const char* what) -> int {               // a launch that reads or writes the stream as fp32 met an fp16 one: a bug above, never garbage
    KVQ_REQUIRE(!x16_cur, KVQ_ERR_UNSUPPORTED, "kvq_swin3d_forward: internal: %s on an fp16 residual stream", what);
    return KVQ_OK;
  };

  // ---- PatchEmbed3D (swin_backbone.py:715-733): one fused launch, or im2col -> GEMM(+bias) -> LayerNorm ----
  const int L0 = pl->D0 * pl->H0 * pl->W0, E = cfg.embed_dim;
  float* cur = xa;
  float* oth = xb;
  bool first_ln1_ready = false;
  if (stage_lo > 0) {
    const StageGeom& g0 = pl->st[stage_lo];
    KVQ_CHECK_HIP(hipMemcpyAsync(xa, io, (size_t)B * g0.L * g0.C * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else if (w->embed_pack && kvq_patch_embed_supported(cfg.in_chans, cfg.patch[0], cfg.patch[1], cfg.patch[2], E, pl->T, pl->H, pl->W)) {
    KvqPatchEmbedArgs ea{};
    ea.x = x; ea.frag = frag; ea.B = B; ea.in_chans = cfg.in_chans; ea.T = pl->T; ea.H = pl->H; ea.W = pl->W;
    ea.pd = cfg.patch[0]; ea.ph = cfg.patch[1]; ea.pw = cfg.patch[2]; ea.embed_dim = E; ea.pack = w->embed_pack;
    ea.has_norm = w->embed_ln_w ? 1 : 0; ea.out = xa; ea.eps = 1e-5f; ea.dtype = pl->dtype; ea.out_f16 = x16[0] ? 1 : 0;
    x16_cur = x16[0];
    const StageGeom& g0 = pl->st[0];
    if (g0.Lp == g0.L && g0.d_dst[0] && w->blocks[0].norm1_w && w->blocks[0].norm1_b) {     // + norm1 / partition of the first block
      ea.next_norm_w = w->blocks[0].norm1_w; ea.next_norm_b = w->blocks[0].norm1_b; ea.next_dst = g0.d_dst[0];
      ea.next_ln = bln; ea.next_rows = g0.Lp;
      first_ln1_ready = true;
    }
    const double px = (double)B * L0 * pl->K0;
    Bracket br(pl, st, KVQ_K_EMBED, (first_ln1_ready ? 1 : 0) + (frag ? 2 : 0), 2.0 * B * L0 * (double)E * pl->K0,
               px * (frag ? 1.0 : 4.0) + (double)B * L0 * E * ((x16[0] ? 2.0 : 4.0) + (first_ln1_ready ? 2.0 : 0.0)));
    KVQ_TRY_UNLESS(64, kvq_patch_embed(&ea, st));
  } else {
    KVQ_REQUIRE(!frag, KVQ_ERR_UNSUPPORTED, "kvq_swin3d_forward_fragments: this plan does not take the fused patch-embedding launch");
    {
      const double px = (double)B * pl->D0 * pl->H0 * pl->W0 * pl->K0;
      Bracket br(pl, st, KVQ_K_IM2COL, 0, 0.0, px * 6.0);
      KVQ_TRY(kvq_patch_im2col(x, B, cfg.in_chans, pl->T, pl->H, pl->W, cfg.patch[0], cfg.patch[1], cfg.patch[2], pl->dtype,
                               bbig, st));
    }
    KVQ_TRY(gemm(pl, st, KVQ_K_GEMM_EMBED, bbig, w->embed_w, w->embed_b, B * L0, E, pl->K0, KVQ_EPI_STORE_F32, nullptr,
                 xb));
    if (w->embed_ln_w) {
      KVQ_TRY(ln(pl, st, xb, nullptr, 1, L0, L0, E, w->embed_ln_w, w->embed_ln_b, nullptr, xa));
    } else {
      cur = xb; oth = xa;
    }
  }
  auto tap = [&](int idx, size_t elems) -> int {          // feats[idx] of the reference's forward (multi / layer)
    if (pl->taps.empty() || !pl->taps[idx]) return KVQ_OK;
    KVQ_TRY(f32_only("a feature tap"));
    KVQ_CHECK_HIP(hipMemcpyAsync(pl->taps[idx], cur, elems * sizeof(float), hipMemcpyDeviceToDevice, st));
    return KVQ_OK;
  };
  if (stage_lo == 0) KVQ_TRY(tap(0, (size_t)B * L0 * E));

  int blk = 0;
  for (int i = 0; i < stage_lo; ++i) blk += pl->st[i].depth;
  size_t out_elems = 0;           // size of the residual stream behind the last stage run
  bool merged_ln1_ready = false;  // the fused merge launch of the previous stage wrote the first norm1 rows of this one
  for (int i = stage_lo; i <= stage_hi; ++i) {
    const StageGeom& g = pl->st[i];
    const int C = g.C, M = B * g.Lp, ML = B * g.L;
    bool ln1_ready = (i == 0 && first_ln1_ready) || merged_ln1_ready;   // the producer (embed / previous tail / merge) already wrote this block's norm1 rows
    bool qkv_ready = false;                                             // the previous block's tail already wrote this block's q | k | v
    merged_ln1_ready = false;
    if (i == stage_lo && i > 0) cur = xa, oth = xb;
    for (int b = 0; b < g.depth; ++b, ++blk) {
      const KvqSwinBlockW& bw = w->blocks[blk];
      KVQ_REQUIRE(bw.norm1_w && bw.rpb_table && bw.qkv_w && bw.proj_w && bw.fc1_w && bw.fc2_w, KVQ_ERR_NULL,
                  "kvq_swin3d_forward: block %d weights incomplete", blk);
      const int par = (b & 1) && g.shifted_any ? 1 : 0;
      // Narrow stages (C <= 128; measured at C = 192: the fused launch loses 19 us to GEMM + attention — the rows would be read twice
      // through the CU's 64 B/clk load path): the qkv GEMM is an HBM-bound launch whose 6 C bytes per row the attention launch reads right
      // back): the attention workgroup of a (window, head) computes its own q | k | v from the norm1 rows (attn.hip,
      // fused_qkv_prologue).  Un-padded partitions on the dense bias only.  Measured (bench.py --legs c2,no_sampler, two runs each,
      // same box): 300.4 -> 314.6 videos/s with the sampler in the step, 315.5 -> 330.1 without; stage-0 launch 133.5 -> 117.3 us.
      const bool fuse_qkv = bw.bias_dense && bw.qkv_b && g.Lp == g.L && C == 96 && g.N <= 400;      // by geometry only, never by batch
      // attn32.hip (bias image) keeps its scores in log2 units: q is scaled by head_dim^-0.5 * log2(e) there; the gather path takes
      // head_dim^-0.5.  Which one a block takes depends on its weights and geometry only, never on the batch.
      const float qs = bw.bias_dense ? kQScaleLog2 : kQScale;
      // norm1 + pad + roll + window_partition
      if (g.Lp != g.L && bw.qkv_b && qkv_ready) {
        // padded partition whose q | k | v rows the previous block's tail wrote (window rows of the tokens; the padding rows are the
        // attention launch's: pad_mask)
      } else if (g.Lp != g.L && bw.qkv_b) {
        // padded partition: norm1 in TOKEN order (written by the previous block's tail when there is one), qkv over the tokens only
        // (rows scattered to their window rows by the epilogue); the padding rows' q | k | v = qkv(0) = bias
        if (!ln1_ready) KVQ_TRY(ln(pl, st, cur, nullptr, 1, g.L, g.L, C, bw.norm1_w, bw.norm1_b, bln, nullptr, x16_cur));
        KVQ_TRY(gemm(pl, st, KVQ_K_GEMM_QKV, bln, bw.qkv_w, bw.qkv_b, ML, 3 * C, C, KVQ_EPI_QKV_BF16, bbig, nullptr, g.nH,
                     qs, g.d_dst[par], g.L, g.Lp));
        // the padding rows' q | k | v = qkv(0) = bias: attention32 writes them into its own K | V images (pad_mask); the gather path
        // reads them from the buffer
        if (!(bw.bias_dense && g.d_padmask[par]))
          KVQ_TRY(kvq_qkv_fill_pad(bbig, bw.qkv_b, g.d_pad[par], g.Lp - g.L, B, g.Lp, g.nH, qs, pl->dtype, st));
      } else if (!qkv_ready) {
        if (!ln1_ready) KVQ_TRY(ln(pl, st, cur, g.d_src[par], 1, g.L, g.Lp, C, bw.norm1_w, bw.norm1_b, bln, nullptr, x16_cur));
        if (!fuse_qkv)
          KVQ_TRY_UNLESS(8 | (i == cfg.num_stages - 1 ? 32 : 0), gemm(pl, st, KVQ_K_GEMM_QKV, bln, bw.qkv_w, bw.qkv_b, M, 3 * C, C, KVQ_EPI_QKV_BF16, bbig, nullptr, g.nH, qs));
      }
      ln1_ready = false;
      qkv_ready = false;
      if (bw.bias_dense) {
        // + the dense bias once per step: 4 B per score of every (window, head)
        Bracket br(pl, st, KVQ_K_ATTN, 4 + par, 4.0 * M * g.N * C + (fuse_qkv ? 6.0 * M * C * C : 0.0),
                   (fuse_qkv ? 2.0 * 2.0 * M * C + 6.0 * C * C : 2.0 * 4.0 * M * C) + (double)kvq_swin3d_bias_dense_bytes(pl, blk));
        KvqAttnDenseArgs aa{};
        aa.qkv = bbig; aa.bias_dense = bw.bias_dense; aa.n_types = bias_types(g, par); aa.BW = B * g.nW; aa.nW = g.nW; aa.N = g.N;
        aa.num_heads = g.nH; aa.dtype = pl->dtype; aa.out = bo; aa.tile_skip = (const uint32_t*)g.d_skip[par];
        // shifted blocks of the (8,7,7) window with a half-window depth shift: the last window slab along D is depth-split
        const int slabs = g.Dp / g.ws[0];
        aa.dsplit_from = (par == 1 && g.N == 392 && g.ws[0] == 8 && g.ws[1] == 7 && g.ws[2] == 7 && g.ss[0] == 4 && slabs >= 1)
                             ? g.nW - g.nW / slabs : -1;
        if (fuse_qkv) { aa.x_ln = bln; aa.w_qkv = bw.qkv_w; aa.b_qkv = bw.qkv_b; aa.q_scale = qs; }
        if (g.Lp != g.L && bw.qkv_b && g.d_padmask[par]) { aa.pad_mask = (const uint32_t*)g.d_padmask[par]; aa.b_qkv = bw.qkv_b; }
        KVQ_TRY_UNLESS(1 | (i == cfg.num_stages - 1 ? 32 : 0), kvq_window_attention32(&aa, st));
      } else {
        // 4*Lp*N*C flops per block; bytes: q,k,v in + o out (16-bit)
        Bracket br(pl, st, KVQ_K_ATTN, (cfg.frag_bias[i] ? 2 : 0) + par, 4.0 * M * g.N * C, 2.0 * 4.0 * M * C);
        KVQ_TRY(kvq_window_attention(bbig, g.d_tok[par], bw.rpb_table, cfg.frag_bias[i] ? bw.fpb_table : nullptr,
                                     bw.bias_pack,
                                     pl->table_len, pl->center, B * g.nW, g.nW, g.N, g.nH, par, pl->dtype, bo,
                                     st));
      }
      const int hidden = cfg.mlp_ratio * C;
      // KVQ_TAIL_MAXC: widest stage that takes the fused tail launch (wider ones run proj / norm2 / fc1 / fc2 as a GEMM chain)
      static const int tail_maxc = getenv("KVQ_TAIL_MAXC") ? atoi(getenv("KVQ_TAIL_MAXC")) : 1 << 30;
      if (bw.tail_pack && C <= tail_maxc && kvq_block_tail_supported(C, hidden)) {
        // proj + window_reverse + roll back + crop + residual + norm2 + Mlp + residual [+ the next block's norm1]
        KvqBlockTailArgs ta{};
        ta.attn = bo; ta.x = cur; ta.scatter_map = g.d_src[par]; ta.map_rows = g.Lp; ta.out_rows = g.L;
        ta.M = M; ta.C = C; ta.hidden = hidden; ta.pack = bw.tail_pack; ta.eps = 1e-5f; ta.dtype = pl->dtype;
        ta.x_f16 = x16_cur ? 1 : 0;
        const int npar = ((b + 1) & 1) && g.shifted_any ? 1 : 0;
        if (g.Lp != g.L) ta.attn_gather = g.d_dst[par];      // padded windows: walk the tokens, not the window rows
        // the next block's norm1 rows in ITS window order (un-padded partitions: token -> window row is a bijection).  Padded
        // partitions could take them in token order (through an identity map; measured on C5: 19 LayerNorm launches / 0.51 ms saved,
        // but the emitting form of the C = 512 tail costs +25 us per launch (it spills): 12.9 vs 13.0 ms serial, 20.9 vs 21.2-22.0
        // videos/s with two steps in flight) — not taken.
        // Round 5: a padded partition takes the next block's q | k | v from this launch too (token -> window row of the NEXT partition,
        // rows of Lp per clip; the padding rows are written by the attention launch, pad_mask) — the LayerNorm launch, the qkv GEMM and
        // the norm1 round trip of every block of Swin-B at 64 x 256 x 256 are gone (C5 23.8 -> 25.05 videos/s, +5.2 %, same box alternating: profiles/r05_padded_qkv_ab.txt);
        // the C = 512 tail in hidden chunks of 128 emits without the spills the comment above met.  KVQ_TAIL_QKV_PADDED=0: the old sequence.
        static const bool tail_qkv_padded = !(getenv("KVQ_TAIL_QKV_PADDED") && atoi(getenv("KVQ_TAIL_QKV_PADDED")) == 0);
        const bool padded_qkv = g.Lp != g.L && tail_qkv_padded && b + 1 < g.depth && g.d_dst[npar] && g.d_padmask[npar] && w->blocks[blk + 1].bias_dense;
        const int32_t* nmap = (g.Lp == g.L || padded_qkv) ? g.d_dst[npar] : nullptr;
        if (b + 1 < g.depth && nmap) {
          const KvqSwinBlockW& nb = w->blocks[blk + 1];
          KVQ_REQUIRE(nb.norm1_w && nb.norm1_b, KVQ_ERR_NULL, "kvq_swin3d_forward: block %d norm1 missing", blk + 1);
          ta.next_norm_w = nb.norm1_w; ta.next_norm_b = nb.norm1_b; ta.next_dst = nmap;
          ta.next_rows = g.Lp;
          // the next block's q | k | v straight from this launch (C = 128 / 192 / 256 / 384 / 512, un-padded partitions, the image path's q scale):
          // no norm1 rows, no qkv GEMM launch.  By geometry and weights only, never by batch.  KVQ_TAIL_QKV=0: rounds 1-4's sequence.
          static const bool tail_qkv = !(getenv("KVQ_TAIL_QKV") && atoi(getenv("KVQ_TAIL_QKV")) == 0);
          const bool next_fuses = nb.bias_dense && nb.qkv_b && g.Lp == g.L && C == 96 && g.N <= 400;      // its attention launch projects q | k | v itself
          if (tail_qkv && !next_fuses && nb.qkv_pack && nb.qkv_b && nb.bias_dense && (g.Lp == g.L || padded_qkv) && kvq_block_tail_qkv_pack_bytes(C, hidden) > 0) {
            ta.next_qkv_pack = nb.qkv_pack; ta.next_qkv_b = nb.qkv_b; ta.qkv_out = bbig; ta.q_scale = kQScaleLog2; ta.num_heads = g.nH;
            qkv_ready = true;
          } else if (g.Lp == g.L) {
            ta.next_ln = bln;
            ln1_ready = true;
          } else {                      // padded partition without the emission: the tail writes x only
            ta.next_norm_w = nullptr; ta.next_norm_b = nullptr; ta.next_dst = nullptr; ta.next_rows = 0;
          }
        }
        Bracket br(pl, st, KVQ_K_TAIL, kvq::tailmm_geometry_code(C, hidden) * 1000 + (C / 32) * 10 + (ln1_ready ? 1 : 0) + (qkv_ready ? 2 : 0), 2.0 * M * C * C + 4.0 * (double)ML * C * hidden + (qkv_ready ? 6.0 * M * C * C : 0.0),
                   (double)M * C * 2.0 + (double)ML * C * ((x16_cur ? 4.0 : 8.0) + (ln1_ready ? 2.0 : 0.0) + (qkv_ready ? 6.0 : 0.0)));
        KVQ_TRY_UNLESS(C <= 192 ? 2 : 4, kvq_block_tail(&ta, st));
        continue;
      }
      KVQ_TRY(f32_only("the proj / MLP GEMM chain"));
      // proj + window_reverse + roll back + crop + residual.  Padded partition: over the tokens (A rows gathered through token ->
      // window row, output in place in token order) instead of over the window rows with the padding rows dropped in the epilogue
      if (g.Lp != g.L) {
        KvqGemmArgs pa{};
        pa.A = bo; pa.W = bw.proj_w; pa.bias = bw.proj_b; pa.M = ML; pa.N = C; pa.K = C; pa.epilogue = KVQ_EPI_RESID_F32; pa.out_f32 = cur;
        pa.dtype = pl->dtype; pa.a_gather = g.d_dst[par]; pa.a_rows = g.L; pa.a_phys_rows = g.Lp;
        Bracket br(pl, st, KVQ_K_GEMM_PROJ, gemm_variant(ML, C, C) * 10 + KVQ_EPI_RESID_F32, 2.0 * M * C * C,
                   2.0 * ((double)ML * C + (double)C * C) + 8.0 * ML * C);
        KVQ_TRY(kvq_gemm_bf16(&pa, st));
      } else {
        KVQ_TRY_UNLESS(i == cfg.num_stages - 1 ? 32 : 0, gemm(pl, st, KVQ_K_GEMM_PROJ, bo, bw.proj_w, bw.proj_b, M, C, C, KVQ_EPI_RESID_F32, nullptr, cur, 0, 1.f,
                     g.d_src[par], g.Lp, g.L));
      }
      // norm2 + fc1 + GELU + fc2 + residual
      KVQ_TRY(ln(pl, st, cur, nullptr, 1, g.L, g.L, C, bw.norm2_w, bw.norm2_b, bln, nullptr));
      KVQ_TRY_UNLESS(i == cfg.num_stages - 1 ? 32 : 0, gemm(pl, st, KVQ_K_GEMM_FC1, bln, bw.fc1_w, bw.fc1_b, ML, cfg.mlp_ratio * C, C, KVQ_EPI_GELU_BF16, bbig,
                   nullptr));
      KVQ_TRY_UNLESS(i == cfg.num_stages - 1 ? 32 : 0, gemm(pl, st, KVQ_K_GEMM_FC2, bbig, bw.fc2_w, bw.fc2_b, ML, C, cfg.mlp_ratio * C, KVQ_EPI_RESID_F32,
                   nullptr, cur));
    }
    if (i < cfg.num_stages - 1) {
      const KvqSwinMergeW& mw = w->merges[i];
      KVQ_REQUIRE(mw.norm_w && mw.norm_b && mw.red_w, KVQ_ERR_NULL, "kvq_swin3d_forward: merge %d weights missing", i);
      const int Ln = g.Dn * g.Hn * g.Wn;
      static const int merge_maxc = getenv("KVQ_MERGE_MAXC") ? atoi(getenv("KVQ_MERGE_MAXC")) : 192;      // 128 = rounds 4's gate (the C = 192 merge as three launches)
      if (mw.merge_pack && kvq_patch_merge_supported(C) && C <= merge_maxc) {
        // concat + LayerNorm(4C) + reduction [+ the next stage's first norm1 in its window order] as one launch (csrc/merge.hip).
        // C = 96: 37.5 us against 26.3 + 25.6 + 15.9 (Swin-T, 4 clips); C = 128: +0.5-1 % on C5.  C = 192 exists and is tested, but
        // the 576 KB matrix streams through LDS for 98 workgroups of one wave per SIMD: 66.9 us against 16.2 + 24.8 + 15.3 alone on the chip;
        // taken since round 5 (merge_maxc = 192): level on the 4-lane line with two launches fewer (profiles/r05_lane_experiments.txt)
        KvqPatchMergeArgs ma{};
        ma.x = cur; ma.merge_map = g.d_merge; ma.B = B; ma.L = g.L; ma.Ln = Ln; ma.C = C; ma.pack = mw.merge_pack; ma.out = oth;
        ma.eps = 1e-5f; ma.dtype = pl->dtype; ma.x_f16 = x16_cur ? 1 : 0; ma.out_f16 = x16[i + 1] ? 1 : 0;
        if (i + 1 <= stage_hi) {
          const StageGeom& gn = pl->st[i + 1];
          const KvqSwinBlockW& nb = w->blocks[blk];           // blk: the first block of stage i + 1
          if (gn.Lp == gn.L && gn.d_dst[0] && nb.norm1_w && nb.norm1_b) {
            ma.next_norm_w = nb.norm1_w; ma.next_norm_b = nb.norm1_b; ma.next_dst = gn.d_dst[0]; ma.next_ln = bln; ma.next_rows = gn.Lp;
            merged_ln1_ready = true;
          }
        }
        Bracket br(pl, st, KVQ_K_MERGE, merged_ln1_ready ? 1 : 0, 2.0 * B * Ln * (double)(2 * C) * (4 * C),
                   (double)B * Ln * 4 * C * (x16_cur ? 2.0 : 4.0) + (double)B * Ln * 2 * C * ((x16[i + 1] ? 2.0 : 4.0) + (merged_ln1_ready ? 2.0 : 0.0)));
        KVQ_TRY_UNLESS(128, kvq_patch_merge(&ma, st));
        x16_cur = x16[i + 1];
      } else {
        KVQ_TRY(ln(pl, st, cur, g.d_merge, 4, g.L, Ln, C, mw.norm_w, mw.norm_b, bln, nullptr, x16_cur));
        x16_cur = x16[i + 1];                                  // the reduction GEMM writes the next stage's stream: fp32, or fp16 rows (fp16 operands)
        if (x16_cur) KVQ_TRY_UNLESS(128, gemm(pl, st, KVQ_K_GEMM_MERGE, bln, mw.red_w, nullptr, B * Ln, 2 * C, 4 * C, KVQ_EPI_BIAS_BF16, reinterpret_cast<uint16_t*>(oth), nullptr));
        else KVQ_TRY_UNLESS(128, gemm(pl, st, KVQ_K_GEMM_MERGE, bln, mw.red_w, nullptr, B * Ln, 2 * C, 4 * C, KVQ_EPI_STORE_F32, nullptr,
                     oth));
      }
      float* t = cur; cur = oth; oth = t;
      out_elems = (size_t)B * Ln * 2 * C;
    } else {
      out_elems = (size_t)ML * C;
    }
    KVQ_TRY(tap(i + 1, out_elems));
  }
  if (io) {
    KVQ_TRY(f32_only("the stage-split output copy"));
    KVQ_CHECK_HIP(hipMemcpyAsync(io, cur, out_elems * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  if (feat && stage_hi == cfg.num_stages - 1) {
    KVQ_TRY(f32_only("the final LayerNorm"));
    const StageGeom& gl = pl->st.back();
    KVQ_TRY(ln(pl, st, cur, nullptr, 1, gl.L, gl.L, gl.C, w->norm_w, w->norm_b, nullptr, feat));
  }
  return KVQ_OK;
}

extern "C" int kvq_swin3d_forward(const KvqSwinPlan* plan, const KvqSwinWeights* w, const float* x, float* feat, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  KVQ_REQUIRE(plan && x && feat, KVQ_ERR_NULL, "kvq_swin3d_forward: NULL pointer");
  return swin_run(plan, w, x, nullptr, 0, plan->cfg.num_stages - 1, nullptr, feat, workspace, workspace_bytes, stream);
}

extern "C" int kvq_swin3d_forward_fragments(const KvqSwinPlan* plan, const KvqSwinWeights* w, const KvqFragmentSource* src, float* feat,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  KVQ_REQUIRE(plan && src && feat, KVQ_ERR_NULL, "kvq_swin3d_forward_fragments: NULL pointer");
  KVQ_REQUIRE(kvq_patch_embed_fragments_supported(src, plan->B, plan->cfg.in_chans, plan->cfg.patch[0], plan->T, plan->H, plan->W),
              KVQ_ERR_UNSUPPORTED, "kvq_swin3d_forward_fragments: the source does not fit the fused read of a %dx%dx%dx%d batch",
              plan->B, plan->T, plan->H, plan->W);
  return swin_run(plan, w, nullptr, src, 0, plan->cfg.num_stages - 1, nullptr, feat, workspace, workspace_bytes, stream);
}

extern "C" int kvq_swin3d_forward_stages(const KvqSwinPlan* plan, const KvqSwinWeights* w, const float* x, int stage_lo,
                                         int stage_hi, float* io, float* feat, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  return swin_run(plan, w, x, nullptr, stage_lo, stage_hi, io, feat, workspace, workspace_bytes, stream);
}
