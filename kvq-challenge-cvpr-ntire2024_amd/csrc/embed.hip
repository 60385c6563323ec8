// PatchEmbed3D as ONE launch (gfx950): strided Conv3d k = s = (pd,4,4) + bias + LayerNorm(E)
// (swin_backbone.py:715-733) [+ norm1 + pad/roll/window_partition of the first block, :416-449].
//
// Token-per-lane, like csrc/tail.hip: Out^T[E][32 tokens] = W[E][K] * patch^T[K][32 tokens] with the weights as the
// MFMA A operand (fragment-major in LDS) and the tokens as the 32 columns of v_mfma_f32_32x32x16.  The B operand is
// read STRAIGHT from the clip: k = (c, kd, kh, kw) with ph*pw = 16, so k-step s is the (c, kd) plane and lane half
// h takes patch rows 2h, 2h+1 — two 16-B loads (4 fp32 pixels each) that are contiguous across the 32 lanes of a
// token row.  No im2col buffer, no separate LayerNorm launch: the clip is read once (4 B/pixel) and the residual
// stream written once, instead of im2col (write 2 B/px) + GEMM (read 2, write 4·E/K) + LayerNorm (read/write 4·E/K).
//
// FRAG variant: the clip is never materialised.  The B operand comes straight out of the decoded uint8 frames through
// the fragment sampler's patch origins (get_spatial_fragments, fusion_datasets.py:22-121: a 4 x 4 patch lies inside one
// fs_h x fs_w mini-patch, so its rows are two 4-byte runs of a source row) and is normalised in registers with the same
// IEEE fp32 (v - mean) / std as kvq_fragment_gather — the operands are bit-identical to the two-launch sequence, which
// wrote the fp32 clip (4 B/px) and read it back (4 B/px).
#include <stdlib.h>

#include "common.hpp"

namespace kvq {

struct FragSrc {
  const uint8_t* video[KVQ_FRAG_MAX_CLIPS];   // clip b: (Cin, T, Hs, Ws)
  const int32_t* hoff[KVQ_FRAG_MAX_CLIPS];    // clip b: [Fh][Fw][T / aligned] patch origins
  const int32_t* woff[KVQ_FRAG_MAX_CLIPS];
  long chan_stride;
  int Hs, Ws, Fw, fsh, fsw, aligned;
  float mean[4], std[4];
  const void* const* table;   // KvqFragmentSource.indirect: video[16] | hoff[16] | woff[16] in device memory, or nullptr
};

struct EmbedParams {
  const float* x;            // (B, Cin, T, H, W); unused by the FRAG variant
  FragSrc frag;
  int B, Cin, T, H, W, pd, D0, H0, W0;
  const unsigned char* pack; // kvq_patch_embed_pack image
  int has_ln;                // patch_embed.norm present
  float* out;                // [B*L0][E] fp32
  int out16;                 // round 6: the residual stream leaves as fp16 rows of 2 E bytes (same pointer)
  const float* nn_w;         // first block's norm1 (EMIT)
  const float* nn_b;
  const int32_t* next_dst;   // token -> window row
  uint16_t* next_ln;         // [B*next_rows][E]
  int next_rows;
  float eps;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

// image: CM panels [KS frag rows][64 lanes][8 x 16-bit] with W[32i+m][16s + 8h + e] (k in Conv3d weight order
// c, kd, kh, kw), then fp32 [bias E][ln_w E][ln_b E]
__global__ void embed_pack_kernel(const uint16_t* w, const float* bias, const float* lnw, const float* lnb, int E, int K,
                                  unsigned char* out, long n_chunks) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KS = K / 16;
  if (g < n_chunks) {
    const int panel = (int)(g / (KS * 64)), rem = (int)(g % (KS * 64));
    const int s = rem >> 6, lane = rem & 63, m = lane & 31, h = lane >> 5;
    uint16_t* o = reinterpret_cast<uint16_t*>(out + g * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = w[(size_t)(32 * panel + m) * K + 16 * s + 8 * h + e];
  } else if (g < n_chunks + 3 * E) {
    const int q = (int)(g - n_chunks);
    const float v = q < E ? bias[q] : (q < 2 * E ? (lnw ? lnw[q - E] : 1.f) : (lnb ? lnb[q - 2 * E] : 0.f));
    reinterpret_cast<float*>(out + n_chunks * 16)[q] = v;
  }
}

// LDS: [weights CM*KS KB][bias | ln_w | ln_b, whole KB][next norm1 gamma | beta][FRAG: 4 x 256 table][STAGED: 4 waves x 32 rows]
__host__ __device__ constexpr int embed_stage_off(int CM, int KS) { return CM * KS * 1024 + 5 * 32 * CM * 4 + 1024 + 4 * 256 * 4; }
// the x0 rows leave through an LDS tile at E = 96 (76 KB per workgroup: two per CU, which measures like five; hipEvent-bracketed
// launch: 52.1 -> 49.8 us reading through the sampler, 48.4 -> 43.3 reading the fp32 clip); at E = 128 the tile would leave one
// workgroup per CU — the accumulator-layout stores stay
// Round 5: OFF.  The tile makes the launch 2-5 us faster ALONE (above) and costs the 4-lane bench line 0.8 % (380.9 / 380.4 -> 384.8 / 382.8
// videos/s without it, same box, alternating): a memory-bound launch whose workgroups hold 76 KB of LDS keeps the other lanes' workgroups
// off its CUs; at 25 KB three of them leave room for an attention or tail workgroup.  -DKVQ_EMBED_STAGED=1 builds the tile form.
#ifndef KVQ_EMBED_STAGED
#define KVQ_EMBED_STAGED 0
#endif
__host__ __device__ constexpr bool embed_staged(int CM, bool) { return CM == 3 && KVQ_EMBED_STAGED; }

template <typename E_, int CM, int KS, bool EMIT, bool FRAG>
__global__ __launch_bounds__(256, embed_staged(CM, FRAG) ? 2 : 3) void patch_embed_kernel(EmbedParams p) {
  fp16_saturate_mode();
  constexpr int E = 32 * CM, WBYTES = CM * KS * 1024;
  constexpr bool STAGED = embed_staged(CM, FRAG);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E_::v8;
  float* prm = reinterpret_cast<float*>(lds + WBYTES);       // [bias][ln_w][ln_b][nn_w][nn_b]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  f32x4 nn_reg = {0.f, 0.f, 0.f, 0.f};
  if (EMIT && tid < E / 2) nn_reg = *reinterpret_cast<const f32x4*>((tid < E / 4 ? p.nn_w : p.nn_b - E) + 4 * tid);
  constexpr int NQ = (WBYTES + 3 * E * 4 + 1023) / 1024;     // weights + parameters, 1 KB wave-loads
  for (int q = wave; q < NQ; q += 4)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.pack + q * 1024 + lane * 16), (lds_ptr_t)(lds + q * 1024), 16, 0, 0);

  // a workgroup walks tiles of 128 tokens blockIdx.x, + gridDim.x, ... (round 5: a launch of fewer, longer-lived workgroups — the weights
  // reach LDS once per workgroup, and in the multi-lane mix the launch holds fewer CU slots while it streams its 200 MB)
  const long L0 = (long)p.D0 * p.H0 * p.W0, total = (long)p.B * L0;
  for (long tile = blockIdx.x; tile * 128 < total; tile += gridDim.x) {
  const bool first_tile = tile == (long)blockIdx.x;
  const long row = tile * 128 + wave * 32 + (lane & 31);
  const long rc = row < total ? row : total - 1;
  const int b = (int)(rc / L0);
  const int tl = (int)(rc - (long)b * L0);
  const int d = tl / (p.H0 * p.W0), hw = tl - d * (p.H0 * p.W0), hh = hw / p.W0, ww = hw - hh * p.W0;
  const bool live = row < total;
  V8 bx[KS];
  uint32_t raw[FRAG ? KS : 1][2];
  float* s_nn = reinterpret_cast<float*>(lds + NQ * 1024);    // past the DMA image (its last KB is padding)
  float* s_tab = s_nn + 2 * E;                                // FRAG: [Cin <= 4][256] normalised pixel values
  if constexpr (FRAG) {
    // a wave's 32 tokens lie in one clip (L0 % 32 == 0, checked on the host): the per-clip pointers are scalar loads
    const FragSrc& f = p.frag;
    const int bu = __builtin_amdgcn_readfirstlane(b);
    const int oy = hh * 4 + 2 * h, ox = ww * 4;
    const int fi = oy / f.fsh, fj = ox / f.fsw;
    const int nt = p.T / f.aligned;
    // the clip's three pointers: launch parameters, or (a recorded forward that serves every video) one scalar load each from
    // the caller's table
    const int32_t* hb = f.table ? (const int32_t*)f.table[KVQ_FRAG_MAX_CLIPS + bu] : f.hoff[bu];
    const int32_t* wb = f.table ? (const int32_t*)f.table[2 * KVQ_FRAG_MAX_CLIPS + bu] : f.woff[bu];
    const int32_t* ho = hb + (fi * f.Fw + fj) * nt;
    const int32_t* wo = wb + (fi * f.Fw + fj) * nt;
    const size_t plane = (size_t)f.Hs * f.Ws;
    const uint8_t* vb = f.table ? (const uint8_t*)f.table[bu] : f.video[bu];
    typedef uint32_t __attribute__((aligned(1))) u32u;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = s / p.pd, t = d * p.pd + (s - c * p.pd), g = t / f.aligned;
      const uint8_t* src = vb + (size_t)c * f.chan_stride + (size_t)t * plane + (size_t)(ho[g] + (oy - fi * f.fsh)) * f.Ws + (wo[g] + (ox - fj * f.fsw));
      raw[s][0] = *reinterpret_cast<const u32u*>(src);
      raw[s][1] = *reinterpret_cast<const u32u*>(src + f.Ws);
    }
    // a pixel is one of 256 bytes: (v - mean) / std — the IEEE fp32 divide of fragment_gather_kernel — once per byte value
    // and channel (Cin divides per thread) instead of once per pixel (8 * KS per lane); looked up below, behind the barrier
    if (first_tile)
      for (int c = 0; c < p.Cin; ++c) s_tab[c * 256 + tid] = ((float)tid - f.mean[c]) / f.std[c];
  } else {
    const size_t plane = (size_t)p.H * p.W;
    const float* base = p.x + (size_t)b * p.Cin * p.T * plane + (size_t)(d * p.pd) * plane + (size_t)(hh * 4 + 2 * h) * p.W + ww * 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = s / p.pd, kd = s - c * p.pd;       // pd is small; KS = Cin*pd is a compile-time count
      const float* src = base + ((size_t)c * p.T + kd) * plane;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(src);
      const f32x4 r1 = *reinterpret_cast<const f32x4*>(src + p.W);
      const u32x4 w = {E_::pack2(r0[0], r0[1]), E_::pack2(r0[2], r0[3]), E_::pack2(r1[0], r1[1]), E_::pack2(r1[2], r1[3])};
      bx[s] = __builtin_bit_cast(V8, w);
    }
  }
  if (first_tile) {
    if (EMIT && tid < E / 2) *reinterpret_cast<f32x4*>(s_nn + 4 * tid) = nn_reg;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if constexpr (FRAG) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* tab = s_tab + (s / p.pd) * 256;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tab[(raw[s][e >> 2] >> (8 * (e & 3))) & 255u];
      const u32x4 w = {E_::pack2(v[0], v[1]), E_::pack2(v[2], v[3]), E_::pack2(v[4], v[5]), E_::pack2(v[6], v[7])};
      bx[s] = __builtin_bit_cast(V8, w);
    }
  }

  f32x16 acc[CM];
#pragma unroll
  for (int i = 0; i < CM; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(prm + 32 * i + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = bq[e];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const V8 a = *reinterpret_cast<const V8*>(lds + (i * KS + s) * 1024 + lane * 16);
      acc[i] = E_::mfma32(a, bx[s], acc[i]);
    }
  }

  // LayerNorm over the E channels of a token: in-lane sums + one exchange with lane^32 (two-pass, as ln.hip)
  auto stats = [&](float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 4) s += (acc[i][r] + acc[i][r + 1]) + (acc[i][r + 2] + acc[i][r + 3]);
    s += __shfl_xor(s, 32);
    mean = s / (float)E;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dd = acc[i][r] - mean;
        sq += dd * dd;
      }
    sq += __shfl_xor(sq, 32);
    rstd = rsqrtf(sq / (float)E + p.eps);
    asm volatile("" : "+v"(mean));     // opaque: no CSE of (acc - mean) between the variance and the normalise pass
  };
  if (p.has_ln) {
    float mean, rstd;
    stats(mean, rstd);
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(prm + E + 32 * i + 8 * q + 4 * h);
        const f32x4 be = *reinterpret_cast<const f32x4*>(prm + 2 * E + 32 * i + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = (acc[i][4 * q + e] - mean) * rstd * g[e] + be[e];
      }
  }
  if constexpr (STAGED) {
    // The wave's 32 tokens are 32 consecutive rows of the residual stream: one contiguous 32 * E * 4 bytes.  Stored from the
    // accumulator layout every store instruction touches 64 different rows (64 cycles in the texture addresser, 24 of them per
    // wave at E = 96: most of this launch's memory-pipe time); through a wave-private LDS tile (row pitch + 16 B: the 32 lanes
    // of a half spread over the banks) the same bytes leave as 1 KB lines.  A wave's LDS accesses complete in order: no barrier.
    constexpr int RP = E * 4 + 16;
    unsigned char* stg = lds + embed_stage_off(CM, KS) + wave * 32 * RP;
    unsigned char* mine = stg + (lane & 31) * RP + 16 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(mine + 128 * i + 32 * q) = (f32x4){acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
    const long row0 = tile * 128 + wave * 32;
    const int nrow = (int)(total - row0 < 32 ? total - row0 : 32);        // <= 0: nothing of this wave is live
    unsigned char* gb = reinterpret_cast<unsigned char*>(p.out + (size_t)row0 * E);
#pragma unroll
    for (int k = 0; k < 32 * E * 4 / 1024; ++k) {
      const int off = k * 1024 + lane * 16, t = off / (E * 4), w = off - t * (E * 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + t * RP + w);
      if (t < nrow) *reinterpret_cast<f32x4*>(gb + off) = v;
    }
  } else if (live && p.out16) {
    uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (size_t)rc * E + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u32x2*>(o + 32 * i + 8 * q) =
            (u32x2){Fp16::pack2(acc[i][4 * q], acc[i][4 * q + 1]), Fp16::pack2(acc[i][4 * q + 2], acc[i][4 * q + 3])};
  } else if (live) {
    float* o = p.out + (size_t)rc * E + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(o + 32 * i + 8 * q) =
            (f32x4){acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  }
  if (EMIT) {
    float mean, rstd;
    stats(mean, rstd);
    // 16 bytes per lane: the lane pair (h = 0 | 1) of a token exchanges the 8-byte pieces of (q, q + 1) by v_permlane32_swap, lane h
    // then owns channels 8 (2 t + h) .. + 7 of a tile — half the row-divergent store instructions (one row per cycle in the addresser)
    const long drow = (long)b * p.next_rows + p.next_dst[tl];
    uint16_t* o = p.next_ln + (size_t)drow * E;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t pk[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = 2 * t + u;
          const f32x4 g = *reinterpret_cast<const f32x4*>(s_nn + 32 * i + 8 * q + 4 * h);
          const f32x4 be = *reinterpret_cast<const f32x4*>(s_nn + E + 32 * i + 8 * q + 4 * h);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[i][4 * q + e] - mean) * rstd * g[e] + be[e];
          pk[u][0] = E_::pack2(y[0], y[1]);
          pk[u][1] = E_::pack2(y[2], y[3]);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        if (live) *reinterpret_cast<u32x4*>(o + 32 * i + 8 * (2 * t + h)) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
      }
  }
  }      // tiles
}

template <typename E_, int CM, int KS>
static int launch_embed(const EmbedParams& p, hipStream_t st) {
  const int lds = embed_stage_off(CM, KS) + (embed_staged(CM, p.x == nullptr) ? 4 * 32 * (32 * CM * 4 + 16) : 0);
  const long total = (long)p.B * p.D0 * p.H0 * p.W0;
  // KVQ_EMBED_WGS: workgroups of the launch (each walks tiles of 128 tokens); 0 = one per tile (rounds 1-4)
  static const long wgs = getenv("KVQ_EMBED_WGS") ? atol(getenv("KVQ_EMBED_WGS")) : 0;
  const long ntile = (total + 127) / 128;
  dim3 grid((unsigned)(wgs > 0 && wgs < ntile ? wgs : ntile)), block(256);
  auto go = [&](auto k) -> int {
    LdsOptIn opt;                             // the opt-in is remembered per (kernel, device) in common.cpp
    if (int rc = opt.ensure(reinterpret_cast<const void*>(k), lds)) return rc;
    hipLaunchKernelGGL(k, grid, block, lds, st, p);
    return KVQ_OK;
  };
  int rc;
  if (p.x == nullptr) rc = p.next_ln ? go(patch_embed_kernel<E_, CM, KS, true, true>) : go(patch_embed_kernel<E_, CM, KS, false, true>);
  else rc = p.next_ln ? go(patch_embed_kernel<E_, CM, KS, true, false>) : go(patch_embed_kernel<E_, CM, KS, false, false>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("patch_embed_kernel");
  return KVQ_OK;
}

}  // namespace kvq

extern "C" int kvq_patch_embed_supported(int in_chans, int pd, int ph, int pw, int embed_dim, int T, int H, int W) {
  // one (c, kd) plane per MFMA k-step; no padded edge (the im2col path pads)
  return ph == 4 && pw == 4 && in_chans * pd == 6 && (embed_dim == 96 || embed_dim == 128) && T % pd == 0 && H % 4 == 0 &&
                 W % 4 == 0 ? 1 : 0;
}

extern "C" size_t kvq_patch_embed_pack_bytes(int embed_dim, int K) {
  if (embed_dim % 32 || K % 16) return 0;
  return (((size_t)embed_dim * K * 2 + (size_t)3 * embed_dim * 4) + 1023) & ~(size_t)1023;
}

extern "C" int kvq_patch_embed_pack(const void* w, const float* bias, const float* ln_w, const float* ln_b, int embed_dim, int K,
                                    void* pack, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(w && bias && pack, KVQ_ERR_NULL, "kvq_patch_embed_pack: NULL pointer");
  KVQ_REQUIRE(kvq_patch_embed_pack_bytes(embed_dim, K) > 0, KVQ_ERR_UNSUPPORTED, "kvq_patch_embed_pack: E=%d K=%d", embed_dim, K);
  const long n_chunks = (long)embed_dim * K * 2 / 16, total = n_chunks + 3 * embed_dim;
  hipLaunchKernelGGL(embed_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)w, bias, ln_w, ln_b, embed_dim, K, (unsigned char*)pack, n_chunks);
  KVQ_CHECK_LAUNCH("embed_pack_kernel");
  return KVQ_OK;
}

extern "C" int kvq_patch_embed_fragments_supported(const KvqFragmentSource* f, int B, int in_chans, int pd, int T, int H, int W) {
  // uint8 frames; whole 4 x 4 patches inside a mini-patch; one clip per wave of 32 tokens; frames of a token in range
  if (!f || !f->src_is_u8 || f->n_clips != B || B > KVQ_FRAG_MAX_CLIPS || in_chans > 4 || pd <= 0) return 0;
  if (f->fs_h <= 0 || f->fs_w <= 0 || f->fs_h % 4 || f->fs_w % 4 || f->Fh * f->fs_h != H || f->Fw * f->fs_w != W) return 0;
  if (f->aligned <= 0 || T % f->aligned || T % pd) return 0;
  if (f->chan_stride < 0 || (f->chan_stride && f->chan_stride < (long)T * f->Hs * f->Ws)) return 0;
  if (f->Hs < H || f->Ws < W) return 0;                      // the upsample fallback is not in the hot path (as kvq_fragment_gather)
  return ((long)(T / pd) * (H / 4) * (W / 4)) % 32 == 0 ? 1 : 0;
}

extern "C" int kvq_patch_embed(const KvqPatchEmbedArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && (a->x || a->frag) && a->pack && a->out, KVQ_ERR_NULL, "kvq_patch_embed: NULL pointer");
  KVQ_REQUIRE(!(a->x && a->frag), KVQ_ERR_UNSUPPORTED, "kvq_patch_embed: both a clip and a fragment source");
  KVQ_REQUIRE(kvq_patch_embed_supported(a->in_chans, a->pd, a->ph, a->pw, a->embed_dim, a->T, a->H, a->W), KVQ_ERR_UNSUPPORTED,
              "kvq_patch_embed: patch (%d,%d,%d) x %d channels -> %d on %dx%dx%d is not the fused shape", a->pd, a->ph, a->pw,
              a->in_chans, a->embed_dim, a->T, a->H, a->W);
  KVQ_REQUIRE(a->B > 0, KVQ_ERR_SHAPE, "kvq_patch_embed: B=%d", a->B);
  KVQ_REQUIRE(!a->next_ln || (a->next_norm_w && a->next_norm_b && a->next_dst && a->next_rows > 0), KVQ_ERR_NULL,
              "kvq_patch_embed: next_ln without its norm / map");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_patch_embed: dtype %d", a->dtype);
  EmbedParams p{};
  if (a->frag) {
    const KvqFragmentSource* f = a->frag;
    KVQ_REQUIRE(kvq_patch_embed_fragments_supported(f, a->B, a->in_chans, a->pd, a->T, a->H, a->W), KVQ_ERR_UNSUPPORTED,
                "kvq_patch_embed: fragment source (%d clips, u8=%d, %dx%d patches of %dx%d, aligned %d) does not fit the fused read "
                "of a %dx%dx%dx%d batch", f->n_clips, f->src_is_u8, f->Fh, f->Fw, f->fs_h, f->fs_w, f->aligned, a->B, a->T, a->H, a->W);
    p.frag.table = f->indirect;
    for (int b = 0; b < a->B && !f->indirect; ++b) {
      KVQ_REQUIRE(f->video[b] && f->hoff[b] && f->woff[b], KVQ_ERR_NULL, "kvq_patch_embed: fragment source clip %d has a NULL pointer", b);
      p.frag.video[b] = (const uint8_t*)f->video[b]; p.frag.hoff[b] = f->hoff[b]; p.frag.woff[b] = f->woff[b];
    }
    p.frag.chan_stride = f->chan_stride ? f->chan_stride : (long)a->T * f->Hs * f->Ws;
    p.frag.Hs = f->Hs; p.frag.Ws = f->Ws; p.frag.Fw = f->Fw; p.frag.fsh = f->fs_h; p.frag.fsw = f->fs_w; p.frag.aligned = f->aligned;
    for (int c = 0; c < 4; ++c) { p.frag.mean[c] = f->normalise ? f->mean[c] : 0.f; p.frag.std[c] = f->normalise ? f->std[c] : 1.f; }   // (v - 0) / 1 == v
  }
  p.x = a->x; p.B = a->B; p.Cin = a->in_chans; p.T = a->T; p.H = a->H; p.W = a->W; p.pd = a->pd;
  p.D0 = a->T / a->pd; p.H0 = a->H / 4; p.W0 = a->W / 4;
  p.pack = (const unsigned char*)a->pack; p.has_ln = a->has_norm; p.out = a->out; p.out16 = a->out_f16; p.nn_w = a->next_norm_w; p.nn_b = a->next_norm_b;
  p.next_dst = a->next_dst; p.next_ln = (uint16_t*)a->next_ln; p.next_rows = a->next_rows; p.eps = a->eps;
  hipStream_t st = (hipStream_t)stream;
  if (a->embed_dim == 96)
    return a->dtype == KVQ_DT_FP16 ? launch_embed<Fp16, 3, 6>(p, st) : launch_embed<Bf16, 3, 6>(p, st);
  return a->dtype == KVQ_DT_FP16 ? launch_embed<Fp16, 4, 6>(p, st) : launch_embed<Bf16, 4, 6>(p, st);
}
