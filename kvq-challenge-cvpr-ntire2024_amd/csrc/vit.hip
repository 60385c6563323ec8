// Small kernels of the KSVQE "CLIP_tool" (the reference's CLIP_extractor_addadapter_cls, CLIP_backbone.py:156-201, over the
// vendored CLIP vision transformer, clip/model.py:184-294): 50..197 tokens per key frame, a handful of frames per video —
// the GEMMs (patch embedding, in_proj, out_proj, c_fc + QuickGELU, c_proj, the CLS adapters) and the LayerNorms are the
// trunk's kernels (gemm.hip, ln.hip); what is left is token assembly, a short-sequence multi-head attention, the CLS
// adapter mix and the cosine map.  All HBM-trivial: written for clarity, fp32 arithmetic throughout.
#include <algorithm>
#include "common.hpp"

namespace kvq {

// x[b][0] = cls + pos[0], x[b][1+i] = tok[b*G+i] + pos[1+i]; then ln_pre over D (two-pass statistics, eps inside rsqrt)
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ lw,
                                                           const float* __restrict__ lb, int G, int D, float eps,
                                                           float* __restrict__ out) {
  const int l = blockIdx.x % (G + 1), b = blockIdx.x / (G + 1), tid = threadIdx.x;
  const float* src = l == 0 ? cls : tok + ((size_t)b * G + l - 1) * D;
  const float* pr = pos + (size_t)l * D;
  __shared__ float red[8];
  float s = 0.f;
  for (int c = tid; c < D; c += 256) s += src[c] + pr[c];
  auto block_sum = [&](float v) -> float {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  const float mean = block_sum(s) / (float)D;
  float q = 0.f;
  for (int c = tid; c < D; c += 256) {
    const float d = src[c] + pr[c] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum(q) / (float)D + eps);
  float* o = out + ((size_t)b * (G + 1) + l) * D;
  for (int c = tid; c < D; c += 256) o[c] = (src[c] + pr[c] - mean) * rstd * lw[c] + lb[c];
}

// Short-sequence multi-head attention, head_dim HD: out[b][i][h*HD..] = softmax_j(scale * q_i . k_j) v_j over the Lk keys of
// batch element b.  q / k / v are 16-bit row-major with their own row strides (the packed in_proj output of
// nn.MultiheadAttention is q = qkv, k = qkv + D, v = qkv + 2D with stride 3D; a cross-attention passes separate tensors).
// One workgroup per (batch element, head) of 64 * ceil(min(Lq, 256) / 64) threads: K and V of the head in LDS (16-bit), a
// thread owns a query and runs the online softmax over the keys in fp32 (K / V rows are LDS broadcasts).  The launches are
// latency-bound (197 queries x 197 keys per head): four waves cover the queries of a CLIP frame in one pass.
struct MhaParams {
  const uint16_t *q, *k, *v;
  long ldq, ldk, ldv;          // row strides in elements
  int Lq, Lk, heads;
  float scale;
  uint16_t* out;               // [B*Lq][heads*HD]
};

// a.lo * b.lo + a.hi * b.hi + c on packed 16-bit pairs, fp32 accumulate
template <typename E>
__device__ __forceinline__ float mha_dot2(uint32_t a, uint32_t b, float c);
template <>
__device__ __forceinline__ float mha_dot2<Fp16>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
}
template <>
__device__ __forceinline__ float mha_dot2<Bf16>(uint32_t a, uint32_t b, float c) {
  return fmaf(__uint_as_float(a << 16), __uint_as_float(b << 16), fmaf(__uint_as_float(a & 0xffff0000u), __uint_as_float(b & 0xffff0000u), c));
}

template <typename E, int HD>
__global__ __launch_bounds__(256) void mha_small_kernel(MhaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem);
  uint16_t* Vs = Ks + (size_t)p.Lk * HD;
  float* part = reinterpret_cast<float*>(Vs + (size_t)p.Lk * HD);       // key-split partials: [ks-1][qwv*64][HD+2]
  const int h = blockIdx.x % p.heads, b = blockIdx.x / p.heads, tid = threadIdx.x, D = p.heads * HD;
  const uint16_t* kb = p.k + (size_t)b * p.Lk * p.ldk + h * HD;
  const uint16_t* vb = p.v + (size_t)b * p.Lk * p.ldv + h * HD;
  const int nthr = blockDim.x;
  for (int i = tid; i < p.Lk * (HD / 8); i += nthr) {
    const int r = i / (HD / 8), c = i % (HD / 8);
    *reinterpret_cast<u32x4*>(Ks + r * HD + c * 8) = *reinterpret_cast<const u32x4*>(kb + (size_t)r * p.ldk + c * 8);
    *reinterpret_cast<u32x4*>(Vs + r * HD + c * 8) = *reinterpret_cast<const u32x4*>(vb + (size_t)r * p.ldv + c * 8);
  }
  __syncthreads();
  // Four waves per (batch element, head).  Up to 64 queries: one query per lane and the KEYS split over the four waves (a
  // CLIP frame at 112x112 has 50 tokens: a single wave walking 50 keys is pure latency), partial (max, sum, o) merged through
  // LDS; up to 128: two query waves x two key halves; more: the waves take 64 queries each and walk all keys.
  const int nw = nthr >> 6, wave = tid >> 6, lane = tid & 63;
  const int ks = p.Lq <= 64 ? nw : (p.Lq <= 128 && nw >= 2 ? nw / 2 : 1);      // key splits
  const int qwv = nw / ks, qstride = qwv * 64;
  const int kpart = wave / qwv, kper = (p.Lk + ks - 1) / ks;
  const int k_lo = kpart * kper, k_hi = min(p.Lk, k_lo + kper);
  const uint16_t* qb = p.q + (size_t)b * p.Lq * p.ldq + h * HD;
  // The loop is issue-bound (LDS instructions first, VALU second), so: 16-byte LDS reads of the K / V rows (a broadcast each),
  // q kept as packed 16-bit pairs and multiplied by v_dot2 (fp16) without conversions, the logit scale applied to the fp32
  // dot, and the online-softmax rescale of the 64 accumulators done once per 8 keys instead of once per key.
  for (int l0 = (wave % qwv) * 64; l0 < p.Lq; l0 += qstride) {
    const int l = l0 + lane;
    const bool live = l < p.Lq;
    const int lq = live ? l : p.Lq - 1;
    uint32_t qp[HD / 2];
    float o[HD];
#pragma unroll
    for (int c8 = 0; c8 < HD / 8; ++c8) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(qb + (size_t)lq * p.ldq + c8 * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) qp[c8 * 4 + e] = t[e];
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
    float mx = -INFINITY, sum = 0.f;
    for (int j0 = k_lo; j0 < k_hi; j0 += 8) {
      float sc[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        float acc = 0.f;
        if (j0 + jj < k_hi) {
          const uint16_t* kr = Ks + (j0 + jj) * HD;
#pragma unroll
          for (int c8 = 0; c8 < HD / 8; ++c8) {
            const u32x4 kv = *reinterpret_cast<const u32x4*>(kr + c8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mha_dot2<E>(qp[c8 * 4 + e], kv[e], acc);
          }
          sc[jj] = acc * p.scale;
        } else {
          sc[jj] = -INFINITY;
        }
      }
      float nm = mx;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) nm = fmaxf(nm, sc[jj]);
      const float corr = __expf(mx - nm);
      sum *= corr;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] *= corr;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (j0 + jj >= k_hi) break;
        const float pj = __expf(sc[jj] - nm);
        sum += pj;
        const uint16_t* vr = Vs + (j0 + jj) * HD;
#pragma unroll
        for (int c8 = 0; c8 < HD / 8; ++c8) {
          const u32x4 vv = *reinterpret_cast<const u32x4*>(vr + c8 * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[c8 * 8 + 2 * e] = fmaf(pj, E::to_f32((uint16_t)(vv[e] & 0xffffu)), o[c8 * 8 + 2 * e]);
            o[c8 * 8 + 2 * e + 1] = fmaf(pj, E::to_f32((uint16_t)(vv[e] >> 16)), o[c8 * 8 + 2 * e + 1]);
          }
        }
      }
      mx = nm;
    }
    if (ks > 1) {           // one pass of the query loop in this mode (qstride >= Lq): the barrier is uniform
      float* mine = part + ((size_t)(kpart > 0 ? kpart - 1 : 0) * qstride + (wave % qwv) * 64 + lane) * (HD + 2);
      if (kpart > 0) {
        mine[0] = mx;
        mine[1] = sum;
#pragma unroll
        for (int c = 0; c < HD; ++c) mine[2 + c] = o[c];
      }
      __syncthreads();
      if (kpart > 0) continue;
      float M = mx;
      for (int s = 0; s < ks - 1; ++s) M = fmaxf(M, part[((size_t)s * qstride + (wave % qwv) * 64 + lane) * (HD + 2)]);
      const float c0 = __expf(mx - M);        // a part without keys carries (-inf, 0, 0): weight exp(-inf) = 0
      sum *= c0;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] *= c0;
      for (int s = 0; s < ks - 1; ++s) {
        const float* pr = part + ((size_t)s * qstride + (wave % qwv) * 64 + lane) * (HD + 2);
        const float cs = __expf(pr[0] - M);
        sum = fmaf(pr[1], cs, sum);
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = fmaf(pr[2 + c], cs, o[c]);
      }
    }
    if (!live) continue;
    const float inv = 1.f / sum;
    uint16_t* dst = p.out + ((size_t)b * p.Lq + l) * D + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 2) *reinterpret_cast<uint32_t*>(dst + c) = E::pack2(o[c] * inv, o[c + 1] * inv);
  }
}

// CLS rows of x (B, L, D) fp32 -> 16-bit [B][D] (the adapter's GEMM operand)
template <typename E>
__global__ void cls_gather_kernel(const float* __restrict__ x, int L, int D, uint16_t* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % D);
  const long b = i / D;
  out[i] = E::cvt(x[(b * L) * (long)D + c]);
}

// x[b][0] = ratio * a[b] + (1 - ratio) * x[b][0]  (CLIP_backbone.py:187-191, ratio 0.5)
template <typename E>
__global__ void cls_mix_kernel(float* __restrict__ x, const uint16_t* __restrict__ a, int L, int D, float ratio, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % D);
  const long b = i / D;
  float* p = x + (b * L) * (long)D + c;
  *p = ratio * E::to_f32(a[i]) + (1.f - ratio) * *p;
}

// torch.cosine_similarity(cls, patches, dim=-1): x (B, L, D) fp32 -> out (B, L-1); eps 1e-8 on each norm as ATen does
__global__ __launch_bounds__(64) void cosine_cls_kernel(const float* __restrict__ x, int L, int D, float* __restrict__ out) {
  const int l = 1 + blockIdx.x % (L - 1), b = blockIdx.x / (L - 1), lane = threadIdx.x;
  const float* c = x + (size_t)b * L * D;
  const float* p = c + (size_t)l * D;
  float dot = 0.f, nc = 0.f, np = 0.f;
  for (int i = lane; i < D; i += 64) {
    dot = fmaf(c[i], p[i], dot);
    nc = fmaf(c[i], c[i], nc);
    np = fmaf(p[i], p[i], np);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dot += __shfl_xor(dot, o);
    nc += __shfl_xor(nc, o);
    np += __shfl_xor(np, o);
  }
  if (lane == 0) out[(size_t)b * (L - 1) + l - 1] = dot / (fmaxf(sqrtf(nc), 1e-8f) * fmaxf(sqrtf(np), 1e-8f));
}

}  // namespace kvq

extern "C" int kvq_vit_embed_ln(const float* tok, const float* cls, const float* pos, const float* ln_w, const float* ln_b, int B,
                                int G, int D, float eps, float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(tok && cls && pos && ln_w && ln_b && out, KVQ_ERR_NULL, "kvq_vit_embed_ln: NULL pointer");
  KVQ_REQUIRE(B > 0 && G > 0 && D > 0, KVQ_ERR_SHAPE, "kvq_vit_embed_ln: bad shape B=%d G=%d D=%d", B, G, D);
  hipLaunchKernelGGL(vit_embed_ln_kernel, dim3((unsigned)(B * (G + 1))), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, ln_w,
                     ln_b, G, D, eps, out);
  KVQ_CHECK_LAUNCH("vit_embed_ln_kernel");
  return KVQ_OK;
}

extern "C" int kvq_mha_cross(const uint16_t* q, long ldq, const uint16_t* k, long ldk, const uint16_t* v, long ldv, int B, int Lq,
                             int Lk, int heads, int head_dim, float scale, int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(q && k && v && out, KVQ_ERR_NULL, "kvq_mha_cross: NULL pointer");
  KVQ_REQUIRE(B > 0 && Lq > 0 && Lk > 0 && Lk <= 320 && heads > 0 && ldq >= (long)heads * head_dim && ldk >= (long)heads * head_dim &&
                  ldv >= (long)heads * head_dim && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0,
              KVQ_ERR_SHAPE, "kvq_mha_cross: bad shape B=%d Lq=%d Lk=%d heads=%d (Lk <= 320, strides multiples of 8)", B, Lq, Lk, heads);
  KVQ_REQUIRE(head_dim == 64, KVQ_ERR_UNSUPPORTED, "kvq_mha_cross: head_dim %d (64 only)", head_dim);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_mha_cross: dtype %d", dtype);
  KVQ_REQUIRE((((size_t)k | (size_t)v) & 15) == 0, KVQ_ERR_SHAPE, "kvq_mha_cross: k / v must be 16-byte aligned");
  MhaParams p{q, k, v, ldq, ldk, ldv, Lq, Lk, heads, scale, out};
  // K + V of the head (16-bit) + the key-split partials of Lq <= 128 (three parts x 64 queries, or one x 128)
  const size_t lds = (size_t)2 * Lk * 64 * sizeof(uint16_t) + (Lq <= 64 ? 3 * 64 : (Lq <= 128 ? 128 : 0)) * (size_t)(64 + 2) * sizeof(float);
  dim3 grid((unsigned)(B * heads)), block(256);
  if (dtype == KVQ_DT_FP16) {
    auto kern = mha_small_kernel<Fp16, 64>;
    if (lds > 64 * 1024) KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, p);
  } else {
    auto kern = mha_small_kernel<Bf16, 64>;
    if (lds > 64 * 1024) KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, p);
  }
  KVQ_CHECK_LAUNCH("mha_small_kernel");
  return KVQ_OK;
}

extern "C" int kvq_mha_small(const uint16_t* qkv, int B, int L, int heads, int head_dim, int dtype, uint16_t* out, void* stream) {
  KVQ_REQUIRE(qkv, KVQ_ERR_NULL, "kvq_mha_small: NULL pointer");
  KVQ_REQUIRE(heads > 0 && head_dim > 0, KVQ_ERR_SHAPE, "kvq_mha_small: bad shape");
  const long D = (long)heads * head_dim;
  return kvq_mha_cross(qkv, 3 * D, qkv + D, 3 * D, qkv + 2 * D, 3 * D, B, L, L, heads, head_dim, 1.0f / sqrtf((float)head_dim), dtype, out,
                       stream);
}

extern "C" int kvq_cls_gather(const float* x, int B, int L, int D, int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out, KVQ_ERR_NULL, "kvq_cls_gather: NULL pointer");
  KVQ_REQUIRE(B > 0 && L > 0 && D > 0, KVQ_ERR_SHAPE, "kvq_cls_gather: bad shape");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_cls_gather: dtype %d", dtype);
  const long total = (long)B * D;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(cls_gather_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, x, L, D, out, total);
  else hipLaunchKernelGGL(cls_gather_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, x, L, D, out, total);
  KVQ_CHECK_LAUNCH("cls_gather_kernel");
  return KVQ_OK;
}

extern "C" int kvq_cls_mix(float* x, const uint16_t* a, int B, int L, int D, float ratio, int dtype, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && a, KVQ_ERR_NULL, "kvq_cls_mix: NULL pointer");
  KVQ_REQUIRE(B > 0 && L > 0 && D > 0, KVQ_ERR_SHAPE, "kvq_cls_mix: bad shape");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_cls_mix: dtype %d", dtype);
  const long total = (long)B * D;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(cls_mix_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, x, a, L, D, ratio, total);
  else hipLaunchKernelGGL(cls_mix_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, x, a, L, D, ratio, total);
  KVQ_CHECK_LAUNCH("cls_mix_kernel");
  return KVQ_OK;
}

extern "C" int kvq_cosine_cls(const float* x, int B, int L, int D, float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out, KVQ_ERR_NULL, "kvq_cosine_cls: NULL pointer");
  KVQ_REQUIRE(B > 0 && L > 1 && D > 0, KVQ_ERR_SHAPE, "kvq_cosine_cls: bad shape B=%d L=%d D=%d", B, L, D);
  hipLaunchKernelGGL(cosine_cls_kernel, dim3((unsigned)(B * (L - 1))), dim3(64), 0, (hipStream_t)stream, x, L, D, out);
  KVQ_CHECK_LAUNCH("cosine_cls_kernel");
  return KVQ_OK;
}

// ---- KSVQE content-distortion modulation (CDM) pieces, KSVQE_model.py:817-835, :934-960 -----------------------------
namespace kvq {

template <typename E>
__global__ void to_half_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
    *reinterpret_cast<u32x2*>(out + i) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
  } else {
    for (long j = i; j < n; ++j) out[j] = E::cvt(x[j]);
  }
}

template <typename E>
__global__ void to_float_kernel(const uint16_t* __restrict__ x, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = E::to_f32(x[i]);
}

// Semantic_Transformation2: per token row m: gama = sigmoid(<wg, x_m> + bg), beta = <wb, x_m> + bb; out_m = gama * in_m + beta
__global__ __launch_bounds__(64) void sem_modulate_kernel(const float* __restrict__ x, const float* __restrict__ inp,
                                                          const float* __restrict__ wg, float bg, const float* __restrict__ wb,
                                                          float bb, int C, float* __restrict__ out) {
  const size_t m = blockIdx.x;
  const int lane = threadIdx.x;
  const float* xr = x + m * C;
  float sg = 0.f, sb = 0.f;
  for (int c = lane; c < C; c += 64) {
    sg = fmaf(wg[c], xr[c], sg);
    sb = fmaf(wb[c], xr[c], sb);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sg += __shfl_xor(sg, o);
    sb += __shfl_xor(sb, o);
  }
  const float gama = 1.f / (1.f + __expf(-(sg + bg))), beta = sb + bb;
  for (int c = lane; c < C; c += 64) out[m * C + c] = fmaf(gama, inp[m * C + c], beta);
}

// Dist_Transformation3: out[b][t][c] = sigmoid(g[b][c]) * in[b][t][c] + beta[b][c]   (g, beta: the two Linear outputs)
template <typename E>
__global__ void dist_modulate_kernel(const float* __restrict__ inp, const uint16_t* __restrict__ g, const uint16_t* __restrict__ beta,
                                     int rows, int C, float* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long b = i / ((long)rows * C);
  const float gm = 1.f / (1.f + __expf(-E::to_f32(g[b * C + c])));
  out[i] = fmaf(gm, inp[i], E::to_f32(beta[b * C + c]));
}

}  // namespace kvq

extern "C" int kvq_convert(const void* src, void* dst, long n, int to_half, int dtype, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(src && dst, KVQ_ERR_NULL, "kvq_convert: NULL pointer");
  KVQ_REQUIRE(n > 0, KVQ_ERR_SHAPE, "kvq_convert: n=%ld", n);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_convert: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (to_half) {
    KVQ_REQUIRE((((size_t)src & 15) | ((size_t)dst & 7)) == 0, KVQ_ERR_SHAPE, "kvq_convert: unaligned");
    dim3 grid((unsigned)((n / 4 + 256) / 256)), block(256);
    if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(to_half_kernel<Fp16>, grid, block, 0, st, (const float*)src, (uint16_t*)dst, n);
    else hipLaunchKernelGGL(to_half_kernel<Bf16>, grid, block, 0, st, (const float*)src, (uint16_t*)dst, n);
  } else {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(to_float_kernel<Fp16>, grid, block, 0, st, (const uint16_t*)src, (float*)dst, n);
    else hipLaunchKernelGGL(to_float_kernel<Bf16>, grid, block, 0, st, (const uint16_t*)src, (float*)dst, n);
  }
  KVQ_CHECK_LAUNCH("convert kernel");
  return KVQ_OK;
}

extern "C" int kvq_sem_modulate(const float* x, const float* input, const float* w_gama, float b_gama, const float* w_beta,
                                float b_beta, int M, int C, float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && input && w_gama && w_beta && out, KVQ_ERR_NULL, "kvq_sem_modulate: NULL pointer");
  KVQ_REQUIRE(M > 0 && C > 0, KVQ_ERR_SHAPE, "kvq_sem_modulate: bad shape M=%d C=%d", M, C);
  hipLaunchKernelGGL(sem_modulate_kernel, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, x, input, w_gama, b_gama, w_beta,
                     b_beta, C, out);
  KVQ_CHECK_LAUNCH("sem_modulate_kernel");
  return KVQ_OK;
}

extern "C" int kvq_dist_modulate(const float* input, const uint16_t* gamma_logit, const uint16_t* beta, int B, int rows, int C,
                                 int dtype, float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(input && gamma_logit && beta && out, KVQ_ERR_NULL, "kvq_dist_modulate: NULL pointer");
  KVQ_REQUIRE(B > 0 && rows > 0 && C > 0, KVQ_ERR_SHAPE, "kvq_dist_modulate: bad shape");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_dist_modulate: dtype %d", dtype);
  const long total = (long)B * rows * C;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16)
    hipLaunchKernelGGL(dist_modulate_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, input, gamma_logit, beta, rows, C, out, total);
  else
    hipLaunchKernelGGL(dist_modulate_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, input, gamma_logit, beta, rows, C, out, total);
  KVQ_CHECK_LAUNCH("dist_modulate_kernel");
  return KVQ_OK;
}

// ---- KSVQE quality-aware region selection (QRS), eval path of RegionNet_CLIP.forward (patchnet.py:461-550) ---------
namespace kvq {

// score (BK, gs, gs) -> nearest upsample to (gh, gw) (legacy nearest: src = floor(dst * gs / g)) -> mean over every kh x kw
// window (F.unfold, stride 1) -> first argmax over the (gh-kh+1) x (gw-kw+1) candidates (min-max normalisation is monotonic).
__global__ __launch_bounds__(64) void qrs_top_region_kernel(const float* __restrict__ score, int gs, int gh, int gw, int kh, int kw,
                                                           int32_t* __restrict__ idx) {
  const int bk = blockIdx.x, lane = threadIdx.x, ny = gh - kh + 1, nx = gw - kw + 1, nreg = ny * nx;
  const float* s = score + (size_t)bk * gs * gs;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int r = lane; r < nreg; r += 64) {
    const int ry = r / nx, rx = r % nx;
    float acc = 0.f;
    for (int y = 0; y < kh; ++y)
      for (int x = 0; x < kw; ++x) {
        const int sy = min((int)floorf((float)(ry + y) * ((float)gs / (float)gh)), gs - 1);
        const int sx = min((int)floorf((float)(rx + x) * ((float)gs / (float)gw)), gs - 1);
        acc += s[sy * gs + sx];
      }
    acc /= (float)(kh * kw);
    if (acc > best) { best = acc; besti = r; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(besti, o);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) idx[bk] = besti;
}

// out[b][c][t][y][x] = x[b][c][t][ry*anchor + y][rx*anchor + x], (ry, rx) = region[b*T + t] decoded over nx candidates per row
__global__ void crop_regions_kernel(const float* __restrict__ x, const int32_t* __restrict__ region, int C, int T, int H, int W,
                                    int anchor, int nx, int oh, int ow, float* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % ow);
  long r = i / ow;
  const int yy = (int)(r % oh); r /= oh;
  const int t = (int)(r % T); r /= T;
  const int c = (int)(r % C);
  const long b = r / C;
  const int reg = region[b * T + t], ry = reg / nx, rx = reg % nx;
  out[i] = x[(((b * C + c) * T + t) * (long)H + ry * anchor + yy) * W + rx * anchor + xx];
}

// the same with 16-byte accesses: anchor, W and the output width are multiples of 4 pixels (32-pixel anchors)
__global__ void crop_regions_vec4_kernel(const float* __restrict__ x, const int32_t* __restrict__ region, int C, int T, int H, int W,
                                         int anchor, int nx, int oh, int ow, float* __restrict__ out, long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int ow4 = ow / 4;
  const int xx = (int)(i % ow4) * 4;
  long r = i / ow4;
  const int yy = (int)(r % oh); r /= oh;
  const int t = (int)(r % T); r /= T;
  const int c = (int)(r % C);
  const long b = r / C;
  const int reg = region[b * T + t], ry = reg / nx, rx = reg % nx;
  *reinterpret_cast<f32x4*>(out + i * 4) =
      *reinterpret_cast<const f32x4*>(x + (((b * C + c) * T + t) * (long)H + ry * anchor + yy) * W + rx * anchor + xx);
}

}  // namespace kvq

extern "C" int kvq_qrs_top_region(const float* score, int BK, int gs, int gh, int gw, int kh, int kw, int32_t* idx, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(score && idx, KVQ_ERR_NULL, "kvq_qrs_top_region: NULL pointer");
  KVQ_REQUIRE(BK > 0 && gs > 0 && gh >= kh && gw >= kw && kh > 0 && kw > 0, KVQ_ERR_SHAPE,
              "kvq_qrs_top_region: bad shape gs=%d grid %dx%d window %dx%d", gs, gh, gw, kh, kw);
  hipLaunchKernelGGL(qrs_top_region_kernel, dim3((unsigned)BK), dim3(64), 0, (hipStream_t)stream, score, gs, gh, gw, kh, kw, idx);
  KVQ_CHECK_LAUNCH("qrs_top_region_kernel");
  return KVQ_OK;
}

extern "C" int kvq_crop_regions(const float* x, const int32_t* region, int B, int C, int T, int H, int W, int anchor, int kh, int kw,
                                float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && region && out, KVQ_ERR_NULL, "kvq_crop_regions: NULL pointer");
  KVQ_REQUIRE(B > 0 && C > 0 && T > 0 && anchor > 0 && H / anchor >= kh && W / anchor >= kw && kh > 0 && kw > 0, KVQ_ERR_SHAPE,
              "kvq_crop_regions: bad shape %dx%d anchor %d window %dx%d", H, W, anchor, kh, kw);
  const int oh = kh * anchor, ow = kw * anchor, nx = W / anchor - kw + 1;
  const long total = (long)B * C * T * oh * ow;
  if (anchor % 4 == 0 && W % 4 == 0 && (((size_t)x | (size_t)out) & 15) == 0) {
    hipLaunchKernelGGL(crop_regions_vec4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, region,
                       C, T, H, W, anchor, nx, oh, ow, out, total / 4);
    KVQ_CHECK_LAUNCH("crop_regions_vec4_kernel");
    return KVQ_OK;
  }
  hipLaunchKernelGGL(crop_regions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, region, C, T,
                     H, W, anchor, nx, oh, ow, out, total);
  KVQ_CHECK_LAUNCH("crop_regions_kernel");
  return KVQ_OK;
}

// F.normalize(x, dim=1) (x / max(||x||_2, 1e-12)) on fp32 rows -> 16-bit (the projector's GEMM operand; CONTRIQUE_model.forward,
// KSVQE_model.py:1654-1656)
namespace kvq {
template <typename E>
__global__ __launch_bounds__(64) void l2_normalize_rows_kernel(const float* __restrict__ x, int D, uint16_t* __restrict__ out) {
  const size_t m = blockIdx.x;
  const int lane = threadIdx.x;
  const float* xr = x + m * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s = fmaf(xr[c], xr[c], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int c = lane; c < D; c += 64) out[m * D + c] = E::cvt(xr[c] * inv);
}
}  // namespace kvq

extern "C" int kvq_l2_normalize_rows(const float* x, int M, int D, int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out, KVQ_ERR_NULL, "kvq_l2_normalize_rows: NULL pointer");
  KVQ_REQUIRE(M > 0 && D > 0, KVQ_ERR_SHAPE, "kvq_l2_normalize_rows: bad shape");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_l2_normalize_rows: dtype %d", dtype);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(l2_normalize_rows_kernel<Fp16>, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, x, D, out);
  else hipLaunchKernelGGL(l2_normalize_rows_kernel<Bf16>, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, x, D, out);
  KVQ_CHECK_LAUNCH("l2_normalize_rows_kernel");
  return KVQ_OK;
}

// out = a * x + b * y (fp32, n elements): the fixed mixes of KSVQE.forward (0.2 / 0.8 adapter blend :1426, (a1 x_d + a2 x_s) / 2 :1482)
namespace kvq {
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fmaf(a, x[i], b * y[i]);
}
}  // namespace kvq

extern "C" int kvq_axpby(const float* x, const float* y, float a, float b, float* out, long n, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && y && out, KVQ_ERR_NULL, "kvq_axpby: NULL pointer");
  KVQ_REQUIRE(n > 0, KVQ_ERR_SHAPE, "kvq_axpby: n=%ld", n);
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, out, n);
  KVQ_CHECK_LAUNCH("axpby_kernel");
  return KVQ_OK;
}
