// Small kernels of the KSVQE "CLIP_tool" (the reference's CLIP_extractor_addadapter_cls, CLIP_backbone.py:156-201, over the
// vendored CLIP vision transformer, clip/model.py:184-294): 50..197 tokens per key frame, a handful of frames per video —
// the GEMMs (patch embedding, in_proj, out_proj, c_fc + QuickGELU, c_proj, the CLS adapters) and the LayerNorms are the
// trunk's kernels (gemm.hip, ln.hip); what is left is token assembly, a short-sequence multi-head attention, the CLS
// adapter mix and the cosine map.  All HBM-trivial: written for clarity, fp32 arithmetic throughout.
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

// nn.MultiheadAttention core on the in_proj output: qkv [B*L][3*D] 16-bit, token-major rows [q | k | v], head h = columns
// h*HD .. of each third.  One workgroup per (batch element, head): K and V of the head in LDS (16-bit), a thread owns a
// query and runs the online softmax over the keys in fp32 (K / V rows are LDS broadcasts).
template <typename E, int HD>
__global__ __launch_bounds__(64) void mha_small_kernel(const uint16_t* __restrict__ qkv, int L, int heads, uint16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem);
  uint16_t* Vs = Ks + (size_t)L * HD;
  const int h = blockIdx.x % heads, b = blockIdx.x / heads, tid = threadIdx.x, D = heads * HD;
  const uint16_t* base = qkv + (size_t)b * L * 3 * D + h * HD;
  for (int i = tid; i < L * (HD / 8); i += 64) {
    const int r = i / (HD / 8), c = i % (HD / 8);
    *reinterpret_cast<u32x4*>(Ks + r * HD + c * 8) = *reinterpret_cast<const u32x4*>(base + (size_t)r * 3 * D + D + c * 8);
    *reinterpret_cast<u32x4*>(Vs + r * HD + c * 8) = *reinterpret_cast<const u32x4*>(base + (size_t)r * 3 * D + 2 * D + c * 8);
  }
  __syncthreads();
  const float scale = rsqrtf((float)HD);
  for (int l = tid; l < L; l += 64) {
    float q[HD], o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      q[c] = E::to_f32(base[(size_t)l * 3 * D + c]) * scale;
      o[c] = 0.f;
    }
    float mx = -INFINITY, sum = 0.f;
    for (int j = 0; j < L; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s = fmaf(q[c], E::to_f32(Ks[j * HD + c]), s);
      const float nm = fmaxf(mx, s), corr = __expf(mx - nm), pj = __expf(s - nm);
      sum = sum * corr + pj;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(pj, E::to_f32(Vs[j * HD + c]), o[c] * corr);
      mx = nm;
    }
    const float inv = 1.f / sum;
    uint16_t* dst = out + ((size_t)b * L + l) * D + h * HD;
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

extern "C" int kvq_mha_small(const uint16_t* qkv, int B, int L, int heads, int head_dim, int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(qkv && out, KVQ_ERR_NULL, "kvq_mha_small: NULL pointer");
  KVQ_REQUIRE(B > 0 && L > 0 && L <= 320 && heads > 0, KVQ_ERR_SHAPE, "kvq_mha_small: bad shape B=%d L=%d heads=%d (L <= 320)", B, L, heads);
  KVQ_REQUIRE(head_dim == 64, KVQ_ERR_UNSUPPORTED, "kvq_mha_small: head_dim %d (64 = CLIP ViT-B)", head_dim);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_mha_small: dtype %d", dtype);
  const size_t lds = (size_t)2 * L * 64 * sizeof(uint16_t);
  dim3 grid((unsigned)(B * heads)), block(64);
  if (dtype == KVQ_DT_FP16) {
    auto k = mha_small_kernel<Fp16, 64>;
    if (lds > 64 * 1024) KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, (hipStream_t)stream, qkv, L, heads, out);
  } else {
    auto k = mha_small_kernel<Bf16, 64>;
    if (lds > 64 * 1024) KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, (hipStream_t)stream, qkv, L, heads, out);
  }
  KVQ_CHECK_LAUNCH("mha_small_kernel");
  return KVQ_OK;
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
