// Window attention with gated relative-position bias (GRPB) for head_dim 32 on gfx950.
//
// Replaces WindowAttention3D.forward's core (swin_backbone.py:261-322): q@k^T, the bias-table
// gathers table[rpi] (:272-288), the fragment gate mix rpb*g + fpb*(1-g) with g = |dfrag| summed
// (:291-302), the 0/-100 shift mask add (:311-316), softmax and attn@v — without materialising
// any (nW,nH,N,N) or (nW,N,N,3) tensor: the bias is rebuilt per score from two int32 per token
// (a linear position code and a packed {frag_h, frag_w, region} descriptor) and the per-head
// tables held in LDS.
//
// One workgroup (5 waves) = one (window, head).  K (swizzled), V^T, the head's (rpb,fpb) table and
// the window's token descriptors are staged in LDS once (~76 KB -> two workgroups per CU).  Each
// wave owns 16-query tiles and computes the TRANSPOSED scores S^T = K * Q^T with
// v_mfma_f32_16x16x32_bf16 (one MFMA per 16x16 tile since head_dim == 32 == MFMA K), so a lane
// holds, for ONE query (lane & 15), 4 keys of every 16-key tile: the whole 392-long softmax row
// lives in 4 lanes' registers.  Row max/sum = in-lane reduction + two wavefront shuffles.
// Key order inside the tiles is permuted (tile t, MFMA row i <-> key 32*(t>>1)+8*(i>>2)+4*(t&1)+(i&3))
// so that after exp/convert the packed probabilities ARE the A-operand fragment of the P*V MFMA
// (lane group g holds keys 32s+8g..+7): no LDS round trip, no cross-lane permute for P.
#include "common.hpp"

namespace kvq {

constexpr int ATT_WAVES = 5;
constexpr int ATT_NT = 26;              // 16-key tiles -> up to 416 keys (N <= 400 supported, 392 used)
constexpr int ATT_KROWS = ATT_NT * 16;  // 416
constexpr int ATT_VPITCH = 400;         // bf16 per V^T row: 800 B = 50 16-B slots == 2 (mod 16) -> conflict-free b128
constexpr int ATT_VT_BYTES = 32 * ATT_VPITCH * 2 + 64;

struct AttnParams {
  const uint16_t* qkv;
  const int32_t* tok;
  const float* rpb;
  const float* fpb;
  int table_len, center, BW, nW, N, nH, use_mask;
  uint16_t* out;
};

__device__ __forceinline__ int k_slot(int row, int g) { return row * 4 + (g ^ ((-(row >> 3)) & 3)); }

template <typename E, bool GATED, bool MASK>
__global__ __launch_bounds__(ATT_WAVES * 64) void window_attention_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* Ks = reinterpret_cast<u32x4*>(smem);                                   // [416*4] 16-B slots
  uint16_t* Vt = reinterpret_cast<uint16_t*>(smem + ATT_KROWS * 64);            // [32][400] (+tail)
  int2* tokL = reinterpret_cast<int2*>(smem + ATT_KROWS * 64 + ATT_VT_BYTES);   // [416] {code, desc}
  float2* tab = reinterpret_cast<float2*>(smem + ATT_KROWS * 64 + ATT_VT_BYTES + ATT_KROWS * 8);

  const int tid = threadIdx.x;
  const int unit = blockIdx.x;
  const int bw = unit / p.nH, h = unit - bw * p.nH;
  const int N = p.N;
  const size_t Mtot = (size_t)p.BW * N;
  const int C = p.nH * 32;
  const uint16_t* Qg = p.qkv + ((size_t)(0 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
  const uint16_t* Kg = p.qkv + ((size_t)(1 * p.nH + h) * Mtot + (size_t)bw * N) * 32;
  const uint16_t* Vg = p.qkv + ((size_t)(2 * p.nH + h) * Mtot + (size_t)bw * N) * 32;

  // ---- stage K (swizzled rows), V^T, token descriptors and this head's bias tables ----
  for (int c = tid; c < ATT_KROWS * 4; c += ATT_WAVES * 64) {
    const int row = c >> 2, g = c & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < N) v = *reinterpret_cast<const u32x4*>(Kg + (size_t)row * 32 + g * 8);
    Ks[k_slot(row, g)] = v;
  }
  for (int c = tid; c < ATT_VPITCH * 4; c += ATT_WAVES * 64) {
    const int key = c >> 2, g = c & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (key < N) v = *reinterpret_cast<const u32x4*>(Vg + (size_t)key * 32 + g * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Vt[(g * 8 + 2 * i) * ATT_VPITCH + key] = (uint16_t)(v[i] & 0xffffu);
      Vt[(g * 8 + 2 * i + 1) * ATT_VPITCH + key] = (uint16_t)(v[i] >> 16);
    }
  }
  if (tid < 32) Vt[32 * ATT_VPITCH + tid] = 0;   // tail read by the last K-step of row 31
  const int w = bw % p.nW;
  for (int n = tid; n < ATT_KROWS; n += ATT_WAVES * 64) {
    int2 t = make_int2(0, 0);
    if (n < N) t = *reinterpret_cast<const int2*>(p.tok + ((size_t)w * N + n) * 2);
    tokL[n] = t;
  }
  for (int i = tid; i < p.table_len; i += ATT_WAVES * 64) {
    const float r = p.rpb[(size_t)i * p.nH + h];
    tab[i] = make_float2(r, GATED ? p.fpb[(size_t)i * p.nH + h] : 0.f);
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int nqt = (N + 15) >> 4;
  const float kLog2e = 1.4426950408889634f;

  for (int qt = wave; qt < nqt; qt += ATT_WAVES) {
    const int q0 = qt * 16;
    const int qrow = min(q0 + j, N - 1);
    // B operand of S^T = K Q^T: lane (j,g) holds Q[q0+j][8g..8g+7]
    using V8 = typename E::v8;
    const V8 qf = *reinterpret_cast<const V8*>(Qg + (size_t)qrow * 32 + g * 8);
    const int2 tq = tokL[qrow];
    const int cq = tq.x + p.center;
    const unsigned fq = (unsigned)tq.y & 0xffffu, rq = (unsigned)tq.y >> 16;

    f32x4 S[ATT_NT];
    // ---- scores + bias + mask, running max.  This lane's keys: 32*(t>>1) + 8g + 4*(t&1) + r ----
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < ATT_NT; ++t) {
      // A operand: MFMA row i = j  <->  key 32*(t>>1) + 8*(i>>2) + 4*(t&1) + (i&3)
      const int krow = 32 * (t >> 1) + 8 * (j >> 2) + 4 * (t & 1) + (j & 3);
      const V8 kf = __builtin_bit_cast(V8, Ks[k_slot(krow, g)]);
      S[t] = E::mfma16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r;
        const int2 tk = tokL[key];
        const float2 b2 = tab[cq - tk.x];
        float bias = b2.x;
        if (GATED) {
          const float gate = (float)__builtin_amdgcn_sad_u8(fq, (unsigned)tk.y & 0xffffu, 0u);
          bias = b2.x * gate + b2.y * (1.0f - gate);
        }
        float s = S[t][r] + bias;
        if (MASK) s += (((unsigned)tk.y >> 16) != rq) ? -100.0f : 0.0f;
        s = key < N ? s : -INFINITY;
        S[t][r] = s;
        mx = fmaxf(mx, s);
      }
      // keep the compiler from hoisting all 104 LDS gathers ahead of their use (that spills):
      __builtin_amdgcn_sched_barrier(0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // ---- exp, bf16 pack (the packed pairs are the P*V A-fragments), row sum of the ROUNDED values ----
    const float mb = mx * kLog2e;
    float sum = 0.f;
    uint32_t P[ATT_NT][2];
#pragma unroll
    for (int t = 0; t < ATT_NT; ++t) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = exp2f(S[t][r] * kLog2e - mb);
      P[t][0] = E::pack2(e[0], e[1]);
      P[t][1] = E::pack2(e[2], e[3]);
      sum += (E::to_f32((uint16_t)(P[t][0] & 0xffffu)) + E::to_f32((uint16_t)(P[t][0] >> 16))) +
             (E::to_f32((uint16_t)(P[t][1] & 0xffffu)) + E::to_f32((uint16_t)(P[t][1] >> 16)));
      __builtin_amdgcn_sched_barrier(0);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // ---- O = P V: 13 K-steps of 32 keys, two 16-wide feature tiles ----
    f32x4 O0 = {0.f, 0.f, 0.f, 0.f}, O1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ATT_NT / 2; ++s) {
      const u32x4 pa = {P[2 * s][0], P[2 * s][1], P[2 * s + 1][0], P[2 * s + 1][1]};
      const V8 pf = __builtin_bit_cast(V8, pa);
      const V8 v0 = *reinterpret_cast<const V8*>(Vt + j * ATT_VPITCH + 32 * s + 8 * g);
      const V8 v1 = *reinterpret_cast<const V8*>(Vt + (j + 16) * ATT_VPITCH + 32 * s + 8 * g);
      O0 = E::mfma16(pf, v0, O0);
      O1 = E::mfma16(pf, v1, O1);
    }
    // ---- normalise + store.  O layout: col = feature j (+16), row = query 4g + r ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = 4 * g + r;
      const float inv = 1.0f / __shfl(sum, qq);
      if (q0 + qq < N) {
        uint16_t* o = p.out + ((size_t)bw * N + q0 + qq) * C + h * 32 + j;
        o[0] = E::cvt(O0[r] * inv);
        o[16] = E::cvt(O1[r] * inv);
      }
    }
  }
}

template <typename E, bool GATED, bool MASK>
static int launch_attn(const AttnParams& p, size_t lds, hipStream_t st) {
  auto kern = window_attention_kernel<E, GATED, MASK>;
  static size_t attr_bytes = 0;   // per instantiation: opt in to > 64 KiB of dynamic LDS once
  if (lds > attr_bytes) {
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_bytes = lds;
  }
  dim3 grid((unsigned)(p.BW * p.nH)), block(ATT_WAVES * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, p);
  KVQ_CHECK_LAUNCH("window_attention_kernel");
  return KVQ_OK;
}

}  // namespace kvq

extern "C" int kvq_window_attention(const uint16_t* qkv, const int32_t* tok, const float* rpb, const float* fpb,
                                    int table_len, int center, int BW, int nW, int N, int num_heads, int use_mask,
                                    int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(qkv && tok && rpb && out, KVQ_ERR_NULL, "kvq_window_attention: NULL pointer");
  KVQ_REQUIRE(BW > 0 && nW > 0 && BW % nW == 0 && num_heads > 0 && table_len > 0, KVQ_ERR_SHAPE,
              "kvq_window_attention: bad shape BW=%d nW=%d nH=%d table_len=%d", BW, nW, num_heads, table_len);
  KVQ_REQUIRE(N >= 1 && N <= 400, KVQ_ERR_UNSUPPORTED,
              "kvq_window_attention: window of %d tokens unsupported (1..400)", N);
  const size_t lds = (size_t)ATT_KROWS * 64 + ATT_VT_BYTES + (size_t)ATT_KROWS * 8 + (size_t)table_len * 8;
  KVQ_REQUIRE(lds <= 160 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_window_attention: bias table of %d entries exceeds LDS",
              table_len);
  AttnParams p{qkv, tok, rpb, fpb, table_len, center, BW, nW, N, num_heads, use_mask, out};
  hipStream_t st = (hipStream_t)stream;
  const bool gated = fpb != nullptr, mask = use_mask != 0;
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_window_attention: dtype %d", dtype);
  if (dtype == KVQ_DT_FP16) {
    if (gated && mask) return launch_attn<Fp16, true, true>(p, lds, st);
    if (gated) return launch_attn<Fp16, true, false>(p, lds, st);
    if (mask) return launch_attn<Fp16, false, true>(p, lds, st);
    return launch_attn<Fp16, false, false>(p, lds, st);
  }
  if (gated && mask) return launch_attn<Bf16, true, true>(p, lds, st);
  if (gated) return launch_attn<Bf16, true, false>(p, lds, st);
  if (mask) return launch_attn<Bf16, false, true>(p, lds, st);
  return launch_attn<Bf16, false, false>(p, lds, st);
}
