// Fused post-attention half of a SwinTransformerBlock3D (gfx950), one launch per block:
//
//   x  <- x + window_reverse(roll(proj(attn_out)))                (swin_backbone.py:323, :472-488, :509)
//   x  <- x + fc2(GELU(fc1(norm2(x))))                            (:490-491, :514, Mlp :84-87)
//   [ ln_next <- window_partition(roll(norm1_next(x))) ]          (:416-449 of the NEXT block)
//
// Everything here is row-local, so the kernel is written TOKEN-PER-LANE: all GEMMs are computed transposed,
//   Out^T[channel][token] = W[channel][k] * In^T[k][token],
// with the weights as the MFMA A operand (from LDS) and 32 tokens as the 32 columns of a 32x32x16 MFMA.  In the
// C/D layout a lane then owns ONE token (column lane&31) and the channel rows (r&3) + 8*(r>>2) + 4*(lane>>5) of
// every 32-channel tile: the residual add, LayerNorm statistics (in-lane sums + one exchange with lane^32), bias
// and GELU all happen in registers, and an accumulator tile converts to the B operand of the next GEMM by a
// 16-bit pack alone — the k order of that operand is a fixed permutation of the channels, which is folded into
// the weights once on the host side of the boundary (kvq_block_tail_pack).  The hidden activations (4C per
// token), the proj output and norm2's output never exist in memory: per token the launch reads attn_out (2C B)
// and x (4C B) and writes x (4C B) [+ 2C B for the next norm1] instead of 48C B over five launches.
//
// Weights stream through a 3-slot LDS ring by LDS-DMA as "panels" (32 x C 16-bit: 32 output channels of proj,
// the W1 rows of 32 hidden units, or the W2 columns of the same 32 hidden units) already laid out fragment-major
// (one 1 KB wave-load = one k-step of A fragments, lane-linear => conflict-free ds_read_b128); a ring item is
// 1 or 2 panels (>= 12 KB).  One counted vmcnt wait + one raw barrier per item, as in gemm.hip.
#include "common.hpp"

namespace kvq {

struct TailParams {
  const uint16_t* attn;      // [M][C] 16-bit, window order
  float* x;                  // [n_batch*out_rows][C] fp32, in place
  const int32_t* map;        // window row -> token of the batch element (or <0 = padding); NULL = identity
  int map_rows, out_rows, M, hidden;
  const unsigned char* pack; // kvq_block_tail_pack image
  const float* nn_w;         // next block's norm1 (EMIT)
  const float* nn_b;
  const int32_t* next_dst;   // token -> window row of the next block's partition
  uint16_t* next_ln;         // [n_batch*next_rows][C]
  int next_rows;
  float eps;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int TAIL_NST = 3;

__host__ __device__ constexpr int tail_ppi(int C) { return 64 * C >= 12288 ? 1 : 2; }     // panels per ring item
__host__ __device__ constexpr int tail_slot_bytes(int C) { return tail_ppi(C) * 64 * C; }
__host__ __device__ constexpr int tail_proj_items(int C) { return (C / 32 + tail_ppi(C) - 1) / tail_ppi(C); }
static size_t tail_items(int C, int hidden) { return tail_proj_items(C) + (size_t)(hidden / 32) * 2 / tail_ppi(C); }
static size_t tail_param_bytes(int C, int hidden) { return (((size_t)(4 * C + hidden) * 4) + 4095) & ~(size_t)4095; }

// ---- weight image ---------------------------------------------------------------------------------------------
// panel = 64*C bytes = C/16 "fragment rows" of 64 lanes x 16 B.  Fragment row f of a panel holds, for lane
// (m = lane&31, h = lane>>5), the 8 k-values an MFMA A operand needs at that lane.  Panel sequence:
//   proj panel i (i < C/32):  f = s (k-step):                Wp[32i+m][16s + 8h + e]            (e = 0..7)
//   zero panels up to a whole number of ring items (tail_ppi(C) panels each), then per 32 hidden units j:
//   W1 panel j:               f = s:                         W1[32j+m][chan(s,h,e)]
//   W2 panel j:               f = 2i + t (t = 0,1):          W2[32i+m][32j + 8(2t + (e>>2)) + 4h + (e&3)]
//   chan(s,h,e) = 32(s>>1) + 8(2(s&1) + (e>>2)) + 4h + (e&3)   — the channel an accumulator register r = 8(s&1)+e
//   of tile s>>1 holds in lane half h (C/D layout of v_mfma_f32_32x32x16).
// After the panels: fp32 [proj_b C][norm2_w C][norm2_b C][fc2_b C][fc1_b hidden], padded to 4 KB.
__global__ void tail_pack_kernel(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b,
                                 const float* n2w, const float* n2b, const float* b1, const float* b2, int C, int hidden,
                                 unsigned char* out, long n_chunks, long n_par) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CM = C / 32, KS = C / 16, first_mlp = tail_proj_items(C) * tail_ppi(C);
  if (g < n_chunks) {
    const int panel = (int)(g / (KS * 64)), rem = (int)(g % (KS * 64));
    const int f = rem >> 6, lane = rem & 63, m = lane & 31, h = lane >> 5;
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint16_t val = 0;
      if (panel < CM) {
        val = wp[(size_t)(32 * panel + m) * C + 16 * f + 8 * h + e];
      } else if (panel >= first_mlp) {
        const int j = (panel - first_mlp) >> 1;
        if (((panel - first_mlp) & 1) == 0) {
          const int ch = 32 * (f >> 1) + 8 * (2 * (f & 1) + (e >> 2)) + 4 * h + (e & 3);
          val = w1[(size_t)(32 * j + m) * C + ch];
        } else {
          const int i = f >> 1, t = f & 1;
          val = w2[(size_t)(32 * i + m) * hidden + 32 * j + 8 * (2 * t + (e >> 2)) + 4 * h + (e & 3)];
        }
      }
      v[e] = val;
    }
    uint16_t* o = reinterpret_cast<uint16_t*>(out + g * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e];
  } else if (g < n_chunks + n_par) {
    const int q = (int)(g - n_chunks);
    float val = 0.f;
    if (q < C) val = proj_b[q];
    else if (q < 2 * C) val = n2w[q - C];
    else if (q < 3 * C) val = n2b[q - 2 * C];
    else if (q < 4 * C) val = b2[q - 3 * C];
    else if (q < 4 * C + hidden) val = b1[q - 4 * C];
    reinterpret_cast<float*>(out + n_chunks * 16)[q] = val;
  }
}

// ---- the kernel -----------------------------------------------------------------------------------------------
template <typename E, int CM, int TN, bool EMIT>
__global__ __launch_bounds__(256, 2) void block_tail_kernel(TailParams p) {
  constexpr int C = 32 * CM, KS = 2 * CM, PPI = tail_ppi(C), PANEL = 64 * C, SLOT = PPI * PANEL, LPW = SLOT / 4096;
  constexpr int NST = TAIL_NST, NPI = tail_proj_items(C);
  static_assert(SLOT % 4096 == 0, "an item is a whole number of 1 KB loads per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;
  float* prm = reinterpret_cast<float*>(lds + NST * SLOT);
  const float* s_pb = prm;
  const float* s_g2 = prm + C;
  const float* s_b2n = prm + 2 * C;
  const float* s_fb2 = prm + 3 * C;
  const float* s_fb1 = prm + 4 * C;

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NJ = p.hidden >> 5, NI = NPI + NJ * 2 / PPI;
  const int PL = (int)(((4 * C + p.hidden) * 4 + 4095) >> 12);      // 1 KB param loads per wave

  // oldest in the queue: the fp32 parameters, straight into LDS
  {
    const unsigned char* src = p.pack + (size_t)NI * SLOT;
    for (int l = 0; l < PL; ++l) {
      const int q = l * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + q * 1024 + lane * 16),
                                       (lds_ptr_t)(lds + NST * SLOT + q * 1024), 16, 0, 0);
    }
  }
  // this lane's tokens
  long orig[TN];
  bool live[TN];
  int tloc[TN], tb[TN];
  V8 bx[TN][KS];
  f32x16 acc[TN][CM];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const long row = (long)blockIdx.x * (128 * TN) + wave * (32 * TN) + t * 32 + (lane & 31);
    const long rc = row < p.M ? row : p.M - 1;
    int b = 0, s = (int)rc;
    if (p.map) {
      b = (int)(rc / p.map_rows);
      s = p.map[rc - (long)b * p.map_rows];
    } else {
      b = (int)(rc / p.out_rows);
      s = (int)(rc - (long)b * p.out_rows);
    }
    live[t] = row < p.M && s >= 0;
    tb[t] = b;
    tloc[t] = s < 0 ? 0 : s;
    orig[t] = (long)b * p.out_rows + tloc[t];
    const uint16_t* ar = p.attn + (size_t)rc * C + 8 * h;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) bx[t][s2] = *reinterpret_cast<const V8*>(ar + 16 * s2);
    const float* xr = p.x + (size_t)orig[t] * C + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 32 * i + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][i][4 * q + e] = v[e];
      }
  }

  auto issue = [&](int it) {
    unsigned char* dst = lds + (it % NST) * SLOT;
    const unsigned char* src = p.pack + (size_t)it * SLOT;   // it % NST: items are issued in order, slot = ring position
#pragma unroll
    for (int l = 0; l < LPW; ++l) {
      const int q = l * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + q * 1024 + lane * 16), (lds_ptr_t)(dst + q * 1024), 16, 0, 0);
    }
  };
  issue(0);
  issue(1);
  int it = 0, slot = 0;
  // item `it` visible to every wave; everybody has left item it-1, whose slot takes item it+NST-1
  auto next_item = [&]() -> const unsigned char* {
    if (it < NI - 1) wait_vmcnt<LPW>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (it + NST - 1 < NI) issue(it + NST - 1);
    const unsigned char* st = lds + slot * SLOT + lane * 16;
    ++it;
    slot = slot + 1 == NST ? 0 : slot + 1;
    return st;
  };

  // ---- proj: acc (= x) += Wp . attn^T ------------------------------------------------------------------------
  const unsigned char* st = nullptr;
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    if (i % PPI == 0) st = next_item();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const V8 a = *reinterpret_cast<const V8*>(st + (i % PPI) * PANEL + s * 1024);
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t][i] = E::mfma32(a, bx[t][s], acc[t][i]);
    }
  }
  // + proj bias (the parameters landed before item 0 and are visible since its barrier)
#pragma unroll
  for (int i = 0; i < CM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(s_pb + 32 * i + 8 * q + 4 * h);
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][i][4 * q + e] += b[e];
    }

  // ---- norm2 in registers: two-pass statistics, biased variance, eps inside the rsqrt (as ln.hip) -----------
  float mean[TN], rstd[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 4) s += (acc[t][i][r] + acc[t][i][r + 1]) + (acc[t][i][r + 2] + acc[t][i][r + 3]);
    s += __shfl_xor(s, 32);
    mean[t] = s / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[t][i][r] - mean[t];
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32);
    rstd[t] = rsqrtf(sq / (float)C + p.eps);
  }
  // normalised rows -> B operands of fc1 (k order = accumulator order, see tail_pack_kernel); x1 + fc2 bias stays in acc
#pragma unroll
  for (int i = 0; i < CM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(s_g2 + 32 * i + 8 * q + 4 * h);
      const f32x4 be = *reinterpret_cast<const f32x4*>(s_b2n + 32 * i + 8 * q + 4 * h);
      const f32x4 fb = *reinterpret_cast<const f32x4*>(s_fb2 + 32 * i + 8 * q + 4 * h);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = (acc[t][i][4 * q + e] - mean[t]) * rstd[t] * g[e] + be[e];
          acc[t][i][4 * q + e] += fb[e];
        }
        u32x4 w = __builtin_bit_cast(u32x4, bx[t][2 * i + (q >> 1)]);
        w[2 * (q & 1)] = E::pack2(y[0], y[1]);
        w[2 * (q & 1) + 1] = E::pack2(y[2], y[3]);
        bx[t][2 * i + (q >> 1)] = __builtin_bit_cast(V8, w);
      }
    }

  // ---- MLP: 32 hidden units per item; they live and die in registers -----------------------------------------
  for (int j = 0; j < NJ; ++j) {
    st = next_item();
    f32x16 hacc[TN];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(s_fb1 + 32 * j + 8 * q + 4 * h);
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) hacc[t][4 * q + e] = b[e];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const V8 a = *reinterpret_cast<const V8*>(st + s * 1024);
#pragma unroll
      for (int t = 0; t < TN; ++t) hacc[t] = E::mfma32(a, bx[t][s], hacc[t]);
    }
    V8 hb[TN][2];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          w[e] = E::pack2(gelu_fast(hacc[t][8 * tt + 2 * e]), gelu_fast(hacc[t][8 * tt + 2 * e + 1]));
        hb[t][tt] = __builtin_bit_cast(V8, w);
      }
    const unsigned char* st2 = PPI == 1 ? next_item() : st + PANEL;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const V8 a = *reinterpret_cast<const V8*>(st2 + (2 * i + tt) * 1024);
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t][i] = E::mfma32(a, hb[t][tt], acc[t][i]);
      }
  }

  // ---- write the residual stream back; optionally the next block's norm1 in ITS window order -----------------
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    if (!live[t]) continue;
    float* xr = p.x + (size_t)orig[t] * C + 4 * h;
#pragma unroll
    for (int i = 0; i < CM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(xr + 32 * i + 8 * q) =
            (f32x4){acc[t][i][4 * q], acc[t][i][4 * q + 1], acc[t][i][4 * q + 2], acc[t][i][4 * q + 3]};
  }
  if (EMIT) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 4) s += (acc[t][i][r] + acc[t][i][r + 1]) + (acc[t][i][r + 2] + acc[t][i][r + 3]);
      s += __shfl_xor(s, 32);
      const float mu = s / (float)C;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[t][i][r] - mu;
          sq += d * d;
        }
      sq += __shfl_xor(sq, 32);
      const float rs = rsqrtf(sq / (float)C + p.eps);
      if (!live[t]) continue;
      const long drow = (long)tb[t] * p.next_rows + p.next_dst[tloc[t]];
      uint16_t* o = p.next_ln + (size_t)drow * C + 4 * h;
#pragma unroll
      for (int i = 0; i < CM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(p.nn_w + 32 * i + 8 * q + 4 * h);
          const f32x4 be = *reinterpret_cast<const f32x4*>(p.nn_b + 32 * i + 8 * q + 4 * h);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (acc[t][i][4 * q + e] - mu) * rs * g[e] + be[e];
          *reinterpret_cast<u32x2*>(o + 32 * i + 8 * q) = (u32x2){E::pack2(y[0], y[1]), E::pack2(y[2], y[3])};
        }
    }
  }
}

template <typename E, int CM, int TN>
static int launch_tail(const TailParams& p, hipStream_t st) {
  constexpr int C = 32 * CM;
  const size_t lds = (size_t)TAIL_NST * tail_slot_bytes(C) + tail_param_bytes(C, p.hidden);
  KVQ_REQUIRE(lds <= 80 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: %zu B of LDS", lds);
  dim3 grid((unsigned)ceil_div(p.M, 128 * TN)), block(256);
  if (p.next_ln) {
    auto k = block_tail_kernel<E, CM, TN, true>;
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, st, p);
  } else {
    auto k = block_tail_kernel<E, CM, TN, false>;
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, st, p);
  }
  KVQ_CHECK_LAUNCH("block_tail_kernel");
  return KVQ_OK;
}

template <typename E>
static int launch_tail_e(const TailParams& p, int C, hipStream_t st) {
  switch (C) {
    case 96: return launch_tail<E, 3, 2>(p, st);
    case 128: return launch_tail<E, 4, 1>(p, st);
    case 192: return launch_tail<E, 6, 1>(p, st);
    default: break;
  }
  KVQ_REQUIRE(false, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d not in {96,128,192}", C);
}

}  // namespace kvq

extern "C" int kvq_block_tail_supported(int C, int hidden) {
  return (C == 96 || C == 128 || C == 192) && hidden % 32 == 0 && hidden > 0 ? 1 : 0;
}

extern "C" size_t kvq_block_tail_pack_bytes(int C, int hidden) {
  if (!kvq_block_tail_supported(C, hidden)) return 0;
  return kvq::tail_items(C, hidden) * kvq::tail_slot_bytes(C) + kvq::tail_param_bytes(C, hidden);
}

extern "C" int kvq_block_tail_pack(const void* proj_w, const float* proj_b, const float* norm2_w, const float* norm2_b,
                                   const void* fc1_w, const float* fc1_b, const void* fc2_w, const float* fc2_b, int C,
                                   int hidden, void* pack, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(proj_w && proj_b && norm2_w && norm2_b && fc1_w && fc1_b && fc2_w && fc2_b && pack, KVQ_ERR_NULL,
              "kvq_block_tail_pack: NULL pointer");
  KVQ_REQUIRE(kvq_block_tail_supported(C, hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail_pack: C=%d hidden=%d", C, hidden);
  const long n_chunks = (long)tail_items(C, hidden) * tail_slot_bytes(C) / 16;
  const long n_par = (long)tail_param_bytes(C, hidden) / 4;
  const long total = n_chunks + n_par;
  hipLaunchKernelGGL(tail_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)proj_w, (const uint16_t*)fc1_w, (const uint16_t*)fc2_w, proj_b, norm2_w, norm2_b,
                     fc1_b, fc2_b, C, hidden, (unsigned char*)pack, n_chunks, n_par);
  KVQ_CHECK_LAUNCH("tail_pack_kernel");
  return KVQ_OK;
}

extern "C" int kvq_block_tail(const KvqBlockTailArgs* a, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(a && a->attn && a->x && a->pack, KVQ_ERR_NULL, "kvq_block_tail: NULL pointer");
  KVQ_REQUIRE(kvq_block_tail_supported(a->C, a->hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d hidden=%d", a->C,
              a->hidden);
  KVQ_REQUIRE(a->M > 0 && a->out_rows > 0 && (!a->scatter_map || a->map_rows > 0), KVQ_ERR_SHAPE, "kvq_block_tail: bad rows");
  KVQ_REQUIRE(!a->next_ln || (a->next_norm_w && a->next_norm_b && a->next_dst && a->next_rows > 0), KVQ_ERR_NULL,
              "kvq_block_tail: next_ln without its norm / map");
  KVQ_REQUIRE(a->dtype == KVQ_DT_BF16 || a->dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_block_tail: dtype %d", a->dtype);
  TailParams p{};
  p.attn = (const uint16_t*)a->attn; p.x = a->x; p.map = a->scatter_map; p.map_rows = a->map_rows; p.out_rows = a->out_rows;
  p.M = a->M; p.hidden = a->hidden; p.pack = (const unsigned char*)a->pack;
  p.nn_w = a->next_norm_w; p.nn_b = a->next_norm_b; p.next_dst = a->next_dst; p.next_ln = (uint16_t*)a->next_ln;
  p.next_rows = a->next_rows; p.eps = a->eps;
  return a->dtype == KVQ_DT_FP16 ? launch_tail_e<Fp16>(p, a->C, (hipStream_t)stream)
                                 : launch_tail_e<Bf16>(p, a->C, (hipStream_t)stream);
}
