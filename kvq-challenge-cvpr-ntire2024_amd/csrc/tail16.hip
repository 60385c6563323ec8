// The fused post-attention launch of csrc/tail.hip for WIDE rows (C = 384: stage 2 of Swin-T/S), on 16x16x32 MFMAs.
//
// Same computation and the same token-per-lane idea (all GEMMs transposed, weights = MFMA A operand from an LDS ring,
// a lane owns one token; residual, LayerNorm, bias and GELU in registers; see tail.hip), but a wave carries 16 tokens
// instead of 32: in the 16x16 C/D layout a lane (token n = lane & 15, group g = lane >> 4) holds rows 4g..4g+3 of
// every 16-channel tile, so the fp32 state of a token's C channels is C/4 registers per lane whatever the tile — 96
// at C = 384, where the 32-token layout of tail.hip would need 192.  Price: a 1 KB A fragment now feeds a 16-cycle MFMA
// (LDS read bandwidth ~ the MFMA rate), and a workgroup (4 waves = 64 tokens) streams all 2.65 MB of proj/fc1/fc2
// once — the same L2 -> LDS volume the three GEMMs of the un-fused chain move, without their x / hidden round trips,
// their epilogues and the two LayerNorm launches.
//
// k orders: B operands built from accumulators take, at position (g, e) of k-step s, channel 32s + 4g + e (e < 4, tile
// 2s) or 32s + 16 + 4g + e - 4 (tile 2s+1); fc1's and fc2's k columns are permuted accordingly in the packed image.
// Ring items are 48 KB = 48 fragment rows: 4 proj tiles x 12 k-steps; {W1 of chunks 0, 1}; then per 32-unit chunk j
// {W2: 24 output tiles x 1 k-step | W1 of chunk j+2: 2 tiles x 12 k-steps} — the software pipeline of tail.hip.
#include "common.hpp"
#include "tail.hpp"

namespace kvq {

typedef __attribute__((address_space(3))) void* lds_ptr16_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr16_t;

constexpr int T16_SLOT = 48 * 1024, T16_NST = 3;     // 3 x 48 KB + 15 KB of parameters = 159 of the 160 KB

bool tail16_supported(int C, int hidden) { return C == 384 && hidden == 4 * C; }

static size_t t16_items(int C, int hidden) { return (size_t)(C / 16) / 4 + 1 + hidden / 32; }
static size_t t16_param_bytes(int C, int hidden) { return (((size_t)(4 * C + hidden) * 4) + 4095) & ~(size_t)4095; }
size_t tail16_pack_bytes(int C, int hidden) {
  return tail16_supported(C, hidden) ? t16_items(C, hidden) * T16_SLOT + t16_param_bytes(C, hidden) : 0;
}

// fragment row = 64 lanes x 16 B: lane (m = lane & 15, g = lane >> 4) holds A[m][8 k-values of group g]
__global__ void tail16_pack_kernel(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b,
                                   const float* n2w, const float* n2b, const float* b1, const float* b2, int C, int hidden,
                                   unsigned char* out, long n_chunks, long n_par) {
  const long gi = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KS = C / 32, CT = C / 16, NJ = hidden / 32, NPI = CT / 4;
  if (gi < n_chunks) {
    const int item = (int)(gi / (48 * 64)), rem = (int)(gi % (48 * 64));
    const int f = rem >> 6, lane = rem & 63, m = lane & 15, g = lane >> 4;
    uint16_t* o = reinterpret_cast<uint16_t*>(out + gi * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // accumulator-order k of a 32-wide k-step: (g, e) -> offset inside the step
      const int kperm = e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
      uint16_t val = 0;
      if (item < NPI) {                                   // proj: tile 4*item + f/KS, k-step f%KS, natural k
        const int tile = 4 * item + f / KS, s = f % KS;
        val = wp[(size_t)(16 * tile + m) * C + 32 * s + 8 * g + e];
      } else {
        int w1_chunk = -1, w1_f = 0, w2_chunk = -1;
        if (item == NPI) {                                // {W1 chunk 0 | W1 chunk 1}
          w1_chunk = f / (2 * KS); w1_f = f % (2 * KS);
        } else {
          const int j = item - NPI - 1;
          if (f < CT) w2_chunk = j;
          else if (j + 2 < NJ) { w1_chunk = j + 2; w1_f = f - CT; }
        }
        if (w2_chunk >= 0) {                              // fragment row f = output tile f
          val = w2[(size_t)(16 * f + m) * hidden + 32 * w2_chunk + kperm];
        } else if (w1_chunk >= 0) {                       // rows: tile t (16 hidden units) x k-step s
          const int t = w1_f / KS, s = w1_f % KS;
          val = w1[(size_t)(32 * w1_chunk + 16 * t + m) * C + 32 * s + kperm];
        }
      }
      o[e] = val;
    }
  } else if (gi < n_chunks + n_par) {
    const int q = (int)(gi - n_chunks);
    float val = 0.f;
    if (q < C) val = proj_b[q];
    else if (q < 2 * C) val = n2w[q - C];
    else if (q < 3 * C) val = n2b[q - 2 * C];
    else if (q < 4 * C) val = b2[q - 3 * C];
    else if (q < 4 * C + hidden) val = b1[q - 4 * C];
    reinterpret_cast<float*>(out + n_chunks * 16)[q] = val;
  }
}

int tail16_pack(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b, const float* n2w, const float* n2b,
                const float* b1, const float* b2, int C, int hidden, unsigned char* out, hipStream_t st) {
  const long n_chunks = (long)t16_items(C, hidden) * T16_SLOT / 16, n_par = (long)t16_param_bytes(C, hidden) / 4;
  const long total = n_chunks + n_par;
  hipLaunchKernelGGL(tail16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wp, w1, w2, proj_b, n2w, n2b, b1,
                     b2, C, hidden, out, n_chunks, n_par);
  KVQ_CHECK_LAUNCH("tail16_pack_kernel");
  return KVQ_OK;
}

template <typename E, int CT, bool EMIT>      // CT = C / 16 channel tiles
__global__ __launch_bounds__(256, 1) void block_tail16_kernel(TailParams p) {
  fp16_saturate_mode();
  constexpr int C = 16 * CT, KS = C / 32, NPI = CT / 4, SLOT = T16_SLOT, NST = T16_NST, LPW = SLOT / 4096;
  static_assert(CT == 2 * KS && 4 * KS == 48 && CT + 2 * KS == 48, "48 fragment rows per item");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  using V8 = typename E::v8;
  float* prm = reinterpret_cast<float*>(lds + NST * SLOT);
  const float* s_g2 = prm + C;
  const float* s_b2n = prm + 2 * C;
  const float* s_fb2 = prm + 3 * C;
  const float* s_fb1 = prm + 4 * C;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NJ = p.hidden >> 5, NI = NPI + 1 + NJ;
  const int NQ = ((4 * C + p.hidden) * 4 + 1023) >> 10;
  float* s_nn = reinterpret_cast<float*>(lds + NST * SLOT + NQ * 1024);

#ifdef KVQ_TAIL_TRACE   // diagnostic build only (tools/tail_trace.py)
  const bool tr = p.trace && tid == 0 && (int)blockIdx.x < p.trace_blocks;
  unsigned long long wait_dma = 0, wait_bar = 0;
#define KVQ_STAMP16(i) if (tr) p.trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter()
#else
#define KVQ_STAMP16(i)
#endif
  KVQ_STAMP16(0);
  f32x4 nn_reg = {0.f, 0.f, 0.f, 0.f};
  if (EMIT && tid < C / 2) nn_reg = *reinterpret_cast<const f32x4*>((tid < C / 4 ? p.nn_w : p.nn_b - C) + 4 * tid);
  {
    const unsigned char* src = p.pack + (size_t)NI * SLOT;
    for (int q = wave; q < NQ; q += 4)
      __builtin_amdgcn_global_load_lds((gbl_ptr16_t)(src + q * 1024 + lane * 16), (lds_ptr16_t)(lds + NST * SLOT + q * 1024), 16, 0, 0);
  }
  // this lane's token
  const long row = (long)blockIdx.x * 64 + wave * 16 + (lane & 15);
  const long rc = row < p.M ? row : p.M - 1;
  int tb, tloc;
  if (p.map) {
    tb = (int)(rc / p.map_rows);
    tloc = p.map[rc - (long)tb * p.map_rows];
  } else {
    tb = (int)(rc / p.out_rows);
    tloc = (int)(rc - (long)tb * p.out_rows);
  }
  const bool live = row < p.M && tloc >= 0;
  tloc = tloc < 0 ? 0 : tloc;
  const long orig = (long)tb * p.out_rows + tloc;
  V8 bx[KS];
  f32x4 acc[CT];
  {
    const uint16_t* ar = p.attn + (size_t)rc * C + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) bx[s] = *reinterpret_cast<const V8*>(ar + 32 * s);
    const float* xr = p.x + (size_t)orig * C + 4 * g;
    const float* pbg = reinterpret_cast<const float*>(p.pack + (size_t)NI * SLOT) + 4 * g;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 16 * i);
      const f32x4 b = *reinterpret_cast<const f32x4*>(pbg + 16 * i);
      acc[i] = v + b;
    }
  }
  auto issue = [&](int item, int slot) {
    unsigned char* dst = lds + slot * SLOT;
    const unsigned char* src = p.pack + (size_t)item * SLOT;
#pragma unroll
    for (int l = 0; l < LPW; ++l) {
      const int q = l * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr16_t)(src + q * 1024 + lane * 16), (lds_ptr16_t)(dst + q * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue(i, i);
  if (EMIT && tid < C / 2) *reinterpret_cast<f32x4*>(s_nn + 4 * tid) = nn_reg;
  int it = 0, slot = 0;
  const unsigned char* pend_src = nullptr;
  unsigned char* pend_dst = nullptr;
  // item `it` has landed (only item it+1 may be younger), everybody has left item it-1, whose slot takes item it+NST-1
  auto next_item = [&]() -> const unsigned char* {
#ifdef KVQ_TAIL_TRACE
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    if (NST == 3 && it + 1 < NI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef KVQ_TAIL_TRACE
    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef KVQ_TAIL_TRACE
    wait_dma += t1 - t0;
    wait_bar += __builtin_readcyclecounter() - t1;
#endif
    const int fill = slot == 0 ? NST - 1 : slot - 1;
    // the refill of the freed slot is issued in pieces between the MFMA groups of this item (stream()): a burst of
    // LPW LDS-DMA instructions here costs ~50 cycles apiece with the fragment reads right behind it
    pend_src = it + NST - 1 < NI ? p.pack + (size_t)(it + NST - 1) * SLOT + lane * 16 : nullptr;
    pend_dst = lds + fill * SLOT;
    const unsigned char* st = lds + slot * SLOT + lane * 16;
    ++it;
    slot = slot + 1 == NST ? 0 : slot + 1;
    return st;
  };

  KVQ_STAMP16(1);
  // The A fragments of an item (48 x 1 KB) are fetched G = 8 ahead of their MFMAs: with one wave per SIMD nothing else
  // hides the LDS latency (a 1-deep prefetch ran the loop at 74 ticks per 16-cycle MFMA).  between(grp) is issued after
  // the loads of group grp+1 and before the MFMAs of group grp — VALU work placed there covers the first group's latency.
  auto stream = [&](const unsigned char* sp, int nk, auto&& row, auto&& mm, auto&& between) __attribute__((always_inline)) {
    constexpr int G = 8;
    V8 a[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) a[0][i] = *reinterpret_cast<const V8*>(sp + row(i) * 1024);
#pragma unroll
    for (int grp = 0; grp < 48 / G; ++grp) {
      if ((grp + 1) * G < nk) {
#pragma unroll
#ifdef T16_NOLDS
        for (int i = 0; i < G; ++i) { a[(grp + 1) & 1][i] = a[grp & 1][i]; asm volatile("" : "+v"(a[(grp + 1) & 1][i])); }
#else
        for (int i = 0; i < G; ++i) a[(grp + 1) & 1][i] = *reinterpret_cast<const V8*>(sp + row((grp + 1) * G + i) * 1024);
#endif
      }
      if (pend_src) {
#pragma unroll
        for (int l = grp * LPW / (48 / G); l < (grp + 1) * LPW / (48 / G); ++l) {
          const int q = l * 4 + wave;
          __builtin_amdgcn_global_load_lds((gbl_ptr16_t)(pend_src + q * 1024), (lds_ptr16_t)(pend_dst + q * 1024), 16, 0, 0);
        }
      }
      between(grp);
      if (grp * G < nk) {
#pragma unroll
        for (int i = 0; i < G; ++i) mm(grp * G + i, a[grp & 1][i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- proj: acc (= x + bias) += Wp . attn^T, 4 output tiles per item ----------------------------------------------
#pragma unroll
  for (int pi = 0; pi < NPI; ++pi) {
    const unsigned char* st = next_item();
    // MFMAs on one accumulator are 4 apart (back-to-back dependent issue stalls the pipe)
    stream(st, 48, [](int k) { return (k % 4) * KS + k / 4; },
           [&](int k, V8 a) { acc[4 * pi + k % 4] = E::mfma16(a, bx[k / 4], acc[4 * pi + k % 4]); }, [&](int) {});
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- norm2 in registers: in-lane sums + exchanges with the three other lane groups of the token ------------------
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[i][r] - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq / (float)C + p.eps);
    float mean_n = mean;
    asm volatile("" : "+v"(mean_n));   // opaque copy: no CSE of (acc - mean) with the variance pass
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {  // k-step s2 = tiles 2*s2 (e < 4) and 2*s2+1 (e >= 4)
      u32x4 w;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int i = 2 * s2 + hh;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(s_g2 + 16 * i + 4 * g);
        const f32x4 be = *reinterpret_cast<const f32x4*>(s_b2n + 16 * i + 4 * g);
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (acc[i][r] - mean_n) * rstd * gm[r] + be[r];
        w[2 * hh] = E::pack2(y[0], y[1]);
        w[2 * hh + 1] = E::pack2(y[2], y[3]);
      }
      bx[s2] = __builtin_bit_cast(V8, w);
      if (s2 & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  KVQ_STAMP16(2);

  // ---- MLP, software-pipelined over chunks of 32 hidden units (two 16-unit tiles) ----------------------------------
  auto h_init = [&](f32x4 (&ha)[2], int j) {
#pragma unroll
    for (int t = 0; t < 2; ++t) ha[t] = *reinterpret_cast<const f32x4*>(s_fb1 + 32 * j + 16 * t + 4 * g);
  };
  auto gelu_pair = [&](const f32x4 (&ha)[2], u32x4& hb, int q) {       // q = 0..3: values (2q, 2q+1) of the 8
#ifdef KVQ_TAIL_NOGELU      // experiment: the loop without its VALU stream
    uint32_t w = E::pack2_raw(fmaxf(ha[q >> 1][2 * (q & 1)], 0.f), fmaxf(ha[q >> 1][2 * (q & 1) + 1], 0.f));
#else
    uint32_t w = E::pack2(gelu_fast(ha[q >> 1][2 * (q & 1)]), gelu_fast(ha[q >> 1][2 * (q & 1) + 1]));
#endif
    asm volatile("" : "+v"(w));      // pins the evaluation between two MFMAs (see tail.hip)
    hb[q] = w;
  };
  f32x4 haA[2], haB[2];
  u32x4 hbA, hbB;
  {
    const unsigned char* st = next_item();       // {W1 chunk 0 | W1 chunk 1}
    h_init(haA, 0);
    h_init(haB, 1);
    stream(st, 48, [](int k) { return (k % 4) * KS + k / 4; }, [&](int k, V8 a) {
      if (k % 4 < 2) haA[k % 4] = E::mfma16(a, bx[k / 4], haA[k % 4]);
      else haB[k % 4 - 2] = E::mfma16(a, bx[k / 4], haB[k % 4 - 2]);
    }, [&](int) {});
#pragma unroll
    for (int q = 0; q < 4; ++q) gelu_pair(haA, hbA, q);
  }
  // one step: Y += W2_j . hb_cur ; ha_cur <- fc1(chunk j+2) ; hb_nxt <- GELU(ha_nxt)    (ha_nxt = chunk j+1)
  auto step = [&](int j, f32x4 (&ha_cur)[2], u32x4& hb_cur, f32x4 (&ha_nxt)[2], u32x4& hb_nxt, bool do_h,
                  bool do_g) __attribute__((always_inline)) {
    const unsigned char* sp = next_item();
    const V8 b = __builtin_bit_cast(V8, hb_cur);
    if (do_h) h_init(ha_cur, j + 2);
    // fc2 (24 independent accumulators) and the two fc1 chains alternate: MFMAs on one fc1 accumulator are 4 apart
    if (do_h) {
      stream(sp, 48, [](int k) { return k % 2 == 0 ? k / 2 : CT + ((k / 2) % 2) * KS + k / 4; }, [&](int k, V8 a) {
        if (k % 2 == 0) acc[k / 2] = E::mfma16(a, b, acc[k / 2]);
        else ha_cur[(k / 2) % 2] = E::mfma16(a, bx[k / 4], ha_cur[(k / 2) % 2]);
      }, [&](int grp) {
        if (do_g && grp < 4) gelu_pair(ha_nxt, hb_nxt, grp);
      });
    } else {
      stream(sp, CT, [](int k) { return k; }, [&](int k, V8 a) { acc[k] = E::mfma16(a, b, acc[k]); }, [&](int grp) {
        if (do_g && grp < 4) gelu_pair(ha_nxt, hb_nxt, grp);
      });
    }
  };
  int j = 0;
  for (; j + 2 < NJ; j += 2) {
    step(j, haA, hbA, haB, hbB, true, true);
    step(j + 1, haB, hbB, haA, hbA, true, true);
  }
  step(j, haA, hbA, haB, hbB, false, true);
  step(j + 1, haB, hbB, haA, hbA, false, false);
  KVQ_STAMP16(3);

  // ---- + fc2 bias; write the residual stream back; optionally the next block's norm1 in ITS window order -----------
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    acc[i] += *reinterpret_cast<const f32x4*>(s_fb2 + 16 * i + 4 * g);
    if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
  if (live) {
    float* xr = p.x + (size_t)orig * C + 4 * g;
#pragma unroll
    for (int i = 0; i < CT; ++i) *reinterpret_cast<f32x4*>(xr + 16 * i) = acc[i];
  }
  if (EMIT) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mu = s / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[i][r] - mu;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rs = rsqrtf(sq / (float)C + p.eps);
    float mu_n = mu;
    asm volatile("" : "+v"(mu_n));
    if (live) {
      const long drow = (long)tb * p.next_rows + p.next_dst[tloc];
      uint16_t* o = p.next_ln + (size_t)drow * C + 4 * g;
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(s_nn + 16 * i + 4 * g);
        const f32x4 be = *reinterpret_cast<const f32x4*>(s_nn + C + 16 * i + 4 * g);
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (acc[i][r] - mu_n) * rs * gm[r] + be[r];
        *reinterpret_cast<u32x2*>(o + 16 * i) = (u32x2){E::pack2(y[0], y[1]), E::pack2(y[2], y[3])};
        if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#ifdef KVQ_TAIL_TRACE
  if (tr) {
    p.trace[blockIdx.x * 8 + 5] = wait_dma;
    p.trace[blockIdx.x * 8 + 6] = wait_bar;
  }
#endif
  KVQ_STAMP16(4);
}

template <typename E>
static int launch16(const TailParams& p, hipStream_t st) {
  constexpr int C = 384;
  const size_t lds = (size_t)T16_NST * T16_SLOT + ((((size_t)(4 * C + p.hidden) * 4) + 1023) & ~(size_t)1023) + (size_t)2 * C * 4;
  dim3 grid((unsigned)ceil_div(p.M, 64)), block(256);
  if (p.next_ln) {
    auto k = block_tail16_kernel<E, C / 16, true>;
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, st, p);
  } else {
    auto k = block_tail16_kernel<E, C / 16, false>;
    KVQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, block, lds, st, p);
  }
  KVQ_CHECK_LAUNCH("block_tail16_kernel");
  return KVQ_OK;
}

int tail16_launch(const TailParams& p, int C, int dtype, hipStream_t st) {
  KVQ_REQUIRE(tail16_supported(C, p.hidden), KVQ_ERR_UNSUPPORTED, "kvq_block_tail: C=%d hidden=%d", C, p.hidden);
  return dtype == KVQ_DT_FP16 ? launch16<Fp16>(p, st) : launch16<Bf16>(p, st);
}

}  // namespace kvq
