// Convolution front-ends for the 2D ResNet-50 (SimpleVQA spatial branch, simpleVQA_model.py:220-264)
// and the SlowFast-R50 3D-conv motion branch (SlowFast_features.py:137-165): activations are
// channels-last 16-bit (N,D,H,W,C); a convolution is an im2col gather (HBM-bound) feeding the MFMA GEMM
// of gemm.hip (BatchNorm folded into weight/bias on the host, ReLU / residual add in the GEMM epilogue);
// 1x1x1 stride-1 convolutions skip the gather entirely.  Pooling and the SimpleVQA mean/std pooling are
// plain HBM-bound reductions.
#include <type_traits>

#include "common.hpp"

namespace kvq {

struct Im2colParams {
  const void* x;
  int src_f32;                       // 1: fp32 source (network input), 0: 16-bit source of type E
  long sb, sc, sd, sh, sw;           // element strides of the source
  int B, C, D, H, W;
  int kd, kh, kw, sdd, shh, sww, pd, ph, pw;
  int Do, Ho, Wo, K, Kpad;
  uint16_t* out;                     // [B*Do*Ho*Wo][Kpad], column order (kd,kh,kw,c), zero padded to Kpad
};

template <typename E>
__global__ __launch_bounds__(256) void im2col_nd_kernel(Im2colParams p) {
  fp16_saturate_mode();
  const int chunks = p.Kpad / 8;
  const long total = (long)p.B * p.Do * p.Ho * p.Wo * chunks;
  const bool vec = !p.src_f32 && (p.C % 8 == 0) && p.sc == 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks);
    long r = i / chunks;
    const int wo = (int)(r % p.Wo); r /= p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int dO = (int)(r % p.Do);
    const int b = (int)(r / p.Do);
    u32x4 o = {0u, 0u, 0u, 0u};
    const int col0 = ch * 8;
    if (vec) {
      if (col0 < p.K) {
        const int c = col0 % p.C;
        int t = col0 / p.C;
        const int kw = t % p.kw; t /= p.kw;
        const int kh = t % p.kh;
        const int kd = t / p.kh;
        const int d = dO * p.sdd - p.pd + kd, h = ho * p.shh - p.ph + kh, w = wo * p.sww - p.pw + kw;
        if (d >= 0 && d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W)
          o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.x) + b * p.sb + d * p.sd + h * p.sh +
                                              w * p.sw + c);
      }
    } else {
      uint16_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int col = col0 + k;
        float f = 0.f;
        uint16_t raw = 0;
        if (col < p.K) {
          const int c = col % p.C;
          int t = col / p.C;
          const int kw = t % p.kw; t /= p.kw;
          const int kh = t % p.kh;
          const int kd = t / p.kh;
          const int d = dO * p.sdd - p.pd + kd, h = ho * p.shh - p.ph + kh, w = wo * p.sww - p.pw + kw;
          if (d >= 0 && d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W) {
            const long off = b * p.sb + c * p.sc + d * p.sd + h * p.sh + w * p.sw;
            if (p.src_f32) f = reinterpret_cast<const float*>(p.x)[off];
            else raw = reinterpret_cast<const uint16_t*>(p.x)[off];
          }
        }
        v[k] = p.src_f32 ? E::cvt(f) : raw;
      }
      o = (u32x4){(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                  (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
    }
    reinterpret_cast<u32x4*>(p.out)[i] = o;
  }
}

struct PoolParams {
  const uint16_t* x;     // (B,D,H,W,C) channels-last 16-bit
  uint16_t* out;         // (B,Do,Ho,Wo,C)
  int B, C, D, H, W, kd, kh, kw, sdd, shh, sww, pd, ph, pw, Do, Ho, Wo, is_max;
  int ldc, coff;         // output rows of ldc channels, this pool's C at channel coff (kvq_pool_nd_strided; else C / 0)
};

// max: padding never wins (-inf); avg: divides by the full window (count_include_pad=True, the
// nn.AvgPool3d default; the SlowFast head pools use no padding anyway).
template <typename E>
__global__ __launch_bounds__(256) void pool_nd_kernel(PoolParams p) {
  const long total = (long)p.B * p.Do * p.Ho * p.Wo * p.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % p.C);
    long r = i / p.C;
    const int wo = (int)(r % p.Wo); r /= p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int dO = (int)(r % p.Do);
    const int b = (int)(r / p.Do);
    float acc = p.is_max ? -INFINITY : 0.f;
    for (int kd = 0; kd < p.kd; ++kd)
      for (int kh = 0; kh < p.kh; ++kh)
        for (int kw = 0; kw < p.kw; ++kw) {
          const int d = dO * p.sdd - p.pd + kd, h = ho * p.shh - p.ph + kh, w = wo * p.sww - p.pw + kw;
          if (d < 0 || d >= p.D || h < 0 || h >= p.H || w < 0 || w >= p.W) continue;
          const float v = E::to_f32(p.x[((((size_t)b * p.D + d) * p.H + h) * p.W + w) * p.C + c]);
          acc = p.is_max ? fmaxf(acc, v) : acc + v;
        }
    if (!p.is_max) acc /= (float)(p.kd * p.kh * p.kw);
    p.out[(i / p.C) * p.ldc + p.coff + c] = E::cvt(acc);
  }
}

// fp32 frames addressed through element strides (b, t, c, h, w) -> 16-bit channels-last (B*T, H, W, 8), channels >= C zero:
// the operand layout the implicit-GEMM conv wants for the 3-channel network input (one 16-byte chunk per pixel).
template <typename E>
__global__ __launch_bounds__(256) void pack_cl8_kernel(const float* __restrict__ x, int T, int Cc, int H, int W, long sb, long st,
                                                       long sc, long sh, long sw, uint16_t* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int w = (int)(i % W);
  long r = i / W;
  const int h = (int)(r % H);
  const long n = r / H;
  const float* src = x + (n / T) * sb + (n % T) * st + (long)h * sh + (long)w * sw;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = c < Cc ? src[c * sc] : 0.f;
  *reinterpret_cast<u32x4*>(out + i * 8) = (u32x4){E::pack2(v[0], v[1]), E::pack2(v[2], v[3]), E::pack2(v[4], v[5]), E::pack2(v[6], v[7])};
}

// C % 8 == 0: a thread owns 8 channels of an output position (16-byte loads / stores; same fp32 arithmetic and order)
template <typename E>
__global__ __launch_bounds__(256) void pool_nd_vec8_kernel(PoolParams p) {
  const int C8 = p.C / 8;
  const long total = (long)p.B * p.Do * p.Ho * p.Wo * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8) * 8;
    long r = i / C8;
    const int wo = (int)(r % p.Wo); r /= p.Wo;
    const int ho = (int)(r % p.Ho); r /= p.Ho;
    const int dO = (int)(r % p.Do);
    const int b = (int)(r / p.Do);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = p.is_max ? -INFINITY : 0.f;
    for (int kd = 0; kd < p.kd; ++kd)
      for (int kh = 0; kh < p.kh; ++kh)
        for (int kw = 0; kw < p.kw; ++kw) {
          const int d = dO * p.sdd - p.pd + kd, h = ho * p.shh - p.ph + kh, w = wo * p.sww - p.pw + kw;
          if (d < 0 || d >= p.D || h < 0 || h >= p.H || w < 0 || w >= p.W) continue;
          const u32x4 v = *reinterpret_cast<const u32x4*>(p.x + ((((size_t)b * p.D + d) * p.H + h) * p.W + w) * p.C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = E::to_f32((uint16_t)(v[e] & 0xffffu)), hi = E::to_f32((uint16_t)(v[e] >> 16));
            acc[2 * e] = p.is_max ? fmaxf(acc[2 * e], lo) : acc[2 * e] + lo;
            acc[2 * e + 1] = p.is_max ? fmaxf(acc[2 * e + 1], hi) : acc[2 * e + 1] + hi;
          }
        }
    const float cnt = (float)(p.kd * p.kh * p.kw);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = p.is_max ? E::pack2(acc[2 * e], acc[2 * e + 1])
                      : E::pack2(acc[2 * e] / cnt, acc[2 * e + 1] / cnt);
    *reinterpret_cast<u32x4*>(p.out + (i / C8) * p.ldc + p.coff + c) = o;
  }
}

// SimpleVQA pooling (simpleVQA_model.py:8-11, 242-252): per (frame, channel) mean and UNBIASED std over
// the H*W positions of a channels-last map; two passes in fp32.  grid (rows, ceil(C/64)), block 256 =
// 64 channels x 4 position groups.
template <typename E, int CH>
__global__ __launch_bounds__(256) void mean_std_pool_kernel(const uint16_t* __restrict__ x, int HW, int C,
                                                            float* __restrict__ out, long out_stride, int mean_off,
                                                            int std_off) {
  // CH channels x GR = 256 / CH position groups per block: CH = 64 for many rows (one frame each), CH = 16 when a few rows
  // pool over thousands of positions (KSVQE's Dist_Transformation3: 4 rows x 3136) and the grid would not fill the chip
  constexpr int GR = 256 / CH;
  __shared__ float red[GR][CH];
  const int cl = threadIdx.x % CH, grp = threadIdx.x / CH;
  const int row = blockIdx.x, c = blockIdx.y * CH + cl;
  const bool live = c < C;
  const uint16_t* xr = x + (size_t)row * HW * C;
  float s = 0.f;
  if (live)
    for (int i = grp; i < HW; i += GR) s += E::to_f32(xr[(size_t)i * C + c]);
  red[grp][cl] = s;
  __syncthreads();
  float tot = 0.f;
  if (GR == 4) {
    tot = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
  } else {
#pragma unroll
    for (int g = 0; g < GR; ++g) tot += red[g][cl];
  }
  const float mean = tot / (float)HW;
  __syncthreads();
  float q = 0.f;
  if (live && std_off >= 0)
    for (int i = grp; i < HW; i += GR) {
      const float d = E::to_f32(xr[(size_t)i * C + c]) - mean;
      q += d * d;
    }
  red[grp][cl] = q;
  __syncthreads();
  if (grp == 0 && live) {
    float ss = 0.f;
    if (GR == 4) {
      ss = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    } else {
#pragma unroll
      for (int g = 0; g < GR; ++g) ss += red[g][cl];
    }
    out[(size_t)row * out_stride + mean_off + c] = mean;
    if (std_off >= 0) out[(size_t)row * out_stride + std_off + c] = sqrtf(ss / (float)(HW - 1));
  }
}


// Few rows, thousands of positions, C % 8 == 0: a block owns 8 channels, a thread strides over the positions with 16-byte loads
// (KSVQE's Dist_Transformation3 at 96 frames: 1 row x 9408 positions x 384 channels -> 48 workgroups of 37 loads per thread)
template <typename E>
__global__ __launch_bounds__(256) void mean_std_pool_vec8_kernel(const uint16_t* __restrict__ x, int HW, int C,
                                                                 float* __restrict__ out, long out_stride, int mean_off,
                                                                 int std_off) {
  __shared__ float red[4][8];
  const int row = blockIdx.x, c0 = blockIdx.y * 8, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* xr = x + (size_t)row * HW * C + c0;
  auto block_sum8 = [&](float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v[e] += __shfl_xor(v[e], o);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  };
  auto load8 = [&](int i, float (&f)[8]) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + (size_t)i * C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = E::to_f32((uint16_t)(v[e] & 0xffffu));
      f[2 * e + 1] = E::to_f32((uint16_t)(v[e] >> 16));
    }
  };
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, f[8];
  for (int i = tid; i < HW; i += 256) {
    load8(i, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += f[e];
  }
  block_sum8(s);
  float mean[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = s[e] / (float)HW;
    q[e] = 0.f;
  }
  if (std_off >= 0) {                                    // mean-only callers (the SlowFast head pools) read the map once
    for (int i = tid; i < HW; i += 256) {
      load8(i, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = f[e] - mean[e];
        q[e] = fmaf(d, d, q[e]);
      }
    }
    block_sum8(q);
  }
  if (tid < 8) {
    out[(size_t)row * out_stride + mean_off + c0 + tid] = mean[tid];
    if (std_off >= 0) out[(size_t)row * out_stride + std_off + c0 + tid] = sqrtf(q[tid] / (float)(HW - 1));
  }
}

// ---- direct stem convolution ---------------------------------------------------------------------------------------
// The SlowFast fast-pathway stem is Conv3d(3 -> 8, k = 5x7x7, stride 1x2x2): K = 735 per output but only 8 output
// channels, so as im2col + GEMM it WRITES a 4.7 GB patch matrix (3.2 M positions x 736 x 2 B: 7.6 ms of the 15 ms
// forward) to feed an MFMA tile that is 3/4 padding.  Here a thread owns one output position and all Cout <= 16
// channels: fp32 FMAs against weights broadcast from LDS ([K][Cout] fp32), the fp32 clip read through L1/L2 (adjacent
// threads share 5/7 of their taps), folded-BN bias + ReLU, one 16-B channels-last store.  fp32 end to end.
struct StemParams {
  const float* x;          // (B, C, D, H, W) contiguous fp32
  const float* w;          // [K][Cout] fp32, K ordered (kd, kh, kw, c) like kvq_im2col_nd
  const float* bias;       // [Cout]
  int B, C, D, H, W, kd, kh, kw, sd, sh, sw, pd, ph, pw, Do, Ho, Wo, relu;
  uint16_t* out;           // (B, Do, Ho, Wo, Cout) 16-bit
};

template <typename E, int COUT>
__global__ __launch_bounds__(256) void conv_stem_direct_kernel(StemParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);
  const int K = p.kd * p.kh * p.kw * p.C;
  for (int i = threadIdx.x; i < K * COUT; i += 256) wl[i] = p.w[i];
  __syncthreads();
  const long total = (long)p.B * p.Do * p.Ho * p.Wo;
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= total) return;
  const int wo = (int)(pos % p.Wo), ho = (int)((pos / p.Wo) % p.Ho), dq = (int)((pos / ((long)p.Wo * p.Ho)) % p.Do);
  const int b = (int)(pos / ((long)p.Wo * p.Ho * p.Do));
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = p.bias[o];
  const size_t plane = (size_t)p.H * p.W, vol = plane * p.D;
  const float* xb = p.x + (size_t)b * p.C * vol;
  const int x0 = wo * p.sw - p.pw, y0 = ho * p.sh - p.ph, t0 = dq * p.sd - p.pd;
  for (int a = 0; a < p.kd; ++a) {
    const int t = t0 + a;
    if (t < 0 || t >= p.D) continue;
    for (int r = 0; r < p.kh; ++r) {
      const int y = y0 + r;
      if (y < 0 || y >= p.H) continue;
      const float* row = xb + (size_t)t * plane + (size_t)y * p.W;
      const float* wrow = wl + (size_t)((a * p.kh + r) * p.kw) * p.C * COUT;
      for (int c2 = 0; c2 < p.kw; ++c2) {
        const int xx = x0 + c2;
        if (xx < 0 || xx >= p.W) continue;
        for (int c = 0; c < p.C; ++c) {
          const float v = row[(size_t)c * vol + xx];
          const float* wv = wrow + (c2 * p.C + c) * COUT;
#pragma unroll
          for (int o = 0; o < COUT; o += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wv + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o + e] = fmaf(v, w4[e], acc[o + e]);
          }
        }
      }
    }
  }
  uint16_t* o16 = p.out + (size_t)pos * COUT;
#pragma unroll
  for (int o = 0; o < COUT; o += 8) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p.relu ? fmaxf(acc[o + e], 0.f) : acc[o + e];
    *reinterpret_cast<u32x4*>(o16 + o) =
        (u32x4){E::pack2(v[0], v[1]), E::pack2(v[2], v[3]), E::pack2(v[4], v[5]), E::pack2(v[6], v[7])};
  }
}

// ---- MFMA stem convolution for few output channels ------------------------------------------------------------------
// The fast-pathway stem again (3 -> 8 channels, 5x7x7, stride 1x2x2), on the matrix cores: the fp32 direct kernel above is
// bound by its loads (24 TFLOP/s).  Here the clip is packed once to 16-bit channels-last with FOUR channels and a zero border
// of 4 pixels left and right (B, T, H, W + 8, 4): the 8 taps (7 + a zero-weight one) x 4 channels of one kernel ROW are 32
// consecutive k values = one v_mfma_f32_16x16x32 slice, and a lane's 8 k values (2 adjacent taps x 4 channels) are 16
// contiguous bytes of the packed clip — no patch matrix; a wave stages the row segment it needs in LDS and reads the fragments there.
// C[channel][position]: A = weights [16 rows (8 used)][32 k] per (kd, kh) from LDS, B = 16 adjacent output columns.  A wave
// owns one output row (up to 7 tiles of 16 columns), so a weight fragment is read once per 7 MFMAs.
template <typename E>
__global__ __launch_bounds__(256) void pack_cl4_border_kernel(const float* __restrict__ x, int Cc, int T, int H, int W, int border,
                                                              uint16_t* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Wp = W + 2 * border;
  const int xp = (int)(i % Wp);
  long r = i / Wp;
  const int y = (int)(r % H); r /= H;
  const int t = (int)(r % T);
  const long b = r / T;
  const int xx = xp - border;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (xx >= 0 && xx < W) {
    const float* src = x + ((b * Cc * T + t) * (long)H + y) * W + xx;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < Cc) v[c] = src[(long)c * T * H * W];
  }
  *reinterpret_cast<u32x2*>(out + i * 4) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
}

struct StemMfmaParams {
  const uint16_t* x4;      // (B, T, H, W + 8, 4) 16-bit, zero border
  const uint16_t* wp;      // [kd*kh][16][32] 16-bit: k = tap * 4 + c, rows >= Cout and tap 7 zero
  const float* bias;       // [8]
  int B, T, H, Wp, kd, kh, sd, sh, pd, ph, Do, Ho, Wo, relu;
  uint16_t* out;           // (B, Do, Ho, Wo, 8)
};

template <typename E>
__global__ __launch_bounds__(256) void conv_stem_mfma_kernel(StemMfmaParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
  const int nsl = p.kd * p.kh, tid = threadIdx.x;
  for (int i = tid; i < nsl * 64; i += 256) *reinterpret_cast<u32x4*>(wl + i * 8) = *reinterpret_cast<const u32x4*>(p.wp + (size_t)i * 8);
  __syncthreads();
  const long rows = (long)p.B * p.Do * p.Ho, row = (long)blockIdx.x * 4 + (tid >> 6);
  if (row >= rows) return;
  const int lane = tid & 63, n = lane & 15, kg = lane >> 4;
  const int ho = (int)(row % p.Ho), dq = (int)((row / p.Ho) % p.Do);
  const long b = row / ((long)p.Ho * p.Do);
  constexpr int NT = 7;
  using v8 = typename E::v8;
  // per-wave staging of the input row segment (256 pixels x 4 channels = 2 KB): the 16 x 8-tap patches of a tile overlap 3.4x,
  // so fetching them per lane from global memory makes the texture-address path the bound (14 loads per kernel row and wave);
  // here a wave fetches the segment once (2 coalesced 16-byte loads per lane) and the fragments come out of LDS
  uint16_t* seg = wl + (size_t)nsl * 512 + (tid >> 6) * 1024;
  // the taps inside the clip form a box [a_lo, a_hi) x [r_lo, r_hi)
  const int t0 = dq * p.sd - p.pd, y0 = ho * p.sh - p.ph;
  const int a_lo = max(0, -t0), a_hi = min(p.kd, p.T - t0), r_lo = max(0, -y0), r_hi = min(p.kh, p.H - y0);
  const int nr = r_hi - r_lo, ns = (a_hi - a_lo) * nr;
  for (int w0 = 0; w0 < p.Wo; w0 += 16 * NT) {
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's two pixel pairs of the segment.  The segment starts at the ODD pixel 2*w0 + 1 of the bordered row (the first
    // tap of output column w0), so that every fragment below is a 16-byte ALIGNED LDS read; the global side then sits on 8-byte
    // boundaries and is fetched as two 8-byte halves
    const int px0 = 2 * w0 + 1 + 2 * lane, px1 = px0 + 128;
    auto pair = [&](const uint16_t* rowp, int px) -> u32x4 {
      if (px + 1 >= p.Wp) return (u32x4){0u, 0u, 0u, 0u};
      const u32x2 lo = *reinterpret_cast<const u32x2*>(rowp + (size_t)px * 4), hi = *reinterpret_cast<const u32x2*>(rowp + (size_t)px * 4 + 4);
      return (u32x4){lo[0], lo[1], hi[0], hi[1]};
    };
    auto gload = [&](int sidx, u32x4& v0, u32x4& v1) {
      const int a = a_lo + sidx / nr, r = r_lo + sidx % nr;
      const uint16_t* rowp = p.x4 + (((b * p.T + t0 + a) * (long)p.H + y0 + r) * p.Wp) * 4;
      v0 = pair(rowp, px0);
      v1 = pair(rowp, px1);
    };
    u32x4 g0 = {0u, 0u, 0u, 0u}, g1 = {0u, 0u, 0u, 0u};
    if (ns > 0) gload(0, g0, g1);
    for (int sidx = 0; sidx < ns; ++sidx) {
      __builtin_amdgcn_wave_barrier();                                 // the previous row's fragment reads are issued
      *reinterpret_cast<u32x4*>(seg + lane * 8) = g0;
      *reinterpret_cast<u32x4*>(seg + 512 + lane * 8) = g1;
      __builtin_amdgcn_wave_barrier();
      if (sidx + 1 < ns) gload(sidx + 1, g0, g1);                      // in flight under this row's MFMAs
      const int a = a_lo + sidx / nr, r = r_lo + sidx % nr;
      const v8 wa = *reinterpret_cast<const v8*>(wl + ((a * p.kh + r) * 16 + n) * 32 + kg * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        // pixel 2*wo - 3 + 2*kg of the image = 2*wo + 1 + 2*kg of the bordered row = 2*(j*16 + n) + 2*kg of the segment
        const u32x4 raw = *reinterpret_cast<const u32x4*>(seg + (size_t)(2 * (j * 16 + n) + 2 * kg) * 4);
        acc[j] = E::mfma16(wa, __builtin_bit_cast(v8, raw), acc[j]);
      }
    }
    if (kg < 2) {                                            // lanes 0..31 hold channels kg*4 .. +3 of position n
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + kg * 4);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int wo = w0 + j * 16 + n;
        if (wo >= p.Wo) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[j][e] + bv[e];
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<u32x2*>(p.out + ((size_t)row * p.Wo + wo) * 8 + kg * 4) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
      }
    }
  }
}

// elementwise max of eight 16-bit values (one 16-byte channel chunk)
template <typename E>
__device__ __forceinline__ u32x4 max8(u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, Fp16>::value) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b)));
  } else {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      r[e] = pack_bf2(fmaxf(bf2f((uint16_t)(a[e] & 0xffffu)), bf2f((uint16_t)(b[e] & 0xffffu))), fmaxf(bf2f((uint16_t)(a[e] >> 16)), bf2f((uint16_t)(b[e] >> 16))));
    return r;
  }
}

// ---- the fast-pathway stem as ONE launch: Conv3d(3 -> 8, (kd,7,7), stride (1,2,2), pad (kd/2,3,3)) + folded BN + ReLU +
// MaxPool3d((1,3,3), stride (1,2,2), pad (0,1,1)) straight from the fp32 clip (SlowFast_features.py:137-165, pytorchvideo's
// create_slowfast stem).  Round 2 ran it as pack (66 us, HBM-bound) + conv_stem_mfma_kernel (224 us: a wave per output row, every
// input row staged 17 times chip-wide, half of every MFMA tile's rows padding) + pool (20 us) for 3 % of the network's flops.  Here:
// * a workgroup owns (clip, frame, 4 pooled rows) = 9 stem rows; per temporal tap it stages the 23 input rows those need ONCE
//   (fp32 planes read with 16-byte loads, converted and interleaved to 4-channel 16-bit pixels in LDS, zero border);
// * stem rows are computed in PAIRS (ho, ho+1): input row y feeds row ho through kernel row kh and row ho+1 through kh-2, so
//   the A operand stacks W[kd][kh] (rows 0-7) on W[kd][kh-2] (rows 8-15): all 16 MFMA rows carry output channels and a pair
//   needs 9 input-row fragments instead of 2 x 7;
// * a wave owns 16-column tiles and walks the 23 staged rows once per tile: one 16-byte fragment read feeds every pair the row
//   belongs to (2 for most rows), 23 LDS reads for 43 MFMAs - the kernel is LDS-read bound, so this is what sets its time;
// * the 9 x Wo x 8 stem tile lands in LDS (bias, ReLU, 16-bit), the 3 x 3 / 2 max-pool reads it and writes 16 bytes per pooled
//   position: the stem tensor (51 MB per 8 clips) never exists in HBM.
// 8 clips of 32 x 224 x 224: 125 us against 285 + 20 (profiles/r03_stem_pool_pmc.txt).  Phases alone: staging 79 us (a chain of five
// load latencies per workgroup at 2 x 4 waves per CU), MFMAs + barriers 75 us; HBM reads = the clip once (FETCH_SIZE 153 MB).
struct StemPoolParams {
  const float* x;          // (B, 3, T, H, W) fp32
  const uint16_t* wp;      // [kd*7][16][32] 16-bit (kvq_conv_stem_mfma's image: k = tap * 4 + c, rows >= 8 and tap 7 zero)
  const float* bias;       // [8]
  int B, T, H, W, kd, Ho, Wo, Hp, Wp, relu;
  uint16_t* out;           // (B, T, Hp, Wp, 8)
};
constexpr int SP_PR = 4, SP_SR = 2 * SP_PR + 1, SP_NP = (SP_SR + 1) / 2, SP_NR = 2 * (SP_SR - 1) + 7;     // pooled rows, stem rows, row pairs, staged input rows

template <typename E>
__global__ __launch_bounds__(256, 2) void conv_stem_pool_kernel(StemPoolParams p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using v8 = typename E::v8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, kg = lane >> 4;
  const int PXP = p.W + 8;                                         // LDS pixel q <-> image column q - 3; 3 zero columns left, 5 right
  uint16_t* rows = reinterpret_cast<uint16_t*>(smem);              // [SP_NR][PXP][4]
  uint16_t* stem = rows + (size_t)SP_NR * PXP * 4;                 // [SP_SR][Wo][8]
  uint16_t* wl = stem + (size_t)SP_SR * p.Wo * 8;                  // [kd*7][8][32]: the weight image's real rows (an L2 round trip per tap otherwise)
  for (int i = tid; i < p.kd * 7 * 8 * 4; i += 256)
    *reinterpret_cast<u32x4*>(wl + (size_t)i * 8) = *reinterpret_cast<const u32x4*>(p.wp + ((size_t)(i >> 5) * 16 + ((i >> 2) & 7)) * 32 + (i & 3) * 8);
  // block -> (clip, row block, frame), frames fastest, each XCD (block id % 8) a contiguous eighth of that order: the five blocks
  // that read a frame's rows (frames t-2 .. t+2 of one row block) run side by side under ONE L2, so the clip crosses the fabric
  // about once instead of five times (round-robin placement measured 155 us for 8 clips: 1.1 GB of L2 misses).
  const int nrb = (p.Hp + SP_PR - 1) / SP_PR, total = p.B * nrb * p.T, chunk = (total + 7) >> 3;
  const int wid = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunk || wid >= total) return;
  const int t = wid % p.T, pr0 = ((wid / p.T) % nrb) * SP_PR, b = wid / (p.T * nrb);
  const int s0 = 2 * pr0 - 1, y_base = 2 * s0 - 3;                 // first stem row / first staged input row (negative at the top edge)
  for (int i = tid; i < SP_NR * 2; i += 256) {                     // zero border columns (never overwritten)
    uint16_t* rp = rows + (size_t)(i >> 1) * PXP * 4;
    if (i & 1) {
#pragma unroll
      for (int e = 0; e < 5; ++e) *reinterpret_cast<u32x2*>(rp + (size_t)(p.W + 3 + e) * 4) = (u32x2){0u, 0u};
    } else {
#pragma unroll
      for (int e = 0; e < 3; ++e) *reinterpret_cast<u32x2*>(rp + e * 4) = (u32x2){0u, 0u};
    }
  }
  const int nct = (p.Wo + 15) >> 4;                                // 16-column tiles (<= 8: two per wave)
  f32x4 acc[2][SP_NP];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int q = 0; q < SP_NP; ++q) acc[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int pd = p.kd / 2;
  const size_t plane = (size_t)p.T * p.H * p.W;
  // staging items of this thread: (staged row, 4 image columns) -> 3 x 16-byte loads, 4 x 8-byte LDS pixels.  The items do not depend
  // on the temporal tap, and the NEXT tap's loads are in flight while this tap's MFMAs run (one HBM latency per tap otherwise).
  // Dword loads (one pixel per lane, conflict-free stores by construction) were tried: 4 x the load instructions made it slower.
  constexpr int ITEMS = (SP_NR * 64 + 255) / 256;                  // W <= 256
  const int qw = p.W >> 2;
  int src_off[ITEMS], dst_off[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int i = tid + 256 * j, r = i / qw, q = i - r * qw, y = y_base + r;
    dst_off[j] = i < SP_NR * qw ? (r * PXP + 4 * q + 3) * 4 : -1;
    src_off[j] = (i < SP_NR * qw && y >= 0 && y < p.H) ? y * p.W + 4 * q : -1;
  }
  f32x4 pf[ITEMS][3];
  auto issue = [&](int tt) {
    const float* x0 = p.x + ((size_t)b * 3 * p.T + tt) * p.H * p.W;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      pf[j][0] = pf[j][1] = pf[j][2] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (src_off[j] >= 0) {
        const float* src = x0 + src_off[j];
        pf[j][0] = *reinterpret_cast<const f32x4*>(src);
        pf[j][1] = *reinterpret_cast<const f32x4*>(src + plane);
        pf[j][2] = *reinterpret_cast<const f32x4*>(src + 2 * plane);
      }
    }
  };
  // a lane's four pixels sit 32 bytes from its neighbour's: stored in lane order every 16-lane group of a ds_write_b64 hits 8 of the
  // 32 banks (4-way: measured as HALF of all LDS cycles of this kernel).  Store pixel e ^ k in pass e, k = (lane / 4) % 4: the four
  // lanes quads of a group then cover the four bank octets.
  const int kx = (lane >> 2) & 3;
  const int a_lo = max(0, pd - t), a_hi = min(p.kd, p.T + pd - t);    // taps inside the clip (block-uniform)
  issue(t - pd + a_lo);
  for (int a = a_lo; a < a_hi; ++a) {
    __syncthreads();                                               // the previous tap's fragment reads are done
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
      if (dst_off[j] >= 0) {
        u32x2 px[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) px[e] = (u32x2){E::pack2(pf[j][0][e], pf[j][1][e]), E::pack2(pf[j][2][e], 0.f)};
#pragma unroll
        for (int sw = 1; sw <= 2; sw <<= 1) {
          const bool on = (kx & sw) != 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(e & sw)) {
              const u32x2 lo = px[e], hi = px[e | sw];
              px[e] = on ? hi : lo;
              px[e | sw] = on ? lo : hi;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x2*>(rows + dst_off[j] + (e ^ kx) * 4) = px[e];
      }
    if (a + 1 < a_hi) issue(t - pd + a + 1);
    __syncthreads();
    // stacked weight fragments of this temporal tap for the input-row offset kk = 0..8 inside a pair: rows 0-7 = W[a][kk], rows 8-15 = W[a][kk-2]
    v8 wa[9];
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
      const int kh = kk - 2 * (n >> 3);
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (kh >= 0 && kh < 7) raw = *reinterpret_cast<const u32x4*>(wl + ((a * 7 + kh) * 8 + (n & 7)) * 32 + kg * 8);
      wa[kk] = __builtin_bit_cast(v8, raw);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ct = wave + 4 * u;
      if (ct < nct) {
        const uint16_t* base = rows + (size_t)(2 * (16 * ct + n + kg)) * 4;       // pixel 2 (wo + kg): taps 2 kg, 2 kg + 1 of column wo
        // row fragments in groups of RG, the next group's reads issued before this group's MFMAs (left to itself the compiler keeps
        // ONE ds_read in flight ahead of two MFMAs: 23 LDS latencies per tile and tap, which was 3/4 of the kernel's time)
        constexpr int RG = 6, NG = (SP_NR + RG - 1) / RG;
        v8 bf[2][RG];
        auto rd = [&](int g, v8* dst) {
#pragma unroll
          for (int e = 0; e < RG; ++e)
            if (g * RG + e < SP_NR) dst[e] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(base + (size_t)(g * RG + e) * PXP * 4));
        };
        rd(0, bf[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) rd(g + 1, bf[(g + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < RG; ++e) {
            const int r = g * RG + e;
#pragma unroll
            for (int q = 0; q < SP_NP; ++q) {
              const int kk = r - 4 * q;                                           // staged row r = 4 q + kk of pair q
              if (r < SP_NR && kk >= 0 && kk < 9) acc[u][q] = E::mfma16(wa[kk], bf[g & 1][e], acc[u][q]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // stem tile: lane (column n, group kg) holds channels 4 (kg & 1) .. +3 of stem row s0 + 2 pair + (kg >> 1)
  const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + (kg & 1) * 4);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int wo = 16 * (wave + 4 * u) + n;
    if (wo < p.Wo) {
#pragma unroll
      for (int q = 0; q < SP_NP; ++q) {
        const int sr = 2 * q + (kg >> 1);
        if (sr < SP_SR) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[u][q][e] + bv[e];
            if (p.relu) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<u32x2*>(stem + ((size_t)sr * p.Wo + wo) * 8 + (kg & 1) * 4) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
        }
      }
    }
  }
  __syncthreads();
  // max-pool 3 x 3 / 2, pad 1 (positions outside the stem map do not take part); 8 channels = 16 bytes per thread
  for (int i = tid; i < SP_PR * p.Wp; i += 256) {
    const int prl = i / p.Wp, pc = i - prl * p.Wp, pr = pr0 + prl;
    if (pr >= p.Hp) continue;
    bool any = false;
    u32x4 best = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int sy = 2 * pr - 1 + dy;
      if (sy < 0 || sy >= p.Ho) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int c = 2 * pc - 1 + dx;
        if (c < 0 || c >= p.Wo) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stem + ((size_t)(sy - s0) * p.Wo + c) * 8);
        best = any ? max8<E>(best, v) : v;
        any = true;
      }
    }
    *reinterpret_cast<u32x4*>(p.out + ((((size_t)b * p.T + t) * p.Hp + pr) * p.Wp + pc) * 8) = best;
  }
}

// ---- the slow-pathway stem as ONE launch: frame selection + Conv3d(3 -> 64, (1,7,7), stride (1,2,2), pad (0,3,3)) + folded BN + ReLU
// + MaxPool3d((1,3,3), (1,2,2), (0,1,1)) from the fp32 clip (SlowFast_features.py:112-165: pack_pathway_output's index_select, then
// block 0 of the slow pathway).  It ran as select_t (17 us) + pack + implicit GEMM (110 us, 103 MB stem map written) + pool (37 us, the
// map read back) per 8 clips for 9 us of MFMA work.  A workgroup of 14 waves owns (clip, selected frame, 2 pooled rows) = 5 stem rows:
// 15 input rows staged once as 4-channel 16-bit pixels (as conv_stem_pool_kernel), wave = (16-column tile, 32-channel half); each row
// fragment feeds the 3-4 stem rows it belongs to.  Pool: vertical max in registers (max commutes with the bias add, ReLU and the
// 16-bit rounding, all monotonic), the 2 x Wo x 64 result through LDS (8-byte slots XOR-swizzled by column), horizontal max and a
// 16-byte store into the caller's channel slice.  8 clips (1792 items): 50 us against 164; by removal: MFMA + vertical max 28 us (40 % MFMA
// utilisation - 14 waves in lockstep between two barriers per item), row loads 7, pool stores 7 (128 of every 160-byte pixel), staging 3.5.
struct Stem64Params {
  const float* x;          // (B, 3, T, H, W) fp32
  const int32_t* t_index;  // device, F selected frames (NULL: frames 0 .. F-1)
  const uint16_t* wimg;    // [7][64][32] 16-bit: [kh][o][kw * 4 + c], kw 7 and c 3 zero
  const float* bias;       // [64]
  int B, T, H, W, F, Ho, Wo, Hp, Wp, relu, out_C, out_coff;
  uint16_t* out;           // (B, F, Hp, Wp, out_C), channels out_coff .. out_coff + 63
};
constexpr int S6_PR = 2, S6_SR = 2 * S6_PR + 1, S6_NR = 2 * (S6_SR - 1) + 7, S6_THREADS = 896;
constexpr int S6_ITEMS = (S6_NR * 56 + S6_THREADS - 1) / S6_THREADS;             // (row, 4 columns) staging items per thread at W <= 224

template <typename E>
__global__ __launch_bounds__(S6_THREADS) void conv_stem64_pool_kernel(Stem64Params p) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using v8 = typename E::v8;
  typedef __attribute__((address_space(3))) void* s6_lds_t;
  typedef const __attribute__((address_space(1))) void* s6_gbl_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, kg = lane >> 4;
  const int PXP = p.W + 8;
  uint16_t* rows = reinterpret_cast<uint16_t*>(smem);              // [S6_NR][PXP][4]
  uint16_t* vm = rows + (size_t)S6_NR * PXP * 4;                   // [S6_PR][Wo][64], 8-byte slot s of column c at slot s ^ ((c & 7) << 1)
  float* raw = reinterpret_cast<float*>(vm + (size_t)S6_PR * p.Wo * 64);       // [3][S6_ITEMS][S6_THREADS][4] fp32: rows in flight
  uint16_t* wl = reinterpret_cast<uint16_t*>(raw + (size_t)3 * S6_ITEMS * S6_THREADS * 4);         // [7][64][32] weight image
  // persistent workgroups: one per CU walks its (clip, frame, row block) items, rows two items ahead in flight - a workgroup of 14
  // waves fills a CU alone, so one item per workgroup exposed an HBM latency, the weight fetch and a dispatch per item.  The rows
  // travel global -> LDS without passing through registers (global_load_lds, 16 bytes per lane), and the weight image sits in LDS
  // with a wave reading 7 fragments per channel tile: anything held in registers across the item loop spilled (128 per wave here).
  const int nrb = (p.Hp + S6_PR - 1) / S6_PR, total = p.B * p.F * nrb, step = gridDim.x;
  const size_t plane = (size_t)p.T * p.H * p.W;
  const int qw = p.W >> 2;
  int dst_off[S6_ITEMS], src_row[S6_ITEMS], src_col[S6_ITEMS];
#pragma unroll
  for (int j = 0; j < S6_ITEMS; ++j) {
    const int i = tid + S6_THREADS * j, r = i / qw, q = i - r * qw;
    dst_off[j] = i < S6_NR * qw ? (r * PXP + 4 * q + 3) * 4 : -1;
    src_row[j] = r;
    src_col[j] = 4 * q;
  }
  auto issue = [&](int it) {                                       // fp32 rows of item `it`: global -> raw
    const int rb = it % nrb, f = (it / nrb) % p.F, b = it / (nrb * p.F);
    const int tt = p.t_index ? p.t_index[f] : f;
    const float* x0 = p.x + ((size_t)b * 3 * p.T + tt) * p.H * p.W;
    const int yb = 2 * (2 * rb * S6_PR - 1) - 3;
#pragma unroll
    for (int j = 0; j < S6_ITEMS; ++j) {
      const int y = yb + src_row[j];
      if (dst_off[j] >= 0 && y >= 0 && y < p.H) {
        const float* src = x0 + (size_t)y * p.W + src_col[j];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          __builtin_amdgcn_global_load_lds((s6_gbl_t)(src + c * plane), (s6_lds_t)(raw + ((size_t)(c * S6_ITEMS + j) * S6_THREADS + wave * 64) * 4), 16, 0, 0);
      }
    }
  };
  const int kx = (lane >> 2) & 3;                                  // store pixel e ^ kx in pass e: conflict-free ds_write_b64 (see conv_stem_pool_kernel)
  auto commit = [&](int it) {                                      // raw -> 4-channel 16-bit pixels in `rows` (zero rows outside the image)
    const int yb = 2 * (2 * (it % nrb) * S6_PR - 1) - 3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // a wave reads back the raw slots its own lanes loaded: no barrier
#pragma unroll
    for (int j = 0; j < S6_ITEMS; ++j)
      if (dst_off[j] >= 0) {
        const int y = yb + src_row[j];
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
        if (y >= 0 && y < p.H) {
          c0 = *reinterpret_cast<const f32x4*>(raw + ((size_t)(0 * S6_ITEMS + j) * S6_THREADS + tid) * 4);
          c1 = *reinterpret_cast<const f32x4*>(raw + ((size_t)(1 * S6_ITEMS + j) * S6_THREADS + tid) * 4);
          c2 = *reinterpret_cast<const f32x4*>(raw + ((size_t)(2 * S6_ITEMS + j) * S6_THREADS + tid) * 4);
        }
        u32x2 px[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) px[e] = (u32x2){E::pack2(c0[e], c1[e]), E::pack2(c2[e], 0.f)};
#pragma unroll
        for (int sw = 1; sw <= 2; sw <<= 1) {
          const bool on = (kx & sw) != 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(e & sw)) {
              const u32x2 lo = px[e], hi = px[e | sw];
              px[e] = on ? hi : lo;
              px[e | sw] = on ? lo : hi;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x2*>(rows + dst_off[j] + (e ^ kx) * 4) = px[e];
      }
  };
  int item = blockIdx.x;
  if (item >= total) return;
  issue(item);
  for (int i = tid; i < 7 * 64 * 4; i += S6_THREADS) *reinterpret_cast<u32x4*>(wl + (size_t)i * 8) = *reinterpret_cast<const u32x4*>(p.wimg + (size_t)i * 8);
  for (int i = tid; i < S6_NR * 2; i += S6_THREADS) {              // zero border columns (never overwritten)
    uint16_t* rp = rows + (size_t)(i >> 1) * PXP * 4;
    if (i & 1) {
#pragma unroll
      for (int e = 0; e < 5; ++e) *reinterpret_cast<u32x2*>(rp + (size_t)(p.W + 3 + e) * 4) = (u32x2){0u, 0u};
    } else {
#pragma unroll
      for (int e = 0; e < 3; ++e) *reinterpret_cast<u32x2*>(rp + e * 4) = (u32x2){0u, 0u};
    }
  }
  commit(item);
  if (item + step < total) issue(item + step);
  __syncthreads();
  const int nct = (p.Wo + 15) >> 4, ct = wave % 7, half = wave / 7, col = 16 * ct + n;
  for (; item < total; item += step) {
    const int rb = item % nrb, f = (item / nrb) % p.F, b = item / (nrb * p.F);
    const int pr0 = rb * S6_PR, s0 = 2 * pr0 - 1;
    // the wave's two 16-channel tiles one after the other: 7 weight fragments + 5 accumulators live at a time
#pragma unroll 1
    for (int c2 = 0; c2 < 2; ++c2) {
      if (ct >= nct) break;
      const int ch0 = 32 * half + 16 * c2;
      v8 wa[7];
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) wa[kh] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(wl + ((size_t)(kh * 64 + ch0 + n)) * 32 + kg * 8));
      f32x4 acc[S6_SR];
#pragma unroll
      for (int s = 0; s < S6_SR; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const uint16_t* base = rows + (size_t)(2 * (16 * ct + n + kg)) * 4;
      constexpr int RG = 5, NG = S6_NR / RG;
      v8 bf[2][RG];
      auto rd = [&](int g, v8* dst) {
#pragma unroll
        for (int e = 0; e < RG; ++e) dst[e] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(base + (size_t)(g * RG + e) * PXP * 4));
      };
      rd(0, bf[0]);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) rd(g + 1, bf[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < RG; ++e) {
          const int r = g * RG + e;
#pragma unroll
          for (int s = 0; s < S6_SR; ++s) {
            const int kh = r - 2 * s;                              // staged row r = 2 s + kh of stem row s0 + s
            if (kh >= 0 && kh < 7) acc[s] = E::mfma16(wa[kh], bf[g & 1][e], acc[s]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // vertical max: lane (column n, group kg) holds channels ch0 + 4 kg .. +3 of every stem row
      if (col < p.Wo) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + ch0 + 4 * kg);
#pragma unroll
        for (int prl = 0; prl < S6_PR; ++prl) {
          float v[4];
          bool any = false;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int sr = 2 * prl + dy;
            if (s0 + sr < 0 || s0 + sr >= p.Ho) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = any ? fmaxf(v[e], acc[sr][e]) : acc[sr][e];
            any = true;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += bv[e];
            if (p.relu) v[e] = fmaxf(v[e], 0.f);
          }
          const int slot = (8 * half + 4 * c2 + kg) ^ ((col & 7) << 1);
          *reinterpret_cast<u32x2*>(vm + ((size_t)prl * p.Wo + col) * 64 + slot * 4) = (u32x2){E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
        }
      }
    }
    __syncthreads();                                               // every wave is done with `rows`; `vm` is complete
    // the next item's rows are converted BEFORE this item's pool stores are issued: the wait on the row loads then does not also wait
    // for stores issued a moment ago (vmcnt is one in-order counter), and the stores drain under the next item's MFMAs
    if (item + step < total) {
      commit(item + step);
      if (item + 2 * step < total) issue(item + 2 * step);
    }
    for (int i = tid; i < S6_PR * p.Wp * 8; i += S6_THREADS) {
      const int prl = i / (p.Wp * 8), rem = i - prl * p.Wp * 8, pc = rem >> 3, ch = rem & 7, pr = pr0 + prl;
      if (pr >= p.Hp) continue;
      bool any = false;
      u32x4 best = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int c = 2 * pc - 1 + dx;
        if (c < 0 || c >= p.Wo) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(vm + ((size_t)prl * p.Wo + c) * 64 + (ch ^ (c & 7)) * 8);
        best = any ? max8<E>(best, v) : v;
        any = true;
      }
      *reinterpret_cast<u32x4*>(p.out + ((((size_t)b * p.F + f) * p.Hp + pr) * p.Wp + pc) * p.out_C + p.out_coff + ch * 8) = best;
    }
    __syncthreads();                                               // `rows` hold the next item; `vm` may be overwritten
  }
}

}  // namespace kvq

extern "C" int kvq_pack_clip_cl4(const float* x, const int32_t dims5[5], int border, int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims5 && out, KVQ_ERR_NULL, "kvq_pack_clip_cl4: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_pack_clip_cl4: dtype %d", dtype);
  const int B = dims5[0], Cc = dims5[1], T = dims5[2], H = dims5[3], W = dims5[4];
  KVQ_REQUIRE(B > 0 && Cc > 0 && Cc <= 4 && T > 0 && H > 0 && W > 0 && border >= 0, KVQ_ERR_SHAPE,
              "kvq_pack_clip_cl4: bad shape B=%d C=%d (1..4) T=%d H=%d W=%d border=%d", B, Cc, T, H, W, border);
  const long total = (long)B * T * H * (W + 2 * border);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(pack_cl4_border_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, x, Cc, T, H, W, border, out, total);
  else hipLaunchKernelGGL(pack_cl4_border_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, x, Cc, T, H, W, border, out, total);
  KVQ_CHECK_LAUNCH("pack_cl4_border_kernel");
  return KVQ_OK;
}

extern "C" int kvq_conv_stem_mfma(const uint16_t* x4, const int32_t dims4[4], const uint16_t* wpack, const float* bias8,
                                  const int32_t kernel3[3], const int32_t stride3[3], const int32_t pad3[3], int relu, int dtype,
                                  uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x4 && dims4 && wpack && bias8 && kernel3 && stride3 && pad3 && out, KVQ_ERR_NULL, "kvq_conv_stem_mfma: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_mfma: dtype %d", dtype);
  KVQ_REQUIRE(kernel3[2] == 7 && stride3[2] == 2 && pad3[2] == 3, KVQ_ERR_UNSUPPORTED,
              "kvq_conv_stem_mfma: kernel width 7 / stride 2 / pad 3 along W only (got %d / %d / %d)", kernel3[2], stride3[2], pad3[2]);
  StemMfmaParams p{};
  p.x4 = x4; p.wp = wpack; p.bias = bias8; p.B = dims4[0]; p.T = dims4[1]; p.H = dims4[2]; p.Wp = dims4[3] + 8;
  p.kd = kernel3[0]; p.kh = kernel3[1]; p.sd = stride3[0]; p.sh = stride3[1]; p.pd = pad3[0]; p.ph = pad3[1]; p.relu = relu; p.out = out;
  KVQ_REQUIRE(p.B > 0 && p.T > 0 && p.H > 0 && dims4[3] > 0 && p.kd > 0 && p.kh > 0 && p.sd > 0 && p.sh > 0, KVQ_ERR_SHAPE,
              "kvq_conv_stem_mfma: bad shape");
  p.Do = (p.T + 2 * p.pd - p.kd) / p.sd + 1; p.Ho = (p.H + 2 * p.ph - p.kh) / p.sh + 1; p.Wo = (dims4[3] + 6 - 7) / 2 + 1;
  KVQ_REQUIRE(p.Do > 0 && p.Ho > 0 && p.Wo > 0, KVQ_ERR_SHAPE, "kvq_conv_stem_mfma: empty output");
  const size_t lds = (size_t)p.kd * p.kh * 16 * 32 * 2 + 4 * 2048;       // weights + a 2 KB row segment per wave
  KVQ_REQUIRE(lds <= 64 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_mfma: %zu B of weights exceed LDS", lds);
  KVQ_REQUIRE(p.Wp % 2 == 0 && (((size_t)x4) & 15) == 0, KVQ_ERR_SHAPE, "kvq_conv_stem_mfma: W must be even and x4 16-byte aligned");
  const long rows = (long)p.B * p.Do * p.Ho;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(conv_stem_mfma_kernel<Fp16>, grid, block, lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(conv_stem_mfma_kernel<Bf16>, grid, block, lds, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("conv_stem_mfma_kernel");
  return KVQ_OK;
}

// the shape limits of kvq_conv_stem_pool (everything but the pointers' alignment): also asked by kvq_convnet_create, so that a plan
// outside them is refused when it is built, not inside every forward
bool kvq::stem_pool_shape_ok(int B, int T, int H, int W, int kd) {
  if (!(B > 0 && T > 0 && H >= 7 && W >= 8 && kd >= 1 && kd <= 7 && (kd & 1) && W % 4 == 0)) return false;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1, Hp = (Ho + 2 - 3) / 2 + 1;
  if (!(Wo <= 128 && (long)B * T * Hp < (1L << 30) && (long)T * H * W < (1L << 27))) return false;
  return (size_t)SP_NR * (W + 8) * 8 + (size_t)SP_SR * Wo * 16 + (size_t)kd * 7 * 512 <= 96 * 1024;
}

extern "C" int kvq_conv_stem_pool(const float* x, const int32_t dims5[5], const uint16_t* wpack, const float* bias8, int kd, int relu,
                                  int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims5 && wpack && bias8 && out, KVQ_ERR_NULL, "kvq_conv_stem_pool: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_pool: dtype %d", dtype);
  StemPoolParams p{};
  p.x = x; p.wp = wpack; p.bias = bias8; p.B = dims5[0]; p.T = dims5[2]; p.H = dims5[3]; p.W = dims5[4]; p.kd = kd; p.relu = relu; p.out = out;
  KVQ_REQUIRE(dims5[1] == 3 && p.B > 0 && p.T > 0 && p.H >= 7 && p.W >= 8 && kd >= 1 && kd <= 7 && (kd & 1), KVQ_ERR_SHAPE,
              "kvq_conv_stem_pool: needs a 3-channel clip and an odd temporal kernel (got C=%d kd=%d)", dims5[1], kd);
  KVQ_REQUIRE(p.W % 4 == 0 && (((size_t)x) & 15) == 0 && (((size_t)out) & 15) == 0, KVQ_ERR_SHAPE,
              "kvq_conv_stem_pool: W %% 4 == 0 and 16-byte aligned clip / output (got W=%d)", p.W);
  p.Ho = (p.H + 6 - 7) / 2 + 1; p.Wo = (p.W + 6 - 7) / 2 + 1;
  p.Hp = (p.Ho + 2 - 3) / 2 + 1; p.Wp = (p.Wo + 2 - 3) / 2 + 1;
  KVQ_REQUIRE(p.Wo <= 128 && (long)p.B * p.T * p.Hp < (1L << 30) && (long)p.T * p.H * p.W < (1L << 27), KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_pool: stem rows of at most 128 columns (W <= 256; got %d)", p.W);
  const size_t lds = (size_t)SP_NR * (p.W + 8) * 8 + (size_t)SP_SR * p.Wo * 16 + (size_t)kd * 7 * 512;
  KVQ_REQUIRE(lds <= 96 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_pool: %zu B of LDS", lds);
  auto launch = [&](auto kern) -> int {
    static LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), 96 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ceil_div(ceil_div(p.Hp, SP_PR) * p.T * p.B, 8))), dim3(256), lds, (hipStream_t)stream, p);
    return KVQ_OK;
  };
  const int rc = dtype == KVQ_DT_FP16 ? launch(conv_stem_pool_kernel<Fp16>) : launch(conv_stem_pool_kernel<Bf16>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("conv_stem_pool_kernel");
  return KVQ_OK;
}

extern "C" int kvq_conv_stem64_pool(const float* x, const int32_t dims5[5], const int32_t* t_index, int n_frames, const uint16_t* wimg,
                                    const float* bias64, int relu, int dtype, uint16_t* out, int out_C, int out_coff, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims5 && wimg && bias64 && out, KVQ_ERR_NULL, "kvq_conv_stem64_pool: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem64_pool: dtype %d", dtype);
  Stem64Params p{};
  p.x = x; p.t_index = t_index; p.wimg = wimg; p.bias = bias64; p.B = dims5[0]; p.T = dims5[2]; p.H = dims5[3]; p.W = dims5[4];
  p.F = n_frames; p.relu = relu; p.out = out; p.out_C = out_C; p.out_coff = out_coff;
  KVQ_REQUIRE(dims5[1] == 3 && p.B > 0 && p.T > 0 && p.H >= 7 && p.W >= 8 && n_frames > 0 && (t_index || n_frames <= p.T), KVQ_ERR_SHAPE,
              "kvq_conv_stem64_pool: needs a 3-channel clip (got C=%d, %d frames of %d)", dims5[1], n_frames, p.T);
  KVQ_REQUIRE(p.W % 4 == 0 && p.W <= 224 && (((size_t)x) & 15) == 0 && (((size_t)out) & 15) == 0 && out_C % 8 == 0 && out_coff % 8 == 0 &&
                  out_coff >= 0 && out_coff + 64 <= out_C, KVQ_ERR_SHAPE,
              "kvq_conv_stem64_pool: W %% 4 == 0, W <= 224, 16-byte aligned clip / output channel slice (got W=%d, C=%d, offset %d)", p.W, out_C, out_coff);
  p.Ho = (p.H - 1) / 2 + 1; p.Wo = (p.W - 1) / 2 + 1;
  p.Hp = (p.Ho - 1) / 2 + 1; p.Wp = (p.Wo - 1) / 2 + 1;
  KVQ_REQUIRE((long)p.B * p.F * p.Hp < (1L << 30), KVQ_ERR_UNSUPPORTED, "kvq_conv_stem64_pool: %d clips x %d frames", p.B, p.F);
  const size_t lds = (size_t)S6_NR * (p.W + 8) * 8 + (size_t)S6_PR * p.Wo * 128 + (size_t)3 * S6_ITEMS * S6_THREADS * 16 + 7 * 64 * 64;
  auto launch = [&](auto kern) -> int {
    static LdsOptIn opt;
    if (int rc = opt.ensure(reinterpret_cast<const void*>(kern), 144 * 1024)) return rc;
    const int total = ceil_div(p.Hp, S6_PR) * p.F * p.B;          // one persistent workgroup per CU (256 on gfx950)
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min(total, 256)), dim3(S6_THREADS), lds, (hipStream_t)stream, p);
    return KVQ_OK;
  };
  const int rc = dtype == KVQ_DT_FP16 ? launch(conv_stem64_pool_kernel<Fp16>) : launch(conv_stem64_pool_kernel<Bf16>);
  if (rc) return rc;
  KVQ_CHECK_LAUNCH("conv_stem64_pool_kernel");
  return KVQ_OK;
}

extern "C" int kvq_im2col_nd(const void* x, int src_f32, int dtype, const int64_t strides5[5], const int32_t dims5[5],
                             const int32_t kernel3[3], const int32_t stride3[3], const int32_t pad3[3], int Kpad,
                             uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out && strides5 && dims5 && kernel3 && stride3 && pad3, KVQ_ERR_NULL, "kvq_im2col_nd: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_im2col_nd: dtype %d", dtype);
  Im2colParams p{};
  p.x = x; p.src_f32 = src_f32;
  p.sb = strides5[0]; p.sc = strides5[1]; p.sd = strides5[2]; p.sh = strides5[3]; p.sw = strides5[4];
  p.B = dims5[0]; p.C = dims5[1]; p.D = dims5[2]; p.H = dims5[3]; p.W = dims5[4];
  p.kd = kernel3[0]; p.kh = kernel3[1]; p.kw = kernel3[2];
  p.sdd = stride3[0]; p.shh = stride3[1]; p.sww = stride3[2];
  p.pd = pad3[0]; p.ph = pad3[1]; p.pw = pad3[2];
  KVQ_REQUIRE(p.B > 0 && p.C > 0 && p.D > 0 && p.H > 0 && p.W > 0 && p.kd > 0 && p.kh > 0 && p.kw > 0 && p.sdd > 0 &&
                  p.shh > 0 && p.sww > 0,
              KVQ_ERR_SHAPE, "kvq_im2col_nd: bad shape");
  p.Do = (p.D + 2 * p.pd - p.kd) / p.sdd + 1;
  p.Ho = (p.H + 2 * p.ph - p.kh) / p.shh + 1;
  p.Wo = (p.W + 2 * p.pw - p.kw) / p.sww + 1;
  p.K = p.kd * p.kh * p.kw * p.C;
  KVQ_REQUIRE(p.Do > 0 && p.Ho > 0 && p.Wo > 0 && Kpad >= p.K && Kpad % 32 == 0, KVQ_ERR_SHAPE,
              "kvq_im2col_nd: Kpad=%d must be a multiple of 32 and >= K=%d", Kpad, p.K);
  p.Kpad = Kpad; p.out = out;
  const long total = (long)p.B * p.Do * p.Ho * p.Wo * (Kpad / 8);
  const int grid = (int)((total + 255) / 256 < 131072 ? (total + 255) / 256 : 131072);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(im2col_nd_kernel<Fp16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(im2col_nd_kernel<Bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("im2col_nd_kernel");
  return KVQ_OK;
}

extern "C" int kvq_pool_nd_strided(const uint16_t* x, int dtype, const int32_t dims5[5], const int32_t kernel3[3],
                                   const int32_t stride3[3], const int32_t pad3[3], int is_max, uint16_t* out, int ldc, int col_off,
                                   void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out && dims5 && kernel3 && stride3 && pad3, KVQ_ERR_NULL, "kvq_pool_nd: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_pool_nd: dtype %d", dtype);
  PoolParams p{};
  p.x = x; p.out = out;
  p.B = dims5[0]; p.C = dims5[1]; p.D = dims5[2]; p.H = dims5[3]; p.W = dims5[4];
  p.kd = kernel3[0]; p.kh = kernel3[1]; p.kw = kernel3[2];
  p.sdd = stride3[0]; p.shh = stride3[1]; p.sww = stride3[2];
  p.pd = pad3[0]; p.ph = pad3[1]; p.pw = pad3[2];
  p.is_max = is_max;
  p.ldc = ldc > 0 ? ldc : dims5[1]; p.coff = ldc > 0 ? col_off : 0;
  KVQ_REQUIRE(p.ldc % 8 == 0 && p.coff % 8 == 0 && p.coff >= 0 && p.coff + dims5[1] <= p.ldc || ldc <= 0, KVQ_ERR_SHAPE,
              "kvq_pool_nd: ldc / col_off must be multiples of 8 with col_off + C <= ldc");
  KVQ_REQUIRE(p.B > 0 && p.C > 0 && p.kd > 0 && p.kh > 0 && p.kw > 0 && p.sdd > 0 && p.shh > 0 && p.sww > 0,
              KVQ_ERR_SHAPE, "kvq_pool_nd: bad shape");
  p.Do = (p.D + 2 * p.pd - p.kd) / p.sdd + 1;
  p.Ho = (p.H + 2 * p.ph - p.kh) / p.shh + 1;
  p.Wo = (p.W + 2 * p.pw - p.kw) / p.sww + 1;
  KVQ_REQUIRE(p.Do > 0 && p.Ho > 0 && p.Wo > 0, KVQ_ERR_SHAPE, "kvq_pool_nd: empty output");
  if (p.C % 8 == 0 && (((size_t)p.x | (size_t)p.out) & 15) == 0) {
    const long tot8 = (long)p.B * p.Do * p.Ho * p.Wo * (p.C / 8);
    const int g8 = (int)((tot8 + 255) / 256 < 131072 ? (tot8 + 255) / 256 : 131072);
    if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(pool_nd_vec8_kernel<Fp16>, dim3(g8), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(pool_nd_vec8_kernel<Bf16>, dim3(g8), dim3(256), 0, (hipStream_t)stream, p);
    KVQ_CHECK_LAUNCH("pool_nd_vec8_kernel");
    return KVQ_OK;
  }
  const long total = (long)p.B * p.Do * p.Ho * p.Wo * p.C;
  const int grid = (int)((total + 255) / 256 < 131072 ? (total + 255) / 256 : 131072);
  if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(pool_nd_kernel<Fp16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pool_nd_kernel<Bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("pool_nd_kernel");
  return KVQ_OK;
}

extern "C" int kvq_pool_nd(const uint16_t* x, int dtype, const int32_t dims5[5], const int32_t kernel3[3],
                           const int32_t stride3[3], const int32_t pad3[3], int is_max, uint16_t* out, void* stream) {
  return kvq_pool_nd_strided(x, dtype, dims5, kernel3, stride3, pad3, is_max, out, 0, 0, stream);
}

extern "C" int kvq_pack_channels_last8(const float* x, const int32_t dims5[5], const int64_t strides5[5], int dtype, uint16_t* out,
                                       void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims5 && strides5 && out, KVQ_ERR_NULL, "kvq_pack_channels_last8: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_pack_channels_last8: dtype %d", dtype);
  const int B = dims5[0], T = dims5[1], Cc = dims5[2], H = dims5[3], W = dims5[4];
  KVQ_REQUIRE(B > 0 && T > 0 && Cc > 0 && Cc <= 8 && H > 0 && W > 0, KVQ_ERR_SHAPE,
              "kvq_pack_channels_last8: bad shape B=%d T=%d C=%d (1..8) H=%d W=%d", B, T, Cc, H, W);
  KVQ_REQUIRE(((size_t)out & 15) == 0, KVQ_ERR_SHAPE, "kvq_pack_channels_last8: out must be 16-byte aligned");
  const long total = (long)B * T * H * W;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == KVQ_DT_FP16)
    hipLaunchKernelGGL(pack_cl8_kernel<Fp16>, grid, block, 0, (hipStream_t)stream, x, T, Cc, H, W, (long)strides5[0], (long)strides5[1],
                       (long)strides5[2], (long)strides5[3], (long)strides5[4], out, total);
  else
    hipLaunchKernelGGL(pack_cl8_kernel<Bf16>, grid, block, 0, (hipStream_t)stream, x, T, Cc, H, W, (long)strides5[0], (long)strides5[1],
                       (long)strides5[2], (long)strides5[3], (long)strides5[4], out, total);
  KVQ_CHECK_LAUNCH("pack_cl8_kernel");
  return KVQ_OK;
}

extern "C" int kvq_mean_std_pool(const uint16_t* x, int dtype, int rows, int HW, int C, float* out, int64_t out_stride,
                                 int mean_off, int std_off, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out, KVQ_ERR_NULL, "kvq_mean_std_pool: NULL pointer");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_mean_std_pool: dtype %d", dtype);
  KVQ_REQUIRE(rows > 0 && HW > 0 && C > 0 && (std_off < 0 || HW > 1), KVQ_ERR_SHAPE, "kvq_mean_std_pool: bad shape");
  // The kernel (and with it the order the positions are summed in) is chosen from HW, C and the pointer's alignment ONLY — never
  // from `rows`, which is the batch in the extractors' head pools: a clip's features must not change in their last bits with what
  // else shares the launch (kvq_convnet_splitk is off there for the same reason).
  hipStream_t st = (hipStream_t)stream;
  // 16-byte loads, a block per 8 channels: maps of >= 256 positions (KSVQE: 4 rows x 3136 positions; SlowFast's slow head pool:
  // 8 rows x 392 positions x 2048 channels, 45.8 -> 8 us); 16-channel tiles when the channels cannot be taken 8 at a time
  const bool wide = C % 8 == 0 && ((size_t)x & 15) == 0 && HW >= 256;
  const bool narrow = HW >= 256;
  if (wide) {
    dim3 g8(rows, C / 8);
    if (dtype == KVQ_DT_FP16) hipLaunchKernelGGL(mean_std_pool_vec8_kernel<Fp16>, g8, dim3(256), 0, st, x, HW, C, out, (long)out_stride, mean_off, std_off);
    else hipLaunchKernelGGL(mean_std_pool_vec8_kernel<Bf16>, g8, dim3(256), 0, st, x, HW, C, out, (long)out_stride, mean_off, std_off);
    KVQ_CHECK_LAUNCH("mean_std_pool_vec8_kernel");
    return KVQ_OK;
  }
  dim3 grid(rows, ceil_div(C, narrow ? 16 : 64));
  if (dtype == KVQ_DT_FP16) {
    if (narrow) hipLaunchKernelGGL((mean_std_pool_kernel<Fp16, 16>), grid, dim3(256), 0, st, x, HW, C, out, (long)out_stride, mean_off, std_off);
    else hipLaunchKernelGGL((mean_std_pool_kernel<Fp16, 64>), grid, dim3(256), 0, st, x, HW, C, out, (long)out_stride, mean_off, std_off);
  } else {
    if (narrow) hipLaunchKernelGGL((mean_std_pool_kernel<Bf16, 16>), grid, dim3(256), 0, st, x, HW, C, out, (long)out_stride, mean_off, std_off);
    else hipLaunchKernelGGL((mean_std_pool_kernel<Bf16, 64>), grid, dim3(256), 0, st, x, HW, C, out, (long)out_stride, mean_off, std_off);
  }
  KVQ_CHECK_LAUNCH("mean_std_pool_kernel");
  return KVQ_OK;
}

extern "C" int kvq_conv_stem_direct(const float* x, const int32_t dims5[5], const float* w, const float* bias, int cout,
                                    const int32_t kernel3[3], const int32_t stride3[3], const int32_t pad3[3], int relu,
                                    int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && dims5 && w && bias && kernel3 && stride3 && pad3 && out, KVQ_ERR_NULL, "kvq_conv_stem_direct: NULL pointer");
  KVQ_REQUIRE(cout == 8 || cout == 16, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_direct: Cout=%d (8 or 16: wider convs are GEMMs)", cout);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_direct: dtype %d", dtype);
  StemParams p{};
  p.x = x; p.w = w; p.bias = bias; p.B = dims5[0]; p.C = dims5[1]; p.D = dims5[2]; p.H = dims5[3]; p.W = dims5[4];
  p.kd = kernel3[0]; p.kh = kernel3[1]; p.kw = kernel3[2]; p.sd = stride3[0]; p.sh = stride3[1]; p.sw = stride3[2];
  p.pd = pad3[0]; p.ph = pad3[1]; p.pw = pad3[2]; p.relu = relu; p.out = out;
  KVQ_REQUIRE(p.B > 0 && p.C > 0 && p.sd > 0 && p.sh > 0 && p.sw > 0, KVQ_ERR_SHAPE, "kvq_conv_stem_direct: bad shape");
  p.Do = (p.D + 2 * p.pd - p.kd) / p.sd + 1; p.Ho = (p.H + 2 * p.ph - p.kh) / p.sh + 1; p.Wo = (p.W + 2 * p.pw - p.kw) / p.sw + 1;
  KVQ_REQUIRE(p.Do > 0 && p.Ho > 0 && p.Wo > 0, KVQ_ERR_SHAPE, "kvq_conv_stem_direct: empty output");
  const size_t lds = (size_t)p.kd * p.kh * p.kw * p.C * cout * 4;
  KVQ_REQUIRE(lds <= 64 * 1024, KVQ_ERR_UNSUPPORTED, "kvq_conv_stem_direct: %zu B of weights exceed LDS", lds);
  const long total = (long)p.B * p.Do * p.Ho * p.Wo;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == KVQ_DT_FP16) {
    if (cout == 8) hipLaunchKernelGGL((conv_stem_direct_kernel<Fp16, 8>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((conv_stem_direct_kernel<Fp16, 16>), grid, block, lds, st, p);
  } else {
    if (cout == 8) hipLaunchKernelGGL((conv_stem_direct_kernel<Bf16, 8>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((conv_stem_direct_kernel<Bf16, 16>), grid, block, lds, st, p);
  }
  KVQ_CHECK_LAUNCH("conv_stem_direct_kernel");
  return KVQ_OK;
}
