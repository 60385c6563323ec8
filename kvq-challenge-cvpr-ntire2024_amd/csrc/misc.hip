// HBM-bound helpers around the trunk: patch-embed im2col, the two score heads, the fragment sampler.
#include "common.hpp"

namespace kvq {

// ------------------------------------------------------------------------------------------------
// PatchEmbed3D im2col (swin_backbone.py:715-726).  One workgroup = one (b, d', h') row of W' tokens:
// it reads the Cin*pd*ph source rows (each W contiguous floats -> coalesced float4 loads), converts and
// transposes them through LDS, and writes the W' GEMM rows (Cin*pd*ph*pw 16-bit values each) as one
// contiguous run with 16-B stores.  Row layout (c,kd,kh,kw) equals Conv3d weight.flatten(1);
// out-of-range (zero-padded tail) pixels read as 0.
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ x, int B, int Cin, int T, int H,
                                                           int W, int pd, int ph, int pw, int D, int Hh, int Ww,
                                                           uint16_t* __restrict__ out) {
  fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) uint16_t tile[];   // [Ww][K] in the output order
  const int K = Cin * pd * ph * pw;
  const int runs = Cin * pd * ph;                 // source rows feeding this token row
  int blk = blockIdx.x;
  const int hq = blk % Hh; blk /= Hh;
  const int dq = blk % D;
  const int b = blk / D;
  const int Wpad = Ww * pw;
  // gather: element (run, col) of the source rows -> tile[col / pw][run * pw + col % pw]
  for (int i = threadIdx.x; i < runs * Wpad; i += blockDim.x) {
    const int run = i / Wpad, col = i - run * Wpad;
    const int kh = run % ph, kd = (run / ph) % pd, c = run / (ph * pd);
    const int tt = dq * pd + kd, hh = hq * ph + kh;
    float v = 0.f;
    if (tt < T && hh < H && col < W) v = x[(((size_t)b * Cin + c) * T + tt) * (size_t)H * W + (size_t)hh * W + col];
    tile[(col / pw) * K + run * pw + (col % pw)] = E::cvt(v);
  }
  __syncthreads();
  uint16_t* o = out + ((((size_t)b * D + dq) * Hh + hq) * Ww) * (size_t)K;
  const int n16 = Ww * K / 8;                       // K % 32 == 0 -> whole 16-B chunks
  for (int i = threadIdx.x; i < n16; i += blockDim.x)
    reinterpret_cast<u32x4*>(o)[i] = reinterpret_cast<const u32x4*>(tile)[i];
}

// ------------------------------------------------------------------------------------------------
// VQAHead (models/head.py:60-68, eval): per token  s = w2 . gelu(W1 f + b1) + b2 ; score = mean_tokens.
// fp32 FMA throughout (0.08 GFLOP/clip — not worth a 16-bit rounding at the very end of the net).
// Lane j owns hidden unit j: W1 is passed TRANSPOSED ([C][hidden]) so the 64 lanes read one coalesced
// 256-B row per channel; the token's feature value is a wave-uniform (broadcast) load.  A wave carries
// HEAD_TOK tokens to amortise the W1 stream (L2-resident, 196 KB); one shuffle reduction per token.
// ------------------------------------------------------------------------------------------------
constexpr int HEAD_TOK = 4;

// One workgroup = HEAD_TOK tokens; its 4 waves split the CHANNELS (the reduction dimension) in four, so the
// dependent chain of W1-row loads per wave is C/4/16 steps instead of C/8 (the kernel is L2-latency bound: 96
// serial steps cost 52 us for 0.3 GFLOP), and the partial sums meet in LDS.
__global__ __launch_bounds__(256) void vqa_head_token_kernel(const float* __restrict__ feat, int B, int L, int C,
                                                             long sb, long sl, long sc, const float* __restrict__ w1t,
                                                             const float* __restrict__ b1, int hidden,
                                                             const float* __restrict__ w2,
                                                             float* __restrict__ tok_score) {
  __shared__ float part[4][HEAD_TOK][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tok0 = (long)blockIdx.x * HEAD_TOK;
  const long total = (long)B * L;
  const float* row[HEAD_TOK];
#pragma unroll
  for (int t = 0; t < HEAD_TOK; ++t) {
    const long tk = min(tok0 + t, total - 1);
    const long b = tk / L, l = tk - b * L;
    row[t] = feat + b * sb + l * sl;
  }
  const int cq = ((C + 3) / 4 + 7) & ~7;                    // channels per wave, a multiple of 8
  const int c_lo = min(wave * cq, C), c_hi = min(c_lo + cq, C);
  float acc[HEAD_TOK];
#pragma unroll
  for (int t = 0; t < HEAD_TOK; ++t) acc[t] = 0.f;
  for (int j0 = 0; j0 < hidden; j0 += 64) {
    const int j = j0 + lane;
    const bool live = j < hidden;
    const int jc = live ? j : hidden - 1;
    float d[HEAD_TOK];
#pragma unroll
    for (int t = 0; t < HEAD_TOK; ++t) d[t] = 0.f;
    if (sc == 1 && (C & 7) == 0 && (hidden & 3) == 0) {
      // channels-last features.  4-B loads are instruction-bound (256 B per wave-load), so the wave is re-shaped for
      // this loop: lane = (hidden quad hq = lane & 15, channel quad cg = lane >> 4); a step covers 16 channels with
      // 16-B loads of W1^T rows and of the tokens, 64 fma per lane; the four channel quads meet by two shuffles.
      const int hq = lane & 15, cg = lane >> 4;
      const int jq = min(j0 + 4 * hq, hidden - 4);
      float dq[HEAD_TOK][4];
#pragma unroll
      for (int t = 0; t < HEAD_TOK; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) dq[t][k] = 0.f;
      for (int c0 = c_lo + 4 * cg; c0 < c_hi; c0 += 16) {      // c_lo, c_hi are multiples of 8: a quad is all in or out
        f32x4 wq[4], xv[HEAD_TOK];
#pragma unroll
        for (int k = 0; k < 4; ++k) wq[k] = *reinterpret_cast<const f32x4*>(w1t + (size_t)(c0 + k) * hidden + jq);
#pragma unroll
        for (int t = 0; t < HEAD_TOK; ++t) xv[t] = *reinterpret_cast<const f32x4*>(row[t] + c0);
#pragma unroll
        for (int t = 0; t < HEAD_TOK; ++t)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) dq[t][e] = fmaf(xv[t][k], wq[k][e], dq[t][e]);
      }
      // lane j of the common layout needs hidden unit j0 + lane: sum the channel quads, then pick element lane & 3 of
      // hidden quad lane >> 2
#pragma unroll
      for (int t = 0; t < HEAD_TOK; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = dq[t][e];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          dq[t][e] = v;
        }
        const int srcl = lane >> 2;                            // the lane (hq = lane>>2, cg = 0) holds the quad
        const float q0 = __shfl(dq[t][0], srcl), q1 = __shfl(dq[t][1], srcl), q2 = __shfl(dq[t][2], srcl),
                    q3 = __shfl(dq[t][3], srcl);
        const int e = lane & 3;
        const float mine = e == 0 ? q0 : (e == 1 ? q1 : (e == 2 ? q2 : q3));
        d[t] = mine;        // live lanes (j < hidden, hidden % 4 == 0) never see a clamped quad
      }
    } else {
#pragma unroll 4
      for (int c = c_lo; c < c_hi; ++c) {
        const float wv = w1t[(size_t)c * hidden + jc];
#pragma unroll
        for (int t = 0; t < HEAD_TOK; ++t) d[t] = fmaf(row[t][c * sc], wv, d[t]);
      }
    }
    __syncthreads();                                         // previous j0 round consumed
#pragma unroll
    for (int t = 0; t < HEAD_TOK; ++t) part[wave][t][lane] = d[t];
    __syncthreads();
    if (wave == 0) {
      const float bj = b1[jc], wj = live ? w2[jc] : 0.f;
#pragma unroll
      for (int t = 0; t < HEAD_TOK; ++t) {
        const float full = (part[0][t][lane] + part[1][t][lane]) + (part[2][t][lane] + part[3][t][lane]);
        acc[t] = fmaf(wj, gelu_erf(full + bj), acc[t]);
      }
    }
  }
  if (wave != 0) return;
#pragma unroll
  for (int t = 0; t < HEAD_TOK; ++t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[t] += __shfl_xor(acc[t], o);
  }
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < HEAD_TOK; ++t)
      if (tok0 + t < total) tok_score[tok0 + t] = acc[t];
  }
}

// The same head on the fp32 matrix pipe, for hidden == 64 and channels-last features: Out^T[64 hidden][16 tokens] =
// W1[64][C] * X^T[C][16 tokens] by v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate: no rounding the VALU form does not
// have).  A lane's 16-byte load is four consecutive channels of its W1 row / of its token, and k-step s of a 16-channel
// macro-step takes element s of every lane's quad (k index = the lane's quad number) — W1 is read in its PyTorch layout, no
// packing.  The four waves split the channels; the partial tiles meet in LDS in a fixed order.  196 workgroups of 16 tokens
// for 4 clips: 12.8 us against 22.2 for the VALU form (784 workgroups that each stream all of W1^T in 12 dependent steps).
// Eight waves with 64 channels of operands requested ahead measured 13.9: the launch is a fixed ~5 us (a 4-workgroup
// mean_rows_kernel takes 4.6) + one workgroup's 192 MFMAs per SIMD + its first load.
__global__ __launch_bounds__(256) void vqa_head_mfma_kernel(const float* __restrict__ feat, int B, int L, int C, long sb, long sl,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, float* __restrict__ tok_score) {
  __shared__ float part[4][64][17];
  __shared__ float red[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok = lane & 15, kq = lane >> 4;
  const long total = (long)B * L;
  const long tk = min((long)blockIdx.x * 16 + tok, total - 1);
  const long b = tk / L, l = tk - b * L;
  const int cw = C / 4, c0 = wave * cw;
  const float* xr = feat + b * sb + l * sl + c0 + 4 * kq;
  const float* wr = w1 + (size_t)tok * C + c0 + 4 * kq;          // + 16 hb rows
  const size_t hbs = (size_t)16 * C;
  f32x4 acc[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) acc[hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 xv = *reinterpret_cast<const f32x4*>(xr), wv[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) wv[hb] = *reinterpret_cast<const f32x4*>(wr + hb * hbs);
  for (int c = 0; c < cw; c += 16) {
    f32x4 xn = xv, wn[4] = {wv[0], wv[1], wv[2], wv[3]};
    if (c + 16 < cw) {                                             // the next macro-step's operands, under this one's MFMAs
      xn = *reinterpret_cast<const f32x4*>(xr + c + 16);
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) wn[hb] = *reinterpret_cast<const f32x4*>(wr + hb * hbs + c + 16);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) acc[hb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[hb][s], xv[s], acc[hb], 0, 0, 0);
    xv = xn;
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) wv[hb] = wn[hb];
  }
#pragma unroll
  for (int hb = 0; hb < 4; ++hb)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][16 * hb + 4 * kq + r][tok] = acc[hb][r];
  __syncthreads();
  // thread = (token tid & 15, hidden quad tid >> 4): channel partials in wave order, bias, GELU, times w2
  const int t2 = tid & 15, hg = tid >> 4;
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int h = 4 * hg + r;
    const float full = (part[0][h][t2] + part[1][h][t2]) + (part[2][h][t2] + part[3][h][t2]);
    v = fmaf(w2[h], gelu_erf(full + b1[h]), v);
  }
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  if (lane < 16) red[wave][lane] = v;
  __syncthreads();
  if (tid < 16 && (long)blockIdx.x * 16 + tid < total)
    tok_score[(long)blockIdx.x * 16 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// score[b] = mean_l tok_score[b*L + l] + b2   (deterministic tree, one block per batch element)
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ v, int L, const float* b2,
                                                        float* __restrict__ out) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) s += v[(size_t)b * L + l];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = red[0] / (float)L + (b2 ? b2[0] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// simpleVQAHead (models/head.py:28-31): Linear(Cin->hidden) -> Linear(hidden->1) per frame, no activation.
// One block per (b, t) frame row: 256 threads split Cin, each accumulates all hidden dots?  hidden=128,
// Cin=9472: W1 is 4.8 MB fp32 and is re-read per frame from L2; 8 frames/video -> negligible.
// Thread j<hidden computes its dot product with a block-cooperative, LDS-staged feature row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void simple_head_frame_kernel(const float* __restrict__ feat, int Cin,
                                                                const float* __restrict__ w1,
                                                                const float* __restrict__ b1, int hidden,
                                                                const float* __restrict__ w2,
                                                                float* __restrict__ frame_score) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* f = feat + (size_t)row * Cin;
  float acc = 0.f;
  for (int j = wave; j < hidden; j += 4) {
    float d = 0.f;
    for (int c = lane; c < Cin; c += 64) d = fmaf(f[c], w1[(size_t)j * Cin + c], d);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    acc = fmaf(w2[j], d + b1[j], acc);
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) frame_score[row] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// Fragment sampler (fusion_datasets.py:22-121 + :1017-1020): grid-mini-patch gather + normalise.
// One thread = 4 consecutive output pixels of one output row; patch origins come from hoff/woff.
// ------------------------------------------------------------------------------------------------
struct FragParams {
  const void* video;
  int src_is_u8, C, T, H, W;
  const int32_t* hoff;
  const int32_t* woff;
  int Fh, Fw, fsh, fsw, aligned;
  float mean[4], std[4];
  int normalise;
  float* out;
  long chan_stride;          // elements between the channel planes of a clip (T * H * W when contiguous)
};

// a batch of clips in one launch (blockIdx.z = clip): per-clip frames / draws, outputs out + z * C * T * OH * OW
struct FragBatch {
  const void* video[KVQ_FRAG_MAX_CLIPS];
  const int32_t* hoff[KVQ_FRAG_MAX_CLIPS];
  const int32_t* woff[KVQ_FRAG_MAX_CLIPS];
};

// One thread = VW consecutive output pixels of one row of one (channel, frame) plane (VW = 4 when the patch width is a multiple of
// 4: the four pixels then sit in one patch, i.e. in one 4-byte run of the source row, and leave as one 16-byte store; the plane is
// blockIdx.y, so the per-pixel index arithmetic is two small divisions per thread instead of five 64-bit ones per pixel — the
// one-pixel-per-thread form of rounds 1-2 spent 30 us per clip on 24 MB of traffic).  (v - mean) / std stays the IEEE fp32 divide:
// bit-equal to the reference's normalisation (fusion_datasets.py:1017-1020).
template <int VW, bool BATCH>
__global__ __launch_bounds__(256) void fragment_gather_kernel(FragParams p, FragBatch fb) {
  if (BATCH) {
    const int z = blockIdx.z;                        // wave-uniform: scalar loads of the clip's pointers
    p.video = fb.video[z]; p.hoff = fb.hoff[z]; p.woff = fb.woff[z];
    p.out += (size_t)z * p.C * p.T * (p.Fh * p.fsh) * (size_t)(p.Fw * p.fsw);
  }
  const int OH = p.Fh * p.fsh, OW = p.Fw * p.fsw, QW = OW / VW;
  const int nt = p.T / p.aligned;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= OH * QW) return;
  const int ct = blockIdx.y, c = ct / p.T, t = ct - c * p.T;           // wave-uniform
  const int oy = q / QW, ox = (q - oy * QW) * VW;
  const int fi = oy / p.fsh, fj = ox / p.fsw;
  const int o = (fi * p.Fw + fj) * nt + t / p.aligned;
  const int sy = p.hoff[o] + (oy - fi * p.fsh), sx = p.woff[o] + (ox - fj * p.fsw);
  const size_t src = (size_t)c * p.chan_stride + ((size_t)t * p.H + sy) * p.W + sx;
  float v[VW];
  if (p.src_is_u8) {
    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(p.video) + src;
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = (float)s8[e];
  } else {
    const float* s32 = reinterpret_cast<const float*>(p.video) + src;
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = s32[e];
  }
  if (p.normalise) {
    const float m = p.mean[c], sd = p.std[c];
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = (v[e] - m) / sd;                 // IEEE fp32 divide
  }
  float* dst = p.out + ((size_t)ct * OH + oy) * OW + ox;
  if (VW == 4) *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2 % VW], v[3 % VW]};
  else dst[0] = v[0];
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize (+ crop + normalise): torchvision.transforms.Resize on a tensor == F.interpolate(
// mode="bilinear", align_corners=False, antialias=False) (get_resize_function, fusion_datasets.py:229-241),
// the centre crop of get_resizecrop_video (:299-316) and the dataset's (v-mean)/std (:903) in one pass.
// Source index as ATen: src = max(0, scale*(dst+0.5)-0.5), scale = in/out in fp32.
// ------------------------------------------------------------------------------------------------
struct ResizeParams {
  const void* video;
  int src_is_u8, C, T, H, W, rh, rw, cy, cx, oh, ow, round_u8, normalise;
  float mean[4], std[4];
  float* out;
};

__global__ __launch_bounds__(256) void resize_bilinear_kernel(ResizeParams p) {
  const long total = (long)p.C * p.T * p.oh * p.ow;
  const float sy = (float)p.H / (float)p.rh, sx = (float)p.W / (float)p.rw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int ox = (int)(r % p.ow); r /= p.ow;
    const int oy = (int)(r % p.oh); r /= p.oh;
    const int t = (int)(r % p.T);
    const int c = (int)(r / p.T);
    const float fy = fmaxf(sy * ((float)(oy + p.cy) + 0.5f) - 0.5f, 0.f);
    const float fx = fmaxf(sx * ((float)(ox + p.cx) + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.H - 1 ? 1 : 0), x1 = x0 + (x0 < p.W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const size_t base = ((size_t)c * p.T + t) * (size_t)p.H * p.W;
    float v00, v01, v10, v11;
    if (p.src_is_u8) {
      const uint8_t* s = reinterpret_cast<const uint8_t*>(p.video) + base;
      v00 = s[(size_t)y0 * p.W + x0]; v01 = s[(size_t)y0 * p.W + x1];
      v10 = s[(size_t)y1 * p.W + x0]; v11 = s[(size_t)y1 * p.W + x1];
    } else {
      const float* s = reinterpret_cast<const float*>(p.video) + base;
      v00 = s[(size_t)y0 * p.W + x0]; v01 = s[(size_t)y0 * p.W + x1];
      v10 = s[(size_t)y1 * p.W + x0]; v11 = s[(size_t)y1 * p.W + x1];
    }
    float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    if (p.round_u8) v = fminf(fmaxf(rintf(v), 0.f), 255.f);      // torchvision rounds when the tensor is integer
    if (p.normalise) v = (v - p.mean[c]) / p.std[c];
    p.out[i] = v;
  }
}

}  // namespace kvq

extern "C" int kvq_resize_bilinear(const void* video, int src_is_u8, int C, int T, int H, int W, int rh, int rw, int cy,
                                   int cx, int oh, int ow, int round_u8, const float* host_mean, const float* host_std,
                                   float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(video && out, KVQ_ERR_NULL, "kvq_resize_bilinear: NULL pointer");
  KVQ_REQUIRE(C > 0 && C <= 4 && T > 0 && H > 0 && W > 0 && rh > 0 && rw > 0 && oh > 0 && ow > 0 && cy >= 0 && cx >= 0 &&
                  cy + oh <= rh && cx + ow <= rw,
              KVQ_ERR_SHAPE, "kvq_resize_bilinear: bad shape / crop outside the resized frame");
  ResizeParams p{};
  p.video = video; p.src_is_u8 = src_is_u8; p.C = C; p.T = T; p.H = H; p.W = W; p.rh = rh; p.rw = rw; p.cy = cy; p.cx = cx;
  p.oh = oh; p.ow = ow; p.round_u8 = round_u8; p.normalise = host_std != nullptr; p.out = out;
  for (int c = 0; c < C; ++c) {
    p.mean[c] = host_mean ? host_mean[c] : 0.f;
    p.std[c] = host_std ? host_std[c] : 1.f;
  }
  const long total = (long)C * T * oh * ow;
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("resize_bilinear_kernel");
  return KVQ_OK;
}

namespace kvq {

// ------------------------------------------------------------------------------------------------
// get_spatial_fragments' fallback for sources smaller than the canvas (fusion_datasets.py:43-50):
//   video = F.interpolate(video / 255.0, scale_factor = 1 / ratio, mode = "bilinear");  video = (video * 255.0).type_as(ovideo)
// i.e. ATen's upsample_bilinear2d (align_corners = False) on the (C, T, H, W) tensor with the SCALE FACTOR handed to the op (the
// source coordinate uses float(1 / scale_factor), not in / out), then a truncating cast back to the frame type.  The arithmetic
// follows ATen's CPU kernel of the image's torch build to the bit, contractions included (found by enumeration against
// F.interpolate, tests/test_gpu_harness.py): src = fma(scale, dst + 0.5, -0.5) clamped at 0, lambda0 = 1 - lambda1,
// row = fma(v0, lx0, v1 * lx1), out = fma(row0, ly0, row1 * ly1).  A flat region must come back as itself or one grey level
// lower exactly where the reference's does — (100 / 255) * 255 truncates to 99 or 100 depending on these roundings.
// ------------------------------------------------------------------------------------------------
struct UpsampleParams {
  const void* video;
  void* out;
  int src_is_u8, C, T, H, W, OH, OW;
  float scale;               // float(1.0 / scale_factor)
};

__global__ __launch_bounds__(256) void upsample_frames_kernel(UpsampleParams p) {
  const long total = (long)p.C * p.T * p.OH * p.OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int ox = (int)(r % p.OW); r /= p.OW;
    const int oy = (int)(r % p.OH);
    const long ct = r / p.OH;
    const float fy = fmaxf(__fmaf_rn(p.scale, (float)oy + 0.5f, -0.5f), 0.f);
    const float fx = fmaxf(__fmaf_rn(p.scale, (float)ox + 0.5f, -0.5f), 0.f);
    const int y0 = min((int)floorf(fy), p.H - 1), x0 = min((int)floorf(fx), p.W - 1);
    const int y1 = y0 + (y0 < p.H - 1 ? 1 : 0), x1 = x0 + (x0 < p.W - 1 ? 1 : 0);
    const float ly1 = fminf(fmaxf(__fsub_rn(fy, (float)y0), 0.f), 1.f), lx1 = fminf(fmaxf(__fsub_rn(fx, (float)x0), 0.f), 1.f);
    const float ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
    const size_t base = (size_t)ct * (size_t)p.H * p.W;
    float v00, v01, v10, v11;
    if (p.src_is_u8) {
      const uint8_t* s = reinterpret_cast<const uint8_t*>(p.video) + base;
      v00 = s[(size_t)y0 * p.W + x0]; v01 = s[(size_t)y0 * p.W + x1];
      v10 = s[(size_t)y1 * p.W + x0]; v11 = s[(size_t)y1 * p.W + x1];
    } else {
      const float* s = reinterpret_cast<const float*>(p.video) + base;
      v00 = s[(size_t)y0 * p.W + x0]; v01 = s[(size_t)y0 * p.W + x1];
      v10 = s[(size_t)y1 * p.W + x0]; v11 = s[(size_t)y1 * p.W + x1];
    }
    v00 = __fdiv_rn(v00, 255.0f); v01 = __fdiv_rn(v01, 255.0f); v10 = __fdiv_rn(v10, 255.0f); v11 = __fdiv_rn(v11, 255.0f);
    const float r0 = __fmaf_rn(v00, lx0, __fmul_rn(v01, lx1));
    const float r1 = __fmaf_rn(v10, lx0, __fmul_rn(v11, lx1));
    const float v = __fmul_rn(__fmaf_rn(r0, ly0, __fmul_rn(r1, ly1)), 255.0f);
    if (p.src_is_u8) reinterpret_cast<uint8_t*>(p.out)[i] = (uint8_t)(int)v;          // .type_as(uint8): truncation
    else reinterpret_cast<float*>(p.out)[i] = v;
  }
}

}  // namespace kvq

extern "C" int kvq_upsample_frames_out_dims(int H, int W, double scale_factor, int32_t out2[2]) {
  using namespace kvq;
  KVQ_REQUIRE(out2, KVQ_ERR_NULL, "kvq_upsample_frames_out_dims: NULL");
  KVQ_REQUIRE(H > 0 && W > 0 && scale_factor > 0.0, KVQ_ERR_SHAPE, "kvq_upsample_frames_out_dims: bad shape / scale");
  // torch.nn.functional.interpolate: math.floor(float(size * scale_factor)) in double
  out2[0] = (int32_t)floor((double)H * scale_factor);
  out2[1] = (int32_t)floor((double)W * scale_factor);
  return KVQ_OK;
}

extern "C" int kvq_upsample_frames(const void* video, int src_is_u8, int C, int T, int H, int W, double scale_factor, void* out,
                                   void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(video && out, KVQ_ERR_NULL, "kvq_upsample_frames: NULL pointer");
  KVQ_REQUIRE(C > 0 && T > 0 && H > 0 && W > 0 && scale_factor > 0.0, KVQ_ERR_SHAPE, "kvq_upsample_frames: bad shape / scale");
  int32_t od[2];
  if (int rc = kvq_upsample_frames_out_dims(H, W, scale_factor, od)) return rc;
  KVQ_REQUIRE(od[0] > 0 && od[1] > 0, KVQ_ERR_SHAPE, "kvq_upsample_frames: empty output");
  UpsampleParams p{};
  p.video = video; p.out = out; p.src_is_u8 = src_is_u8; p.C = C; p.T = T; p.H = H; p.W = W; p.OH = od[0]; p.OW = od[1];
  p.scale = (float)(1.0 / scale_factor);
  const long total = (long)C * T * p.OH * p.OW;
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(upsample_frames_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  KVQ_CHECK_LAUNCH("upsample_frames_kernel");
  return KVQ_OK;
}

extern "C" int kvq_patch_im2col(const float* x, int B, int Cin, int T, int H, int W, int pd, int ph, int pw,
                                int dtype, uint16_t* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(x && out, KVQ_ERR_NULL, "kvq_patch_im2col: NULL pointer");
  KVQ_REQUIRE(B > 0 && Cin > 0 && T > 0 && H > 0 && W > 0 && pd > 0 && ph > 0 && pw > 0, KVQ_ERR_SHAPE,
              "kvq_patch_im2col: bad shape");
  const int D = ceil_div(T, pd), Hh = ceil_div(H, ph), Ww = ceil_div(W, pw);
  const int K = Cin * pd * ph * pw;
  const size_t lds = (size_t)Ww * K * sizeof(uint16_t);
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_patch_im2col: dtype %d", dtype);
  KVQ_REQUIRE(K % 8 == 0 && lds <= 64 * 1024, KVQ_ERR_UNSUPPORTED,
              "kvq_patch_im2col: need K%%8==0 and a token row (%d x %d) of at most 64 KiB", Ww, K);
  const int grid = B * D * Hh;
  if (dtype == KVQ_DT_FP16)
    hipLaunchKernelGGL(patch_im2col_kernel<Fp16>, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, B, Cin, T, H, W,
                       pd, ph, pw, D, Hh, Ww, out);
  else
    hipLaunchKernelGGL(patch_im2col_kernel<Bf16>, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, B, Cin, T, H, W,
                       pd, ph, pw, D, Hh, Ww, out);
  KVQ_CHECK_LAUNCH("patch_im2col_kernel");
  return KVQ_OK;
}

extern "C" int kvq_vqa_head(const float* feat, int B, int L, int C, int64_t stride_b, int64_t stride_l,
                            int64_t stride_c, const float* w1t, const float* w1, const float* b1, int hidden, const float* w2,
                            const float* b2, float* scratch, float* score, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(feat && (w1t || w1) && b1 && w2 && scratch && score, KVQ_ERR_NULL, "kvq_vqa_head: NULL pointer");
  KVQ_REQUIRE(B > 0 && L > 0 && C > 0 && hidden > 0, KVQ_ERR_SHAPE, "kvq_vqa_head: bad shape");
  const bool mfma = w1 && stride_c == 1 && hidden == 64 && C % 64 == 0 && stride_b % 4 == 0 && stride_l % 4 == 0 &&
                    (((size_t)feat | (size_t)w1) & 15) == 0;
  KVQ_REQUIRE(mfma || w1t, KVQ_ERR_NULL, "kvq_vqa_head: this shape takes the VALU kernel, which reads w1t");
  if (mfma) {
    hipLaunchKernelGGL(vqa_head_mfma_kernel, dim3((unsigned)(((long)B * L + 15) / 16)), dim3(256), 0, (hipStream_t)stream, feat, B, L, C,
                       (long)stride_b, (long)stride_l, w1, b1, w2, scratch);
    KVQ_CHECK_LAUNCH("vqa_head_mfma_kernel");
  } else {
    const int grid = (int)(((long)B * L + HEAD_TOK - 1) / HEAD_TOK);
    hipLaunchKernelGGL(vqa_head_token_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, B, L, C,
                       (long)stride_b, (long)stride_l, (long)stride_c, w1t, b1, hidden, w2, scratch);
    KVQ_CHECK_LAUNCH("vqa_head_token_kernel");
  }
  hipLaunchKernelGGL(mean_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scratch, L, b2, score);
  KVQ_CHECK_LAUNCH("mean_rows_kernel");
  return KVQ_OK;
}

namespace kvq {
// ------------------------------------------------------------------------------------------------
// VQAHead's other two branches (models/head.py:60-68): pre_pool (AdaptiveAvgPool3d((1,1,1)) in front of the MLP) and
// num_class > 1 (nn.Softmax() with its implicit dim — dim 1, the classes, for a 5-D input — over fc_last's K outputs, then the
// mean over the token grid).  No reference config sets either, so this is a plain fp32 VALU path: lane j owns hidden unit j
// (W1 transposed, one coalesced row per channel), one wave per token, the K logits meet in LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void feat_mean_kernel(const float* __restrict__ feat, int L, int C, long sb, long sl, long sc,
                                                        float* __restrict__ pooled) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* f = feat + b * sb + c * sc;
  float s = 0.f;
  for (int l = 0; l < L; ++l) s += f[l * sl];                 // sequential in l: the order of a plain sum, deterministic
  pooled[(size_t)b * C + c] = s / (float)L;
}

__global__ __launch_bounds__(64) void vqa_head_classes_kernel(const float* __restrict__ feat, int L, int C, long sb, long sl, long sc,
                                                              const float* __restrict__ w1t, const float* __restrict__ b1, int hidden,
                                                              const float* __restrict__ w2, const float* __restrict__ b2, int K,
                                                              float* __restrict__ tok_prob) {
  extern __shared__ float logit[];                            // [K]
  const int lane = threadIdx.x;
  const long tk = blockIdx.x;
  const long b = tk / L, l = tk - b * L;
  const float* row = feat + b * sb + l * sl;
  for (int k = lane; k < K; k += 64) logit[k] = 0.f;
  __syncthreads();
  for (int j0 = 0; j0 < hidden; j0 += 64) {
    const int j = j0 + lane;
    const bool live = j < hidden;
    const int jc = live ? j : hidden - 1;
    float d = 0.f;
    for (int c = 0; c < C; ++c) d = fmaf(row[c * sc], w1t[(size_t)c * hidden + jc], d);
    const float h = live ? gelu_erf(d + b1[jc]) : 0.f;
    for (int k = 0; k < K; ++k) {
      float v = live ? w2[(size_t)k * hidden + jc] * h : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) logit[k] += v;
    }
  }
  __syncthreads();
  if (lane == 0) {
    float* out = tok_prob + tk * K;
    if (K == 1) {
      out[0] = logit[0] + b2[0];
    } else {
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) m = fmaxf(m, logit[k] + b2[k]);
      float z = 0.f;
      for (int k = 0; k < K; ++k) z += expf(logit[k] + b2[k] - m);
      for (int k = 0; k < K; ++k) out[k] = expf(logit[k] + b2[k] - m) / z;
    }
  }
}

// score[b][k] = mean_l tok_prob[(b*L + l)*K + k]
__global__ __launch_bounds__(256) void mean_classes_kernel(const float* __restrict__ v, int L, int K, float* __restrict__ out) {
  __shared__ float red[256];
  const int b = blockIdx.x, k = blockIdx.y;
  float s = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) s += v[((size_t)b * L + l) * K + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)b * K + k] = red[0] / (float)L;
}

}  // namespace kvq

extern "C" int kvq_vqa_head_classes(const float* feat, int B, int L, int C, int64_t stride_b, int64_t stride_l, int64_t stride_c,
                                    const float* w1t, const float* b1, int hidden, const float* w2, const float* b2, int num_class,
                                    int pre_pool, float* scratch, float* score, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(feat && w1t && b1 && w2 && b2 && scratch && score, KVQ_ERR_NULL, "kvq_vqa_head_classes: NULL pointer");
  KVQ_REQUIRE(B > 0 && L > 0 && C > 0 && hidden > 0 && num_class > 0 && num_class <= 4096, KVQ_ERR_SHAPE,
              "kvq_vqa_head_classes: bad shape");
  long sb = stride_b, sl = stride_l, sc = stride_c;
  int Lh = L;
  float* tok = scratch;
  if (pre_pool) {                     // scratch = pooled [B][C] | token probabilities [B][num_class]
    hipLaunchKernelGGL(feat_mean_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, feat, L, C, sb, sl, sc, scratch);
    KVQ_CHECK_LAUNCH("feat_mean_kernel");
    feat = scratch;
    tok = scratch + (size_t)B * C;
    sb = C, sl = 0, sc = 1, Lh = 1;
  }
  hipLaunchKernelGGL(vqa_head_classes_kernel, dim3((unsigned)((long)B * Lh)), dim3(64), num_class * sizeof(float), (hipStream_t)stream,
                     feat, Lh, C, sb, sl, sc, w1t, b1, hidden, w2, b2, num_class, tok);
  KVQ_CHECK_LAUNCH("vqa_head_classes_kernel");
  hipLaunchKernelGGL(mean_classes_kernel, dim3(B, num_class), dim3(256), 0, (hipStream_t)stream, tok, Lh, num_class, score);
  KVQ_CHECK_LAUNCH("mean_classes_kernel");
  return KVQ_OK;
}

extern "C" int kvq_simple_vqa_head(const float* feat, int B, int T, int Cin, const float* w1, const float* b1,
                                   int hidden, const float* w2, const float* b2, float* scratch, float* score,
                                   void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(feat && w1 && b1 && w2 && scratch && score, KVQ_ERR_NULL, "kvq_simple_vqa_head: NULL pointer");
  KVQ_REQUIRE(B > 0 && T > 0 && Cin > 0 && hidden > 0, KVQ_ERR_SHAPE, "kvq_simple_vqa_head: bad shape");
  hipLaunchKernelGGL(simple_head_frame_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, feat, Cin, w1, b1,
                     hidden, w2, scratch);
  KVQ_CHECK_LAUNCH("simple_head_frame_kernel");
  hipLaunchKernelGGL(mean_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scratch, T, b2, score);
  KVQ_CHECK_LAUNCH("mean_rows_kernel");
  return KVQ_OK;
}

extern "C" int kvq_fragment_gather(const void* video, int src_is_u8, int C, int T, int H, int W,
                                   const int32_t* hoff, const int32_t* woff, int Fh, int Fw, int fs_h, int fs_w,
                                   int aligned, const float* host_mean, const float* host_std, float* out,
                                   void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(video && hoff && woff && out, KVQ_ERR_NULL, "kvq_fragment_gather: NULL pointer");
  KVQ_REQUIRE(C > 0 && C <= 4 && T > 0 && Fh > 0 && Fw > 0 && fs_h > 0 && fs_w > 0 && aligned > 0, KVQ_ERR_SHAPE,
              "kvq_fragment_gather: bad shape");
  // reference: assert dur_t % aligned == 0, "Please provide match vclip and align index" (fusion_datasets.py:60)
  KVQ_REQUIRE(T % aligned == 0, KVQ_ERR_SHAPE, "Please provide match vclip and align index");
  // a source smaller than the canvas is legal (the caller ran the upsample fallback, fusion_datasets.py:43-50, whose output may
  // stay one pixel short of the canvas: floor(H * scale)); what must hold is that a mini-patch fits
  KVQ_REQUIRE(H >= fs_h && W >= fs_w, KVQ_ERR_UNSUPPORTED, "kvq_fragment_gather: source %dx%d smaller than one %dx%d mini-patch", H, W, fs_h, fs_w);
  FragParams p{};
  p.video = video; p.src_is_u8 = src_is_u8; p.C = C; p.T = T; p.H = H; p.W = W;
  p.hoff = hoff; p.woff = woff; p.Fh = Fh; p.Fw = Fw; p.fsh = fs_h; p.fsw = fs_w; p.aligned = aligned;
  p.normalise = host_std != nullptr;
  for (int c = 0; c < C; ++c) {
    p.mean[c] = host_mean ? host_mean[c] : 0.f;
    p.std[c] = host_std ? host_std[c] : 1.f;
  }
  p.out = out;
  const long plane = (long)Fh * fs_h * Fw * fs_w;
  KVQ_REQUIRE(plane < (1L << 30) && (long)C * T < 65536, KVQ_ERR_SHAPE, "kvq_fragment_gather: output plane / plane count too large");
  const bool vec = fs_w % 4 == 0 && ((size_t)out & 15) == 0;
  p.chan_stride = (long)T * H * W;
  FragBatch none{};
  if (vec) hipLaunchKernelGGL((fragment_gather_kernel<4, false>), dim3((unsigned)((plane / 4 + 255) / 256), (unsigned)(C * T)), dim3(256), 0, (hipStream_t)stream, p, none);
  else hipLaunchKernelGGL((fragment_gather_kernel<1, false>), dim3((unsigned)((plane + 255) / 256), (unsigned)(C * T)), dim3(256), 0, (hipStream_t)stream, p, none);
  KVQ_CHECK_LAUNCH("fragment_gather_kernel");
  return KVQ_OK;
}

extern "C" int kvq_fragment_gather_batch(const KvqFragmentSource* f, int C, int T, float* out, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(f && out, KVQ_ERR_NULL, "kvq_fragment_gather_batch: NULL pointer");
  KVQ_REQUIRE(!f->indirect, KVQ_ERR_UNSUPPORTED, "kvq_fragment_gather_batch: a source with an indirect pointer table (the fused read's form)");
  KVQ_REQUIRE(f->n_clips > 0 && f->n_clips <= KVQ_FRAG_MAX_CLIPS && C > 0 && C <= 4 && T > 0 && f->Fh > 0 && f->Fw > 0 && f->fs_h > 0 &&
                  f->fs_w > 0 && f->aligned > 0, KVQ_ERR_SHAPE, "kvq_fragment_gather_batch: bad shape");
  KVQ_REQUIRE(T % f->aligned == 0, KVQ_ERR_SHAPE, "Please provide match vclip and align index");
  KVQ_REQUIRE(f->Hs >= f->fs_h && f->Ws >= f->fs_w, KVQ_ERR_UNSUPPORTED,
              "kvq_fragment_gather_batch: source %dx%d smaller than one %dx%d mini-patch", f->Hs, f->Ws, f->fs_h, f->fs_w);
  KVQ_REQUIRE(f->chan_stride == 0 || f->chan_stride >= (int64_t)T * f->Hs * f->Ws, KVQ_ERR_SHAPE, "kvq_fragment_gather_batch: channel stride");
  FragParams p{};
  FragBatch fb{};
  for (int b = 0; b < f->n_clips; ++b) {
    KVQ_REQUIRE(f->video[b] && f->hoff[b] && f->woff[b], KVQ_ERR_NULL, "kvq_fragment_gather_batch: clip %d has a NULL pointer", b);
    fb.video[b] = f->video[b]; fb.hoff[b] = f->hoff[b]; fb.woff[b] = f->woff[b];
  }
  p.src_is_u8 = f->src_is_u8; p.C = C; p.T = T; p.H = f->Hs; p.W = f->Ws;
  p.Fh = f->Fh; p.Fw = f->Fw; p.fsh = f->fs_h; p.fsw = f->fs_w; p.aligned = f->aligned;
  p.normalise = f->normalise;
  for (int c = 0; c < C; ++c) { p.mean[c] = f->normalise ? f->mean[c] : 0.f; p.std[c] = f->normalise ? f->std[c] : 1.f; }
  p.out = out;
  p.chan_stride = f->chan_stride ? f->chan_stride : (long)T * f->Hs * f->Ws;
  const long plane = (long)f->Fh * f->fs_h * f->Fw * f->fs_w;
  KVQ_REQUIRE(plane < (1L << 30) && (long)C * T < 65536, KVQ_ERR_SHAPE, "kvq_fragment_gather_batch: output plane / plane count too large");
  const bool vec = f->fs_w % 4 == 0 && ((size_t)out & 15) == 0;
  if (vec) hipLaunchKernelGGL((fragment_gather_kernel<4, true>), dim3((unsigned)((plane / 4 + 255) / 256), (unsigned)(C * T), (unsigned)f->n_clips), dim3(256), 0, (hipStream_t)stream, p, fb);
  else hipLaunchKernelGGL((fragment_gather_kernel<1, true>), dim3((unsigned)((plane + 255) / 256), (unsigned)(C * T), (unsigned)f->n_clips), dim3(256), 0, (hipStream_t)stream, p, fb);
  KVQ_CHECK_LAUNCH("fragment_gather_kernel");
  return KVQ_OK;
}

// ---- trilinear resize, channels-last (the torch.cat of multi=True feature taps, swin_backbone.py:1070-1075) --------
namespace kvq {
// ATen area_pixel_compute_source_index, align_corners = False, linear modes: max(0, scale * (dst + 0.5) - 0.5)
__device__ __forceinline__ void lin_src(int dst, int in, int out, int& i0, int& i1, float& lam) {
  if (in == out) { i0 = i1 = dst; lam = 0.f; return; }
  const float scale = (float)in / (float)out;
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  lam = s - (float)i0;
}

__global__ void resize_trilinear_cl_kernel(const float* __restrict__ src, int D, int H, int W, int C, float* __restrict__ dst,
                                           int Do, int Ho, int Wo, int c_total, int c_off, long total) {
  const long gi = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total) return;
  const int c = (int)(gi % C);
  long r = gi / C;
  const int wo = (int)(r % Wo); r /= Wo;
  const int ho = (int)(r % Ho); r /= Ho;
  const int dd = (int)(r % Do);
  const long b = r / Do;
  int d0, d1, h0, h1, w0, w1;
  float ld, lh, lw;
  lin_src(dd, D, Do, d0, d1, ld);
  lin_src(ho, H, Ho, h0, h1, lh);
  lin_src(wo, W, Wo, w0, w1, lw);
  auto at = [&](int d, int h, int w) { return src[(((b * D + d) * H + h) * (long)W + w) * C + c]; };
  // ATen upsample_trilinear3d order: w, then h, then d
  const float v00 = (1.f - lw) * at(d0, h0, w0) + lw * at(d0, h0, w1);
  const float v01 = (1.f - lw) * at(d0, h1, w0) + lw * at(d0, h1, w1);
  const float v10 = (1.f - lw) * at(d1, h0, w0) + lw * at(d1, h0, w1);
  const float v11 = (1.f - lw) * at(d1, h1, w0) + lw * at(d1, h1, w1);
  const float v0 = (1.f - lh) * v00 + lh * v01, v1 = (1.f - lh) * v10 + lh * v11;
  dst[(((b * Do + dd) * Ho + ho) * (long)Wo + wo) * c_total + c_off + c] = (1.f - ld) * v0 + ld * v1;
}
}  // namespace kvq

extern "C" int kvq_resize_trilinear_cl(const float* src, int B, int D, int H, int W, int C, float* dst, int Do, int Ho, int Wo,
                                       int c_total, int c_off, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(src && dst, KVQ_ERR_NULL, "kvq_resize_trilinear_cl: NULL pointer");
  KVQ_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Do > 0 && Ho > 0 && Wo > 0 && c_off >= 0 && c_off + C <= c_total,
              KVQ_ERR_SHAPE, "kvq_resize_trilinear_cl: bad geometry");
  const long total = (long)B * Do * Ho * Wo * C;
  hipLaunchKernelGGL(resize_trilinear_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     D, H, W, C, dst, Do, Ho, Wo, c_total, c_off, total);
  KVQ_CHECK_LAUNCH("resize_trilinear_cl_kernel");
  return KVQ_OK;
}
