// Shared between tail.hip (token per lane, C <= 192) and tailmm.hip (feature-sliced GEMM chain, C >= 256): launch parameters of the fused
// post-attention launch and the entry points of the wide variant.
#pragma once
#include "common.hpp"

namespace kvq {

struct TailParams {
  const uint16_t* attn;      // [M][C] 16-bit, window order
  float* x;                  // [n_batch*out_rows][C] fp32, in place (x16: the same rows as fp16, 2 C bytes each)
  int x16;                   // round 6: the residual stream of this stage is kept in fp16 (csrc/tail.hip, C <= 192)
  const int32_t* map;        // window row -> token of the batch element (or <0 = padding); NULL = identity
  const int32_t* gather;     // token -> window row (attn_gather): the launch walks n_tok tokens instead of M window rows
  int n_tok;
  int map_rows, out_rows, M, hidden;
  const unsigned char* pack; // kvq_block_tail_pack image
  const float* nn_w;         // next block's norm1 (EMIT)
  const float* nn_b;
  const int32_t* next_dst;   // token -> window row of the next block's partition
  uint16_t* next_ln;         // [n_batch*next_rows][C]
  int next_rows;
  // instead of next_ln: the next block's q | k | v, head-major [3][nH][n_batch*next_rows][32] in ITS window order (tailmm.hip, round 5)
  const unsigned char* qkv_pack;   // kvq_block_tail_qkv_pack image of the NEXT block's qkv weight
  const float* qkv_b;              // [3C]
  uint16_t* qkv_out;
  float q_scale;
  int num_heads;
  long qkv_rows;                   // n_batch * next_rows
  float eps;
  unsigned long long* trace;   // diagnostic stamps (kvq_debug_gemm_trace; -DKVQ_TAIL_TRACE builds only)
  int trace_blocks;
};

// csrc/tailmm.hip: the same launch for C = 256 / 384 / 512 as a register-blocked 32x32x16 GEMM chain
bool tailmm_supported(int C, int hidden);
size_t tailmm_pack_bytes(int C, int hidden);
int tailmm_pack(const uint16_t* wp, const uint16_t* w1, const uint16_t* w2, const float* proj_b, const float* n2w, const float* n2b,
                const float* b1, const float* b2, int C, int hidden, unsigned char* out, hipStream_t st);
int tailmm_launch(const TailParams& p, int C, int dtype, hipStream_t st);
int tailmm_geometry_code(int C, int hidden);           // 0: not a tailmm width; else (hidden chunk / 128) * 10 + token tiles of 32 per workgroup (profile records)
size_t tailmm_qkv_pack_bytes(int C, int hidden);      // 0: this width cannot emit q | k | v
int tailmm_qkv_pack(const uint16_t* qkv_w, int C, int hidden, unsigned char* out, hipStream_t st);

}  // namespace kvq
