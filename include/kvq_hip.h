/*
 * kvq_hip.h — C ABI of libkvq_hip.so: the MI355X (gfx950) hot path of the KVQ per-video
 * forward (fragment sampler -> Swin-3D(GRPB) trunk -> VQAHead; SimpleVQA head).
 *
 * The reference is pure PyTorch: there is no native FFI to mirror, so each entry point
 * cites the reference *Python* call it replaces (file:line under /root/reference).
 * The binding a maintainer adds on the reference side is a ctypes stub — INTEGRATION.md.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless marked "host"; the caller (PyTorch) owns all
 *     buffers, the library borrows them for the duration of a call and never frees them;
 *   - scratch is caller-provided (kvq_swin3d_workspace_bytes), except the small integer index
 *     maps a plan owns (allocated in kvq_swin3d_plan_create, freed in kvq_swin3d_plan_destroy);
 *   - every launch goes to the hipStream_t passed in (void* here so the header needs no HIP
 *     include); calls are asynchronous, never synchronise the device, and are graph-capturable;
 *   - return value: 0 = ok, <0 = KvqStatus; kvq_last_error() gives the message (thread-local);
 *   - 16-bit operands (bf16 or fp16, see KvqDtype) travel as raw uint16_t bit patterns
 *     (round-to-nearest-even of fp32).
 *   - no exceptions / abort() cross this boundary.
 */
#ifndef KVQ_HIP_H
#define KVQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVQ_ABI_VERSION 31
#define KVQ_MAX_STAGES 4

/* 16-bit MFMA operand type of every uint16_t buffer below (activations AND weights of one call must
 * agree).  Same MFMA rate and bytes; fp16 (11-bit mantissa, saturating conversions) is the default of
 * the host wrapper because it holds the 1e-3 MOS parity gate where bf16 (8-bit mantissa) does not. */
typedef enum { KVQ_DT_BF16 = 0, KVQ_DT_FP16 = 1 } KvqDtype;

typedef enum {
  KVQ_OK = 0,
  KVQ_ERR_NULL = -1,        /* required pointer is NULL                    */
  KVQ_ERR_SHAPE = -2,       /* unsupported / inconsistent shape            */
  KVQ_ERR_UNSUPPORTED = -3, /* configuration outside what the kernels cover */
  KVQ_ERR_WORKSPACE = -4,   /* workspace too small                         */
  KVQ_ERR_HIP = -5          /* a HIP runtime call failed                   */
} KvqStatus;

int kvq_abi_version(void);
const char* kvq_last_error(void);
/* Name of the device the library sees ("" if none); fills at most n bytes. */
int kvq_device_name(char* host_buf, int n);

/* ---------------------------------------------------------------------------------------------
 * Trunk configuration = constructor kwargs of SwinTransformer3D
 * (models/backbones/swin_backbone.py:760-783).  head_dim must be 32 (it is for every
 * reference configuration: embed_dim*2^i / num_heads[i] == 32).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t patch[3];                  /* (2,4,4) */
  int32_t in_chans;                  /* 3 */
  int32_t embed_dim;                 /* 96 (tiny/small), 128 (base) */
  int32_t num_stages;                /* 4 */
  int32_t depths[KVQ_MAX_STAGES];    /* 2,2,6,2 */
  int32_t num_heads[KVQ_MAX_STAGES]; /* 3,6,12,24 */
  int32_t window[3];                 /* (8,7,7) */
  int32_t mlp_ratio;                 /* 4 */
  int32_t frag_bias[KVQ_MAX_STAGES]; /* 1,1,1,0 for the GRPB trunk; 0,0,0,0 for swin_3d_tiny */
  int32_t adaptive_window[3];        /* ABI 30.  (0,0,0): off.  Otherwise forward(adaptive_window_size=True)'s resized window
                                      * (swin_backbone.py:54-61, :1050-1055; each entry <= window[]): every stage partitions by
                                      * it (clamped per stage as usual, :408-413, :667-671) while the shift stays window[]/2, and
                                      * a token's bias index is its coordinate INSIDE the resized window
                                      * (relative_position_index[:d,:h,:w,:d,:h,:w], :266-271) instead of the [:N,:N] slice */
} KvqSwinCfg;

/* Weights of one SwinTransformerBlock3D (swin_backbone.py:385-405).  GEMM weights are bf16
 * [out][in] exactly as nn.Linear stores them; everything else fp32. */
typedef struct {
  const float* norm1_w;    /* [C] */
  const float* norm1_b;
  const float* rpb_table;  /* relative_position_bias_table  [table_len][nH] */
  const float* fpb_table;  /* fragment_position_bias_table  [table_len][nH] or NULL */
  const float* bias_pack;  /* optional, derived: [nH][TLe][2], TLe = table_len rounded up to even,
                              = {fpb (rpb if no fpb), rpb-fpb (0)}: a head's table stages into LDS as one
                              contiguous 16-B aligned copy; NULL = gather from the raw tables */
  const uint16_t* qkv_w;   /* [3C][C], rows = [q | k | v], each head-major */
  const float* qkv_b;      /* [3C] */
  const uint16_t* proj_w;  /* [C][C] */
  const float* proj_b;
  const float* norm2_w;
  const float* norm2_b;
  const uint16_t* fc1_w;   /* [4C][C] */
  const float* fc1_b;
  const uint16_t* fc2_w;   /* [C][4C] */
  const float* fc2_b;
  const void* tail_pack;   /* optional, derived: kvq_block_tail_pack image of (proj, norm2, fc1, fc2).  Non-NULL (and
                              a fused width: kvq_block_tail_supported) -> proj+residual+norm2+Mlp+residual run as ONE launch
                              (kvq_block_tail); NULL -> GEMM/LayerNorm launches */
  const void* qkv_pack;    /* optional, derived: kvq_block_tail_qkv_pack image of THIS block's qkv_w.  Non-NULL -> the fused tail launch
                              of the PREVIOUS block of the stage writes this block's q | k | v itself (no norm1 rows, no qkv GEMM launch) */
  const void* bias_dense;  /* optional, derived, valid ONLY with the plan it was built for
                              (kvq_swin3d_bias_dense_build): the block's attention bias per (window type, head), gate,
                              shift mask and padding included.  Non-NULL -> kvq_window_attention32; NULL -> the
                              bias is rebuilt per score from the tables (kvq_window_attention) */
} KvqSwinBlockW;

typedef struct {            /* PatchMerging (swin_backbone.py:527-531) */
  const float* norm_w;      /* [4C] */
  const float* norm_b;
  const uint16_t* red_w;    /* [2C][4C], no bias */
  const void* merge_pack;   /* optional, derived: kvq_patch_merge_pack image; non-NULL (and kvq_patch_merge_supported(C)) ->
                               the merge's LayerNorm + GEMM (+ the next block's norm1) run as ONE launch */
} KvqSwinMergeW;

typedef struct {
  const uint16_t* embed_w;  /* patch_embed.proj.weight as bf16 [E][in*pd*ph*pw] */
  const float* embed_b;     /* [E] */
  const float* embed_ln_w;  /* patch_embed.norm */
  const float* embed_ln_b;
  const void* embed_pack;   /* optional, derived: kvq_patch_embed_pack image; non-NULL (and a fused shape, see
                               kvq_patch_embed_supported) -> im2col + GEMM + LayerNorm run as ONE launch */
  const KvqSwinBlockW* blocks; /* host array, sum(depths) entries, stage-major */
  KvqSwinMergeW merges[KVQ_MAX_STAGES - 1];
  const float* norm_w;      /* final LayerNorm [C_out] */
  const float* norm_b;
} KvqSwinWeights;

/* Opaque plan: all shape-dependent integer index maps (window gather/scatter with shift,
 * padding and crop; per-token fragment ids and shift-mask regions; patch-merge neighbours)
 * for one (cfg, B, T, H, W).  Replaces the lru_cached tensors of compute_mask
 * (swin_backbone.py:559-586) and global_position_index (:21-50). */
typedef struct KvqSwinPlan KvqSwinPlan;

int kvq_swin3d_plan_create(const KvqSwinCfg* cfg, int B, int T, int H, int W, int dtype /*KvqDtype*/,
                           KvqSwinPlan** out);
void kvq_swin3d_plan_destroy(KvqSwinPlan* plan);
size_t kvq_swin3d_workspace_bytes(const KvqSwinPlan* plan);
/* Output geometry: C_out, D, H', W' of the (B, C_out, D, H', W') feature map. */
int kvq_swin3d_out_dims(const KvqSwinPlan* plan, int32_t out4[4]);

#define KVQ_FRAG_MAX_CLIPS 16
/* A batch of clips that is still (decoded frames, sampler draws): what kvq_fragment_gather would be called with, clip by clip.
 * kvq_patch_embed / kvq_swin3d_forward_fragments read the patch-embedding operand straight from it — get_spatial_fragments
 * (fusion_datasets.py:22-121) + (v - mean) / std (:1017-1020) happen in registers, with the same fp32 arithmetic, and the fp32
 * (B,3,T,H,W) clip (4 B/pixel written, 4 B/pixel read back) never exists. */
typedef struct {
  const void* video[KVQ_FRAG_MAX_CLIPS];     /* clip b: uint8 (C, T, Hs, Ws), device; frames contiguous, channel planes
                                                chan_stride ELEMENTS apart (a clip may be a run of frames of a longer video) */
  const int32_t* hoff[KVQ_FRAG_MAX_CLIPS];   /* clip b: int32 [Fh][Fw][T/aligned] absolute patch origins, device    */
  const int32_t* woff[KVQ_FRAG_MAX_CLIPS];
  int64_t chan_stride;                       /* in ELEMENTS of the frame type; 0 = T * Hs * Ws (contiguous clips)    */
  int32_t n_clips, src_is_u8, Hs, Ws, Fh, Fw, fs_h, fs_w, aligned;
  int32_t normalise;                         /* 0: raw pixel values                                                  */
  float mean[4], std[4];
  const void* const* indirect;               /* ABI 30.  NULL, or a DEVICE array of 3 * KVQ_FRAG_MAX_CLIPS pointers — video[16] | hoff[16] |
                                                woff[16] — that the embedding launch reads INSTEAD of the three arrays above (which may
                                                then stay NULL): the launch parameters no longer carry per-video addresses, so a recorded
                                                hipGraph of the forward serves every video — the caller rewrites the 384-byte table on the
                                                stream in front of each replay.  kvq_fragment_gather_batch refuses it. */
} KvqFragmentSource;

/* SwinTransformer3D.forward (swin_backbone.py:1044-1080), multi=False, layer=-1.
 *   x     fp32 (B,3,T,H,W) contiguous                      — batch['technical']
 *   feat  fp32 channels-LAST (B, D, H', W', C_out); the host wrapper returns the
 *         (B,C_out,D,H',W') permuted view, which is what the reference returns.
 * When score != NULL the VQAHead (models/head.py:60-68) is applied too (see kvq_vqa_head). */
int kvq_swin3d_forward(const KvqSwinPlan* plan, const KvqSwinWeights* w, const float* x, float* feat,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same forward on a batch that is still (frames, sampler draws) (KvqFragmentSource above): K1 fused into the patch
 * embedding's operand read.  Bit-identical to kvq_fragment_gather per clip + kvq_swin3d_forward.  KVQ_ERR_UNSUPPORTED when
 * the plan does not take the fused embedding launch or kvq_patch_embed_fragments_supported says no (the caller then runs
 * the two calls). */
int kvq_swin3d_forward_fragments(const KvqSwinPlan* plan, const KvqSwinWeights* w, const KvqFragmentSource* src,
                                 float* feat, void* workspace, size_t workspace_bytes, void* stream);

/* Feature taps of SwinTransformer3D.forward (swin_backbone.py:1060-1078: ``feats = [embed, stage 0, ..., stage n-1]``,
 * read by ``multi=True`` and ``layer > -1``).  taps[i] (i = 0..num_stages) is NULL or a caller-owned fp32 channels-LAST
 * buffer (B, D, H_i, W_i, C_i) that every following kvq_swin3d_forward on this plan fills with feats[i] (the residual
 * stream after the patch embedding / after stage i-1 including its PatchMerging); taps == NULL clears them.
 * kvq_swin3d_tap_dims: C_i, D, H_i, W_i. */
int kvq_swin3d_set_taps(KvqSwinPlan* plan, float* const* taps);
int kvq_swin3d_tap_dims(const KvqSwinPlan* plan, int index, int32_t out4[4]);

/* F.interpolate(mode="trilinear", align_corners=False) on a channels-last fp32 volume, written into channels
 * [c_off, c_off + C) of a channels-last destination with c_total channels (the torch.cat of multi=True,
 * swin_backbone.py:1070-1075).  src (B, D, H, W, C) -> dst (B, Do, Ho, Wo, c_total). */
int kvq_resize_trilinear_cl(const float* src, int B, int D, int H, int W, int C, float* dst, int Do, int Ho, int Wo,
                            int c_total, int c_off, void* stream);

/* Stages stage_lo .. stage_hi only (0-based, inclusive) — what KSVQE.forward needs to modulate the stream between stages
 * (KSVQE_model.py:1433-1486).  stage_lo == 0 starts from the clip x; otherwise from io, the residual stream in front of
 * stage_lo: fp32 channels-last (B, D, H, W, C) of that stage (kvq_swin3d_tap_dims(plan, stage_lo)).  On return io (may be
 * NULL when feat is taken) holds the stream behind stage_hi (its PatchMerging included: kvq_swin3d_tap_dims(plan,
 * stage_hi + 1)); feat (may be NULL) additionally receives the final LayerNorm when stage_hi is the last stage. */
int kvq_swin3d_forward_stages(const KvqSwinPlan* plan, const KvqSwinWeights* w, const float* x, int stage_lo, int stage_hi,
                              float* io, float* feat, void* workspace, size_t workspace_bytes, void* stream);

/* Dense attention bias of weights->blocks[block] for this plan's geometry (see kvq_attn_bias32_build): size
 * and builder.  Independent of the batch size; rebuild when the block's tables change. */
size_t kvq_swin3d_bias_dense_bytes(const KvqSwinPlan* plan, int block);
int kvq_swin3d_bias_dense_build(const KvqSwinPlan* plan, int block, const float* rpb_table, const float* fpb_table,
                                void* out, float* max_abs /* device, may be NULL: see kvq_attn_bias32_build */,
                                void* stream);

/* Per-kernel-class GPU time of the most recent profiled forward.  Profiling brackets every
 * launch with hipEvents on the launch stream; enable with kvq_swin3d_profile(plan, 1). */
enum {
  KVQ_K_IM2COL = 0, KVQ_K_LAYERNORM, KVQ_K_GEMM_QKV, KVQ_K_ATTN, KVQ_K_GEMM_PROJ, KVQ_K_GEMM_FC1,
  KVQ_K_GEMM_FC2, KVQ_K_GEMM_MERGE, KVQ_K_GEMM_EMBED, KVQ_K_TAIL, KVQ_K_EMBED, KVQ_K_MERGE, KVQ_K_COUNT
};
typedef struct {
  int32_t kind;     /* KVQ_K_*                                                                    */
  int32_t variant;  /* kernel instantiation: GEMM (MI*100+BK)*10+epilogue; attention 2*gated+mask */
  float ms;         /* hipEventElapsedTime around the launch, on the launch stream                */
  double flops;     /* algorithmic flops of the launch (2MNK; attention 4*rows*N*C)               */
  double bytes;     /* algorithmic HBM bytes of the launch (operands read once, result written once) */
} KvqProfRecord;
int kvq_swin3d_profile(KvqSwinPlan* plan, int enable);
/* Synchronises the recorded events (host blocks) and copies up to max_records per-launch records
 * (in launch order) of every forward issued since profiling was enabled / last read. */
int kvq_swin3d_profile_read(KvqSwinPlan* plan, KvqProfRecord* host_out, int max_records, int* n_records);

/* ---------------------------------------------------------------------------------------------
 * Individual kernels (exported for the parity tests; kvq_swin3d_forward is built from them).
 * ------------------------------------------------------------------------------------------- */

/* LayerNorm over gathered rows: nn.LayerNorm + F.pad + torch.roll + window_partition
 * (swin_backbone.py:416-449), norm2 (:491), PatchMerging's 4-neighbour concat + norm (:546-552)
 * and the final norm (:1066-1068).
 *   x        fp32 [n_batch * rows_in][Cin]
 *   map      int32 [rows_out][nparts] source row within a batch element, -1 = zero; NULL = identity
 *   out row r of batch b = LN(concat_p x[b*rows_in + map[r][p]]) over nparts*Cin channels.
 *   nparts==1 and map<0  -> the whole output row is 0 (pad AFTER the norm, :424);
 *   nparts>1  and map<0  -> that part is 0 and takes part in the statistics (pad BEFORE, :544).
 *   out_h (16-bit, type = dtype) / out_f32: exactly one is non-NULL. */
int kvq_layernorm_rows(const float* x, const int32_t* map, int nparts, int n_batch, int rows_in,
                       int rows_out, int Cin, const float* gamma, const float* beta, float eps,
                       uint16_t* out_h, int dtype, float* out_f32, void* stream);

/* bf16 MFMA GEMM  acc[m][n] = sum_k A[m][k] * W[n][k]  (fp32 accumulate) with a fused epilogue. */
typedef enum {
  KVQ_EPI_BIAS_BF16 = 0,   /* out_bf16[m][n] = acc + bias                      (generic Linear)      */
  KVQ_EPI_GELU_BF16 = 1,   /* out_bf16[m][n] = gelu_erf(acc + bias)            (Mlp.fc1+act, :84-85) */
  KVQ_EPI_QKV_BF16 = 2,    /* head-major split: out[(which*nH+h)*M + m][e], q scaled (:253-260)      */
  KVQ_EPI_RESID_F32 = 3,   /* out_f32[row(m)][n] += acc + bias, row(m) via scatter map (:472-488,509,514) */
  KVQ_EPI_STORE_F32 = 4,   /* out_f32[m][n] = acc (+ bias if non-NULL)         (reduction :553, embed) */
  KVQ_EPI_RELU_BF16 = 5,   /* out_bf16[m][n] = relu(acc + bias [+ resid_bf16[m][n]])   conv+BN(+identity)+ReLU */
  KVQ_EPI_QGELU_BF16 = 6   /* out_bf16[m][n] = y * sigmoid(1.702 y), y = acc + bias     CLIP QuickGELU (clip/model.py:179-181) */
} KvqEpilogue;

typedef struct {
  const uint16_t* A;     /* bf16 [M][K] */
  const uint16_t* W;     /* bf16 [N][K] */
  const float* bias;     /* [N] or NULL */
  int32_t M, N, K;       /* N % 8 == 0 (QKV: N == 96*num_heads), K % 32 == 0 */
  int32_t epilogue;      /* KvqEpilogue */
  uint16_t* out_bf16;
  float* out_f32;
  /* KVQ_EPI_QKV_BF16 */
  int32_t num_heads;     /* N == 3*32*num_heads */
  float q_scale;         /* head_dim^-0.5 */
  /* KVQ_EPI_RESID_F32: output row of GEMM row m = b*rows_out + map[m % rows_in_map]; <0 = drop */
  const int32_t* scatter_map; /* NULL = identity */
  int32_t map_rows;      /* rows per batch element in the map (windowed rows)  */
  int32_t out_rows;      /* rows per batch element in the output               */
  int32_t dtype;         /* KvqDtype of A, W and out_bf16                      */
  const uint16_t* resid_bf16; /* KVQ_EPI_RELU_BF16: optional [M][N] identity branch, same dtype */
  const float* resid_f32;     /* KVQ_EPI_RELU_BF16: optional fp32 [M][N] identity branch; with this epilogue a
                                 non-NULL out_f32 additionally receives the fp32 (un-rounded) result */
  /* Split-K (long K, few output tiles: the late convolutions of the conv nets, stage 3 of the trunk): NULL = never.  With a
   * scratch buffer the launch MAY cut K into S ranges (kvq_gemm_splitk_factor), each workgroup writing its fp32 partial tile
   * [S][M][N] there; a second launch sums the S partials IN ORDER (bit-reproducible) and applies the epilogue.  Not for QKV. */
  void* splitk_ws;
  size_t splitk_ws_bytes;     /* >= kvq_gemm_splitk_bytes(M, N, K) or the launch stays un-split */
  /* 16-bit row-major epilogues (BIAS / GELU / QGELU / RELU): the N columns land at columns col_off .. col_off+N-1 of rows of
   * ldc elements — a channel concatenation (torch.cat along C of channels-last tensors) for free.  ldc = 0: ldc = N, col_off 0. */
  int32_t ldc, col_off;
  /* Optional row gather of A (plain GEMMs only): GEMM row m reads A row (m / a_rows) * a_phys_rows + a_gather[m % a_rows] — the proj of a
   * padded window partition run over the real tokens (a_gather = token -> window row) instead of over the padding rows as well. */
  const int32_t* a_gather;
  int32_t a_rows, a_phys_rows;
} KvqGemmArgs;
/* S the launch would use (1 = no split) and the scratch bytes that S needs */
int kvq_gemm_splitk_factor(int M, int N, int K);
size_t kvq_gemm_splitk_bytes(int M, int N, int K);

int kvq_gemm_bf16(const KvqGemmArgs* host_args, void* stream);
/* Which main loop kvq_gemm_bf16 / kvq_conv_implicit take (process-wide; returns the previous mode): -1 = by shape (default; the
 * 256 x 256 x 64 eight-phase kernel of csrc/gemm256.hip when the tile grid fills the chip, else the 128 x 128 x 32 ring kernel),
 * 0 = never the wide tile, 1 = the wide tile whenever the shape is eligible (K % 64 == 0, K >= 128).  Any other value only reads.
 * Initial value: environment KVQ_GEMM8P.  Used by the parity tests to run every epilogue through both kernels. */
int kvq_gemm_tile_mode(int mode);
/* Diagnostic: while dev_buf != NULL every GEMM block b < max_blocks writes uint64 stamps
 * dev_buf[8*b + {0:start, 1:first slice landed, 2:K loop done, 3:epilogue done}] (shader clock) and
 * [4] = XCC id << 32 | HW_ID.  Pass NULL to switch it off. */
int kvq_debug_gemm_trace(void* dev_buf, int max_blocks);

/* PatchEmbed3D (swin_backbone.py:715-733) as one launch, token-per-lane MFMA (csrc/embed.hip): the strided
 * Conv3d reads its patches straight from the clip (no im2col buffer), + bias + LayerNorm(E), optionally + the
 * first block's norm1 in its window order.  Fused shape: patch (pd,4,4), in_chans*pd == 6, E in {96,128}, clip
 * dimensions multiples of the patch (anything else takes the im2col + GEMM + LayerNorm launches). */
typedef struct {
  const float* x;              /* fp32 (B, in_chans, T, H, W); NULL when frag is set                    */
  int32_t B, in_chans, T, H, W;
  int32_t pd, ph, pw, embed_dim;
  const void* pack;            /* kvq_patch_embed_pack image                                            */
  int32_t has_norm;            /* patch_embed.norm present (the image carries identity vectors if not)  */
  float* out;                  /* fp32 [B*D0*H0*W0][E]                                                  */
  const float* next_norm_w;    /* the following four: only with next_ln != NULL                         */
  const float* next_norm_b;
  const int32_t* next_dst;     /* token -> row of the first block's window order (no padding)           */
  void* next_ln;               /* 16-bit [B*next_rows][E]                                               */
  int32_t next_rows;
  float eps;
  int32_t dtype;
  const KvqFragmentSource* frag; /* host struct or NULL: read the clip through the sampler (T, H, W = the sampled clip's) */
  int32_t out_f16;             /* ABI 31: `out` receives the residual stream as fp16 [B*D0*H0*W0][E] (2 E bytes per row) instead of fp32 */
} KvqPatchEmbedArgs;
int kvq_patch_embed_supported(int in_chans, int pd, int ph, int pw, int embed_dim, int T, int H, int W);
/* 1 when the fused read applies to a (B, in_chans, T, H, W) batch sampled from src: uint8 frames, 4 x 4 patches inside the
 * mini-patches (fs_h, fs_w multiples of 4), Fh*fs_h == H, Fw*fs_w == W, source >= canvas, one clip per 32 tokens. */
int kvq_patch_embed_fragments_supported(const KvqFragmentSource* src, int B, int in_chans, int pd, int T, int H, int W);
size_t kvq_patch_embed_pack_bytes(int embed_dim, int K);
/* w: 16-bit [E][K = in_chans*pd*ph*pw] (Conv3d weight order); ln_w / ln_b may be NULL (no norm). */
int kvq_patch_embed_pack(const void* w, const float* bias, const float* ln_w, const float* ln_b, int embed_dim, int K,
                         void* pack, void* stream);
int kvq_patch_embed(const KvqPatchEmbedArgs* host_args, void* stream);

/* PatchMerging (swin_backbone.py:533-556) as one launch, token-per-lane MFMA (csrc/merge.hip), C = 96 / 128 / 192: the 4-neighbour concat
 * (x0 x1 x2 x3 = (h,w) (h+1,w) (h,w+1) (h+1,w+1), F.pad zeros for odd H / W), LayerNorm(4C) and Linear(4C -> 2C, no bias),
 * optionally + the next stage's first norm1 in its window order.  LayerNorm is folded around the GEMM:
 * W (gamma (x - mean) rstd + beta) = rstd (W diag(gamma)) (x - mean) + W beta; the pack holds W diag(gamma) (16-bit MFMA fragments)
 * and W beta (fp32).  kvq_patch_merge_pack reads the fp32 weights. */
typedef struct {
  const float* x;              /* fp32 [B*L][C] residual stream                                        */
  const int32_t* merge_map;    /* int32 [Ln][4]: token (within a clip) of each neighbour, -1 = padding  */
  int32_t B, L, Ln, C;
  const void* pack;            /* kvq_patch_merge_pack image                                            */
  float* out;                  /* fp32 [B*Ln][2C]                                                       */
  const float* next_norm_w;    /* the following four: only with next_ln != NULL                         */
  const float* next_norm_b;
  const int32_t* next_dst;     /* merged token -> row of the next block's window order (no padding)     */
  void* next_ln;               /* 16-bit [B*next_rows][2C]                                              */
  int32_t next_rows;
  float eps;
  int32_t dtype;
  int32_t x_f16, out_f16;      /* ABI 31: x / out are fp16 residual streams (rows of 2 C / 4 C bytes) instead of fp32                */
} KvqPatchMergeArgs;
int kvq_patch_merge_supported(int C);
size_t kvq_patch_merge_pack_bytes(int C);
int kvq_patch_merge_pack(const float* red_w /* fp32 [2C][4C] */, const float* norm_w, const float* norm_b, int C,
                         int dtype /*KvqDtype*/, void* pack, void* stream);
int kvq_patch_merge(const KvqPatchMergeArgs* host_args, void* stream);

/* Fused post-attention half of SwinTransformerBlock3D, one launch, token-per-lane MFMA (csrc/tail.hip):
 *   x <- x + window_reverse(roll(proj(attn)))        (swin_backbone.py:323, :472-488, :509)
 *   x <- x + fc2(GELU(fc1(norm2(x))))                (:490-491, :514; Mlp :84-87)
 *   optionally next_ln <- window_partition(roll(norm1_next(x)))   (:416-449 of the next block)
 * The hidden activations, proj output and norm2 output stay in registers.  Weights come as the image
 * kvq_block_tail_pack builds (MFMA-fragment-major panels, k order of fc1/fc2 permuted to the accumulator
 * layout, followed by the fp32 bias / norm vectors). */
typedef struct {
  const void* attn;            /* 16-bit [M][C], window order (kvq_window_attention output)              */
  float* x;                    /* fp32 [n_batch*out_rows][C] residual stream, updated in place           */
  const int32_t* scatter_map;  /* window row -> token within the batch element, <0 = padding; NULL = id. */
  int32_t map_rows, out_rows;  /* rows per batch element in the map / in x                               */
  int32_t M, C, hidden;        /* M = n_batch*map_rows window rows; C in {96,128,192} (hidden % 64 == 0), 256, 384, 512 or 768 (hidden = 4 C) */
  const void* pack;            /* kvq_block_tail_pack image                                              */
  const float* next_norm_w;    /* the following four: only with next_ln != NULL                          */
  const float* next_norm_b;
  const int32_t* next_dst;     /* token -> row of the next block's window order (a bijection: no padding) */
  void* next_ln;               /* 16-bit [n_batch*next_rows][C]                                          */
  int32_t next_rows;
  float eps;                   /* 1e-5                                                                   */
  int32_t dtype;               /* KvqDtype of attn, the packed weights and next_ln                        */
  const int32_t* attn_gather;  /* optional, [out_rows]: token -> window row of THIS block's partition (the inverse of scatter_map over
                                  the real tokens).  When set the launch walks the n_batch*out_rows TOKENS and fetches each one's
                                  attention row through it, instead of walking the M window rows: padded geometries (Swin-B at
                                  256x256: 1.2x .. 3x the rows) then do no work on padding rows.  scatter_map is not read. */
  /* Instead of next_ln (leave it NULL; next_norm_w / _b / next_dst / next_rows as above): the NEXT block's q | k | v (swin_backbone.py:
   * 252-260), head-major [3][num_heads][n_batch*next_rows][32] in its window order, q scaled by q_scale — what the qkv GEMM's
   * KVQ_EPI_QKV_BF16 epilogue writes.  C with kvq_block_tail_qkv_pack_bytes(C, hidden) > 0 only (128 / 192 / 256 / 384 / 512 / 768). */
  const void* next_qkv_pack;   /* kvq_block_tail_qkv_pack image of the next block's qkv weight               */
  const float* next_qkv_b;     /* [3C]                                                                       */
  void* qkv_out;               /* 16-bit; non-NULL selects this form                                         */
  float q_scale;
  int32_t num_heads;
  int32_t x_f16;               /* ABI 31: x is the residual stream kept in fp16: rows of 2 C bytes, read and written in place            */
} KvqBlockTailArgs;
/* Padded window partitions (Swin-B at 256x256, KSVQE at 288x288): the q|k|v of a PADDING row is qkv(0) = bias (the reference pads after
 * norm1, swin_backbone.py:416-449) and takes part in the softmax of its window as a key.  Instead of multiplying zero rows, the qkv
 * GEMM runs over the real tokens — KVQ_EPI_QKV_BF16 with scatter_map = token -> window row, map_rows = tokens and out_rows = window rows
 * per batch element — and this launch writes bias (q scaled) into the n_pad padding rows pad_rows[] of every batch element. */
int kvq_qkv_fill_pad(void* qkv, const float* qkv_bias, const int32_t* pad_rows, int n_pad, int n_batch, int rows_per_batch,
                     int num_heads, float q_scale, int dtype, void* stream);
int kvq_block_tail_supported(int C, int hidden);                 /* 1 / 0 */
size_t kvq_block_tail_pack_bytes(int C, int hidden);             /* 0 when unsupported */
/* proj_w [C][C], fc1_w [hidden][C], fc2_w [C][hidden]: 16-bit nn.Linear layouts; the rest fp32. */
int kvq_block_tail_pack(const void* proj_w, const float* proj_b, const float* norm2_w, const float* norm2_b,
                        const void* fc1_w, const float* fc1_b, const void* fc2_w, const float* fc2_b, int C, int hidden,
                        void* pack, void* stream);
/* qkv_w [3C][C] 16-bit (rows q | k | v, head-major): the image the fused tail of the PREVIOUS block streams to emit q | k | v. */
size_t kvq_block_tail_qkv_pack_bytes(int C, int hidden);         /* 0: this width's tail cannot emit q | k | v */
int kvq_block_tail_qkv_pack(const void* qkv_w, int C, int hidden, void* pack, void* stream);
int kvq_block_tail(const KvqBlockTailArgs* host_args, void* stream);

/* WindowAttention3D core (swin_backbone.py:261-322) for head_dim 32: S = q k^T + bias, where
 * bias = rpb*g + fpb*(1-g) (gated, :299-302) or rpb, + shift mask (0/-100, :583), softmax, @v.
 *   qkv      bf16 head-major [3][nH][BW*N][32] (q pre-scaled)   — output of KVQ_EPI_QKV_BF16
 *   tok      int32 [nW*N][2]: {bias code c = d*(2Wh-1)(2Ww-1)+h*(2Ww-1)+w of the token's
 *            coordinate in the configured window raster, desc = fh | fw<<8 | region<<16}
 *   rpb/fpb  fp32 [table_len][nH]; fpb NULL = no fragment gate
 *   bias_pack optional fp32 [nH][TLe][2] = {fpb|rpb, rpb-fpb|0}, TLe = table_len rounded up to even; else NULL
 *   center   (Wd-1)*(2Wh-1)*(2Ww-1) + (Wh-1)*(2Ww-1) + (Ww-1)
 *   out      bf16 [BW*N][nH*32]  (== (attn@v).transpose(1,2).reshape(B_,N,C), :322)        */
int kvq_window_attention(const uint16_t* qkv, const int32_t* tok, const float* rpb, const float* fpb,
                         const float* bias_pack, int table_len, int center, int BW, int nW, int N, int num_heads, int use_mask,
                         int dtype, uint16_t* out, void* stream);

/* The same attention with the bias PRE-BUILT per (window type, head) (csrc/attn32.hip; the trunk's default): bias[w][h][i][j] = what
 * kvq_window_attention rebuilds per score (table gather, fragment gate, -100 shift mask), stored as an fp16 image in the kernel's MFMA
 * accumulator layout — [n_types][nH][ceil(N/32)][13][2][64 lanes][8] + one 2 KB pad block, 16-byte aligned: register r = 8 half + e of lane
 * (q = lane & 31, hi = lane >> 5) of the 32 x 32 score block (qb, kb) is query 32 qb + q against key 32 kb + (r & 3) + 8 (r >> 2) + 4 hi;
 * keys >= N hold -60000.  2 B per score of HBM / L2 traffic, shared by all clips of a step and by the windows of one TYPE: n_types <= nW
 * divides nW and window w uses bias w % n_types (un-shifted windows that differ only in depth index share one; pass the descriptors of
 * the first n_types windows to the builder).  Stored is bias - max_key bias of the query's row (softmax is invariant to a per-row shift):
 * the entries that carry the probability mass sit next to 0, where fp16 resolves them to <= 2^-11.  The builder also reports max |bias|
 * (un-masked entries) through max_abs (device float, zero it first; NULL = skip); the host mirror keeps the exact per-score path above a
 * (generous) cap.  tok, rpb, fpb, table_len, center, use_mask: as kvq_window_attention (fpb NULL = no gate).
 *
 * The kernel: S^T = K Q^T on v_mfma_f32_32x32x16, the bias tile widened / scaled by log2(e) / shifted by the row's running maximum in one
 * v_fma_mix_f32 per score as the MFMA's C operand (flash-attention running maximum with a deferred rescale, threshold 2^8), row sums
 * by v_dot2c on the packed probabilities, O^T = V^T P^T through the hardware transpose read; one workgroup of four waves per (window,
 * head, clip[, q-part]), K | V staged by LDS-DMA, 32-query blocks from an LDS ticket.
 *   - q must arrive scaled by head_dim^-0.5 * log2(e) (scores are kept in log2 units; the image stays in natural units);
 *   - tile_skip[nW] (or NULL): bit t of tile_skip[w] set = rows 16t .. 16t+15 of window w are padding rows only (padded partitions); a
 *     32-row q-block is passed over when both its 16-row tiles are — its output rows are never read (the consumers walk the tokens);
 *   - dsplit_from >= 0 (shifted blocks of the (8,7,7) window, N = 392): windows w >= dsplit_from of every clip are DEPTH-SPLIT — the
 *     cyclic shift put depth positions Dp-4.. and the wrapped 0..3 into one window and the shift mask (swin_backbone.py:563-579)
 *     separates the two halves of 196 tokens — so a q-block of one half passes over the 32-key blocks of the other half, whose scores
 *     are the image's -100 and leave the exponential as zeros (46 % fewer score blocks in those windows).  First-half rows come out bit
 *     for bit as with dsplit_from = -1; second-half rows start their running maximum at another block and may differ in the last bit of
 *     the 16-bit output.  The caller vouches for the geometry (the plan derives it from the window layout). */
size_t kvq_attn_bias32_bytes(int n_types, int N, int num_heads);      /* 0 for N > 400 */
int kvq_attn_bias32_build(const int32_t* tok, const float* rpb, const float* fpb, int table_len, int center,
                          int n_types, int N, int num_heads, int use_mask, void* out, float* max_abs, void* stream);
typedef struct {
  const uint16_t* qkv;        /* [3][nH][BW*N][32], q scaled by head_dim^-0.5 * log2(e) */
  const void* bias_dense;     /* kvq_attn_bias32_build image of n_types window types */
  int32_t n_types, BW, nW, N, num_heads, dtype;
  uint16_t* out;              /* [BW*N][nH*32] */
  const uint32_t* tile_skip;  /* optional, see above */
  int32_t dsplit_from;        /* -1 = no depth-split windows */
  /* Fused qkv projection (x_ln != NULL; C = 32 num_heads = 96; always one workgroup per (window, head)): the workgroup of a (window,
   * head) computes its q | k | v from the window's norm1 rows — x_ln 16-bit [BW*N][C] in window order (no padding rows: un-padded
   * partitions only), w_qkv 16-bit [3C][C], b_qkv fp32 [3C] (swin_backbone.py:252-260) — instead of reading them: k and v go
   * straight into the LDS images, q (scaled by q_scale = head_dim^-0.5 * log2(e)) into the q third of `qkv` ([num_heads][BW*N][32];
   * the k / v thirds are not touched and need not exist).  Replaces kvq_gemm_bf16(KVQ_EPI_QKV_BF16) where that launch is HBM-bound. */
  const uint16_t* x_ln;
  const uint16_t* w_qkv;
  const float* b_qkv;
  float q_scale;
  /* Padded partitions (pad_mask != NULL, with b_qkv): uint32 [nW][13], bit (r & 31) of word r >> 5 of window w set = window row r is a
   * PADDING row.  The reference computes qkv(0) = bias for such rows (F.pad after norm1, swin_backbone.py:416-449) and they take part in
   * their window's softmax as keys: the kernel writes k | v = the 16-bit rounding of b_qkv's k / v thirds into those rows of its LDS
   * images itself (what kvq_qkv_fill_pad would have put into `qkv`: those rows of `qkv` are then never read as keys) and takes their q as
   * zero (their output rows are never read).  Saves the fill launch and its HBM round trip (Swin-B at 256 x 256: 79 MB per block). */
  const uint32_t* pad_mask;
} KvqAttnDenseArgs;
int kvq_window_attention32(const KvqAttnDenseArgs* host_args, void* stream);

/* im2col of PatchEmbed3D's stride==kernel Conv3d (swin_backbone.py:715-726): zero pads the tail
 * of each axis, emits bf16 rows [B*D*H'*W'][in*pd*ph*pw] in (c,kd,kh,kw) order. */
int kvq_patch_im2col(const float* x, int B, int Cin, int T, int H, int W, int pd, int ph, int pw,
                     int dtype, uint16_t* out, void* stream);

/* VQAHead.forward in eval mode (models/head.py:60-68): fp32 throughout.
 *   feat  fp32 with explicit element strides (so both (B,C,D,H,W) and channels-last work)
 *   w1t   fc_hid.weight TRANSPOSED: fp32 [C][hidden] (lane j of a wave owns hidden unit j) — the VALU kernel's layout
 *   w1    fc_hid.weight as it is: fp32 [hidden][C], or NULL.  With hidden == 64, channels-last features (stride_c == 1),
 *         C % 64 == 0 and 16-byte aligned rows the head runs on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: fp32 products
 *         and sums, a different summation order) and reads w1; w1t may then be NULL.
 *   score fp32 [B] = mean_tokens( w2 . gelu(W1 f + b1) + b2 ).  scratch: fp32 [B*L]. */
int kvq_vqa_head(const float* feat, int B, int L, int C, int64_t stride_b, int64_t stride_l,
                 int64_t stride_c, const float* w1t, const float* w1, const float* b1, int hidden, const float* w2,
                 const float* b2, float* scratch, float* score, void* stream);

/* VQAHead.forward's other branches (models/head.py:60-68; no reference config sets them): pre_pool != 0 averages the token grid
 * first (AdaptiveAvgPool3d((1,1,1)), head.py:61-62); num_class > 1 applies nn.Softmax() — implicit dim 1 = the classes — to
 * fc_last's outputs per token (head.py:66-67) before the mean over the tokens (head.py:68).
 *   w1t fp32 [C][hidden], w2 fp32 [num_class][hidden], b2 fp32 [num_class]; score fp32 [B][num_class];
 *   scratch fp32 [B*L*num_class] (pre_pool: [B*C + B*num_class]).  num_class == 1: no softmax, as the reference. */
int kvq_vqa_head_classes(const float* feat, int B, int L, int C, int64_t stride_b, int64_t stride_l, int64_t stride_c,
                         const float* w1t, const float* b1, int hidden, const float* w2, const float* b2, int num_class,
                         int pre_pool, float* scratch, float* score, void* stream);

/* simpleVQAHead.forward (models/head.py:28-31): Linear(Cin->hidden) -> Linear(hidden->1), mean over
 * frames.  feat fp32 [B][T][Cin]; score fp32 [B]; scratch fp32 [B*T]. */
int kvq_simple_vqa_head(const float* feat, int B, int T, int Cin, const float* w1, const float* b1,
                        int hidden, const float* w2, const float* b2, float* scratch, float* score,
                        void* stream);

/* get_spatial_fragments + (v-mean)/std (datasets/fusion_datasets.py:22-121, 1017-1020) with the
 * drawn offsets as inputs (the reference draws them inside with torch.randint, :87-98).
 *   video  uint8 or fp32 (C,T,H,W) on device (src_is_u8 selects)
 *   hoff/woff int32 [Fh][Fw][T/aligned] absolute patch origins = grid + random offset
 *   out    fp32 (C,T,Fh*fs_h,Fw*fs_w);  mean/std: host fp32 [C] (std==NULL -> no normalisation) */
int kvq_fragment_gather(const void* video, int src_is_u8, int C, int T, int H, int W, const int32_t* hoff,
                        const int32_t* woff, int Fh, int Fw, int fs_h, int fs_w, int aligned,
                        const float* host_mean, const float* host_std, float* out, void* stream);
/* The same for every clip of a KvqFragmentSource in ONE launch (uint8 or fp32 frames, clips may be frame runs of a longer
 * video: chan_stride): out fp32 (n_clips, C, T, Fh*fs_h, Fw*fs_w).  What FragmentSource.materialise() runs. */
int kvq_fragment_gather_batch(const KvqFragmentSource* src, int C, int T, float* out, void* stream);

/* torchvision Resize on a tensor (= bilinear, align_corners=False, no antialias; get_resize_function,
 * fusion_datasets.py:229-241) + crop + (v-mean)/std: get_resized_video (:244-252), get_resizecrop_video
 * (:299-316).  video u8|fp32 (C,T,H,W) -> resized to (rh,rw) -> crop [cy:cy+oh, cx:cx+ow] -> out fp32
 * (C,T,oh,ow).  round_u8: round+clamp to 0..255 before normalising (what torchvision does to integer tensors). */
int kvq_resize_bilinear(const void* video, int src_is_u8, int C, int T, int H, int W, int rh, int rw, int cy,
                        int cx, int oh, int ow, int round_u8, const float* host_mean, const float* host_std,
                        float* out, void* stream);

/* get_spatial_fragments' fallback for sources smaller than the canvas (fusion_datasets.py:43-50): F.interpolate(video / 255.0,
 * scale_factor = s, mode = "bilinear") * 255.0 cast back to the frame type (uint8: truncation), with ATen's CPU arithmetic to the
 * bit.  video u8|fp32 (C,T,H,W) -> out of the same type (C,T,floor(H*s),floor(W*s)) — kvq_upsample_frames_out_dims gives the size. */
int kvq_upsample_frames_out_dims(int H, int W, double scale_factor, int32_t out2[2]);
int kvq_upsample_frames(const void* video, int src_is_u8, int C, int T, int H, int W, double scale_factor, void* out,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution front-ends (2D ResNet-50 of SimpleVQA, simpleVQA_model.py:220-264; SlowFast-R50 3D convs,
 * SlowFast_features.py:137-165).  Activations are channels-last 16-bit (B,D,H,W,C); conv = im2col -> GEMM
 * (BatchNorm folded on the host, ReLU / identity add = KVQ_EPI_RELU_BF16); 1x1x1 stride-1 convs need no im2col.
 * ------------------------------------------------------------------------------------------- */

/* ---------------------------------------------------------------------------------------------
 * KSVQE "CLIP_tool": CLIP_extractor_addadapter_cls.forward (models/backbones/CLIP_backbone.py:156-201) over the vendored
 * CLIP vision transformer (models/backbones/clip/model.py:184-294).  Its Linear layers, LayerNorms and the 16x16 patch
 * embedding run on kvq_gemm_bf16 / kvq_layernorm_rows / kvq_patch_im2col; these are the remaining pieces.
 * ------------------------------------------------------------------------------------------- */
/* Token assembly + ln_pre (:163-171): out[b][0] = LN(cls + pos[0]), out[b][1+i] = LN(tok[b*G+i] + pos[1+i]); fp32;
 * pos is the positional embedding already resized to the G-token grid (resize_pos_embed2d :35-70 is host-side). */
int kvq_vit_embed_ln(const float* tok, const float* cls, const float* pos, const float* ln_w, const float* ln_b, int B, int G,
                     int D, float eps, float* out, void* stream);
/* nn.MultiheadAttention(x, x, x) core (clip/model.py:199-201), no mask: qkv 16-bit [B*L][3*D] = the in_proj output
 * (rows [q | k | v], head h = columns h*64.. of each third), q scaled by head_dim^-0.5 here; out 16-bit [B*L][D]
 * (heads concatenated) = the out_proj input.  L <= 320, head_dim == 64. */
int kvq_mha_small(const uint16_t* qkv, int B, int L, int heads, int head_dim, int dtype, uint16_t* out, void* stream);
/* The general form: q [B*Lq] / k, v [B*Lk] rows with their own element strides (multiples of 8; k, v 16-byte aligned) and an
 * explicit logit scale — KSVQE's crossattention1 (KSVQE_model.py:1553-1586: trunk tokens attend CLIP / distortion tokens,
 * scale dim^-0.5, no output projection) and its temporal Attention (:1508-1551: 16 frames per position, scale head_dim^-0.5). */
int kvq_mha_cross(const uint16_t* q, long ldq, const uint16_t* k, long ldk, const uint16_t* v, long ldv, int B, int Lq, int Lk,
                  int heads, int head_dim, float scale, int dtype, uint16_t* out, void* stream);
/* CLS adapter plumbing (CLIP_backbone.py:183-191): x (B, L, D) fp32; gather x[:, 0] as the 16-bit GEMM operand [B][D];
 * x[:, 0] = ratio * a + (1 - ratio) * x[:, 0] with a = the adapter's 16-bit output [B][D]. */
int kvq_cls_gather(const float* x, int B, int L, int D, int dtype, uint16_t* out, void* stream);
int kvq_cls_mix(float* x, const uint16_t* a, int B, int L, int D, float ratio, int dtype, void* stream);
/* torch.cosine_similarity(x[:, :1], x[:, 1:], dim=-1) (:199): out fp32 [B][L-1]. */
int kvq_cosine_cls(const float* x, int B, int L, int D, float* out, void* stream);

/* fp32 <-> 16-bit element conversion of an activation (n elements; to_half: src fp32 16-B aligned). */
int kvq_convert(const void* src, void* dst, long n, int to_half, int dtype, void* stream);
/* Semantic_Transformation2.forward (KSVQE_model.py:829-835) on channels-last token rows: per row m
 * gama = sigmoid(<w_gama, x_m> + b_gama), beta = <w_beta, x_m> + b_beta (the two 1x1 convs C -> 1), out_m = gama*input_m + beta. */
int kvq_sem_modulate(const float* x, const float* input, const float* w_gama, float b_gama, const float* w_beta, float b_beta, int M,
                     int C, float* out, void* stream);
/* Dist_Transformation3.forward (:952-960), last step: out[b][r][c] = sigmoid(gamma_logit[b][c]) * input[b][r][c] + beta[b][c];
 * gamma_logit / beta = get_gamma(std) / get_beta(mean), 16-bit [B][C] (GEMM outputs). */
int kvq_dist_modulate(const float* input, const uint16_t* gamma_logit, const uint16_t* beta, int B, int rows, int C, int dtype,
                      float* out, void* stream);

/* KSVQE quality-aware region selection, eval path of RegionNet_CLIP.forward (models/backbones/patchnet.py:461-550): the CLIP
 * CLS-to-patch map of a key frame (BK maps of gs x gs, fp32) is nearest-upsampled to the anchor grid gh x gw, averaged over
 * every kh x kw window (F.unfold, stride 1) and the best window's index (row-major over (gh-kh+1) x (gw-kw+1), first
 * maximum) is returned; kvq_crop_regions then cuts that window (kh*anchor x kw*anchor pixels at (ry*anchor, rx*anchor))
 * out of every frame: x (B, C, T, H, W) fp32, region int32 [B*T] -> out (B, C, T, kh*anchor, kw*anchor). */
int kvq_qrs_top_region(const float* score, int BK, int gs, int gh, int gw, int kh, int kw, int32_t* idx, void* stream);
int kvq_crop_regions(const float* x, const int32_t* region, int B, int C, int T, int H, int W, int anchor, int kh, int kw,
                     float* out, void* stream);

/* F.normalize(x, dim=1) of fp32 rows [M][D] -> 16-bit [M][D] (CONTRIQUE_model.forward, KSVQE_model.py:1654-1656). */
int kvq_l2_normalize_rows(const float* x, int M, int D, int dtype, uint16_t* out, void* stream);

/* out = a * x + b * y over n fp32 elements (out may alias x or y): the fixed blends of KSVQE.forward (KSVQE_model.py:1426, :1482). */
int kvq_axpby(const float* x, const float* y, float a, float b, float* out, long n, void* stream);

/* Implicit-GEMM convolution (nn.Conv2d / nn.Conv3d + folded BatchNorm [+ identity] [+ ReLU], the Bottleneck convs of
 * simpleVQA_model.py:85-126 and the SlowFast res blocks): the GEMM's A tiles are fetched straight from the channels-LAST
 * 16-bit activation x (B, D, H, W, C), C % 8 == 0 — no patch matrix.  W [N][Kpad], columns ordered (kd,kh,kw,c) like
 * kvq_im2col_nd and zero padded to Kpad (a multiple of 32).  taps: device int32 [Kpad/8][4], one row per 8-channel
 * chunk of K: {kd, kh, kw, ((kd*H + kh)*W + kw)*C + c0}, last entry -1 for chunks of the K padding (depends on H, W, C).
 * The table defines K: taps that only ever read the zero border (3x3 / pad 1 on a 1x1 map: eight of nine) may be left out
 * of it together with their columns of W — Kpad is 8 x the table's rows, not necessarily >= kd*kh*kw*C.
 * taps may be NULL when C % 32 == 0 and Kpad >= kd*kh*kw*C (the full tap set): the kernel then walks (kd, kh, kw, c) itself with
 * wave-uniform counters — no table loads between the slice transfers (res4's 3x1x1 over 1024 channels: 70 -> 48 us per launch),
 * bit-identical results.
 * Output rows = output pixels (b, do, ho, wo), i.e. channels-last again.  epilogue: KVQ_EPI_RELU_BF16 | KVQ_EPI_BIAS_BF16 |
 * KVQ_EPI_STORE_F32 (out_f32 [M][N] = acc + bias: the projection shortcuts, kept in fp32). */
typedef struct {
  const uint16_t* x;
  const uint16_t* W;
  const float* bias;
  const int32_t* taps;
  int32_t dims5[5];       /* B, C, D, H, W */
  int32_t kernel3[3], stride3[3], pad3[3];
  int32_t Kpad, N;
  int32_t epilogue, dtype;
  uint16_t* out_bf16;     /* [B*Do*Ho*Wo][N] */
  float* out_f32;         /* optional fp32 copy (RELU epilogue) */
  const uint16_t* resid_bf16;
  const float* resid_f32;
  void* splitk_ws;        /* optional split-K scratch, as in KvqGemmArgs (M = B*Do*Ho*Wo, K = Kpad) */
  size_t splitk_ws_bytes;
  int32_t ldc, col_off;   /* as in KvqGemmArgs: out_bf16 rows of ldc channels, this conv's N at channel col_off (0 = N / 0) */
} KvqConvArgs;
int kvq_conv_implicit(const KvqConvArgs* host_args, void* stream);

/* Gather conv patches: x with explicit ELEMENT strides5 = {b,c,d,h,w} over dims5 = {B,C,D,H,W} (fp32 when
 * src_f32, else the 16-bit dtype) -> out [B*Do*Ho*Wo][Kpad], columns ordered (kd,kh,kw,c) and zero padded
 * from K = kd*kh*kw*C to Kpad (a multiple of 32).  Zero padding of the borders as nn.ConvNd(padding=pad3). */
int kvq_im2col_nd(const void* x, int src_f32, int dtype, const int64_t strides5[5], const int32_t dims5[5],
                  const int32_t kernel3[3], const int32_t stride3[3], const int32_t pad3[3], int Kpad,
                  uint16_t* out, void* stream);
/* The 3-channel network input as an implicit-GEMM operand (stem conv 7x7/2 of ResNet-50, simpleVQA_model.py:220-223 /
 * torchvision resnet50 inside CONTRIQUE_model, KSVQE_model.py:1630): fp32 frames addressed through ELEMENT strides5 =
 * {b,t,c,h,w} over dims5 = {B,T,C,H,W} (C <= 8) -> 16-bit channels-last (B*T, H, W, 8), channels >= C zero. */
int kvq_pack_channels_last8(const float* x, const int32_t dims5[5], const int64_t strides5[5], int dtype, uint16_t* out,
                            void* stream);
/* Direct Conv3d for the few-output-channel stems (SlowFast fast pathway: 3 -> 8, k 5x7x7, SlowFast_features.py:140):
 * x fp32 (B,C,D,H,W) contiguous, w fp32 [K][cout] with K ordered (kd,kh,kw,c), bias fp32 [cout] (BatchNorm folded),
 * cout in {8,16}; out 16-bit channels-last (B,Do,Ho,Wo,cout).  fp32 arithmetic; no patch matrix is materialised. */
int kvq_conv_stem_direct(const float* x, const int32_t dims5[5], const float* w, const float* bias, int cout,
                         const int32_t kernel3[3], const int32_t stride3[3], const int32_t pad3[3], int relu, int dtype,
                         uint16_t* out, void* stream);
/* The same stem on the matrix cores (SlowFast fast pathway, SlowFast_features.py:140: Conv3d(3, 8, (5,7,7), stride (1,2,2),
 * padding (2,3,3)) + BatchNorm + ReLU).  kvq_pack_clip_cl4: x fp32 (B,C<=4,T,H,W) contiguous -> 16-bit channels-last with four
 * channels and `border` zero pixels left and right, (B, T, H, W + 2*border, 4).  kvq_conv_stem_mfma: x4 packed with border 4,
 * dims4 = {B, T, H, W}; wpack 16-bit [kd*kh][16][32], entry [a*kh + r][o][tap*4 + c] = w[o][c][a][r][tap] (BatchNorm folded;
 * rows o >= 8, tap 7 and c >= C zero); bias8 fp32 [8]; kernel width 7 / stride 2 / pad 3 along W; out 16-bit (B,Do,Ho,Wo,8).
 * 16-bit operands, fp32 accumulate; no patch matrix. */
int kvq_pack_clip_cl4(const float* x, const int32_t dims5[5], int border, int dtype, uint16_t* out, void* stream);
int kvq_conv_stem_mfma(const uint16_t* x4, const int32_t dims4[4], const uint16_t* wpack, const float* bias8,
                       const int32_t kernel3[3], const int32_t stride3[3], const int32_t pad3[3], int relu, int dtype,
                       uint16_t* out, void* stream);
/* The whole fast-pathway stem in ONE launch (SlowFast_features.py:137-165 block 0, fast pathway): Conv3d(3, 8, (kd,7,7), stride
 * (1,2,2), padding (kd/2,3,3)) + folded BatchNorm [+ ReLU] + MaxPool3d((1,3,3), stride (1,2,2), padding (0,1,1)) read straight
 * from the fp32 clip x (B,3,T,H,W), W % 4 == 0, W <= 256; wpack / bias8 as for kvq_conv_stem_mfma; out 16-bit channels-last
 * (B, T, Hp, Wp, 8) with Hp = (Ho - 1) / 2 + 1 over the stem's Ho = (H - 1) / 2 + 1.  Same arithmetic as pack + stem + pool
 * (16-bit operands, fp32 accumulate, 16-bit stem values before the max); neither the packed clip nor the stem map is stored. */
int kvq_conv_stem_pool(const float* x, const int32_t dims5[5], const uint16_t* wpack, const float* bias8, int kd, int relu,
                       int dtype, uint16_t* out, void* stream);
/* The slow-pathway stem in ONE launch (SlowFast_features.py:112-165: pack_pathway_output's frame selection, then block 0 of the slow
 * pathway): frames t_index[0 .. n_frames) (device int32; NULL = frames 0 .. n_frames-1) of the fp32 clip x (B,3,T,H,W) ->
 * Conv3d(3, 64, (1,7,7), stride (1,2,2), padding (0,3,3)) + folded BatchNorm [+ ReLU] + MaxPool3d((1,3,3), (1,2,2), (0,1,1)).
 * wimg 16-bit [7][64][32], entry [kh][o][kw*4 + c] = w[o][c][0][kh][kw] (kw 7 and c 3 zero); bias64 fp32 [64]; W % 4 == 0, W <= 224.
 * t_index lives on the device: its entries must be in [0, T) (the library cannot check them; kvq_convnet_create does for its plans).
 * out 16-bit channels-last (B, n_frames, Hp, Wp, out_C): channels out_coff .. out_coff + 63 are written (out_C, out_coff % 8 == 0). */
int kvq_conv_stem64_pool(const float* x, const int32_t dims5[5], const int32_t* t_index, int n_frames, const uint16_t* wimg,
                         const float* bias64, int relu, int dtype, uint16_t* out, int out_C, int out_coff, void* stream);
/* ---- Whole-network entry for the convolutional branches (csrc/convnet.hip) ------------------------------------------------
 * The reference sequences these networks layer by layer from Python (SlowFast_features.py:137-165: blocks 0-4 of
 * pytorchvideo's slowfast_r50 + the head pools; simpleVQA_model.py:220-264: ResNet-50 + avg / std pooling).  Here the layer
 * table is handed over ONCE (kvq_convnet_create: shapes checked, tap tables and the workspace layout built) and a forward is one
 * call that enqueues every launch on the caller's stream — what kvq_swin3d_forward is for the trunk.  Slots 0 .. n_inputs-1 of
 * the tensor table are the caller's input pointers (fp32 planar clips), the others live in the caller's workspace. */
typedef enum {
  KVQ_NET_CONV = 0,       /* Conv3d/2d + folded BatchNorm [+ identity src2] [+ ReLU]; 1x1x1 stride 1 over C % 32 == 0 runs as a plain GEMM */
  KVQ_NET_POOL = 1,       /* max / average pool */
  KVQ_NET_STEM8 = 2,      /* <= 8-channel fp32 planar input -> packed 8-channel rows -> implicit GEMM with a 1 x kh x kw kernel */
  KVQ_NET_STEM_MFMA = 3,  /* <= 4-channel fp32 planar input, kernel kd x kh x 7, W stride 2, pad 3, 8 outputs (SlowFast fast stem) */
  KVQ_NET_MEAN_STD = 4,   /* mean (and unbiased std) over the positions of every row -> fp32 output `dst` of the caller */
  KVQ_NET_SELECT_T = 5,   /* frames t_index[k] of an fp32 planar clip (pathway packing, SlowFast_features.py:112-135) */
  KVQ_NET_STEM_POOL = 7,  /* STEM_MFMA with kernel kd x 7 x 7, stride (1,2,2), followed by the (1,3,3) / (1,2,2) / (0,1,1) max-pool, in one
                             launch (kvq_conv_stem_pool): dst is the POOLED map */
  KVQ_NET_STEM64_POOL = 8, /* frame selection (t_index / n_index) + 3 -> 64 stem (1 x 7 x 7, stride (1,2,2)) + the (1,3,3) max-pool in one launch
                             (kvq_conv_stem64_pool): src the fp32 clip, dst the POOLED map (channels dst_coff .. dst_coff + 63), w its weight image */
  KVQ_NET_BOTTLENECK_S = 9, /* one identity residual block of the SLOW pathway (conv_a 1x1x1) in one launch (kvq_slow_bottleneck): w = the
                             packed image, kpad = inner channels, cout = output channels (written at channel 0 of dst rows) */
  KVQ_NET_BOTTLENECK = 6  /* one residual block of SlowFast's fast pathway in ONE launch (kvq_fast_bottleneck): w = the packed image,
                             kpad = inner channels, cout = output channels, n_index = 1 when the block has a projection shortcut,
                             stride3[1] = stride3[2] = its spatial stride */
} KvqNetOpKind;
typedef enum {
  KVQ_NET_T_ACT16 = 0,       /* 16-bit channels-last (B,D,H,W,C) */
  KVQ_NET_T_F32_PLANAR = 1,  /* fp32 (B,C,D,H,W): network inputs */
  KVQ_NET_T_ACT32 = 2        /* fp32 channels-last: an un-rounded residual stream (identity branch / fp32 copy of a conv output) */
} KvqNetTensorKind;
typedef struct {
  int32_t B, D, H, W, C;
  int32_t kind;           /* KvqNetTensorKind */
} KvqNetTensor;
typedef struct {
  int32_t kind;           /* KvqNetOpKind */
  int32_t src;            /* slot read */
  int32_t src2;           /* CONV: slot of the identity branch ([M][cout], 16-bit or ACT32; ReLU only) or -1 */
  int32_t dst;            /* slot written (an ACT32 dst: conv + bias stored in fp32, no ReLU — a projection shortcut kept
                             un-rounded); MEAN_STD: index into the caller's output pointers */
  int32_t dst32;          /* CONV with ReLU: ACT32 slot that additionally receives the un-rounded result, or -1 */
  int32_t kernel3[3], stride3[3], pad3[3];
  int32_t cout;           /* CONV / STEM*: output channels */
  int32_t kpad;           /* CONV / STEM8: columns of w, (kd,kh,kw,c)-ordered and zero padded to a multiple of 32 */
  int32_t relu;
  int32_t is_max;         /* POOL: max (1) or average (0) */
  int32_t dst_coff;       /* CONV / POOL: first channel of this op inside a wider dst (torch.cat along C for free) */
  int32_t per_frame;      /* MEAN_STD: rows = B*D frames pooled over H*W (1) or B clips pooled over D*H*W (0) */
  int32_t mean_off, std_off;   /* MEAN_STD: float offsets inside an output row; std_off < 0: mean only */
  int64_t out_stride;     /* MEAN_STD: floats per output row */
  const void* w;          /* device: CONV / STEM8 16-bit [cout][kpad]; STEM_MFMA the kvq_conv_stem_mfma weight image */
  const float* bias;      /* device fp32 [cout] (BatchNorm folded) */
  const int32_t* t_index; /* SELECT_T: HOST frame indices (copied) */
  int32_t n_index;
  int32_t lane;           /* 0: the caller's stream; 1: the plan's own second stream — two independent pathways (SlowFast's slow and
                             fast) then run side by side; ops of different lanes that touch the same workspace bytes are ordered by
                             events worked out in kvq_convnet_create, the second stream is forked off / joined back into the
                             caller's stream inside every forward (hipGraph-capturable).  One forward of a plan at a time. */
} KvqNetOp;
typedef struct KvqConvNet KvqConvNet;
int kvq_convnet_create(const KvqNetOp* ops, int n_ops, const KvqNetTensor* tensors, int n_tensors, int n_inputs, int n_outputs,
                       int dtype, KvqConvNet** out);
void kvq_convnet_destroy(KvqConvNet* net);
/* Split-K on / off for every launch of the plan (default on).  Whether a convolution splits depends on its tile count, i.e. on the
 * batch: with it on, a clip's features differ in the last bits with how many clips share its forward.  The feature extractor
 * (SlowFast_features.py writes per-clip .npy files the reference computes at batch 1) switches it off. */
int kvq_convnet_splitk(KvqConvNet* net, int enable);
size_t kvq_convnet_workspace_bytes(const KvqConvNet* net);
/* inputs[n_inputs]: device pointers of the input slots; outputs[n_outputs]: fp32 device buffers of the MEAN_STD ops */
int kvq_convnet_forward(const KvqConvNet* net, const void* const* inputs, float* const* outputs, void* workspace,
                        size_t workspace_bytes, void* stream);
/* measurement only: with enable != 0 the following forwards bracket every op with HIP events on the forward's stream;
 * kvq_convnet_profile_read waits for the last profiled forward and returns ms[i] = milliseconds of op i (n_ops of them). */
int kvq_convnet_profile(KvqConvNet* net, int enable);
int kvq_convnet_profile_read(const KvqConvNet* net, float* ms, int capacity, int* n_ops);

/* SlowFast fast-pathway residual block, fused (csrc/bottleneck.hip): conv_a 3x1x1 (pad 1,0,0) + BN + ReLU -> conv_b 1x3x3 (pad 0,1,1,
 * stride 1) + BN + ReLU -> conv_c 1x1x1 + BN -> + shortcut (identity when projection == 0, else the block's 1x1x1 conv + BN) -> ReLU,
 * as pytorchvideo's ResBlock runs them inside SlowFast_features.py:137-165.  x / out: 16-bit channels-last (B,T,H,W,cin) /
 * (B,T,H,W,cout), dims4 = {B,T,H,W}.  pack (kvq_fast_bottleneck_pack_bytes bytes, 16-byte aligned) holds the BatchNorm-folded 16-bit
 * weights as MFMA A fragments — fragment f, lane (m = lane & 31, h = lane >> 5), element e = W[row0 + m][k(f,h,e)], rows / k past the
 * matrix zero — in this order: conv_a [ceil(3 cin / 16)] with k = 16 f + 8 h + e over (dt, c); conv_b [ceil(9 ci / 16)] with the same
 * k over (dy, dx, c); conv_c [cout / 32][ceil(ci / 16)] with k = 16 f + 8 (e >> 2) + 4 h + (e & 3) (the accumulator order of conv_b);
 * the projection [cout / 32][ceil(cin / 16)] with natural k; then fp32 bias_a[32] bias_b[32] bias_c[cout] (+ the projection's bias),
 * zero padded to a multiple of 1 KB.  stride (1 | 2) is the spatial stride of conv_b AND of the projection (a stage's first block,
 * pad 0,1,1: the output map is ceil(H / stride) x ceil(W / stride)).  Built (cin, ci, cout, projection, stride): (8,8,32,1,1)
 * (32,8,32,0,1) (64,16,64,0,1) (128,32,128,0,1) (32,16,64,1,2) (64,32,128,1,2); kvq_fast_bottleneck_pack_bytes returns 0 for anything
 * else. */
size_t kvq_fast_bottleneck_pack_bytes(int cin, int ci, int cout, int projection, int stride);
int kvq_fast_bottleneck(const uint16_t* x, const int32_t dims4[4], int cin, int ci, int cout, int projection, int stride,
                        const void* pack, int dtype, uint16_t* out, void* stream);

/* SlowFast SLOW-pathway identity residual block at res2, fused (csrc/slowneck.hip): conv_a 1x1x1 + BN + ReLU -> conv_b 1x3x3 (pad 0,1,1,
 * stride 1) + BN + ReLU -> conv_c 1x1x1 + BN -> + x -> ReLU (SlowFast_features.py:137-165, blocks 1 .. of the slow pathway's res2).
 * x 16-bit channels-last (B,T,H,W,cin), dims4 = {B,T,H,W}; out rows of out_C >= cout channels (out_C <= 0: dense), channels 0 .. cout-1
 * written (a stage's last block writes into the lateral-concat tensor).  pack as for kvq_fast_bottleneck with conv_a's k over c only and
 * TWO row tiles for conv_a / conv_b: conv_a [2][cin / 16] | conv_b [2][9 ci / 16] | conv_c [cout / 32][ci / 16] (accumulator order) |
 * fp32 bias_a[64] bias_b[64] bias_c[cout], padded to 2 KB.  Built (cin, ci, cout): (256, 64, 256); kvq_slow_bottleneck_pack_bytes returns 0
 * for anything else. */
size_t kvq_slow_bottleneck_pack_bytes(int cin, int ci, int cout);
int kvq_slow_bottleneck(const uint16_t* x, const int32_t dims4[4], int cin, int ci, int cout, const void* pack, int dtype,
                        uint16_t* out, int out_C, void* stream);

/* nn.MaxPool / nn.AvgPool (count_include_pad) on channels-last 16-bit (B,D,H,W,C). */
int kvq_pool_nd(const uint16_t* x, int dtype, const int32_t dims5[5], const int32_t kernel3[3],
                const int32_t stride3[3], const int32_t pad3[3], int is_max, uint16_t* out, void* stream);
/* the same, writing the C channels at channel col_off of output rows of ldc channels (ldc <= 0: dense) */
int kvq_pool_nd_strided(const uint16_t* x, int dtype, const int32_t dims5[5], const int32_t kernel3[3],
                        const int32_t stride3[3], const int32_t pad3[3], int is_max, uint16_t* out, int ldc, int col_off,
                        void* stream);
/* avgpool + global_std_pool2d (simpleVQA_model.py:8-11, 242-252): x 16-bit [rows][HW][C] -> fp32
 * out[row*out_stride + mean_off + c] = mean, out[row*out_stride + std_off + c] = UNBIASED std (std_off < 0: skip). */
int kvq_mean_std_pool(const uint16_t* x, int dtype, int rows, int HW, int C, float* out, int64_t out_stride,
                      int mean_off, int std_off, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KVQ_HIP_H */
