// Whole-network entry for the convolutional branches (kvq_slowfast_r50_forward / kvq_resnet50_simplevqa_forward):
// a plan = the network's layer table (convolutions with folded BatchNorm, pools, stems, global pools) + the shapes of one
// batch; a forward = ONE C call that enqueues every launch on the caller's stream, as kvq_swin3d_forward does for the trunk.
//
// Replaces the per-layer Python sequencing of the reference forwards (SlowFast_features.py:137-165: blocks 0-4 of
// pytorchvideo's slowfast_r50 + the head pools; simpleVQA_model.py:220-264: ResNet-50 + avg / std pooling): no torch kernels
// between the layers (pathway packing = a frame-select launch, torch.cat of the lateral connections = the producing convs
// write their channels at an offset of the wider tensor), no host work per layer beyond the launches themselves.
//
// Tensors are 16-bit channels-last activations (B, D, H, W, C) living in the caller's workspace (slots are recycled after their
// last reader) or fp32 planar network inputs (B, C, T, H, W); weights are borrowed device pointers.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.hpp"

namespace kvq {

struct NetTensorState {
  KvqNetTensor t;
  size_t bytes, off;
  int last_use;        // index of the last op that reads it (-1: never read -> kept to the end)
  bool placed;
  unsigned lanes;      // bit l: some op of lane l touches it
};

struct NetOpState {
  KvqNetOp op;
  int32_t* d_taps;     // CONV (implicit) / STEM8: device tap table
  int tmp;             // STEM8 / STEM_MFMA: internal slot of the packed input
  int Do, Ho, Wo;
  bool pointwise;
  std::vector<int32_t> t_index;
  int wait_ev;         // two-lane plans: event of the other lane's op this one has to see finished (-1: none)
  int signal_ev;       //                 event recorded behind this op (-1: nobody on the other lane waits for it)
};

// pathway packing (SlowFast_features.py:112-135): frames idx[k] of an fp32 (B, C, T, H, W) clip -> (B, C, n, H, W)
__global__ __launch_bounds__(256) void select_frames_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int n, long hw4,
                                                            const int32_t* __restrict__ idx, long total4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long p = i % hw4;
  const long r = i / hw4;                 // (b * C + c) * n + k
  const int k = (int)(r % n);
  const long bc = r / n;
  reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(x)[(bc * T + idx[k]) * hw4 + p];
}

}  // namespace kvq

struct KvqConvNet {
  std::vector<kvq::NetTensorState> tensors;
  std::vector<kvq::NetOpState> ops;
  std::vector<void*> owned;
  int n_inputs, n_outputs, dtype;
  size_t ws_bytes, sk_off, sk_bytes;
  bool sk_on = true;      // kvq_convnet_splitk: false = no launch of this plan cuts K (results independent of the batch size)
  // kvq_convnet_profile: one event before the first op and one after every op of the NEXT forwards (measurement only)
  mutable std::vector<hipEvent_t> events;
  mutable bool recorded = false;
  // ops with lane == 1 run on the plan's own stream (two independent pathways side by side); sync[0] forks it off the
  // caller's stream, sync[1] joins it back, the others order ops of different lanes that touch the same workspace bytes
  hipStream_t lane1 = nullptr;
  std::vector<hipEvent_t> sync;
  size_t sk_stride = 0;
};

namespace kvq {

static size_t tensor_bytes(const KvqNetTensor& t) {
  const size_t n = (size_t)t.B * t.D * t.H * t.W * t.C;
  return ((t.kind == KVQ_NET_T_ACT16 ? n * 2 : n * 4) + 255) & ~(size_t)255;
}

static int upload_i32(KvqConvNet* net, const std::vector<int32_t>& h, int32_t** out) {
  void* d = nullptr;
  KVQ_CHECK_HIP(hipMalloc(&d, h.size() * sizeof(int32_t)));
  net->owned.push_back(d);
  KVQ_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  *out = (int32_t*)d;
  return KVQ_OK;
}

// tap table of kvq_conv_implicit: one row per 8-channel chunk of the (kd, kh, kw, c)-ordered K axis
static std::vector<int32_t> build_taps(const int32_t k3[3], int C, int H, int W, int kpad) {
  std::vector<int32_t> rows((size_t)kpad / 8 * 4, 0);
  size_t q = 0;
  for (int a = 0; a < k3[0]; ++a)
    for (int b = 0; b < k3[1]; ++b)
      for (int c = 0; c < k3[2]; ++c)
        for (int c0 = 0; c0 < C; c0 += 8, ++q) {
          rows[4 * q] = a; rows[4 * q + 1] = b; rows[4 * q + 2] = c; rows[4 * q + 3] = ((a * H + b) * W + c) * C + c0;
        }
  for (; q < (size_t)kpad / 8; ++q) rows[4 * q + 3] = -1;
  return rows;
}

}  // namespace kvq

extern "C" void kvq_convnet_destroy(KvqConvNet* net) {
  if (!net) return;
  for (void* d : net->owned) (void)hipFree(d);
  for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : net->sync) (void)hipEventDestroy(e);
  if (net->lane1) (void)hipStreamDestroy(net->lane1);
  delete net;
}

extern "C" int kvq_convnet_profile(KvqConvNet* net, int enable) {
  using namespace kvq;
  KVQ_REQUIRE(net, KVQ_ERR_NULL, "kvq_convnet_profile: NULL plan");
  for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
  net->events.clear();
  net->recorded = false;
  if (enable) {
    net->events.resize(net->ops.size() + 1);
    for (hipEvent_t& e : net->events) KVQ_CHECK_HIP(hipEventCreate(&e));
  }
  return KVQ_OK;
}

extern "C" int kvq_convnet_profile_read(const KvqConvNet* net, float* ms, int capacity, int* n_ops) {
  using namespace kvq;
  KVQ_REQUIRE(net && ms && n_ops, KVQ_ERR_NULL, "kvq_convnet_profile_read: NULL pointer");
  KVQ_REQUIRE(!net->events.empty() && net->recorded, KVQ_ERR_SHAPE, "kvq_convnet_profile_read: no profiled forward (kvq_convnet_profile(net, 1) first)");
  KVQ_REQUIRE(capacity >= (int)net->ops.size(), KVQ_ERR_WORKSPACE, "kvq_convnet_profile_read: room for %d ops, the plan has %zu", capacity, net->ops.size());
  KVQ_CHECK_HIP(hipEventSynchronize(net->events.back()));
  for (size_t i = 0; i < net->ops.size(); ++i) KVQ_CHECK_HIP(hipEventElapsedTime(&ms[i], net->events[i], net->events[i + 1]));
  *n_ops = (int)net->ops.size();
  return KVQ_OK;
}

extern "C" size_t kvq_convnet_workspace_bytes(const KvqConvNet* net) { return net ? net->ws_bytes : 0; }

extern "C" int kvq_convnet_create(const KvqNetOp* ops, int n_ops, const KvqNetTensor* tensors, int n_tensors, int n_inputs,
                                  int n_outputs, int dtype, KvqConvNet** out) {
  using namespace kvq;
  KVQ_REQUIRE(ops && tensors && out && n_ops > 0 && n_tensors > 0 && n_inputs > 0 && n_inputs <= n_tensors, KVQ_ERR_NULL,
              "kvq_convnet_create: NULL / empty argument");
  KVQ_REQUIRE(dtype == KVQ_DT_BF16 || dtype == KVQ_DT_FP16, KVQ_ERR_UNSUPPORTED, "kvq_convnet_create: dtype %d", dtype);
  KvqConvNet* net = new KvqConvNet();
  net->n_inputs = n_inputs; net->n_outputs = n_outputs; net->dtype = dtype;
  for (int i = 0; i < n_tensors; ++i) {
    NetTensorState t{};
    t.t = tensors[i]; t.bytes = tensor_bytes(tensors[i]); t.off = 0; t.last_use = -1; t.placed = false; t.lanes = 0;
    net->tensors.push_back(t);
  }
  auto fail = [&](int code) { kvq_convnet_destroy(net); return code; };
#define NET_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      kvq::set_error(__VA_ARGS__);    \
      return fail(KVQ_ERR_SHAPE);     \
    }                                 \
  } while (0)
  size_t max_sk = 0;
  for (int i = 0; i < n_ops; ++i) {
    NetOpState o{};
    o.op = ops[i]; o.d_taps = nullptr; o.tmp = -1; o.pointwise = false; o.wait_ev = -1; o.signal_ev = -1;
    NET_REQUIRE(ops[i].lane == 0 || ops[i].lane == 1, "kvq_convnet_create: op %d lane %d", i, ops[i].lane);
    const KvqNetOp& p = ops[i];
    NET_REQUIRE(p.src >= 0 && p.src < (int)net->tensors.size(), "kvq_convnet_create: op %d reads slot %d", i, p.src);
    const KvqNetTensor s = net->tensors[p.src].t;
    net->tensors[p.src].last_use = i;
    const bool has_dst = p.kind != KVQ_NET_MEAN_STD;
    KvqNetTensor d{};
    if (has_dst) {
      NET_REQUIRE(p.dst >= n_inputs && p.dst < n_tensors, "kvq_convnet_create: op %d writes slot %d", i, p.dst);
      d = net->tensors[p.dst].t;
    } else {
      NET_REQUIRE(p.dst >= 0 && p.dst < n_outputs, "kvq_convnet_create: op %d writes output %d of %d", i, p.dst, n_outputs);
    }
    auto odim = [&](int n, int a) { return (n + 2 * p.pad3[a] - p.kernel3[a]) / p.stride3[a] + 1; };
    switch (p.kind) {
      case KVQ_NET_CONV: {
        NET_REQUIRE(s.kind == KVQ_NET_T_ACT16 && (d.kind == KVQ_NET_T_ACT16 || d.kind == KVQ_NET_T_ACT32) && s.C % 8 == 0,
                    "kvq_convnet_create: op %d (conv) operand kinds", i);
        NET_REQUIRE(d.kind == KVQ_NET_T_ACT16 || (!p.relu && p.src2 < 0 && p.dst32 < 0 && d.C == p.cout),
                    "kvq_convnet_create: op %d: an fp32 destination takes conv + bias only, dense", i);
        o.Do = odim(s.D, 0); o.Ho = odim(s.H, 1); o.Wo = odim(s.W, 2);
        NET_REQUIRE(d.B == s.B && d.D == o.Do && d.H == o.Ho && d.W == o.Wo && p.cout % 8 == 0 && p.dst_coff % 8 == 0 && d.C % 8 == 0 &&
                    p.dst_coff + p.cout <= d.C, "kvq_convnet_create: op %d (conv) output shape (%d,%d,%d,%d,%d) vs (%d,%d,%d,%d,>=%d)", i,
                    d.B, d.D, d.H, d.W, d.C, s.B, o.Do, o.Ho, o.Wo, p.dst_coff + p.cout);
        const bool one = p.kernel3[0] == 1 && p.kernel3[1] == 1 && p.kernel3[2] == 1 && p.stride3[0] == 1 && p.stride3[1] == 1 && p.stride3[2] == 1;
        o.pointwise = one && s.C % 32 == 0;
        NET_REQUIRE(p.w && p.kpad % 32 == 0 && p.kpad >= (o.pointwise ? s.C : p.kernel3[0] * p.kernel3[1] * p.kernel3[2] * s.C),
                    "kvq_convnet_create: op %d (conv) weight columns %d", i, p.kpad);
        NET_REQUIRE(!o.pointwise || p.kpad == s.C, "kvq_convnet_create: op %d (1x1x1 conv) needs kpad == C", i);
        if (p.src2 >= 0) {
          NET_REQUIRE(p.src2 < n_tensors && p.relu, "kvq_convnet_create: op %d identity slot %d", i, p.src2);
          const KvqNetTensor r = net->tensors[p.src2].t;
          NET_REQUIRE((r.kind == KVQ_NET_T_ACT16 || r.kind == KVQ_NET_T_ACT32) && r.C == p.cout &&
                      (size_t)r.B * r.D * r.H * r.W == (size_t)d.B * d.D * d.H * d.W, "kvq_convnet_create: op %d identity branch shape", i);
          net->tensors[p.src2].last_use = i;
        }
        if (p.dst32 >= 0) {
          NET_REQUIRE(p.dst32 >= n_inputs && p.dst32 < n_tensors && p.relu && d.C == p.cout, "kvq_convnet_create: op %d fp32 copy slot %d", i, p.dst32);
          const KvqNetTensor r = net->tensors[p.dst32].t;
          NET_REQUIRE(r.kind == KVQ_NET_T_ACT32 && r.C == p.cout && r.B == d.B && r.D == d.D && r.H == d.H && r.W == d.W,
                      "kvq_convnet_create: op %d fp32 copy shape", i);
        }
        if (!o.pointwise && s.C % 32 != 0) {     // C % 32 == 0: the kernel walks the taps itself
          int rc = upload_i32(net, build_taps(p.kernel3, s.C, s.H, s.W, p.kpad), &o.d_taps);
          if (rc) return fail(rc);
        }
        max_sk = std::max(max_sk, kvq_gemm_splitk_bytes(s.B * o.Do * o.Ho * o.Wo, p.cout, p.kpad));
        break;
      }
      case KVQ_NET_POOL: {
        NET_REQUIRE(s.kind == KVQ_NET_T_ACT16 && d.kind == KVQ_NET_T_ACT16, "kvq_convnet_create: op %d (pool) operand kinds", i);
        o.Do = odim(s.D, 0); o.Ho = odim(s.H, 1); o.Wo = odim(s.W, 2);
        NET_REQUIRE(d.B == s.B && d.D == o.Do && d.H == o.Ho && d.W == o.Wo && p.dst_coff + s.C <= d.C && (d.C == s.C || (d.C % 8 == 0 && s.C % 8 == 0 && p.dst_coff % 8 == 0)),
                    "kvq_convnet_create: op %d (pool) output shape", i);
        break;
      }
      case KVQ_NET_STEM8:
      case KVQ_NET_STEM_MFMA: {
        NET_REQUIRE(s.kind == KVQ_NET_T_F32_PLANAR && d.kind == KVQ_NET_T_ACT16 && s.C <= (p.kind == KVQ_NET_STEM8 ? 8 : 4) && p.w,
                    "kvq_convnet_create: op %d (stem) operand kinds", i);
        o.Do = odim(s.D, 0); o.Ho = odim(s.H, 1); o.Wo = odim(s.W, 2);
        NET_REQUIRE(d.B == s.B && d.D == o.Do && d.H == o.Ho && d.W == o.Wo && d.C == p.cout, "kvq_convnet_create: op %d (stem) output shape", i);
        KvqNetTensor tmp{};
        tmp.kind = KVQ_NET_T_ACT16; tmp.B = s.B; tmp.D = s.D; tmp.H = s.H;
        if (p.kind == KVQ_NET_STEM8) {
          NET_REQUIRE(p.kernel3[0] == 1 && p.kpad % 32 == 0 && p.kpad >= p.kernel3[1] * p.kernel3[2] * 8 && p.cout % 8 == 0,
                      "kvq_convnet_create: op %d (stem8) needs a 1 x kh x kw kernel over the 8-channel packed input", i);
          tmp.W = s.W; tmp.C = 8;
          int rc = upload_i32(net, build_taps(p.kernel3, 8, s.H, s.W, p.kpad), &o.d_taps);
          if (rc) return fail(rc);
        } else {
          NET_REQUIRE(p.cout == 8 && p.kernel3[2] == 7 && p.stride3[2] == 2 && p.pad3[2] == 3, "kvq_convnet_create: op %d (stem_mfma) geometry", i);
          tmp.W = s.W + 8; tmp.C = 4;
        }
        NetTensorState ts{};
        ts.t = tmp; ts.bytes = tensor_bytes(tmp); ts.last_use = i; ts.placed = false; ts.lanes = 0;
        o.tmp = (int)net->tensors.size();
        net->tensors.push_back(ts);
        break;
      }
      case KVQ_NET_STEM_POOL: {
        NET_REQUIRE(s.kind == KVQ_NET_T_F32_PLANAR && d.kind == KVQ_NET_T_ACT16 && s.C == 3 && p.w && p.bias, "kvq_convnet_create: op %d (stem + pool) operand kinds", i);
        NET_REQUIRE(p.cout == 8 && p.kernel3[1] == 7 && p.kernel3[2] == 7 && p.stride3[0] == 1 && p.stride3[1] == 2 && p.stride3[2] == 2 &&
                        p.pad3[0] == p.kernel3[0] / 2 && p.pad3[1] == 3 && p.pad3[2] == 3 && (p.kernel3[0] & 1),
                    "kvq_convnet_create: op %d (stem + pool) geometry", i);
        NET_REQUIRE(stem_pool_shape_ok(s.B, s.D, s.H, s.W, p.kernel3[0]),
                    "kvq_convnet_create: op %d (stem + pool): clip %d x %d x %d with a %d-frame kernel is outside the fused stem's limits (W %% 4 == 0, "
                    "W <= 256, temporal kernel <= 7, 96 KB of LDS)", i, s.D, s.H, s.W, p.kernel3[0]);
        o.Do = s.D; o.Ho = ((s.H - 1) / 2 + 1 - 1) / 2 + 1; o.Wo = ((s.W - 1) / 2 + 1 - 1) / 2 + 1;
        NET_REQUIRE(d.B == s.B && d.D == o.Do && d.H == o.Ho && d.W == o.Wo && d.C == 8, "kvq_convnet_create: op %d (stem + pool) output shape", i);
        break;
      }
      case KVQ_NET_STEM64_POOL: {
        NET_REQUIRE(s.kind == KVQ_NET_T_F32_PLANAR && d.kind == KVQ_NET_T_ACT16 && s.C == 3 && p.w && p.bias && p.t_index && p.n_index > 0,
                    "kvq_convnet_create: op %d (stem64 + pool) operands", i);
        NET_REQUIRE(p.cout == 64 && p.kernel3[0] == 1 && p.kernel3[1] == 7 && p.kernel3[2] == 7 && p.stride3[0] == 1 && p.stride3[1] == 2 &&
                        p.stride3[2] == 2 && p.pad3[0] == 0 && p.pad3[1] == 3 && p.pad3[2] == 3 && s.W <= 224 && s.W % 4 == 0,
                    "kvq_convnet_create: op %d (stem64 + pool) geometry", i);
        o.Do = p.n_index; o.Ho = ((s.H - 1) / 2 + 1 - 1) / 2 + 1; o.Wo = ((s.W - 1) / 2 + 1 - 1) / 2 + 1;
        NET_REQUIRE(d.B == s.B && d.D == o.Do && d.H == o.Ho && d.W == o.Wo && p.dst_coff % 8 == 0 && d.C % 8 == 0 && p.dst_coff + 64 <= d.C,
                    "kvq_convnet_create: op %d (stem64 + pool) output shape", i);
        for (int k = 0; k < p.n_index; ++k) NET_REQUIRE(p.t_index[k] >= 0 && p.t_index[k] < s.D, "kvq_convnet_create: op %d frame index %d", i, p.t_index[k]);
        o.t_index.assign(p.t_index, p.t_index + p.n_index);
        int rc = upload_i32(net, o.t_index, &o.d_taps);
        if (rc) return fail(rc);
        break;
      }
      case KVQ_NET_MEAN_STD:
        NET_REQUIRE(s.kind == KVQ_NET_T_ACT16 && p.out_stride > 0 && p.mean_off >= 0, "kvq_convnet_create: op %d (mean/std pool)", i);
        break;
      case KVQ_NET_SELECT_T: {
        NET_REQUIRE(s.kind == KVQ_NET_T_F32_PLANAR && d.kind == KVQ_NET_T_F32_PLANAR && p.t_index && p.n_index > 0 && d.B == s.B && d.C == s.C &&
                    d.D == p.n_index && d.H == s.H && d.W == s.W && (s.H * s.W) % 4 == 0, "kvq_convnet_create: op %d (frame select) shape", i);
        o.t_index.assign(p.t_index, p.t_index + p.n_index);
        for (int v : o.t_index) NET_REQUIRE(v >= 0 && v < s.D, "kvq_convnet_create: op %d frame index %d", i, v);
        int rc = upload_i32(net, o.t_index, &o.d_taps);
        if (rc) return fail(rc);
        break;
      }
      case KVQ_NET_BOTTLENECK_S: {
        NET_REQUIRE(s.kind == KVQ_NET_T_ACT16 && d.kind == KVQ_NET_T_ACT16 && p.w, "kvq_convnet_create: op %d (slow bottleneck) operand kinds", i);
        NET_REQUIRE(kvq_slow_bottleneck_pack_bytes(s.C, p.kpad, p.cout) && d.B == s.B && d.D == s.D && d.H == s.H && d.W == s.W && d.C >= p.cout &&
                        d.C % 8 == 0 && p.dst_coff == 0,
                    "kvq_convnet_create: op %d (slow bottleneck %d -> %d -> %d) shape", i, s.C, p.kpad, p.cout);
        break;
      }
      case KVQ_NET_BOTTLENECK: {
        NET_REQUIRE(s.kind == KVQ_NET_T_ACT16 && d.kind == KVQ_NET_T_ACT16 && p.w, "kvq_convnet_create: op %d (bottleneck) operand kinds", i);
        const int bs = p.stride3[1];
        NET_REQUIRE(p.stride3[0] == 1 && p.stride3[2] == bs && (bs == 1 || bs == 2), "kvq_convnet_create: op %d (bottleneck) stride", i);
        NET_REQUIRE(d.B == s.B && d.D == s.D && d.H == (s.H + bs - 1) / bs && d.W == (s.W + bs - 1) / bs && d.C == p.cout,
                    "kvq_convnet_create: op %d (bottleneck) output shape", i);
        NET_REQUIRE(kvq_fast_bottleneck_pack_bytes(s.C, p.kpad, p.cout, p.n_index, bs) > 0,
                    "kvq_convnet_create: op %d (bottleneck) channels %d -> %d -> %d (projection %d, stride %d) not built", i, s.C, p.kpad, p.cout,
                    p.n_index, bs);
        break;
      }
      default:
        NET_REQUIRE(false, "kvq_convnet_create: op %d unknown kind %d", i, p.kind);
    }
    o.op.t_index = nullptr;          // the host table was copied
    net->ops.push_back(o);
  }
#undef NET_REQUIRE
  // ---- which lanes touch which slot ----
  bool two_lanes = false;
  auto slots_of = [&](const NetOpState& o, int* rd, int& nr, int* wr, int& nw) {
    nr = nw = 0;
    rd[nr++] = o.op.src;
    if (o.op.kind == KVQ_NET_CONV && o.op.src2 >= 0) rd[nr++] = o.op.src2;
    if (o.tmp >= 0) { rd[nr++] = o.tmp; wr[nw++] = o.tmp; }
    if (o.op.kind != KVQ_NET_MEAN_STD) wr[nw++] = o.op.dst;
    if (o.op.kind == KVQ_NET_CONV && o.op.dst32 >= 0) wr[nw++] = o.op.dst32;
  };
  for (const NetOpState& o : net->ops) {
    int rd[4], wr[4], nr, nw;
    slots_of(o, rd, nr, wr, nw);
    for (int k = 0; k < nr; ++k) net->tensors[rd[k]].lanes |= 1u << o.op.lane;
    for (int k = 0; k < nw; ++k) net->tensors[wr[k]].lanes |= 1u << o.op.lane;
    two_lanes = two_lanes || o.op.lane == 1;
  }
  // ---- workspace layout: slots are placed when first written and recycled after their last reader (first fit).  A released
  // range is only handed to a slot touched by the SAME lanes, so recycling never makes one pathway wait for the other ----
  struct Free { size_t off, bytes; unsigned lanes; };
  std::vector<Free> free_list;
  size_t top = 0;
  auto place = [&](int slot) {
    NetTensorState& t = net->tensors[slot];
    if (t.placed || slot < n_inputs) return;
    for (size_t f = 0; f < free_list.size(); ++f)
      if (free_list[f].bytes >= t.bytes && free_list[f].lanes == t.lanes) {
        t.off = free_list[f].off;
        free_list[f].off += t.bytes; free_list[f].bytes -= t.bytes;
        t.placed = true;
        return;
      }
    t.off = top; top += t.bytes; t.placed = true;
  };
  for (int i = 0; i < (int)net->ops.size(); ++i) {
    const NetOpState& o = net->ops[i];
    if (o.tmp >= 0) place(o.tmp);
    if (o.op.kind != KVQ_NET_MEAN_STD) place(o.op.dst);
    if (o.op.kind == KVQ_NET_CONV && o.op.dst32 >= 0) place(o.op.dst32);
    for (int slot = n_inputs; slot < (int)net->tensors.size(); ++slot) {
      NetTensorState& t = net->tensors[slot];
      if (t.placed && t.last_use == i && t.bytes) {
        free_list.push_back({t.off, t.bytes, t.lanes});
        t.last_use = -2;              // released
      }
    }
  }
  net->sk_stride = (max_sk + 255) & ~(size_t)255;
  net->sk_off = top; net->sk_bytes = max_sk;
  net->ws_bytes = top + net->sk_stride * (two_lanes ? 2 : 1);
  // ---- two lanes: op i waits for the latest op of the OTHER lane whose workspace bytes it conflicts with (one of the two
  // accesses a write; the same wide tensor written at different channel offsets counts as a conflict: conservative) ----
  if (two_lanes) {
    KVQ_CHECK_HIP(hipStreamCreateWithFlags(&net->lane1, hipStreamNonBlocking));
    net->sync.resize(2);
    auto range = [&](int slot, size_t& lo, size_t& hi) {
      if (slot < n_inputs) return false;            // the caller's inputs are only read
      lo = net->tensors[slot].off; hi = lo + net->tensors[slot].bytes;
      return hi > lo;
    };
    auto overlap = [&](int a, int b) {
      size_t la, ha, lb, hb;
      return range(a, la, ha) && range(b, lb, hb) && la < hb && lb < ha;
    };
    int waited[2] = {-1, -1};                        // latest op of the other lane each lane already waits for
    const int n = (int)net->ops.size();
    std::vector<int> wait_op(n, -1);
    for (int i = 0; i < n; ++i) {
      int ri[4], wi[4], nri, nwi;
      slots_of(net->ops[i], ri, nri, wi, nwi);
      const int lane = net->ops[i].op.lane;
      int dep = -1;
      for (int j = i - 1; j > waited[lane] && dep < 0; --j) {
        if (net->ops[j].op.lane == lane) continue;
        int rj[4], wj[4], nrj, nwj;
        slots_of(net->ops[j], rj, nrj, wj, nwj);
        bool hit = false;
        for (int a = 0; a < nwi && !hit; ++a) {
          for (int b = 0; b < nrj && !hit; ++b) hit = overlap(wi[a], rj[b]);
          for (int b = 0; b < nwj && !hit; ++b) hit = overlap(wi[a], wj[b]);
        }
        for (int a = 0; a < nri && !hit; ++a)
          for (int b = 0; b < nwj && !hit; ++b) hit = overlap(ri[a], wj[b]);
        if (hit) dep = j;
      }
      if (dep >= 0) { wait_op[i] = dep; waited[lane] = dep; }
    }
    for (int i = 0; i < n; ++i) {
      if (wait_op[i] < 0) continue;
      NetOpState& src_op = net->ops[wait_op[i]];
      if (src_op.signal_ev < 0) { src_op.signal_ev = (int)net->sync.size(); net->sync.push_back(nullptr); }
      net->ops[i].wait_ev = src_op.signal_ev;
    }
    for (hipEvent_t& e : net->sync) KVQ_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  *out = net;
  return KVQ_OK;
}

#define KVQ_TRY(expr)      \
  do {                     \
    int _rc = (expr);      \
    if (_rc) return _rc;   \
  } while (0)

extern "C" int kvq_convnet_forward(const KvqConvNet* net, const void* const* inputs, float* const* outputs, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  using namespace kvq;
  KVQ_REQUIRE(net && inputs && workspace, KVQ_ERR_NULL, "kvq_convnet_forward: NULL pointer");
  KVQ_REQUIRE(workspace_bytes >= net->ws_bytes, KVQ_ERR_WORKSPACE, "kvq_convnet_forward: workspace %zu < %zu bytes", workspace_bytes, net->ws_bytes);
  unsigned char* ws = (unsigned char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  auto ptr_of = [&](int slot) -> void* { return slot < net->n_inputs ? const_cast<void*>(inputs[slot]) : (void*)(ws + net->tensors[slot].off); };
  for (int i = 0; i < net->n_inputs; ++i) KVQ_REQUIRE(inputs[i], KVQ_ERR_NULL, "kvq_convnet_forward: input %d is NULL", i);
  const bool prof = !net->events.empty();
  const bool fork = net->lane1 && !prof;          // a profiled forward runs every op on the caller's stream, one after the other
  const hipStream_t caller = st;
  if (prof) KVQ_CHECK_HIP(hipEventRecord(net->events[0], st));
  if (fork) {
    KVQ_CHECK_HIP(hipEventRecord(net->sync[0], caller));
    KVQ_CHECK_HIP(hipStreamWaitEvent(net->lane1, net->sync[0], 0));
  }
  size_t op_index = 0;
  int prev_signal = -1;
  // the ops run inside a lambda: an op that fails must not leave the second stream forked off — the join below runs either way,
  // so the caller's stream (and whoever frees or reuses the inputs / workspace after the error) is ordered behind lane 1
  const int rc_ops = [&]() -> int {
  for (const NetOpState& o : net->ops) {
    if (prof && op_index) KVQ_CHECK_HIP(hipEventRecord(net->events[op_index], st));
    if (fork && prev_signal >= 0) KVQ_CHECK_HIP(hipEventRecord(net->sync[prev_signal], st));     // behind the previous op, on ITS stream
    ++op_index;
    const KvqNetOp& p = o.op;
    st = fork && p.lane == 1 ? net->lane1 : caller;
    if (fork && o.wait_ev >= 0) KVQ_CHECK_HIP(hipStreamWaitEvent(st, net->sync[o.wait_ev], 0));
    prev_signal = o.signal_ev;
    unsigned char* const sk_ws = ws + net->sk_off + (fork && p.lane == 1 ? net->sk_stride : 0);
    const KvqNetTensor& s = net->tensors[p.src].t;
    switch (p.kind) {
      case KVQ_NET_CONV: {
        const KvqNetTensor& d = net->tensors[p.dst].t;
        const int M = s.B * o.Do * o.Ho * o.Wo;
        const bool f32dst = d.kind == KVQ_NET_T_ACT32;
        const int epi = f32dst ? KVQ_EPI_STORE_F32 : (p.relu ? KVQ_EPI_RELU_BF16 : KVQ_EPI_BIAS_BF16);
        const bool wide = d.C != p.cout;
        const bool r32 = p.src2 >= 0 && net->tensors[p.src2].t.kind == KVQ_NET_T_ACT32;
        if (o.pointwise) {
          KvqGemmArgs a{};
          a.A = (const uint16_t*)ptr_of(p.src); a.W = (const uint16_t*)p.w; a.bias = p.bias; a.M = M; a.N = p.cout; a.K = p.kpad;
          a.epilogue = epi; a.dtype = net->dtype;
          if (f32dst) a.out_f32 = (float*)ptr_of(p.dst);
          else a.out_bf16 = (uint16_t*)ptr_of(p.dst);
          if (p.dst32 >= 0) a.out_f32 = (float*)ptr_of(p.dst32);
          a.resid_bf16 = p.src2 >= 0 && !r32 ? (const uint16_t*)ptr_of(p.src2) : nullptr;
          a.resid_f32 = r32 ? (const float*)ptr_of(p.src2) : nullptr;
          a.ldc = wide ? d.C : 0; a.col_off = wide ? p.dst_coff : 0;
          if (net->sk_on && net->sk_bytes) { a.splitk_ws = sk_ws; a.splitk_ws_bytes = net->sk_bytes; }
          KVQ_TRY(kvq_gemm_bf16(&a, st));
        } else {
          KvqConvArgs a{};
          a.x = (const uint16_t*)ptr_of(p.src); a.W = (const uint16_t*)p.w; a.bias = p.bias; a.taps = o.d_taps;
          a.dims5[0] = s.B; a.dims5[1] = s.C; a.dims5[2] = s.D; a.dims5[3] = s.H; a.dims5[4] = s.W;
          memcpy(a.kernel3, p.kernel3, sizeof(a.kernel3)); memcpy(a.stride3, p.stride3, sizeof(a.stride3)); memcpy(a.pad3, p.pad3, sizeof(a.pad3));
          a.Kpad = p.kpad; a.N = p.cout; a.epilogue = epi; a.dtype = net->dtype;
          if (f32dst) a.out_f32 = (float*)ptr_of(p.dst);
          else a.out_bf16 = (uint16_t*)ptr_of(p.dst);
          if (p.dst32 >= 0) a.out_f32 = (float*)ptr_of(p.dst32);
          a.resid_bf16 = p.src2 >= 0 && !r32 ? (const uint16_t*)ptr_of(p.src2) : nullptr;
          a.resid_f32 = r32 ? (const float*)ptr_of(p.src2) : nullptr;
          a.ldc = wide ? d.C : 0; a.col_off = wide ? p.dst_coff : 0;
          if (net->sk_on && net->sk_bytes) { a.splitk_ws = sk_ws; a.splitk_ws_bytes = net->sk_bytes; }
          KVQ_TRY(kvq_conv_implicit(&a, st));
        }
        break;
      }
      case KVQ_NET_POOL: {
        const KvqNetTensor& d = net->tensors[p.dst].t;
        const int32_t dims5[5] = {s.B, s.C, s.D, s.H, s.W};
        KVQ_TRY(kvq_pool_nd_strided((const uint16_t*)ptr_of(p.src), net->dtype, dims5, p.kernel3, p.stride3, p.pad3, p.is_max,
                                    (uint16_t*)ptr_of(p.dst), d.C != s.C ? d.C : 0, d.C != s.C ? p.dst_coff : 0, st));
        break;
      }
      case KVQ_NET_STEM8: {
        // the 3-channel input packed to 8 channels (one 16-byte chunk per pixel), then the implicit GEMM over kh x kw taps
        const int32_t dims5[5] = {s.B, s.D, s.C, s.H, s.W};                               // {B, T, C, H, W} addressed through strides
        const int64_t strides5[5] = {(int64_t)s.C * s.D * s.H * s.W, (int64_t)s.H * s.W, (int64_t)s.D * s.H * s.W, s.W, 1};
        uint16_t* x8 = (uint16_t*)ptr_of(o.tmp);
        KVQ_TRY(kvq_pack_channels_last8((const float*)ptr_of(p.src), dims5, strides5, net->dtype, x8, st));
        KvqConvArgs a{};
        a.x = x8; a.W = (const uint16_t*)p.w; a.bias = p.bias; a.taps = o.d_taps;
        a.dims5[0] = s.B; a.dims5[1] = 8; a.dims5[2] = s.D; a.dims5[3] = s.H; a.dims5[4] = s.W;
        memcpy(a.kernel3, p.kernel3, sizeof(a.kernel3)); memcpy(a.stride3, p.stride3, sizeof(a.stride3)); memcpy(a.pad3, p.pad3, sizeof(a.pad3));
        a.Kpad = p.kpad; a.N = p.cout; a.epilogue = p.relu ? KVQ_EPI_RELU_BF16 : KVQ_EPI_BIAS_BF16; a.dtype = net->dtype;
        a.out_bf16 = (uint16_t*)ptr_of(p.dst);
        KVQ_TRY(kvq_conv_implicit(&a, st));
        break;
      }
      case KVQ_NET_STEM_MFMA: {
        const int32_t dims5[5] = {s.B, s.C, s.D, s.H, s.W};
        uint16_t* x4 = (uint16_t*)ptr_of(o.tmp);
        KVQ_TRY(kvq_pack_clip_cl4((const float*)ptr_of(p.src), dims5, 4, net->dtype, x4, st));
        const int32_t dims4[4] = {s.B, s.D, s.H, s.W};
        KVQ_TRY(kvq_conv_stem_mfma(x4, dims4, (const uint16_t*)p.w, p.bias, p.kernel3, p.stride3, p.pad3, p.relu, net->dtype,
                                   (uint16_t*)ptr_of(p.dst), st));
        break;
      }
      case KVQ_NET_STEM_POOL: {
        const int32_t dims5[5] = {s.B, s.C, s.D, s.H, s.W};
        KVQ_TRY(kvq_conv_stem_pool((const float*)ptr_of(p.src), dims5, (const uint16_t*)p.w, p.bias, p.kernel3[0], p.relu, net->dtype,
                                   (uint16_t*)ptr_of(p.dst), st));
        break;
      }
      case KVQ_NET_STEM64_POOL: {
        const int32_t dims5[5] = {s.B, s.C, s.D, s.H, s.W};
        KVQ_TRY(kvq_conv_stem64_pool((const float*)ptr_of(p.src), dims5, o.d_taps, p.n_index, (const uint16_t*)p.w, p.bias, p.relu, net->dtype,
                                     (uint16_t*)ptr_of(p.dst), net->tensors[p.dst].t.C, p.dst_coff, st));
        break;
      }
      case KVQ_NET_BOTTLENECK_S: {
        const int32_t dims4[4] = {s.B, s.D, s.H, s.W};
        KVQ_TRY(kvq_slow_bottleneck((const uint16_t*)ptr_of(p.src), dims4, s.C, p.kpad, p.cout, p.w, net->dtype, (uint16_t*)ptr_of(p.dst),
                                    net->tensors[p.dst].t.C, st));
        break;
      }
      case KVQ_NET_BOTTLENECK: {
        const int32_t dims4[4] = {s.B, s.D, s.H, s.W};
        KVQ_TRY(kvq_fast_bottleneck((const uint16_t*)ptr_of(p.src), dims4, s.C, p.kpad, p.cout, p.n_index, p.stride3[1], p.w, net->dtype,
                                    (uint16_t*)ptr_of(p.dst), st));
        break;
      }
      case KVQ_NET_MEAN_STD:
        KVQ_REQUIRE(outputs && outputs[p.dst], KVQ_ERR_NULL, "kvq_convnet_forward: output %d is NULL", p.dst);
        KVQ_TRY(kvq_mean_std_pool((const uint16_t*)ptr_of(p.src), net->dtype, p.per_frame ? s.B * s.D : s.B,
                                  p.per_frame ? s.H * s.W : s.D * s.H * s.W, s.C, outputs[p.dst], p.out_stride, p.mean_off, p.std_off, st));
        break;
      case KVQ_NET_SELECT_T: {
        const long hw4 = (long)s.H * s.W / 4, total4 = (long)s.B * s.C * p.n_index * hw4;
        hipLaunchKernelGGL(select_frames_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, (const float*)ptr_of(p.src),
                           (float*)ptr_of(p.dst), s.D, p.n_index, hw4, o.d_taps, total4);
        KVQ_CHECK_LAUNCH("select_frames_kernel");
        break;
      }
      default:
        break;
    }
  }
  return KVQ_OK;
  }();
  if (fork) {
    hipError_t je = hipSuccess;
    if (rc_ops == KVQ_OK && prev_signal >= 0) je = hipEventRecord(net->sync[prev_signal], st);
    if (je == hipSuccess) je = hipEventRecord(net->sync[1], net->lane1);
    if (je == hipSuccess) je = hipStreamWaitEvent(caller, net->sync[1], 0);
    if (je != hipSuccess) (void)hipStreamSynchronize(net->lane1);      // last resort: never return with lane 1 un-joined
    if (rc_ops == KVQ_OK && je != hipSuccess) return kvq::hip_fail(je, "kvq_convnet_forward: joining the second stream");
  }
  if (rc_ops) return rc_ops;
  if (prof) {
    KVQ_CHECK_HIP(hipEventRecord(net->events[net->ops.size()], caller));
    net->recorded = true;
  }
  return KVQ_OK;
}

extern "C" int kvq_convnet_splitk(KvqConvNet* net, int enable) {
  KVQ_REQUIRE(net, KVQ_ERR_NULL, "kvq_convnet_splitk: NULL plan");
  net->sk_on = enable != 0;
  return KVQ_OK;
}
