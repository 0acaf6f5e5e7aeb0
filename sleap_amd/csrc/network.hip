// Whole-network entry points of the C ABI: a compiled launch plan executed from C, so that a non-Python host can run the hot path
// (network forward, and forward + peak finding + PAF grouping) through include/sleap_amd.h alone.
//
// Replaces, for one batch: `keras_model(imgs)` inside InferenceLayer.call (sleap/nn/inference.py:2864-2890) and
// `BottomUpInferenceLayer.call` (:2938-3003). The plan is what `sleap_amd.nn.engine.DeviceNetwork` compiles from the Keras graph
// of `best_model.h5`; `DeviceNetwork.plan_words()` serialises it and `DeviceNetwork.forward` itself runs through
// sa_network_forward, so the Python product path and a C host execute the same code.
//
// Host-only translation unit (no kernels): every launch goes through the per-layer entry points of this same library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sa_common.h"

namespace {

constexpr int64_t PLAN_MAGIC = 0x53414E4554303032LL;  // "SANET002" (002: + the activation layout word)

enum OpKind : int64_t {
  K_STEM2 = 1, K_STEM = 2, K_CONV = 3, K_PAIR = 4, K_CONV1X1 = 5, K_CONVT2 = 6, K_CONVT = 7, K_POOLG = 8, K_IMGCONV = 9,
  K_ADD = 10, K_HEAD = 11, K_POOL = 12, K_UP = 13, K_BNECK = 14,
};

struct Buf {
  int cp, num, den, kind;  // kind: 0 = 16-bit activations, 1 = float32 (head output), 2 = virtual (shape only: a tensor the
                           // fusion passes keep on chip; never allocated)
};
struct Out {
  int buf, c, is_f32;
};
struct Op {
  int64_t kind;
  std::vector<int64_t> a;
};

template <typename T>
T* P(int64_t v) {
  return reinterpret_cast<T*>(static_cast<uintptr_t>(v));
}

}  // namespace

struct sa_network {
  std::vector<Buf> bufs;
  std::vector<Out> outs;
  std::vector<Op> ops;
  int in_channels = 0, max_stride = 1;
  int layout = SA_LAYOUT_NHWC;  // of every 16-bit plan tensor (SA_LAYOUT_PLANES16: the compiler checked that each launch of the plan supports it)
};

namespace {

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// byte offsets of every plan buffer inside the workspace for a (B, H, W) input; returns the total
size_t layout(const sa_network* n, int B, int H, int W, std::vector<size_t>* off) {
  size_t total = 0;
  if (off) off->assign(n->bufs.size(), 0);
  for (size_t i = 0; i < n->bufs.size(); ++i) {
    const Buf& b = n->bufs[i];
    if (b.kind == 2) continue;
    const size_t h = (size_t)H * b.num / b.den, w = (size_t)W * b.num / b.den;
    if (off) (*off)[i] = total;
    total += align256((size_t)B * h * w * b.cp * (b.kind == 1 ? 4 : 2));
  }
  return total;
}

}  // namespace

extern "C" {

int sa_network_create(const int64_t* plan, size_t n_words, sa_network_t** out) {
  SA_REQUIRE(plan && out && n_words >= 8, "sa_network_create: NULL / short plan");
  SA_REQUIRE(plan[0] == PLAN_MAGIC, "sa_network_create: bad magic (plan words come from DeviceNetwork.plan_words())");
  size_t i = 1;
  const int64_t n_buf = plan[i++], n_out = plan[i++], n_ops = plan[i++];
  auto* n = new sa_network;
  n->in_channels = (int)plan[i++];
  n->max_stride = (int)plan[i++];
  n->layout = (int)plan[i++];
  if (n->layout != SA_LAYOUT_NHWC && n->layout != SA_LAYOUT_PLANES16) {
    delete n;
    return sa::fail(SA_ERR_INVALID_ARG, "sa_network_create: unknown activation layout %d", n->layout);
  }
  if (!(n_buf >= 0 && n_out > 0 && n_ops > 0)) {
    delete n;
    return sa::fail(SA_ERR_INVALID_ARG, "sa_network_create: empty plan");
  }
  auto need = [&](size_t k) { return i + k <= n_words; };
  bool ok = need((size_t)n_buf * 4);
  for (int64_t b = 0; ok && b < n_buf; ++b, i += 4) n->bufs.push_back({(int)plan[i], (int)plan[i + 1], (int)plan[i + 2], (int)plan[i + 3]});
  ok = ok && need((size_t)n_out * 3);
  for (int64_t o = 0; ok && o < n_out; ++o, i += 3) n->outs.push_back({(int)plan[i], (int)plan[i + 1], (int)plan[i + 2]});
  for (int64_t o = 0; ok && o < n_ops; ++o) {
    ok = need(2);
    if (!ok) break;
    Op op;
    op.kind = plan[i++];
    const int64_t na = plan[i++];
    ok = na >= 0 && need((size_t)na);
    if (!ok) break;
    op.a.assign(plan + i, plan + i + na);
    i += (size_t)na;
    n->ops.push_back(std::move(op));
  }
  if (!ok || i != n_words) {
    delete n;
    return sa::fail(SA_ERR_INVALID_ARG, "sa_network_create: truncated or over-long plan (%zu of %zu words consumed)", i, n_words);
  }
  for (const Out& o : n->outs)
    if (o.buf < 0 || o.buf >= (int)n->bufs.size() || n->bufs[(size_t)o.buf].kind == 2) {
      delete n;
      return sa::fail(SA_ERR_INVALID_ARG, "sa_network_create: output refers to buffer %d", o.buf);
    }
  if (n->layout == SA_LAYOUT_PLANES16) {
    bool fits = true;
    for (const Op& op : n->ops) {
      // (round 4: the tap GEMM -- 1x1 / k x k / transposed convs -- has a plane variant; stand-alone pools and adds run per plane)
      fits = fits && (op.kind == K_STEM2 || op.kind == K_PAIR || op.kind == K_UP || op.kind == K_CONV || op.kind == K_IMGCONV ||
                      op.kind == K_HEAD || op.kind == K_CONV1X1 || op.kind == K_CONVT2 || op.kind == K_POOLG || op.kind == K_POOL ||
                      op.kind == K_ADD || op.kind == K_BNECK);
      if (op.kind == K_CONV && op.a.size() >= 10) {  // plain / concat sources; the upsampling source mode without fused heads
        const bool ext = op.a[10 + 5 * (size_t)op.a[9]] != 0;  // and without the extended (BN / residual) epilogue
        fits = fits && (op.a[2] == SA_SRC1_NONE || op.a[2] == SA_SRC1_DIRECT ||
                        (op.a[2] == SA_SRC1_UPSAMPLE2X && op.a[9] == 0 && !ext));
      }
      if (op.kind == K_IMGCONV) fits = fits && n->bufs[(size_t)op.a[0]].cp % 16 == 0;
      if (op.kind == K_HEAD) {  // the matrix-core head kernel: <= 64 maps, weights within its 64 KiB of LDS (sa_conv1x1_head)
        const int cp = n->bufs[(size_t)op.a[0]].cp;
        fits = fits && op.a[3] <= 64 && cp % 16 == 0 && (size_t)cp / 16 * 2048 * (op.a[3] <= 32 ? 1 : 2) <= 64 * 1024;
      }
    }
    for (const Out& o : n->outs) fits = fits && (o.is_f32 || n->bufs[(size_t)o.buf].cp == 16);
    if (!fits) {
      delete n;
      return sa::fail(SA_ERR_UNSUPPORTED, "sa_network_create: SA_LAYOUT_PLANES16 plan holds a launch or an output that only exists for NHWC tensors");
    }
  }
  *out = n;
  return SA_OK;
}

void sa_network_destroy(sa_network_t* net) { delete net; }

int sa_network_n_outputs(const sa_network_t* net) { return net ? (int)net->outs.size() : 0; }
int sa_network_in_channels(const sa_network_t* net) { return net ? net->in_channels : 0; }
int sa_network_max_stride(const sa_network_t* net) { return net ? net->max_stride : 0; }
int sa_network_layout(const sa_network_t* net) { return net ? net->layout : 0; }

int sa_network_output_shape(const sa_network_t* net, int index, int H, int W, int* oh, int* ow, int* oc) {
  SA_REQUIRE(net && index >= 0 && index < (int)net->outs.size(), "sa_network_output_shape: bad index");
  const Buf& b = net->bufs[net->outs[index].buf];
  if (oh) *oh = H * b.num / b.den;
  if (ow) *ow = W * b.num / b.den;
  if (oc) *oc = net->outs[index].c;
  return SA_OK;
}

size_t sa_network_workspace_bytes(const sa_network_t* net, int B, int H, int W) {
  if (!net || B <= 0 || H <= 0 || W <= 0) return 0;
  // head outputs live in the caller's `outputs`, not in the workspace, but are laid out all the same: a caller may pass
  // outputs = NULL entries to keep them there (sa_network_buffer)
  return layout(net, B, H, W, nullptr);
}

/* Device address of plan buffer `buf` inside `workspace` for this input shape (diagnostics: the fp16 range scan and the
 * per-layer tests of the Python engine read intermediate tensors through it). */
void* sa_network_buffer(const sa_network_t* net, int buf, int B, int H, int W, void* workspace, int* h, int* w, int* cp, int* is_f32) {
  if (!net || buf < 0 || buf >= (int)net->bufs.size() || net->bufs[buf].kind == 2) return nullptr;
  std::vector<size_t> off;
  layout(net, B, H, W, &off);
  const Buf& b = net->bufs[buf];
  if (h) *h = H * b.num / b.den;
  if (w) *w = W * b.num / b.den;
  if (cp) *cp = b.cp;
  if (is_f32) *is_f32 = b.kind == 1;
  return static_cast<unsigned char*>(workspace) + off[buf];
}

int sa_network_forward(const sa_network_t* net, const void* images, int images_are_u8, int B, int H, int W, int C,
                       float* const* outputs, void* workspace, size_t ws_bytes, sa_stream_t stream) {
  SA_REQUIRE(net && images && outputs && workspace, "sa_network_forward: NULL pointer");
  SA_REQUIRE(B > 0 && H > 0 && W > 0, "sa_network_forward: bad shape");
  SA_REQUIRE(C == net->in_channels, "sa_network_forward: model expects %d input channels, got %d", net->in_channels, C);
  SA_REQUIRE(H % net->max_stride == 0 && W % net->max_stride == 0,
             "sa_network_forward: input size (%d, %d) must be a multiple of the model stride %d", H, W, net->max_stride);
  std::vector<size_t> off;
  const size_t need = layout(net, B, H, W, &off);
  if (ws_bytes < need) return sa::fail(SA_ERR_WORKSPACE, "sa_network_forward: workspace %zu < %zu bytes", ws_bytes, need);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  // f32 head outputs are written straight into the caller's tensors when given
  std::vector<void*> ptr(net->bufs.size(), nullptr);
  for (size_t i = 0; i < net->bufs.size(); ++i)
    if (net->bufs[i].kind != 2) ptr[i] = ws + off[i];
  for (size_t o = 0; o < net->outs.size(); ++o)
    if (net->outs[o].is_f32 && outputs[o]) ptr[net->outs[o].buf] = outputs[o];
  auto bp = [&](int64_t id) -> void* { return id < 0 ? nullptr : ptr[(size_t)id]; };
  auto bh = [&](int64_t id) { const Buf& b = net->bufs[(size_t)id]; return H * b.num / b.den; };
  auto bw = [&](int64_t id) { const Buf& b = net->bufs[(size_t)id]; return W * b.num / b.den; };
  auto bc = [&](int64_t id) { return net->bufs[(size_t)id].cp; };
  const int lay = net->layout;

  for (const Op& op : net->ops) {
    const int64_t* a = op.a.data();
    int rc = SA_OK;
    switch (op.kind) {
      case K_STEM2: {  // [cin, w0, b0, c0p, relu0, w1, b1, coutp, relu1, o_buf, opool_buf, stem16_blob]
        if (images_are_u8 && a[11])
          rc = sa_stem16_u8_bf16(images, B, H, W, (int)a[0], P<void>(a[11]), (int)a[4], (int)a[8], bp(a[9]), bp(a[10]), stream);
        else
          rc = sa_stem_conv3x3x2_bf16(images, images_are_u8, B, H, W, (int)a[0], P<float>(a[1]), P<float>(a[2]), (int)a[3],
                                      (int)a[4], P<void>(a[5]), P<float>(a[6]), (int)a[7], (int)a[8], bp(a[9]), bp(a[10]), lay, stream);
        break;
      }
      case K_STEM:  // [o_buf, w, bias, cin, relu]
        rc = sa_stem_conv3x3(images, images_are_u8, B, H, W, (int)a[3], P<float>(a[1]), P<float>(a[2]), bc(a[0]), (int)a[4],
                             bp(a[0]), stream);
        break;
      case K_CONV: {
        // [s0, s1, mode, w, bias, o_buf, relu, opool_buf, store, n_heads, (hw, hb, hc, hact, hdst_buf) x n_heads,
        //  has_ext, ps, pt, res_buf, res_mode, relu_last]; o_buf always names the output tensor (its size), `store` says
        //  whether it is written
        const int64_t s0 = a[0], s1 = a[1], o = a[5];
        const int nh = (int)a[9];
        const int64_t* e = a + 10 + 5 * nh;
        const int oh = bh(o), ow = bw(o), coutp = bc(o);
        void* dst = a[8] ? bp(o) : nullptr;
        const void* src1 = s1 >= 0 ? bp(s1) : nullptr;
        const int c1p = s1 >= 0 ? bc(s1) : 0;
        if (nh > 0) {
          const float* hw[2];
          const float* hb[2];
          int hc[2], ha[2];
          float* hd[2];
          for (int k = 0; k < nh; ++k) {
            hw[k] = P<float>(a[10 + 5 * k]);
            hb[k] = P<float>(a[11 + 5 * k]);
            hc[k] = (int)a[12 + 5 * k];
            ha[k] = (int)a[13 + 5 * k];
            hd[k] = static_cast<float*>(bp(a[14 + 5 * k]));
          }
          if (e[0])  // extended epilogue (no residual: DeviceNetwork._fuse_heads) + heads
            rc = sa_conv3x3_ex_heads_bf16(bp(s0), bc(s0), src1, c1p, (int)a[2] | lay, P<void>(a[3]), P<float>(a[4]), coutp, (int)a[6], B,
                                          oh, ow, dst, P<float>(e[1]), P<float>(e[2]), (int)e[5], nh, hw, hb, hc, ha, hd, stream);
          else
            rc = sa_conv3x3_heads_bf16(bp(s0), bc(s0), src1, c1p, (int)a[2] | lay, P<void>(a[3]), P<float>(a[4]), coutp, (int)a[6], B, oh,
                                       ow, dst, nh, hw, hb, hc, ha, hd, stream);
        } else if (e[0]) {
          rc = sa_conv3x3_ex_bf16(bp(s0), bc(s0), src1, c1p, (int)a[2] | lay, P<void>(a[3]), P<float>(a[4]), coutp, (int)a[6], B, oh, ow,
                                  dst, bp(a[7]), P<float>(e[1]), P<float>(e[2]), bp(e[3]), (int)e[4], (int)e[5], stream);
        } else {
          rc = sa_conv3x3_bf16(bp(s0), bc(s0), src1, c1p, (int)a[2] | lay, P<void>(a[3]), P<float>(a[4]), coutp, (int)a[6], B, oh, ow,
                               dst, bp(a[7]), stream);
        }
        break;
      }
      case K_BNECK: {
        // [src, w, bias, relu, ps, pt, relu_last | xw, xbias, xrelu, xps, xpt, xres, xrelu_last, x_out | has_y, yw, ybias, yrelu,
        //  yps, ypt, yrelu_last, y_out]  (DeviceNetwork._bneck_words)
        const int64_t xo = a[14];
        rc = sa_conv3x3_bneck_bf16(bp(a[0]), bc(a[0]), lay, P<void>(a[1]), P<float>(a[2]), (int)a[3], P<float>(a[4]), P<float>(a[5]),
                                   (int)a[6], B, bh(xo), bw(xo), P<void>(a[7]), P<float>(a[8]), P<float>(a[10]), P<float>(a[11]),
                                   bp(a[12]), (int)a[9], (int)a[13], bc(xo), bp(xo), a[15] ? P<void>(a[16]) : nullptr,
                                   a[15] ? P<float>(a[17]) : nullptr, a[15] ? P<float>(a[19]) : nullptr,
                                   a[15] ? P<float>(a[20]) : nullptr, (int)a[18], (int)a[21], a[15] ? bc(a[22]) : 0,
                                   a[15] ? bp(a[22]) : nullptr, stream);
        break;
      }
      case K_PAIR: {  // [s0, wa, ba, relu_a, c1p, wb, bb, relu_b, o_buf, store, opool_buf, (mid_buf)]
        // round 6: the 32 -> 64 -> 64 block names its intermediate buffer; a launch with fewer than ~6 tiles per CU runs as the
        // two convolutions through it (one persistent workgroup per CU has too few tiles to amortise its pipeline: 8 frames of
        // 1024^2 0.790 vs 0.780 ms per step; from 16 frames on the fused launch wins -- engine.py:_fuse_pairs)
        const int64_t mid = op.a.size() > 11 ? a[11] : -1;
        static int n_cu = 0;
        if (!n_cu) {
          int dev = 0;
          if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
        }
        const int64_t tiles = (int64_t)B * ((bh(a[8]) + 15) / 16) * ((bw(a[8]) + 31) / 32);
        if (mid >= 0 && tiles < 6 * (int64_t)n_cu) {
          rc = sa_conv3x3_bf16(bp(a[0]), bc(a[0]), nullptr, 0, lay, P<void>(a[1]), P<float>(a[2]), (int)a[4], (int)a[3], B, bh(a[8]), bw(a[8]),
                               bp(mid), nullptr, stream);
          if (rc == SA_OK)
            rc = sa_conv3x3_bf16(bp(mid), (int)a[4], nullptr, 0, lay, P<void>(a[5]), P<float>(a[6]), bc(a[8]), (int)a[7], B, bh(a[8]), bw(a[8]),
                                 a[9] ? bp(a[8]) : nullptr, bp(a[10]), stream);
          break;
        }
        rc = sa_conv3x3_pair_bf16(bp(a[0]), bc(a[0]), P<void>(a[1]), P<float>(a[2]), (int)a[3], (int)a[4], P<void>(a[5]),
                                  P<float>(a[6]), (int)a[7], bc(a[8]), B, bh(a[8]), bw(a[8]), a[9] ? bp(a[8]) : nullptr, bp(a[10]),
                                  lay, stream);
        break;
      }
      case K_CONV1X1:  // [s0, w, bias, relu, stride, has_ext, ps, pt, res_buf, relu_last, o_buf]
        // (stride word: bits 0-7 the stride of a 1x1 conv, bits 8+ the window size k of a stride-1 "same" k x k conv)
        if ((a[4] >> 8) > 1)
          rc = sa_convk_bf16(bp(a[0]), bc(a[0]), P<void>(a[1]), (int)(a[4] >> 8), P<float>(a[2]), bc(a[10]), (int)a[3] | lay, B, bh(a[0]),
                             bw(a[0]), a[5] ? P<float>(a[6]) : nullptr, a[5] ? P<float>(a[7]) : nullptr, bp(a[8]),
                             a[5] ? (int)a[9] : 0, bp(a[10]), stream);
        else
          rc = sa_conv1x1_bf16(bp(a[0]), bc(a[0]), P<void>(a[1]), P<float>(a[2]), bc(a[10]), (int)a[3] | lay, B, bh(a[0]), bw(a[0]),
                               (int)(a[4] & 255), a[5] ? P<float>(a[6]) : nullptr, a[5] ? P<float>(a[7]) : nullptr, bp(a[8]),
                               a[5] ? (int)a[9] : 0, bp(a[10]), stream);
        break;
      case K_CONVT2: {  // [s, w0, w1, w2, w3, ksz, bias, relu, has_ext, ps, pt, relu_last, o_buf]
        const void* wp[4] = {P<void>(a[1]), P<void>(a[2]), P<void>(a[3]), P<void>(a[4])};
        rc = sa_convt_s2_bf16(bp(a[0]), bc(a[0]), wp, (int)a[5], P<float>(a[6]), bc(a[12]), (int)a[7] | lay, B, bh(a[0]), bw(a[0]),
                              a[8] ? P<float>(a[9]) : nullptr, a[8] ? P<float>(a[10]) : nullptr, a[8] ? (int)a[11] : 0, bp(a[12]),
                              stream);
        break;
      }
      case K_CONVT:  // [s, w, bias, relu, o_buf]
        rc = sa_convt3x3s2_bf16(bp(a[0]), bc(a[0]), P<void>(a[1]), P<float>(a[2]), bc(a[4]), (int)a[3], B, bh(a[0]), bw(a[0]),
                                bp(a[4]), stream);
        break;
      case K_POOLG: {  // [s, o, k, stride, pad (-1 = TF SAME), pad_is_zero]
        const int sh = bh(a[0]), sw = bw(a[0]), oh = bh(a[1]), ow = bw(a[1]), k = (int)a[2], st = (int)a[3];
        int pt = (int)a[4], pl = (int)a[4];
        if (a[4] < 0) {  // TF SAME: pad_total = max((out - 1) * s + k - in, 0), pad_before = pad_total / 2
          const int th = (oh - 1) * st + k - sh, tw = (ow - 1) * st + k - sw;
          pt = (th > 0 ? th : 0) / 2;
          pl = (tw > 0 ? tw : 0) / 2;
        }
        // (planes: every 16-channel plane of every frame is a frame of 16 channels)
        if (lay == SA_LAYOUT_PLANES16)
          rc = sa_maxpool_bf16(bp(a[0]), B * (bc(a[0]) / 16), sh, sw, 16, k, st, pt, pl, (int)a[5], oh, ow, bp(a[1]), stream);
        else
          rc = sa_maxpool_bf16(bp(a[0]), B, sh, sw, bc(a[0]), k, st, pt, pl, (int)a[5], oh, ow, bp(a[1]), stream);
        break;
      }
      case K_IMGCONV: {
        // [o, w, bias, src_c, relu, kh, kw, stride, ps, pt, cin_w, in_affine, has_pads, pad_t, pad_l, mf_w, mf_bias, has_mean, mf_cin_w]
        const int oh = bh(a[0]), ow = bw(a[0]), kh = (int)a[5], kw = (int)a[6], st = (int)a[7];
        int pt = (int)a[13], pl = (int)a[14];
        if (!a[12]) {
          const int th = (oh - 1) * st + kh - H, tw = (ow - 1) * st + kw - W;
          pt = (th > 0 ? th : 0) / 2;
          pl = (tw > 0 ? tw : 0) / 2;
        }
        if (images_are_u8 && a[15])
          // (a[18]: weight channels of the packed operand -- 1 for a tiled grayscale frame under a 3-channel kernel; older plans: a[10])
          rc = sa_imgconv_u8_bf16(images, B, H, W, (int)a[3], op.a.size() > 18 ? (int)a[18] : (int)a[10], kh, st, pt, pl, oh, ow, P<void>(a[15]), P<float>(a[16]),
                                  bc(a[0]), (int)a[4] | lay, (int)a[17], P<float>(a[8]), P<float>(a[9]), bp(a[0]), stream);
        else
          rc = sa_image_conv_bf16(images, images_are_u8, B, H, W, (int)a[3], (int)a[10], P<float>(a[11]), kh, kw, st, pt, pl, oh, ow,
                                  P<float>(a[1]), P<float>(a[2]), bc(a[0]), (int)a[4] | lay, P<float>(a[8]), P<float>(a[9]), bp(a[0]),
                                  stream);
        break;
      }
      case K_ADD:  // [a, b, b_half_res, relu, o]
        if (lay == SA_LAYOUT_PLANES16)
          rc = sa_add_bf16(bp(a[0]), bp(a[1]), B * (bc(a[4]) / 16), bh(a[4]), bw(a[4]), 16, (int)a[2], (int)a[3], bp(a[4]), stream);
        else
          rc = sa_add_bf16(bp(a[0]), bp(a[1]), B, bh(a[4]), bw(a[4]), bc(a[4]), (int)a[2], (int)a[3], bp(a[4]), stream);
        break;
      case K_HEAD:  // [s, w, bias, c, act, o]
        rc = sa_conv1x1_head(bp(a[0]), bc(a[0]), P<float>(a[1]), P<float>(a[2]), (int)a[3], (int)a[4] | lay, B, bh(a[0]), bw(a[0]),
                             static_cast<float*>(bp(a[5])), stream);
        break;
      case K_POOL:  // [s, o]
        if (lay == SA_LAYOUT_PLANES16)
          rc = sa_maxpool2x2_bf16(bp(a[0]), B * (bc(a[0]) / 16), bh(a[0]), bw(a[0]), 16, bp(a[1]), stream);
        else
          rc = sa_maxpool2x2_bf16(bp(a[0]), B, bh(a[0]), bw(a[0]), bc(a[0]), bp(a[1]), stream);
        break;
      case K_UP:  // [s, o, bilinear]; planes: every 16-channel plane of every frame is a frame of 16 channels
        if (lay == SA_LAYOUT_PLANES16)
          rc = sa_upsample2x_bf16(bp(a[0]), B * (bc(a[0]) / 16), bh(a[0]), bw(a[0]), 16, (int)a[2], bp(a[1]), stream);
        else
          rc = sa_upsample2x_bf16(bp(a[0]), B, bh(a[0]), bw(a[0]), bc(a[0]), (int)a[2], bp(a[1]), stream);
        break;
      default:
        return sa::fail(SA_ERR_INVALID_ARG, "sa_network_forward: unknown op kind %lld", (long long)op.kind);
    }
    if (rc != SA_OK) return rc;
  }
  // 16-bit feature tensors that are model outputs: float32 copies for the caller
  for (size_t o = 0; o < net->outs.size(); ++o) {
    const Out& out = net->outs[o];
    if (out.is_f32 || !outputs[o]) continue;
    const int rc = sa_bf16_to_f32(ptr[out.buf], B * bh(out.buf) * bw(out.buf), bc(out.buf), out.c, outputs[o], stream);
    if (rc != SA_OK) return rc;
  }
  return SA_OK;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * BottomUpInferenceLayer.call for one batch (inference.py:2938-3003): network forward -> find_peaks (local peaks of the
 * confidence maps, refinement, x cm_output_stride) -> PAFScorer.predict (score, match, group), all on `stream`, nothing
 * allocated, nothing synchronised. Scratch for the post-processing stage is carved from the same workspace.
 * ------------------------------------------------------------------------------------------------------------------- */
size_t sa_bottomup_workspace_bytes(const sa_network_t* net, const sa_bottomup_params* q, int B, int H, int W) {
  if (!net || !q) return 0;
  size_t t = align256(sa_network_workspace_bytes(net, B, H, W));
  int oh = 0, ow = 0, oc = 0;
  for (int i = 0; i < (int)net->outs.size(); ++i) {  // the head tensors themselves
    sa_network_output_shape(net, i, H, W, &oh, &ow, &oc);
    t += align256((size_t)B * oh * ow * oc * 4);
  }
  const int N = q->n_nodes, E = q->n_edges, NP = q->max_node_peaks, MP = q->max_peaks;
  t += align256(sa_bottomup_postproc_workspace(B, MP, E, N, NP));
  t += align256((size_t)B * MP * 2 * 4) + 2 * align256((size_t)B * MP * 4) + align256((size_t)B * 4);  // peak_xy, val, chan, count
  t += align256((size_t)B * N * 4) + align256((size_t)B * N * NP * 4) + align256((size_t)B * E * NP * NP * 4);  // node tables, line scores
  t += 2 * align256((size_t)B * E * NP * 4);                                                                    // match_dst / score
  return t;
}

int sa_bottomup_predict(const sa_network_t* net, const sa_bottomup_params* q, const void* images, int images_are_u8, int B, int H,
                        int W, int C, float* instance_peaks, float* instance_peak_vals, float* instance_scores,
                        int32_t* n_instances, int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream) {
  SA_REQUIRE(net && q && images && instance_peaks && instance_peak_vals && instance_scores && n_instances && status && workspace,
             "sa_bottomup_predict: NULL pointer");
  const int n_out = (int)net->outs.size();
  SA_REQUIRE(q->confmaps_ind >= 0 && q->confmaps_ind < n_out && q->pafs_ind >= 0 && q->pafs_ind < n_out &&
                 q->offsets_ind < n_out, "sa_bottomup_predict: head indices out of range");
  SA_REQUIRE(q->edges && q->sorted_edge_inds && q->n_nodes > 0 && q->n_edges >= 0, "sa_bottomup_predict: skeleton tables missing");
  const size_t need = sa_bottomup_workspace_bytes(net, q, B, H, W);
  if (ws_bytes < need) return sa::fail(SA_ERR_WORKSPACE, "sa_bottomup_predict: workspace %zu < %zu bytes", ws_bytes, need);
  unsigned char* p = static_cast<unsigned char*>(workspace);
  auto take = [&](size_t bytes) {
    void* r = p;
    p += align256(bytes);
    return r;
  };
  void* net_ws = take(sa_network_workspace_bytes(net, B, H, W));
  std::vector<float*> heads((size_t)n_out);
  std::vector<int> oh((size_t)n_out), ow((size_t)n_out), oc((size_t)n_out);
  for (int i = 0; i < n_out; ++i) {
    sa_network_output_shape(net, i, H, W, &oh[i], &ow[i], &oc[i]);
    heads[(size_t)i] = static_cast<float*>(take((size_t)B * oh[i] * ow[i] * oc[i] * 4));
  }
  const int N = q->n_nodes, E = q->n_edges, NP = q->max_node_peaks, MP = q->max_peaks;
  const size_t pp_ws_bytes = sa_bottomup_postproc_workspace(B, MP, E, N, NP);
  void* pp_ws = take(pp_ws_bytes);
  float* peak_xy = static_cast<float*>(take((size_t)B * MP * 2 * 4));
  float* peak_val = static_cast<float*>(take((size_t)B * MP * 4));
  int32_t* peak_chan = static_cast<int32_t*>(take((size_t)B * MP * 4));
  int32_t* peak_count = static_cast<int32_t*>(take((size_t)B * 4));
  int32_t* node_count = static_cast<int32_t*>(take((size_t)B * N * 4));
  int32_t* node_peaks = static_cast<int32_t*>(take((size_t)B * N * NP * 4));
  float* line_scores = static_cast<float*>(take((size_t)B * E * NP * NP * 4));
  int32_t* match_dst = static_cast<int32_t*>(take((size_t)B * E * NP * 4));
  float* match_score = static_cast<float*>(take((size_t)B * E * NP * 4));
  hipStream_t st = (hipStream_t)stream;
  SA_HIP_CHECK(hipMemsetAsync(status, 0, (size_t)B * 4, st));

  int rc = sa_network_forward(net, images, images_are_u8, B, H, W, C, heads.data(), net_ws, sa_network_workspace_bytes(net, B, H, W),
                              stream);
  if (rc != SA_OK) return rc;
  const int ci = q->confmaps_ind, pi = q->pafs_ind, oi = q->offsets_ind;
  SA_REQUIRE(oc[ci] == N, "sa_bottomup_predict: confidence maps have %d channels, skeleton has %d nodes", oc[ci], N);
  SA_REQUIRE(oc[pi] == 2 * E, "sa_bottomup_predict: PAFs have %d channels, skeleton has %d edges", oc[pi], E);
  const int refinement = oi >= 0 ? SA_REFINE_OFFSETS : q->refinement;
  // PAFScorer.max_edge_length (paf_grouping.py:469-473): ratio * max over (H, W, 2E) of the PAF tensor * stride, f32 products
  int mx = oh[pi] > ow[pi] ? oh[pi] : ow[pi];
  if (2 * E > mx) mx = 2 * E;
  const float max_edge_length = q->max_edge_length_ratio * (float)mx * q->pafs_stride;
  return sa_bottomup_postproc(heads[ci], oi >= 0 ? heads[oi] : nullptr, B, oh[ci], ow[ci], N, q->peak_threshold, refinement,
                              q->integral_patch_size, q->cm_output_stride, MP, heads[pi], oh[pi], ow[pi], E, q->edges,
                              q->sorted_edge_inds, q->n_sorted, N, q->n_points, q->pafs_stride, max_edge_length,
                              q->dist_penalty_weight, NP, q->min_line_scores, q->min_instance_peaks, q->max_instances, peak_xy,
                              peak_val, peak_chan, peak_count, node_count, node_peaks, line_scores, match_dst, match_score,
                              instance_peaks, instance_peak_vals, instance_scores, n_instances, status, pp_ws, pp_ws_bytes, stream);
}

}  // extern "C"
