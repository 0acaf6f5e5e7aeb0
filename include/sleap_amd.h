/* sleap_amd.h -- C ABI of libsleap_amd.so: the MI355X (gfx950) kernels behind the SLEAP
 * bottom-up inference hot path.
 *
 * The reference (talmolab/sleap v1.4.1) has no FFI for this path: it is pure Python on
 * TensorFlow. Each entry point below therefore names the reference *Python* function (or the
 * TensorFlow op that function lowers to) whose arithmetic it replaces; the Python host layer in
 * `sleap_amd/nn/` mirrors the reference's signatures and binds these symbols with `ctypes`
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - every function returns 0 on success, a negative `SA_ERR_*` code otherwise;
 *     `sa_last_error()` returns a thread-local message for the last failure
 *   - all pointers except those marked HOST are DEVICE pointers owned by the caller; nothing is
 *     allocated behind the caller's back
 *   - `stream` is a `hipStream_t` passed as `void*`; all work is enqueued asynchronously
 *   - image-like tensors are NHWC; activations inside the network are 16-bit with the channel count
 *     padded to a multiple of 16 ("CP" below), network heads / maps are float32 with exact channels
 *   - the 16-bit storage type of activations and packed conv weights is a BUILD property of the
 *     library: libsleap_amd_fp16.so stores IEEE half (the default build the Python layer loads),
 *     libsleap_amd.so stores bfloat16; `sa_storage_dtype()` names it. Both export this same ABI;
 *     the entry points keep their historical `_bf16` suffix and "bf16" in the comments below
 *     means "the build's storage type". Accumulation is float32 in both.
 */
#ifndef SLEAP_AMD_H
#define SLEAP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_ABI_VERSION 8  /* 8 (round 6): + sa_h264_decode_slice. 7 (round 6): + sa_find_local_peaks_rough; sa_conv3x3_pair_bf16 accepts 32 -> 64 -> 64. 6 (round 5): + sa_conv3x3_set_persistent. 5 (round 4): + sa_conv3x3_ex_heads_bf16. 4 (round 4): + sa_tensor_absmax, sa_imgconv_pack_tiled, sa_pack_pointwise_weights,
                            sa_pointwise_packed_elems, sa_conv3x3_bneck_bf16; SA_LAYOUT_PLANES16 accepted in the `relu` argument of
                            sa_conv1x1_bf16 / sa_convk_bf16 / sa_convt_s2_bf16 */

#define SA_OK 0
#define SA_ERR_INVALID_ARG (-1)
#define SA_ERR_HIP (-2)
#define SA_ERR_UNSUPPORTED (-3)
#define SA_ERR_WORKSPACE (-4)

/* per-frame status bits written by the post-processing kernels into `status[b]` */
#define SA_STATUS_PEAK_OVERFLOW 1      /* more local peaks than `max_peaks` */
#define SA_STATUS_NODE_PEAK_OVERFLOW 2 /* more peaks of one node type than `max_node_peaks` */
#define SA_STATUS_INSTANCE_OVERFLOW 4  /* more instances than `max_instances` */
#define SA_STATUS_LSA_INFEASIBLE 8     /* scipy would raise "cost matrix is infeasible" */
#define SA_STATUS_PAF_OOB 16           /* a PAF line sample fell outside the PAF tensor (TF-CPU raises) */
#define SA_STATUS_NONFINITE 32         /* the confidence maps hold inf / NaN (an fp16-storage network overflowed) */

typedef void* sa_stream_t;

int sa_abi_version(void);
/* "fp16" or "bf16": the 16-bit storage type this build of the library was compiled for. */
const char* sa_storage_dtype(void);
const char* sa_last_error(void);
/* HOST out-params. `arch` receives e.g. "gfx950". */
int sa_device_info(int device, int* n_cu, int* lds_bytes, int* wave_size, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Peak finding -- replaces sleap/nn/peak_finding.py
 * ---------------------------------------------------------------------------------------------- */

#define SA_REFINE_NONE 0
#define SA_REFINE_INTEGRAL 1 /* find_local_peaks(refinement="integral")  peak_finding.py:451-532 */
#define SA_REFINE_LOCAL 2    /* find_local_peaks(refinement="local")     peak_finding.py:78-132 */
#define SA_REFINE_OFFSETS 3  /* find_local_peaks_with_offsets            peak_finding.py:646-707 */

size_t sa_find_local_peaks_workspace(int B, int max_peaks);

/* find_local_peaks_rough (peak_finding.py:249-308: tf.nn.dilation2d NMS + tf.where + gather_nd)
 * followed by the refinement of find_local_peaks / find_local_peaks_with_offsets
 * (crop_bboxes -> tf.image.crop_and_resize, integral_regression / find_offsets_local_direction).
 *
 *   cms      [B,H,W,C] f32          offsets [B,H,W,2C] f32 or NULL
 *   peak_xy  [B,max_peaks,2] f32  = (rough + offset) * xy_scale, order row-major (y,x,c) per frame
 *   peak_val [B,max_peaks] f32      peak_chan [B,max_peaks] i32      peak_count [B] i32
 *   status   [B] i32 (bits OR-ed in; caller zeroes)
 * `xy_scale` = 1 reproduces the module function (grid units); BottomUpInferenceLayer.find_peaks
 * (inference.py:2892-2936) passes cm_output_stride. */
int sa_find_local_peaks(const float* cms, const float* offsets, int B, int H, int W, int C,
                        float threshold, int refinement, int patch_size, float xy_scale,
                        int max_peaks, float* peak_xy, float* peak_val, int32_t* peak_chan,
                        int32_t* peak_count, int32_t* status, void* workspace, size_t ws_bytes,
                        sa_stream_t stream);

/* find_local_peaks_rough alone (peak_finding.py:249-308): the NMS scan of the confidence maps -- the one pass of the
 * post-processing that reads every map value (H W C x 4 bytes per frame: the HBM-bound kernel bench.py's `roofline_postproc`
 * is quoted on). keys [B,max_peaks] u32 = the linear (y, x, c) index of every local maximum above `threshold`, in arrival
 * order (sa_find_local_peaks sorts them: row-major order); peak_count [B] i32 (may exceed max_peaks: status bit
 * SA_STATUS_PEAK_OVERFLOW); status [B] i32 (bits OR-ed in, the caller zeroes). */
int sa_find_local_peaks_rough(const float* cms, int B, int H, int W, int C, float threshold, int max_peaks, uint32_t* keys,
                              int32_t* peak_count, int32_t* status, sa_stream_t stream);

/* find_global_peaks_rough + find_global_peaks / find_global_peaks_with_offsets
 * (peak_finding.py:193-246, 337-420, 566-643). Row and column argmax are taken independently.
 *   peak_xy [B,C,2] f32 (NaN below threshold), peak_val [B,C] f32 */
int sa_find_global_peaks(const float* cms, const float* offsets, int B, int H, int W, int C,
                         float threshold, int refinement, int patch_size, float xy_scale,
                         float* peak_xy, float* peak_val, sa_stream_t stream);

/* crop_bboxes (peak_finding.py:135-190) for the top-down path (CentroidCrop, inference.py:1919-1929):
 * tf.image.crop_and_resize(bilinear, extrapolation 0) of `crop` x `crop` boxes centred on fractional (x, y)
 * centres (make_centered_bboxes, instance_cropping.py:124-166), cast back to the image dtype.
 *   images [B,H,W,C] u8|f32; centres_xy [n,2] f32; sample_inds [n] i32; out [n,crop,crop,C] same dtype */
int sa_crop_and_resize(const void* images, int is_u8, int H, int W, int C, const float* centres_xy,
                       const int32_t* sample_inds, int n, int crop, void* out, sa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * PAF grouping -- replaces sleap/nn/paf_grouping.py
 * ---------------------------------------------------------------------------------------------- */

/* get_connection_candidates + make_line_subs + get_paf_lines + score_paf_lines
 * (paf_grouping.py:82-142, 145-275, 325-403; batch loop :406-550).
 *
 *   pafs [B,Hp,Wp,2E] f32;  peaks as produced by sa_find_local_peaks (image pixels)
 *   edges [E,2] i32 (src node, dst node)
 *   node_count [B,N] i32; node_peaks [B,N,max_node_peaks] i32 (index into the frame's peak list)
 *   line_scores [B,E,max_node_peaks,max_node_peaks] f32 -- entry [k][s][d] is the candidate
 *     (s-th peak of edge k's src node, d-th peak of its dst node); unused entries untouched
 * `max_edge_length` is max_edge_length_ratio * max(Hp,Wp,2E) * pafs_stride, computed by the host
 * exactly as paf_grouping.py:469-473. */
int sa_paf_score(const float* pafs, int B, int Hp, int Wp, int E, const float* peak_xy,
                 const int32_t* peak_chan, const int32_t* peak_count, int max_peaks,
                 const int32_t* edges, int N, int n_points, float pafs_stride,
                 float max_edge_length, float dist_penalty_weight, int max_node_peaks,
                 int32_t* node_count, int32_t* node_peaks, float* line_scores, int32_t* status,
                 sa_stream_t stream);

/* match_candidates_sample (paf_grouping.py:553-670) with the Hungarian solve of
 * sleap/nn/utils.py:79-98 (scipy.optimize.linear_sum_assignment, restated in csrc/lsa.h).
 *   match_dst [B,E,max_node_peaks] i32: for src peak s of edge k the matched dst peak or -1
 *   match_score [B,E,max_node_peaks] f32 */
int sa_paf_match(const float* line_scores, const int32_t* node_count, const int32_t* edges, int B,
                 int E, int N, int max_node_peaks, int32_t* match_dst, float* match_score,
                 int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream);

/* Bytes of device workspace sa_paf_match / sa_paf_group need (one buffer may serve both, sequentially). */
/* The pieces of the scoring stage the reference exposes (and tests) as module-level functions; the hot path runs the fused
 * sa_paf_score, which shares their arithmetic. All per SAMPLE, flat candidate lists (K candidates).
 *   sa_paf_line_subs      make_line_subs (paf_grouping.py:145-222): peaks_xy [n_peaks,2] f32, edge_peak_inds [K,2] i32,
 *                         edge_inds [K] i32 -> subs [K,n_points,2,3] i32 = [row, col, 2*edge + {0,1}] (linspace, round half to
 *                         even, no clipping -- as the reference)
 *   sa_gather_nd3         tf.gather_nd(pafs_sample [H,W,C] f32, subs [n,3]) of get_paf_lines (:225-275) -> out [n];
 *                         out-of-range subscripts read 0 and OR SA_STATUS_PAF_OOB into *status (may be NULL)
 *   sa_paf_line_scores    score_paf_lines (:325-403): paf_lines [K,n_points,2] -> line_scores [K]
 *   sa_distance_penalty   compute_distance_penalty (:278-322), elementwise over n lengths */
int sa_paf_line_subs(const float* peaks_xy, const int32_t* edge_peak_inds, const int32_t* edge_inds, int K, int n_points,
                     float pafs_stride, int32_t* subs, sa_stream_t stream);
int sa_gather_nd3(const float* src, int H, int W, int C, const int32_t* subs, int n, float* out, int32_t* status,
                  sa_stream_t stream);
int sa_paf_line_scores(const float* paf_lines, const float* peaks_xy, const int32_t* edge_peak_inds, int K, int n_points,
                       float max_edge_length, float dist_penalty_weight, float* line_scores, sa_stream_t stream);
int sa_distance_penalty(const float* lengths, int n, float max_edge_length, float dist_penalty_weight, float* out,
                        sa_stream_t stream);

/* group_instances with the connections given as an explicit, ordered list instead of match tables -- the dictionary form
 * of assign_connections_to_instances / make_predicted_instances (paf_grouping.py:799-981; the reference walks
 * `connections.items()` in insertion order, any number of connections per edge type):
 *   conn_edge/src/dst [B,conn_stride] i32, conn_score [B,conn_stride] f32, conn_count [B] i32: connection q of frame b joins
 *   peak `src` of node edges[edge][0] with peak `dst` of node edges[edge][1] (indices within the node's peak list)
 *   assign_out [B,N,max_node_peaks] i32 or NULL: the raw instance id of every (node, peak) slot (-1 = unassigned, < -1 =
 *   dropped by min_instance_peaks) = the reference's `instance_assignments` dictionary
 *   assign_in / order_in [B,N,max_node_peaks] i32 + order_count [B] i32, or all NULL: make_predicted_instances on GIVEN
 *   assignments -- the greedy walk is skipped, `assign_in` is the slot -> instance id table and `order_in` lists the assigned
 *   slots (node * max_node_peaks + peak) in the dictionary's key order (later keys overwrite earlier ones in a cell)
 * Other arguments as sa_paf_group. */
int sa_paf_group_connections(const float* peak_xy, const float* peak_val, const int32_t* node_count, const int32_t* node_peaks,
                             int max_peaks, const int32_t* conn_edge, const int32_t* conn_src, const int32_t* conn_dst,
                             const float* conn_score, const int32_t* conn_count, int conn_stride, const int32_t* edges, int B,
                             int E, int N, int max_node_peaks, float min_line_scores, int min_instance_peaks, int max_instances,
                             float* instance_peaks, float* instance_peak_vals, float* instance_scores, int32_t* n_instances,
                             int32_t* assign_out, const int32_t* assign_in, const int32_t* order_in, const int32_t* order_count,
                             int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream);

/* find_peaks + PAFScorer.predict of BottomUpInferenceLayer.call (inference.py:2892-2936, paf_grouping.py:1629-1705) in TWO
 * launches: the NMS scan over all confidence maps, then one workgroup per frame that sorts + refines its peaks, scores the
 * candidate connections, matches every edge (one wavefront per edge) and assembles the instances (sa_find_local_peaks +
 * sa_paf_score + sa_paf_match + sa_paf_group fused; same arithmetic, same outputs, same status bits). All the intermediate
 * tables are outputs (caller-owned, shapes as in the separate entry points). Capacities beyond one workgroup's LDS fall back
 * to the separate kernels inside the call. workspace: sa_bottomup_postproc_workspace bytes, ZEROED ONCE by the caller after
 * allocation (it starts with the NMS scan's per-frame counters, which every call hands back zeroed -- no memset launch in front
 * of the scan); keep one workspace per in-flight call. */
size_t sa_bottomup_postproc_workspace(int B, int max_peaks, int E, int N, int max_node_peaks);
int sa_bottomup_postproc(const float* cms, const float* offsets, int B, int H, int W, int C, float threshold, int refinement,
                         int patch_size, float xy_scale, int max_peaks, const float* pafs, int Hp, int Wp, int E,
                         const int32_t* edges, const int32_t* sorted_edge_inds, int n_sorted, int N, int n_points,
                         float pafs_stride, float max_edge_length, float dist_penalty_weight, int max_node_peaks,
                         float min_line_scores, int min_instance_peaks, int max_instances, float* peak_xy, float* peak_val,
                         int32_t* peak_chan, int32_t* peak_count, int32_t* node_count, int32_t* node_peaks, float* line_scores,
                         int32_t* match_dst, float* match_score, float* instance_peaks, float* instance_peak_vals,
                         float* instance_scores, int32_t* n_instances, int32_t* status, void* workspace, size_t ws_bytes,
                         sa_stream_t stream);

size_t sa_paf_workspace(int B, int E, int N, int max_node_peaks);

/* group_instances_sample = assign_connections_to_instances + make_predicted_instances
 * (paf_grouping.py:984-1112, 799-914, 917-981).
 *   sorted_edge_inds [n_sorted] i32: PAFScorer.sorted_edge_inds (toposort_edges :1293-1315)
 *   instance_peaks [B,max_instances,N,2] f32 NaN-padded; instance_peak_vals [B,max_instances,N];
 *   instance_scores [B,max_instances]; n_instances [B] i32 */
int sa_paf_group(const float* peak_xy, const float* peak_val, const int32_t* node_count,
                 const int32_t* node_peaks, int max_peaks, const int32_t* match_dst,
                 const float* match_score, const int32_t* edges, const int32_t* sorted_edge_inds,
                 int n_sorted, int B, int E, int N, int max_node_peaks, float min_line_scores,
                 int min_instance_peaks, int max_instances, float* instance_peaks,
                 float* instance_peak_vals, float* instance_scores, int32_t* n_instances,
                 int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream);

/* HOST-side Hungarian solve (same code as the device path; for tests / tooling).
 * cost [nr,nc] f64 row-major; row_ind/col_ind sized min(nr,nc). Returns #pairs or -1 if infeasible. */
int sa_lsa_host(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind);
/* Same contract; runs the wave-cooperative form of the solver that the matching kernel uses (one wavefront per (frame, edge):
 * column scan, dual update and initialisation spread over 64 lanes, SciPy's tie rule kept by an order-aware reduction) with
 * its lanes emulated on the host -- the CPU-side check of that code path against SciPy. */
int sa_lsa_host_wave(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind);

/* ------------------------------------------------------------------------------------------------
 * Network layers -- replace the TensorFlow/cuDNN ops behind the Keras graph built by
 * sleap/nn/architectures/{encoder_decoder,unet,hourglass}.py and sleap/nn/heads.py
 * ---------------------------------------------------------------------------------------------- */

/* InferenceLayer.preprocess (inference.py:940-967: ensure_float = x * 1/255) fused with the first
 * Conv2D(k3, same)+bias+ReLU of the encoder (encoder_decoder.py:117-131) for Cin in {1,3}.
 *   src [B,H,W,Cin] u8 (src_is_u8=1, scaled by 1/255) or f32;  w [3][3][Cin][CoutP] f32;  bias [CoutP] f32
 *   dst [B,H,W,CoutP] bf16 */
int sa_stem_conv3x3(const void* src, int src_is_u8, int B, int H, int W, int Cin, const float* w,
                    const float* bias, int CoutP, int relu, void* dst, sa_stream_t stream);

/* The first TWO layers of the encoder in one kernel: sa_stem_conv3x3 computed on the VALU straight into the LDS
 * tile of the following Conv2D(k3)+bias+ReLU (MFMA), whose output is stored at full resolution and/or max-pooled
 * (encoder_decoder.py:109-131: conv, conv, then the next block's MaxPool2D). The full-resolution activation of
 * the first conv (B*H*W*C0P bf16) never touches HBM.
 *   src [B,H,W,Cin] u8|f32; w0 [3][3][Cin][C0P] f32, bias0 [C0P] f32 (C0P in {16,32});
 *   w1 packed (sa_pack_conv3x3_weights with C0 = first conv's channels), bias1 [CoutP] f32, CoutP <= 64 */
int sa_stem_conv3x3x2_bf16(const void* src, int src_is_u8, int B, int H, int W, int Cin, const float* w0,
                           const float* bias0, int C0P, int relu0, const void* w1, const float* bias1, int CoutP,
                           int relu1, void* dst, void* dst_pool, int layout, sa_stream_t stream);

/* Specialisation of sa_stem_conv3x3x2_bf16 for uint8 images and <= 16 channels in both convs (the default
 * `filters: 16` of SLEAP's UNet profiles) on v_mfma_f32_16x16x32_bf16, all weights register resident.
 *   blob: device copy of the buffer filled by sa_stem16_pack (HOST helper: Keras conv0 (3,3,Cin,C0) and conv1
 *   (3,3,C0,C1) f32 kernels + biases -> per-lane MFMA fragments; sa_stem16_blob_bytes() bytes). */
int sa_stem16_u8_bf16(const void* src, int B, int H, int W, int Cin, const void* blob, int relu0, int relu1,
                      void* dst, void* dst_pool, sa_stream_t stream);
int sa_stem16_pack(const float* k0, const float* b0, int Cin, int C0, const float* k1, const float* b1, int C1,
                   void* blob);
size_t sa_stem16_blob_bytes(void);

#define SA_SRC1_NONE 0
#define SA_SRC1_DIRECT 1     /* Concatenate([src0, src1]) (encoder_decoder.py:360-362) */
#define SA_SRC1_UPSAMPLE2X 2 /* Concatenate([src0, UpSampling2D(2, bilinear)(src1)]) (:335-339) */
#define SA_SRC0_POOL2X 4     /* flag OR-ed in: src0 is read through MaxPool2D(2, s2) (:109-114) */

/* Layout of the 16-bit activation tensors (the `layout` argument of the fused-block entry points; OR-ed into `mode` for the
 * sa_conv3x3_* family, where it applies to src0, src1, dst and dst_pool alike):
 *   SA_LAYOUT_NHWC      [B,H,W,CP]                                              (default)
 *   SA_LAYOUT_PLANES16  [B,CP/16,H,W,16]: 16-channel planes. The 3x3 kernels consume the input channels in chunks of 16; with
 *     planes a chunk of a halo-tile row is ONE contiguous run (34 pixels x 32 bytes) instead of 34 slices of 32 bytes at a
 *     pixel stride of 2 CP bytes -- every fetched cache line is used completely, once -- and an epilogue store instruction
 *     writes 1 KiB contiguously. Measured on MI355X (profiles/r02_planar_ab.md): every layer of the benchmark UNet 6-19 %
 *     faster, bitwise the same values. A tensor with CP = 16 is the same bytes in both layouts. Supported by
 *     sa_conv3x3_bf16 / _heads_bf16 (modes NONE / DIRECT, 16-channel chunks), sa_conv3x3_pair_bf16, sa_stem_conv3x3x2_bf16,
 *     sa_stem16_u8_bf16 (16 channels: nothing to do) and sa_upsample2x_bf16 (call it with B*CP/16 frames of 16 channels);
 *     a network plan (sa_network_create) is in ONE layout throughout. */
#define SA_LAYOUT_NHWC 0
#define SA_LAYOUT_PLANES16 0x100

/* Conv2D(k3, s1, same) + bias + optional ReLU on bf16 NHWC activations as an implicit GEMM on
 * MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
 *   src0 [B,H0,W0,C0P] bf16 (H0=H, or 2H when SA_SRC0_POOL2X), src1 [B,H1,W1,C1P] bf16 or NULL
 *   w packed by sa_pack_conv3x3_weights: [ceil(CoutP/32)][(C0P+C1P)/16][9 taps][64 lanes][8] bf16 (MFMA
 *   A-fragment order);  bias [CoutP] f32
 *   dst [B,H,W,CoutP] bf16 and/or dst_pool [B,H/2,W/2,CoutP] bf16 = MaxPool2D(2,s2) of the same output fused
 *   into the epilogue (either may be NULL; dst_pool only with mode NONE/DIRECT, or SA_SRC1_UPSAMPLE2X on planes)
 *   SA_SRC1_UPSAMPLE2X: src1 is [B,H/2,W/2,C1P] and enters through UpSampling2D(2, bilinear), the values a stored upsampled
 *   tensor would hold (bitwise). With SA_LAYOUT_PLANES16 the half-resolution tile is copied by the DMA pipeline and expanded
 *   in LDS; on NHWC tensors the register-staged first-generation kernel does it (no pooled output, no fused heads there). */
int sa_conv3x3_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                    const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, void* dst_pool,
                    sa_stream_t stream);

/* One encoder block in one launch: Conv2D(k3, 16 -> 32)+bias+ReLU -> Conv2D(k3, 32 -> 32)+bias+ReLU [-> dst] [-> MaxPool2D(2)
 * -> dst_pool] (encoder_decoder.py:109-131). The 32-channel intermediate lives only in LDS (bf16, the rounding a stored tensor
 * would get); results are bitwise those of two sa_conv3x3_bf16 calls. wa / wb: sa_pack_conv3x3_weights of the two kernels.
 * Implemented: C0P = 16, C1P = C2P = 32 (block 1; csrc/convpair.hip) and, round 6, C0P = 32, C1P = C2P = 64 (block 2;
 * csrc/convpair64.hip: one persistent 158.5-KiB workgroup per CU, src / dst / dst_pool in `layout`); SA_ERR_UNSUPPORTED otherwise. */
int sa_conv3x3_pair_bf16(const void* src, int C0P, const void* wa, const float* bias_a, int relu_a, int C1P,
                         const void* wb, const float* bias_b, int relu_b, int C2P, int B, int H, int W, void* dst,
                         void* dst_pool, int layout, sa_stream_t stream);

/* Same convolution with the extended epilogue the hourglass / ResNet graphs need:
 *   v = acc + bias; if relu: v = max(v, 0);
 *   if post_scale: v = v * post_scale[c] + post_shift[c]     (BatchNormalization AFTER the activation, hourglass.py:36-45)
 *   if residual:   v += residual[pixel][c]                   (Add layer; res_mode 1: residual is [B,H/2,W/2,CoutP] and is
 *                                                             read with UpSampling2D(2,"nearest"), hourglass.py:183-191)
 *   if relu_last:  v = max(v, 0)                             (ResNet: relu(bn(conv) + shortcut), resnet.py:168-253)
 * post_scale/post_shift [CoutP] f32 or NULL, residual bf16 or NULL. mode NONE/DIRECT only. */
int sa_conv3x3_ex_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                       const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, void* dst_pool,
                       const float* post_scale, const float* post_shift, const void* residual, int res_mode,
                       int relu_last, sa_stream_t stream);

/* ResNet bottleneck tail in ONE launch (resnet.py:168-253: ... -> Conv2D(k3, 64) + BN + ReLU -> Conv2D(k1, 4 x 64) + BN ->
 * Add(shortcut) -> ReLU, and the next block's Conv2D(k1, -> 64) + BN + ReLU). The 3x3 conv (src [B,H,W,CinP] -> 64 padded maps,
 * `w` / `bias` as for sa_conv3x3_bf16, epilogue relu / post_scale / post_shift / relu_last as sa_conv3x3_ex_bf16) never stores
 * its activation: it is the B operand of the EXPAND stage
 *     y = act_last(affine(act(W2 h + xp_bias)) + xp_res)          -> xp_dst [B,H,W,CoutX], CoutX % 32 == 0, <= 256
 * and y, rounded as stored, is the B operand of the optional REDUCE stage (rd_w != NULL)
 *     z = act_last(affine(act(W1' y + rd_bias)))                  -> rd_dst [B,H,W,64]
 * xp_w / rd_w: sa_pack_pointwise_weights of the Keras 1x1 kernels. `layout`: SA_LAYOUT_NHWC or SA_LAYOUT_PLANES16 for all
 * tensors. Same operations per value as the un-fused sa_conv3x3_ex_bf16 / sa_conv1x1_bf16 launches; the matrix cores add a
 * k-step's 16 products in a different order, so results agree to float32 rounding (<= 1 ulp of the storage type), not bitwise. */
size_t sa_pointwise_packed_elems(int CinP, int CoutP);
int sa_pack_pointwise_weights(const float* w, int Cin, int CinP, int Cout, int CoutP, void* packed);
int sa_conv3x3_bneck_bf16(const void* src, int CinP, int layout, const void* w, const float* bias, int relu, const float* post_scale,
                          const float* post_shift, int relu_last, int B, int H, int W, const void* xp_w, const float* xp_bias,
                          const float* xp_scale, const float* xp_shift, const void* xp_res, int xp_relu, int xp_relu_last,
                          int CoutX, void* xp_dst, const void* rd_w, const float* rd_bias, const float* rd_scale,
                          const float* rd_shift, int rd_relu, int rd_relu_last, int CoutR, void* rd_dst, sa_stream_t stream);

/* Same convolution with up to two 1x1 heads (Head.make_head, heads.py:42-62) fused into the epilogue and
 * computed from the fp32 accumulators (no bf16 rounding of the features the heads see). Needs CoutP <= 64 and
 * mode NONE/DIRECT. HOST arrays of length n_heads: head_w[i] -> device [head_c[i]][CoutP] f32, head_b[i] -> device
 * [head_c[i]] f32, head_c[i] <= 64 (32 per pass of the matrix-core head GEMM), head_act[i] (0 linear, 1 sigmoid), head_dst[i] -> device [B,H,W,head_c[i]] f32.
 * dst (bf16 features) may be NULL when only the heads consume this layer. */
int sa_conv3x3_heads_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w,
                          const float* bias, int CoutP, int relu, int B, int H, int W, void* dst, int n_heads,
                          const float* const* head_w, const float* const* head_b, const int* head_c,
                          const int* head_act, float* const* head_dst, sa_stream_t stream);

/* sa_conv3x3_heads_bf16 behind the extended epilogue (sa_conv3x3_ex_bf16 without a residual or a pooled output): the ResNet
 * decoder's Conv2D + BatchNormalization + ReLU in front of its heads (upsampling.py:140-199 refine convs -> heads.py:42-62).
 * The heads see the value the extended epilogue stores. Needs 33..64 padded output channels; mode NONE/DIRECT.
 * dst may be NULL when only the heads consume this layer. */
int sa_conv3x3_ex_heads_bf16(const void* src0, int C0P, const void* src1, int C1P, int mode, const void* w, const float* bias,
                             int CoutP, int relu, int B, int H, int W, void* dst, const float* post_scale, const float* post_shift,
                             int relu_last, int n_heads, const float* const* head_w, const float* const* head_b,
                             const int* head_c, const int* head_act, float* const* head_dst, sa_stream_t stream);

/* Launch policy of the 3x3 MFMA kernels (process-wide, HOST). By default a layer with more tiles than the chip holds
 * workgroups is launched PERSISTENT: occupancy x CUs workgroups, each walking its share of the (frame, tile, cout-tile)
 * list with the first chunk of its next tile prefetched into the idle LDS stage while the current tile's epilogue runs.
 *   n  > 0  launch at most n workgroups in total (tests: a handful of workgroups over many tiles, uneven tails); raised to
 *           min(tiles, 8): the tile schedule hands each of the 8 XCDs a contiguous range, fewer workgroups would skip ranges.
 *           Also honoured by the persistent 32 -> 64 -> 64 block of sa_conv3x3_pair_bf16
 *   n == 0  automatic (default; SA_CONV_PERSIST=0 in the environment = one workgroup per tile, =k = k workgroups per CU)
 *   n  < 0  one workgroup per tile
 * Returns the previous value. Results do not depend on it (each tile's arithmetic is the same). */
int sa_conv3x3_set_grid_limit(int n);

/* The persistent tile loop of the TWO-workgroups-per-CU plain kernels (round 5; `PERS` in csrc/conv3x3.hip: the next tile's first
 * chunk is copied while the epilogue runs, and the wait in front of it is `s_waitcnt vmcnt(S)`, S = the epilogue's store
 * instructions, instead of vmcnt(0)). Bitwise neutral; measured 3-10 % SLOWER than one workgroup per tile on every layer of the
 * benchmark plan (profiles/r05_ab_session.md), so it is OFF by default -- this switch exists for A/B runs and for the tests that
 * keep the kernels correct.  mode: 0 off, 1 every plain multi-chunk layer, 2 few-chunk layers only, 3 many-chunk layers only,
 * -1 = back to the environment's choice (SA_CONV_PERS, default 0). Process-wide, HOST. Returns the previous value. */
int sa_conv3x3_set_persistent(int mode);


/* HOST helper: Keras Conv2D kernel (3,3,Cin,Cout) f32 -> the packed bf16 layout above. The input
 * channel axis is the concatenation [C0 | C1]; each part is zero-padded to C0P / C1P. */
int sa_pack_conv3x3_weights(const float* keras_kernel, int C0, int C0P, int C1, int C1P, int Cout,
                            int CoutP, uint16_t* packed);
size_t sa_conv3x3_packed_elems(int C0P, int C1P, int CoutP);

/* Conv2DTranspose(k3, s2, same) + bias + optional ReLU (encoder_decoder.py:304-310; out = 2n, full
 * transposed conv cropped at the end).  w [3][3][CoutP][CinP] bf16 (Keras layout, padded) */
int sa_convt3x3s2_bf16(const void* src, int CinP, const void* w, const float* bias, int CoutP,
                       int relu, int B, int H, int W, void* dst, sa_stream_t stream);

/* First-layer convolution of any kernel size / stride on the raw image with explicit TF SAME pads, e.g. the
 * hourglass stem Conv2D(k7, s2, same)+ReLU+BatchNormalization (hourglass.py:75-85). ensure_float fused for u8.
 * ResNet's input Lambdas fold in as well (resnet.py:326-362): CinW = 3 weight channels over a Cin = 1 image is
 * tile_channels, in_affine = {scale[CinW], shift[CinW]} applied to every in-bounds tap is imagenet_preproc_v1
 * (the BGR flip is a permutation of the weight channels done by the caller).
 *   src [B,H,W,Cin] u8|f32; w [kh][kw][CinW][CoutP] f32; post_scale/post_shift [CoutP] or NULL (BN after ReLU);
 *   dst [B,Ho,Wo,CoutP] bf16.  `relu`: bit 0 = ReLU; `relu | SA_LAYOUT_PLANES16` writes dst as 16-channel planes
 *   [B,CoutP/16,Ho,Wo,16] (sa_imgconv_u8_bf16 alike) -- the first launch of a plan that runs on planes */
int sa_image_conv_bf16(const void* src, int src_is_u8, int B, int H, int W, int Cin, int CinW, const float* in_affine,
                       int kh, int kw, int stride, int pad_top, int pad_left, int Ho, int Wo, const float* w, const float* bias, int CoutP, int relu,
                       const float* post_scale, const float* post_shift, void* dst, sa_stream_t stream);

/* The k7 stems on the matrix cores for uint8 images (hourglass.py:75-85, resnet.py:109-121): raw pixels are exact bf16 B
 * operands, the fp32 kernel x input scale enters as hi + lo bf16 fragments, the ImageNet means ride in the bias plus an
 * exact out-of-image indicator term (csrc/imgconv.hip). sa_imgconv_pack (host): w [k][k][CinW][Cout] f32, in_scale[CinW]
 * applied to the raw 0..255 value, mean[CinW] or NULL, bias_io [CoutP] in/out. relu / post affine as sa_image_conv_bf16. */
size_t sa_imgconv_packed_elems(int ksize, int CinW, int CoutP);
int sa_imgconv_pack(const float* w, int ksize, int CinW, int Cout, int CoutP, const float* in_scale, const float* mean,
                    uint16_t* packed, float* bias_io);
/* The same packer for ResNet's `tile_channels` input (resnet.py:326-362: a single-channel frame repeated three times in front of
 * a 3-channel first conv): w3 [k][k][3][Cout], in_scale3 / mean3 per weight channel. The three products of a tap share their pixel,
 * so the packed operand has ONE K slot per tap (float64 channel sum, then the hi + lo split) -- the CinW = 1 layout; run it with
 * sa_imgconv_u8_bf16(Cin = 1, CinW = 1, has_mean as packed). */
int sa_imgconv_pack_tiled(const float* w3, int ksize, int Cout, int CoutP, const float* in_scale3, const float* mean3,
                          uint16_t* packed, float* bias_io);
int sa_imgconv_u8_bf16(const void* src, int B, int H, int W, int Cin, int CinW, int ksize, int stride, int pad_top,
                       int pad_left, int Ho, int Wo, const void* wfrag, const float* bias, int CoutP, int relu,
                       int has_mean, const float* post_scale, const float* post_shift, void* dst, sa_stream_t stream);

/* Add layer (hourglass.py:190, resnet.py:249): dst = a + b [+ ReLU]; b may be [B,H/2,W/2,CP] read with
 * UpSampling2D(2, "nearest") (b_half_res = 1). Used when the addition can not be folded into a conv epilogue. */
int sa_add_bf16(const void* a, const void* b, int B, int H, int W, int CP, int b_half_res, int relu, void* dst,
                sa_stream_t stream);

/* Conv2D(k1, stride 1|2, valid) as a GEMM on the matrix cores with the extended epilogue of sa_conv3x3_ex_bf16:
 * ResNet bottleneck 1x1 convs + BatchNormalization [+ Add(shortcut) + ReLU] (resnet.py:168-253) and the 1x1 skip
 * projection of UpsamplingStack (upsampling.py:205-215).
 *   src [B,Hs,Ws,CinP] bf16; w from sa_pack_tapconv_weights(n_taps = 1); dst/residual [B,Ho,Wo,CoutP] bf16 with
 *   Ho = (Hs - 1) / stride + 1 */
int sa_conv1x1_bf16(const void* src, int CinP, const void* w, const float* bias, int CoutP, int relu, int B, int Hs, int Ws,
                    int stride, const float* post_scale, const float* post_shift, const void* residual, int relu_last,
                    void* dst, sa_stream_t stream);

/* Conv2D(k x k, stride 1, padding "same") [+ BatchNormalization] [+ residual] [+ ReLU] on a feature tensor, k up to 9: the
 * 7 x 7 convolutions of the UNet stem blocks (unet.py:105-127: SimpleConvBlock with kernel_size = stem_kernel_size;
 * encoder_decoder.py:113-121). The k*k taps run through the tap GEMM of sa_conv1x1_bf16 (tap t = ky*k + kx at offset
 * (ky, kx) - (k-1)/2, TF SAME), zero fill outside the image by the hardware.
 *   src [B,H,W,CinP] bf16; w from sa_pack_tapconv_weights(n_taps = k*k) of keras_kernel.reshape(k*k, Cin, Cout);
 *   dst / residual [B,H,W,CoutP] bf16 */
int sa_convk_bf16(const void* src, int CinP, const void* w, int ksize, const float* bias, int CoutP, int relu, int B, int H, int W,
                  const float* post_scale, const float* post_shift, const void* residual, int relu_last, void* dst,
                  sa_stream_t stream);

/* Conv2DTranspose(k4 | k3, stride 2, padding "same") [+ BatchNormalization] [+ ReLU] on the matrix cores
 * (upsampling.py:177-191; encoder_decoder.py:304-310): one GEMM launch per output phase (oy & 1, ox & 1).
 *   w_phase[4]: per phase a*2+b the packed weights (sa_pack_tapconv_weights) of the kernel taps listed by
 *   sa_convt_s2_phase_taps, each tap slice [Cin][Cout] = keras_kernel[ky, kx].T (Keras layout (kh,kw,Cout,Cin));
 *   dst [B,2Hs,2Ws,CoutP] bf16 */
int sa_convt_s2_bf16(const void* src, int CinP, const void* const* w_phase, int ksize, const float* bias, int CoutP,
                     int relu, int B, int Hs, int Ws, const float* post_scale, const float* post_shift, int relu_last,
                     void* dst, sa_stream_t stream);
int sa_convt_s2_phase_taps(int ksize, int phase, int* ky, int* kx);

/* host: w [n_taps][Cin][Cout] f32 -> MFMA A-operand fragments [CoutP/32][n_taps][CinP/16][64][8] bf16 */
size_t sa_tapconv_packed_elems(int n_taps, int CinP, int CoutP);
int sa_pack_tapconv_weights(const float* w, int n_taps, int Cin, int CinP, int Cout, int CoutP, uint16_t* packed);

/* MaxPooling2D(k, stride) with explicit top/left padding; padded taps are skipped (Keras "same") or, with
 * pad_is_zero = 1, take part as zeros (ZeroPadding2D + MaxPooling2D(3, s2, valid), resnet.py:125-126). */
int sa_maxpool_bf16(const void* src, int B, int H, int W, int CP, int k, int stride, int pad_top, int pad_left,
                    int pad_is_zero, int Ho, int Wo, void* dst, sa_stream_t stream);

/* MaxPooling2D(2, s2, same) on even sizes; UpSampling2D(2, "bilinear"|"nearest") */
int sa_maxpool2x2_bf16(const void* src, int B, int H, int W, int CP, void* dst, sa_stream_t stream);
int sa_upsample2x_bf16(const void* src, int B, int H, int W, int CP, int bilinear, void* dst,
                       sa_stream_t stream);

/* Head.make_head: Conv2D(k1, activation) (heads.py:42-62).  w [Cout][CinP] f32, dst [B,H,W,Cout] f32
 * act: 0 linear (all pose heads), 1 sigmoid (ClassMapsHead; identity heads are otherwise out of scope);
 * `act | SA_LAYOUT_PLANES16`: src is 16-channel planes [B,CinP/16,H,W,16] (Cout <= 64: the matrix-core kernel; dst NHWC f32) */
int sa_conv1x1_head(const void* src, int CinP, const float* w, const float* bias, int Cout, int act,
                    int B, int H, int W, float* dst, sa_stream_t stream);

/* resize_image (data/resizing.py:71-105): tf.image.resize bilinear, half-pixel centres, no antialias, on the float
 * image (InferenceLayer.preprocess applies ensure_float first). src [B,H,W,C] f32 -> dst [B,Ho,Wo,C] f32 */
int sa_resize_bilinear_f32(const float* src, int B, int H, int W, int C, int Ho, int Wo, float* dst,
                           sa_stream_t stream);
/* The same resize on uint8 frames whose taps enter as float(v) * in_scale: ensure_float (x * 1/255, normalization.py:49) folded
 * into the resize that follows it in InferenceLayer.preprocess (inference.py:940-967); same float32 operations, same results as
 * converting first. */
int sa_resize_bilinear_u8_f32(const void* src, int B, int H, int W, int C, int Ho, int Wo, float in_scale, float* dst,
                              sa_stream_t stream);

/* dtype plumbing: f32 NHWC [.., C] <-> bf16 NHWC [.., CP] (zero padded) */
int sa_f32_to_bf16_padded(const float* src, int n_pix, int C, int CP, void* dst, sa_stream_t stream);
int sa_bf16_to_f32(const void* src, int n_pix, int CP, int C, float* dst, sa_stream_t stream);
/* Range scan of one stored activation tensor (no reference counterpart: the reference computes and stores float32,
 * sleap/nn/inference.py:1047-1090 hands tf.float32 tensors around; fp16 storage has a finite range).
 * x: n elements of this library's 16-bit storage type (is_f32 = 0) or float32 (is_f32 = 1), 16-byte aligned, on the device.
 * out2 (device, 2 dwords, written asynchronously on `stream`): out2[0] = the largest FINITE |x| as float,
 * ((uint32_t*)out2)[1] = flags, bit 0: an infinity was seen, bit 1: a NaN was seen. One wave-reduced atomic per wavefront. */
int sa_tensor_absmax(const void* x, size_t n, int is_f32, float* out2, sa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Cross-frame identity tracking (host code). Replaces sleap.nn.tracking.Tracker with the
 * `simple` / `simplemaxtracks` candidate makers (tracking.py:442-507, 542-841) and sleap/nn/tracker/components.py
 * (similarities :33-196, greedy / Hungarian matching :199-226, pre-cull :229-417, FrameMatches :469-640,
 * connect_single_track_breaks :419-466), and with the `flow` / `flowmaxtracks` candidate makers (tracking.py:108-440, 1194-1240:
 * queued instances shifted into the current frame by sparse pyramidal Lucas-Kanade flow -- the sa_flow_* kernels; img_scale = 1
 * and save_shifted_instances = False only). The Kalman tracker (pykalman) is not covered.
 * Tracks are integers: the index into the tracker's list of spawned tracks (reference name "track_<index>").
 * ------------------------------------------------------------------------------------------------ */
enum { SA_SIM_INSTANCE = 0, SA_SIM_CENTROID = 1, SA_SIM_IOU = 2, SA_SIM_NORMALIZED_INSTANCE = 3, SA_SIM_OBJECT_KEYPOINT = 4 };
enum { SA_MATCH_GREEDY = 0, SA_MATCH_HUNGARIAN = 1 };

typedef struct sa_tracker_config {   /* Tracker.make_tracker_by_name arguments, tracking.py:844-876 */
  int max_tracks_mode;               /* 0 = SimpleCandidateMaker, 1 = SimpleMaxTracksCandidateMaker */
  int similarity;                    /* SA_SIM_* */
  int match;                         /* SA_MATCH_* */
  int track_window;
  double robust;                     /* robust_best_instance: in (0,1) -> np.quantile of the track's similarities, else max */
  int min_new_track_points;
  int min_match_points;
  int target_instance_count;
  int pre_cull_to_target;
  double pre_cull_iou_threshold;     /* <= 0: score-only culling */
  int max_tracks;
  int max_tracking;
  int oks_n_errors;                  /* 0 = keypoint_errors None (= 1) */
  const double* oks_errors;
  int oks_score_weighting;
  int oks_normalization;             /* 0 "all", 1 "ref", 2 "union" */
  int flow;                          /* 0 = Simple[MaxTracks]CandidateMaker, 1 = Flow[MaxTracks]CandidateMaker (tracking.py:108-440, 1194-1240) */
  int of_window_size;                /* flow: Lucas-Kanade window (default 21; 3..31) */
  int of_max_levels;                 /* flow: pyramid levels above the frame (default 3) */
  int save_shifted_instances;        /* flow (not flowmaxtracks): chain the flow through the latest shifted copy of a queued frame's
                                        instances (FlowCandidateMaker.save_shifted_instances, tracking.py:146-208, 239-256) */
  double img_scale;                  /* flow: FlowCandidateMaker.img_scale (tracking.py:116-131); 0 or 1 = frames as they are (ABI 3) */
} sa_tracker_config;

void* sa_tracker_create(const sa_tracker_config* cfg);   /* NULL on invalid configuration (sa_last_error) */
void sa_tracker_destroy(void* tracker);
int sa_tracker_reset(void* tracker);                     /* Tracker.reset_candidates */
int sa_tracker_n_tracks(void* tracker);                  /* tracks spawned so far */
/* Tracker.last_matches.has_only_first_choice_matches (components.py:478-480) of the last tracked frame that had instances:
 * 1 / 0, -1 before the first such frame (the Kalman tracker's test for a "good" initialisation frame, tracking.py:1243-1277) */
int sa_tracker_last_first_choice(void* tracker);

/* Tracker.track for ONE frame (tracking.py:642-773). points [n, n_nodes, 2] f32 (x, y; NaN = missing node),
 * point_scores [n, n_nodes] or NULL, inst_scores [n] or NULL, t < 0 = infer the time step.
 * Returns the tracked instances in the reference's order (matches first, then newly spawned tracks):
 * out_index[k] = index of the k-th returned instance in the input, out_track[k], out_score[k] = tracking_score
 * (match similarity; 0 for spawned tracks); *n_out <= n (instances can be culled or refused a new track). */
int sa_tracker_track(void* tracker, int n, int n_nodes, const float* points, const float* point_scores,
                     const float* inst_scores, int img_h, int img_w, int t, int* out_index, int* out_track,
                     double* out_score, int* n_out);

/* The same over n_frames consecutive frames laid out like the predictor's output: points [F, I, N, 2] NaN padded,
 * point_scores [F, I, N], inst_scores [F, I], n_valid [F]. out_track [F, I] (-1 = not tracked), out_score [F, I],
 * out_order [F, I] = position in the frame's returned list (or -1); t0 < 0 = infer every step. */
int sa_tracker_track_frames(void* tracker, int n_frames, int max_inst, int n_nodes, const float* points,
                            const float* point_scores, const float* inst_scores, const int* n_valid, int img_h, int img_w,
                            int t0, int* out_track, double* out_score, int* out_order);

/* Flow trackers look at the frames (Tracker.uses_image, tracking.py:139-141). sa_tracker_set_image hands over the frame that
 * the NEXT sa_tracker_track call belongs to: DEVICE [H,W,C] uint8 (C = 1 or 3); its pyramid is built on `stream` and kept by the
 * tracker (device memory it owns) while a queued instance refers to that time step. A no-op for the simple candidate makers.
 * sa_tracker_track_frames_images = sa_tracker_track_frames with images [F,frame_h,frame_w,C] (DEVICE; img_h / img_w stay what
 * normalized_instance_similarity divides by) and, optionally, per-frame time steps frame_t [F] (HOST; NULL: t0 + f or inferred). */
int sa_tracker_set_image(void* tracker, const void* image, int H, int W, int C, sa_stream_t stream);
int sa_tracker_track_frames_images(void* tracker, int n_frames, int max_inst, int n_nodes, const float* points,
                                   const float* point_scores, const float* inst_scores, const int* n_valid, int img_h, int img_w,
                                   int t0, const int* frame_t, const void* images, int frame_h, int frame_w, int C,
                                   sa_stream_t stream, int* out_track, double* out_score, int* out_order);

/* connect_single_track_breaks (components.py:419-466) in place on a [F, I] track table (-1 = empty slot). */
int sa_connect_single_track_breaks(int n_frames, int max_inst, const int* order, int* track, int instance_count);

/* ------------------------------------------------------------------------------------------------
 * Sparse pyramidal Lucas-Kanade optical flow on the device: replaces the cv2.calcOpticalFlowPyrLK call of
 * FlowCandidateMaker.flow_shift_instances (sleap/nn/tracking.py:258-356; winSize (w, w), maxLevel L, criteria (EPS | COUNT,
 * 30, 0.01), default flags / minEigThreshold 1e-4). OpenCV's integer conventions are kept (csrc/flow.hip, oracle/optical_flow.py).
 *   pyramid  one frame's levels (uint8) and Scharr derivatives (int16 x 2) in ONE caller-owned device buffer of
 *            sa_flow_pyramid_bytes(H, W, win, max_level) bytes; sa_flow_pyramid_levels = how many levels exist (an image must
 *            stay larger than the window). image: DEVICE [H,W,C] uint8, C = 1, or 3 (gray by cv2's COLOR_BGR2GRAY weights applied
 *            to the given channel order, as the reference does to its RGB frames). win in 3..31.
 *   lk       n points: prev_pts [n,2] (x, y; DEVICE) in the frame whose pyramid is pyr_prev[i] (DEVICE array of n device
 *            pointers -- points of several reference frames can share one launch) -> next_pts [n,2] in the frame of pyr_next,
 *            status [n] (1 = found), err [n] (mean |J - I| / 32 over the window; 0 where the point left the image). NaN points
 *            report status 0. All pyramids of one call must come from frames of the same (H, W, win, max_level). */
int sa_flow_pyramid_levels(int H, int W, int win, int max_level);
size_t sa_flow_pyramid_bytes(int H, int W, int win, int max_level);
int sa_flow_pyramid_build(const void* image, int H, int W, int C, int win, int max_level, void* pyramid, sa_stream_t stream);
/* F frames at once (images [F,H,W,C]; pyramids: DEVICE array of F device pointers, one buffer per frame): 2 + 2 levels launches
 * for the whole batch instead of per frame. */
int sa_flow_pyramid_build_batch(const void* images, int F, int H, int W, int C, int win, int max_level, void* const* pyramids,
                                sa_stream_t stream);
/* FlowCandidateMaker.img_scale != 1 (tracking.py:116-131, 311-314, 321, 333): both frames go through
 * cv2.resize(img, None, None, scale, scale) (INTER_LINEAR on the gray uint8 frame, OpenCV's 11-bit fixed-point form) before
 * the pyramid, points are multiplied by the scale before and divided by it after the flow. sa_flow_scaled_size = cv::resize's
 * dsize (cvRound(size * scale)); sa_flow_pyramid_build_scaled builds the pyramids of F frames [F,H,W,C] at that size
 * (`pyramid` for F == 1, else `pyramids` = DEVICE array of F buffers of sa_flow_pyramid_bytes(Hs, Ws, ...) bytes); the
 * Lucas-Kanade calls then take (Hs, Ws). */
int sa_flow_scaled_size(int H, int W, double img_scale, int* Hs, int* Ws);
int sa_flow_pyramid_build_scaled(const void* images, int F, int H, int W, int C, double img_scale, int win, int max_level,
                                 void* pyramid, void* const* pyramids, sa_stream_t stream);
int sa_flow_lk(const void* const* pyr_prev, const void* pyr_next, int H, int W, int win, int max_level, int n,
               const float* prev_pts, float* next_pts, uint8_t* status, float* err, int max_count, float epsilon,
               sa_stream_t stream);
/* The same with one TARGET pyramid per point (pyr_next: DEVICE array of n device pointers): every (reference frame, target frame)
 * pair of a whole batch of frames in one launch -- what the flow tracker does for a run of frames (the shifts depend on the
 * detections only, not on the track assignments, so they need not wait for the frame-by-frame matching). */
int sa_flow_lk_pairs(const void* const* pyr_prev, const void* const* pyr_next, int H, int W, int win, int max_level, int n,
                     const float* prev_pts, float* next_pts, uint8_t* status, float* err, int max_count, float epsilon,
                     sa_stream_t stream);

/* Top-down glue on the device (CentroidCrop / FindInstancePeaks, inference.py:1747-1966, 2059-2200): the number of crops per
 * frame is data dependent in the reference (ragged); here every frame has K crop slots, so nothing between the centroid
 * network and the instance network needs the host.
 *   sa_select_centroids   local peaks of the centroid maps (sa_find_local_peaks outputs, grid x stride units) -> per frame K
 *     slots: centroids [B,K,2] ((p / input_scale) + 0.5 when input_scale != 1, x precrop_resize; NaN in empty slots),
 *     centroid_vals [B,K], crop_centres [B,K,2] (= centroids, a finite far-outside point for empty slots: an all-zero crop),
 *     crop_offsets [B,K,2] = centroid - crop_size / 2, n_valid [B]. Peaks keep their order; with 0 <= max_instances < count
 *     the max_instances strongest in tf.math.top_k order (value descending, ties by lower index). max_instances < 0 = None.
 *     More than K -> SA_STATUS_INSTANCE_OVERFLOW in status[b].
 *   sa_finish_instance_peaks   in place on peaks [B*K,N,2] / vals [B*K,N] from sa_find_global_peaks of the crops: (p /
 *     input_scale) + 0.5 when input_scale != 1, + crop_offsets / input_scale (crop_offsets may be NULL), NaN for slots >=
 *     n_valid[b]. */
int sa_select_centroids(const float* peak_xy, const float* peak_val, const int32_t* peak_count, int B, int max_peaks, int K,
                        int max_instances, float input_scale, float precrop_resize, int crop_size, float* centroids,
                        float* centroid_vals, float* crop_centres, float* crop_offsets, int32_t* n_valid, int32_t* status,
                        sa_stream_t stream);
int sa_finish_instance_peaks(float* peaks, float* vals, const float* crop_offsets, const int32_t* n_valid, int B, int K, int N,
                             float input_scale, sa_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole network / whole layer in one call -- replaces `keras_model(imgs)` (sleap/nn/inference.py:2864-2890) and
 * `BottomUpInferenceLayer.call` (:2938-3003) for one batch. A non-Python host runs the hot path through these alone.
 *
 * `plan` (HOST, int64 words) is the launch plan `sleap_amd.nn.engine.DeviceNetwork` compiles from the Keras graph of
 * `best_model.h5` (`DeviceNetwork.plan_words()`): buffer table, output table and one record per fused launch. It is
 * process-local: weight operands appear as DEVICE ADDRESSES of tensors the caller uploaded (packed with the sa_pack_* helpers)
 * and keeps alive while the handle exists. The handle itself owns no device memory.
 *   workspace   caller-owned device scratch of sa_network_workspace_bytes(net, B, H, W) bytes holding every stored activation
 *               tensor of the plan (256-byte aligned slots); zero it once after allocation
 *   outputs     HOST array of sa_network_n_outputs(net) device pointers, model-output order (`keras_model.output_names`):
 *               [B, h, w, c] float32 each (sa_network_output_shape). 1x1 heads are written there directly; a 16-bit feature
 *               tensor that is a model output is converted. An entry may be NULL (head kept in the workspace, see
 *               sa_network_buffer).
 *   images      [B,H,W,C] uint8 (`images_are_u8`) or float32 in [0,1]; H, W multiples of sa_network_max_stride(net)
 * Nothing is allocated or synchronised; all launches go to `stream`. Errors: SA_ERR_INVALID_ARG (shape / channel mismatch),
 * SA_ERR_WORKSPACE (workspace too small), plus whatever a layer entry point reports. */
typedef struct sa_network sa_network_t;
int sa_network_create(const int64_t* plan, size_t n_words, sa_network_t** out);
void sa_network_destroy(sa_network_t* net);
int sa_network_n_outputs(const sa_network_t* net);
int sa_network_in_channels(const sa_network_t* net);
int sa_network_max_stride(const sa_network_t* net);
/* SA_LAYOUT_NHWC or SA_LAYOUT_PLANES16: how the plan's 16-bit tensors lie in the workspace (sa_network_buffer). The plan
 * compiler picks planes when every launch of the plan supports them (the UNet family); outputs are [B,h,w,c] float32 always. */
int sa_network_layout(const sa_network_t* net);
int sa_network_output_shape(const sa_network_t* net, int index, int H, int W, int* h, int* w, int* c);
size_t sa_network_workspace_bytes(const sa_network_t* net, int B, int H, int W);
void* sa_network_buffer(const sa_network_t* net, int buf, int B, int H, int W, void* workspace, int* h, int* w, int* cp,
                        int* is_f32);
int sa_network_forward(const sa_network_t* net, const void* images, int images_are_u8, int B, int H, int W, int C,
                       float* const* outputs, void* workspace, size_t ws_bytes, sa_stream_t stream);

/* Parameters of BottomUpInferenceLayer + PAFScorer (inference.py:2793-2860, paf_grouping.py:1318-1404) for
 * sa_bottomup_predict. `edges` [n_edges][2] and `sorted_edge_inds` [n_sorted] are DEVICE int32 tables (skeleton edge node
 * indices; PAFScorer.sorted_edge_inds). */
typedef struct sa_bottomup_params {
  int confmaps_ind, pafs_ind, offsets_ind; /* model-output indices (find_head); offsets_ind = -1 without the head */
  float peak_threshold;                    /* 0.2 */
  int refinement;                          /* SA_REFINE_INTEGRAL / SA_REFINE_LOCAL / SA_REFINE_NONE */
  int integral_patch_size;                 /* 5 */
  float cm_output_stride, pafs_stride;
  int n_nodes, n_edges;
  const int32_t* edges;
  const int32_t* sorted_edge_inds;
  int n_sorted;
  float max_edge_length_ratio, dist_penalty_weight; /* 0.25, 1.0 */
  int n_points;                                     /* 10 */
  float min_line_scores;                            /* 0.25 */
  int min_instance_peaks;                           /* 0 */
  int max_peaks, max_node_peaks, max_instances;     /* capacities of the fixed-shape buffers (status bits report overflow) */
} sa_bottomup_params;

size_t sa_bottomup_workspace_bytes(const sa_network_t* net, const sa_bottomup_params* params, int B, int H, int W);
/* -> instance_peaks [B,max_instances,n_nodes,2] f32 (NaN padded, image pixels of the network input), instance_peak_vals
 * [B,max_instances,n_nodes], instance_scores [B,max_instances], n_instances [B] i32, status [B] i32 (SA_STATUS_* bits; zeroed
 * here). The +0.5 / input_scale un-scaling of inference.py:2985-2990 is the caller's (it applies only when input_scale != 1). */
int sa_bottomup_predict(const sa_network_t* net, const sa_bottomup_params* params, const void* images, int images_are_u8, int B,
                        int H, int W, int C, float* instance_peaks, float* instance_peak_vals, float* instance_scores,
                        int32_t* n_instances, int32_t* status, void* workspace, size_t ws_bytes, sa_stream_t stream);

/* ---- video ingest: slice_data() of one H.264 picture, HOST code. The reference reads video through
 * cv2.VideoCapture / FFmpeg (sleap/io/video.py:340-504, `MediaVideo.get_frame`); neither exists in this image. The container,
 * parameter sets, slice header, picture order counts and reference lists are handled by the caller (sleap_amd/io/_h264.py,
 * whose pure-Python slice decoder is this function's checker: tests/test_h264_native.py, bit-exact picture by picture); this
 * entry decodes the macroblocks of one slice = one picture: entropy decoding (CABAC with cabac_init_idc 0, or CAVLC), intra and
 * inter prediction (P_Skip, spatial / temporal direct, weighted prediction), residual reconstruction, the in-loop edge filter.
 * Progressive Baseline / Main / High (8-bit 4:2:0, flat scaling matrices) profile; returns SA_ERR_INVALID_ARG with the reason for anything else or for a slice whose
 * entropy decode does not end exactly on the last macroblock. */
typedef struct sa_h264_pic {   /* buffers of one picture, all HOST memory owned by the caller */
  uint8_t *y, *cb, *cr;        /* [16 mb_h][16 mb_w], [8 mb_h][8 mb_w] x 2 */
  int16_t* mv;                 /* [2 lists][4 mb_h][4 mb_w][2]: quarter-sample motion vector of every 4x4 block */
  int8_t* ref;                 /* [2][4 mb_h][4 mb_w]: reference index into the slice's list, -1 = list not used */
  int32_t* refid;              /* [2][4 mb_h][4 mb_w]: `id` of the referenced picture, -1 */
  uint8_t* intra4;             /* [4 mb_h][4 mb_w]: block belongs to an intra macroblock */
  int32_t poc, id;             /* picture order count; identity (unique among the pictures alive at one time) */
} sa_h264_pic;
typedef struct sa_h264_slice {
  int32_t mb_w, mb_h;
  int32_t slice_type;                      /* 0 P, 1 B, 2 I */
  int32_t cabac;                           /* entropy_coding_mode_flag */
  int32_t qp, chroma_qp_offset;            /* SliceQPY, chroma_qp_index_offset */
  int32_t disable_deblock, filter_offset_a, filter_offset_b;
  int32_t direct_spatial, direct_8x8_inference;
  int32_t nref[2];                         /* num_ref_idx_l0 / l1_active */
  int32_t weighted_mode;                   /* 0 default, 1 explicit (`weights`), 2 implicit (from the picture order counts) */
  int32_t luma_log2_denom, chroma_log2_denom;
  int32_t weights[2][32][3][2];            /* explicit mode: [list][refIdx][Y, Cb, Cr][weight, offset] */
  int32_t data_bit_offset;                 /* first bit of slice_data() in `rbsp` (behind cabac_alignment_one_bit) */
  int32_t transform_8x8_mode;              /* PPS transform_8x8_mode_flag (High profile) */
  int32_t chroma_qp_offset_cr;             /* second_chroma_qp_index_offset (= chroma_qp_offset when the PPS has none) */
} sa_h264_slice;
/* rbsp: the slice NAL unit's payload without header byte and emulation prevention bytes; list0 / list1: nref[] entries each
 * (an entry with y == NULL is "no reference picture"); cur: written completely (planes and motion data); stats (8 ints or
 * NULL): Intra4x4, Intra16x16, skipped and inter macroblock counts, bits left at the end of the slice data, Intra8x8 and 8x8-
 * transform inter macroblock counts. */
int sa_h264_decode_slice(const sa_h264_slice* s, const uint8_t* rbsp, int64_t n_bytes, const sa_h264_pic* list0,
                         const sa_h264_pic* list1, sa_h264_pic* cur, int32_t* stats);

/* The colour conversion behind cv2.VideoCapture.read as `MediaVideo` sees it (video.py:470-485: BGR frames, channel 0 kept for a
 * grayscale video): limited-range BT.601 YCbCr 4:2:0 planes -> [height][width][channels] uint8, channels 3 = BGR, 1 = blue alone, with
 * libswscale's SIMD arithmetic (13-bit coefficients, truncation). HOST buffers; strides in bytes. Checker: io/_h264_intra.py swscale_bgr. */
int sa_yuv420_to_bgr(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int width, int height, int y_stride, int c_stride,
                     uint8_t* out, int channels);

#ifdef __cplusplus
}
#endif
#endif /* SLEAP_AMD_H */
