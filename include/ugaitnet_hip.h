/*
 * ugaitnet_hip.h -- C ABI of libugaitnet_hip.so (gfx950 / MI355X).
 *
 * The reference (avagait/ugaitnet) has no FFI/plugin layer: every op below is an *implicit* TensorFlow
 * primitive reached through the Keras graph that nets/mj_uwyhNets_ba.py builds.  Each entry point cites
 * the reference call site (path relative to the reference root) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; no ownership transfer;
 *   - tensors are dense, channels-last (NHWC), fp32; index maps are uint8;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work;
 *   - return 0 on success, a positive hipError_t on a HIP failure, UGN_EINVAL on a shape the kernels do
 *     not implement.  ugn_last_error() returns a static description of the last failure of the thread;
 *   - thread-compatible: no global mutable state besides that message and ONE process-wide launch setting,
 *     ugn_set_persistent_wgs (how many CUs the persistent launches occupy; results do not depend on it).
 */
#ifndef UGAITNET_HIP_H
#define UGAITNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UGN_EINVAL (-22)
#define UGN_ABI_VERSION 1

/* fusion modes of fMerge (nets/mj_uwyhNets_ba.py:814,1189) */
#define UGN_FUSE_SIGN_MAX 0 /* mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:169-178 */
#define UGN_FUSE_MAX 1      /* keras Maximum */
#define UGN_FUSE_AVG 2      /* keras Average */

int ugn_abi_version(void);
const char* ugn_last_error(void);

/* ---- first layer: ZeroPadding2D(2) + Conv2D(32, 5x5, same, no bias) + LeakyReLU(0.3) ---------------
 * nets/mj_uwyhNets_ba.py:428-430.  x [n,60,60,cin] (cin 1 or 2), w HWIO [5,5,cin,32], a1 [n,64,64,32]. */
int ugn_conv5x5_in_fwd(const float* x, const float* w, float* a1, uint32_t* a1_sign, int n, int cin, void* stream);
/* a1_sign (optional, [n,64,64] words): bit c of a pixel's word = (a1[..,c] > 0).  With it the backward never re-reads a1 for
 * its LeakyReLU' factor: the a2 data gradient leaves dL/da1 in dz1 and ugn_conv5x5_in_wgrad applies the factor from the bits.
 * dw [5,5,cin,32] = sum over frames/pixels of x (padded) * dz1 [n,64,64,32] (* LeakyReLU'(a1) when a1_sign is given).
 * ws: >= ugn_conv5x5_in_wgrad_ws(). */
size_t ugn_conv5x5_in_wgrad_ws(int n, int cin);
int ugn_conv5x5_in_wgrad(const float* x, const float* dz1, const uint32_t* a1_sign, float* dw, int n, int cin, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- 3x3 convolutions (TimeDistributed Conv2D / Conv2D, same, no bias), nets/mj_uwyhNets_ba.py:431-462 --
 * Forward weights are consumed in packed [9][cout][cin] order produced by ugn_pack3x3 from HWIO. */
int ugn_pack3x3(const float* w_hwio, float* w_packed, int cin, int cout, void* stream);
/* out = LeakyReLU(conv(in)).  pool != 0: additionally MaxPooling2D(2,2): out is [n,hw/2,hw/2,cout] and
 * out_idx (uint8, same shape) holds the first maximum of each window in row-major order (0..3). */
int ugn_conv3x3_fwd(const float* in, const float* w_packed, float* out, uint8_t* out_idx, int n, int hw, int cin,
                    int cout, int pool, void* stream);
/* Data gradient.  Forward layer: cin -> cout at hw x hw.  dz is the gradient w.r.t. the layer's pre-activation
 * [n,hw,hw,cout]; if dz_idx != NULL, dz is given at pooled resolution [n,hw/2,hw/2,cout] (already multiplied by
 * LeakyReLU') and is scattered through dz_idx on the fly (MaxPool backward).  w is the HWIO weight.
 *   t   = conv_transpose(dz, w) (+ addend if addend != NULL)
 *   raw_out = t                           (if raw_out != NULL)
 *   out = t * (act > 0 ? 1 : 0.3)         (if act != NULL, else out = t); all [n,hw,hw,cin]. */
int ugn_conv3x3_dgrad(const float* dz, const uint8_t* dz_idx, const float* w_hwio, const float* act,
                      const float* addend, float* out, float* raw_out, int n, int hw, int cin, int cout,
                      void* stream);
/* Weight gradient dw HWIO [3,3,cin,cout] = sum in (x) dz; dz/dz_idx as in dgrad. */
size_t ugn_conv3x3_wgrad_ws(int n, int hw, int cin, int cout);
int ugn_conv3x3_wgrad(const float* in, const float* dz, const uint8_t* dz_idx, float* dw, int n, int hw, int cin,
                      int cout, void* ws, size_t ws_bytes, void* stream);

/* Winograd F(2x2,3x3) variants of the forward / data-gradient convolutions (same results up to fp32 rounding, 2.25x
 * fewer matrix FLOPs).  u_packed: 16*cin*cout floats produced by ugn_wino_pack from the HWIO weight
 * (dgrad = 0 for ugn_conv3x3_fwd_wino; for ugn_conv3x3_dgrad_wino 1 when dz is full-resolution and 3 when dz is the
 * pooled gradient + argmax of a MaxPool'ed layer: the kernels for the two cases consume different filter layouts).
 * Arguments otherwise as the direct versions; in the data gradient `addend` and `raw_out` require `act`. */
int ugn_wino_pack(const float* w_hwio, float* u_packed, int cin, int cout, int dgrad, void* stream);
/* Same for up to 64 (layer, direction) jobs in ONE launch (all branches of a 3-modality model: 54); HOST arrays of length njobs. */
int ugn_wino_pack_multi(const float* const* w_hwio_host, float* const* u_packed_host, const int* cin_host,
                        const int* cout_host, const int* dgrad_host, int njobs, void* stream);
int ugn_conv3x3_fwd_wino(const float* in, const float* u_packed, float* out, uint8_t* out_idx, int n, int hw, int cin,
                         int cout, int pool, void* stream);
int ugn_conv3x3_dgrad_wino(const float* dz, const uint8_t* dz_idx, const float* u_packed, const float* act,
                           const float* addend, float* out, float* raw_out, int n, int hw, int cin, int cout,
                           void* stream);
/* The same two operators for TWO convolutions of one shape in a single launch (arrays of length 2: pointers and image
 * counts per job).  The encoder of nets/mj_uwyhNets_ba.py:431-462 applies every 3x3 shape twice, to the L frames of a clip
 * and to the set-pooled map of the global branch; the second has 1/L of the work and rides along with the first.  In the
 * pair data gradient both jobs must use the same subset of act / addend / raw_out and both or neither a pooled dz. */
int ugn_conv3x3_fwd_wino_pair(const float* const* in, const float* const* u_packed, float* const* out,
                              uint8_t* const* out_idx, const int* n, int hw, int cin, int cout, int pool, void* stream);
int ugn_conv3x3_dgrad_wino_pair(const float* const* dz, const uint8_t* const* dz_idx, const float* const* u_packed,
                                const float* const* act, const float* const* addend, float* const* out,
                                float* const* raw_out, const int* n, int hw, int cin, int cout, void* stream);
/* The same for up to 6 convolutions of one shape in a single launch (arrays of length njobs): the three modality encoders
 * of nets/mj_uwyhNets_ba.py:1102-1140 are the same ten layer shapes, so a 3-modality step issues ONE launch per layer for the
 * frame-level convolutions of all modalities and their set-level twins.  Items of all jobs form one list that the
 * persistent workgroups stride over; a job's filter slices stay resident in LDS across its items where they fit.  bf16 must be 0
 * (the bf16-operand Winograd kernels of rounds 1-4 are retired: UGN_EINVAL).  Rules for act / addend / raw_out / dz_idx as in the pair form. */
int ugn_conv3x3_fwd_wino_multi(const float* const* in, const float* const* u_packed, float* const* out,
                               uint8_t* const* out_idx, const int* n, int njobs, int hw, int cin, int cout, int pool, int bf16,
                               void* stream);
int ugn_conv3x3_dgrad_wino_multi(const float* const* dz, const uint8_t* const* dz_idx, const float* const* u_packed,
                                 const float* const* act, const float* const* addend, float* const* out,
                                 float* const* raw_out, const int* n, int njobs, int hw, int cin, int cout, int bf16,
                                 void* stream);
/* Data gradient whose addend is the set-max gradient of the layer's output (the Add of the two gradient paths into p2 / p4,
 * nets/mj_uwyhNets_ba.py:435,451): out = (dgrad + ((act == smax_m[clip]) ? smax_g[clip] : 0)) * LeakyReLU'(act), clip =
 * image / frames; smax_m [n/frames,hw,hw,cin] = the set maxima (ugn_setmax_fwd_cnt), smax_g = dL/dm / #maxima (ugn_div of
 * the incoming gradient by the count).  Shapes: (hw 32, cin 32, cout 64) and (hw 16, cin 64, cout 128). */
int ugn_conv3x3_dgrad_wino_routed(const float* dz, const float* u_packed, const float* act, const float* smax_m,
                                  const float* smax_g, int frames, float* out, int n, int hw, int cin, int cout,
                                  void* stream);
/* Winograd F(2x2,3x3) weight gradient (Conv2DBackpropFilter of the same layers); arguments as ugn_conv3x3_wgrad,
 * workspace size from ugn_conv3x3_wgrad_wino_ws (0 = unsupported shape).  Deterministic (fixed summation order). */
size_t ugn_conv3x3_wgrad_wino_ws(int n, int hw, int cin, int cout);
int ugn_conv3x3_wgrad_wino(const float* in, const float* dz, const uint8_t* dz_idx, float* dw, int n, int hw, int cin,
                           int cout, void* ws, size_t ws_bytes, void* stream);
/* Two weight gradients of one shape in a single launch (arrays of length 2); workspace: ugn_conv3x3_wgrad_wino_ws(n0 + n1). */
int ugn_conv3x3_wgrad_wino_pair(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx,
                                float* const* dw, const int* n, int hw, int cin, int cout, void* ws, size_t ws_bytes,
                                void* stream);
/* Up to 6 weight gradients of one shape in a single launch.  The 8x16-pixel regions of all jobs form one list and every
 * workgroup group owns an equal contiguous share of it (balanced to one region whatever the jobs' sizes); a share that crosses
 * a job boundary leaves one partial-sum slab per job.  Workspace: ugn_conv3x3_wgrad_wino_ws (independent of n). */
int ugn_conv3x3_wgrad_wino_multi(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx,
                                 float* const* dw, const int* n, int njobs, int hw, int cin, int cout, void* ws,
                                 size_t ws_bytes, int bf16, void* stream);

/* ---- set pooling over the L frames: tf.math.reduce_max(x, axis=1), nets/mj_uwyhNets_ba.py:435,451,463 ----
 * p [b,l,s] -> m [b,s]; if addend != NULL also sum_out = m + addend (the Add layers :452,:465). */
int ugn_setmax_fwd(const float* p, const float* addend, float* m, float* sum_out, int b, int l, size_t s,
                   void* stream);
/* out[b,l,s] = (p == max_l p) ? dm / (#maxima) : 0, times LeakyReLU'(p) when apply_lrelu != 0. */
/* Forward that also returns cnt [b,s] = the number of frames holding the maximum (fp32; l <= 32). */
int ugn_setmax_fwd_cnt(const float* p, const float* addend, float* m, float* sum_out, float* cnt, int b, int l, size_t s,
                       void* stream);
/* out = g * LeakyReLU'(act) elementwise (n a multiple of 4; out may alias g): the LeakyReluGrad of a set-level map whose
 * data gradient was computed with a plain epilogue so that it could share a launch with its frame-level twin. */
int ugn_lrelu_bwd(const float* g, const float* act, float* out, size_t n, void* stream);
/* x *= factor elementwise, in place (global-batch data parallelism: every replica holds the head's gradient of the WHOLE
 * batch, so it is pre-scaled by 1/replicas before the summing all-reduce). */
int ugn_scale(float* x, float factor, size_t n, void* stream);
/* out = a / b elementwise (n a multiple of 4): dL/dm divided by the number of maxima (TF's reduce_max gradient). */
int ugn_div(const float* a, const float* b, float* out, size_t n, void* stream);
/* addend (optional, [b,l,s], may alias out): a second gradient path into p, summed before the LeakyReLU' factor:
 * out = ((p == m) ? dm / #maxima : 0) + addend) * (apply_lrelu ? LeakyReLU'(p) : 1). */
int ugn_setmax_bwd(const float* p, const float* dm, const float* addend, float* out, int b, int l, size_t s,
                   int apply_lrelu, void* stream);

/* The pooling / elementwise steps above for up to 4 tensors of one shape in a single launch (arrays of length njobs; b[j] clips
 * in job j): the modality branches of nets/mj_uwyhNets_ba.py:1102-1140 run the same steps on their own tensors. */
int ugn_setmax_fwd_multi(const float* const* p, const float* const* addend, float* const* m, float* const* sum_out,
                         const int* b, int njobs, int l, size_t s, void* stream);
int ugn_setmax_bwd_multi(const float* const* p, const float* const* dm, const float* const* addend, float* const* out,
                         const int* b, int njobs, int l, size_t s, int apply_lrelu, void* stream);
/* Set pooling with ROUTING WORDS: the forward pass also writes, per (clip, element), two u32 words -- route [b][s/4][2][4]: word 0 of
 * a channel = bit t set iff frame t holds the maximum, word 1 = bit t set iff frame t is positive (l <= 32) -- and the gradient reads
 * those 8 bytes per element INSTEAD of the l frames (4 l bytes): the same results as ugn_setmax_fwd_multi / ugn_setmax_bwd_multi bit
 * for bit (reduce_max + its gradient, nets/mj_uwyhNets_ba.py:435,451,463). */
int ugn_setmax_fwd_routed_multi(const float* const* p, const float* const* addend, float* const* m, float* const* sum_out,
                                uint32_t* const* route, const int* b, int njobs, int l, size_t s, void* stream);
int ugn_setmax_bwd_routed_multi(const uint32_t* const* route, const float* const* dm, const float* const* addend, float* const* out,
                                const int* b, int njobs, int l, size_t s, int apply_lrelu, void* stream);
int ugn_lrelu_bwd_multi(const float* const* g, const float* const* act, float* const* out, const size_t* n, int njobs,
                        void* stream);

/* ---- horizontal pyramid pooling, nets/mj_uwyhNets_ba.py:468-481.  a, s3 [b,16,16,128] -> feat [62,b,128] */
int ugn_hpp_fwd(const float* a, const float* s3, float* feat, int b, void* stream);
/* dfeat [62,b,128] -> dm3 = dL/da + dL/ds3 (a also feeds s3 = b4 + a), dzb4 = dL/ds3 * LeakyReLU'(b4). */
int ugn_hpp_bwd(const float* a, const float* s3, const float* b4, const float* dfeat, float* dm3, float* dzb4,
                int b, void* stream);
/* the same for up to 4 modality branches in one launch (arrays of length njobs) */
int ugn_hpp_fwd_multi(const float* const* a, const float* const* s3, float* const* feat, const int* b, int njobs, void* stream);
int ugn_hpp_bwd_multi(const float* const* a, const float* const* s3, const float* const* b4, const float* const* dfeat,
                      float* const* dm3, float* const* dzb4, const int* b, int njobs, void* stream);

/* ---- MatMul layer (62 per-bin FCs), nets/mj_uwyhNets_ba.py:23-40: feat [62,b,128] x w [62,128,256] ------- */
int ugn_binfc_fwd(const float* feat, const float* w, float* out, int b, void* stream);
int ugn_binfc_bwd(const float* feat, const float* w, const float* dout, float* dw, float* dfeat, int b,
                  void* stream);
/* the same for up to 4 modality branches in one launch (arrays of length njobs) */
int ugn_binfc_fwd_multi(const float* const* feat, const float* const* w, float* const* out, const int* b, int njobs, void* stream);
int ugn_binfc_bwd_multi(const float* const* feat, const float* const* w, const float* const* dout, float* const* dw,
                        float* const* dfeat, const int* b, int njobs, void* stream);
/* the two halves separately: parts = 1 the weight gradient dw only, 2 the feature gradient dfeat only (the one the rest of the
 * backward pass waits for), 3 both (= ugn_binfc_bwd_multi) */
int ugn_binfc_bwd_parts_multi(const float* const* feat, const float* const* w, const float* const* dout, float* const* dw,
                              float* const* dfeat, const int* b, int njobs, int parts, void* stream);

/* ---- gate (:51-54) + fMerge (:814,:1189).  outs/uses: HOST arrays of nmod device pointers; x_m [62,b,256],
 * use_m [b].  fused [62,b,256]; sel uint8 (selected modality). */
int ugn_gate_fuse_fwd(const float* const* outs_host, const float* const* uses_host, int nmod, int mode,
                      float* fused, uint8_t* sel, int b, void* stream);
int ugn_gate_fuse_bwd(const float* dfused, const uint8_t* sel, const float* const* uses_host,
                      float* const* douts_host, int nmod, int mode, int b, void* stream);
/* ---- tf.math.l2_normalize(x, axis=1) on [62,b,256] (axis 1 = batch), :817,:1191 ----------------------- */
int ugn_l2norm_batch_fwd(const float* f, float* sig, int b, void* stream);
int ugn_l2norm_batch_bwd(const float* f, const float* sig, const float* dsig, float* df, int b, void* stream);
/* the gate / fMerge and the batch-axis normalisation in one launch, and their gradients in one (b <= 32 clips per replica): the same
 * arithmetic and the same outputs as the two calls each replaces */
int ugn_gate_norm_fwd(const float* const* outs_host, const float* const* uses_host, int nmod, int mode, float* fused, uint8_t* sel,
                      float* sig, int b, void* stream);
int ugn_gate_norm_bwd(const float* f, const float* sig, const float* dsig, const uint8_t* sel, const float* const* uses_host,
                      float* const* douts_host, int nmod, int mode, int b, void* stream);

/* ---- classification head: transpose+Flatten+Dense(softmax) + categorical cross-entropy, :848-850,:865 ---
 * sig [62,b,256], wc [15872,ncls], bc [ncls].  part: workspace [248,b,ncls] floats.
 * probs [b,ncls]; row_loss [b] (= -sum t log p); dlogits [b,ncls] = (p - t) * grad_scale; hit [b] (argmax match). */
int ugn_head_fwd(const float* sig, const float* wc, const float* bc, const float* onehot, float* part,
                 float* probs, float* row_loss, float* dlogits, float* hit, float grad_scale, int b, int ncls,
                 void* stream);
/* dwc, dbc written; dsig (+)= dlogits x wc^T in [62,b,256] layout (accumulate != 0 adds to dsig). */
int ugn_head_bwd(const float* sig, const float* wc, const float* dlogits, float* dwc, float* dbc, float* dsig,
                 int accumulate, int b, int ncls, void* stream);

/* ---- batch-all triplet loss, nets/triplet_loss_all.py:8-77 --------------------------------------------
 * Host helper: hp/hn pair-index lists (row-major boolean_mask order).  Returns 0, or UGN_EINVAL when the
 * pair counts are not divisible by m (the reference's tf.reshape([n,m,-1,1]) would raise). */
int ugn_triplet_indices_host(const int32_t* labels_host, int m, int32_t* hp_host, int32_t* hn_host, int* kp,
                             int* kn);
/* sig [62,m,256]; hp [m*kp], hn [m*kn] device int32.  bin_loss [62] (sum/num, 0 if num==0), bin_num [62].
 * dsig = grad_scale/(62*num_k) * d(sum_k)/dsig (written, not accumulated).  m <= 128. */
int ugn_triplet_fwd_bwd(const float* sig, const int32_t* hp, const int32_t* hn, int kp, int kn, float margin,
                        float* bin_loss, float* bin_num, float* dsig, float grad_scale, int m, void* stream);
/* Batch-HARD triplet loss per bin, the mode `compile_hard` names (nets/mj_uwyhNets_ba.py:1301-1306, tfa.losses.TripletHardLoss
 * with soft = False and L2 distances): per anchor the largest distance to another sample of its identity and the smallest to a
 * sample of another identity (tfa's masked maximum / minimum, including their empty-set values), L_bin = mean_a max(hp - hn +
 * margin, 0), mean over the 62 bins.  labels: int32 [m] in device memory.  bin_num = anchors with a positive hinge.  The
 * reference itself never calls compile_hard, and tfa's loss takes [batch, dim] embeddings: applying it per bin is this
 * build's reading of north_star's "batch-hard triplet". */
int ugn_triplet_hard_fwd_bwd(const float* sig, const int32_t* labels, float margin, float* bin_loss, float* bin_num,
                             float* dsig, float grad_scale, int m, void* stream);

/* ---- keras Adam (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:227): p -= lr_t * m / (sqrt(v) + eps) -------- */
int ugn_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                  float eps, float grad_scale, void* stream);
/* Same update, lr_t read from device memory (one float): the launch can be captured in a hipGraph and replayed every step. */
int ugn_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* lr_t_dev, float b1, float b2,
                      float eps, float grad_scale, void* stream);

/* ---- device-side batch assembly ----------------------------------------------------------
 * One modality of one batch: decode + re-layout + row expansion of data/mj_dataGeneratorMMUWYHsingle_repetitions.py
 * (`__load_dd` :300-318, gaitset layout :746-753, expansion / disabling :732-737,776-806).
 * raw: device array [nbase][60][60][25*channels], int16 (is_int16) or uint8, exactly the `data` array of a sample file.
 * src_row [nrows] int32: the base sample an output row copies, or -1 for an absent / disabled modality.
 * x_out [nrows][25][60][60][channels] = ((clip(v) / divisor) * post_mul) - offset, or `noise` everywhere for a -1 row;
 * clip: |v| > clip_max -> 1e-8, |v| < clip_min -> 1e-8 (each only when > 0).  flag_out [nrows] = 1 / 0. */
int ugn_assemble_modality(const void* raw, int is_int16, const int32_t* src_row, int nrows, int channels, float divisor,
                          float offset, float post_mul, float clip_max, float clip_min, float noise, float* x_out,
                          float* flag_out, void* stream);

/* Row gather / scatter of the mask-skipping step: a masked (clip, modality) pair is multiplied by 0 in the gate
 * (nets/mj_uwyhNets_ba.py:51-54), so a modality's encoder may run on the clips whose flag is 1 only.  t = [outer][rows][row_floats]
 * fp32, idx [nidx] int64 (device) row numbers, row_floats a multiple of 4, 16-byte aligned pointers.
 * gather: dst [outer][nidx][row_floats] = src[o][idx[i]][:]   (src has src_rows rows per outer slice);
 * scatter: dst[o][idx[i]][:] = src[o][i][:]                   (dst has dst_rows rows per outer slice; other rows are left alone). */
int ugn_gather_rows(const float* src, const int64_t* idx, float* dst, int outer, int src_rows, int nidx, size_t row_floats, void* stream);
int ugn_scatter_rows(const float* src, const int64_t* idx, float* dst, int outer, int dst_rows, int nidx, size_t row_floats, void* stream);

/* ---- evaluation: k-NN over gait signatures ----------------------------------------------
 * Replaces sklearn KNeighborsClassifier(n_neighbors=k).fit(gallery, labels).predict(probes) of
 * mains/mj_testUWYHGaitNet_open_tum.py:328-341: Euclidean, uniform weights, majority vote, smallest label on a tied
 * vote (nearer-first, lower gallery index on equal distances).  gallery [ngallery,dim], probes [nprobe,dim] fp32;
 * pred [nprobe]; neighbours (optional) [nprobe,k] gallery indices, nearest first.  1 <= k <= 16. */
size_t ugn_knn_ws(int ngallery, int nprobe);
int ugn_knn_predict(const float* gallery, const int32_t* gallery_labels, const float* probes, int ngallery, int nprobe,
                    int dim, int k, int32_t* pred, int32_t* neighbours, void* ws, size_t ws_bytes, void* stream);

/* Persistent workgroups per launch: 8..256, 0 = default (256 = one per CU).  A process-wide setting (the one piece of mutable state
 * besides the error message): under data parallelism a value below 256 leaves 256 - n CUs to RCCL's channels while the backward pass
 * runs (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:342-349: the gradient all-reduce of MirroredStrategy).  It sizes EVERY persistent
 * launch that owns its CUs' whole LDS: the x3, bf16 (and, where built, f16x2) 3x3 forward / data-gradient kernels (items stride over n, or 2n,
 * workgroups; the x3 weight gradients keep their 256 fixed shares), their weight gradients (the grid holds the largest multiple of 8 groups per block combination that fits n; a
 * workgroup then walks several of the launch's FIXED shares in turn) and the 5x5 forward (4n workgroups).  Every result is
 * bit-identical for every n.  Call it between launches, not concurrently with them. */
int ugn_set_persistent_wgs(int n);
/* the grid in force (8..256): what a report of a data-parallel run should state beside its collective timings */
int ugn_get_persistent_wgs(void);

/* The f16x2 ("H2") kernel set of rounds 3-4 is an opt-in build (python -m ugaitnet_amd.build --h2): its entry points are declared
 * in include/ugaitnet_hip_h2.h and are NOT exported by the default library. */

/* ---- bf16 path: BASELINE.json configs[4] ("MFMA bf16 conv tiles + fp32 accumulate"), SURVEY 8(d) C5 ----------------------------
 * bf16 activations, gradients and saved tensors in HBM ([pixel][c] bf16, half the bytes of fp32), one v_mfma_f32_32x32x16_bf16 per
 * product, fp32 accumulate; weight gradients, master weights and Adam fp32.  Same reference lines as the fp32 / H2 entry points with
 * the same names (nets/mj_uwyhNets_ba.py:428-481); `GaitCore(conv_precision="bf16")`.
 * pooled_host[j] (data-gradient jobs): the layer is MaxPool'ed, i.e. its data gradient reads a pooled dz (32-channel chunks). */
int ugn_bf_pack_multi(const float* const* w_hwio_host, uint16_t* const* wpk_host, const int* cin_host, const int* cout_host,
                      const int* dgrad_host, const int* pooled_host, int njobs, void* stream);
int ugn_bf_conv3x3_fwd_multi(const uint16_t* const* in, const uint16_t* const* wpk, uint16_t* const* out, uint8_t* const* out_idx,
                             const int* n, int njobs, int hw, int cin, int cout, int pool, void* stream);
int ugn_bf_conv3x3_dgrad_multi(const uint16_t* const* dz, const uint8_t* const* dz_idx, const uint16_t* const* wpk,
                               const uint16_t* const* act, uint16_t* const* out, const int* n, int njobs, int hw, int cin, int cout,
                               void* stream);
size_t ugn_bf_conv3x3_wgrad_ws(int hw, int cin, int cout);
int ugn_bf_conv3x3_wgrad_multi(const uint16_t* const* in, const uint16_t* const* dz, const uint8_t* const* dz_idx, float* const* dw,
                               const int* n, int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes, void* stream);
int ugn_conv5x5_in_fwd_bf(const float* x, const float* w, uint16_t* a1, uint32_t* a1_sign, int n, int cin, void* stream);
int ugn_conv5x5_in_wgrad_bf(const float* x, const uint16_t* dz1, const uint32_t* a1_sign, float* dw, int n, int cin, void* ws,
                            size_t ws_bytes, void* stream);
int ugn_bf_setmax_fwd_multi(const uint16_t* const* p, const uint16_t* const* addend, uint16_t* const* m, uint16_t* const* sum,
                            const int* b, int njobs, int l, int npix, int c, void* stream);
int ugn_bf_setmax_fwd_f32_multi(const uint16_t* const* p, const uint16_t* const* addend, float* const* m, float* const* sum,
                                const int* b, int njobs, int l, int npix, int c, void* stream);
int ugn_bf_setmax_bwd_multi(const uint16_t* const* p, const void* const* dm, int dm_is_f32, const uint16_t* const* addend,
                            uint16_t* const* out, const int* b, int njobs, int l, int npix, int c, int apply_lrelu, void* stream);
/* routed forms (as ugn_h2_setmax_*_routed_multi): the forward pass also writes the routing words u32 [b][npix][2][c], the gradient
 * reads them instead of the l frames */
int ugn_bf_setmax_fwd_routed_multi(const uint16_t* const* p, const uint16_t* const* addend, uint16_t* const* m, uint16_t* const* sum,
                                   uint32_t* const* route, const int* b, int njobs, int l, int npix, int c, void* stream);
int ugn_bf_setmax_fwd_f32_routed_multi(const uint16_t* const* p, const uint16_t* const* addend, float* const* m, float* const* sum,
                                       uint32_t* const* route, const int* b, int njobs, int l, int npix, int c, void* stream);
int ugn_bf_setmax_bwd_routed_multi(const uint32_t* const* route, const void* const* dm, int dm_is_f32, const uint16_t* const* addend,
                                   uint16_t* const* out, const int* b, int njobs, int l, int npix, int c, int apply_lrelu,
                                   void* stream);
int ugn_bf_lrelu_bwd_multi(const uint16_t* const* g, const uint16_t* const* act, uint16_t* const* out, const size_t* npix, int njobs,
                           int c, void* stream);
int ugn_bf_convert_multi(const float* const* x, uint16_t* const* y, const size_t* n, int njobs, void* stream);
int ugn_hpp_bwd_b4bf_multi(const float* const* a, const float* const* s3, const uint16_t* const* b4, const float* const* dfeat,
                           float* const* dm3, float* const* dzb4, const int* b, int njobs, void* stream);

/* ---- "x3": the 3x3 layers of nets/mj_uwyhNets_ba.py:431-462 on IEEE fp32 tensors, multiplied on the bf16 matrix pipe through the
 * EXACT three-way bf16 split of both operands (x = x0 + x1 + x2, 24 = 8 + 8 + 8 bits, fp32's exponent range): six of the nine
 * partial products per fp32 product, fp32 accumulate; the dropped ones are below 2^-23 of the product, i.e. below the rounding of the
 * accumulation (csrc/x3_common.h; measured error against fp64 at or below an fp32 MFMA chain's).  Tensors, index maps, shapes and
 * epilogues are those of the fp32 entries above (ugn_conv3x3_fwd_wino_multi ...); only the filters are consumed in a packed form.
 * ugn_x3_pack_multi: HWIO fp32 [3,3,cin,cout] -> three bf16 planes in the order the kernels stream them, 54 * cin * cout bytes per
 * job; dgrad = 1 packs the flipped, transposed filter of the data gradient.  Up to 64 (layer, direction) jobs per launch. */
/* The split itself (tests, documentation of the format): planes[k * n + i] = bf16 bit pattern of x_k[i], k = 0, 1, 2, with
 * x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1 (round to nearest even); x0 + x1 + x2 == x exactly for every finite fp32 x. */
int ugn_x3_split(const float* x, uint16_t* planes, size_t n, void* stream);
int ugn_x3_pack_multi(const float* const* w_hwio_host, uint16_t* const* wpk_host, const int* cin_host, const int* cout_host,
                      const int* dgrad_host, int njobs, void* stream);
/* The first layer (ZeroPadding2D(2) + Conv2D(32, 5x5) + LeakyReLU, nets/mj_uwyhNets_ba.py:428-430) and its weight gradient in the same
 * arithmetic: arguments, tensors and the optional sign words exactly as ugn_conv5x5_in_fwd / ugn_conv5x5_in_wgrad (workspace:
 * ugn_conv5x5_in_wgrad_ws), six bf16 products per fp32 product on v_mfma_f32_32x32x16_bf16 instead of the fp32 MFMA. */
int ugn_x3_conv5x5_in_fwd(const float* x, const float* w, float* a1, uint32_t* a1_sign, int n, int cin, void* stream);
int ugn_x3_conv5x5_in_wgrad(const float* x, const float* dz1, const uint32_t* a1_sign, float* dw, int n, int cin, void* ws,
                            size_t ws_bytes, void* stream);
/* out = LeakyReLU(conv(in)) (+ MaxPooling2D(2,2) and first-maximum argmax bytes when pool != 0) for up to 6 jobs of one shape
 * (the frame-level layer and the set-level twin of every modality): Conv2D + LeakyReLU + MaxPooling2D, :431-462. */
/* products: 6 = the default arithmetic; 9 = every partial product of the split operands, i.e. the EXACT product (1.5x the matrix time):
 * for verification -- the two agree to the rounding of the fp32 accumulation (tests/test_x3_gpu.py). */
int ugn_x3_conv3x3_fwd_multi(const float* const* in, const uint16_t* const* wpk, float* const* out, uint8_t* const* out_idx,
                             const int* n, int njobs, int hw, int cin, int cout, int pool, int products, void* stream);
/* Data gradient of the layer cin -> cout at hw x hw (Conv2DBackpropInput; with dz_idx: dz is the POOLED gradient [n,hw/2,hw/2,cout]
 * and is scattered through the argmax bytes while it is staged = MaxPoolGrad; with act [n,hw,hw,cin]: out *= LeakyReLU'(act) =
 * LeakyReluGrad).  All jobs or none take dz_idx / act.  wpk from ugn_x3_pack_multi(dgrad = 1). */
int ugn_x3_conv3x3_dgrad_multi(const float* const* dz, const uint8_t* const* dz_idx, const uint16_t* const* wpk,
                               const float* const* act, float* const* out, const int* n, int njobs, int hw, int cin, int cout,
                               int products, void* stream);
/* Weight gradient dw HWIO [3,3,cin,cout] = sum over images and pixels of in (x) dz (Conv2DBackpropFilter (+ MaxPoolGrad when dz_idx
 * is given)), both operands split in registers while they are staged.  ws: >= ugn_x3_conv3x3_wgrad_ws(hw, cin, cout) bytes for the
 * partial-sum slabs (fixed-order reduction, no atomics: bitwise reproducible). */
size_t ugn_x3_conv3x3_wgrad_ws(int hw, int cin, int cout);
int ugn_x3_conv3x3_wgrad_multi(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx, float* const* dw,
                               const int* n, int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes, int products, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UGAITNET_HIP_H */
