/*
 * ugaitnet_hip_h2.h -- entry points of the OPT-IN f16x2 ("H2") kernel set of libugaitnet_hip.so (gfx950 / MI355X).
 *
 * Not part of the default build (round 6): `python -m ugaitnet_amd.build --h2` (or UGN_BUILD_H2=1) adds conv3x3_mm.hip,
 * wgrad3x3_mm.hip and h2_elem.hip and exports what is declared here; `GaitCore(conv_precision="h2")` needs such a build.  The set
 * holds activations / gradients as two f16 halves + ONE block exponent per tensor (22 significant bits: narrower than the
 * reference's fp32, and what a clip gets depends on the largest clip of its batch), which is why it is neither the default
 * arithmetic nor credited anywhere; the default 3x3 set is "x3" (ugaitnet_hip.h), on IEEE fp32 tensors.
 * Conventions as in ugaitnet_hip.h.
 */
#ifndef UGAITNET_HIP_H2_H
#define UGAITNET_HIP_H2_H

#include "ugaitnet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- "H2" tensors: the 3x3 layers on the f16 matrix pipe at fp32-class accuracy (round 3) ---------------------------
 * Same reference call sites as the fp32 convolutions above (nets/mj_uwyhNets_ba.py:431-462); what changes is how the
 * activations and gradients between those layers are HELD.  An H2 tensor [pixels][c] is stored as f16 bit patterns
 * [pixels][2][c] -- plane 0 = H = f16(x * 2^e), plane 1 = L = f16(x * 2^e - H) -- with a ugn_h2meta {e, amax}:
 * x = (H + L) * 2^-e (22 significant bits); 4 bytes per element like the fp32 tensor it replaces.  Products run as
 * aH*bH + aH*bL + aL*bH on v_mfma_f32_32x32x16_f16 with fp32 accumulation: error <= that of an fp32 FMA chain
 * (tools/probe_split.hip).  `e` is chosen by the producing kernel from a rigorous bound of its output, `amax` (bits of the
 * largest |stored| value) is gathered by the producer with atomicMax and tells the consumer the true range.
 * EVERY ugn_h2meta THAT A KERNEL WRITES MUST BE ZERO ON ENTRY (one hipMemsetAsync over the model's meta array per step). */
typedef struct { int32_t e; uint32_t amax_bits; } ugn_h2meta;
/* packed filter halves of a 3x3 layer: {block exponent, L1 bound, bits of max|w|, unused}; filled by ugn_mm_pack_multi (the
 * kernels read the first two words) */
typedef struct { int32_t e; float l1; uint32_t amax_bits; uint32_t reserved; } ugn_wmeta;
/* meta <- {0, bits(max|x|)} of an fp32 tensor (meta zero on entry): what a kernel that turns fp32 into H2 needs first */
int ugn_absmax(const float* x, size_t n, void* meta, void* stream);
/* fp32 [npix][c] <-> H2 [npix][2][c] (tests, tools, the edges of the H2 part of the path); encode zeroes and fills meta */
int ugn_h2_encode(const float* x, uint16_t* y, void* meta, size_t npix, int c, void* stream);
int ugn_h2_decode(const uint16_t* y, const void* meta, float* x, size_t npix, int c, void* stream);
/* Filters of up to 64 (layer, direction) jobs -> the order the kernels stream them (9*cin*cout halves x 2 planes =
 * 36*cin*cout bytes per job) + their ugn_wmeta.  dgrad = 1 packs the flipped, transposed filter of the data gradient. */
int ugn_mm_pack_multi(const float* const* w_hwio_host, uint16_t* const* wpk_host, void* const* wmeta_host,
                      const int* cin_host, const int* cout_host, const int* dgrad_host, int njobs, void* stream);
/* out = LeakyReLU(conv(in)) (+ MaxPool 2x2 + first-max argmax when pool), up to 6 jobs of one shape per launch; arrays are
 * HOST arrays of device pointers.  in [n][hw][hw][2][cin], out [n][ho][ho][2][cout], out_idx uint8 [n][ho][ho][cout]. */
int ugn_mm_conv3x3_fwd_multi(const uint16_t* const* in, const void* const* in_meta, const uint16_t* const* wpk,
                             const void* const* wmeta, uint16_t* const* out, uint8_t* const* out_idx, void* const* out_meta,
                             const int* n, int njobs, int hw, int cin, int cout, int pool, void* stream);
/* Data gradient of the forward layer cin -> cout at hw x hw.  dz [n][hw][hw][2][cout], or with dz_idx the POOLED gradient
 * [n][hw/2][hw/2][2][cout] + argmax bytes (MaxPool backward while staging).  act (optional, H2 [n][hw][hw][2][cin]):
 * out = conv_transpose(dz, w) * LeakyReLU'(act), the sign taken from act's H plane.  wpk from ugn_mm_pack_multi(dgrad = 1). */
int ugn_mm_conv3x3_dgrad_multi(const uint16_t* const* dz, const uint8_t* const* dz_idx, const void* const* dz_meta,
                               const uint16_t* const* wpk, const void* const* wmeta, const uint16_t* const* act,
                               uint16_t* const* out, void* const* out_meta, const int* n, int njobs, int hw, int cin, int cout,
                               void* stream);

/* Data gradient of the pooled 32 -> 32 layer (a2: Conv2DBackpropInput + MaxPoolGrad, nets/mj_uwyhNets_ba.py:431-434) fused with the
 * weight gradient of the 5x5 first layer (Conv2DBackpropFilter of :428-430, LeakyReluGrad from the a1 sign words): dL/da1 is never
 * written.  dz: dL/dp2 as H2 [n][32][32][2][32] + argmax bytes; x: the network input fp32 [n][60][60][cin], x_meta = {0, bits(max|x|)};
 * dw5[j]: [5][5][cin][32] fp32; scale[j]: a scratch ugn_h2meta record; ws: ugn_mm_dgrad32_wgrad5_ws(njobs) bytes. */
size_t ugn_mm_dgrad32_wgrad5_ws(int njobs);
int ugn_mm_dgrad32_wgrad5_multi(const uint16_t* const* dz, const void* const* dz_meta, const uint8_t* const* dz_idx,
                                const uint16_t* const* wpk, const void* const* wmeta, const float* const* x,
                                const void* const* x_meta, const uint32_t* const* a1_sign, float* const* dw5,
                                void* const* scale, const int* n, const int* cin, int njobs, void* ws, size_t ws_bytes,
                                void* stream);
/* Weight gradient dw HWIO [3,3,cin,cout] (fp32) = sum in (x) dz over images and pixels; in H2 [n][hw][hw][2][cin], dz as in
 * the data gradient (pooled + argmax bytes when dz_idx is given).  ws: >= ugn_mm_conv3x3_wgrad_ws(hw, cin, cout) bytes of
 * scratch for the partial-sum slabs (fixed-order reduction, no atomics: bitwise reproducible). */
size_t ugn_mm_conv3x3_wgrad_ws(int hw, int cin, int cout);
int ugn_mm_conv3x3_wgrad_multi(const uint16_t* const* in, const void* const* in_meta, const uint16_t* const* dz,
                               const uint8_t* const* dz_idx, const void* const* dz_meta, float* const* dw, const int* n,
                               int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes, void* stream);

/* ---- the steps around the 3x3 layers on H2 tensors (same reference lines as their fp32 versions above) ------------------
 * first layer (nets/mj_uwyhNets_ba.py:428-430) with a1 written as H2 [n][64][64][2][32]; x_meta = {0, bits(max|x|)} */
int ugn_conv5x5_in_fwd_h2(const float* x, const void* x_meta, const float* w, uint16_t* a1, void* a1_meta, uint32_t* a1_sign,
                          int n, int cin, void* stream);
/* its weight gradient with dz1 given as H2 [n][64][64][2][32] */
int ugn_conv5x5_in_wgrad_h2(const float* x, const uint16_t* dz1, const void* dz1_meta, const uint32_t* a1_sign, float* dw, int n,
                            int cin, void* ws, size_t ws_bytes, void* stream);
/* the same weight gradient multiplied on the f16 matrix pipe (the default path since round 4): the input patch is split into f16
 * halves of x * 2^ex (x_meta = {0, bits(max|x|)} as for ugn_conv5x5_in_fwd_h2), the gradient's halves are used as stored, and
 * LeakyReLU'(a1) = 0.3 + 0.7 [a1 > 0] is applied as 0.3 * sum + 0.7 * (sum over the pixels whose a1_sign bit is set) */
int ugn_conv5x5_in_wgrad_h2x(const float* x, const void* x_meta, const uint16_t* dz1, const void* dz1_meta, const uint32_t* a1_sign,
                             float* dw, int n, int cin, void* ws, size_t ws_bytes, void* stream);
/* meta[j] <- {0, bits(max|x[j]|)} for up to 6 fp32 tensors (metas zero on entry) */
int ugn_absmax_multi(const float* const* x, const size_t* n, void* const* meta, int njobs, void* stream);
/* fp32 [npix][c] -> H2 for up to 6 tensors; amax_meta[j] from ugn_absmax_multi, meta[j] (another record) is filled */
int ugn_h2_encode_multi(const float* const* x, const void* const* amax_meta, uint16_t* const* y, void* const* meta,
                        const size_t* npix, int njobs, int c, void* stream);
/* set pooling over the l frames of each clip, tf.math.reduce_max(axis=1) (+ Add of the set-level addend), :435,451-452,463-465.
 * p H2 [b*l][npix][2][c]; addend H2 [b][npix][2][c] (optional); m (optional) = maxima, sum = m + addend, both H2. */
int ugn_h2_setmax_fwd_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                            const void* const* addend_meta, uint16_t* const* m, void* const* m_meta, uint16_t* const* sum,
                            void* const* sum_meta, const int* b, int njobs, int l, int npix, int c, void* stream);
/* the same with fp32 outputs [b][npix][c]: the last set pooling feeds HPP */
int ugn_h2_setmax_fwd_f32_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                const void* const* addend_meta, float* const* m, float* const* sum, const int* b, int njobs,
                                int l, int npix, int c, void* stream);
/* its gradient: out = ((p == max ? dm / #maxima : 0) + addend) * (apply_lrelu ? LeakyReLU'(p) : 1).  dm: H2 [b][npix][2][c], or
 * (dm_is_f32) fp32 [b][npix][c] with dm_meta = {0, bits(max|dm|)}.  addend (optional) H2 [b*l][npix][2][c]; out may alias its
 * data, out_meta must be another record than addend_meta. */
int ugn_h2_setmax_bwd_multi(const uint16_t* const* p, const void* const* p_meta, const void* const* dm,
                            const void* const* dm_meta, int dm_is_f32, const uint16_t* const* addend,
                            const void* const* addend_meta, uint16_t* const* out, void* const* out_meta, const int* b, int njobs,
                            int l, int npix, int c, int apply_lrelu, void* stream);
/* Routed set pooling (the default of the f16x2 path since round 4).  The forward pass also writes ROUTING WORDS, u32
 * [b][npix][2][c]: plane 0 = bit t set where frame t of the clip holds the maximum (ties: several bits), plane 1 = bit t set where
 * frame t is positive; l <= 32.  The gradient reads them instead of the l frames -- bit-identical to ugn_h2_setmax_bwd_multi, a
 * third (with an addend) to a half (without) fewer bytes. */
int ugn_h2_setmax_fwd_routed_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                   const void* const* addend_meta, uint16_t* const* m, void* const* m_meta, uint16_t* const* sum,
                                   void* const* sum_meta, uint32_t* const* route, const int* b, int njobs, int l, int npix, int c,
                                   void* stream);
int ugn_h2_setmax_fwd_f32_routed_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                       const void* const* addend_meta, float* const* m, float* const* sum, uint32_t* const* route,
                                       const int* b, int njobs, int l, int npix, int c, void* stream);
int ugn_h2_setmax_bwd_routed_multi(const uint32_t* const* route, const void* const* dm, const void* const* dm_meta, int dm_is_f32,
                                   const uint16_t* const* addend, const void* const* addend_meta, uint16_t* const* out,
                                   void* const* out_meta, const int* b, int njobs, int l, int npix, int c, int apply_lrelu,
                                   void* stream);
/* out = g * LeakyReLU'(act), all H2 [npix][2][c] */
int ugn_h2_lrelu_bwd_multi(const uint16_t* const* g, const void* const* g_meta, const uint16_t* const* act,
                           uint16_t* const* out, void* const* out_meta, const size_t* npix, int njobs, int c, void* stream);
/* HPP backward (nets/mj_uwyhNets_ba.py:468-481) with b4 held as H2 [b][16][16][2][128]: only its sign is used */
int ugn_hpp_bwd_b4h2_multi(const float* const* a, const float* const* s3, const uint16_t* const* b4, const float* const* dfeat,
                           float* const* dm3, float* const* dzb4, const int* b, int njobs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UGAITNET_HIP_H2_H */
