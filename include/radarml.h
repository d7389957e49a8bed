/*
 * radarml.h -- C ABI of libradarml_hip.so, the MI355X (gfx950) implementation of the
 * radar-ml batched hot path: 3-D radar volume -> (xz, yz, xy) projections -> feature
 * rows -> RBF-SVM / linear decision function -> labels.
 *
 * The reference (goruck/radar-ml) is pure Python and has no FFI of its own; its
 * boundary for this path is three Python call surfaces.  Each entry
 * point below names the reference site it replaces (paths relative to the reference
 * tree; "sk:" = the scikit-learn package the reference calls into).  The Python side
 * binds these with ctypes (radar-ml_amd/_lib.py; INTEGRATION.md shows the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - every function returns RML_OK (0) or a negative rml_status; no exceptions, no
 *     aborts, nothing printed.  rml_last_error() returns a thread-local message.
 *   - all array arguments are CALLER-OWNED DEVICE pointers unless a parameter is
 *     documented "host".  The library never frees caller memory.
 *   - launches are asynchronous on the caller's hipStream_t (passed as void*; NULL =
 *     the default stream).  A context is bound to one device.
 *   - threads and streams: a context may be shared.  The entry points that use the context's workspace, second
 *     stream or caches (rml_svm_decision, rml_svm_kernel_matrix, rml_project_svm, rml_derive_targets,
 *     rml_resize_bicubic) serialise on it: they hold a per-context mutex for the duration of the (asynchronous) call and
 *     make their stream wait for the device work the previous such call queued, whatever stream that was on.  Calls on
 *     different streams are therefore correct but do not overlap on the device; use one context per stream for that.
 *   - layouts: volumes V[b][i][j][k], C-contiguous, (x=theta index i, y=phi index j,
 *     z=range index k); element type float32 (vdtype RML_VOL_F32) or uint8 (RML_VOL_U8: the
 *     radar's native 0..255 magnitudes as stored by the data sets -- a quarter of the HBM
 *     bytes; results are identical to the float32 path on the same values).  Feature rows are [xz | yz | xy] (the tuple
 *     order of common.py:40), each plane C-order, i.e. exactly
 *     np.concatenate((xz, yz, xy), axis=None) of common.py:146.
 */
#ifndef RADARML_H
#define RADARML_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rml_status {
    RML_OK = 0,
    RML_ERR_INVALID = -1,      /* bad argument (NULL pointer, non-positive size, ...) */
    RML_ERR_UNSUPPORTED = -2,  /* shape / mode this build has no kernel for          */
    RML_ERR_HIP = -3,          /* a HIP runtime call failed (see rml_last_error)     */
    RML_ERR_NOMEM = -4,
    RML_ERR_STATE = -5         /* e.g. exact-integer path requested on a model that is not integer valued */
} rml_status;

typedef struct rml_ctx rml_ctx;
typedef struct rml_svm rml_svm;
typedef struct rml_linear rml_linear;

/* projection modes */
#define RML_MODE_MAX   0   /* BASELINE-named max-projection: xz=max_j V, yz=max_i V, xy=max_k V -- np.max, NaN policy included
                              (SURVEY 8 a-1'): a line that holds a NaN gives NaN (round 6: gfx950's IEEE-754-2019 maximum,
                              v_maximum3_f32, in every kernel family at the cost of the maxNum instruction it replaced; rounds 1-5
                              ignored NaNs here).  Radar magnitudes are integers 0..255 (common.py:30-31): the reference path never
                              meets one; tests/test_projection_gpu.py pins the behaviour of every kernel. */
#define RML_MODE_SLICE 1   /* reference-faithful plane slices through (i,j,k): predict.py:102-107       */
#define RML_MODE_SUM   2   /* sum-projection (the reductions of common.py:51-53), float32 accumulation  */
#define RML_MODE_MAX_NAN 3 /* = RML_MODE_MAX since round 6 (kept for callers of rounds 3-5, when NumPy's NaN policy was this opt-in
                              mode on an untuned kernel).
                              Non-finite rows and the SVM: a row with a NaN is off the code grid, so the asynchronous front doors
                              (rml_svm_decision, rml_project_svm) score it on the float64 path, whose RBF epilogue clamps the squared
                              distance with "d2 > 0 ? d2 : 0" -- the row's decision values are finite and meaningless; rows next to
                              it are untouched.  The reference raises instead (predict.py:60 -> sklearn validate_data): the
                              sklearn-protocol classes of radar-ml_amd/svm.py test their input rows and raise the same ValueError
                              (tests/test_svm_gpu.py::test_nan_row_through_the_svm_raises_like_scikit_learn); callers of the bare ABI
                              that may meet NaNs test the row flags / projections themselves. */

/* element type of the volumes */
#define RML_VOL_F32    0
#define RML_VOL_U8     1

/* projection mask bits, positional order of common.ProjMask (common.py:40) */
#define RML_MASK_XZ 1u
#define RML_MASK_YZ 2u
#define RML_MASK_XY 4u
#define RML_MASK_ALL 7u

/* kernel types of the SVC grid in train.py:474-476 */
#define RML_KERNEL_RBF    0
#define RML_KERNEL_LINEAR 1

/* GEMM path selection for rml_svm_decision */
#define RML_PATH_AUTO  0   /* exact-integer path where model and rows are on the code grid, else f64 */
#define RML_PATH_F32   1   /* v_mfma_f32_32x32x2_f32: opt-in, approximate (f32 accumulate, ~1e-4)     */
#define RML_PATH_I8    2   /* v_mfma_i32_32x32x32_i8 on u8 codes, exact int32 dot products            */
#define RML_PATH_F64   3   /* v_mfma_f64_16x16x4_f64 on widened float32 rows: libsvm-class accuracy   */
#define RML_PATH_DIGITS 4  /* general rows as four balanced int8 digits of a 32-bit fixed-point value: ten exact
                              digit-plane products on v_mfma_i32_32x32x32_i8 (|d(u.u)| ~ 1e-8); rows outside the
                              model's fixed-point range take RML_PATH_F64.  RML_PATH_AUTO picks it for large batches */

const char* rml_version(void);
const char* rml_last_error(void);

/* ---- context ------------------------------------------------------------------------- */
int rml_ctx_create(int device, rml_ctx** out);
int rml_ctx_destroy(rml_ctx* ctx);
int rml_ctx_device(const rml_ctx* ctx);

/* Context options (rml_ctx_set_option / rml_ctx_get_option).  The defaults are the tuned configuration and give the documented
 * results; every other value gives THE SAME RESULTS through another kernel family or schedule -- they exist for A/B measurements and
 * for the tests that pin one family against another.  (Until round 5 these were environment variables read inside the launch paths.)
 *   RML_OPT_PROJECT_SHARE_CU (0 / 1, default 0): the stand-alone projection entry points launch their persistent kernels the way
 *     the fused pipeline does beside its GEMM -- one workgroup per CU with a padded LDS request -- so that a kernel the caller runs on
 *     ANOTHER stream finds room on every CU.
 *   RML_OPT_WAVEFRAME (default 1): wave-per-frame projection kernels; 0 off, 2 quarter-plane buffers everywhere, 3 also for short rows
 *     stand-alone.          RML_OPT_LINPLANE (default 1): the linear-plane kernel for rows of 40 / 44 / 48 / 56 quads.
 *   RML_OPT_STAGE_CODES (default 1): per-wave LDS stage of the code rows.   RML_OPT_SLICE_WAVE (default 1): wave-per-row slice kernel.
 *   RML_OPT_DERIVE_FUSED (default 1): one-pass derive -> slice kernel (0: rml_derive_slice runs sum planes + top-k + slices, and
 *     rml_derive_project_svm returns RML_ERR_UNSUPPORTED).
 *   RML_OPT_CODE_RMW (default -1 = the measured rule, rml_code_rmw_default; 0 / 1 forced): read-compare-write code-row stores.
 *   RML_OPT_GEMM_BIG (default -1 = the whole-round rule; 0 never, 1 for every chunk of >= 256 rows): 256 x 256 ring GEMM.
 *   RML_OPT_CHUNK (default 0 = chosen per batch): rows per chunk of the chunked front doors (>= 128; rounded up to 128).
 *   RML_OPT_C1_PK (default 1): packed first-layer kernels of the SGAN branches. */
#define RML_OPT_PROJECT_SHARE_CU 1
#define RML_OPT_WAVEFRAME 2
#define RML_OPT_LINPLANE 3
#define RML_OPT_STAGE_CODES 4
#define RML_OPT_SLICE_WAVE 5
#define RML_OPT_DERIVE_FUSED 6
#define RML_OPT_CODE_RMW 7
#define RML_OPT_GEMM_BIG 8
#define RML_OPT_CHUNK 9
#define RML_OPT_C1_PK 10
int rml_ctx_set_option(rml_ctx* ctx, int option, int value);
int rml_ctx_get_option(const rml_ctx* ctx, int option, int* value);

/* The context's workspace (the chunk buffers of rml_project_svm / rml_svm_decision / rml_derive_targets ...) grows on demand: the
 * first call that needs more than any call before it allocates a bigger block with hipMalloc (the outgrown one is released once the
 * work queued before is done) -- a host-side allocation, not capturable into a HIP graph and possibly synchronising.  Every other
 * call is asynchronous on the caller's stream.  rml_ctx_reserve_workspace(ctx, bytes) does the growth up front (and releases
 * outgrown blocks; it synchronises the device); a warm-up call of the same shape does the same.  A call that would have to grow
 * the workspace while its stream is being captured fails with RML_ERR_INVALID and says so. */
int rml_ctx_reserve_workspace(rml_ctx* ctx, int64_t bytes);
int64_t rml_ctx_workspace_bytes(const rml_ctx* ctx);

/* In-situ timing of the projection launches issued by rml_project_svm (bench.py's roofline line):
 * while enabled, a hipEvent pair is recorded around every projection launch on the stream it is
 * launched on.  rml_profile_read synchronises, returns the number of launches, the summed
 * elapsed time and the frames they covered, and resets the counters. */
int rml_profile_enable(rml_ctx* ctx, int on);
int rml_profile_read(rml_ctx* ctx, int64_t* launches, double* total_ms, int64_t* frames);
/* The same for the GEMM + finish kernels of every chunk (the context's second stream): launches, summed
 * milliseconds, and the algorithmic operations 2*D*M per frame they covered. */
int rml_profile_read_gemm(rml_ctx* ctx, int64_t* launches, double* total_ms, double* ops);

/* What a pure streaming READ of `bytes` resident bytes reaches on this device, in GB/s: `reps` launches of a persistent
 * non-temporal 16-byte-load kernel over `buf` (16-byte aligned, >= 1 MiB), timed with a hipEvent pair on `stream`; synchronises.
 * bench.py's second roofline denominator beside the 8 TB/s specification (SURVEY.md 8d "measured copy bandwidth"); replaces no
 * reference interface -- the reference publishes no bandwidth figure. */
int rml_probe_stream(rml_ctx* ctx, const void* buf, int64_t bytes, int reps, double* gb_per_s, void* stream);

/* 1 when the fused pipelines store a frame's code row read-compare-write (read the old words, store what changed) for frames of
 * frame_bytes volume bytes and D codes, 0 for plain stores (the rule and its measurements: csrc/rml_internal.h rml_code_rmw;
 * RML_OPT_CODE_RMW overrides it per context).  Reporting only: bench.py prints it beside the with/without pair. */
int rml_code_rmw_default(int64_t D, int64_t frame_bytes, int derive, int u8);

/* Feature-row length for a grid and mask: X*Z + Y*Z + X*Y over the selected planes
 * (train_svc.log:19 "Feature vector length: 10010" at (22,31,176)). */
int64_t rml_feature_len(int X, int Y, int Z, uint32_t mask);

/* ---- projection + feature assembly ---------------------------------------------------
 * Replaces, for a batch of B frames, the inline slicing of predict.py:102-107 /
 * ground_truth_samples.py:413-419 (mode SLICE), the max-projection named by
 * BASELINE.json (mode MAX) and the reductions of common.py:51-53 (mode SUM), followed by
 * common.process_samples at zoom 1 (common.py:141-148): mask-select, ravel, concatenate
 * in (xz,yz,xy) order, optional "/ RADAR_MAX" (scale_div = 255.0f; 0 or 1 = no scaling;
 * a true float32 division, bit-identical to NumPy's).
 *
 *   V        B*X*Y*Z float32
 *   ijk      B*3 int32 (mode SLICE only; negative indices wrap like Python's)
 *   feat     B*ld_feat float32 or NULL, row b at feat + b*ld_feat (ld_feat >= D)
 *   feat_q   B*ld_q uint8 or NULL: the same rows as integer codes c = value (before
 *            scaling), stored biased for the signed i8 MFMA (byte = c XOR 0x80 = int8 c-128),
 *            pad columns [D, ld_q) zeroed; valid for a row iff row_flags[b]==1.
 *            Needs ld_q % 4 == 0 and a 4-byte aligned base.
 *   row_isum / row_isq   B int32 / B int64 or NULL: sum c and sum c^2 of each code row
 *   row_flags            B int32 or NULL: 1 iff every selected value of the row is an
 *            integer in [0,255] (the exact-integer SVM path may then be used)
 */
int rml_project(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                const int32_t* ijk, float scale_div, uint32_t mask,
                float* feat, int64_t ld_feat,
                uint8_t* feat_q, int64_t ld_q, int32_t* row_isum, int64_t* row_isq, int32_t* row_flags,
                void* stream);

/* Mode SLICE with T targets per frame: the reference classifies every target of one raw image
 * (`for target in targets:` over one `image`, predict.py:93-119; ground_truth_samples.py:366-440).  Row r = b*T + t of
 * every output is sliced from frame b at ijk[r] -- the volume is read once per target but never duplicated.
 *   ijk      B*T*3 int32;  feat / feat_q / row_* have B*T rows, otherwise as rml_project. */
int rml_project_slices(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int T,
                       const int32_t* ijk, float scale_div, uint32_t mask,
                       float* feat, int64_t ld_feat,
                       uint8_t* feat_q, int64_t ld_q, int32_t* row_isum, int64_t* row_isq, int32_t* row_flags,
                       void* stream);

/* The three planes as separate arrays (B,X,Z) (B,Y,Z) (B,X,Y); any may be NULL.
 * Same modes; no scaling.  (The tuple a reference caller packs at predict.py:113.) */
int rml_project_planes(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                       const int32_t* ijk, float* xz, float* yz, float* xy, void* stream);

/* DerivedTarget.get_derived_targets, common.py:49-80, batched: the three energy
 * profiles (sum over the other two axes) and their top-num_targets indices, ascending
 * by value (argpartition(-n)[-n:] then argsort).  Ties: the LOWER index ranks lower.
 *   ijk      B*num_targets*3 int32   (t-th row = t-th entry of each of the three index lists,
 *                                     zipped as common.py:80)
 *   profiles B*(X+Y+Z) float32 or NULL: [s_theta | s_phi | s_r] per frame
 */
int rml_derive_targets(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                       int num_targets, int32_t* ijk, float* profiles, void* stream);

/* The reference-faithful path in ONE pass over the volumes: derive the num_targets strongest targets of every frame
 * (common.py:49-80, as rml_derive_targets) and slice the three planes through each of them (predict.py:98-107) into feature
 * rows -- what predict.py does per frame when the SDK reports no target and get_derived_targets stands in
 * (ground_truth_samples.py:357).  Row r = b*num_targets + t of every row output belongs to target t of frame b (ascending by
 * energy, like rml_derive_targets).  The frame is streamed once for the three energy profiles (no sum planes through HBM);
 * the planes of its targets are gathered right behind (4*D more bytes per target).
 *   ijk      B*num_targets*3 int32 or NULL (the derived indices, as rml_derive_targets writes them)
 *   profiles B*(X+Y+Z) float32 or NULL
 *   feat, feat_q, row_*   B*num_targets rows, as rml_project
 * Shapes without the fused kernel (rows that are not whole 16-byte quads, Z > 256, odd part of Z/4 above 15, a misaligned V)
 * run rml_derive_targets + rml_project_slices internally.  rml_derive_slice_supported: 1 when the one-pass kernel takes the
 * shape on this context (ctx may be NULL: default options; V may be NULL: alignment not checked). */
int rml_derive_slice(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int num_targets,
                     int32_t* ijk, float* profiles, float scale_div, uint32_t mask,
                     float* feat, int64_t ld_feat,
                     uint8_t* feat_q, int64_t ld_q, int32_t* row_isum, int64_t* row_isq, int32_t* row_flags,
                     void* stream);
int rml_derive_slice_supported(const rml_ctx* ctx, const void* V, int vdtype, int X, int Y, int Z, int num_targets);

/* Assemble feature rows from already separate projection planes (the list-of-tuples
 * input of common.process_samples, common.py:123-149, stacked per plane), zoom 1.
 * Planes are (B,X,Z),(B,Y,Z),(B,X,Y) float32; NULL for a masked-out plane. */
int rml_assemble_features(rml_ctx* ctx, const float* xz, const float* yz, const float* xy,
                          int64_t B, int X, int Y, int Z, float scale_div, uint32_t mask,
                          float* feat, int64_t ld_feat, void* stream);

/* common.process_samples with NON-UNIT zoom (common.py:141-148 when proj_zoom != 1; predict.py:34-54,109-116):
 * scipy.ndimage.zoom(p, zoom) with SciPy's defaults (order-3 spline, mode 'constant', prefilter) on every
 * selected plane, then ravel + concatenate + optional "/ RADAR_MAX".  out_shape (HOST, 6 ints: xz, yz, xy as
 * (rows, cols)) is round(in * zoom) evaluated by the caller with Python's round().  Planes as in
 * rml_assemble_features; feat rows have D = sum of the selected output planes. */
int rml_zoom_features(rml_ctx* ctx, const float* xz, const float* yz, const float* xy,
                      int64_t B, int X, int Y, int Z, const int32_t* out_shape,
                      float scale_div, uint32_t mask, float* feat, int64_t ld_feat, void* stream);

/* Quantise float32 feature rows to uint8 codes + row stats, for callers that bring (N,D)
 * features instead of volumes.  A value v is on the code grid iff it is bit-identical to
 * float32(c / scale_div) for an integer c in [0,255] (scale_div = 255: the "p / 255." of
 * train.py:667; scale_div <= 1: v == c).  flags as in rml_project. */
int rml_quantize_rows(rml_ctx* ctx, const float* feat, int64_t N, int64_t D, int64_t ld_feat,
                      float scale_div, uint8_t* feat_q, int64_t ld_q,
                      int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream);

/* ---- RBF / linear SVC (libsvm C-SVC, one-vs-one) ------------------------------------------
 * rml_svm_load ingests the arrays of a fitted sklearn SVC (HOST pointers, float64, the
 * private libsvm-order attributes): support_vectors_ (M,D), _dual_coef_ (C-1,M),
 * _intercept_ (P=C(C-1)/2), _n_support (C), _gamma; optional sigmoid calibrators
 * a_, b_ (C each) of CalibratedClassifierCV (train.py:722-724).  It builds the device
 * copies both GEMM paths need (float32 centred SVs + norms; uint8 codes when every SV
 * is an integer multiple of 1/code_scale).
 */
int rml_svm_load(rml_ctx* ctx, const double* sv, int64_t M, int64_t D,
                 const double* dual_coef, const double* intercept, const int32_t* n_support,
                 int n_classes, int kernel, double gamma, double code_scale /* e.g. 255, or 1 */,
                 const double* calib_a, const double* calib_b, rml_svm** out);
int rml_svm_free(rml_ctx* ctx, rml_svm* m);
int rml_svm_is_exact(const rml_svm* m);   /* 1 when the uint8-code path is available */
int64_t rml_svm_num_sv(const rml_svm* m);
int64_t rml_svm_dim(const rml_svm* m);

/* SVC.decision_function / SVC.predict / CalibratedClassifierCV.predict_proba / .predict
 * (train.py:217,723-724; predict.py:60) for N feature rows:
 *   sk:svm/src/libsvm/svm.cpp:461-475,514  K = exp(-gamma * ||x - sv||^2)
 *   sk:svm/src/libsvm/svm.cpp:2864-2890    dec[p] = sum coef*K - rho[p]; vote
 *   sk:utils/multiclass.py:542-584         ovr = votes + s/(3(|s|+1))
 *   sk:calibration.py:727-784,928-942      expit(-(a*T+b)), normalise, argmax
 * Inputs: either feat (float32 rows, general path) or feat_q (+row stats, exact path).
 * Outputs (any may be NULL): dec_ovo N*P f64, dec_ovr N*C f64, proba N*C f64,
 * label_vote N int32 (class index of SVC.predict), label_calib N int32 (class index of
 * the calibrated predict; requires calibrators).
 */
int rml_svm_decision(rml_ctx* ctx, const rml_svm* m, int path,
                     const float* feat, int64_t ld_feat,
                     const uint8_t* feat_q, int64_t ld_q, const int32_t* row_isum, const int64_t* row_isq,
                     const int32_t* row_flags,
                     int64_t N,
                     double* dec_ovo, double* dec_ovr, double* proba,
                     int32_t* label_vote, int32_t* label_calib, void* stream);

/* SVC(probability=True).predict_proba (the estimator train.py:478 constructs): libsvm's Platt sigmoid on every
 * pair value followed by pairwise coupling (sk:svm/src/libsvm/svm.cpp:2032-2104, 2918-2952).
 * rml_svm_set_platt: probA/probB HOST, P values each (SVC._probA / _probB); copied to the device once, at load time
 * (synchronous, like rml_svm_load).  rml_svm_pairwise_proba: dec_ovo DEVICE N*P libsvm pair values (dec_ovo of
 * rml_svm_decision); proba DEVICE N*C; an ordinary asynchronous launch on `stream`; RML_ERR_STATE when the model has
 * no Platt coefficients. */
int rml_svm_set_platt(rml_ctx* ctx, rml_svm* m, const double* probA, const double* probB);
int rml_svm_pairwise_proba(rml_ctx* ctx, const rml_svm* m, const double* dec_ovo, int64_t N, double* proba, void* stream);

/* ---- training-time augmentation ---------------------------------------------------------------
 * train.DataGenerator (train.py:84-185) on the GPU, one batch of equally shaped projection planes per call:
 *   RML_AUG_ROTATE  scipy.ndimage.rotate(p, angle, reshape=False) + clamp to [0,1] (train.py:87-94); params: 6 doubles per
 *                   plane, the affine map of ndimage.rotate: m00 m01 m10 m11 off0 off1 with
 *                   [[c, s], [-s, c]], offset = centre - M centre, c/s = cos/sin of the angle in degrees
 *   RML_AUG_ZOOM    clipped_zoom(p, f) + clamp (train.py:96-144); params: 1 double per plane, the zoom factor
 *   RML_AUG_NOISE   sparse_noise: ONE draw per plane added to its non-zero entries + clamp (train.py:146-154); params:
 *                   1 double per plane, the draw
 * src, dst: DEVICE B*H*W float32 (may not alias); params: DEVICE.  The random draws are the caller's (the Python mirror
 * makes them in the reference's order from the reference's generators). */
#define RML_AUG_ROTATE 0
#define RML_AUG_ZOOM   1
#define RML_AUG_NOISE  2
int rml_augment(rml_ctx* ctx, int op, const float* src, int64_t B, int H, int W, const double* params, float* dst, void* stream);

/* Kernel values against the model's support vectors, K[n][m] = k(x_n, sv_m) for m < M, float64, exact to the
 * same arithmetic as rml_svm_decision (path AUTO / I8 / F64).  With the training rows loaded as the "support vectors"
 * of a model (any coefficients) this is the Gram matrix sklearn's SVC(kernel='precomputed') is fitted on and
 * K(test, train) it predicts with -- the rbf evaluations libsvm would otherwise do one pair at a time
 * (train.py:462-491 grid search).  kmat: N x ld_k doubles, ld_k >= M. */
int rml_svm_kernel_matrix(rml_ctx* ctx, const rml_svm* m, int path, const float* feat, int64_t ld_feat, int64_t N,
                          double* kmat, int64_t ld_k, void* stream);

/* Fused front door: volumes -> projection (mode, mask fixed at load: D must match) ->
 * SVM outputs, features never returned to the caller.  Workspace is owned by the ctx and
 * grows on demand. */
int rml_project_svm(rml_ctx* ctx, const rml_svm* m, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                    int mode, const int32_t* ijk, float scale_div, uint32_t mask,
                    double* dec_ovo, double* dec_ovr, double* proba,
                    int32_t* label_vote, int32_t* label_calib, void* stream);

/* The same front door for the reference-faithful projection with DERIVED targets: per frame the strongest derived target
 * (common.py:49-80, num_targets = 1), the three slices through it (predict.py:98-107) and the SVM outputs, the frame read once
 * (+ 4*D bytes for the slices).  ijk_out: B*3 int32 or NULL (the derived indices).  RML_ERR_UNSUPPORTED when
 * rml_derive_slice_supported(ctx, V, ...) is 0: call rml_derive_targets and rml_project_svm(RML_MODE_SLICE, ijk) then. */
int rml_derive_project_svm(rml_ctx* ctx, const rml_svm* m, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                           float scale_div, uint32_t mask, int32_t* ijk_out,
                           double* dec_ovo, double* dec_ovr, double* proba,
                           int32_t* label_vote, int32_t* label_calib, void* stream);

/* ---- linear classifier (SGDClassifier(loss='log'), train.py:350-381; predict 421,433) --- */
int rml_linear_load(rml_ctx* ctx, const double* coef /* host (C,D) */, const double* intercept /* host C */,
                    int n_classes, int64_t D, const double* calib_a, const double* calib_b, rml_linear** out);
int rml_linear_free(rml_ctx* ctx, rml_linear* m);
int rml_linear_decision(rml_ctx* ctx, const rml_linear* m, const float* feat, int64_t ld_feat, int64_t N,
                        double* dec /* N*C */, double* proba /* N*C or NULL */, int32_t* label /* argmax dec */,
                        int32_t* label_calib, void* stream);

/* ---- PIL bicubic resize in front of the dnn / sgan classifiers ----------------------------------------------
 * Image.fromarray(p).resize((out_w, out_h), resample=Image.BICUBIC) of float32 planes (dnn.py:240-245,
 * sgan.py:676-681; Pillow's Resample.c: antialiased, double-precision taps, float32 intermediate), bit-identical
 * to Pillow.  When div != 0 the reference's [-1,1] scaling (dnn.py:202-205) is applied first: v = (p - sub) / div
 * (sub = div = RADAR_MAX/2 = 127.5).  in: B planes of H x W float32, in_stride floats from one sample to the next
 * (a plane inside a feature row [xz|yz|xy] is addressed directly); out: B x out_h x out_w contiguous, float32
 * (out_bf16 = 0) or bf16 (out_bf16 = 1, the operand type of rml_dnn_trunk). */
int rml_resize_bicubic(rml_ctx* ctx, const float* in, int64_t in_stride, int64_t B, int H, int W, int out_h, int out_w,
                       float sub, float div, void* out, int out_bf16, void* stream);
/* ---- the same preprocessing for the bf16 conv trunk, all three projections of a feature row in ONE launch ------------
 * dnn.py:200-254 ((p - 127.5) / 127.5, then the bicubic resize of xz (X x Z), yz (Y x Z) and xy (X x Y) to out_h x out_w) with
 * bf16 outputs (the operand type of rml_dnn_trunk).  Pillow's windows and normalised weights -- the tables of
 * rml_resize_bicubic -- applied in float32 (fused multiply-adds, partial sums): NOT bit-identical to Pillow, within ~1e-6 of it
 * before the bf16 rounding (the bf16 value differs from the rounded exact one by at most one bf16 ulp on a small fraction of
 * the pixels); rml_resize_bicubic stays the Pillow-exact surface.  Input per row: the float32 feature row [xz | yz | xy]
 * (rows, ld floats apart) and / or the biased uint8 code row rml_project writes (codes, ldq bytes apart, 16-byte aligned);
 * flags[b] != 0 selects the code row of row b (the per-row "every value is an integer in [0, 255]" flag of rml_project), 0 its
 * float row; a row whose kind was not passed is left untouched (flags with ONE kind of row: only the rows of that kind are
 * written); flags = NULL: every row, from the one kind given.  Outputs: (B, out_h, out_w) bf16, 8-byte aligned.
 * Shapes: Z % 16 == 0, out_w % 4 == 0, out_w <= 256, out_h >= X and >= Y (no vertical shrink), (X + Y) * Z <= 16 384,
 * X * Y <= 4 096 -- rml_dnn_preprocess_supported says; others: RML_ERR_UNSUPPORTED (rml_resize_bicubic per projection). */
int rml_dnn_preprocess_supported(int X, int Y, int Z, int out_h, int out_w);
int rml_dnn_preprocess_rows(rml_ctx* ctx, const float* rows, int64_t ld, const uint8_t* codes, int64_t ldq, const int32_t* flags,
                            int64_t B, int X, int Y, int Z, int out_h, int out_w, uint16_t* xz, uint16_t* yz, uint16_t* xy,
                            void* stream);
/* Volumes -> trunk inputs (the front of BASELINE configs[3]): projection (mode as rml_project; ijk for RML_MODE_SLICE) into code rows
 * + row flags (no float rows through HBM), for float32 volumes a float-row pass predicated ON THE DEVICE on "some row left the
 * code grid" (it exits at once for radar data, integers 0..255: common.py:30-31) -- queued on the context's second stream beside
 * the code rows' preprocessing, joined before the call returns its stream --, then rml_dnn_preprocess_rows.  Scratch is the
 * caller's: codes B x ldq bytes (ldq >= D, % 16 == 0), flags B + 1 int32 (flags[B]: every row on the grid), rows B x ld float32
 * (float32 volumes only; NULL for uint8 volumes, which cannot leave the grid). */
int rml_dnn_preprocess_volumes(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode, const int32_t* ijk,
                               uint8_t* codes, int64_t ldq, int32_t* flags, float* rows, int64_t ld, int out_h, int out_w,
                               uint16_t* xz, uint16_t* yz, uint16_t* xy, void* stream);
/* ---- dnn.py convolutional trunk (dnn.py:45-52,68-76), fused ----------------------------------------------
 * Per branch Conv2D(1->64,3x3,s2,'same',relu) -> Conv2D(64->32,3x3,s2,'same',relu); the three branches concatenated
 * on channels and flattened NHWC: feat[b][(h*(W/4)+w)*96 + branch*32 + n], bf16.  Inputs (B,H,W) already scaled to
 * [-1,1] and resized (dnn.py:200-254; rml_resize_bicubic), float32 (in_bf16 = 0; H, W multiples of 4; rounded to
 * bf16 on load) or bf16 (in_bf16 = 1; W a multiple of 8): the results are identical.  Weights (DEVICE): w1 [3][64][9] float32
 * (tap = ky*3+kx), b1 [3][64], w2t [3][32][576] bf16 with k = (ky*3+kx)*64 + cin, b2 [3][32]. */
int rml_dnn_trunk(rml_ctx* ctx, const void* xz, const void* yz, const void* xy, int in_bf16, int64_t B, int H, int W,
                  const float* w1, const float* b1, const uint16_t* w2t, const float* b2,
                  uint16_t* feat, void* stream);
/* The same values in the layout rml_dnn_dense_tail streams best ("K-block"): feat[kb][b][64] bf16 with the K axis ordered (branch,
 * pixel, channel) -- element (branch, pixel, channel) of sample b at block kb = (branch * P + pixel) / 2, offset (pixel & 1) * 32 +
 * channel, P = (H/4) * (W/4) even.  A 128-sample tile of one K-step is then 16 KB of contiguous memory (row-major rows put those
 * 128 pieces of 128 B 76.8 KB apart: every piece its own DRAM page, 3.3-3.6 TB/s for every GEMM that was tried on it).
 * RML_ERR_UNSUPPORTED for planes the register-resident trunk kernel does not take. */
int rml_dnn_trunk_kblock(rml_ctx* ctx, const void* xz, const void* yz, const void* xy, int in_bf16, int64_t B, int H, int W,
                         const float* w1, const float* b1, const uint16_t* w2t, const float* b2,
                         uint16_t* feat, void* stream);

/* The same two convolutions at float32-class accuracy (csrc/dnn_x3.hip, round 6) -- the reference's model.predict is float32 Keras
 * (dnn.py:373-381): every operand is carried as `parts` bf16 numbers and every product as the matrix-core products of the part
 * pairs (i, j) with i + j < parts, float32 accumulation.  parts = 2 ("x3": 16 significant bits, three products, ~2^-16 relative
 * per product where rml_dnn_trunk has bf16's 2^-9; about 3.5 x the time of rml_dnn_trunk per sample) or 3 ("x6": 24 bits, six
 * products, what is dropped is the size of float32's own rounding; about twice x3).  Planes float32 (rml_resize_bicubic's
 * Pillow-bit-identical output), weights float32: w1 [3][64][9], b1 [3][64], w2 [3][32][576] (k = (ky*3+kx)*64 + cin; NOT transposed
 * to bf16), b2 [3][32]; feat[b][(h*(W/4)+w)*96 + branch*32 + n] float32 (Keras' Flatten order).  Planes, w2 and feat 16-byte aligned.
 * The second opinion of the margin guard (radar-ml_amd/dnn.py), not the fast path.
 * rml_dnn_trunk_x3_supported: H, W multiples of 4 and three float32 planes + 72 KB of weight fragments (one plane + 108 KB for
 * parts = 3) within the 160 KB LDS -- the 80 x 80 of dnn.py:33 does; otherwise RML_ERR_UNSUPPORTED. */
int rml_dnn_trunk_x3_supported(int H, int W);
int rml_dnn_trunk_x3(rml_ctx* ctx, const float* xz, const float* yz, const float* xy, int64_t B, int H, int W,
                     const float* w1, const float* b1, const float* w2, const float* b2, int parts, float* feat, void* stream);

/* Volumes -> float32-class features in one call (the re-scoring front of the margin guard): frames rows[0..n) of V (rows = NULL: the
 * first n frames) gathered into scratch, projected exactly (rml_project, float32 rows, mode MAX / SUM / MAX_NAN), scaled to [-1, 1]
 * and resized with rml_resize_bicubic (Pillow-bit-identical float32 planes), then rml_dnn_trunk_x3 with `parts`.  scratch:
 * rml_dnn_exact_features_scratch_bytes(vdtype, n, X, Y, Z, out_h, out_w, rows != NULL) bytes, 256-byte aligned. */
int64_t rml_dnn_exact_features_scratch_bytes(int vdtype, int64_t n, int X, int Y, int Z, int out_h, int out_w, int gathered);
int rml_dnn_exact_features(rml_ctx* ctx, const void* V, int vdtype, const int64_t* rows, int64_t n, int X, int Y, int Z, int mode,
                           int out_h, int out_w, const float* w1, const float* b1, const float* w2, const float* b2, int parts,
                           void* scratch, int64_t scratch_bytes, float* feat, void* stream);

/* ---- the margin guard's device side (radar-ml_amd/dnn.py Classifier._guard; labels of model.predict, dnn.py:373-381) --------------
 * rml_dnn_top2_gap: gap[r] = largest - second largest of proba[r][0..C) (rows ld floats apart, 2 <= C <= 16); 0 for a row that
 * holds a non-finite value (it counts as a tie).
 * rml_dnn_guard_apply: one re-scoring round's bookkeeping in one launch -- proba[rows[i]] <- fresh[i] (fresh: n x C contiguous),
 * stats[0] = atomic max over the rows where both are finite of |old - new| as float32 BITS (non-negative floats order like unsigned
 * integers; zero stats before the call), stats[1] += rows whose new top-2 gap is below thr_close, close[i] = that test,
 * gap[rows[i]] = +inf when gap is given (re-scored: never a candidate again). */
int rml_dnn_top2_gap(rml_ctx* ctx, const float* proba, int64_t ld, int64_t N, int C, float* gap, void* stream);
int rml_dnn_guard_apply(rml_ctx* ctx, float* proba, int64_t ld, int C, const int64_t* rows, int64_t n, const float* fresh,
                        float thr_close, float* gap, uint32_t* stats, uint8_t* close, void* stream);

/* ---- dnn.py dense tail (dnn.py:78-88), fused -----------------------------------------------------------------
 * Dense 64 relu -> Dense 64 relu -> Dense n_classes softmax on the bf16 feature rows of rml_dnn_trunk (Dropout is inactive at
 * inference): proba[b][c] float32.  The first layer runs on the bf16 matrix cores (float32 accumulation, split-K with the partial
 * sums added in a fixed order: deterministic), the two small layers and the softmax in float32.  feat: N rows of K bf16, ld_feat
 * elements apart (kblock = 0; K % 64 == 0, ld_feat % 8 == 0, 16-byte aligned), or the [K/64][N][64] layout of rml_dnn_trunk_kblock
 * (kblock = 1: w1 is then blocked the same way, [K/64][64 out][64], the K axis in the (branch, pixel, channel) order); w1 [64][K]
 * bf16 (kblock = 0: torch Linear layout, out x in), b1 [64];
 * w2t [64 in][64 out] float32 -- the TRANSPOSE of torch's (out, in) = the Keras kernel layout (in, out) --, b2 [64]; w3
 * [n_classes][64] float32 (torch layout), b3 [n_classes]; n_classes <= 16.  workspace: rml_dnn_dense_workspace_bytes(ctx, N, K)
 * bytes of device memory, 16-byte aligned (the split-K partial sums). */
int64_t rml_dnn_dense_workspace_bytes(rml_ctx* ctx, int64_t N, int64_t K);
int rml_dnn_dense_tail(rml_ctx* ctx, const uint16_t* feat, int64_t ld_feat, int kblock, int64_t N, int64_t K, const uint16_t* w1, const float* b1,
                       const float* w2t, const float* b2, const float* w3, const float* b3, int n_classes, float* workspace,
                       int64_t workspace_bytes, float* proba, void* stream);

/* The same tail on FLOAT32 feature rows (feat [N][ld_feat], w1 [64][K] float32 in torch's Linear layout, the other operands as
 * above): the last stage of the margin guard's re-scoring (radar-ml_amd/dnn.py Classifier._tail_float32; Keras runs these layers in
 * float32, dnn.py:78-88).  Its result for a row depends on that row alone -- the K axis is cut into splits that are a function of
 * K only, every partial sum is one fma chain in ascending k, the splits are added in order -- so a row scored alone, inside any
 * candidate set, or twice has the same bits (a library float32 GEMM that splits K with atomics does not).  K and ld_feat multiples
 * of 4; workspace: rml_dnn_dense_tail_f32_workspace_bytes(N, K) bytes, 16-byte aligned. */
int64_t rml_dnn_dense_tail_f32_workspace_bytes(int64_t N, int64_t K);
int rml_dnn_dense_tail_f32(rml_ctx* ctx, const float* feat, int64_t ld_feat, int64_t N, int64_t K, const float* w1, const float* b1,
                           const float* w2t, const float* b2, const float* w3, const float* b3, int n_classes, float* workspace,
                           int64_t workspace_bytes, float* proba, void* stream);

/* ---- SGAN discriminator branches: fused BatchNorm(train) + LeakyReLU + 'same' pad (sgan.py:137-158) -------------
 * x: N x H x W x C (NHWC, dense) float16 (dtype 0) or bfloat16 (dtype 1), the convolution output; y: N x (H+pad_h) x
 * (W+pad_w) x C, the zero-padded input of the next stride-2 'same' convolution (pad 0: plain output).  Batch
 * statistics in float32/float64 (deterministic), running estimates updated as torch.nn.BatchNorm2d does
 * (momentum = 1 - Keras momentum).  workspace: rml_bn_workspace_floats(ctx, C) + 2*C floats (all three entry points).  backward: dy has y's
 * layout; returns dx (x's layout) and the float32 parameter gradients. */
int64_t rml_bn_workspace_floats(rml_ctx* ctx, int C);
int rml_bn_lrelu_pad_forward(rml_ctx* ctx, const void* x, int dtype, int64_t N, int H, int W, int C, int pad_h, int pad_w,
                             const float* gamma, const float* beta, float eps, float momentum, float slope,
                             float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                             float* workspace, void* y, void* stream);
int rml_bn_lrelu_pad_backward(rml_ctx* ctx, const void* x, const void* dy, int dtype, int64_t N, int H, int W, int C,
                              int pad_h, int pad_w, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_rstd, float slope, float* workspace, void* dx, float* dgamma, float* dbeta,
                              void* stream);

/* First layer of a branch, [3x3 stride-2 'same' convolution of a 1-channel image] + BatchNorm + LeakyReLU + pad with the
 * convolution folded in (the image needs no gradient): the convolution output is never stored; every pass recomputes
 * it (9 FMAs per element) from the zero-padded half-precision image N x (2H+1) x (2W+1) and the float32 weights
 * weight[tap][c].  forward: image -> y; backward: image, dy -> dweight[tap][c], dgamma, dbeta. */
int rml_conv1_bn_lrelu_pad_forward(rml_ctx* ctx, const void* image, const float* weight, int dtype, int64_t N, int H, int W,
                                   int C, int pad_h, int pad_w, const float* gamma, const float* beta, float eps,
                                   float momentum, float slope, float* running_mean, float* running_var, float* save_mean,
                                   float* save_rstd, float* img_stats /* 54 floats out: sums of the taps and of their products */,
                                   float* workspace, void* y, void* stream);
int rml_conv1_bn_lrelu_pad_backward(rml_ctx* ctx, const void* image, const float* weight, const void* dy, int dtype, int64_t N,
                                    int H, int W, int C, int pad_h, int pad_w, const float* gamma, const float* beta,
                                    const float* save_mean, const float* save_rstd, const float* img_stats /* from forward */,
                                    float slope, float* workspace, float* dweight, float* dgamma, float* dbeta, void* stream);

/* ---- Adam update of the SGAN discriminator (sgan.py:206, 214: Adam(lr=0.0002, beta_1=0.5) inside train_on_batch, sgan.py:525-532) ----
 * ONE pass over all parameters with the loss-scale bookkeeping of a half-precision step on the device (csrc/optim.hip).
 * table: DEVICE array of n_tensors + 1 records of rml_adam_entry_bytes() bytes each,
 *     { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t start; }
 * start = the tensor's first element in the concatenation of all tensors, record n_tensors holds { 0, 0, 0, 0, total }.  The four
 * tensors of a record are dense float32 with identical strides (the update runs over the storage order).
 * step: device float, the number of updates applied so far (incremented here unless the step is skipped).
 * scale: device float loss scale, or NULL (no scaling: gradients taken as they are, inv_scale = 1).  With check != 0 every gradient is
 * tested first; a non-finite one skips the whole update and multiplies the scale by `backoff`, `growth_interval` clean steps in
 * a row multiply it by `growth` (torch.amp.GradScaler's rule).  state: device int32[3] (non-finite count, growth tracker, skip flag
 * of the last call), zero-initialised by the caller once.  inv_scale: device float scratch.
 * The update is torch.optim.Adam's: m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2, p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps). */
int rml_adam_entry_bytes(void);
int rml_adam_step(rml_ctx* ctx, const void* table, int n_tensors, int64_t total, float lr, float beta1, float beta2, float eps,
                  float* step, float* scale, int32_t* state, float* inv_scale, int check, float growth, float backoff,
                  int growth_interval, void* stream);

/* ---- synthetic data (bench / tests) --------------------------------------- */
int rml_synth_volumes(rml_ctx* ctx, uint64_t seed, int64_t frame0, int64_t B, int X, int Y, int Z,
                      int n_classes, float* V, int32_t* cls /* B or NULL */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RADARML_H */
