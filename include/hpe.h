/*
 * hpe.h -- C ABI of libhpe_hip.so: the MI355X (gfx950) implementation of the per-image forward hot
 * path of maxpit/human-pose-estimation (ResNet-50 v1 encoder -> 3-stage iterative SMPL-parameter
 * regressor -> SMPL linear blend skinning -> orthographic reprojection).
 *
 * The reference has NO plugin / operator / FFI interface: the Python class
 * `Predictor` (reference: src/predictor.py:26-163) *is* the interface.  This header is therefore the
 * boundary a maintainer's ctypes binding would call from `Predictor.__init__` / `Predictor.predict`;
 * every entry point cites the reference code it replaces.  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - plain C: pointers + sizes, int return codes (HPE_OK == 0), no exceptions cross the ABI,
 *     `hpe_last_error()` returns a thread-local message for the last failing call.
 *   - "host" pointers are read during the call and not retained; "dev" pointers are HIP device
 *     pointers owned by the caller (e.g. torch tensors' data_ptr()).  All tensors are dense float32.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  No call synchronises the
 *     device except hpe_create / hpe_load_* / hpe_finalize / hpe_destroy / hpe_get_timings / hpe_device_status,
 *     and hpe_mesh_loss / hpe_val_losses ONCE when they meet a problem larger than the loss workspace
 *     hpe_finalize sized (max_batch images of 224 x 224, 6890 vertices): that call synchronises, frees the
 *     old workspace and allocates the larger one (such a call cannot be captured into a hipGraph).
 *   - a hpe_finalize that fails on the device (e.g. out of memory) releases everything it had allocated
 *     and leaves the ctx dead: every later call returns HPE_ERR_STATE, only hpe_destroy is valid.
 *   - one ctx per device; a ctx is not re-entrant (the reference is not either: SMPL.J_transformed is
 *     mutated per call, src/tf_smpl/batch_smpl.py:135).
 *   - theta layout (kept from the reference, src/predictor.py:136-138):
 *         [ s, tx, ty | 72 axis-angle (root first) | 10 betas ]  = 85 floats.
 */
#ifndef HPE_H_
#define HPE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define HPE_OK 0
#define HPE_ERR_INVALID 1   /* bad argument / shape */
#define HPE_ERR_HIP 2       /* a HIP runtime call failed */
#define HPE_ERR_STATE 3     /* call order violated (e.g. forward before finalize) */
#define HPE_ERR_NO_DEVICE 4 /* no gfx950 device visible */

#define HPE_NUM_CONV 53      /* ResNet-50 v1 conv layers, order of hpe_conv_layer_name() */
#define HPE_NUM_DENSE 3      /* RegressionNetwork: 2133->1024->1024->85 */
#define HPE_NUM_VERTS 6890
#define HPE_NUM_JOINTS 24
#define HPE_NUM_BETAS 10
#define HPE_NUM_POSE_BASIS 207
#define HPE_THETA_DIM 85
#define HPE_FEATURE_DIM 2048
#define HPE_MAX_KP 24
#define HPE_IMG_SIZE 224

typedef struct hpe_ctx hpe_ctx;

typedef struct HpeConfig {
    int struct_size; /* sizeof(HpeConfig) of the header the caller was built with; written by hpe_config_init, checked by hpe_create
                      * (HPE_ERR_INVALID on a mismatch: the struct has grown every round, a short or zero-initialised one is refused) */
    int device;     /* HIP device ordinal */
    int max_batch;  /* workspace is sized for this many images per call (<= 1024) */
    int num_stage;  /* IEF iterations; reference default 3 (src/config.py:39) */
    float bn_eps;   /* BatchNorm epsilon: 1e-3 (keras_applications 1.0.8) or 1.001e-5 (tf.keras >= 2.2) */
    int encoder_dtype; /* 0 = fp32 MFMA (default), 1 = bf16 MFMA with fp32 accumulate (config 4) */
    /* Plan options that change WHICH kernels run (and therefore the rounding of the result, never its meaning).  -1 = the
     * default: the environment variable named on the right if it is set, else the built-in value.  Fill the struct with
     * hpe_config_init() first; two contexts with different options can live in one process. */
    int n_streams;         /* HPE_STREAMS          concurrent batch-chunk streams of the encoder, 1..4 (2) */
    int dual_gemm;         /* HPE_DUAL             conv_block expand + projection shortcut as one dual-source GEMM (1) */
    int stem_fused;        /* HPE_STEM_FUSED       conv1 + BN + ReLU + max-pool as one kernel (1); 0 = pad / im2col GEMM / pool */
    int wino_min_c;        /* HPE_WINO_MINC        3x3 layers with >= this many channels run as Winograd (128); 0 = direct conv everywhere */
    int wino_min_items;    /* HPE_WINO_MIN_ITEMS   ... when the launch has >= this many work items (128) */
    int wino_fused;        /* HPE_WINO_FUSED       input transform fused into the Winograd GEMM on the large maps (1) */
    int wino_fused_min_hw; /* HPE_WINO_FUSED_MINHW smallest map side on the fused path (28) */
    int mesh_a2b;          /* HPE_MESH_A2B         pixel -> vertex search of the mesh loss: 0 cell grid (default), 1 VALU full
                            *                      search, 2 matrix-core full search */
    int wino_f4;           /* HPE_WINO_F4          map sizes whose 3x3 layers run as Winograd F(4x4,3x3) instead of F(2x2,3x3) / direct: bit mask
                            *                      1 = 7x7, 2 = 14x14, 4 = 28x28, 8 = 56x56 maps (7) */
    int wino4_fused;       /* HPE_WINO4_FUSED      map sizes (4 = 28x28, 8 = 56x56) whose F(4x4) layers transform their input inside the GEMM kernel
                            *                      (no V round trip; takes precedence over wino_f4 for those maps) */
    int bf16_p8;           /* HPE_BF16_P8          bf16 layer kinds on the 256 x 256 phase-interleaved GEMM kernel (N % 256 == 0, K >= 512 only):
                            *                      1 the 3x3 layers of stage 4, 2 those of stage 5, 4 1x1 / strided layers, 8 the dual-source
                            *                      launch of res5a, 16 the other dual-source launches */
    int wino4_ksplit;      /* HPE_WINO4_KSPLIT     small F(4x4) launches cut their channel axis into 2-4 parts that are added in part order
                            *                      (1); 0 = never (one summation order per output whatever the batch) */
    int chain_fuse;        /* HPE_CHAIN            bf16 encoder: stages (1 = stage 2, 2 = stage 3) whose identity blocks run res*_branch2c + add +
                            *                      ReLU and the NEXT block's res*_branch2a + ReLU as one launch: the 4C-wide block output is
                            *                      written once and not read back; 4 = the same for res2a (branch2c + branch1 + add + ReLU, the
                            *                      dual-source GEMM, + res2b_branch2a) (7); same bf16 rounding points as the separate launches.
                            *                      fp32 encoder: 8 = res2b_branch2c + add + ReLU and res2c_branch2a + ReLU as one launch
                            *                      (conv_chain_f32.hip) (8); 16 = bf16 stage 4 too (one 128-pixel workgroup per CU; a
                            *                      measured tie, off); 0 = one launch per layer */
    int halo3;             /* HPE_HALO3            bf16 encoder: map sizes whose 3x3 layers run on the halo-resident kernel (conv3_halo_bf16.hip:
                            *                      the activation tile + halo staged in LDS once per 64 input channels, the 9 taps read it at 9
                            *                      row shifts) instead of the implicit GEMM: bit mask as wino_f4 (1 = 7x7 ... 8 = 56x56) (15); same
                            *                      operands and rounding points, fp32 summation order differs */
} HpeConfig;

/* defaults: struct_size = sizeof(HpeConfig), device 0, max_batch 8, num_stage 3, bn_eps 1e-3, fp32, every plan option -1.
 * ALWAYS start from this call: hpe_create refuses a struct whose struct_size is not the library's. */
void hpe_config_init(HpeConfig* cfg);

/* SMPL constants as the reference holds them after SMPL.__init__ (src/tf_smpl/batch_smpl.py:31-81),
 * as dense host arrays in the pickle's own layouts. */
typedef struct HpeSmplModel {
    const float* v_template;   /* [6890,3] */
    const float* shapedirs;    /* [6890,3,10] */
    const float* posedirs;     /* [6890,3,207] */
    const float* J_regressor;  /* [24,6890]  (the sparse matrix, densified) */
    const float* weights;      /* [6890,24] */
    const float* kp_regressor; /* [num_kp,6890]  cocoplus_regressor (or its first 14 rows for 'lsp') */
    const int* parents;        /* [24] kintree_table[0] as int32, parents[0] == -1, parents[i] < i */
    int num_kp;                /* 19 (cocoplus, the reference default) or 14 (lsp) */
} HpeSmplModel;

/* Device output pointers of one IEF stage.  Any pointer may be NULL (that output is not written). */
typedef struct HpeOutputs {
    float* verts;          /* [B,6890,3]    generated_verts   (src/predictor.py:155) */
    float* joints;         /* [B,num_kp,3]  generated_joints  (src/predictor.py:154) */
    float* cams;           /* [B,3]         generated_cams    (src/predictor.py:156) */
    float* theta;          /* [B,85] */
    float* J_transformed;  /* [B,24,3]      smpl.J_transformed (src/tf_smpl/batch_smpl.py:135) */
    float* kp2d;           /* [B,num_kp,2]  batch_orth_proj_idrot(joints, cams) (src/trainer.py:274) */
    float* verts2d;        /* [B,6890,2]    reproject_vertices(verts, cams, [224,224]) (src/trainer.py:285) */
    float* Rs;             /* [B,24,3,3]    rotation matrices (third return of SMPL.__call__) */
} HpeOutputs;

const char* hpe_last_error(void);
const char* hpe_version(void);

/* Keras layer name of conv `idx` ("conv1", "res2a_branch2a", ... ) and of its BatchNorm. */
const char* hpe_conv_layer_name(int idx);
const char* hpe_bn_layer_name(int idx);
/* geometry of conv `idx`: out[0..6] = KH, KW, Cin, Cout, stride, Hin, Hout */
int hpe_conv_layer_geometry(int idx, int out[7]);

/* -- lifetime: replaces Predictor.__init__ (src/predictor.py:27-86) minus renderer/optimizers/critic -- */
int hpe_create(const HpeConfig* cfg, hpe_ctx** out);
int hpe_destroy(hpe_ctx* ctx);

/* SMPL(pkl_path) constants (src/predictor.py:55 -> src/tf_smpl/batch_smpl.py:26-86). */
int hpe_load_smpl(hpe_ctx* ctx, const HpeSmplModel* host_model);
/* One Keras Conv2D + its BatchNorm, Keras layouts: kernel HWIO [KH,KW,Cin,Cout], bias/gamma/beta/
 * moving_mean/moving_variance [Cout] (EncoderNetwork weights, src/models.py:35-41; restored from the
 * checkpoint's `feature_extractor`, src/predictor.py:79-86). */
int hpe_load_conv(hpe_ctx* ctx, int idx, const float* kernel_hwio, const float* bias, const float* gamma,
                  const float* beta, const float* moving_mean, const float* moving_variance);
/* One Keras Dense of RegressionNetwork: kernel [in,out], bias [out] (src/models.py:60-74). */
int hpe_load_dense(hpe_ctx* ctx, int idx, const float* kernel_in_out, const float* bias);
/* mean theta [85] (load_mean_param, src/predictor.py:88-110 / checkpoint's `inital_theta`). */
int hpe_load_mean_theta(hpe_ctx* ctx, const float* mean85);
/* Pack weights for the kernels, fold BN into per-channel scale/shift, precompute the joint-regressor
 * basis on the device.  Must be called after all hpe_load_* and before any compute call. */
int hpe_finalize(hpe_ctx* ctx);

/* -- the hot path: replaces the body of Predictor.predict (src/predictor.py:114-158) --------------
 * images_dev [B,224,224,3] NHWC float32 in [-1,1].  stage_outs[i] (i < n_outs) receives IEF stage
 * (num_stage - n_outs + i): n_outs == 1 gives the reference's `predict` result (last stage only, SMPL
 * of the earlier stages -- dead work in the reference -- is skipped); n_outs == num_stage gives what
 * Trainer.val_step consumes (src/trainer.py:242-298). */
int hpe_forward(hpe_ctx* ctx, const float* images_dev, int B, const HpeOutputs* stage_outs, int n_outs, void* stream);

/* The same forward, software-pipelined ACROSS calls for steady-state serving: the encoder of this batch is enqueued on `stream`,
 * its regressor + SMPL tail (a chain of small, latency-bound launches) on the ctx's own tail stream behind an event, so that the
 * NEXT call's encoder overlaps it.  The outputs of a call are complete only after hpe_join(ctx, s) has made stream `s` wait for
 * the tail (or after work enqueued on hpe_tail_stream(ctx) itself, e.g. hpe_val_losses or a collective on the outputs).
 * Successive tails are ordered among themselves; the caller must not reuse an output buffer before joining the call that
 * wrote it.  Per-batch latency is that of hpe_forward; throughput gains the tail time (fp32 3 %, bf16 encoder 10 %). */
int hpe_forward_pipelined(hpe_ctx* ctx, const float* images_dev, int B, const HpeOutputs* stage_outs, int n_outs, void* stream);
int hpe_join(hpe_ctx* ctx, void* stream);
/* The regressor + SMPL half of hpe_forward alone: features_dev [B,2048] (what hpe_encoder wrote) -> stage_outs as in hpe_forward
 * (src/predictor.py:126-148).  With hpe_encoder it lets a caller software-pipeline batches inside ONE stream-ordered step that a
 * hipGraph can capture -- fork; hpe_tail(features of batch k) on a side stream || hpe_encoder(images of batch k+1); join --
 * where hpe_forward_pipelined keeps its tail stream outside the caller's ordering and cannot be captured.  Uses the tail's own
 * split-K workspace, so it may run concurrently with hpe_encoder of the same ctx (and with nothing else of it). */
int hpe_tail(hpe_ctx* ctx, const float* features_dev, int B, const HpeOutputs* stage_outs, int n_outs, void* stream);
/* the ctx's tail stream (hipStream_t) for enqueuing consumers of a pipelined call's outputs without stalling `stream` */
void* hpe_tail_stream(hpe_ctx* ctx);

/* -- operators of the path, individually (same kernels as hpe_forward) -------------------------- */
/* image_feature_extractor.predict(images) (src/predictor.py:125): -> features_dev [B,2048] */
int hpe_encoder(hpe_ctx* ctx, const float* images_dev, int B, float* features_dev, void* stream);
/* one IEF step: theta_out = theta_prev + generator3d([features | theta_prev]) (src/predictor.py:129-133).
 * theta_prev_dev == NULL means tile(mean_var) (src/predictor.py:126). */
int hpe_regress_stage(hpe_ctx* ctx, const float* features_dev, const float* theta_prev_dev, int B, float* theta_out_dev,
                      void* stream);
/* self.smpl(shapes, poses, get_skin=True) + proj_fn on theta rows [B,85] (src/predictor.py:136-141). */
int hpe_smpl(hpe_ctx* ctx, const float* theta_dev, int B, const HpeOutputs* outs, void* stream);
/* batch_orth_proj_idrot (src/tf_smpl/projection.py:23-33): X [B,P,3], cam [B,3] -> out [B,P,2] */
int hpe_orth_proj(const float* X_dev, const float* cam_dev, int B, int P, float* out_dev, void* stream);
/* reproject_vertices (src/tf_smpl/projection.py:45-56): out = (proj + 1) * 0.5 * im_size */
int hpe_reproject_vertices(const float* verts_dev, const float* cam_dev, int B, int P, float im_w, float im_h, float* out_dev,
                           void* stream);
/* kp_reprojection_loss (src/ops.py:35-47): kp_gt [B,K,3], kp_pred [B,K,2] -> out_dev[0] = sum(vis*|d|),
 * out_dev[1] = 2*#visible (the SUM_BY_NONZERO_WEIGHTS denominator), out_dev[2] = loss (0 if nothing visible).
 * Numerator and count are returned separately so that ranks can all-reduce them before dividing. */
int hpe_kp_loss(const float* kp_gt_dev, const float* kp_pred_dev, int B, int K, float* out_dev, void* stream);
/* mesh_reprojection_loss (src/ops.py:117-137) forward: seg_dev [B,H,W] (>0 = silhouette), verts2d_dev
 * [B,P,2] pixels -> out_dev[0] = sum_i bidirectional_dist_i / (3 + P).  workspace from the ctx.
 * Nearest neighbours (find_nearest_neighbors, src/ops.py:60-71) are exact: the argmin of the expanded squared distance
 * -2 a.b + |a|^2 + |b|^2 evaluated in fp32, ties to the lowest index as tf.argmin; any H, W, P (the pixel -> vertex search
 * prunes by a cell grid over the vertices when they fit its LDS image, else -- and for meshes concentrated in a few cells --
 * it evaluates every pair; the two give the same neighbours). */
int hpe_mesh_loss(hpe_ctx* ctx, const float* seg_dev, const float* verts2d_dev, int B, int H, int W, int P, float* out_dev,
                  void* stream);

/* Both reprojection losses of all n_stage IEF stages in ONE call -- what Trainer.val_step evaluates per step
 * (src/trainer.py:274-296): the work that depends only on seg_gts (tf.where compaction, src/trainer.py:291;
 * the silhouette bitmap) is done once per call instead of once per stage.
 * kp2d_dev[i] [B,K,2] and verts2d_dev[i] [B,P,2] are host arrays of n_stage device pointers; seg_dev /
 * verts2d_dev may be NULL (keypoint loss only).  out_dev [n_stage][4] = {kp numerator, kp count, kp loss,
 * mesh loss sum}: a rank all-reduces the whole [n_stage][4] block once and re-divides column 0 by column 1. */
int hpe_val_losses(hpe_ctx* ctx, const float* seg_dev, const float* kp_gt_dev, const float* const* kp2d_dev,
                   const float* const* verts2d_dev, int n_stage, int B, int K, int H, int W, int P, float* out_dev, void* stream);

/* -- the steps right before / after the path -------------------------- */
/* preprocess_image (preview.py:18-35) = resize_img + scale_and_crop (src/util/image.py:7-39) + [-1,1] normalisation,
 * fused: img_dev uint8 [H,W,C] (C = 3 or 4, RGB first) -> out224_dev float [224,224,3].
 * proc_param (host, out) = {start_pt.x, start_pt.y, end_pt.x, end_pt.y, img_size}; scale = 224 / max(H, W). */
int hpe_preprocess_u8(const unsigned char* img_dev, int H, int W, int C, float* out224_dev, int proc_param[5], void* stream);
/* The same for a batch of frames in ONE launch.  frames_dev: uint8 frames in one device buffer; frame i starts at byte
 * offsets[i] and is [sizes_hw[2i], sizes_hw[2i+1], C].  offsets == NULL: B frames of one size sizes_hw[0] x sizes_hw[1], back to
 * back (a camera / video stream) -- then no per-image table is needed and table_dev may be NULL.  Otherwise table_dev is a
 * caller-owned device scratch of >= 32 * B bytes; the per-image table is copied into it with a SYNCHRONOUS copy after
 * `stream` has drained (this variant blocks the host and cannot be captured; equal-sized frames never take it).  out_dev [B,224,224,3] float, proc_params (host, out) [B][5] as hpe_preprocess_u8. */
int hpe_preprocess_u8_batch(const unsigned char* frames_dev, const long long* offsets, const int* sizes_hw, int B, int C,
                            float* out_dev, int* proc_params, void* table_dev, void* stream);
/* get_original (src/util/renderer.py:260-283): vert_shifted_dev [B,P,3] = verts + [tx, ty, 500 / (0.5*img_size*s)];
 * cam_for_render (host, out) = {flength/scale, principal point x, y in the original image};
 * kp_original_host [B*K*2] = (joints2d_host + start_pt - img_size/2) / scale (both optional host arrays). */
int hpe_get_original(const float* verts_dev, const float* cam_dev, int B, int P, int K, const int start_pt[2], float scale,
                     int img_size, float* vert_shifted_dev, float cam_for_render[3], float* kp_original_host,
                     const float* joints2d_host, void* stream);

/* -- test / measurement hooks -------------------------------------------------------------------- */
/* Run loaded conv layer `idx` (+BN, optional residual, optional ReLU) on x_dev [B,Hin,Hin,Cin] ->
 * y_dev [B,Hout,Hout,Cout]; for idx 0 the input is the raw [B,224,224,3] image and y is the
 * post-ReLU conv1 output [B,112,112,64].  On a bf16 context (idx > 0 only) x / residual are rounded to bf16 on the way in, the
 * layer runs through the bf16 kernel the plan picks for this batch, and y is its bf16 output widened to float. */
int hpe_debug_conv(hpe_ctx* ctx, int idx, const float* x_dev, int B, const float* residual_dev, int relu, float* y_dev,
                   void* stream);
/* bf16 contexts: the chained launch of conv_chain_bf16.hip alone.  idx2c = res{2,3}{b..}_branch2c of a block that is followed by an
 * identity block: t2_dev [B,H,H,C], residual_dev [B,H,H,4C] (rounded to bf16 on the way in) -> t3_dev [B,H,H,4C] =
 * relu(bn(conv2c(t2)) + residual) and u1_dev [B,H,H,C] = relu(bn(conv2a_next(t3))), both widened to float.  idx2c = res2a_branch2c: the
 * conv_block form, residual_dev is the block INPUT [B,56,56,64] and t3 = relu(bn(conv2c(t2)) + bn(conv1(input))).  occupancy (host,
 * optional, 3 ints): resident workgroups per CU of the three instantiations (the design needs 2).  fp32 contexts: idx2c = res2b_branch2c
 * only (conv_chain_f32.hip), operands in fp32, occupancy[0] = that kernel's. */
int hpe_debug_chain(hpe_ctx* ctx, int idx2c, const float* t2_dev, const float* residual_dev, int B, float* t3_dev, float* u1_dev,
                    int* occupancy, void* stream);
/* The fused stem kernel alone (conv1_pad + conv1 + bn_conv1 + ReLU + pool1_pad + MaxPooling2D(3,2) of the Keras ResNet50,
 * src/models.py:39): images_dev [B,224,224,3] -> y_dev [B,56,56,64].  rows_per_strip: pooled rows per workgroup
 * (1, 2, 4, 7 or 8; 0 = the default for this batch).  fp32 contexts only. */
int hpe_debug_stem(hpe_ctx* ctx, const float* images_dev, int B, int rows_per_strip, float* y_dev, void* stream);
/* Raw GEMM through the conv kernel (dense mode): y[M,N] = act(x[M,K] . wt[n][k]^T (+ residual)), wt has w_rows >= N
 * rounded up to the tile width rows of K floats; K % 32 == 0; tile: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 64x128. */
int hpe_debug_gemm(hpe_ctx* ctx, const float* x_dev, const float* wt_dev, int M, int N, int K, int w_rows, int tile,
                   const float* residual_dev, int relu, float* y_dev, void* stream);
/* Diagnostics builds only (-DHPE_ABLATION): device buffer of 2 x u64 per workgroup that hpe_debug_gemm fills with
 * {shader-clock cycles, 100 MHz ticks} of the main loop; NULL disables. */
int hpe_debug_set_dbg(hpe_ctx* ctx, void* dbg_dev);
/* ZeroPad(1)+MaxPool3x3/2: x [B,H,H,C] -> y [B,H/2,H/2,C];  global average pool x [B,HW,C] -> y [B,C] */
int hpe_debug_maxpool(const float* x_dev, int B, int H, int C, float* y_dev, void* stream);
int hpe_debug_avgpool(const float* x_dev, int B, int HW, int C, float* y_dev, void* stream);
/* generic regressor: out[n,k,c] = sum_v X[n,v,c] * reg[v,k]  (the 6890 -> K joint regressor kernel) */
int hpe_debug_joint_regress(hpe_ctx* ctx, const float* X_dev, int n, int use_kp_regressor, float* out_dev, void* stream);
/* Timing: level 1 brackets the phases of hpe_forward / hpe_encoder with HIP events recorded on `stream`;
 * level 2 additionally brackets every conv launch (adds ~100 event records per call).
 * hpe_get_timings synchronises on the last event and returns milliseconds of the LAST call:
 *   ms[0] encoder (pad + 53 convs + pools), ms[1] sum over the 53 conv launches (level 2, else 0),
 *   ms[2] regressor + SMPL stages, ms[3] reserved (0), ms[4] whole call. */
int hpe_enable_timing(hpe_ctx* ctx, int level);
/* Synchronises `stream` and reports the ctx's device error word: HPE_ERR_HIP if a kernel flagged an invalid result since
 * the last check (today only the opt-in HPE_WINO_STREAMK path can: a bounded inter-workgroup wait that timed out). */
int hpe_device_status(hpe_ctx* ctx, void* stream);
int hpe_get_timings(hpe_ctx* ctx, float ms[5]);
/* Encoder span (first launch to last completion on `stream`, HIP events) over ALL timed hpe_forward* / hpe_encoder calls since
 * hpe_enable_timing (the last 64 at most): ms[0] mean, ms[1] min, ms[2] max; *n_calls = calls averaged.  Synchronises. */
int hpe_get_span_stats(hpe_ctx* ctx, float ms[3], int* n_calls);
/* last timed hpe_val_losses call: ms[0] = whole call, ms[1] = sum over the stages of the pixel -> nearest-vertex search
 * (nn_a2b_mfma_kernel, the dominant kernel of the mesh loss) */
int hpe_get_loss_timings(hpe_ctx* ctx, float ms[2]);
/* Work counter of the pixel -> vertex search: while counter_dev (device, 2 x u64, caller-zeroed) is set, every hpe_val_losses /
 * hpe_mesh_loss call adds the v_mfma_f32_32x32x2_f32 instructions it issues (1024 (pixel, vertex) pairs each) to
 * counter_dev[0] (cell-grid search) and counter_dev[1] (full search).  NULL disables (the default). */
int hpe_debug_set_loss_counter(hpe_ctx* ctx, void* counter_dev);
/* per-conv-layer milliseconds of the last level-2 timed call: ms53[HPE_NUM_CONV] */
int hpe_get_conv_timings(hpe_ctx* ctx, float* ms53);

#ifdef __cplusplus
}
#endif
#endif /* HPE_H_ */
