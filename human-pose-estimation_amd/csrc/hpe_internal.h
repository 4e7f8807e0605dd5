// hpe_internal.h -- shared declarations of libhpe_hip.so (not part of the C ABI; see include/hpe.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// GEMM_DUAL: two A sources summed into one accumulator (K concatenated): k-slabs [0, k1_slabs) come from the dense matrix x
// (row pitch lda), the rest from the strided NHWC tensor x2 (geometry in Hi / Wi / Cin / Ho / Wo / stride) -- the last 1x1
// convolution of a ResNet conv_block and its projection shortcut as ONE launch (weights concatenated along k, BN scales folded in)
enum GemmMode { GEMM_DENSE = 0, GEMM_STRIDED = 1, GEMM_CONV3 = 2, GEMM_STEM = 3, GEMM_DUAL = 4 };
enum GemmTile { TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x64 = 2, TILE_64x128 = 3, TILE_128x128_W8 = 4, TILE_128x64_W8 = 5, TILE_256x128_W8 = 6,
                TILE_P8_256x256 = 7 /* bf16 only: the phase-interleaved 8-wave kernel of conv_gemm_bf16_p8.hip */ };

// Arguments of the implicit-GEMM kernel (conv_gemm.hip).  All offsets are in floats.
struct GemmArgs {
    const float* x;      // A source: activations (NHWC) or a dense [M, lda] matrix
    const float* w;      // packed weights Wt[n][k], row pitch ldw, w_rows rows (zero padded)
    const float* scale;  // [N]
    const float* shift;  // [N]
    const float* res;    // optional residual [M, ldres]
    float* y;            // [M, ldy]
    int M, N, K;         // K is a multiple of 32 (weights are zero padded along k)
    int lda, ldw, ldy, ldres, w_rows;
    int relu;
    int Hi, Wi, Cin, Ho, Wo, stride, cin_slabs;
    int n_mtiles, n_ntiles;  // filled by the launcher
    int split_k;             // filled by the launcher (> 1: small grid cut along K, reduced by the fix-up kernel)
    float* partial;          // split-K workspace of the launching stream, partial_floats floats; nullptr disables splitting
    size_t partial_floats;
    const float* zero;       // >= 16 B of zeros in global memory (source of out-of-image taps for the LDS-DMA path)
    unsigned long long* dbg;  // diagnostics only (HPE_ABLATION builds): per-workgroup {shader clocks, 100 MHz ticks}
    const float* x2;  // GEMM_DUAL: second A source (strided NHWC)
    int k1_slabs;     // GEMM_DUAL: k-slabs taken from x
    int res_prefetch;  // filled by the launcher: 8-wave tiles read their residual rows before the main loop (HPE_RES_PREFETCH=0: in the epilogue)
    int y_slab8;  // 1: y is written channel-slab major, y[(n / 8) * M + m][n % 8] (the layout the fused Winograd kernel reads); needs N % 8 == 0
};

hipError_t hpe_launch_gemm(GemmArgs p, int mode, int tile, hipStream_t st);

// conv_wino.hip: 3x3/s1 SAME convolution as Winograd F(2x2,3x3); U = G g G^T blocked [N/64][C/8][16][2][64][4], V workspace of
// hpe_wino_v_floats(B, H, W, C) floats; C % 32 == 0, N % 64 == 0
size_t hpe_wino_v_floats(int B, int H, int W, int C);
hipError_t hpe_wino_init_device();  // per-device kernel attributes; call with the target device current
// optional persistent stream-K scheduling of the GEMM: n_wg workgroups (one per CU), ws = n_wg * HPE_WINO_WS_FLOATS floats and
// flags = n_wg zero-initialised words owned by the launching stream, epoch unique per launch (never 0)
#define HPE_WINO_WS_FLOATS 65536
struct WinoStreamK {
    float* ws;
    unsigned* flags;
    unsigned epoch;
    int n_wg;
    unsigned* err;  // device error word (bit 0 set when a wait timed out), may be nullptr
};
hipError_t hpe_launch_wino_conv3(const float* x, int lda, const float* U, const float* scale, const float* shift, float* y, int ldy,
                                 int B, int H, int W, int C, int N, int relu, float* V, const WinoStreamK* sk, hipStream_t st);

// fused-transform variant (56x56 / 28x28 maps): xs is channel-slab major [C/8][B*H*W][8] (GemmArgs::y_slab8 of the producer)
hipError_t hpe_launch_wino_fused_conv3(const float* xs, const float* U, const float* scale, const float* shift, const float* zero16, float* y,
                                       int ldy, int B, int H, int W, int C, int N, int relu, hipStream_t st);
int hpe_wino_fused_items(int B, int H, int W, int N);  // work items of that launch, 0 if the geometry is not supported
hipError_t hpe_launch_nhwc_to_slab8(const float* x, float* xs, long M, int C, hipStream_t st);

// conv_wino4.hip: the same convolution as Winograd F(4x4,3x3); U = G g G^T blocked [N/64][C/4][36][64][4], V workspace of
// hpe_wino4_v_floats(B, H, W, C) floats (blocked [tiles/32][C/4][36][32][4]); C % 32 == 0, N % 64 == 0; co_running = batch chunks
// launching the same layer on other streams at the same time (picks between the 64- and the 32-cout GEMM)
size_t hpe_wino4_v_floats(int B, int H, int W, int C);
int hpe_wino4_items(int B, int H, int W, int N);  // workgroups of the GEMM launch
hipError_t hpe_wino4_init_device();
hipError_t hpe_launch_wino4_conv3(const float* x, int lda, const float* U, const float* scale, const float* shift, float* y, int ldy, int B,
                                  int H, int W, int C, int N, int relu, float* V, hipStream_t st, int co_running = 1, float* split_ws = nullptr);
// split_ws: hpe_wino4_split_ws_floats() floats whose LAST 256 (unsigned counters) are zero, owned by launches that are alone on the device
// (nullptr: the C axis is never cut)
size_t hpe_wino4_split_ws_floats();

// fused-transform F(4x4,3x3) (56x56 / 28x28 maps): xs is channel-slab major [C/8][B*H*W][8] (GemmArgs::y_slab8 of the producer)
hipError_t hpe_launch_wino4_fused_conv3(const float* xs, const float* U, const float* scale, const float* shift, const float* zero16, float* y,
                                        int ldy, int B, int H, int W, int C, int N, int relu, hipStream_t st);
int hpe_wino4_fused_items(int B, int H, int W, int N);  // workgroups of that launch, 0 if the geometry is not supported

// conv_gemm_bf16.hip (x / w / res / y of GemmArgs point to bf16 data; offsets are in bf16 elements; K % 64 == 0)
hipError_t hpe_launch_gemm_bf16(GemmArgs p, int mode, int tile, int ring_depth, hipStream_t st);  // ring_depth 2..4 LDS slab buffers
// conv_gemm_bf16_p8.hip: 256 x 256 x 64 tile, 8 waves, phase-interleaved main loop, split-K through p.partial (DENSE / STRIDED / CONV3 / DUAL)
hipError_t hpe_launch_gemm_bf16_p8(GemmArgs p, int mode, hipStream_t st);
// conv_chain_bf16.hip: t3 = relu(bn(W2c . t2) + res) and u1 = relu(bn'(W2a' . t3)) in one launch (the last 1x1 of identity block i and the
// first 1x1 of identity block i + 1); all tensors bf16, row-major [M, channels], weights packed [n][k]
struct ChainArgs {
    const __bf16* t2;   // [M, C]
    const __bf16* res;  // [M, 4C]  (identity block; nullptr in the conv_block form)
    const __bf16* x2;   // [M, C2]  conv_block form: the block input, second A source of the expand GEMM (nullptr otherwise)
    const __bf16* w2c;  // [4C][ldw2c]
    const __bf16* w2a;  // [C'][ldw2a]
    const float *scaleA, *shiftA;  // [4C]
    const float *scaleB, *shiftB;  // [C']
    __bf16* t3;  // [M, 4C]
    __bf16* u1;  // [M, C']
    int M, ldw2c, ldw2a;
};
// conv3_halo_bf16.hip: 3x3 / stride 1 / SAME, bf16, activation tile + halo resident in LDS (the 9 taps read one staged image)
struct Halo3Args {
    const __bf16* x;     // [M, Cin] NHWC, M = B * HW * HW
    const __bf16* w;     // packed [N][ldw], k = tap * Cin + c
    const float* scale;  // [N]
    const float* shift;  // [N]
    __bf16* y;           // [M, N]
    int M, N, ldw, relu;
    int n_ntiles;        // filled by the launcher
    int two;             // map sizes (bit mask: 1 = 7x7, 2 = 14x14, 4 = 28x28) on the two-workgroups-per-CU form (64 channels per workgroup, one image buffer)
};
bool hpe_halo3_bf16_supported(int HW, int Cin, int N);
hipError_t hpe_launch_halo3_bf16(const Halo3Args& p, int HW, int Cin, hipStream_t st);

bool hpe_chain_bf16_supported(int C, int C4, int CP, int C2);
hipError_t hpe_launch_chain_bf16(const ChainArgs& p, int C, int C4, int CP, int C2, hipStream_t st);
hipError_t hpe_chain_bf16_occupancy(int out[3]);
// conv_chain_f32.hip: the same chained launch in fp32 (identity blocks of stage 2, C = 64; opt-in).  u1_slab8: u1 is written channel-slab
// major, u1[(n / 8) * M + m][n % 8] (what the fused Winograd 3x3 kernel of the next block reads)
struct ChainArgsF32 {
    const float *t2, *res, *w2c, *w2a, *scaleA, *shiftA, *scaleB, *shiftB;
    float *t3, *u1;
    int M, ldw2c, ldw2a, u1_slab8;
};
bool hpe_chain_f32_supported(int C, int C4, int CP);
hipError_t hpe_launch_chain_f32(const ChainArgsF32& p, int C, int C4, int CP, hipStream_t st);
hipError_t hpe_chain_f32_occupancy(int* out);
hipError_t hpe_launch_f32_to_bf16(const float* x, void* y, long n, hipStream_t st);  // round to nearest even
hipError_t hpe_launch_bf16_to_f32(const void* x, float* y, long n, hipStream_t st);
hipError_t hpe_launch_pad_input_bf16(const float* img, void* out, int B, int H, int W, int Hp, int Wp, hipStream_t st);
hipError_t hpe_launch_maxpool_bf16(const void* x, void* y, int B, int H, int C, hipStream_t st);
hipError_t hpe_launch_avgpool_bf16(const void* x, float* y, int B, int HW, int C, int ldy, hipStream_t st);

// stem_fused.hip: conv1_pad + conv1 + bn_conv1 + ReLU + pool1_pad + max-pool in one kernel; img [B,224,224,3] fp32 ->
// y [B,56,56,64] (fp32, or bf16 when bf16 != 0).  w: fp32 [64][160] (k = kh * 22 + 1 + kw * 3 + c, other slots zero) or
// bf16 [64][7][32] (k = kh * 32 + kw * 4 + c, other slots zero).  R pooled rows per workgroup, 56 % R == 0, R <= 8.
hipError_t hpe_launch_stem_fused(const float* img, const void* w, const float* scale, const float* shift, void* y, int B, int R, int bf16,
                                 hipStream_t st);
hipError_t hpe_stem_fused_init_device();
int hpe_stem_fused_pick_rows(int B);

// encoder_ops.hip
hipError_t hpe_launch_pad_input(const float* img, float* out, int B, int H, int W, int Hp, int Wp, hipStream_t st);
hipError_t hpe_launch_maxpool(const float* x, float* y, int B, int H, int C, hipStream_t st);
hipError_t hpe_launch_avgpool(const float* x, float* y, int B, int HW, int C, int ldy, hipStream_t st);
// Dense layer at M <= 4 rows (one launch, no split-K): wt packed [n][K], K % 4 == 0
hipError_t hpe_launch_dense_gemv(const float* x, int lda, int M, int K, const float* wt, int N, const float* scale, const float* shift, const float* res,
                                 int ldres, int relu, float* y, int ldy, hipStream_t st);
hipError_t hpe_launch_tile_theta(const float* mean85, float* theta, int B, int ld, hipStream_t st);
hipError_t hpe_launch_copy_theta(const float* src, int lds, float* dst, int ldd, int B, int n, hipStream_t st);

// smpl.hip
struct SmplDev {
    // constants (device)
    const float* v_template;  // [V*3]
    const float* shapedirs;   // [10][V*3]
    const float* posedirs;    // [207][V*3]
    const float* weights;     // [V][24]
    const float* kp_reg;      // [V][KP_PITCH] (zero padded columns)
    const float* j_reg;       // [V][KP_PITCH]
    const float* j_basis;     // [11][24][3]  J of (v_template, shapedirs[0..9]) -- from the regress kernel
    const int* parents;       // [24]
    const int* depth;         // [24]
    int num_kp;
    int max_depth;
};
#define SMPL_V 6890
#define SMPL_KP_PITCH 24
#define SMPL_IMG_TILE 8
#define SMPL_SMALL_B 8  // batches up to this size take the latency path of smpl.hip

struct SmplWork {
    float* pfT;   // [207][Bpad]  pose feature, transposed
    float* betaT; // [10][Bpad]
    float* A;     // [Bpad][24][12]
    float* cams;  // [Bpad][4] (s, tx, ty, 0)
    float* verts_tmp; // [Bpad][V][3] used when the caller does not want verts but wants joints
    float* kp_part;   // [SMPL_SMALL_B][108][72] keypoint-regressor partials of the small-batch path (nullptr: always the large-batch kernels)
    int Bpad;
};

struct HpeOutputs;
hipError_t hpe_launch_smpl(const SmplDev& d, const SmplWork& w, const float* theta, int ldtheta, int B, const HpeOutputs* o,
                           hipStream_t st);
hipError_t hpe_launch_joint_regress(const float* X, const float* reg, int n, int K, float* out, const float* cams,
                                    float* kp2d, hipStream_t st);
hipError_t hpe_launch_orth_proj(const float* X, const float* cam, int B, int P, float sx, float sy, int pixels, float* out,
                                hipStream_t st);

// losses.hip
hipError_t hpe_launch_kp_loss(const float* gt, const float* pred, int n, float* out, hipStream_t st);
hipError_t hpe_losses_init_device();  // per-device kernel attributes of the loss kernels
size_t hpe_mesh_loss_ws_floats(int B, int H, int W, int P);
// a2b_mode: 0 cell-grid search (default), 1 VALU full search, 2 matrix-core full search; counter: optional 2 x u64 (MFMAs issued
// by the grid / full search)
int hpe_mesh_a2b_mode_from_env();
hipError_t hpe_launch_mesh_loss(const float* seg, const float* v2d, int B, int H, int W, int P, float* ws, float* out,
                                hipStream_t st, int a2b_mode, unsigned long long* counter);
// the same in two halves: silhouette compaction + bitmap (once per step), then the searches of one vertex set (once per IEF stage)
hipError_t hpe_launch_mesh_loss_prepare(const float* seg, int B, int H, int W, int P, float* ws, hipStream_t st);
// (ev_a2b0 / ev_a2b1: optional events recorded around the pixel -> vertex search, the dominant kernel of the loss)
hipError_t hpe_launch_mesh_loss_search(const float* v2d, int B, int H, int W, int P, float* ws, float* out, hipStream_t st,
                                       hipEvent_t ev_a2b0, hipEvent_t ev_a2b1, int a2b_mode, unsigned long long* counter);

// prepost.hip
hipError_t hpe_launch_preprocess_u8(const unsigned char* img, int H, int W, int C, int newH, int newW, int start_x, int start_y,
                                    int margin, float* out, int S, hipStream_t st);
// batch of frames in one launch: frame b at img + (table ? table[b].offset : b * H * W * C); table == nullptr: all frames share the
// geometry in `uni`
struct PreprocFrame {
    long long offset;
    int H, W, newH, newW, start_x, start_y;
};
hipError_t hpe_launch_preprocess_u8_batch(const unsigned char* img, const PreprocFrame* table_dev, PreprocFrame uni, int B, int C, int margin,
                                          float* out, int S, hipStream_t st);
hipError_t hpe_launch_shift_verts(const float* verts, const float* cam, int B, int P, float flength, float img_size, float* out,
                                  hipStream_t st);
