// hpe_api.hip -- the C ABI of include/hpe.h: context, weight ingestion (Keras layouts), plan and dispatch.
// Host logic only; the kernels live in conv_gemm.hip / encoder_ops.hip / smpl.hip / losses.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hpe.h"
#include "hpe_internal.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            return fail(HPE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +      \
                                         std::to_string(__LINE__) + ")");                                         \
    } while (0)

struct ConvSpec {
    char name[24];
    char bn[24];
    int kh, kw, cin, cout, stride, hin, hout;
};

// ResNet-50 v1 layer table, Keras names / order [2a, 2b, 2c, (1)] per block.
std::vector<ConvSpec> build_specs() {
    std::vector<ConvSpec> v;
    auto add = [&](const std::string& n, const std::string& b, int kh, int cin, int cout, int s, int hin, int hout) {
        ConvSpec c;
        snprintf(c.name, sizeof c.name, "%s", n.c_str());
        snprintf(c.bn, sizeof c.bn, "%s", b.c_str());
        c.kh = c.kw = kh;
        c.cin = cin;
        c.cout = cout;
        c.stride = s;
        c.hin = hin;
        c.hout = hout;
        v.push_back(c);
    };
    add("conv1", "bn_conv1", 7, 3, 64, 2, 224, 112);
    const int nblk[4] = {3, 4, 6, 3};
    const int filt[4][3] = {{64, 64, 256}, {128, 128, 512}, {256, 256, 1024}, {512, 512, 2048}};
    int cin = 64, h = 56;
    for (int st = 0; st < 4; ++st) {
        for (int b = 0; b < nblk[st]; ++b) {
            const bool first = b == 0;
            const int s = (first && st > 0) ? 2 : 1;
            const int hout = h / s;
            char base[24], bn[24];
            snprintf(base, sizeof base, "res%d%c_branch", st + 2, 'a' + b);
            snprintf(bn, sizeof bn, "bn%d%c_branch", st + 2, 'a' + b);
            add(std::string(base) + "2a", std::string(bn) + "2a", 1, cin, filt[st][0], s, h, hout);
            add(std::string(base) + "2b", std::string(bn) + "2b", 3, filt[st][0], filt[st][1], 1, hout, hout);
            add(std::string(base) + "2c", std::string(bn) + "2c", 1, filt[st][1], filt[st][2], 1, hout, hout);
            if (first) add(std::string(base) + "1", std::string(bn) + "1", 1, cin, filt[st][2], s, h, hout);
            cin = filt[st][2];
            h = hout;
        }
    }
    return v;
}

const std::vector<ConvSpec>& specs() {
    static const std::vector<ConvSpec> s = build_specs();
    return s;
}

inline int round_up(int x, int m) { return ((x + m - 1) / m) * m; }

struct ConvLayer {
    std::vector<float> kernel, bias, gamma, beta, mean, var;  // host staging (Keras layouts)
    bool loaded = false;
    float* w = nullptr;  // device, packed [n_pad][k_pad] (fp32) or bf16 [n_pad][k_pad16] in bf16 mode
    float* scale = nullptr;
    float* shift = nullptr;
    // *_branch2c of a conv_block only: [scale2c * W2c | scale1 * W1] concatenated along k and the summed shifts -- the expand
    // convolution and the projection shortcut as one dual-source GEMM (GEMM_DUAL)
    float* w_dual = nullptr;
    float* shift_dual = nullptr;
    int k_dual = 0, k1_dual = 0;
    void* stem_w = nullptr;   // conv1 only: weights in the k enumeration of stem_fused.hip (fp32 [64][160] / bf16 [64][7][32])
    float* wino_u = nullptr;  // device, G g G^T in the blocked layout of conv_wino.hip (3x3 layers on the Winograd path only)
    float* wino4_u = nullptr;  // device, the F(4x4,3x3) G g G^T in the blocked layout of conv_wino4.hip (layers selected by wino_f4)
    int n_pad = 0, k_pad = 0;
};

inline unsigned short f2bf(float f) {  // round-to-nearest-even fp32 -> bf16 (finite inputs)
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

struct DenseLayer {
    std::vector<float> kernel, bias;
    bool loaded = false;
};

constexpr int STEM_HP = 230;  // 224 + 2*3
constexpr int STEM_WP = 232;  // 224 + 2*3 + 2 (8th tap column of the last window, zero weights)
constexpr int THETA_LD = 96;  // theta rows padded to 3 k-slabs of 32
// Winograd V workspace: image-major with this per-image pitch; a chunk of >= 32 images starting at image i0 owns
// [i0 * pitch, (i0 + n) * pitch): n * 802816 floats of transformed tiles (16 * tiles * C <= 802816 per image for every
// 3x3 layer) + up to 63 padding tiles * 16 * 512 = 516096 floats <= n * 16384
constexpr size_t WINO_V_PITCH = 802816 + 16384;
constexpr size_t WINO_V_SLACK = 524288;

}  // namespace

// tile-selection overrides (HPE_TILE_WIDE / HPE_TILE_NARROW / HPE_SHORTK_TILE / HPE_TILE_BF16), read once in hpe_finalize
struct TileKnobs {
    int force_wide = -1, force_narrow = -1, shortk = TILE_128x64_W8, force_bf16 = -1;
    int bf16_128_min_tiles = 192;  // HPE_BF16_128_MIN_TILES: concurrent chunk launches take 128x128 from 0.75 tiles per CU on
    int concurrent_tiles = 0;      // HPE_CONCURRENT_TILES=1: the tile rule of concurrent chunk launches on every launch (profiling passes with HPE_STREAMS=1)
    int wide128_min_tiles = 384;  // HPE_WIDE128_MIN_TILES: 1.5 tiles per CU (0 = the round-1 rule everywhere)
    int force_expand = -1;   // HPE_EXPAND_TILE: fp32 tile of the identity-block expand layers (experiment knob)
    int expand_small_grid = 128;  // HPE_EXPAND_SMALL_GRID: expand layers with fewer 128x64 tiles than this take the 64x64 split-K tile (0 = never)
    int bf16_w8_min_tiles = 128;  // HPE_BF16_W8_MIN_TILES: N > 64 bf16 launches with >= this many 128 x 128 tiles use the 8-wave tile (0 = never)
    int force_ns_bf16 = -1;  // HPE_NS_BF16: LDS ring depth of the bf16 GEMM (2..4), -1 = per-layer rule
    int bf16_rules = 1;      // HPE_BF16_RULES=0: the round-1 tile rule (128x128 / 64x128 by grid size, double buffer)
    int bf16_p8 = 0;         // HPE_BF16_P8: layer kinds that take the 256 x 256 phase-interleaved kernel (bit mask, see pick_bf16)
    int bf16_p8_min_n = 256; // HPE_BF16_P8_MINN
    int bf16_p8_min_k = 512; // HPE_BF16_P8_MINK
};

struct hpe_ctx {
    HpeConfig cfg{};
    bool finalized = false;
    bool dead = false;  // hpe_finalize failed part-way: everything it had allocated was released, the ctx can only be destroyed
    TileKnobs knobs;
    bool bf16 = false;  // encoder_dtype == 1
    bool have_encoder = false, have_regressor = false, have_smpl = false;
    ConvLayer conv[HPE_NUM_CONV];
    DenseLayer dense[HPE_NUM_DENSE];
    // SMPL host staging
    bool smpl_loaded = false, mean_loaded = false;
    std::vector<float> h_vt, h_sd, h_pd, h_jreg, h_w, h_kreg;
    std::vector<int> h_par;
    int num_kp = 19;
    float h_mean[HPE_THETA_DIM];
    // device: regressor
    float *w1f = nullptr, *w1t = nullptr, *w2 = nullptr, *w3 = nullptr, *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
    float *ones = nullptr, *zeros = nullptr, *mean_dev = nullptr;
    // device: SMPL
    SmplDev smpl{};
    SmplWork work{};
    float* smpl_basis_src = nullptr;  // [11][V*3]: v_template | shapedirs^T
    // device: activations
    float *padded = nullptr, *X0 = nullptr, *X1 = nullptr, *T1 = nullptr, *T2 = nullptr, *SC = nullptr;
    float *feat = nullptr, *P1 = nullptr, *H1 = nullptr, *H2 = nullptr, *thA = nullptr, *thB = nullptr;
    float* loss_ws = nullptr;
    unsigned long long* dbg = nullptr;  // diagnostics buffer (hpe_debug_set_dbg)
    size_t loss_ws_floats = 0;
    std::vector<void*> allocs;
    // batch-chunk streams
    int n_streams = 1;
    float* partial = nullptr;  // split-K workspace (small grids only run unchunked on the caller's stream)
    size_t partial_floats = 0;
    int chunk_images = 0;
    float* wino_v = nullptr;  // Winograd input-transform workspace (nullptr: direct convolution everywhere)
    float* wino_ws = nullptr;       // stream-K parking space, one slot of n_cu workgroups per chunk stream (nullptr: plain grid)
    unsigned* wino_flags = nullptr;
    unsigned* dev_err = nullptr;  // device error word (bit 0: a stream-K wait timed out -> wrong output), see hpe_device_status
    unsigned wino_epoch = 0;
    int n_cu = 0;
    int wino_min_c = 128;     // 3x3 layers with at least this many channels take the Winograd path
    int wino_min_items = 128; // ... when the launch has at least this many workgroups
    int wino_fused_min_hw = 28;  // smallest map side on the fused path (HPE_WINO_FUSED_MINHW)
    int dual_gemm = 1;        // conv_block: branch2c + branch1 in one launch (HPE_DUAL=0: two launches through the shortcut buffer)
    int stem_fused = 1;       // conv1 + BN + ReLU + max-pool as one kernel reading the raw images (HPE_STEM_FUSED=0: pad / im2col GEMM / pool)
    int wino4_min_items = 64;  // F(4x4) launches need at least this many 32-cout workgroups (HPE_WINO4_MIN_ITEMS), else F(2x2) / direct by their rules
    int wino4_fused = 0;      // map sizes (bits as wino_f4: 4 = 28x28, 8 = 56x56) whose F(4x4) layers take the fused-transform kernel (HPE_WINO4_FUSED)
    int wino_f4 = 0;          // map sizes whose 3x3 layers run as Winograd F(4x4,3x3): bit 0: 7x7, 1: 14x14, 2: 28x28, 3: 56x56 (HPE_WINO_F4)
    int mesh_a2b = 0;         // pixel -> vertex search of the mesh loss: 0 cell grid, 1 VALU full search, 2 matrix-core full search
    bool loss_attr_done = false;  // per-device kernel attributes of the loss kernels set (hpe_finalize, or the first loss call of a loss-only ctx)
    unsigned long long* loss_counter = nullptr;  // hpe_debug_set_loss_counter
    int wino_fused = 1;       // 56x56 / 28x28 maps: input transform inside the GEMM kernel, fed by a slab-major 1x1 producer
    hipStream_t aux[3]{};
    int min_chunk = 32;  // HPE_MIN_CHUNK: smallest batch chunk that still gets its own stream
    hipEvent_t ev_fork{}, ev_join[3]{};
    // software pipeline across calls (hpe_forward_pipelined): the regressor + SMPL tail of batch k runs on `tail_st` while the
    // caller's stream already runs the encoder of batch k+1; features alternate between two buffers, the Dense layers of the tail
    // have their own split-K workspace
    hipStream_t tail_st{};
    hipEvent_t ev_enc{}, ev_tail{}, ev_feat_free[2]{};
    bool feat_free_valid[2] = {false, false};
    bool tail_pending = false;
    unsigned pipe_idx = 0;
    float* feat_alt = nullptr;
    int wino4_ksplit = 1;  // plan option wino4_ksplit / HPE_WINO4_KSPLIT
    int halo3_two = 4;     // HPE_HALO3_TWO: map sizes of halo3 on the two-workgroups-per-CU form of that kernel (default: 28x28)
    int halo3 = 0;         // bf16 only: map sizes (1 = 7x7, 2 = 14x14, 4 = 28x28, 8 = 56x56) whose 3x3 layers run on conv3_halo_bf16.hip; plan option halo3 / HPE_HALO3
    int chain_fuse = 0;    // bf16 only: stages (bit 0: stage 2, bit 1: stage 3) whose identity blocks run branch2c + the next block's branch2a as
                           // one launch (conv_chain_bf16.hip); plan option chain_fuse / HPE_CHAIN
    int co_running = 1;    // chunk streams of the encoder call being enqueued (launch-size rules of the F(4x4) kernels)
    float* w4_split = nullptr;  // F(4x4) C-axis split workspaces + counters (4 x hpe_wino4_split_ws_floats: one per chunk-stream slot)
    float* partial_tail = nullptr;
    size_t partial_tail_floats = 0;
    bool dense_on_tail = false;  // set while a pipelined tail is being enqueued: run_dense then uses partial_tail
    // timing
    int timing = 0;
    hipEvent_t ev[8]{};
    // encoder span of every timed call since hpe_enable_timing (ring of the last SPAN_RING calls): hpe_get_span_stats
    static constexpr int SPAN_RING = 64;
    hipEvent_t span0[SPAN_RING]{}, span1[SPAN_RING]{};
    unsigned span_n = 0;
    hipEvent_t cev0[HPE_NUM_CONV]{}, cev1[HPE_NUM_CONV]{};
    hipEvent_t lev0[16]{}, lev1[16]{}, lev_all[2]{};  // hpe_val_losses: around each stage's pixel -> vertex search / the whole call
    int loss_timed_stages = 0;
    bool ev_ok = false, timed_valid = false, conv_timed_valid = false;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

int dev_alloc(hpe_ctx* c, float** p, size_t n_floats, bool zero) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, n_floats * sizeof(float)));
    c->allocs.push_back(q);
    if (zero) HIP_TRY(hipMemset(q, 0, n_floats * sizeof(float)));
    *p = static_cast<float*>(q);
    return HPE_OK;
}

int upload(hpe_ctx* c, float** p, const std::vector<float>& h) {
    int rc = dev_alloc(c, p, h.size(), false);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return HPE_OK;
}

int pick_tile(const TileKnobs& kn, int M, int N, int K, bool residual_expand = false, bool concurrent = false) {
    // prefer the largest tile that still gives >= 2 workgroups per CU; N == 64 layers use 64-wide tiles
    const bool wide = N > 64;
    // identity-block expand layers: 8 waves (see below) -- unless the grid is so small that the launch is DMA latency: then the 4-wave
    // 64x64 tile, which the launcher cuts along K (single frames: res5*_branch2c 21 -> 9 us)
    if (wide && residual_expand) {
        if (kn.force_expand >= 0) return kn.force_expand;
        return (long)((M + 127) / 128) * ((N + 63) / 64) < kn.expand_small_grid ? TILE_64x64 : TILE_128x64_W8;
    }
    if (wide && kn.force_wide >= 0) return kn.force_wide;
    if (!wide && kn.force_narrow >= 0) return kn.force_narrow;
    // Measured on MI355X (profiles/r01/d_tile_sweep.txt): with LDS-DMA staging the small tiles with 3-5 workgroups
    // per CU beat 128x128 at 2 per CU except on the huge-M layers of stages 2-3.
    if (!wide) return TILE_128x64;
    // Identity-block expand layers (above) and K <= 128 on the huge-M maps (the C -> 4C expand / projection layers of stages 2 and
    // 3): the launch is mostly epilogue -> 8 waves to issue the row stores and residual loads win; 128x64 beats 128x128
    // (profiles/r01/h_tile_128x64w8.txt: res2*_branch2c 0.41-0.43 -> 0.37-0.38 ms, res3*_branch2c 0.31 -> 0.28 ms; stages 4-5:
    // equal to the 64x64 tile within 1 %, profiles/r02/fp32_expand_tile.txt)
    if (K <= 128 && M >= 150000) return kn.shortk;
    // Launches of concurrent batch chunks: 128x128 wherever it still leaves >= 1.5 tiles per CU (round 2, pipelined steps + two chunk
    // streams at B = 256: 17,440 -> 17,830 img/s, B = 128: +0.7 %, although most of these layers are 5-10 % SLOWER with it when they
    // run alone -- fewer, longer workgroups leave the co-running chunk's kernels more room).  A single-chunk batch keeps the
    // round-1 rule (B = 64: -0.6 ... -1.2 % with 128x128).  Thresholds 300 / 390 / 700 tiles: 17,805 / 17,843 / 17,806 img/s.
    if ((concurrent || kn.concurrent_tiles) && kn.wide128_min_tiles > 0 && (long)((M + 127) / 128) * ((N + 127) / 128) >= kn.wide128_min_tiles) return TILE_128x128;
    if (M >= 150000) return TILE_64x128;
    return TILE_64x64;
}

struct Bf16Plan {
    int tile, ns;
};

// bf16 tile + ring depth per layer kind, from the per-layer sweeps in profiles/r02 (B = 256):
//  * identity-block expand layers (1x1, K = C, N = 4C, + residual): all epilogue -> 128x64 with 8 waves issuing the row stores
//    and residual loads (res2b_branch2c 0.273 -> 0.182 ms = 5.1 TB/s, res3* 0.157 -> 0.108, res4* 0.079 -> 0.062, res5* 0.061 -> 0.048)
//  * everything with a long k axis on the small maps (stage 5: M = 49 B): 256x128, 8 waves (res5*_branch2b 0.112 -> 0.083 ms)
//  * otherwise 128x128 while that still gives >= 512 workgroups, else 64x128
Bf16Plan pick_bf16(const TileKnobs& kn, int M, int N, int K, bool residual_expand, bool concurrent = false, int mode = GEMM_DENSE) {
    Bf16Plan pl{TILE_128x64, 2};
    // 256 x 256 phase-interleaved kernel (conv_gemm_bf16_p8.hip), per layer kind -- bits of bf16_p8:
    //   1: 3x3 layers with N == 256 (stage 4), 2: 3x3 layers with N >= 512 (stage 5), 4: 1x1 / strided layers,
    //   8: dual-source launches with N >= 2048 (res5a), 16: the other dual-source launches
    if (kn.bf16_p8 && N >= kn.bf16_p8_min_n && N % 256 == 0 && K >= kn.bf16_p8_min_k && !residual_expand) {
        const int bit = mode == GEMM_CONV3 ? (N == 256 ? 1 : 2) : (mode == GEMM_DUAL ? (N >= 2048 ? 8 : 16) : 4);
        if (kn.bf16_p8 & bit) return Bf16Plan{TILE_P8_256x256, 2};
    }
    if (N > 64) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        pl.tile = t128 >= ((concurrent || kn.concurrent_tiles) ? kn.bf16_128_min_tiles : 512) ? TILE_128x128 : TILE_64x128;
        if (kn.bf16_rules) {
            if (residual_expand) pl.tile = TILE_128x64_W8;
            else if (M <= 16384 && M >= 8192 && K >= 1024 && N >= 256) pl.tile = TILE_256x128_W8;
            // Round 4: these launches are paced by the ISSUE of their LDS-DMA instructions (60-180 cycles each for the issuing wave),
            // not by the matrix pipe (without any multiplies the 1x1 layers take 0.96-0.99 of their time; a deeper ring is slower):
            // the 128 x 128 tile with EIGHT waves halves the DMA instructions per wave and slab.  Every N > 64 layer of the B = 256
            // step is equal or faster with it (serial pass 3.59 -> 3.50 ms, step 75.1 -> 76.7 k img/s); small grids keep the old rules.
            if (kn.bf16_w8_min_tiles > 0 && t128 >= kn.bf16_w8_min_tiles) pl.tile = TILE_128x128_W8;
        }
        if (kn.force_bf16 >= 0) pl.tile = kn.force_bf16;
    }
    if (kn.force_ns_bf16 >= 2) pl.ns = kn.force_ns_bf16;  // experiment knob: a 3-deep ring lost everywhere (conv_gemm_bf16.hip)
    return pl;
}

#define HIPE(expr)                               \
    do {                                         \
        hipError_t _e = (expr);                  \
        if (_e != hipSuccess) return _e;         \
    } while (0)

// one conv layer (+BN fold, +residual, +ReLU) through the implicit-GEMM kernel

inline int f4_bit(int hin) { return hin <= 7 ? 1 : (hin <= 14 ? 2 : (hin <= 28 ? 4 : 8)); }

// the 3x3 layer idx runs as Winograd F(4x4,3x3) for this batch (blocked V through the workspace)
bool use_wino4(const hpe_ctx* c, int idx, int B) {
    const ConvSpec& s = specs()[idx];
    return !c->bf16 && c->conv[idx].wino4_u && s.kh == 3 && s.stride == 1 && (c->wino_f4 & f4_bit(s.hin)) &&
           hpe_wino4_items(B, s.hin, s.hin, s.cout) >= (c->wino_min_items < c->wino4_min_items ? c->wino_min_items : c->wino4_min_items);
}

// ... with the input transform inside the GEMM kernel (its 1x1 producer then writes channel-slab major); takes precedence over use_wino4
bool use_wino4_fused(const hpe_ctx* c, int idx, int B) {
    const ConvSpec& s = specs()[idx];
    return !c->bf16 && c->conv[idx].wino4_u && s.kh == 3 && s.stride == 1 && (c->wino4_fused & f4_bit(s.hin)) &&
           hpe_wino4_fused_items(B, s.hin, s.hin, s.cout) >= (c->wino_min_items < c->wino4_min_items ? c->wino_min_items : c->wino4_min_items);
}

// the 3x3 layer idx runs as the fused F(2x2) Winograd kernel for this batch (its 1x1 producer then writes channel-slab major)
bool use_wino_fused(const hpe_ctx* c, int idx, int B) {
    const ConvSpec& s = specs()[idx];
    if (use_wino4_fused(c, idx, B) || use_wino4(c, idx, B)) return false;
    return c->wino_fused && !c->bf16 && c->conv[idx].wino_u && s.kh == 3 && s.stride == 1 && s.hin >= c->wino_fused_min_hw &&
           hpe_wino_fused_items(B, s.hin, s.hin, s.cout) >= c->wino_min_items;
}

enum { CONV_OUT_SLAB8 = 1, CONV_IN_SLAB8 = 2, CONV_CONCURRENT = 4 };

hipError_t run_conv(hpe_ctx* c, int idx, const float* x, int B, const float* res, int relu, float* y, hipStream_t st,
                    float* wino_v = nullptr, int slot = 0, int flags = 0) {
    const ConvSpec& s = specs()[idx];
    const ConvLayer& L = c->conv[idx];
    if ((flags & CONV_IN_SLAB8) && use_wino4_fused(c, idx, B))
        return hpe_launch_wino4_fused_conv3(x, L.wino4_u, L.scale, L.shift, c->zeros, y, s.cout, B, s.hin, s.hin, s.cin, s.cout, relu, st);
    if (flags & CONV_IN_SLAB8)
        return hpe_launch_wino_fused_conv3(x, L.wino_u, L.scale, L.shift, c->zeros, y, s.cout, B, s.hin, s.hin, s.cin, s.cout, relu, st);
    if (wino_v && !res && use_wino4(c, idx, B))
        return hpe_launch_wino4_conv3(x, s.cin, L.wino4_u, L.scale, L.shift, y, s.cout, B, s.hin, s.hin, s.cin, s.cout, relu, wino_v, st,
                                      (flags & CONV_CONCURRENT) ? c->co_running : 1,
                                      c->w4_split && slot >= 0 && slot < 4 ? c->w4_split + (size_t)slot * hpe_wino4_split_ws_floats() : nullptr);
    // Winograd needs enough (64-tile x 64-cout) work items to occupy the 256 CUs (one 8-wave workgroup each); below that
    // the direct kernel with split-K is faster (measured crossover: batch ~32, profiles/r01/g_wino_small_batch.txt)
    if (L.wino_u && wino_v && !res && s.cin >= c->wino_min_c &&
        (long)((B * ((s.hin + 1) / 2) * ((s.hin + 1) / 2) + 63) / 64) * (s.cout / 64) >= c->wino_min_items)
    {
        WinoStreamK sk{};
        if (c->wino_ws && slot >= 0 && slot < 4) {
            sk.ws = c->wino_ws + (size_t)slot * c->n_cu * HPE_WINO_WS_FLOATS;
            sk.flags = c->wino_flags + (size_t)slot * c->n_cu;
            sk.epoch = ++c->wino_epoch;
            if (sk.epoch == 0) sk.epoch = ++c->wino_epoch;
            sk.n_wg = c->n_cu;
            sk.err = c->dev_err;
        }
        return hpe_launch_wino_conv3(x, s.cin, L.wino_u, L.scale, L.shift, y, s.cout, B, s.hin, s.hin, s.cin, s.cout, relu, wino_v,
                                     c->wino_ws ? &sk : nullptr, st);
    }
    GemmArgs p{};
    p.x = x;
    p.w = L.w;
    p.scale = L.scale;
    p.shift = L.shift;
    p.res = res;
    p.y = y;
    p.M = B * s.hout * s.hout;
    p.N = s.cout;
    p.K = L.k_pad;
    p.ldw = L.k_pad;
    p.w_rows = L.n_pad;
    p.ldy = s.cout;
    p.ldres = s.cout;
    p.relu = relu;
    p.Hi = p.Wi = s.hin;
    p.Cin = s.cin;
    p.Ho = p.Wo = s.hout;
    p.stride = s.stride;
    p.cin_slabs = s.cin / 32;
    p.lda = s.cin;
    p.zero = c->zeros;
    p.y_slab8 = (flags & CONV_OUT_SLAB8) ? 1 : 0;
    // The ctx has ONE split-K workspace: only a launch that is alone on the device may use it.  Batch chunks running on
    // concurrent streams (CONV_CONCURRENT) never split K, whatever their size (their grids overlap each other instead).
    if (!(flags & CONV_CONCURRENT)) {
        p.partial = c->partial;
        p.partial_floats = c->partial_floats;
    }
    int mode;
    if (idx == 0) {
        mode = GEMM_STEM;
        p.Hi = STEM_HP;
        p.Wi = STEM_WP;
        p.Cin = 4;
    } else if (s.kh == 3) {
        mode = GEMM_CONV3;
    } else if (s.stride == 1) {
        mode = GEMM_DENSE;
    } else {
        mode = GEMM_STRIDED;
    }
    if (c->bf16 && mode == GEMM_CONV3 && !res && s.stride == 1 && (c->halo3 & f4_bit(s.hin) ? true : false) &&
        hpe_halo3_bf16_supported(s.hin, s.cin, s.cout) && L.k_pad >= 9 * s.cin)
    {
        Halo3Args h{};
        h.x = reinterpret_cast<const __bf16*>(x);
        h.w = reinterpret_cast<const __bf16*>(L.w);
        h.scale = L.scale;
        h.shift = L.shift;
        h.y = reinterpret_cast<__bf16*>(y);
        h.M = p.M;
        h.N = s.cout;
        h.ldw = L.k_pad;
        h.relu = relu;
        h.two = c->halo3_two;
        return hpe_launch_halo3_bf16(h, s.hin, s.cin, st);
    }
    if (c->bf16) {
        p.cin_slabs = s.cin / 64;
        const Bf16Plan pl = pick_bf16(c->knobs, p.M, p.N, p.K, mode == GEMM_DENSE && res != nullptr && s.cout == 4 * s.cin, (flags & CONV_CONCURRENT) != 0, mode);
        return hpe_launch_gemm_bf16(p, mode, pl.tile, pl.ns, st);
    }
    return hpe_launch_gemm(p, mode, pick_tile(c->knobs, p.M, p.N, p.K, mode == GEMM_DENSE && res != nullptr && s.cout == 4 * s.cin, (flags & CONV_CONCURRENT) != 0), st);
}

// branch2c (+BN) + branch1 (+BN) + add + ReLU of a conv_block as one dual-source GEMM: t2 [M, K1] dense, x NHWC strided
hipError_t run_dual(hpe_ctx* c, int i2c, int i1, const float* t2, const float* x, int B, float* y, hipStream_t st, int flags) {
    const ConvSpec& s2 = specs()[i2c];
    const ConvSpec& s1 = specs()[i1];
    const ConvLayer& L = c->conv[i2c];
    const int slab = c->bf16 ? 64 : 32;
    GemmArgs p{};
    p.x = t2;
    p.x2 = x;
    p.w = L.w_dual;
    p.scale = c->ones;
    p.shift = L.shift_dual;
    p.y = y;
    p.M = B * s2.hout * s2.hout;
    p.N = s2.cout;
    p.K = L.k_dual;
    p.k1_slabs = L.k1_dual / slab;
    p.lda = s2.cin;
    p.ldw = L.k_dual;
    p.w_rows = round_up(s2.cout, 128);
    p.ldy = s2.cout;
    p.relu = 1;
    p.Hi = p.Wi = s1.hin;
    p.Cin = s1.cin;
    p.Ho = p.Wo = s1.hout;
    p.stride = s1.stride;
    p.zero = c->zeros;
    if (!(flags & CONV_CONCURRENT)) {
        p.partial = c->partial;
        p.partial_floats = c->partial_floats;
    }
    if (c->bf16) {
        const Bf16Plan pl = pick_bf16(c->knobs, p.M, p.N, p.K, false, (flags & CONV_CONCURRENT) != 0, GEMM_DUAL);
        return hpe_launch_gemm_bf16(p, GEMM_DUAL, pl.tile, pl.ns, st);
    }
    return hpe_launch_gemm(p, GEMM_DUAL, pick_tile(c->knobs, p.M, p.N, p.K, false, (flags & CONV_CONCURRENT) != 0), st);
}

// the bf16 identity-block pair branch2c (idx i2c, + residual + ReLU) -> next block's branch2a (idx i2c + 1) as one launch
// `first`: the conv_block form -- branch2c + the projection shortcut branch1 (idx i2c + 1, stride 1: stage 2 only) as the dual-source GEMM,
// chained with the next block's branch2a (idx i2c + 2); bit 2 of chain_fuse
bool use_chain(const hpe_ctx* c, int stg, int i2c, bool first, bool has_next) {
    if (!has_next) return false;
    const ConvSpec& s2 = specs()[i2c];
    if (!c->bf16) {
        // fp32: identity blocks of stage 2 only (conv_chain_f32.hip; bit 3 of chain_fuse, on by default: A/B on two boxes +0.3 ... +1.4 % at
        // B = 256, +1.6 % at B = 64)
        if (first || stg != 0 || !(c->chain_fuse & 8)) return false;
        const ConvSpec& sn = specs()[i2c + 1];
        return sn.kh == 1 && sn.stride == 1 && sn.cin == s2.cout && hpe_chain_f32_supported(s2.cin, s2.cout, sn.cout);
    }
    if (first) {
        const ConvSpec& s1 = specs()[i2c + 1];
        const ConvSpec& sn = specs()[i2c + 2];
        return stg == 0 && (c->chain_fuse & 4) && c->conv[i2c].w_dual && s1.stride == 1 && s1.hin == s2.hin && sn.kh == 1 && sn.stride == 1 &&
               sn.cin == s2.cout && c->conv[i2c].k_dual == s2.cin + s1.cin && hpe_chain_bf16_supported(s2.cin, s2.cout, sn.cout, s1.cin);
    }
    // identity blocks: bit 0 = stage 2, bit 1 = stage 3, bit 4 (value 16) = stage 4 (128-pixel workgroups, one per CU)
    const int bit = stg == 0 ? 1 : stg == 1 ? 2 : stg == 2 ? 16 : 0;
    if (!(c->chain_fuse & bit)) return false;
    const ConvSpec& sn = specs()[i2c + 1];
    return sn.kh == 1 && sn.stride == 1 && sn.cin == s2.cout && hpe_chain_bf16_supported(s2.cin, s2.cout, sn.cout, 0);
}

// res: the block input -- the residual of an identity block, the second A source of a conv_block
hipError_t run_chain(hpe_ctx* c, int i2c, bool first, const float* t2, const float* res, int B, float* t3, float* u1, hipStream_t st,
                     bool u1_slab8 = false) {
    const ConvSpec& s2 = specs()[i2c];
    const int inext = i2c + (first ? 2 : 1);
    const ConvSpec& sn = specs()[inext];
    const ConvLayer& L2 = c->conv[i2c];
    const ConvLayer& Ln = c->conv[inext];
    if (!c->bf16) {
        ChainArgsF32 q{};
        q.t2 = t2;
        q.res = res;
        q.w2c = L2.w;
        q.w2a = Ln.w;
        q.scaleA = L2.scale;
        q.shiftA = L2.shift;
        q.scaleB = Ln.scale;
        q.shiftB = Ln.shift;
        q.t3 = t3;
        q.u1 = u1;
        q.M = B * s2.hout * s2.hout;
        q.ldw2c = L2.k_pad;
        q.ldw2a = Ln.k_pad;
        q.u1_slab8 = u1_slab8 ? 1 : 0;
        return hpe_launch_chain_f32(q, s2.cin, s2.cout, sn.cout, st);
    }
    ChainArgs p{};
    p.t2 = reinterpret_cast<const __bf16*>(t2);
    if (first) {
        p.x2 = reinterpret_cast<const __bf16*>(res);
        p.w2c = reinterpret_cast<const __bf16*>(L2.w_dual);
        p.scaleA = c->ones;
        p.shiftA = L2.shift_dual;
        p.ldw2c = L2.k_dual;
    } else {
        p.res = reinterpret_cast<const __bf16*>(res);
        p.w2c = reinterpret_cast<const __bf16*>(L2.w);
        p.scaleA = L2.scale;
        p.shiftA = L2.shift;
        p.ldw2c = L2.k_pad;
    }
    p.w2a = reinterpret_cast<const __bf16*>(Ln.w);
    p.scaleB = Ln.scale;
    p.shiftB = Ln.shift;
    p.t3 = reinterpret_cast<__bf16*>(t3);
    p.u1 = reinterpret_cast<__bf16*>(u1);
    p.M = B * s2.hout * s2.hout;
    p.ldw2a = Ln.k_pad;
    return hpe_launch_chain_bf16(p, s2.cin, s2.cout, sn.cout, first ? specs()[i2c + 1].cin : 0, st);
}

hipError_t run_dense(hpe_ctx* c, const float* x, int lda, int M, int K, const float* w, int w_rows, int N, const float* scale,
                     const float* shift, const float* res, int ldres, int relu, float* y, int ldy, hipStream_t st) {
    // single frames and very small batches: one launch per layer (the implicit-GEMM kernel would need split-K + a fix-up launch)
    if (M <= 4) return hpe_launch_dense_gemv(x, lda, M, K, w, N, scale, shift, res, ldres, relu, y, ldy, st);
    GemmArgs p{};
    p.zero = shift;  // any readable 16 B: dense mode never takes the zero-page path
    // the Dense layers run after the chunk streams have joined; in the pipelined forward they overlap the NEXT batch's encoder,
    // whose unchunked launches may split K too -> separate workspace
    p.partial = c->dense_on_tail ? c->partial_tail : c->partial;
    p.partial_floats = c->dense_on_tail ? c->partial_tail_floats : c->partial_floats;
    p.x = x;
    p.w = w;
    p.scale = scale;
    p.shift = shift;
    p.res = res;
    p.y = y;
    p.M = M;
    p.N = N;
    p.K = K;
    p.lda = lda;
    p.ldw = K;
    p.w_rows = w_rows;
    p.ldy = ldy;
    p.ldres = ldres;
    p.relu = relu;
    return hpe_launch_gemm(p, GEMM_DENSE, TILE_64x64, st);
}

hipError_t timed_conv(hpe_ctx* c, int idx, const float* x, int B, const float* res, int relu, float* y, hipStream_t st,
                      float* wino_v = nullptr, int slot = 0, int flags = 0) {
    const bool t2 = c->timing >= 2;
    if (t2) HIPE(hipEventRecord(c->cev0[idx], st));
    HIPE(run_conv(c, idx, x, B, res, relu, y, st, wino_v, slot, flags));
    if (t2) HIPE(hipEventRecord(c->cev1[idx], st));
    return hipSuccess;
}

// the encoder on images [i0, i0+B) of the batch (all workspace buffers are image-major)
hipError_t encoder_chunk(hpe_ctx* c, const float* images, int i0, int B, float* features, int ldfeat, hipStream_t st, int slot = 0,
                         bool concurrent = false) {
    const int cf = concurrent ? CONV_CONCURRENT : 0;
    // all workspace buffers are image-major; in bf16 mode the same allocations hold bf16 elements (half the bytes)
    const int esz = c->bf16 ? 2 : 4;
    auto at = [&](float* base, size_t elems) { return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + elems * esz); };
    const size_t o_img = (size_t)i0 * HPE_IMG_SIZE * HPE_IMG_SIZE * 3;
    const size_t o_pad = (size_t)i0 * STEM_HP * STEM_WP * 4;
    const size_t o_big = (size_t)i0 * 802816;
    const size_t o_mid = (size_t)i0 * 200704;
    float* padded = at(c->padded, o_pad);
    float* SC = at(c->SC, o_big);
    float* T1 = at(c->T1, o_mid);
    float* T2 = at(c->T2, o_mid);
    float* cur = at(c->X0, o_big);
    float* nxt = at(c->X1, o_big);
    // the chunk's slice of the Winograd workspace (chunks of < 32 images only occur unchunked, i0 == 0: the slack at the end covers them)
    float* wv = (c->wino_v && (i0 == 0 || B >= 32)) ? c->wino_v + (size_t)i0 * WINO_V_PITCH : nullptr;
    // the fused stem stages whole 16-byte chunks of the caller's rows; an images pointer that is only float-aligned (e.g. a
    // tensor view at an odd offset) takes the pad / im2col / pool path, which reads the images with scalar loads
    if (c->stem_fused && (reinterpret_cast<uintptr_t>(images + o_img) & 15) == 0) {
        // conv1_pad .. pool1 in one kernel straight from the caller's images (stem_fused.hip); timed as conv layer 0
        const bool t2 = c->timing >= 2;
        if (t2) HIPE(hipEventRecord(c->cev0[0], st));
        HIPE(hpe_launch_stem_fused(images + o_img, c->conv[0].stem_w, c->conv[0].scale, c->conv[0].shift, cur, B, hpe_stem_fused_pick_rows(B),
                                   c->bf16 ? 1 : 0, st));
        if (t2) HIPE(hipEventRecord(c->cev1[0], st));
    } else if (c->bf16) {
        HIPE(hpe_launch_pad_input_bf16(images + o_img, padded, B, HPE_IMG_SIZE, HPE_IMG_SIZE, STEM_HP, STEM_WP, st));
        HIPE(timed_conv(c, 0, padded, B, nullptr, 1, SC, st, nullptr, 0, cf));
        HIPE(hpe_launch_maxpool_bf16(SC, cur, B, 112, 64, st));
    } else {
        HIPE(hpe_launch_pad_input(images + o_img, padded, B, HPE_IMG_SIZE, HPE_IMG_SIZE, STEM_HP, STEM_WP, st));
        HIPE(timed_conv(c, 0, padded, B, nullptr, 1, SC, st, nullptr, 0, cf));
        HIPE(hpe_launch_maxpool(SC, cur, B, 112, 64, st));
    }
    int ci = 1;
    const int nblk[4] = {3, 4, 6, 3};
    bool have_2a = false;  // the previous block's chained launch has already written this block's branch2a output to T1
    for (int stg = 0; stg < 4; ++stg) {
        for (int b = 0; b < nblk[stg]; ++b) {
            const bool first = b == 0;
            const int i2a = ci, i2b = ci + 1, i2c = ci + 2, i1 = ci + 3;
            const bool fz = use_wino_fused(c, i2b, B) || use_wino4_fused(c, i2b, B);  // then T1 is channel-slab major and never leaves this pair of launches
            if (have_2a) {
                if (c->timing >= 2) {
                    HIPE(hipEventRecord(c->cev0[i2a], st));
                    HIPE(hipEventRecord(c->cev1[i2a], st));
                }
            } else {
                HIPE(timed_conv(c, i2a, cur, B, nullptr, 1, T1, st, nullptr, 0, cf | (fz ? CONV_OUT_SLAB8 : 0)));
            }
            have_2a = false;
            HIPE(timed_conv(c, i2b, T1, B, nullptr, 1, T2, st, wv, slot, cf | (fz ? CONV_IN_SLAB8 : 0)));
            const float* res = cur;
            if (use_chain(c, stg, i2c, first, b + 1 < nblk[stg])) {
                // identity block followed by an identity block (bf16): relu(bn(W2c t2) + x) and the next block's relu(bn(W2a' .)) in one
                // launch; the 4C-wide sum is written once and not read back (timed as layer i2c; the next branch2a then shows 0)
                const bool t2 = c->timing >= 2;
                if (t2) HIPE(hipEventRecord(c->cev0[i2c], st));
                // (fp32: the next block's 3x3 layer may be the fused Winograd kernel, which reads its input channel-slab major)
                const int i2b_next = i2c + (first ? 3 : 2);
                const bool slab8_next = !c->bf16 && (use_wino_fused(c, i2b_next, B) || use_wino4_fused(c, i2b_next, B));
                HIPE(run_chain(c, i2c, first, T2, cur, B, nxt, T1, st, slab8_next));
                if (t2) {
                    HIPE(hipEventRecord(c->cev1[i2c], st));
                    if (first) {  // the projection shortcut is inside the launch
                        HIPE(hipEventRecord(c->cev0[i1], st));
                        HIPE(hipEventRecord(c->cev1[i1], st));
                    }
                }
                have_2a = true;
            } else if (first && c->conv[i2c].w_dual) {
                // conv_block: expand convolution + projection shortcut + add + ReLU as one dual-source GEMM (timed as layer i2c)
                const bool t2 = c->timing >= 2;
                if (t2) HIPE(hipEventRecord(c->cev0[i2c], st));
                HIPE(run_dual(c, i2c, i1, T2, cur, B, nxt, st, cf));
                if (t2) {
                    HIPE(hipEventRecord(c->cev1[i2c], st));
                    HIPE(hipEventRecord(c->cev0[i1], st));
                    HIPE(hipEventRecord(c->cev1[i1], st));
                }
            } else {
                if (first) {
                    // projection shortcut (conv_block), no ReLU before the add
                    HIPE(timed_conv(c, i1, cur, B, nullptr, 0, SC, st, nullptr, 0, cf));
                    res = SC;
                }
                HIPE(timed_conv(c, i2c, T2, B, res, 1, nxt, st, nullptr, 0, cf));
            }
            ci += first ? 4 : 3;
            float* t = cur;
            cur = nxt;
            nxt = t;
        }
    }
    if (c->bf16) return hpe_launch_avgpool_bf16(cur, features + (size_t)i0 * ldfeat, B, 49, HPE_FEATURE_DIM, ldfeat, st);
    return hpe_launch_avgpool(cur, features + (size_t)i0 * ldfeat, B, 49, HPE_FEATURE_DIM, ldfeat, st);
}

// Batch chunks run on separate HIP streams (fork/join with events around the caller's stream): images are
// independent, so while one chunk's launch drains its last partial round of workgroups (49*2^k tiles never fill
// 256 CUs x 2 evenly) the other chunk's kernels fill the idle CUs.  Per-conv event timing (level 2) needs
// back-to-back launches on one stream and therefore runs unchunked.
hipError_t encoder_impl(hpe_ctx* c, const float* images, int B, float* features, int ldfeat, hipStream_t st) {
    int nstream = c->n_streams;
    // a chunk needs >= 32 images to keep its own launches efficient.  Rounds 1-2 had 44 (B = 64 was 7 % faster unchunked,
    // profiles/r01/g_wino_chunk_rule.txt); with the 32-cout / C-split F(4x4) launches of round 3 two chunks of 32-40 win: B = 64 / 72 / 80
    // 15.0 / 15.1 / 15.8 k img/s in two chunks against 14.1 / 13.4 / 14.1 k unchunked, B = 56 13.3 against 13.6 k, B = 40 12.5 against
    // 12.8 k (profiles/r03/chunk_rule.txt); B = 128 best with 2 chunks, B = 256 equal for 2-3, 4 chunks of 64 lose 5 %
    if (nstream > B / c->min_chunk) nstream = B / c->min_chunk;
    if (c->timing >= 2 || nstream < 2) nstream = 1;
    if (nstream == 1) return encoder_chunk(c, images, 0, B, features, ldfeat, st);
    // chunk size: about HPE_CHUNK images (default: one chunk per stream), never below min_chunk -- smaller chunks are launch bound
    //  -- and all chunks of equal size +-1; chunks go round-robin over the streams
    int nchunk = nstream;
    if (c->chunk_images > 0) {
        const int want = c->chunk_images < c->min_chunk ? c->min_chunk : c->chunk_images;
        nchunk = B / want;
        if (nchunk < nstream) nchunk = nstream;
    }
    const int per = (B + nchunk - 1) / nchunk;
    nchunk = (B + per - 1) / per;
    HIPE(hipEventRecord(c->ev_fork, st));
    for (int k = 1; k < nstream; ++k) HIPE(hipStreamWaitEvent(c->aux[k - 1], c->ev_fork, 0));
    c->co_running = nstream;
    for (int k = 0; k < nchunk; ++k) {
        const int i0 = k * per;
        const int n = (i0 + per <= B) ? per : (B - i0);
        const int sid = k % nstream;
        hipStream_t s = (sid == 0) ? st : c->aux[sid - 1];
        const hipError_t ec = encoder_chunk(c, images, i0, n, features, ldfeat, s, sid, true);
        if (ec != hipSuccess) {
            c->co_running = 1;
            return ec;
        }
    }
    c->co_running = 1;
    for (int k = 1; k < nstream; ++k) {
        HIPE(hipEventRecord(c->ev_join[k - 1], c->aux[k - 1]));
        HIPE(hipStreamWaitEvent(st, c->ev_join[k - 1], 0));
    }
    return hipSuccess;
}

// one IEF step on padded theta rows [B, THETA_LD]; P1 = features . W1[:2048] must be current
hipError_t regress_impl(hpe_ctx* c, const float* th_prev, float* th_next, int B, hipStream_t st) {
    HIPE(run_dense(c, th_prev, THETA_LD, B, THETA_LD, c->w1t, 1024, 1024, c->ones, c->b1, c->P1, 1024, 1, c->H1, 1024, st));
    HIPE(run_dense(c, c->H1, 1024, B, 1024, c->w2, 1024, 1024, c->ones, c->b2, nullptr, 0, 1, c->H2, 1024, st));
    return run_dense(c, c->H2, 1024, B, 1024, c->w3, 128, HPE_THETA_DIM, c->ones, c->b3, th_prev, THETA_LD, 0, th_next, THETA_LD, st);
}

hipError_t features_proj(hpe_ctx* c, const float* features, int B, hipStream_t st) {
    return run_dense(c, features, HPE_FEATURE_DIM, B, HPE_FEATURE_DIM, c->w1f, 1024, 1024, c->ones, c->zeros, nullptr, 0, 0, c->P1,
                     1024, st);
}

enum { NEED_ENC = 1, NEED_REG = 2, NEED_SMPL = 4 };

int check_ready(hpe_ctx* c, int B, int need) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    if (c->dead) return fail(HPE_ERR_STATE, "hpe_finalize failed on this ctx: destroy it and create a new one");
    if (!c->finalized) return fail(HPE_ERR_STATE, "hpe_finalize() has not been called");
    if (B < 1 || B > c->cfg.max_batch) return fail(HPE_ERR_INVALID, "batch " + std::to_string(B) + " outside [1, max_batch]");
    if ((need & NEED_ENC) && !c->have_encoder) return fail(HPE_ERR_STATE, "encoder weights were not loaded before hpe_finalize");
    if ((need & NEED_REG) && !c->have_regressor) return fail(HPE_ERR_STATE, "regressor weights / mean theta were not loaded");
    if ((need & NEED_SMPL) && !c->have_smpl) return fail(HPE_ERR_STATE, "SMPL model was not loaded before hpe_finalize");
    return HPE_OK;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

static void release_device_state(hpe_ctx* c);

const char* hpe_last_error(void) { return g_err.c_str(); }
const char* hpe_version(void) { return "hpe_hip 0.1 (gfx950)"; }

const char* hpe_conv_layer_name(int idx) { return (idx >= 0 && idx < HPE_NUM_CONV) ? specs()[idx].name : nullptr; }
const char* hpe_bn_layer_name(int idx) { return (idx >= 0 && idx < HPE_NUM_CONV) ? specs()[idx].bn : nullptr; }

int hpe_conv_layer_geometry(int idx, int out[7]) {
    if (idx < 0 || idx >= HPE_NUM_CONV || !out) return fail(HPE_ERR_INVALID, "bad conv index");
    const ConvSpec& s = specs()[idx];
    out[0] = s.kh;
    out[1] = s.kw;
    out[2] = s.cin;
    out[3] = s.cout;
    out[4] = s.stride;
    out[5] = s.hin;
    out[6] = s.hout;
    return HPE_OK;
}

void hpe_config_init(HpeConfig* cfg) {
    if (!cfg) return;
    cfg->struct_size = (int)sizeof(HpeConfig);
    cfg->device = 0;
    cfg->max_batch = 8;
    cfg->num_stage = 3;
    cfg->bn_eps = 1e-3f;
    cfg->encoder_dtype = 0;
    cfg->n_streams = cfg->dual_gemm = cfg->stem_fused = cfg->wino_min_c = cfg->wino_min_items = cfg->wino_fused = -1;
    cfg->wino_fused_min_hw = cfg->mesh_a2b = cfg->wino_f4 = cfg->wino4_fused = cfg->bf16_p8 = cfg->wino4_ksplit = cfg->chain_fuse = cfg->halo3 = -1;
}

int hpe_create(const HpeConfig* cfg, hpe_ctx** out) {
    if (!cfg || !out) return fail(HPE_ERR_INVALID, "null argument");
    // The struct has grown every round: a caller built against another header (or one that zero-initialised the struct instead of
    // calling hpe_config_init) is refused here instead of having plan options read from past the end of its struct.
    if (cfg->struct_size != (int)sizeof(HpeConfig))
        return fail(HPE_ERR_INVALID, "HpeConfig.struct_size is " + std::to_string(cfg->struct_size) + ", this library expects " +
                                         std::to_string(sizeof(HpeConfig)) + ": fill the struct with hpe_config_init() of the same header");
    if (cfg->n_streams > 4 || cfg->n_streams == 0) return fail(HPE_ERR_INVALID, "n_streams must be -1 (default) or 1..4");
    if (cfg->mesh_a2b > 2) return fail(HPE_ERR_INVALID, "mesh_a2b must be -1 (default), 0 (grid), 1 (valu) or 2 (mfma)");
    if (cfg->max_batch < 1 || cfg->max_batch > 1024) return fail(HPE_ERR_INVALID, "max_batch must be in [1,1024]");
    if (cfg->num_stage < 1 || cfg->num_stage > 16) return fail(HPE_ERR_INVALID, "num_stage must be in [1,16]");
    if (cfg->encoder_dtype != 0 && cfg->encoder_dtype != 1) return fail(HPE_ERR_INVALID, "encoder_dtype must be 0 (fp32) or 1 (bf16)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(HPE_ERR_NO_DEVICE, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(HPE_ERR_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HPE_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    hpe_ctx* c = new hpe_ctx();
    c->cfg = *cfg;
    c->bf16 = cfg->encoder_dtype == 1;
    if (c->cfg.bn_eps <= 0.f) c->cfg.bn_eps = 1e-3f;
    *out = c;
    return HPE_OK;
}

int hpe_destroy(hpe_ctx* c) {
    if (!c) return HPE_OK;
    DeviceGuard g(c->cfg.device);
    (void)hipDeviceSynchronize();
    release_device_state(c);
    delete c;
    return HPE_OK;
}

int hpe_load_smpl(hpe_ctx* c, const HpeSmplModel* m) {
    if (!c || !m) return fail(HPE_ERR_INVALID, "null argument");
    if (c->finalized) return fail(HPE_ERR_STATE, "already finalized");
    if (!m->v_template || !m->shapedirs || !m->posedirs || !m->J_regressor || !m->weights || !m->kp_regressor || !m->parents)
        return fail(HPE_ERR_INVALID, "null SMPL array");
    if (m->num_kp < 1 || m->num_kp > HPE_MAX_KP) return fail(HPE_ERR_INVALID, "num_kp must be in [1,24]");
    if (m->parents[0] >= 0) return fail(HPE_ERR_INVALID, "parents[0] must be negative (root)");
    for (int i = 1; i < 24; ++i)
        if (m->parents[i] < 0 || m->parents[i] >= i) return fail(HPE_ERR_INVALID, "parents[i] must satisfy 0 <= parents[i] < i");
    const int V = HPE_NUM_VERTS;
    c->h_vt.assign(m->v_template, m->v_template + V * 3);
    c->h_sd.assign(m->shapedirs, m->shapedirs + (size_t)V * 3 * 10);
    c->h_pd.assign(m->posedirs, m->posedirs + (size_t)V * 3 * 207);
    c->h_jreg.assign(m->J_regressor, m->J_regressor + (size_t)24 * V);
    c->h_w.assign(m->weights, m->weights + (size_t)V * 24);
    c->h_kreg.assign(m->kp_regressor, m->kp_regressor + (size_t)m->num_kp * V);
    c->h_par.assign(m->parents, m->parents + 24);
    c->num_kp = m->num_kp;
    c->smpl_loaded = true;
    return HPE_OK;
}

int hpe_load_conv(hpe_ctx* c, int idx, const float* kernel, const float* bias, const float* gamma, const float* beta,
                  const float* mean, const float* var) {
    if (!c || idx < 0 || idx >= HPE_NUM_CONV) return fail(HPE_ERR_INVALID, "bad conv index");
    if (c->finalized) return fail(HPE_ERR_STATE, "already finalized");
    if (!kernel || !bias || !gamma || !beta || !mean || !var) return fail(HPE_ERR_INVALID, "null conv array");
    const ConvSpec& s = specs()[idx];
    ConvLayer& L = c->conv[idx];
    L.kernel.assign(kernel, kernel + (size_t)s.kh * s.kw * s.cin * s.cout);
    L.bias.assign(bias, bias + s.cout);
    L.gamma.assign(gamma, gamma + s.cout);
    L.beta.assign(beta, beta + s.cout);
    L.mean.assign(mean, mean + s.cout);
    L.var.assign(var, var + s.cout);
    L.loaded = true;
    return HPE_OK;
}

int hpe_load_dense(hpe_ctx* c, int idx, const float* kernel, const float* bias) {
    if (!c || idx < 0 || idx >= HPE_NUM_DENSE) return fail(HPE_ERR_INVALID, "bad dense index");
    if (c->finalized) return fail(HPE_ERR_STATE, "already finalized");
    if (!kernel || !bias) return fail(HPE_ERR_INVALID, "null dense array");
    const int din[3] = {2133, 1024, 1024}, dout[3] = {1024, 1024, 85};
    c->dense[idx].kernel.assign(kernel, kernel + (size_t)din[idx] * dout[idx]);
    c->dense[idx].bias.assign(bias, bias + dout[idx]);
    c->dense[idx].loaded = true;
    return HPE_OK;
}

int hpe_load_mean_theta(hpe_ctx* c, const float* mean85) {
    if (!c || !mean85) return fail(HPE_ERR_INVALID, "null argument");
    if (c->finalized) return fail(HPE_ERR_STATE, "already finalized");
    memcpy(c->h_mean, mean85, sizeof(float) * HPE_THETA_DIM);
    c->mean_loaded = true;
    return HPE_OK;
}

static int finalize_impl(hpe_ctx* c);

// release everything a (possibly partial) hpe_finalize created
static void release_device_state(hpe_ctx* c) {
    for (void* p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    for (auto& a : c->aux)
        if (a) {
            (void)hipStreamDestroy(a);
            a = nullptr;
        }
    auto kill = [](hipEvent_t& e) {
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    };
    kill(c->ev_fork);
    kill(c->ev_enc);
    kill(c->ev_tail);
    for (auto& e : c->ev_feat_free) kill(e);
    if (c->tail_st) {
        (void)hipStreamDestroy(c->tail_st);
        c->tail_st = nullptr;
    }
    for (auto& e : c->ev_join) kill(e);
    for (auto& e : c->ev) kill(e);
    for (auto& e : c->span0) kill(e);
    for (auto& e : c->span1) kill(e);
    for (auto& e : c->cev0) kill(e);
    for (auto& e : c->cev1) kill(e);
    for (auto& e : c->lev0) kill(e);
    for (auto& e : c->lev1) kill(e);
    for (auto& e : c->lev_all) kill(e);
    c->ev_ok = false;
}

int hpe_finalize(hpe_ctx* c) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    if (c->dead) return fail(HPE_ERR_STATE, "an earlier hpe_finalize failed: destroy this ctx and create a new one");
    if (c->finalized) return fail(HPE_ERR_STATE, "already finalized");
    const int rc = finalize_impl(c);
    if (rc != HPE_OK && rc != HPE_ERR_STATE) {
        // a device-side failure part-way (out of memory, ...): nothing of the half-built state survives, so a retry cannot
        // leak it or double-allocate (the thread-local error message of the failing call is kept)
        const std::string keep = g_err;
        DeviceGuard g(c->cfg.device);
        (void)hipDeviceSynchronize();
        release_device_state(c);
        c->dead = true;
        g_err = keep;
    }
    return rc;
}

static int finalize_impl(hpe_ctx* c) {
    {
        int nconv = 0, ndense = 0;
        for (int i = 0; i < HPE_NUM_CONV; ++i) nconv += c->conv[i].loaded ? 1 : 0;
        for (int i = 0; i < HPE_NUM_DENSE; ++i) ndense += c->dense[i].loaded ? 1 : 0;
        if (nconv != 0 && nconv != HPE_NUM_CONV) {
            for (int i = 0; i < HPE_NUM_CONV; ++i)
                if (!c->conv[i].loaded) return fail(HPE_ERR_STATE, std::string("conv layer not loaded: ") + specs()[i].name);
        }
        if (ndense != 0 && (ndense != HPE_NUM_DENSE || !c->mean_loaded))
            return fail(HPE_ERR_STATE, "regressor needs all 3 dense layers and the mean theta");
        c->have_encoder = nconv == HPE_NUM_CONV;
        c->have_regressor = ndense == HPE_NUM_DENSE && c->mean_loaded;
        c->have_smpl = c->smpl_loaded;
        if (!c->have_encoder && !c->have_regressor && !c->have_smpl) return fail(HPE_ERR_STATE, "nothing was loaded");
    }
    DeviceGuard g(c->cfg.device);
    int rc;
    {
        // plan options: HpeConfig field if >= 0, else the environment variable, else the built-in default
        auto opt = [](int cfg_val, const char* env, int dflt) {
            if (cfg_val >= 0) return cfg_val;
            const char* v = getenv(env);
            return v ? atoi(v) : dflt;
        };
        c->wino_min_c = opt(c->cfg.wino_min_c, "HPE_WINO_MINC", 128);  // 0 disables the Winograd path
        c->wino_min_items = opt(c->cfg.wino_min_items, "HPE_WINO_MIN_ITEMS", 128);
        c->wino_fused = opt(c->cfg.wino_fused, "HPE_WINO_FUSED", 1) && c->wino_min_c > 0;
        c->wino_fused_min_hw = opt(c->cfg.wino_fused_min_hw, "HPE_WINO_FUSED_MINHW", 28);
        c->stem_fused = opt(c->cfg.stem_fused, "HPE_STEM_FUSED", 1);
        c->dual_gemm = opt(c->cfg.dual_gemm, "HPE_DUAL", 1);
        // F(4x4,3x3) on the 28x28 / 14x14 / 7x7 maps by default (A/B on one box: 17,720 -> 18,790 img/s; with the 7x7 and 14x14 maps only
        // 18,540; the 56x56 maps lose: their V round trip costs more than the direct kernel's extra multiplies)
        c->wino_f4 = c->wino_min_c > 0 ? opt(c->cfg.wino_f4, "HPE_WINO_F4", 7) : 0;
        c->wino4_min_items = opt(-1, "HPE_WINO4_MIN_ITEMS", c->wino4_min_items);
        c->wino4_ksplit = opt(c->cfg.wino4_ksplit, "HPE_WINO4_KSPLIT", 1);
        c->chain_fuse = c->bf16 ? (opt(c->cfg.chain_fuse, "HPE_CHAIN", 7) & 23) : (opt(c->cfg.chain_fuse, "HPE_CHAIN", 8) & 8);
        c->halo3 = c->bf16 ? (opt(c->cfg.halo3, "HPE_HALO3", 15) & 15) : 0;
        c->halo3_two = opt(-1, "HPE_HALO3_TWO", 4) & 7;
        c->wino4_fused = c->wino_min_c > 0 ? (opt(c->cfg.wino4_fused, "HPE_WINO4_FUSED", 0) & 12) : 0;
        const char* e;
        e = getenv("HPE_CONCURRENT_TILES");
        c->knobs.concurrent_tiles = e ? atoi(e) : 0;
        e = getenv("HPE_BF16_128_MIN_TILES");
        if (e) c->knobs.bf16_128_min_tiles = atoi(e);
        e = getenv("HPE_WIDE128_MIN_TILES");
        if (e) c->knobs.wide128_min_tiles = atoi(e);
        e = getenv("HPE_TILE_WIDE");
        c->knobs.force_wide = e ? atoi(e) : -1;
        e = getenv("HPE_TILE_NARROW");
        c->knobs.force_narrow = e ? atoi(e) : -1;
        e = getenv("HPE_SHORTK_TILE");
        c->knobs.shortk = e ? atoi(e) : TILE_128x64_W8;
        e = getenv("HPE_TILE_BF16");
        c->knobs.force_bf16 = e ? atoi(e) : -1;
        e = getenv("HPE_EXPAND_TILE");
        c->knobs.force_expand = e ? atoi(e) : -1;
        e = getenv("HPE_EXPAND_SMALL_GRID");
        if (e) c->knobs.expand_small_grid = atoi(e);
        e = getenv("HPE_BF16_W8_MIN_TILES");
        if (e) c->knobs.bf16_w8_min_tiles = atoi(e);
        e = getenv("HPE_NS_BF16");
        c->knobs.force_ns_bf16 = e ? atoi(e) : -1;
        e = getenv("HPE_BF16_RULES");
        c->knobs.bf16_rules = e ? atoi(e) : 1;
        c->knobs.bf16_p8 = opt(c->cfg.bf16_p8, "HPE_BF16_P8", c->knobs.bf16_p8);
        e = getenv("HPE_BF16_P8_MINN");
        if (e) c->knobs.bf16_p8_min_n = atoi(e);
        e = getenv("HPE_BF16_P8_MINK");
        if (e) c->knobs.bf16_p8_min_k = atoi(e);
        // per-device function attributes (dynamic LDS above 64 KB) of the Winograd and stem kernels
        HIP_TRY(hpe_wino_init_device());
        HIP_TRY(hpe_wino4_init_device());
        HIP_TRY(hpe_stem_fused_init_device());
        HIP_TRY(hpe_losses_init_device());
        c->mesh_a2b = c->cfg.mesh_a2b >= 0 ? c->cfg.mesh_a2b : hpe_mesh_a2b_mode_from_env();
        c->loss_attr_done = true;
    }
    // ---- conv_block (first block of a stage): out = relu(bn2c(W2c . t2) + bn1(W1 . x_strided)).  Both convolutions are 1x1,
    //      so they are ONE GEMM over the concatenated k axis once each BN scale is folded into its weights:
    //      out = relu([s2c W2c | s1 W1] . [t2 ; x] + (shift2c + shift1))   -- no shortcut tensor in HBM, one launch instead of two
    if (c->have_encoder && c->dual_gemm) {
        int ci = 1;
        const int nblk[4] = {3, 4, 6, 3};
        for (int stg = 0; stg < 4; ++stg) {
            const int i2c = ci + 2, i1 = ci + 3;
            const ConvSpec& s2 = specs()[i2c];
            const ConvSpec& s1 = specs()[i1];
            ConvLayer& L2 = c->conv[i2c];
            const ConvLayer& L1 = c->conv[i1];
            const int K1 = s2.cin, K2 = s1.cin, N = s2.cout;
            const int slab = c->bf16 ? 64 : 32;
            if (K1 % slab == 0 && K2 % slab == 0) {
                const int n_pad = round_up(N, 128), K = K1 + K2;
                std::vector<float> wt((size_t)n_pad * K, 0.f), sh(N);
                for (int n = 0; n < N; ++n) {
                    const double inv2 = (double)L2.gamma[n] / std::sqrt((double)L2.var[n] + (double)c->cfg.bn_eps);
                    const double inv1 = (double)L1.gamma[n] / std::sqrt((double)L1.var[n] + (double)c->cfg.bn_eps);
                    for (int k = 0; k < K1; ++k) wt[(size_t)n * K + k] = (float)(inv2 * (double)L2.kernel[(size_t)k * N + n]);
                    for (int k = 0; k < K2; ++k) wt[(size_t)n * K + K1 + k] = (float)(inv1 * (double)L1.kernel[(size_t)k * N + n]);
                    sh[n] = (float)((((double)L2.bias[n] - (double)L2.mean[n]) * inv2 + (double)L2.beta[n]) +
                                    (((double)L1.bias[n] - (double)L1.mean[n]) * inv1 + (double)L1.beta[n]));
                }
                if (c->bf16) {
                    std::vector<unsigned short> wb(wt.size());
                    for (size_t q = 0; q < wt.size(); ++q) wb[q] = f2bf(wt[q]);
                    void* qd = nullptr;
                    HIP_TRY(hipMalloc(&qd, wb.size() * 2));
                    c->allocs.push_back(qd);
                    HIP_TRY(hipMemcpy(qd, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
                    L2.w_dual = static_cast<float*>(qd);
                } else {
                    if ((rc = upload(c, &L2.w_dual, wt))) return rc;
                }
                if ((rc = upload(c, &L2.shift_dual, sh))) return rc;
                L2.k_dual = K;
                L2.k1_dual = K1;
            }
            ci += 4 + 3 * (nblk[stg] - 1);
        }
    }
    // ---- encoder weights: HWIO -> Wt[n][k] (k = (kh,kw,cin), cin fastest), zero padded; BN -> scale/shift
    for (int i = 0; c->have_encoder && i < HPE_NUM_CONV; ++i) {
        const ConvSpec& s = specs()[i];
        ConvLayer& L = c->conv[i];
        L.n_pad = round_up(s.cout, 128);
        if (c->bf16) {
            // bf16: 64-element slabs; stem slab s = kernel rows (2s, 2s+1), each 8 px x 4 ch
            L.k_pad = (i == 0) ? 4 * 64 : round_up(s.kh * s.kw * s.cin, 64);
            std::vector<unsigned short> wt((size_t)L.n_pad * L.k_pad, 0);
            for (int kh = 0; kh < s.kh; ++kh)
                for (int kw = 0; kw < s.kw; ++kw)
                    for (int ci = 0; ci < s.cin; ++ci) {
                        const int k = (i == 0) ? (kh * 32 + kw * 4 + ci) : ((kh * s.kw + kw) * s.cin + ci);
                        const float* src = &L.kernel[(((size_t)kh * s.kw + kw) * s.cin + ci) * s.cout];
                        for (int n = 0; n < s.cout; ++n) wt[(size_t)n * L.k_pad + k] = f2bf(src[n]);
                    }
            void* q = nullptr;
            HIP_TRY(hipMalloc(&q, wt.size() * 2));
            c->allocs.push_back(q);
            HIP_TRY(hipMemcpy(q, wt.data(), wt.size() * 2, hipMemcpyHostToDevice));
            L.w = static_cast<float*>(q);
        } else {
        L.k_pad = (i == 0) ? 7 * 32 : round_up(s.kh * s.kw * s.cin, 32);
        std::vector<float> wt((size_t)L.n_pad * L.k_pad, 0.f);
        for (int kh = 0; kh < s.kh; ++kh)
            for (int kw = 0; kw < s.kw; ++kw)
                for (int ci = 0; ci < s.cin; ++ci) {
                    const int k = (i == 0) ? (kh * 32 + kw * 4 + ci) : ((kh * s.kw + kw) * s.cin + ci);
                    const float* src = &L.kernel[(((size_t)kh * s.kw + kw) * s.cin + ci) * s.cout];
                    for (int n = 0; n < s.cout; ++n) wt[(size_t)n * L.k_pad + k] = src[n];
                }
        if ((rc = upload(c, &L.w, wt))) return rc;
        if (c->wino_min_c > 0 && s.kh == 3 && s.stride == 1 && s.cin % 32 == 0 && s.cout % 64 == 0 &&
            (s.cin >= c->wino_min_c || (c->wino_fused && s.hin >= c->wino_fused_min_hw))) {
            // U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], in double; layout [cout/64][cin/8][16][2][64][4]
            static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const int S = s.cin / 8;
            std::vector<float> U((size_t)16 * s.cin * s.cout);
            for (int ci = 0; ci < s.cin; ++ci)
                for (int n = 0; n < s.cout; ++n) {
                    double g[3][3];
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) g[a][b] = L.kernel[(((size_t)a * 3 + b) * s.cin + ci) * s.cout + n];
                    const size_t base = ((((size_t)(n >> 6) * S + (ci >> 3)) * 16) * 2 + ((ci >> 2) & 1)) * 256 + (size_t)(n & 63) * 4 + (ci & 3);
                    for (int xi = 0; xi < 4; ++xi)
                        for (int nu = 0; nu < 4; ++nu) {
                            double u = 0.0;
                            for (int a = 0; a < 3; ++a)
                                for (int b = 0; b < 3; ++b) u += G[xi][a] * G[nu][b] * g[a][b];
                            U[base + (size_t)(xi * 4 + nu) * 512] = (float)u;
                        }
                }
            if ((rc = upload(c, &L.wino_u, U))) return rc;
        }
        if (s.kh == 3 && s.stride == 1 && s.cin % 32 == 0 && s.cout % 64 == 0 && ((c->wino_f4 | c->wino4_fused) & f4_bit(s.hin))) {
            // F(4x4,3x3): U = G g G^T with G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], in double;
            // layout [cout/64][cin/4][36][64][4]
            static const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
            const int S4 = s.cin / 4;
            std::vector<float> U((size_t)36 * s.cin * s.cout);
            for (int ci = 0; ci < s.cin; ++ci)
                for (int n = 0; n < s.cout; ++n) {
                    double g[3][3];
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) g[a][b] = L.kernel[(((size_t)a * 3 + b) * s.cin + ci) * s.cout + n];
                    const size_t base = (((size_t)(n >> 6) * S4 + (ci >> 2)) * 36) * 256 + (size_t)(n & 63) * 4 + (ci & 3);
                    for (int xi = 0; xi < 6; ++xi)
                        for (int nu = 0; nu < 6; ++nu) {
                            double u = 0.0;
                            for (int a = 0; a < 3; ++a)
                                for (int b = 0; b < 3; ++b) u += G4[xi][a] * G4[nu][b] * g[a][b];
                            U[base + (size_t)(xi * 6 + nu) * 256] = (float)u;
                        }
                }
            if ((rc = upload(c, &L.wino4_u, U))) return rc;
        }
        }
        if (i == 0) {  // fused stem: same weights in the k enumeration of stem_fused.hip
            void* q = nullptr;
            if (c->bf16) {
                std::vector<unsigned short> wp((size_t)64 * 7 * 32, 0);
                for (int kh = 0; kh < 7; ++kh)
                    for (int kw = 0; kw < 7; ++kw)
                        for (int ci = 0; ci < 3; ++ci)
                            for (int n = 0; n < 64; ++n)
                                wp[((size_t)n * 7 + kh) * 32 + kw * 4 + ci] = f2bf(L.kernel[(((size_t)kh * 7 + kw) * 3 + ci) * 64 + n]);
                HIP_TRY(hipMalloc(&q, wp.size() * 2));
                c->allocs.push_back(q);
                HIP_TRY(hipMemcpy(q, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
            } else {
                std::vector<float> wp((size_t)64 * 160, 0.f);
                for (int kh = 0; kh < 7; ++kh)
                    for (int kw = 0; kw < 7; ++kw)
                        for (int ci = 0; ci < 3; ++ci)
                            for (int n = 0; n < 64; ++n)
                                wp[(size_t)n * 160 + kh * 22 + 1 + kw * 3 + ci] = L.kernel[(((size_t)kh * 7 + kw) * 3 + ci) * 64 + n];
                HIP_TRY(hipMalloc(&q, wp.size() * 4));
                c->allocs.push_back(q);
                HIP_TRY(hipMemcpy(q, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
            }
            L.stem_w = q;
        }
        std::vector<float> sc(s.cout), sh(s.cout);
        for (int n = 0; n < s.cout; ++n) {
            const double inv = (double)L.gamma[n] / std::sqrt((double)L.var[n] + (double)c->cfg.bn_eps);
            sc[n] = (float)inv;
            sh[n] = (float)(((double)L.bias[n] - (double)L.mean[n]) * inv + (double)L.beta[n]);
        }
        if ((rc = upload(c, &L.scale, sc))) return rc;
        if ((rc = upload(c, &L.shift, sh))) return rc;
        std::vector<float>().swap(L.kernel);
    }
    // constants every part uses: the zero page is the LDS-DMA source of out-of-image taps / halo pixels
    if ((rc = upload(c, &c->ones, std::vector<float>(2048, 1.f)))) return rc;
    if ((rc = upload(c, &c->zeros, std::vector<float>(1024, 0.f)))) return rc;
    // ---- regressor: Dense kernels [in,out] -> [out_pad][in_pad]; W1 split into features / theta parts
    if (c->have_regressor) {
        const std::vector<float>& k1 = c->dense[0].kernel;  // [2133][1024]
        std::vector<float> w1f((size_t)1024 * 2048), w1t((size_t)1024 * THETA_LD, 0.f);
        for (int n = 0; n < 1024; ++n) {
            for (int k = 0; k < 2048; ++k) w1f[(size_t)n * 2048 + k] = k1[(size_t)k * 1024 + n];
            for (int k = 0; k < HPE_THETA_DIM; ++k) w1t[(size_t)n * THETA_LD + k] = k1[(size_t)(2048 + k) * 1024 + n];
        }
        const std::vector<float>& k2 = c->dense[1].kernel;
        std::vector<float> w2((size_t)1024 * 1024);
        for (int n = 0; n < 1024; ++n)
            for (int k = 0; k < 1024; ++k) w2[(size_t)n * 1024 + k] = k2[(size_t)k * 1024 + n];
        const std::vector<float>& k3 = c->dense[2].kernel;  // [1024][85]
        std::vector<float> w3((size_t)128 * 1024, 0.f);
        for (int n = 0; n < HPE_THETA_DIM; ++n)
            for (int k = 0; k < 1024; ++k) w3[(size_t)n * 1024 + k] = k3[(size_t)k * HPE_THETA_DIM + n];
        if ((rc = upload(c, &c->w1f, w1f))) return rc;
        if ((rc = upload(c, &c->w1t, w1t))) return rc;
        if ((rc = upload(c, &c->w2, w2))) return rc;
        if ((rc = upload(c, &c->w3, w3))) return rc;
        if ((rc = upload(c, &c->b1, c->dense[0].bias))) return rc;
        if ((rc = upload(c, &c->b2, c->dense[1].bias))) return rc;
        std::vector<float> b3(128, 0.f);
        for (int n = 0; n < HPE_THETA_DIM; ++n) b3[n] = c->dense[2].bias[n];
        if ((rc = upload(c, &c->b3, b3))) return rc;
        if ((rc = upload(c, &c->mean_dev, std::vector<float>(c->h_mean, c->h_mean + HPE_THETA_DIM)))) return rc;
    }
    // ---- SMPL constants in kernel layouts
    if (c->have_smpl) {
        const int V = HPE_NUM_VERTS, V3 = V * 3;
        // basis source [11][V*3]: row 0 v_template, rows 1..10 shapedirs^T  (shapedirs [V,3,10] -> [10][V*3])
        std::vector<float> src((size_t)11 * V3);
        memcpy(src.data(), c->h_vt.data(), sizeof(float) * V3);
        for (int i = 0; i < V3; ++i)
            for (int k = 0; k < 10; ++k) src[(size_t)(1 + k) * V3 + i] = c->h_sd[(size_t)i * 10 + k];
        if ((rc = upload(c, &c->smpl_basis_src, src))) return rc;
        c->smpl.v_template = c->smpl_basis_src;
        c->smpl.shapedirs = c->smpl_basis_src + V3;
        // posedirs [V,3,207] -> [207][V*3]
        std::vector<float> pd((size_t)207 * V3);
        for (int i = 0; i < V3; ++i)
            for (int k = 0; k < 207; ++k) pd[(size_t)k * V3 + i] = c->h_pd[(size_t)i * 207 + k];
        float* p = nullptr;
        if ((rc = upload(c, &p, pd))) return rc;
        c->smpl.posedirs = p;
        if ((rc = upload(c, &p, c->h_w))) return rc;
        c->smpl.weights = p;
        // regressors [K,V] -> [V][24] zero padded
        std::vector<float> jr((size_t)V * SMPL_KP_PITCH, 0.f), kr((size_t)V * SMPL_KP_PITCH, 0.f);
        for (int j = 0; j < 24; ++j)
            for (int v = 0; v < V; ++v) jr[(size_t)v * SMPL_KP_PITCH + j] = c->h_jreg[(size_t)j * V + v];
        for (int j = 0; j < c->num_kp; ++j)
            for (int v = 0; v < V; ++v) kr[(size_t)v * SMPL_KP_PITCH + j] = c->h_kreg[(size_t)j * V + v];
        if ((rc = upload(c, &p, jr))) return rc;
        c->smpl.j_reg = p;
        if ((rc = upload(c, &p, kr))) return rc;
        c->smpl.kp_reg = p;
        int depth[24], maxd = 0;
        for (int j = 0; j < 24; ++j) {
            depth[j] = c->h_par[j] < 0 ? 0 : depth[c->h_par[j]] + 1;
            if (depth[j] > maxd) maxd = depth[j];
        }
        void* ip = nullptr;
        HIP_TRY(hipMalloc(&ip, sizeof(int) * 48));
        c->allocs.push_back(ip);
        HIP_TRY(hipMemcpy(ip, c->h_par.data(), sizeof(int) * 24, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(static_cast<int*>(ip) + 24, depth, sizeof(int) * 24, hipMemcpyHostToDevice));
        c->smpl.parents = static_cast<int*>(ip);
        c->smpl.depth = static_cast<int*>(ip) + 24;
        c->smpl.max_depth = maxd;
        c->smpl.num_kp = c->num_kp;
        // 24-joint basis through the 6890 -> 24 joint-regressor kernel
        float* jb = nullptr;
        if ((rc = dev_alloc(c, &jb, 11 * 24 * 3, true))) return rc;
        HIP_TRY(hpe_launch_joint_regress(c->smpl_basis_src, c->smpl.j_reg, 11, 24, jb, nullptr, nullptr, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        c->smpl.j_basis = jb;
    }
    // ---- workspace for max_batch images
    {
        const size_t B = (size_t)c->cfg.max_batch;
        const size_t Bpad = (size_t)round_up(c->cfg.max_batch, SMPL_IMG_TILE);
        if (c->have_encoder) {
            if ((rc = dev_alloc(c, &c->padded, B * STEM_HP * STEM_WP * 4 + 64, true))) return rc;
            if ((rc = dev_alloc(c, &c->X0, B * 802816, false))) return rc;
            if ((rc = dev_alloc(c, &c->X1, B * 802816, false))) return rc;
            if ((rc = dev_alloc(c, &c->SC, B * 802816, false))) return rc;
            if ((rc = dev_alloc(c, &c->T1, B * 200704, false))) return rc;
            if ((rc = dev_alloc(c, &c->T2, B * 200704, false))) return rc;
            if ((rc = dev_alloc(c, &c->feat, B * HPE_FEATURE_DIM, true))) return rc;
            if ((rc = dev_alloc(c, &c->feat_alt, B * HPE_FEATURE_DIM, true))) return rc;
        }
        if (c->have_encoder && !c->bf16 && c->wino_min_c > 0) {
            if ((rc = dev_alloc(c, &c->wino_v, B * WINO_V_PITCH + WINO_V_SLACK, false))) return rc;
            if (c->wino_f4 && c->wino4_ksplit) {
                // one workspace per chunk-stream slot (16 MB each), block counters zeroed
                const size_t nws = hpe_wino4_split_ws_floats();
                if ((rc = dev_alloc(c, &c->w4_split, 4 * nws, false))) return rc;
                for (int k = 0; k < 4; ++k) HIP_TRY(hipMemset(c->w4_split + (k + 1) * nws - 256, 0, 256 * sizeof(unsigned)));
            }
            // persistent stream-K scheduling of the Winograd GEMM: opt-in.  It removes the partial last round of workgroups
            // (-7 % on a res4 layer, -2 % on the step with HPE_STREAMS=1) but with the default batch-chunk streams, whose
            // kernels already fill those idle CUs, the step time is unchanged within noise (profiles/r01/g_wino_streamk.txt)
            const char* e = getenv("HPE_WINO_STREAMK");
            if (e && atoi(e) != 0) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, c->cfg.device));
                c->n_cu = prop.multiProcessorCount;
                if ((rc = dev_alloc(c, &c->wino_ws, (size_t)4 * c->n_cu * HPE_WINO_WS_FLOATS, false))) return rc;
                float* fl = nullptr;
                if ((rc = dev_alloc(c, &fl, (size_t)4 * c->n_cu + 4, true))) return rc;
                c->wino_flags = reinterpret_cast<unsigned*>(fl);
                c->dev_err = c->wino_flags + (size_t)4 * c->n_cu;
            }
        }
        {
            c->partial_floats = (size_t)512 * 128 * 128;  // 512 slices of the largest tile (32 MB)
            if ((rc = dev_alloc(c, &c->partial, c->partial_floats, false))) return rc;
            c->partial_tail_floats = (size_t)64 * 128 * 128;  // Dense layers: <= 16 slices of <= 64 tiles of 64 x 64 (4 MB)
            if ((rc = dev_alloc(c, &c->partial_tail, c->partial_tail_floats, false))) return rc;
        }
        if (c->have_regressor) {
            if ((rc = dev_alloc(c, &c->P1, B * 1024, true))) return rc;
            if ((rc = dev_alloc(c, &c->H1, B * 1024, true))) return rc;
            if ((rc = dev_alloc(c, &c->H2, B * 1024, true))) return rc;
            if ((rc = dev_alloc(c, &c->thA, B * THETA_LD, true))) return rc;
            if ((rc = dev_alloc(c, &c->thB, B * THETA_LD, true))) return rc;
        }
        if (c->have_smpl) {
            if ((rc = dev_alloc(c, &c->work.pfT, 207 * Bpad, true))) return rc;
            if ((rc = dev_alloc(c, &c->work.betaT, 10 * Bpad, true))) return rc;
            if ((rc = dev_alloc(c, &c->work.A, Bpad * 288, true))) return rc;
            if ((rc = dev_alloc(c, &c->work.cams, Bpad * 4, true))) return rc;
            if ((rc = dev_alloc(c, &c->work.verts_tmp, B * HPE_NUM_VERTS * 3, false))) return rc;
            if ((rc = dev_alloc(c, &c->work.kp_part, (size_t)SMPL_SMALL_B * ((HPE_NUM_VERTS + 63) / 64) * 72, true))) return rc;
            // reprojection-loss workspace for the geometry the path itself produces (config 5); other sizes grow it on demand
            c->loss_ws_floats = hpe_mesh_loss_ws_floats(c->cfg.max_batch, HPE_IMG_SIZE, HPE_IMG_SIZE, HPE_NUM_VERTS);
            if ((rc = dev_alloc(c, &c->loss_ws, c->loss_ws_floats, true))) return rc;
        }
        c->work.Bpad = (int)Bpad;
    }
    {
        // Two chunk streams by default: with the tail stream of the pipelined forward that makes 3 busy queues per process, and a
        // 4th for RCCL.  A 5th concurrently busy queue is expensive on this part whatever GPU_MAX_HW_QUEUES says -- with a process
        // group alive 3 chunk streams cost 6 % in fp32 and 24 % in bf16 (profiles/r02/streams_vs_rccl.txt) -- while 2 and 3 chunk
        // streams are equal without one (17,306 vs 17,337 img/s).
        const char* e = getenv("HPE_STREAMS");
        int ns = c->cfg.n_streams > 0 ? c->cfg.n_streams : (e ? atoi(e) : 2);
        if (ns < 1) ns = 1;
        if (ns > 4) ns = 4;
        c->n_streams = ns;
        // bf16 launches are short enough to leave CUs idle at small batches: two chunks pay from 2 x 24 images on (B = 48 / 64 / 80:
        // 38.5 / 44.8 / 48.9 k img/s against 34.7 / 39.2 / 44.5 k as one chunk); fp32 from 2 x 32 (see encoder_impl)
        c->min_chunk = c->bf16 ? 24 : 32;
        e = getenv("HPE_MIN_CHUNK");
        if (e && atoi(e) >= 8) c->min_chunk = atoi(e);
        e = getenv("HPE_CHUNK");
        c->chunk_images = e ? atoi(e) : 0;
        for (int i = 0; i < ns - 1; ++i) HIP_TRY(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&c->tail_st, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_enc, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
        for (auto& ev : c->ev_feat_free) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (auto& ev : c->ev_join) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->span0) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->span1) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->cev0) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->cev1) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->lev0) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->lev1) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->lev_all) HIP_TRY(hipEventCreate(&e));
    c->ev_ok = true;
    HIP_TRY(hipDeviceSynchronize());
    c->finalized = true;
    return HPE_OK;
}

int hpe_encoder(hpe_ctx* c, const float* images, int B, float* features, void* stream) {
    int rc = check_ready(c, B, NEED_ENC);
    if (rc) return rc;
    if (!images || !features) return fail(HPE_ERR_INVALID, "null pointer");
    DeviceGuard g(c->cfg.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[0], st));
        HIP_TRY(hipEventRecord(c->span0[c->span_n % hpe_ctx::SPAN_RING], st));
    }
    HIP_TRY(encoder_impl(c, images, B, features, HPE_FEATURE_DIM, st));
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->span1[c->span_n % hpe_ctx::SPAN_RING], st));
        ++c->span_n;
        HIP_TRY(hipEventRecord(c->ev[1], st));
        HIP_TRY(hipEventRecord(c->ev[4], st));
        c->timed_valid = true;
        c->conv_timed_valid = c->timing >= 2;
    }
    return HPE_OK;
}

int hpe_regress_stage(hpe_ctx* c, const float* features, const float* theta_prev, int B, float* theta_out, void* stream) {
    int rc = check_ready(c, B, NEED_REG);
    if (rc) return rc;
    if (!features || !theta_out) return fail(HPE_ERR_INVALID, "null pointer");
    DeviceGuard g(c->cfg.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->tail_pending) {  // shares the regressor buffers with a pipelined call's tail
        HIP_TRY(hipStreamWaitEvent(st, c->ev_tail, 0));
        c->tail_pending = false;
    }
    HIP_TRY(features_proj(c, features, B, st));
    if (theta_prev)
        HIP_TRY(hpe_launch_copy_theta(theta_prev, HPE_THETA_DIM, c->thA, THETA_LD, B, HPE_THETA_DIM, st));
    else
        HIP_TRY(hpe_launch_tile_theta(c->mean_dev, c->thA, B, THETA_LD, st));
    HIP_TRY(regress_impl(c, c->thA, c->thB, B, st));
    HIP_TRY(hpe_launch_copy_theta(c->thB, THETA_LD, theta_out, HPE_THETA_DIM, B, HPE_THETA_DIM, st));
    return HPE_OK;
}

int hpe_smpl(hpe_ctx* c, const float* theta, int B, const HpeOutputs* outs, void* stream) {
    int rc = check_ready(c, B, NEED_SMPL);
    if (rc) return rc;
    if (!theta || !outs) return fail(HPE_ERR_INVALID, "null pointer");
    DeviceGuard g(c->cfg.device);
    if (c->tail_pending) {  // shares the SMPL work buffers with a pipelined call's tail
        HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->ev_tail, 0));
        c->tail_pending = false;
    }
    HIP_TRY(hpe_launch_smpl(c->smpl, c->work, theta, HPE_THETA_DIM, B, outs, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

// features [B,2048] -> feature projection (hoisted W1 block), then num_stage x (regressor step, SMPL of the stages that are returned);
// feat_free (optional) is recorded once the features have been consumed
static hipError_t tail_impl(hpe_ctx* c, const float* feat, int B, const HpeOutputs* stage_outs, int n_outs, hipStream_t ts, hipEvent_t feat_free) {
    hipError_t e = features_proj(c, feat, B, ts);
    if (e == hipSuccess && feat_free) e = hipEventRecord(feat_free, ts);
    if (e == hipSuccess) e = hpe_launch_tile_theta(c->mean_dev, c->thA, B, THETA_LD, ts);
    float* prev = c->thA;
    float* next = c->thB;
    const int first_out = c->cfg.num_stage - n_outs;
    for (int s = 0; e == hipSuccess && s < c->cfg.num_stage; ++s) {
        e = regress_impl(c, prev, next, B, ts);
        if (e == hipSuccess && s >= first_out) e = hpe_launch_smpl(c->smpl, c->work, next, THETA_LD, B, &stage_outs[s - first_out], ts);
        float* t = prev;
        prev = next;
        next = t;
    }
    return e;
}

// encoder on `st`; regressor + SMPL stages on `st` (pipelined == false) or on the ctx's tail stream behind an event (true)
static int forward_impl(hpe_ctx* c, const float* images, int B, const HpeOutputs* stage_outs, int n_outs, hipStream_t st, bool pipelined) {
    int rc = check_ready(c, B, NEED_ENC | NEED_REG | NEED_SMPL);
    if (rc) return rc;
    if (!images || !stage_outs) return fail(HPE_ERR_INVALID, "null pointer");
    if (n_outs < 1 || n_outs > c->cfg.num_stage) return fail(HPE_ERR_INVALID, "n_outs must be in [1, num_stage]");
    DeviceGuard g(c->cfg.device);
    const bool tm = c->timing != 0;
    if (c->timing >= 2) pipelined = false;  // per-launch event timing wants one serial stream
    float* feat = c->feat;
    hipStream_t ts = st;
    if (pipelined) {
        // features alternate between two buffers: the tail of batch k reads one while the encoder of batch k+1 fills the other;
        // the buffer used now was last read by the feature projection of two calls ago (long finished: the wait is a formality)
        const unsigned slot = c->pipe_idx & 1u;
        feat = slot ? c->feat_alt : c->feat;
        if (c->feat_free_valid[slot]) HIP_TRY(hipStreamWaitEvent(st, c->ev_feat_free[slot], 0));
        ts = c->tail_st;
    } else if (c->tail_pending) {
        // a serial call after pipelined ones: its tail shares buffers with the pending tail -> order them
        HIP_TRY(hipStreamWaitEvent(st, c->ev_tail, 0));
        c->tail_pending = false;
    }
    if (tm) {
        HIP_TRY(hipEventRecord(c->ev[0], st));
        HIP_TRY(hipEventRecord(c->span0[c->span_n % hpe_ctx::SPAN_RING], st));
    }
    HIP_TRY(encoder_impl(c, images, B, feat, HPE_FEATURE_DIM, st));
    if (tm) {
        HIP_TRY(hipEventRecord(c->ev[1], st));
        HIP_TRY(hipEventRecord(c->span1[c->span_n % hpe_ctx::SPAN_RING], st));
        ++c->span_n;
    }
    if (pipelined) {
        HIP_TRY(hipEventRecord(c->ev_enc, st));
        HIP_TRY(hipStreamWaitEvent(ts, c->ev_enc, 0));
        c->dense_on_tail = true;
    }
    hipEvent_t feat_free = nullptr;
    if (pipelined) {
        const unsigned slot = c->pipe_idx & 1u;
        feat_free = c->ev_feat_free[slot];
        c->feat_free_valid[slot] = true;
    }
    hipError_t e = tail_impl(c, feat, B, stage_outs, n_outs, ts, feat_free);
    c->dense_on_tail = false;
    if (e != hipSuccess) return fail(HPE_ERR_HIP, std::string("forward tail: ") + hipGetErrorString(e));
    if (pipelined) {
        HIP_TRY(hipEventRecord(c->ev_tail, ts));
        c->tail_pending = true;
        ++c->pipe_idx;
    }
    if (tm) {
        HIP_TRY(hipEventRecord(c->ev[4], ts));
        c->timed_valid = true;
        c->conv_timed_valid = c->timing >= 2;
    }
    return HPE_OK;
}

int hpe_forward(hpe_ctx* c, const float* images, int B, const HpeOutputs* stage_outs, int n_outs, void* stream) {
    return forward_impl(c, images, B, stage_outs, n_outs, static_cast<hipStream_t>(stream), false);
}

int hpe_forward_pipelined(hpe_ctx* c, const float* images, int B, const HpeOutputs* stage_outs, int n_outs, void* stream) {
    return forward_impl(c, images, B, stage_outs, n_outs, static_cast<hipStream_t>(stream), true);
}

int hpe_tail(hpe_ctx* c, const float* features, int B, const HpeOutputs* stage_outs, int n_outs, void* stream) {
    int rc = check_ready(c, B, NEED_REG | NEED_SMPL);
    if (rc) return rc;
    if (!features || !stage_outs) return fail(HPE_ERR_INVALID, "null pointer");
    if (n_outs < 1 || n_outs > c->cfg.num_stage) return fail(HPE_ERR_INVALID, "n_outs must be in [1, num_stage]");
    DeviceGuard g(c->cfg.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->tail_pending) {  // shares the regressor / SMPL buffers with a pipelined call's tail
        HIP_TRY(hipStreamWaitEvent(st, c->ev_tail, 0));
        c->tail_pending = false;
    }
    c->dense_on_tail = true;  // its own split-K workspace: may overlap hpe_encoder of the next batch
    const hipError_t e = tail_impl(c, features, B, stage_outs, n_outs, st, nullptr);
    c->dense_on_tail = false;
    if (e != hipSuccess) return fail(HPE_ERR_HIP, std::string("hpe_tail: ") + hipGetErrorString(e));
    return HPE_OK;
}

int hpe_join(hpe_ctx* c, void* stream) {
    if (!c || !c->finalized) return fail(HPE_ERR_STATE, "needs a finalized ctx");
    DeviceGuard g(c->cfg.device);
    if (c->tail_pending) HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->ev_tail, 0));
    return HPE_OK;
}

void* hpe_tail_stream(hpe_ctx* c) { return (c && c->finalized) ? static_cast<void*>(c->tail_st) : nullptr; }

int hpe_orth_proj(const float* X, const float* cam, int B, int P, float* out, void* stream) {
    if (!X || !cam || !out || B < 1 || P < 1) return fail(HPE_ERR_INVALID, "bad argument");
    HIP_TRY(hpe_launch_orth_proj(X, cam, B, P, 0.f, 0.f, 0, out, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_reproject_vertices(const float* verts, const float* cam, int B, int P, float im_w, float im_h, float* out, void* stream) {
    if (!verts || !cam || !out || B < 1 || P < 1) return fail(HPE_ERR_INVALID, "bad argument");
    HIP_TRY(hpe_launch_orth_proj(verts, cam, B, P, im_w, im_h, 1, out, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

// preview.py:22-29 / image.py:7-15,17-39 -- all index arithmetic in double like numpy; false if the frame is too thin
static bool preprocess_geometry(int H, int W, PreprocFrame* f, int proc_param[5]) {
    const int S = HPE_IMG_SIZE;
    const int mx = H > W ? H : W;
    const double scale = (mx != S) ? ((double)S / (double)mx) : 1.0;
    const int newH = (int)std::floor(H * scale), newW = (int)std::floor(W * scale);
    if (newH < 1 || newW < 1) return false;
    const double fy = (double)newH / (double)H, fx = (double)newW / (double)W;  // actual_factor [y, x]
    const double cy = std::nearbyint(H / 2.0), cx = std::nearbyint(W / 2.0);     // np.round: half to even
    const int csx = (int)std::nearbyint(cx * fx), csy = (int)std::nearbyint(cy * fy);
    const int margin = S / 2;
    const int start_x = csx + margin - margin, start_y = csy + margin - margin;  // center_pad - margin
    proc_param[0] = start_x;
    proc_param[1] = start_y;
    proc_param[2] = start_x + 2 * margin;
    proc_param[3] = start_y + 2 * margin;
    proc_param[4] = S;
    f->H = H;
    f->W = W;
    f->newH = newH;
    f->newW = newW;
    f->start_x = start_x;
    f->start_y = start_y;
    return true;
}

int hpe_preprocess_u8(const unsigned char* img, int H, int W, int C, float* out224, int proc_param[5], void* stream) {
    if (!img || !out224 || !proc_param || H < 1 || W < 1 || (C != 3 && C != 4)) return fail(HPE_ERR_INVALID, "bad argument");
    PreprocFrame f{};
    if (!preprocess_geometry(H, W, &f, proc_param)) return fail(HPE_ERR_INVALID, "image too thin");
    HIP_TRY(hpe_launch_preprocess_u8(img, H, W, C, f.newH, f.newW, f.start_x, f.start_y, HPE_IMG_SIZE / 2, out224, HPE_IMG_SIZE,
                                     static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_preprocess_u8_batch(const unsigned char* frames, const long long* offsets, const int* sizes_hw, int B, int C, float* out,
                            int* proc_params, void* table_dev, void* stream) {
    if (!frames || !sizes_hw || !out || !proc_params || B < 1 || (C != 3 && C != 4)) return fail(HPE_ERR_INVALID, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!offsets) {
        PreprocFrame f{};
        if (sizes_hw[0] < 1 || sizes_hw[1] < 1 || !preprocess_geometry(sizes_hw[0], sizes_hw[1], &f, proc_params))
            return fail(HPE_ERR_INVALID, "bad frame size");
        for (int b = 1; b < B; ++b) memcpy(proc_params + 5 * b, proc_params, 5 * sizeof(int));
        HIP_TRY(hpe_launch_preprocess_u8_batch(frames, nullptr, f, B, C, HPE_IMG_SIZE / 2, out, HPE_IMG_SIZE, st));
        return HPE_OK;
    }
    if (!table_dev) return fail(HPE_ERR_INVALID, "per-image sizes need table_dev (32 * B bytes of device scratch)");
    static_assert(sizeof(PreprocFrame) == 32, "table_dev is documented as 32 bytes per frame");
    std::vector<PreprocFrame> tab((size_t)B);
    for (int b = 0; b < B; ++b) {
        if (offsets[b] < 0 || sizes_hw[2 * b] < 1 || sizes_hw[2 * b + 1] < 1 ||
            !preprocess_geometry(sizes_hw[2 * b], sizes_hw[2 * b + 1], &tab[b], proc_params + 5 * b))
            return fail(HPE_ERR_INVALID, "bad frame " + std::to_string(b));
        tab[b].offset = offsets[b];
    }
    // pageable source: the runtime stages the bytes before it returns, so `tab` may die at the end of this call
    // The table lives on this call's stack: the copy must have READ it before the call returns.  hipMemcpyAsync from pageable memory
    // happens to stage the bytes before returning on this runtime, but that is not a contract of the API -- so: wait for the work already
    // on `st` (the scratch may still be read by an earlier launch), then a synchronous copy.  This path (frames of different sizes) is
    // host-blocking and cannot be captured; the uniform-size path above needs no table at all.
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(table_dev, tab.data(), tab.size() * sizeof(PreprocFrame), hipMemcpyHostToDevice));
    HIP_TRY(hpe_launch_preprocess_u8_batch(frames, static_cast<const PreprocFrame*>(table_dev), PreprocFrame{}, B, C, HPE_IMG_SIZE / 2, out,
                                           HPE_IMG_SIZE, st));
    return HPE_OK;
}

int hpe_get_original(const float* verts, const float* cam, int B, int P, int K, const int start_pt[2], float scale, int img_size,
                     float* vert_shifted, float cam_for_render[3], float* kp_original_host, const float* joints2d_host,
                     void* stream) {
    if (!verts || !cam || !vert_shifted || !cam_for_render || !start_pt || B < 1 || P < 1 || scale <= 0.f || img_size < 1)
        return fail(HPE_ERR_INVALID, "bad argument");
    const float flength = 500.f;
    const float undo = 1.f / scale;
    HIP_TRY(hpe_launch_shift_verts(verts, cam, B, P, flength, (float)img_size, vert_shifted, static_cast<hipStream_t>(stream)));
    // renderer.py:273-276
    const float pp = img_size / 2.f;
    cam_for_render[0] = flength * undo;
    cam_for_render[1] = (pp + (start_pt[0] - 0.5f * img_size)) * undo;
    cam_for_render[2] = (pp + (start_pt[1] - 0.5f * img_size)) * undo;
    // renderer.py:281-282: kp_original = (joints + start_pt - margin) * undo_scale (host arrays, K x 2 per image)
    if (kp_original_host && joints2d_host) {
        const int margin = img_size / 2;
        for (int i = 0; i < B * K; ++i) {
            kp_original_host[2 * i] = (joints2d_host[2 * i] + start_pt[0] - margin) * undo;
            kp_original_host[2 * i + 1] = (joints2d_host[2 * i + 1] + start_pt[1] - margin) * undo;
        }
    }
    return HPE_OK;
}

int hpe_kp_loss(const float* kp_gt, const float* kp_pred, int B, int K, float* out, void* stream) {
    if (!kp_gt || !kp_pred || !out || B < 1 || K < 1) return fail(HPE_ERR_INVALID, "bad argument");
    HIP_TRY(hpe_launch_kp_loss(kp_gt, kp_pred, B * K, out, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

// Loss workspace: sized in hpe_finalize for max_batch images of 224 x 224 and 6890 vertices (what the path produces); any
// other geometry grows it here -- the one case in which a compute call synchronises (documented in hpe.h).
static int ensure_loss_ws(hpe_ctx* c, int B, int H, int W, int P) {
    if (!c->loss_attr_done) {
        // a ctx that was never finalized (loss operators only): the search kernel's dynamic-LDS attribute is set here
        HIP_TRY(hpe_losses_init_device());
        c->mesh_a2b = c->cfg.mesh_a2b >= 0 ? c->cfg.mesh_a2b : hpe_mesh_a2b_mode_from_env();
        c->loss_attr_done = true;
    }
    const size_t need = hpe_mesh_loss_ws_floats(B, H, W, P);
    if (need <= c->loss_ws_floats) return HPE_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (c->loss_ws) {
        for (auto it = c->allocs.begin(); it != c->allocs.end(); ++it)
            if (*it == c->loss_ws) {
                c->allocs.erase(it);
                break;
            }
        (void)hipFree(c->loss_ws);
        c->loss_ws = nullptr;
        c->loss_ws_floats = 0;
    }
    float* p = nullptr;
    int rc = dev_alloc(c, &p, need, true);
    if (rc) return rc;
    c->loss_ws = p;
    c->loss_ws_floats = need;
    return HPE_OK;
}

int hpe_mesh_loss(hpe_ctx* c, const float* seg, const float* verts2d, int B, int H, int W, int P, float* out, void* stream) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    if (!seg || !verts2d || !out || B < 1 || H < 1 || W < 1 || P < 1) return fail(HPE_ERR_INVALID, "bad argument");
    DeviceGuard g(c->cfg.device);
    int rc = ensure_loss_ws(c, B, H, W, P);
    if (rc) return rc;
    HIP_TRY(hpe_launch_mesh_loss(seg, verts2d, B, H, W, P, c->loss_ws, out, static_cast<hipStream_t>(stream), c->mesh_a2b, c->loss_counter));
    return HPE_OK;
}

int hpe_val_losses(hpe_ctx* c, const float* seg, const float* kp_gt, const float* const* kp2d, const float* const* verts2d, int n_stage,
                   int B, int K, int H, int W, int P, float* out, void* stream) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    if (!kp_gt || !kp2d || !out || n_stage < 1 || n_stage > 16 || B < 1 || K < 1) return fail(HPE_ERR_INVALID, "bad argument");
    const bool mesh = seg && verts2d;
    if (mesh && (H < 1 || W < 1 || P < 1)) return fail(HPE_ERR_INVALID, "bad silhouette / vertex geometry");
    for (int s = 0; s < n_stage; ++s)
        if (!kp2d[s] || (mesh && !verts2d[s])) return fail(HPE_ERR_INVALID, "null stage pointer");
    DeviceGuard g(c->cfg.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tm = c->timing != 0 && c->ev_ok;
    c->loss_timed_stages = 0;
    if (tm) HIP_TRY(hipEventRecord(c->lev_all[0], st));
    if (mesh) {
        int rc = ensure_loss_ws(c, B, H, W, P);
        if (rc) return rc;
        HIP_TRY(hpe_launch_mesh_loss_prepare(seg, B, H, W, P, c->loss_ws, st));
    }
    for (int s = 0; s < n_stage; ++s) {
        HIP_TRY(hpe_launch_kp_loss(kp_gt, kp2d[s], B * K, out + 4 * s, st));  // writes out[4s .. 4s+2]
        if (mesh)
            HIP_TRY(hpe_launch_mesh_loss_search(verts2d[s], B, H, W, P, c->loss_ws, out + 4 * s + 3, st, tm ? c->lev0[s] : nullptr,
                                                tm ? c->lev1[s] : nullptr, c->mesh_a2b, c->loss_counter));
        else
            HIP_TRY(hipMemsetAsync(out + 4 * s + 3, 0, sizeof(float), st));
    }
    if (tm) {
        HIP_TRY(hipEventRecord(c->lev_all[1], st));
        c->loss_timed_stages = mesh ? n_stage : -1;
    }
    return HPE_OK;
}

int hpe_debug_conv(hpe_ctx* c, int idx, const float* x, int B, const float* residual, int relu, float* y, void* stream) {
    int rc = check_ready(c, B, NEED_ENC);
    if (rc) return rc;
    if (idx < 0 || idx >= HPE_NUM_CONV || !x || !y) return fail(HPE_ERR_INVALID, "bad argument");
    DeviceGuard g(c->cfg.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->bf16) {
        if (idx == 0) return fail(HPE_ERR_STATE, "hpe_debug_conv: conv1 of a bf16 context runs inside the fused stem only");
        const ConvSpec& s = specs()[idx];
        const long nin = (long)B * s.hin * s.hin * s.cin, nout = (long)B * s.hout * s.hout * s.cout;
        HIP_TRY(hpe_launch_f32_to_bf16(x, c->X0, nin, st));
        if (residual) HIP_TRY(hpe_launch_f32_to_bf16(residual, c->SC, nout, st));
        HIP_TRY(run_conv(c, idx, c->X0, B, residual ? c->SC : nullptr, relu, c->X1, st));
        HIP_TRY(hpe_launch_bf16_to_f32(c->X1, y, nout, st));
        return HPE_OK;
    }
    const float* in = x;
    if (idx == 0) {
        HIP_TRY(hpe_launch_pad_input(x, c->padded, B, HPE_IMG_SIZE, HPE_IMG_SIZE, STEM_HP, STEM_WP, st));
        in = c->padded;
    }
    if (!residual && (use_wino_fused(c, idx, B) || use_wino4_fused(c, idx, B))) {
        // the fused Winograd kernel reads channel-slab major input; in the network its 1x1 producer writes that directly
        const ConvSpec& s = specs()[idx];
        HIP_TRY(hpe_launch_nhwc_to_slab8(in, c->T1, (long)B * s.hin * s.hin, s.cin, st));
        HIP_TRY(run_conv(c, idx, c->T1, B, nullptr, relu, y, st, nullptr, 0, CONV_IN_SLAB8));
        return HPE_OK;
    }
    HIP_TRY(run_conv(c, idx, in, B, residual, relu, y, st, c->wino_v));
    return HPE_OK;
}

int hpe_debug_chain(hpe_ctx* c, int idx2c, const float* t2, const float* residual, int B, float* t3, float* u1, int* occupancy, void* stream) {
    int rc = check_ready(c, B, NEED_ENC);
    if (rc) return rc;
    if (idx2c < 1 || idx2c + 2 >= HPE_NUM_CONV || !t2 || !residual || !t3 || !u1) return fail(HPE_ERR_INVALID, "bad argument");
    const ConvSpec& s2 = specs()[idx2c];
    if (!c->bf16) {
        // fp32 contexts: the identity blocks of stage 2 (conv_chain_f32.hip); operands used in place, u1 row-major
        const ConvSpec& sn1 = specs()[idx2c + 1];
        if (s2.kh != 1 || sn1.kh != 1 || sn1.stride != 1 || sn1.cin != s2.cout || !hpe_chain_f32_supported(s2.cin, s2.cout, sn1.cout))
            return fail(HPE_ERR_INVALID, "hpe_debug_chain (fp32): idx2c must be res2b_branch2c (an identity block of stage 2 followed by an identity block)");
        DeviceGuard g32(c->cfg.device);
        HIP_TRY(run_chain(c, idx2c, false, t2, residual, B, t3, u1, static_cast<hipStream_t>(stream), false));
        if (occupancy) {
            occupancy[0] = occupancy[1] = occupancy[2] = 0;
            HIP_TRY(hpe_chain_f32_occupancy(&occupancy[0]));
        }
        return HPE_OK;
    }
    // the conv_block form when idx2c is the branch2c of a block's first unit (its branch1 follows in the layer table)
    const bool first = specs()[idx2c + 1].kh == 1 && specs()[idx2c + 1].cout == s2.cout && strstr(specs()[idx2c + 1].name, "branch1") != nullptr;
    const ConvSpec& sn = specs()[idx2c + (first ? 2 : 1)];
    const int C2 = first ? specs()[idx2c + 1].cin : 0;
    if (s2.kh != 1 || s2.cout != 4 * s2.cin || sn.kh != 1 || sn.stride != 1 || sn.cin != s2.cout || !hpe_chain_bf16_supported(s2.cin, s2.cout, sn.cout, C2) ||
        (first && (!c->conv[idx2c].w_dual || specs()[idx2c + 1].stride != 1)))
        return fail(HPE_ERR_INVALID, "hpe_debug_chain: idx2c must be the branch2c of a stage-2 / stage-3 block that is followed by an identity block");
    DeviceGuard g(c->cfg.device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long M = (long)B * s2.hout * s2.hout;
    HIP_TRY(hpe_launch_f32_to_bf16(t2, c->T2, M * s2.cin, st));
    HIP_TRY(hpe_launch_f32_to_bf16(residual, c->X0, M * (first ? C2 : s2.cout), st));
    HIP_TRY(run_chain(c, idx2c, first, c->T2, c->X0, B, c->X1, c->T1, st));
    HIP_TRY(hpe_launch_bf16_to_f32(c->X1, t3, M * s2.cout, st));
    HIP_TRY(hpe_launch_bf16_to_f32(c->T1, u1, M * sn.cout, st));
    if (occupancy) HIP_TRY(hpe_chain_bf16_occupancy(occupancy));
    return HPE_OK;
}

int hpe_debug_stem(hpe_ctx* c, const float* images, int B, int rows_per_strip, float* y, void* stream) {
    int rc = check_ready(c, B, NEED_ENC);
    if (rc) return rc;
    if (!images || !y) return fail(HPE_ERR_INVALID, "null pointer");
    if (c->bf16) return fail(HPE_ERR_STATE, "hpe_debug_stem works on fp32 contexts only");
    DeviceGuard g(c->cfg.device);
    const int R = rows_per_strip > 0 ? rows_per_strip : hpe_stem_fused_pick_rows(B);
    HIP_TRY(hpe_launch_stem_fused(images, c->conv[0].stem_w, c->conv[0].scale, c->conv[0].shift, y, B, R, 0, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_debug_gemm(hpe_ctx* c, const float* x, const float* wt, int M, int N, int K, int w_rows, int tile, const float* residual,
                   int relu, float* y, void* stream) {
    if (!c || !c->finalized || !c->have_regressor) return fail(HPE_ERR_STATE, "needs a finalized ctx with the regressor loaded");
    if (!x || !wt || !y || N > 1024) return fail(HPE_ERR_INVALID, "bad argument (N <= 1024)");
    DeviceGuard g(c->cfg.device);
    GemmArgs p{};
    p.x = x;
    p.w = wt;
    p.scale = c->ones;
    p.shift = c->zeros;
    p.res = residual;
    p.y = y;
    p.M = M;
    p.N = N;
    p.K = K;
    p.lda = K;
    p.ldw = K;
    p.w_rows = w_rows;
    p.ldy = N;
    p.ldres = N;
    p.relu = relu;
    p.zero = c->zeros;
    p.dbg = c->dbg;
    HIP_TRY(hpe_launch_gemm(p, GEMM_DENSE, tile, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_debug_set_dbg(hpe_ctx* c, void* dbg_dev) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    c->dbg = static_cast<unsigned long long*>(dbg_dev);
    return HPE_OK;
}

int hpe_debug_maxpool(const float* x, int B, int H, int C, float* y, void* stream) {
    if (!x || !y || B < 1) return fail(HPE_ERR_INVALID, "bad argument");
    HIP_TRY(hpe_launch_maxpool(x, y, B, H, C, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_debug_avgpool(const float* x, int B, int HW, int C, float* y, void* stream) {
    if (!x || !y || B < 1) return fail(HPE_ERR_INVALID, "bad argument");
    HIP_TRY(hpe_launch_avgpool(x, y, B, HW, C, C, static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_debug_joint_regress(hpe_ctx* c, const float* X, int n, int use_kp, float* out, void* stream) {
    if (!c || !c->finalized || !c->have_smpl) return fail(HPE_ERR_STATE, "SMPL not finalized");
    if (!X || !out || n < 1) return fail(HPE_ERR_INVALID, "bad argument");
    DeviceGuard g(c->cfg.device);
    HIP_TRY(hpe_launch_joint_regress(X, use_kp ? c->smpl.kp_reg : c->smpl.j_reg, n, use_kp ? c->num_kp : 24, out, nullptr, nullptr,
                                     static_cast<hipStream_t>(stream)));
    return HPE_OK;
}

int hpe_device_status(hpe_ctx* c, void* stream) {
    if (!c || !c->finalized) return fail(HPE_ERR_STATE, "needs a finalized ctx");
    DeviceGuard g(c->cfg.device);
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (c->dev_err) {
        unsigned w = 0;
        HIP_TRY(hipMemcpy(&w, c->dev_err, sizeof w, hipMemcpyDeviceToHost));
        if (w) {
            (void)hipMemset(c->dev_err, 0, sizeof w);
            return fail(HPE_ERR_HIP, "device error word " + std::to_string(w) +
                                         ": a stream-K wait of the Winograd GEMM timed out, outputs since the last check are invalid");
        }
    }
    return HPE_OK;
}

int hpe_enable_timing(hpe_ctx* c, int enable) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    c->timing = enable;
    c->span_n = 0;
    c->timed_valid = false;
    c->conv_timed_valid = false;
    return HPE_OK;
}

int hpe_get_timings(hpe_ctx* c, float ms[5]) {
    if (!c || !ms) return fail(HPE_ERR_INVALID, "null argument");
    if (!c->timed_valid) return fail(HPE_ERR_STATE, "no timed call recorded (hpe_enable_timing first)");
    DeviceGuard g(c->cfg.device);
    HIP_TRY(hipEventSynchronize(c->ev[4]));
    for (int i = 0; i < 5; ++i) ms[i] = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms[4], c->ev[0], c->ev[4]));
    if (c->conv_timed_valid) {
        for (int i = 0; i < HPE_NUM_CONV; ++i) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, c->cev0[i], c->cev1[i]));
            ms[1] += t;
        }
    }
    ms[2] = ms[4] - ms[0];  // regressor + SMPL stages
    return HPE_OK;
}

int hpe_get_span_stats(hpe_ctx* c, float ms[3], int* n_calls) {
    if (!c || !ms || !n_calls) return fail(HPE_ERR_INVALID, "null argument");
    if (c->span_n == 0) return fail(HPE_ERR_STATE, "no timed call recorded (hpe_enable_timing first)");
    DeviceGuard g(c->cfg.device);
    const unsigned n = c->span_n < (unsigned)hpe_ctx::SPAN_RING ? c->span_n : (unsigned)hpe_ctx::SPAN_RING;
    double sum = 0.0;
    float lo = 1e30f, hi = 0.f;
    for (unsigned i = 0; i < n; ++i) {
        float t = 0.f;
        HIP_TRY(hipEventSynchronize(c->span1[i]));
        HIP_TRY(hipEventElapsedTime(&t, c->span0[i], c->span1[i]));
        sum += t;
        lo = t < lo ? t : lo;
        hi = t > hi ? t : hi;
    }
    ms[0] = (float)(sum / n);
    ms[1] = lo;
    ms[2] = hi;
    *n_calls = (int)n;
    return HPE_OK;
}

int hpe_get_loss_timings(hpe_ctx* c, float ms[2]) {
    if (!c || !ms) return fail(HPE_ERR_INVALID, "null argument");
    if (c->loss_timed_stages == 0) return fail(HPE_ERR_STATE, "no timed hpe_val_losses call recorded (hpe_enable_timing first)");
    DeviceGuard g(c->cfg.device);
    HIP_TRY(hipEventSynchronize(c->lev_all[1]));
    HIP_TRY(hipEventElapsedTime(&ms[0], c->lev_all[0], c->lev_all[1]));
    ms[1] = 0.f;
    for (int s = 0; s < c->loss_timed_stages; ++s) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, c->lev0[s], c->lev1[s]));
        ms[1] += t;
    }
    return HPE_OK;
}

int hpe_debug_set_loss_counter(hpe_ctx* c, void* counter_dev) {
    if (!c) return fail(HPE_ERR_INVALID, "null ctx");
    c->loss_counter = static_cast<unsigned long long*>(counter_dev);
    return HPE_OK;
}

int hpe_get_conv_timings(hpe_ctx* c, float* ms) {
    if (!c || !ms) return fail(HPE_ERR_INVALID, "null argument");
    if (!c->conv_timed_valid) return fail(HPE_ERR_STATE, "no level-2 timed call recorded");
    DeviceGuard g(c->cfg.device);
    HIP_TRY(hipEventSynchronize(c->ev[4]));
    for (int i = 0; i < HPE_NUM_CONV; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], c->cev0[i], c->cev1[i]));
    return HPE_OK;
}

}  // extern "C"
#pragma GCC visibility pop
