// conv_wino4.hip -- 3x3 / stride 1 / SAME convolutions (res*_branch2b; reference: src/models.py:39 -> keras_applications resnet50
// conv_block / identity_block) as Winograd F(4x4, 3x3) in fp32 (round 3):
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 output tile (6x6 input tile), summed over input channels
//
// 36 multiplies per 16 outputs and channel pair: 4x fewer than the direct convolution, 1.78x fewer than the F(2x2, 3x3) of
// conv_wino.hip (on the 7x7 maps both cover 8x8, on the 14x14 maps this one covers 16x16: 1.36x there).  The transformed input
// is also SMALLER than F(2x2)'s (36 components per 16 pixels = 2.25 floats per input float instead of 4), so the blocked-V round
// trip through HBM that F(2x2) could only afford on the small maps is the cheap part here.  Arithmetic is fp32 end to end, U in
// double on the host.  Error against the fp64 convolution: max 1e-5 of the layer's largest output (F(2x2) / direct: 1.5e-6); it
// does not accumulate through the network -- features, vertices and keypoints stay where they were (tests).
//
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]           (Lavin & Gray, interpolation points 0, +-1, +-2)
//
//   w4_input_kernel   x [B,H,W,C] NHWC -> V, HBM bound.  V is stored in the byte image of the GEMM's LDS staging:
//                     V[tile block of 32][slab of 4 ch][comp 36][tile 32][4 ch]  (a k-slab of a workgroup = 18 KB contiguous)
//   w4_gemm_kernel    36 independent GEMMs M_c[tile][cout] = sum_ch V_c[tile][ch] * U_c[cout][ch] for a 32-tile x 64-cout block, all
//                     36 components in one workgroup of 12 waves: wave (xi, nb) owns row xi of the 6x6 component grid for cout
//                     half nb = six 32x32 MFMA blocks = 96 accumulator VGPRs.  LDS-DMA double buffering of 54-KB slabs (V 18 KB +
//                     U 36 KB), two slabs in flight with the barrier in the middle of a slab's MFMA work (as wino_gemm_kernel).
//                     Output transform: the wave applies (.) A to its own row in registers (6 -> 4 values), the rows meet in LDS
//                     for A^T (.), then BN scale/shift, ReLU and 16-B NHWC stores.
#include <stdlib.h>

#include "hpe_internal.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int W4_T = 32;                      // tiles per workgroup
constexpr int W4_N = 64;                      // output channels per workgroup
constexpr int W4_VS = 36 * W4_T * 4;          // floats of a V slab (4 channels): 4608 = 18 KB
constexpr int W4_US = 36 * W4_N * 4;          // floats of a U slab: 9216 = 36 KB
constexpr int W4_SLAB = W4_VS + W4_US;        // 13824 floats = 54 KB
constexpr int W4_LDS_BYTES = 2 * W4_SLAB * (int)sizeof(float);  // 108 KB (the epilogue's 96-KB exchange area reuses it)
constexpr int W4_THREADS = 768;
constexpr int W4_DMA = W4_SLAB / 256;         // 54 wave-instructions of 1 KB per slab

// row transform of B^T (used for rows, then for columns)
#define W4_BT(o0, o1, o2, o3, o4, o5, d0, d1, d2, d3, d4, d5) \
    do {                                                       \
        o0 = 4.f * (d0) - 5.f * (d2) + (d4);                   \
        o1 = (d3) + (d4) - 4.f * ((d1) + (d2));                \
        o2 = 4.f * ((d1) - (d2)) - (d3) + (d4);                \
        o3 = 2.f * ((d3) - (d1)) - (d2) + (d4);                \
        o4 = 2.f * ((d1) - (d3)) - (d2) + (d4);                \
        o5 = 4.f * (d1) - 5.f * (d3) + (d5);                   \
    } while (0)

// ------------------------------------------------------------------------------------------------ input transform
// one thread = (tile, 4 channels); a wave = 8 tiles x 32 channels: full 128-B lines on the read side, 128-B segments on the write side
__global__ __launch_bounds__(256) void w4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C, int TW, int TT,
                                                       int T, int Tpad, int lda) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int lane = gid & 63;
    const int wv = gid >> 6;
    const int ncb = C >> 5;
    const int tg = wv / ncb;
    const int cb = wv - tg * ncb;
    const int t = tg * 8 + (lane >> 3);
    if (t >= Tpad) return;
    const int c = cb * 32 + (lane & 7) * 4;
    const int S = C >> 2;
    float* dst = V + ((size_t)(t >> 5) * S + (c >> 2)) * W4_VS + (t & 31) * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (t >= T) {
#pragma unroll
        for (int k = 0; k < 36; ++k) *reinterpret_cast<f32x4*>(dst + k * (W4_T * 4)) = z;
        return;
    }
    const int b = t / TT;
    const int rem = t - b * TT;
    const int ty = rem / TW;
    const int tx = rem - ty * TW;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    f32x4 r[6][6];  // r[xi][e] = (B^T d)[xi][e]
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int ix = x0 + e;
        const bool okx = ix >= 0 && ix < W;
        f32x4 d[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int iy = y0 + a;
            const bool ok = okx && iy >= 0 && iy < H;
            d[a] = ok ? *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + iy) * W + ix) * lda + c) : z;
        }
        W4_BT(r[0][e], r[1][e], r[2][e], r[3][e], r[4][e], r[5][e], d[0], d[1], d[2], d[3], d[4], d[5]);
    }
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
        f32x4 o[6];
        W4_BT(o[0], o[1], o[2], o[3], o[4], o[5], r[xi][0], r[xi][1], r[xi][2], r[xi][3], r[xi][4], r[xi][5]);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<f32x4*>(dst + (xi * 6 + nu) * (W4_T * 4)) = o[nu];
    }
}

// ------------------------------------------------------------------------------------------------ 36-component GEMM + output transform
struct W4Args {
    const float* V;
    const float* U;
    const float* scale;
    const float* shift;
    float* y;
    int H, W, N;     // output map, output channels
    int TW, TT, T;   // tiles per row / per image / total
    int S;           // k-slabs (C / 4)
    int n_tb, n_nt;  // tile blocks, cout blocks
    int ldy, relu;
    // w4_gemm32_kernel only: the C axis cut into ksplit parts (1 = whole), one workgroup each; partial outputs meet in split_ws
    // ([output block][part][W4_SPLIT_BLOCK floats]), the part that finishes last (split_cnt, self-resetting) adds them in part order
    int ksplit;
    float* split_ws;
    unsigned* split_cnt;
};

// output transform of one workgroup's 32-tile x 64-cout block: (.) A in registers, A^T (.) across the six waves of a cout half through
// LDS (96 KB exchange area at lds[0]), BN scale/shift, ReLU, NHWC stores.  Entered with the LDS free (all waves past their last slab).
__device__ __forceinline__ void w4_epilogue(const W4Args& p, float* lds, f32x16 (&acc)[6], int tb, int nt, int t, int lane, int xi, int nb) {
    const int hi = lane >> 5;
    // ---- output transform.  (.) A in registers: P[j] = sum_nu A^T[j][nu] M[xi][nu]
    f32x16 P[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float a0 = acc[0][e], a1 = acc[1][e], a2 = acc[2][e], a3 = acc[3][e], a4 = acc[4][e], a5 = acc[5][e];
        const float s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
        P[0][e] = a0 + s12 + s34;
        P[1][e] = d12 + 2.f * d34;
        P[2][e] = s12 + 4.f * s34;
        P[3][e] = d12 + 8.f * d34 + a5;
    }
    // A^T (.) across the six waves of a cout half, through LDS, two output columns per pass:
    //   exchange area [xi 6][jj 2][tile 32][cout 64] floats = 96 KB
    const int em = t >> 4;        // gather role (threads 0 .. 511): tile within the block
    const int eq = (t & 15) * 4;  // cout quad
    const int n = nt * W4_N + eq;
    const bool gather = t < 512;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    int b = 0, ty = 0, tx = 0;
    bool live = false;
    if (gather) {
        sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const int tg = tb * W4_T + em;
        live = tg < p.T;
        if (live) {
            b = tg / p.TT;
            const int rem = tg - b * p.TT;
            ty = rem / p.TW;
            tx = rem - ty * p.TW;
        }
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if (jp) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
                lds[((xi * 2 + jj) * W4_T + m) * W4_N + 32 * nb + (lane & 31)] = P[2 * jp + jj][e];
            }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 Q[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) Q[k] = *reinterpret_cast<const f32x4*>(&lds[((k * 2 + jj) * W4_T + em) * W4_N + eq]);
                const f32x4 s12 = Q[1] + Q[2], d12 = Q[1] - Q[2], s34 = Q[3] + Q[4], d34 = Q[3] - Q[4];
                f32x4 o[4];
                o[0] = Q[0] + s12 + s34;
                o[1] = d12 + 2.f * d34;
                o[2] = s12 + 4.f * s34;
                o[3] = d12 + 8.f * d34 + Q[5];
                const int ox = 4 * tx + 2 * jp + jj;
                if (ox < p.W) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int oy = 4 * ty + i;
                        if (oy < p.H) {
                            f32x4 v = o[i] * sc + sh;
                            if (p.relu) {
                                v.x = fmaxf(v.x, 0.f);
                                v.y = fmaxf(v.y, 0.f);
                                v.z = fmaxf(v.z, 0.f);
                                v.w = fmaxf(v.w, 0.f);
                            }
                            *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * p.ldy + n) = v;
                        }
                    }
                }
            }
        }
    }
}

// ABL (diagnostics builds, -DHPE_ABLATION + HPE_W4_ABL; results wrong): 1 = no V DMA, 2 = no U DMA, 4 = no MFMAs, 8 = no fragment reads
template <int ABL>
__global__ __launch_bounds__(W4_THREADS, 1) void w4_gemm_kernel(W4Args p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 slabs; reused by the epilogue

    const int total = p.n_tb * p.n_nt;
    const int bid = blockIdx.x;
    // XCD-aware bijective remap: consecutive logical ids (the cout blocks of one tile block) share an XCD and its L2
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tb = lid / p.n_nt;
    const int nt = lid - tb * p.n_nt;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int xi = wave >> 1;  // component row owned by this wave
    const int nb = wave & 1;   // cout half (32 couts)
    const int hi = lane >> 5;

    const float* vsrc = p.V + (size_t)tb * p.S * W4_VS + lane * 4;
    const float* usrc = p.U + (size_t)nt * p.S * W4_US + lane * 4;

    // slab s -> buffer buf: 54 wave-instructions of 1 KB, instruction k = wave + 12 i (V for k < 18, U after)
    auto issue = [&](int s, int buf) {
        const float* sv = vsrc + (size_t)s * W4_VS;
        const float* su = usrc + (size_t)s * W4_US;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = wave + 12 * i;
            if (k < W4_DMA && !((ABL & 1) && k < 18) && !((ABL & 2) && k >= 18)) {
                const float* src = (k < 18) ? sv + k * 256 : su + (k - 18) * 256;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + buf * W4_SLAB + k * 256), 16, 0, 0);
            }
        }
    };

    f32x16 acc[6];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nu][e] = 0.f;

    // fragment offsets (floats): V [comp][tile 32][4], U [comp][cout 64][4]; lanes 0-31 take channels 0-1, lanes 32-63 channels 2-3
    const int fv = (xi * 6) * (W4_T * 4) + (lane & 31) * 4 + 2 * hi;
    const int fu = W4_VS + (xi * 6) * (W4_N * 4) + (32 * nb + (lane & 31)) * 4 + 2 * hi;
    const int S = p.S;

    // Two slabs in flight.  Per slab s: [fragment reads, MFMAs of components 0-2] [barrier] [DMA of slab s+2 into the buffer slab s has
    // just vacated] [MFMAs of components 3-5]: the barrier comes after this wave has pulled ALL its fragments of slab s into registers,
    // so the buffer is free for slab s+2, and that DMA has a whole slab of MFMA time to land before barrier s+1 waits for it.
    issue(0, 0);
    if (S > 1) issue(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (once per workgroup: not worth a per-wave counted wait)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int s = 0; s < S; ++s) {
        const int cur = (s & 1) * W4_SLAB;
        f32x2 fa[6], fb[6];
        if (!(ABL & 8)) {
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                fa[nu] = *reinterpret_cast<const f32x2*>(&lds[cur + fv + nu * (W4_T * 4)]);
                fb[nu] = *reinterpret_cast<const f32x2*>(&lds[cur + fu + nu * (W4_N * 4)]);
            }
        } else {
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) fa[nu] = fb[nu] = f32x2{1.f, 1.f};
        }
        if (!(ABL & 4)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nu][ks], fb[nu][ks], acc[nu], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // lgkmcnt(0): my fragments of slab s are in registers; vmcnt(0): my part of slab s+1 has landed.  Written out: with the
        // wave-uniform branches of issue() in the loop hipcc's __syncthreads() emitted only the lgkmcnt wait here (found as whole
        // tile blocks that came out wrong, now and then, once >= 200 workgroups stretched the DMA latency)
        __builtin_amdgcn_s_waitcnt(0x0070);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 2 < S) issue(s + 2, s & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nu = 3; nu < 6; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nu][ks], fb[nu][ks], acc[nu], 0, 0, 0);
        } else {
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) acc[nu][0] += fa[nu][0] * fb[nu][1];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();  // all waves out of the last slab before the epilogue reuses the LDS

    w4_epilogue(p, lds, acc, tb, nt, t, lane, xi, nb);
}

// ---- 32-cout variant for launches that would leave CUs empty (batches around 64 images; the 7x7 layers of a 128-image chunk):
// a workgroup = 32 tiles x 32 couts, SIX waves (wave xi = row xi of the component grid, the same six 32x32 accumulator blocks), slabs of
// V 18 KB + U 18 KB double buffered = 72 KB, so TWO workgroups share a CU: twice the workgroups for the same launch, the same 12 waves per
// CU when they are all there.  Reads the same blocked U (a 32-cout half of a 64-cout block: per-lane source addresses, 512 B per component).
constexpr int W4_US32 = 36 * 32 * 4;                 // 4608 floats
constexpr int W4_SLAB32 = W4_VS + W4_US32;           // 9216 floats = 36 KB
constexpr int W4_LDS32_BYTES = 2 * W4_SLAB32 * (int)sizeof(float);  // 72 KB
constexpr int W4_THREADS32 = 384;
constexpr int W4_SPLIT_BLOCK = 32 * 16 * 32;  // floats of one workgroup's output block (32 tiles x 16 pixels x 32 couts)
constexpr int W4_SPLIT_SLOTS = 256;           // parked blocks the workspace holds (16 MB) = workgroups of a split launch

__global__ __launch_bounds__(W4_THREADS32, 2) void w4_gemm32_kernel(W4Args p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int n_nt = 2 * p.n_nt;  // 32-cout blocks
    const int total = p.n_tb * n_nt;
    const int KS = p.ksplit;
    const int bid = KS > 1 ? (int)blockIdx.x / KS : (int)blockIdx.x;
    const int part = KS > 1 ? (int)blockIdx.x - bid * KS : 0;
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tb = lid / n_nt;
    const int nt = lid - tb * n_nt;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int xi = __builtin_amdgcn_readfirstlane(t >> 6);  // wave = component row
    const int hi = lane >> 5;
    const int S = p.S;
    const int s_begin = (int)((long)part * S / KS), s_end = (int)((long)(part + 1) * S / KS);  // this workgroup's slabs

    const float* vsrc = p.V + (size_t)tb * S * W4_VS + lane * 4;
    // U instruction k (0..17) moves components 2k, 2k+1: lane -> (component 2k + (lane >> 5), cout 32 (nt & 1) + (lane & 31))
    const float* usrc = p.U + (size_t)(nt >> 1) * S * W4_US + (lane >> 5) * (W4_N * 4) + (32 * (nt & 1) + (lane & 31)) * 4;

    auto issue = [&](int s, int buf) {
        const float* sv = vsrc + (size_t)s * W4_VS;
        const float* su = usrc + (size_t)s * W4_US;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int k = xi + 6 * i;  // 36 wave-instructions, 6 per wave
            const float* src = (k < 18) ? sv + k * 256 : su + (k - 18) * (2 * W4_N * 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf * W4_SLAB32 + k * 256), 16, 0, 0);
        }
    };

    f32x16 acc[6];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nu][e] = 0.f;
    const int fv = (xi * 6) * (W4_T * 4) + (lane & 31) * 4 + 2 * hi;
    const int fu = W4_VS + (xi * 6) * (32 * 4) + (lane & 31) * 4 + 2 * hi;

    issue(s_begin, 0);
    if (s_begin + 1 < s_end) issue(s_begin + 1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int s = s_begin; s < s_end; ++s) {
        const int cur = ((s - s_begin) & 1) * W4_SLAB32;
        f32x2 fa[6], fb[6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) {
            fa[nu] = *reinterpret_cast<const f32x2*>(&lds[cur + fv + nu * (W4_T * 4)]);
            fb[nu] = *reinterpret_cast<const f32x2*>(&lds[cur + fu + nu * (32 * 4)]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nu][ks], fb[nu][ks], acc[nu], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), written out (see w4_gemm_kernel)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 2 < s_end) issue(s + 2, (s - s_begin) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nu = 3; nu < 6; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nu][ks], fb[nu][ks], acc[nu], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    // ---- output transform as w4_epilogue, 32 couts wide: exchange area [xi 6][jj 2][tile 32][cout 32] = 48 KB
    f32x16 P[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float a0 = acc[0][e], a1 = acc[1][e], a2 = acc[2][e], a3 = acc[3][e], a4 = acc[4][e], a5 = acc[5][e];
        const float s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
        P[0][e] = a0 + s12 + s34;
        P[1][e] = d12 + 2.f * d34;
        P[2][e] = s12 + 4.f * s34;
        P[3][e] = d12 + 8.f * d34 + a5;
    }
    const int em = t >> 3;       // gather role (threads 0 .. 255): tile within the block
    const int eq = (t & 7) * 4;  // cout quad
    const int n = nt * 32 + eq;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    int b = 0, ty = 0, tx = 0;
    bool live = false;
    if (t < 256) {
        sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const int tg = tb * W4_T + em;
        live = tg < p.T;
        if (live) {
            b = tg / p.TT;
            const int rem = tg - b * p.TT;
            ty = rem / p.TW;
            tx = rem - ty * p.TW;
        }
    }
    auto pack2 = [](float a, float b) { return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32); };
    unsigned long long* ws_block = reinterpret_cast<unsigned long long*>(p.split_ws) + (size_t)lid * KS * (W4_SPLIT_BLOCK / 2) + (t & 255);
    unsigned long long* ws_mine = ws_block + (size_t)part * (W4_SPLIT_BLOCK / 2);
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if (jp) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
                lds[((xi * 2 + jj) * W4_T + m) * 32 + (lane & 31)] = P[2 * jp + jj][e];
            }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 Q[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) Q[k] = *reinterpret_cast<const f32x4*>(&lds[((k * 2 + jj) * W4_T + em) * 32 + eq]);
                const f32x4 s12 = Q[1] + Q[2], d12 = Q[1] - Q[2], s34 = Q[3] + Q[4], d34 = Q[3] - Q[4];
                f32x4 o[4];
                o[0] = Q[0] + s12 + s34;
                o[1] = d12 + 2.f * d34;
                o[2] = s12 + 4.f * s34;
                o[3] = d12 + 8.f * d34 + Q[5];
                const int ox = 4 * tx + 2 * jp + jj;
                if (KS > 1) {
                    // a part of the C axis: park the partial outputs (relaxed agent-scope atomics = sc1 stores, see conv_wino.hip)
                    unsigned long long* w = ws_mine + (size_t)((jp * 2 + jj) * 4) * 2 * 256;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __hip_atomic_store(w + (2 * i) * 256, pack2(o[i].x, o[i].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(w + (2 * i + 1) * 256, pack2(o[i].z, o[i].w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else if (ox < p.W) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int oy = 4 * ty + i;
                        if (oy < p.H) {
                            f32x4 v = o[i] * sc + sh;
                            if (p.relu) {
                                v.x = fmaxf(v.x, 0.f);
                                v.y = fmaxf(v.y, 0.f);
                                v.z = fmaxf(v.z, 0.f);
                                v.w = fmaxf(v.w, 0.f);
                            }
                            *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * p.ldy + n) = v;
                        }
                    }
                }
            }
        }
    }
    if (KS > 1) {
        // Hand-off form (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", first row of the table of
        // hand-offs measured with sc1 loads in place of the acquire): EVERY parked byte is stored sc1 (relaxed agent-scope atomic store =
        // write-through, 8 B), every storing wave drains vmcnt, the workgroup meets at a barrier, ONE lane adds to the block's counter at
        // agent scope, the workgroup whose add returns KS - 1 is last; its other waves read only after the barrier that lane then joins,
        // and EVERY read of the parked bytes is an sc1 load (relaxed agent-scope atomic load, bypasses the reader's L1).  No fence: an
        // agent-scope release / acquire pair writes back / invalidates the XCD's whole L2 (measured: 30 % slower per layer).
        // The parts of one block run on different XCDs; the form above is what makes that safe, not the memory model's fences.
        // every wave's parked stores must be acknowledged before the counter moves (written out: hipcc emits no vmcnt wait for a
        // workgroup-scope fence / __syncthreads() here)
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        int* flag = reinterpret_cast<int*>(lds);
        if (t == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.split_cnt + lid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(KS - 1);
            if (last) __hip_atomic_store(p.split_cnt + lid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch / graph replay
            *flag = last;
        }
        __syncthreads();
        if (!*flag || !live) return;
        // last part of this output block: add the parts in part order (bitwise the same result whichever part is last), BN, ReLU, store
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int ox = 4 * tx + 2 * jp + jj;
                if (ox >= p.W) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int oy = 4 * ty + i;
                    if (oy >= p.H) continue;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    for (int k = 0; k < KS; ++k) {
                        const unsigned long long* w = ws_block + (size_t)k * (W4_SPLIT_BLOCK / 2) + (size_t)(((jp * 2 + jj) * 4 + i) * 2) * 256;
                        const unsigned long long lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long hi2 = __hip_atomic_load(w + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a.x += __uint_as_float((unsigned)lo);
                        a.y += __uint_as_float((unsigned)(lo >> 32));
                        a.z += __uint_as_float((unsigned)hi2);
                        a.w += __uint_as_float((unsigned)(hi2 >> 32));
                    }
                    f32x4 v = a * sc + sh;
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * p.ldy + n) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ fused input transform (56x56 / 28x28 maps)
// The same GEMM without V in memory: the producing 1x1 convolution writes its output channel-slab major, Xs[slab of 8 ch][pixel][8]
// (GemmArgs::y_slab8, as for wino_fused_kernel), and a workgroup owns R consecutive tile rows of the batch -- 2 x 14 tiles on the 56x56
// maps, 4 x 7 on the 28x28 maps: 28 of its 32 tile slots -- for one 64-cout block.  Per 4-channel slab (half of an 8-channel record: 16 B
// per pixel) the 6 input rows of every tile row arrive by LDS-DMA as raw[tile row][6][W + 2][4 ch] (zero page for the halo; <= 720 chunks =
// ONE wave-instruction per wave), and 168 (tile, xi) pairs, 14 lanes of every wave, turn them into the V slab image the MFMA loop reads:
// row xi of B^T d for the six columns, then (.) B.  The MFMA work of slab s overlaps the transform of slab s+1 and the DMAs of
// U(s+1) and raw(s+2); one barrier per slab.  LDS: V 2 x 18 KB + U 2 x 36 KB + raw 2 x 12 KB = 132 KB.
struct W4FusedArgs {
    const float* Xs;     // [C/8][M][8]
    const float* U;
    const float* scale;
    const float* shift;
    const float* zero;   // >= 16 B of zeros
    float* y;
    int H, W, TW, TH;    // map, tiles per row / per column (W / 4, H / 4)
    int R, NG;           // tile rows per workgroup, tile rows in the batch (B * TH)
    int n_blk, n_nt, S;  // workgroups along the tile rows, cout blocks, 4-channel slabs (C / 4)
    int NC;              // raw chunks per slab (R * 6 * (W + 2))
    long M;              // pixels in the batch (B * H * W)
    int ldy, relu;
};

constexpr int W4F_RAW = 768 * 4;  // floats per raw buffer (768 chunks >= 2*6*58 = 696 and 4*6*30 = 720)
constexpr int W4F_BUF = W4_VS + W4_US + W4F_RAW;  // 16896 floats
constexpr int W4F_LDS_BYTES = 2 * W4F_BUF * (int)sizeof(float);  // 132 KB

__constant__ float w4_bt[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};

// MAPW: the map side (56 or 28), a template parameter so that the 36 raw reads of the transform are ONE address register plus immediate
// offsets (with a run-time row pitch hipcc keeps 36 address VGPRs alive and spills the accumulators)
template <int MAPW>
__global__ __launch_bounds__(W4_THREADS, 1) void w4_fused_kernel(W4FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [buf 2][V | U | raw]

    const int total = p.n_blk * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int blk = lid / p.n_nt;
    const int nt = lid - blk * p.n_nt;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int xi = wave >> 1;
    const int nb = wave & 1;
    const int hi = lane >> 5;
    const int S = p.S;
    constexpr int Wp = MAPW + 2;
    constexpr int TW = MAPW / 4;

    // ---- raw DMA source of this lane: chunk c = wave * 64 + lane -> (tile row, input row a, padded column xp)
    long rsrc = -1;  // float offset inside an 8-channel slab of Xs (without the half), or -1 for the zero page
    {
        const int c = wave * 64 + lane;
        if (c < p.NC) {
            const int xp = c % Wp;
            const int ra = c / Wp;
            const int a = ra % 6;
            const int trl = ra / 6;
            const int g = blk * p.R + trl;
            if (g < p.NG) {
                const int b = g / p.TH;
                const int ty = g - b * p.TH;
                const int iy = 4 * ty - 1 + a, ix = xp - 1;
                if (iy >= 0 && iy < MAPW && ix >= 0 && ix < MAPW) rsrc = ((long)(b * MAPW + iy) * MAPW + ix) * 8;
            }
        }
    }
    const bool raw_wave = wave * 64 < p.NC;
    const float* usrc = p.U + (size_t)nt * S * W4_US + lane * 4;

    auto issue_raw = [&](int s, int buf) {
        if (!raw_wave) return;
        const float* src = rsrc >= 0 ? p.Xs + (size_t)(s >> 1) * p.M * 8 + rsrc + (s & 1) * 4 : p.zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + buf * W4F_BUF + W4_VS + W4_US + wave * 256), 16, 0, 0);
    };
    auto issue_u = [&](int s, int buf) {
        const float* su = usrc + (size_t)s * W4_US;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = wave + 12 * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(su + k * 256),
                                             (__attribute__((address_space(3))) void*)(lds + buf * W4F_BUF + W4_VS + k * 256), 16, 0, 0);
        }
    };

    // ---- transform role: wave (xi, nb) computes row xi of B^T d -- the same row it multiplies, so the six coefficients are
    //      wave-uniform (SGPRs) -- for tiles 14 nb .. 14 nb + 13, two channels per lane: lanes 0 .. 27 = (tile, channel pair)
    const int n_tiles = p.R * TW;
    const int t_tile = 14 * nb + (lane >> 1);
    const int chp = lane & 1;
    const bool t_live = lane < 28 && t_tile < n_tiles;
    const int t_trl = t_live ? t_tile / TW : 0;
    const int t_tx = t_live ? t_tile - t_trl * TW : 0;
    float cf[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) cf[a] = w4_bt[xi][a];
    const int raw0 = ((t_trl * 6) * Wp + 4 * t_tx) * 4 + 2 * chp;  // float offset of input row 0, tile column 0, this lane's channel pair
    auto transform = [&](int rbuf, int vbuf) {
        if (!t_live) return;
        const float* rw = lds + rbuf * W4F_BUF + W4_VS + W4_US + raw0;
        f32x2 r[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            f32x2 d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) d[a] = *reinterpret_cast<const f32x2*>(rw + (a * Wp + e) * 4);
            r[e] = cf[0] * d[0] + cf[1] * d[1] + cf[2] * d[2] + cf[3] * d[3] + cf[4] * d[4] + cf[5] * d[5];
        }
        f32x2 o[6];
        W4_BT(o[0], o[1], o[2], o[3], o[4], o[5], r[0], r[1], r[2], r[3], r[4], r[5]);
        float* v = lds + vbuf * W4F_BUF + (xi * 6) * (W4_T * 4) + t_tile * 4 + 2 * chp;
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<f32x2*>(v + nu * (W4_T * 4)) = o[nu];
    };

    f32x16 acc[6];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nu][e] = 0.f;
    const int fv = (xi * 6) * (W4_T * 4) + (lane & 31) * 4 + 2 * hi;
    const int fu = W4_VS + (xi * 6) * (W4_N * 4) + (32 * nb + (lane & 31)) * 4 + 2 * hi;

    issue_raw(0, 0);
    issue_u(0, 0);
    if (S > 1) issue_raw(1, 1);
    // tile slots that belong to no tile are never written by the transform: clear them once in both V buffers
    if (n_tiles < W4_T) {
        const int dead = W4_T - n_tiles;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = t; i < 2 * 36 * dead; i += W4_THREADS) {
            const int row = n_tiles + i % dead;
            const int comp = (i / dead) % 36;
            const int buf = i / (dead * 36);
            *reinterpret_cast<f32x4*>(lds + buf * W4F_BUF + (comp * W4_T + row) * 4) = z;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    transform(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int s = 0; s < S; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < S) issue_u(s + 1, nxt);
        if (s + 2 < S) issue_raw(s + 2, cur);
        const int cb = cur * W4F_BUF;
        // first-half fragments and MFMAs, then the transform of the next slab while they run, then the second half: the transform's
        // ~50 VGPRs and a half's 12 fragment VGPRs are never live together with the other half's (96 accumulators + 168-VGPR budget)
        {
            f32x2 fa[3], fb[3];
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) {
                fa[nu] = *reinterpret_cast<const f32x2*>(&lds[cb + fv + nu * (W4_T * 4)]);
                fb[nu] = *reinterpret_cast<const f32x2*>(&lds[cb + fu + nu * (W4_N * 4)]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nu = 0; nu < 3; ++nu) acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nu][ks], fb[nu][ks], acc[nu], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < S) transform(nxt, nxt);  // raw(s+1) landed behind the barrier that ended slab s-1
        __builtin_amdgcn_sched_barrier(0);
        {
            f32x2 fa[3], fb[3];
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) {
                fa[nu] = *reinterpret_cast<const f32x2*>(&lds[cb + fv + (nu + 3) * (W4_T * 4)]);
                fb[nu] = *reinterpret_cast<const f32x2*>(&lds[cb + fu + (nu + 3) * (W4_N * 4)]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nu = 0; nu < 3; ++nu) acc[nu + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nu][ks], fb[nu][ks], acc[nu + 3], 0, 0, 0);
        }
        // U(s+1) and raw(s+2) have landed, V(s+1) is written, every wave is done with V(s) / U(s)
        __builtin_amdgcn_s_waitcnt(0x0070);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- output transform (w4_epilogue with this kernel's tile -> pixel mapping)
    f32x16 P[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float a0 = acc[0][e], a1 = acc[1][e], a2 = acc[2][e], a3 = acc[3][e], a4 = acc[4][e], a5 = acc[5][e];
        const float s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
        P[0][e] = a0 + s12 + s34;
        P[1][e] = d12 + 2.f * d34;
        P[2][e] = s12 + 4.f * s34;
        P[3][e] = d12 + 8.f * d34 + a5;
    }
    const int em = t >> 4;
    const int eq = (t & 15) * 4;
    const int n = nt * W4_N + eq;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    int b = 0, ty = 0, tx = 0;
    bool live = false;
    if (t < 512) {
        sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const int trl = em / TW;
        tx = em - trl * TW;
        const int g = blk * p.R + trl;
        live = em < n_tiles && g < p.NG;
        if (live) {
            b = g / p.TH;
            ty = g - b * p.TH;
        }
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if (jp) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
                lds[((xi * 2 + jj) * W4_T + m) * W4_N + 32 * nb + (lane & 31)] = P[2 * jp + jj][e];
            }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 Q[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) Q[k] = *reinterpret_cast<const f32x4*>(&lds[((k * 2 + jj) * W4_T + em) * W4_N + eq]);
                const f32x4 s12 = Q[1] + Q[2], d12 = Q[1] - Q[2], s34 = Q[3] + Q[4], d34 = Q[3] - Q[4];
                f32x4 o[4];
                o[0] = Q[0] + s12 + s34;
                o[1] = d12 + 2.f * d34;
                o[2] = s12 + 4.f * s34;
                o[3] = d12 + 8.f * d34 + Q[5];
                const int ox = 4 * tx + 2 * jp + jj;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int oy = 4 * ty + i;
                    f32x4 v = o[i] * sc + sh;
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * p.ldy + n) = v;
                }
            }
        }
    }
}

}  // namespace

// hipFuncSetAttribute applies to the CURRENT device: hpe_finalize calls this once per ctx under its device guard
size_t hpe_wino4_split_ws_floats() { return (size_t)W4_SPLIT_SLOTS * W4_SPLIT_BLOCK + W4_SPLIT_SLOTS; }

hipError_t hpe_wino4_init_device() {
#ifdef HPE_ABLATION
    for (const void* f : {(const void*)w4_gemm_kernel<1>, (const void*)w4_gemm_kernel<2>, (const void*)w4_gemm_kernel<3>, (const void*)w4_gemm_kernel<4>,
                          (const void*)w4_gemm_kernel<8>, (const void*)w4_gemm_kernel<12>})
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES);
#endif
    hipError_t e32 = hipFuncSetAttribute(reinterpret_cast<const void*>(w4_gemm32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS32_BYTES);
    if (e32 != hipSuccess) return e32;
    hipError_t ef = hipFuncSetAttribute(reinterpret_cast<const void*>(w4_fused_kernel<56>), hipFuncAttributeMaxDynamicSharedMemorySize, W4F_LDS_BYTES);
    if (ef != hipSuccess) return ef;
    ef = hipFuncSetAttribute(reinterpret_cast<const void*>(w4_fused_kernel<28>), hipFuncAttributeMaxDynamicSharedMemorySize, W4F_LDS_BYTES);
    if (ef != hipSuccess) return ef;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(w4_gemm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES);
}

// fused-transform variant: maps with H, W multiples of 4 and W / 4 <= 32 tiles per row; xs channel-slab major [C/8][B*H*W][8]
static bool w4_fused_geometry(int B, int H, int W, int* R_out, int* NC_out) {
    if (H != W || (W != 56 && W != 28)) return false;  // the two instantiations of w4_fused_kernel
    const int TW = W / 4;
    if (TW > W4_T) return false;
    int R = W4_T / TW;
    while (R > 1 && R * 6 * (W + 2) > 768) --R;
    if (R * 6 * (W + 2) > 768) return false;
    if (R > (H / 4) * B) R = (H / 4) * B;
    *R_out = R;
    *NC_out = R * 6 * (W + 2);
    return R >= 1;
}

int hpe_wino4_fused_items(int B, int H, int W, int N) {
    int R, NC;
    if (!w4_fused_geometry(B, H, W, &R, &NC)) return 0;
    return ((B * (H / 4) + R - 1) / R) * (N / W4_N);
}

hipError_t hpe_launch_wino4_fused_conv3(const float* xs, const float* U, const float* scale, const float* shift, const float* zero16, float* y,
                                        int ldy, int B, int H, int W, int C, int N, int relu, hipStream_t st) {
    int R, NC;
    if (C % 8 != 0 || N % 64 != 0 || ldy % 4 != 0 || B < 1 || !xs || !U || !y || !zero16 || !w4_fused_geometry(B, H, W, &R, &NC)) return hipErrorInvalidValue;
    W4FusedArgs p{};
    p.Xs = xs;
    p.U = U;
    p.scale = scale;
    p.shift = shift;
    p.zero = zero16;
    p.y = y;
    p.H = H;
    p.W = W;
    p.TW = W / 4;
    p.TH = H / 4;
    p.R = R;
    p.NG = B * p.TH;
    p.n_blk = (p.NG + R - 1) / R;
    p.n_nt = N / W4_N;
    p.S = C / 4;
    p.NC = NC;
    p.M = (long)B * H * W;
    p.ldy = ldy;
    p.relu = relu;
    if (W == 56)
        hipLaunchKernelGGL(w4_fused_kernel<56>, dim3(p.n_blk * p.n_nt), dim3(W4_THREADS), W4F_LDS_BYTES, st, p);
    else
        hipLaunchKernelGGL(w4_fused_kernel<28>, dim3(p.n_blk * p.n_nt), dim3(W4_THREADS), W4F_LDS_BYTES, st, p);
    return hipGetLastError();
}

size_t hpe_wino4_v_floats(int B, int H, int W, int C) {
    const int TH = (H + 3) / 4, TW = (W + 3) / 4;
    const size_t T = (size_t)B * TH * TW;
    return ((T + W4_T - 1) / W4_T) * W4_T * 36 * (size_t)C;
}

int hpe_wino4_items(int B, int H, int W, int N) {
    // workgroups of the launch: 32-tile x 32-cout ones (the small-launch variant) -- what the dispatch threshold counts
    const int TH = (H + 3) / 4, TW = (W + 3) / 4;
    return (int)(((long)B * TH * TW + W4_T - 1) / W4_T) * (N / 32);
}

hipError_t hpe_launch_wino4_conv3(const float* x, int lda, const float* U, const float* scale, const float* shift, float* y, int ldy, int B,
                                  int H, int W, int C, int N, int relu, float* V, hipStream_t st, int co_running, float* split_ws) {
    if (C % 32 != 0 || N % 64 != 0 || lda % 4 != 0 || ldy % 4 != 0 || B < 1 || H < 1 || W < 1 || !x || !U || !V || !y) return hipErrorInvalidValue;
    const int TH = (H + 3) / 4, TW = (W + 3) / 4, TT = TH * TW;
    const long Tl = (long)B * TT;
    if (Tl > (1L << 30)) return hipErrorInvalidValue;
    const int T = (int)Tl;
    const int Tpad = (T + W4_T - 1) / W4_T * W4_T;
    {
        const long waves = (long)(Tpad / 8) * (C / 32);
        const int blocks = (int)((waves + 3) / 4);
        hipLaunchKernelGGL(w4_input_kernel, dim3(blocks), dim3(256), 0, st, x, V, H, W, C, TW, TT, T, Tpad, lda);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    W4Args p{};
    p.V = V;
    p.U = U;
    p.scale = scale;
    p.shift = shift;
    p.y = y;
    p.H = H;
    p.W = W;
    p.N = N;
    p.TW = TW;
    p.TT = TT;
    p.T = T;
    p.S = C / 4;
    p.n_tb = Tpad / W4_T;
    p.n_nt = N / W4_N;
    p.ldy = ldy;
    p.relu = relu;
#ifdef HPE_ABLATION
    static const int abl = [] {
        const char* e = getenv("HPE_W4_ABL");
        return e ? atoi(e) : 0;
    }();
    const dim3 g(p.n_tb * p.n_nt), b(W4_THREADS);
    switch (abl) {
        case 1: hipLaunchKernelGGL(w4_gemm_kernel<1>, g, b, W4_LDS_BYTES, st, p); return hipGetLastError();
        case 2: hipLaunchKernelGGL(w4_gemm_kernel<2>, g, b, W4_LDS_BYTES, st, p); return hipGetLastError();
        case 3: hipLaunchKernelGGL(w4_gemm_kernel<3>, g, b, W4_LDS_BYTES, st, p); return hipGetLastError();
        case 4: hipLaunchKernelGGL(w4_gemm_kernel<4>, g, b, W4_LDS_BYTES, st, p); return hipGetLastError();
        case 8: hipLaunchKernelGGL(w4_gemm_kernel<8>, g, b, W4_LDS_BYTES, st, p); return hipGetLastError();
        case 12: hipLaunchKernelGGL(w4_gemm_kernel<12>, g, b, W4_LDS_BYTES, st, p); return hipGetLastError();
        default: break;
    }
#endif
    // Fewer 64-cout workgroups on the device than CUs -- this launch's, times the batch chunks running beside it on other streams -- :
    // the 32-cout variant (twice the workgroups, two per CU).  From 256 = one per CU on the 64-cout kernel is the faster one
    // (profiles/r03/w4_n32_ab.txt).  HPE_WINO4_N32: workgroup count below which the variant is used (0 = never)
    static const int n32_below = [] {
        const char* e = getenv("HPE_WINO4_N32");
        return e ? atoi(e) : 256;
    }();
    p.ksplit = 1;
    if (p.n_tb * p.n_nt * (co_running > 1 ? co_running : 1) < n32_below) {
        // Still fewer workgroups than half the CUs, and a long C axis (the 7x7 layers up to ~64 images: 128 slabs = a 0.14 ms chain of
        // barriers whatever the batch): cut C into 2-4 parts of >= 16 slabs.  Needs a workspace that no concurrent launch uses (the caller
        // passes one per chunk stream; nullptr = plan option wino4_ksplit off).  HPE_WINO4_KSPLIT_WGS: workgroups on the device to aim for.
        static const int ksplit_wgs = [] {
            const char* e = getenv("HPE_WINO4_KSPLIT_WGS");
            return e ? atoi(e) : 512;  // two 6-wave workgroups per CU
        }();
        const int wgs = p.n_tb * p.n_nt * 2 * (co_running > 1 ? co_running : 1);  // on the device, with the co-running chunks' launches
        if (split_ws && 2 * wgs <= ksplit_wgs) {
            int ks = ksplit_wgs / wgs;
            if (ks > 4) ks = 4;
            while (ks > 1 && (p.S / ks < 16 || p.n_tb * p.n_nt * 2 * ks > W4_SPLIT_SLOTS)) --ks;
            if (ks > 1) {
                p.ksplit = ks;
                p.split_ws = split_ws;
                p.split_cnt = reinterpret_cast<unsigned*>(split_ws + (size_t)W4_SPLIT_SLOTS * W4_SPLIT_BLOCK);
            }
        }
        hipLaunchKernelGGL(w4_gemm32_kernel, dim3(p.n_tb * p.n_nt * 2 * p.ksplit), dim3(W4_THREADS32), W4_LDS32_BYTES, st, p);
    } else
        hipLaunchKernelGGL(w4_gemm_kernel<0>, dim3(p.n_tb * p.n_nt), dim3(W4_THREADS), W4_LDS_BYTES, st, p);
    return hipGetLastError();
}
