// conv1x1_stream_bf16.hip -- the expand convolution of an identity block in the bf16 encoder (res*_branch2c: 1x1, C -> 4C,
// + BN + residual add + ReLU; reference: src/models.py:39 -> keras_applications resnet50 identity_block) as a STREAMING
// kernel.  In bf16 these layers are pure data movement (K = C is 64 .. 256: one to four 64-deep k-slabs; the matrix work is
// < 15 % of the time the bytes need), and the generic implicit-GEMM kernel spends them in its LDS round trips: slab DMA,
// barrier, accumulators -> LDS -> rows, barrier.  Here nothing goes through LDS and there is no barrier:
//   * orientation Y^T = W . X^T: output channels are the MFMA's rows, pixels its columns.  A wave owns NB blocks of 32
//     output channels for good -- their weights (BN scale folded in) live in its registers for the whole kernel -- and walks
//     over tiles of 32 pixels;
//   * the activation tile is the B operand straight from global memory: lane (pixel, half h) loads K contiguous bytes of its
//     pixel row (k is only a summation label, the weights are packed on the host in the matching order);
//   * the BN shift enters through one extra MFMA k-step (shift split into three bf16 terms against a constant-one fragment),
//     so no per-channel vectors sit in registers;
//   * the accumulators hold, per lane, 4-channel runs of one pixel; v_permlane32_swap pairs them to 8-channel runs, so the
//     residual arrives and the result leaves as 16 B per lane, NHWC rows, without a transpose (cdna_hip_programming.md T21).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpe_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StreamArgs {
    const __bf16* x;    // [M][K] activations
    const bf16x8* w;    // [N/32][KS+1][64] A fragments in lane order (hpe_pack_stream_weights)
    const __bf16* res;  // [M][N] residual (may be nullptr)
    __bf16* y;          // [M][N]
    int M, N, K, relu;
    int ntiles;         // ceil(M / 32)
    int nsplit;         // workgroups per pixel tile = N / (128 * NB)
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return u;
}

template <int KS, int NB>
__global__ __launch_bounds__(256) void conv1x1_stream_bf16_kernel(StreamArgs p) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int ns = blockIdx.x % p.nsplit;
    const int blk0 = (ns * 4 + wave) * NB;  // first 32-channel block of this wave

    // ---- this wave's weights: NB blocks x (KS + 1) k-steps, 16 B per lane each, resident for the whole kernel
    bf16x8 wr[NB][KS + 1];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s <= KS; ++s) wr[nb][s] = p.w[((size_t)(blk0 + nb) * (KS + 1) + s) * 64 + lane];
    // constant B fragment of the shift k-step: ones in the three slots that carry the shift's bf16 terms (lower half only)
    bf16x8 onesf;
#pragma unroll
    for (int j = 0; j < 8; ++j) onesf[j] = (__bf16)((h == 0 && j < 3) ? 1.0f : 0.0f);

    const int tstride = gridDim.x / p.nsplit;
    for (int tile = blockIdx.x / p.nsplit; tile < p.ntiles; tile += tstride) {
        const int px = tile * 32 + c;
        const bool live = px < p.M;
        const size_t row = (size_t)(live ? px : p.M - 1);
        // ---- activations: k-step s of lane half h = channels (K/2) h + 8 s .. + 7 of the pixel
        const bf16x8* xp = reinterpret_cast<const bf16x8*>(p.x + row * p.K + h * (KS * 8));
        bf16x8 xf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = xp[s];
        // ---- residual in store layout: piece q of block nb = channels 32 (blk0 + nb) + 16 q + 8 h .. + 7
        u32x4 rq[NB][2];
        if (p.res) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    rq[nb][q] = *reinterpret_cast<const u32x4*>(p.res + row * p.N + 32 * (blk0 + nb) + 16 * q + 8 * h);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[nb][KS], onesf, acc, 0, 0, 0);  // acc = BN shift of the row's channel
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[nb][s], xf[s], acc, 0, 0, 0);
            // C layout: column = pixel c, register e = channel (e & 3) + 8 (e >> 2) + 4 h of the block
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // registers 8q .. 8q+3 (channels 16q + 4h + i) and 8q+4 .. 8q+7 (channels 16q + 8 + 4h + i)
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[8 * q + i];
                if (p.res) {
                    // un-swap the residual piece into this layout: (a, b) = the two channel quads of this lane
                    const auto r0 = __builtin_amdgcn_permlane32_swap(rq[nb][q][0], rq[nb][q][2], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(rq[nb][q][1], rq[nb][q][3], false, false);
                    v[0] += bf_lo(r0[0]);
                    v[1] += bf_hi(r0[0]);
                    v[2] += bf_lo(r1[0]);
                    v[3] += bf_hi(r1[0]);
                    v[4] += bf_lo(r0[1]);
                    v[5] += bf_hi(r0[1]);
                    v[6] += bf_lo(r1[1]);
                    v[7] += bf_hi(r1[1]);
                }
                if (p.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                unsigned ax = pack2(v[0], v[1]), ay = pack2(v[2], v[3]);  // quad 2q     : channels 16q + 4h + 0..3
                unsigned bx = pack2(v[4], v[5]), by = pack2(v[6], v[7]);  // quad 2q + 1 : channels 16q + 8 + 4h + 0..3
                const auto s0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                // lanes 0-31: [own quad 2q | upper's quad 2q] = channels 16q .. 16q+7; lanes 32-63: channels 16q+8 .. 16q+15
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (live) *reinterpret_cast<u32x4*>(p.y + row * p.N + 32 * (blk0 + nb) + 16 * q + 8 * h) = o;
            }
        }
    }
}

template <int KS, int NB>
hipError_t launch_stream(StreamArgs& p, hipStream_t st) {
    p.nsplit = p.N / (128 * NB);
    int wgs = p.ntiles * p.nsplit;
    const int cap = 256 * 12;  // persistent: a few workgroups per CU, each walks a strided set of tiles
    if (wgs > cap) wgs = (cap / p.nsplit) * p.nsplit;
    hipLaunchKernelGGL((conv1x1_stream_bf16_kernel<KS, NB>), dim3(wgs), dim3(256), 0, st, p);
    return hipGetLastError();
}

}  // namespace

// shapes the streaming kernel covers: K in {64, 128, 256}, N a multiple of 128 * NB
int hpe_stream_bf16_supported(int N, int K) {
    if (K == 64 || K == 128) return (N % 256) == 0;
    if (K == 256) return (N % 128) == 0;
    return 0;
}

// number of bf16x8 fragments of the packed weights: [N/32][K/16 + 1][64]
size_t hpe_stream_bf16_weight_frags(int N, int K) { return (size_t)(N / 32) * (K / 16 + 1) * 64; }

hipError_t hpe_launch_conv1x1_stream_bf16(const void* x, const void* w_packed, const void* res, void* y, int M, int N, int K, int relu,
                                          hipStream_t st) {
    if (!x || !w_packed || !y || M < 1 || !hpe_stream_bf16_supported(N, K)) return hipErrorInvalidValue;
    if (((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15)) return hipErrorInvalidValue;
    StreamArgs p{};
    p.x = static_cast<const __bf16*>(x);
    p.w = static_cast<const bf16x8*>(w_packed);
    p.res = static_cast<const __bf16*>(res);
    p.y = static_cast<__bf16*>(y);
    p.M = M;
    p.N = N;
    p.K = K;
    p.relu = relu;
    p.ntiles = (M + 31) / 32;
    switch (K) {
        case 64: return launch_stream<4, 2>(p, st);
        case 128: return launch_stream<8, 2>(p, st);
        case 256: return launch_stream<16, 1>(p, st);
        default: return hipErrorInvalidValue;
    }
}
