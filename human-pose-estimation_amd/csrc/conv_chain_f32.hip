// conv_chain_f32.hip -- the chained 1x1 launch of conv_chain_bf16.hip in fp32, for the identity blocks of stage 2 (C = 64):
//
//     t3 = relu(bn2c(W2c . t2) + x)      res2*_branch2c + add + ReLU          (Keras identity_block, src/models.py:35-41)
//     u1 = relu(bn2a'(W2a' . t3))        res2*_branch2a + ReLU of the NEXT block
//
// in one launch: t3 is written once and not read back (10 C instead of 14 C floats per pixel).  Plan option chain_fuse bit 8, on by default.
// The round-3 review asked for the bf16 kernel "in fp32 for res2{b,c}_branch2c" (72 TFLOP/s, the weakest fp32 layers).  In fp32 the pair is
// 52 GFLOP = 0.33 ms of MFMA at peak AND 2.05 GB = 0.37-0.40 ms of HBM, against 0.61 ms for the two launches: both bounds bind at once, so
// the launch itself only gains 7 % (0.603 -> 0.559 ms in a serial pass) -- but a kernel that keeps the matrix pipe AND the memory system busy
// fills the concurrent step better than an HBM-bound launch followed by an MFMA-bound one: step +0.3 ... +1.4 % at B = 256 (two boxes),
// +1.6 % at B = 64.
//
// Structure = the bf16 kernel's (64 pixels x all 256 channels per workgroup, 4 waves, two workgroups per CU, 80 KB of LDS; chunks of 64
// channels; transposed accumulators; in-place epilogue on the residual chunk; every wave issues a quarter of every DMA group behind
// counted vmcnt waits), with 32-float k-slabs (the same 128-byte rows), v_mfma_f32_32x32x2_f32 (one ds_read_b128 per lane feeds four
// MFMAs: lanes 0-31 take k = 8 g .. 8 g + 3, lanes 32-63 k = 8 g + 4 .. 8 g + 7) and 16-byte epilogue accesses (4 channels per quad).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpe_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cfloat_p;
typedef const __attribute__((address_space(4))) f32x4* cf32x4_p;

namespace {

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void dma16_nt(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_dma_leaving(int n, bool exact_counts) {
    if (!exact_counts) n = 0;
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int C, int CP>
__global__ __launch_bounds__(256, 2) void chain_expand_reduce_f32_kernel(ChainArgsF32 p) {
    constexpr int BM = 64, NC = 64, C4 = 4 * C;
    constexpr int KSA = C / 32, KSB = NC / 32, NCH = C4 / NC;
    constexpr int SLAB = BM * 128;  // bytes of one [64 rows x 32 floats] slab
    constexpr int QBYTES = KSB * SLAB, QBUFS = 32768 / QBYTES;
    constexpr int WA_SLAB = NC * 128, WB_SLAB = CP * 128;
    constexpr int AT_OFF = 0, Q_OFF = KSA * SLAB, WA_OFF = Q_OFF + QBUFS * QBYTES, WB_OFF = WA_OFF + KSA * WA_SLAB;
    constexpr int LDS_BYTES = WB_OFF + KSB * WB_SLAB;
    constexpr int RI = KSB * 2, WIB = KSB * CP / 32, ST = NC / 16;  // per-wave instruction counts: residual chunk, GEMM-B weights, t3 stores
    static_assert(C == 64 && CP == 64, "stage 2 only: one 32-channel block per wave in both GEMMs");
    static_assert(QBUFS == 2 && NCH == 4 && LDS_BYTES <= 80 * 1024, "geometry");
    static_assert(RI == 4 && WIB == 4 && ST == 4, "the counted waits below know 0 / 4 / 8 only");

    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const bool full_tile = m0 + BM <= p.M;  // partial last tile: skipped stores make the counts inexact -> every wait drains the queue

    const int drow = lane >> 3;
    const int r0 = 8 * wave + drow;
    const unsigned swz_a = ((lane & 7) ^ ((r0 >> 1) & 7)) * 16;  // bytes
    unsigned off_t2[2], off_res[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m >= p.M) m = p.M - 1;
        off_t2[i] = (unsigned)m * (C * 4) + swz_a;
        off_res[i] = (unsigned)m * (C4 * 4) + swz_a;
    }
    const unsigned off_wa = (unsigned)r0 * (unsigned)(p.ldw2c * 4) + swz_a;
    const unsigned off_wb = (unsigned)r0 * (unsigned)(p.ldw2a * 4) + swz_a;
    const char* T2b = reinterpret_cast<const char*>(p.t2);
    const char* RESb = reinterpret_cast<const char*>(p.res);
    const char* WAb = reinterpret_cast<const char*>(p.w2c);
    const char* WBb = reinterpret_cast<const char*>(p.w2a);

    auto issue_at = [&]() {
#pragma unroll
        for (int s = 0; s < KSA; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16(T2b + s * 128 + off_t2[i], lds + AT_OFF + s * SLAB + (wave + 4 * i) * 1024);
    };
    auto issue_res = [&](int c) {
#pragma unroll
        for (int s = 0; s < KSB; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                dma16_nt(RESb + (c * NC + s * 32) * 4 + off_res[i], lds + Q_OFF + (c % QBUFS) * QBYTES + s * SLAB + (wave + 4 * i) * 1024);
    };
    auto issue_wa = [&](int c) {  // W2c rows [c * NC, + NC), k-slabs 0 .. KSA-1
#pragma unroll
        for (int s = 0; s < KSA; ++s) {
            const char* base = WAb + ((size_t)(c * NC) * p.ldw2c + s * 32) * 4;
#pragma unroll
            for (int i = 0; i < NC / 32; ++i)
                dma16(base + (size_t)(32 * i) * p.ldw2c * 4 + off_wa, lds + WA_OFF + s * WA_SLAB + (wave + 4 * i) * 1024);
        }
    };
    auto issue_wb = [&](int c) {  // W2a' rows [0, CP), k-slabs of chunk c
#pragma unroll
        for (int s = 0; s < KSB; ++s) {
            const char* base = WBb + (c * NC + s * 32) * 4;
#pragma unroll
            for (int i = 0; i < CP / 32; ++i)
                dma16(base + (size_t)(32 * i) * p.ldw2a * 4 + off_wb, lds + WB_OFF + s * WB_SLAB + (wave + 4 * i) * 1024);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;  // pixel half, channel half (one 32 x 32 block per wave in both GEMMs)
    const int ar = wm * 32 + (lane & 31);
    const int a_off = ar * 128, a_x = (ar >> 1) & 7;
    const int wr = wn * 32 + (lane & 31);
    const int w_off = wr * 128, w_x = (wr >> 1) & 7;

    f32x16 accB;
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[e] = 0.f;

    // issue order (what the counted waits rely on): t2 tile, res(0), A(0), B(0), res(1); per chunk c: [barrier 1] B(c), res(c + 1) (c > 0)
    // [barrier 2] A(c + 1), t3 stores of chunk c
    issue_at();
    issue_res(0);
    issue_wa(0);
    issue_wb(0);
    issue_res(1);

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int qb = c % QBUFS;
        wait_dma_leaving(c == 0 ? WIB + (QBUFS - 1) * RI : ST, full_tile);
        lds_barrier();
        if (c > 0) issue_wb(c);
        if (c > 0 && c - 1 + QBUFS < NCH) issue_res(c - 1 + QBUFS);

        // ---- GEMM-A (transposed): accA[channel][pixel] = W2c[chunk c] . t2 tile, K = C
        f32x16 accA;
#pragma unroll
        for (int e = 0; e < 16; ++e) accA[e] = 0.f;
#pragma unroll
        for (int sa = 0; sa < KSA; ++sa)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 2 * g + hi;
                const f32x4 fa = *reinterpret_cast<const f32x4*>(lds + AT_OFF + sa * SLAB + a_off + ((lc ^ a_x) << 4));
                const f32x4 fw = *reinterpret_cast<const f32x4*>(lds + WA_OFF + sa * WA_SLAB + w_off + ((lc ^ w_x) << 4));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[ks], fa[ks], accA, 0, 0, 0);
            }

        // ---- epilogue-A in place on Q[qb]: lane = pixel ar, quad q = channels 32 wn + 8 q + 4 hi + (0..3) = chunk 2 q + hi of slab wn
        {
            unsigned char* Qs = lds + Q_OFF + qb * QBYTES + wn * SLAB + a_off;
            cfloat_p scp = (cfloat_p)(p.scaleA + c * NC + wn * 32);
            cfloat_p shp = (cfloat_p)(p.shiftA + c * NC + wn * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned char* qp = Qs + (((2 * q + hi) ^ a_x) << 4);
                const f32x4 rv = *reinterpret_cast<const f32x4*>(qp);
                const f32x4 sc_lo = *reinterpret_cast<cf32x4_p>(scp + 8 * q), sc_hi = *reinterpret_cast<cf32x4_p>(scp + 8 * q + 4);
                const f32x4 sh_lo = *reinterpret_cast<cf32x4_p>(shp + 8 * q), sh_hi = *reinterpret_cast<cf32x4_p>(shp + 8 * q + 4);
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sc = hi ? sc_hi[k] : sc_lo[k];
                    const float sh = hi ? sh_hi[k] : sh_lo[k];
                    o[k] = fmaxf(accA[4 * q + k] * sc + sh + rv[k], 0.f);
                }
                *reinterpret_cast<f32x4*>(qp) = o;
            }
        }
        wait_dma_leaving(c == 0 ? (QBUFS - 1) * RI : (c - 1 + QBUFS < NCH ? RI : 0), full_tile);
        lds_barrier();
        if (c + 1 < NCH) issue_wa(c + 1);

        // ---- t3 row stores (16 B units: 16 per row) + GEMM-B partial sums
        {
            const unsigned char* Q = lds + Q_OFF + qb * QBYTES;
            constexpr int UPR = NC / 4;
#pragma unroll
            for (int pass = 0; pass < (BM * UPR) / 256; ++pass) {
                const int idx = pass * 256 + t;
                const int r = idx / UPR, u = idx - r * UPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Q + (u >> 3) * SLAB + r * 128 + (((u & 7) ^ ((r >> 1) & 7)) << 4));
                const int m = m0 + r;
                if (m < p.M) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.t3 + (size_t)m * C4 + c * NC + u * 4));
            }
        }
#pragma unroll
        for (int sb = 0; sb < KSB; ++sb) {
            const unsigned char* Q = lds + Q_OFF + qb * QBYTES + sb * SLAB;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 2 * g + hi;
                const f32x4 fa = *reinterpret_cast<const f32x4*>(Q + a_off + ((lc ^ a_x) << 4));
                const f32x4 fw = *reinterpret_cast<const f32x4*>(lds + WB_OFF + sb * WB_SLAB + w_off + ((lc ^ w_x) << 4));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[ks], fa[ks], accB, 0, 0, 0);
            }
        }
    }

    // ---- epilogue-B: u1 = relu(accB * scale' + shift') -> LDS (the t2 tile's space) -> 16-B stores, row-major or channel-slab major
    {
        unsigned char* Us = lds + AT_OFF + wn * SLAB + a_off;
        cfloat_p scp = (cfloat_p)(p.scaleB + wn * 32);
        cfloat_p shp = (cfloat_p)(p.shiftB + wn * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 sc_lo = *reinterpret_cast<cf32x4_p>(scp + 8 * q), sc_hi = *reinterpret_cast<cf32x4_p>(scp + 8 * q + 4);
            const f32x4 sh_lo = *reinterpret_cast<cf32x4_p>(shp + 8 * q), sh_hi = *reinterpret_cast<cf32x4_p>(shp + 8 * q + 4);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sc = hi ? sc_hi[k] : sc_lo[k];
                const float sh = hi ? sh_hi[k] : sh_lo[k];
                o[k] = fmaxf(accB[4 * q + k] * sc + sh, 0.f);
            }
            *reinterpret_cast<f32x4*>(Us + (((2 * q + hi) ^ a_x) << 4)) = o;
        }
    }
    lds_barrier();
    {
        const unsigned char* U = lds + AT_OFF;
#pragma unroll
        for (int pass = 0; pass < (BM * (CP / 4)) / 256; ++pass) {
            const int idx = pass * 256 + t;
            int r, u;
            if (p.u1_slab8) {  // y[(n / 8) * M + m][n % 8]: consecutive lanes = the two halves of consecutive pixels of one 8-channel slab
                u = ((idx >> 7) << 1) | (idx & 1);
                r = (idx >> 1) & 63;
            } else {
                r = idx / (CP / 4);
                u = idx - r * (CP / 4);
            }
            const f32x4 v = *reinterpret_cast<const f32x4*>(U + (u >> 3) * SLAB + r * 128 + (((u & 7) ^ ((r >> 1) & 7)) << 4));
            const int m = m0 + r;
            if (m < p.M) {
                if (p.u1_slab8) *reinterpret_cast<f32x4*>(p.u1 + ((size_t)(u >> 1) * p.M + m) * 8 + (u & 1) * 4) = v;
                else *reinterpret_cast<f32x4*>(p.u1 + (size_t)m * CP + u * 4) = v;
            }
        }
    }
}

}  // namespace

bool hpe_chain_f32_supported(int C, int C4, int CP) { return C == 64 && C4 == 256 && CP == 64; }

hipError_t hpe_launch_chain_f32(const ChainArgsF32& p, int C, int C4, int CP, hipStream_t st) {
    if (!hpe_chain_f32_supported(C, C4, CP)) return hipErrorInvalidValue;
    if (p.M <= 0 || !p.t2 || !p.res || !p.w2c || !p.w2a || !p.t3 || !p.u1 || !p.scaleA || !p.shiftA || !p.scaleB || !p.shiftB) return hipErrorInvalidValue;
    if (p.ldw2c < C || p.ldw2a < C4 || (p.ldw2c % 4) != 0 || (p.ldw2a % 4) != 0) return hipErrorInvalidValue;
    if ((((uintptr_t)p.t2 | (uintptr_t)p.res | (uintptr_t)p.w2c | (uintptr_t)p.w2a | (uintptr_t)p.t3 | (uintptr_t)p.u1) & 15) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((chain_expand_reduce_f32_kernel<64, 64>), dim3((p.M + 63) / 64), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t hpe_chain_f32_occupancy(int* out) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, reinterpret_cast<const void*>(chain_expand_reduce_f32_kernel<64, 64>), 256, 0);
}
