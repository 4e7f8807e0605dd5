// conv_chain_bf16.hip -- back-to-back 1x1 fusion inside the ResNet bottleneck chain (bf16 encoder, BASELINE config 4):
//
//     t3 = relu(bn2c(W2c . t2) + x)          last 1x1 ("expand", C -> 4C) of identity block i      (Keras res*_branch2c + add + ReLU)
//     u1 = relu(bn2a'(W2a' . t3))            first 1x1 ("reduce", 4C -> C') of identity block i+1   (Keras res*_branch2a + ReLU)
//
// as ONE launch (reference semantics: keras.applications.ResNet50 identity_block, src/models.py:35-41).
// The two layers as separate launches move (C + 4C + 4C) + (4C + C') bf16 per pixel; here t3 is written once (it is the next
// block's residual) and never read back: (C + 4C + 4C + C') -- 10 C instead of 14 C per pixel, -29 % of the bytes of the pair,
// and the encoder is HBM-bound in bf16 (stage 2: the two launches run at 4.5-5.5 TB/s today).
//
// Rounding points are those of the two-launch path: t3 is rounded to bf16 where it is stored, and the second GEMM multiplies exactly
// those bf16 values; only the fp32 summation order inside a k-slab differs.
//
// Structure (one workgroup = 64 pixels x all 4C channels, 4 waves, two workgroups per CU; three in the conv_block form):
//   * the workgroup walks the 4C axis in chunks of NC channels (128 in stage 2, 64 in stage 3).  Per chunk: GEMM-A (64 x NC, K = C) from
//     the resident t2 tile, epilogue IN PLACE on the residual chunk that LDS-DMA has put in the A-operand slab layout (so the result is
//     at once the bytes to store and the A operand of the next GEMM), 16-B row stores of t3, GEMM-B partial sums (64 x C', K = NC of
//     this chunk) in registers across the chunks.  The 4C-wide tensor never exists whole on the chip.  Two barriers per chunk.
//   * accumulators are kept TRANSPOSED (the weight fragment is the MFMA's A operand): a lane owns one pixel and 4 consecutive
//     channels per register quad, so the in-place epilogue is one 8-byte LDS read + one 8-byte LDS write per quad.
//   * a workgroup computes for ~1 us per tile and HBM answers after ~3 us, so everything a tile needs is requested up front: the t2
//     tile and a 32 KB ring of residual chunks (all of them in stage 2, four of eight in stage 3; refilled as chunks retire) -- two
//     workgroups per CU keep ~100 KB in flight.  The weights of a chunk (16 KB for GEMM-A, 16 KB for GEMM-B, L2 hits) have one slot
//     each and are requested as soon as the previous chunk has released the slot.
//   * every wave issues a QUARTER of every DMA group and of the t3 stores (one loader wave only reaches ~25 GB/s; a version with
//     dedicated loader waves was slower), so all four waves hold the same queue, and every barrier is preceded by a COUNTED vmcnt
//     wait that names exactly what may stay in flight behind the youngest thing the barrier publishes (vmcnt is per wave and counts
//     LDS-DMA, loads and stores together, in issue order); the issue order is fixed in the prologue comment below.  On the last,
//     partial tile the counts are not exact (skipped stores) and every wait drains the queue.
//   * every wait in front of a barrier that publishes LDS-DMA data is written out: hipcc's __syncthreads() only waits for lgkmcnt
//     behind some DMA patterns and for vmcnt(0) behind others (rounds 3-4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "hpe_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// per-channel scale / shift are read through the CONSTANT address space: the address is wave-uniform, so the loads become s_load (the
// kernel also stores to global memory, which makes hipcc fall back to per-lane global_load for plain pointers -- those would queue
// behind the LDS-DMA in flight, vmcnt counts in issue order)
typedef const __attribute__((address_space(4))) float* cfloat_p;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f32x4* cf32x4_p;

#include "bf16_rows.h"

namespace {

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// The same with the non-temporal cache policy (aux bit 1) for the residual, which is read exactly once; the t3 rows (4C wide, next read a
// whole 3x3 layer later) are stored non-temporal too.  What that buys is cache room for the C-wide u1 rows, which the next launch (the
// 3x3 layer) reads: A/B on one box (tools/chain_nt_ab.sh, -DHPE_CHAIN_NO_NT = plain policy): res2c_branch2b 0.109 -> 0.097 ms, the chained
// launches themselves 0.200 -> 0.195 / 0.132 -> 0.129 ms, bf16 step 69.4-69.5 k -> 70.1-70.8 k img/s.
__device__ __forceinline__ void dma16_nt(const void* src, void* lds_dst) {
#ifndef HPE_CHAIN_NO_NT
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}

// Barriers are written out: __syncthreads() is a workgroup fence + s_barrier, and behind pending LDS-DMA hipcc puts a vmcnt(0) in
// front of some of them (here: the one after the in-place epilogue, which would make the waves wait for the residual chunk and the
// weight slot they have only just requested) and only lgkmcnt(0) in front of others.  lds_barrier: LDS traffic of this wave is done, DMAs stay in flight;
// the caller adds wait_dma() where the barrier publishes this wave's LDS-DMA data.
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// counted wait: all but the wave's n youngest vector-memory operations (LDS-DMA, loads and stores count together, in issue order) are
// done.  n is a compile-time constant after unrolling; inline asm needs an immediate.
__device__ __forceinline__ void wait_dma_leaving(int n, bool exact_counts) {
    if (!exact_counts) n = 0;
    switch (n) {
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // 0, and anything unforeseen: over-waiting is safe
    }
}

// C = channels of t2, 4C = channels of t3, CP = output channels of the reduce GEMM, NC = channels per chunk.
// C2 > 0: the conv_block form -- no residual; the expand GEMM has a second A source x2 [M, C2] (the block input, stride 1) behind t2 along
// k, i.e. branch2c and the projection shortcut branch1 as one GEMM over [t2 | x2] with the BN scales folded into the weights (GEMM_DUAL of
// conv_gemm_bf16.hip), followed by the next block's branch2a.
// BM = pixels per workgroup: 64 (4 waves, two or three workgroups per CU) or 128 (8 waves, ONE workgroup per CU with all 160 KB of LDS:
// stage 4, where a chunk's weights are 64 KB -- twice the pixels per workgroup halve the weight stream per pixel).
template <int C, int CP, int NC, int C2, int BM = 64>
__global__ __launch_bounds__(BM * 4, BM == 128 ? 2 : (C2 > 0 ? 3 : 2)) void chain_expand_reduce_bf16_kernel(ChainArgs p) {
    constexpr int C4 = 4 * C;
    constexpr int NW = BM / 16;    // waves: wave (wm, wn) owns pixels 32 wm .. + 31 and one half of the channels
    constexpr int NTHR = 64 * NW;
    constexpr bool HAS_RES = C2 == 0;
    constexpr int KSA = (C + C2) / 64, KSB = NC / 64, NCH = C4 / NC;
    constexpr int SLAB = BM * 128;  // bytes of one [BM rows x 64 bf16] slab
    // residual ring: 32 KB of chunks requested ahead (identity blocks); without a residual (conv_block form) Q only stages the t3 rows
    // of the current chunk: one buffer
    constexpr int QBYTES = KSB * SLAB, QBUFS = HAS_RES ? 32768 / QBYTES : 1;
    constexpr int WA_SLAB = NC * 128, WB_SLAB = CP * 128;  // one k-slab of the GEMM-A / GEMM-B weights of a chunk
    constexpr int AT_OFF = 0, Q_OFF = KSA * SLAB, WA_OFF = Q_OFF + QBUFS * QBYTES, WB_OFF = WA_OFF + KSA * WA_SLAB;
    constexpr int LDS_BYTES = WB_OFF + KSB * WB_SLAB;
    constexpr int NTA = NC / 64, NTB = CP / 64;  // 32-wide channel blocks per wave (a wave owns half the chunk / half of C')
    // vector-memory instructions per wave of each group (every wave issues a quarter of every group)
    constexpr int RI = HAS_RES ? KSB * 2 : 0, WIB = KSB * CP / (8 * NW), ST = (BM * NC / 8) / NTHR;  // residual chunk, GEMM-B weight slot, t3 stores
    static_assert(C % 64 == 0 && C2 % 64 == 0 && C4 % NC == 0 && (NC == 64 || NC == 128) && CP % 64 == 0 && (BM == 64 || BM == 128), "geometry");
    static_assert(NC % (8 * NW) == 0 && CP % (8 * NW) == 0 && (BM * NC / 8) % NTHR == 0 && (BM * CP / 8) % NTHR == 0, "every wave issues an equal share");
    static_assert(QBUFS >= 1 && QBUFS <= NCH && (!HAS_RES || QBUFS * QBYTES == 32768), "residual ring");
    static_assert(KSA * SLAB >= (CP / 64) * SLAB, "the u1 tile reuses the t2 tile's space");
    // identity blocks: two workgroups per CU (72-80 KB each, ~40 KB of HBM requests in flight per workgroup).  The conv_block form has no
    // residual ring to keep in flight (24 KB per workgroup: the [t2 | x] tile and the first weight slots), so it is built for THREE
    // workgroups per CU (48 KB: 64-channel chunks, one Q buffer)
    static_assert(LDS_BYTES <= (BM == 128 ? 160 : (HAS_RES ? 80 : 53)) * 1024, "workgroups per CU");

    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    // The counted waits assume that every wave has issued every store instruction of the schedule.  On the last, partial tile a wave whose
    // rows all lie beyond M skips its stores (exec == 0 branches over them): there every wait drains the queue instead.
    const bool full_tile = m0 + BM <= p.M;

    // ---- DMA sources.  One wave-instruction fills 8 rows x 128 B; lane l -> row l >> 3, 16-B position l & 7 holds logical chunk
    //      (l & 7) ^ ((row >> 1) & 7) (the swizzle the fragment reads undo).  Every address is a wave-uniform base (SGPR pair) + a
    //      32-bit per-lane byte offset (64-bit per-lane addresses for every unrolled DMA cost 255 VGPRs).
    const int drow = lane >> 3;
    // Every wave issues an equal share of every group: instructions ii = wave, wave + NW, ... (rows 8 ii .. + 7).  The swizzle term
    // ((row >> 1) & 7) = (4 ii + (drow >> 1)) & 7 depends on the parity of ii only, i.e. on the wave (NW is even).
    const int r0 = 8 * wave + drow;
    const unsigned swz_a = (((lane & 7) ^ ((r0 >> 1) & 7)) * 8) * 2;
    unsigned off_t2[2], off_res[2], off_x2[2];  // activation rows r0, r0 + 8 NW, clamped to M - 1 on the last tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + r0 + 8 * NW * i;
        if (m >= p.M) m = p.M - 1;
        off_t2[i] = (unsigned)m * (C * 2) + swz_a;
        off_res[i] = (unsigned)m * (C4 * 2) + swz_a;
        off_x2[i] = (unsigned)m * (C2 * 2) + swz_a;
    }
    const unsigned off_wa = (unsigned)r0 * (unsigned)(p.ldw2c * 2) + swz_a;  // weight rows r0 + 32 i: the row step is wave-uniform
    const unsigned off_wb = (unsigned)r0 * (unsigned)(p.ldw2a * 2) + swz_a;
    const char* T2b = reinterpret_cast<const char*>(p.t2);
    const char* RESb = reinterpret_cast<const char*>(p.res);
    const char* WAb = reinterpret_cast<const char*>(p.w2c);
    const char* WBb = reinterpret_cast<const char*>(p.w2a);

    const char* X2b = reinterpret_cast<const char*>(p.x2);
    auto issue_at = [&]() {  // [t2 | x2] tile -> AT: KSA slabs
#pragma unroll
        for (int s = 0; s < KSA; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* src = s < C / 64 ? T2b + s * 128 + off_t2[i] : X2b + (s - C / 64) * 128 + off_x2[i];
                dma16(src, lds + AT_OFF + s * SLAB + (wave + NW * i) * 1024);
            }
    };
    auto issue_res = [&](int c) {  // residual chunk c -> Q[c % QBUFS]: KSB slabs
#pragma unroll
        for (int s = 0; s < KSB; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                dma16_nt(RESb + (c * NC + s * 64) * 2 + off_res[i], lds + Q_OFF + (c % QBUFS) * QBYTES + s * SLAB + (wave + NW * i) * 1024);
    };
    auto issue_wa = [&](int c) {  // W2c rows [c * NC, + NC), all KSA k-slabs -> slot A
#pragma unroll
        for (int s = 0; s < KSA; ++s) {
            const char* base = WAb + ((size_t)(c * NC) * p.ldw2c + s * 64) * 2;
#pragma unroll
            for (int i = 0; i < NC / (8 * NW); ++i)
                dma16(base + (size_t)(8 * NW * i) * p.ldw2c * 2 + off_wa, lds + WA_OFF + s * WA_SLAB + (wave + NW * i) * 1024);
        }
    };
    auto issue_wb = [&](int c) {  // W2a' rows [0, CP), k-slabs of chunk c -> slot B
#pragma unroll
        for (int s = 0; s < KSB; ++s) {
            const char* base = WBb + (c * NC + s * 64) * 2;
#pragma unroll
            for (int i = 0; i < CP / (8 * NW); ++i)
                dma16(base + (size_t)(8 * NW * i) * p.ldw2a * 2 + off_wb, lds + WB_OFF + s * WB_SLAB + (wave + NW * i) * 1024);
        }
    };

    // ---- fragment addressing (byte offsets inside a slab)
    const int wm = wave >> 1;  // pixel half (32 rows) in both GEMMs
    const int wn = wave & 1;   // channel half
    const int ar = wm * 32 + (lane & 31);
    const int a_off = ar * 128, a_x = (ar >> 1) & 7;
    int wa_off[NTA], wa_x[NTA], wb_off[NTB], wb_x[NTB];
#pragma unroll
    for (int j = 0; j < NTA; ++j) {
        const int r = wn * (NC / 2) + j * 32 + (lane & 31);
        wa_off[j] = r * 128;
        wa_x[j] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        const int r = wn * (CP / 2) + j * 32 + (lane & 31);
        wb_off[j] = r * 128;
        wb_x[j] = (r >> 1) & 7;
    }

    f32x16 accB[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accB[j][e] = 0.f;

    // ---- prologue: everything the first chunks need, at once.  The issue ORDER below is what the counted waits rely on:
    //      t2 tile, res(0), A(0), B(0), res(1 .. QBUFS-1); then per chunk c: [barrier 1] B(c), res(c - 1 + QBUFS) (c > 0)
    //      [barrier 2] A(c + 1), t3 stores of chunk c.
    issue_at();
    if (HAS_RES) issue_res(0);
    issue_wa(0);
    issue_wb(0);
    if (HAS_RES) {
#pragma unroll
        for (int c = 1; c < QBUFS; ++c) issue_res(c);
    }

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int qb = c % QBUFS;
        // ---- barrier 1: the GEMM-A weights and the residual of this chunk (and the t2 tile when c == 0) have landed -- the youngest of
        //      them is A(c); what was issued after it stays in flight -- and every wave is out of chunk c - 1.
        wait_dma_leaving(c == 0 ? WIB + (QBUFS - 1) * RI : ST, full_tile);
        lds_barrier();
        if (c > 0) issue_wb(c);                                      // slot B: GEMM-B of chunk c - 1 is done
        if (HAS_RES && c > 0 && c - 1 + QBUFS < NCH) issue_res(c - 1 + QBUFS);  // Q[(c - 1) % QBUFS]: read and stored

        // ================= GEMM-A: accA[n][m] = W2c[chunk c] . t2 tile
        f32x16 accA[NTA];
#pragma unroll
        for (int j = 0; j < NTA; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) accA[j][e] = 0.f;
#pragma unroll
        for (int sa = 0; sa < KSA; ++sa)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 2 * g + hi;
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(lds + AT_OFF + sa * SLAB + a_off + ((lc ^ a_x) << 4));
#pragma unroll
                for (int j = 0; j < NTA; ++j) {
                    const bf16x8 fw = *reinterpret_cast<const bf16x8*>(lds + WA_OFF + sa * WA_SLAB + wa_off[j] + ((lc ^ wa_x[j]) << 4));
                    accA[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fa, accA[j], 0, 0, 0);
                }
            }

        // ================= epilogue-A, in place on Q[qb]: q = bf16(relu(acc * scale + shift + q))
        // lane: pixel row ar, channels nl0 + 8 g + 4 hi + (0..3) of block j  ->  8 bytes of the slab row
        {
            unsigned char* Q = lds + Q_OFF + qb * QBYTES;
#pragma unroll
            for (int j = 0; j < NTA; ++j) {
                const int nl0 = wn * (NC / 2) + j * 32;  // first channel of the block inside the chunk (wave-uniform)
                cfloat_p scp = (cfloat_p)(p.scaleA + c * NC + nl0);
                cfloat_p shp = (cfloat_p)(p.shiftA + c * NC + nl0);
                unsigned char* Qs = Q + (nl0 >> 6) * SLAB + a_off;
                const int kc0 = (nl0 & 63) >> 3;  // 16-B chunk of channel nl0 inside the 64-channel slab row
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned char* q = Qs + (((kc0 + g) ^ a_x) << 4) + hi * 8;
                    bf16x4 rv = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    if (HAS_RES) rv = *reinterpret_cast<const bf16x4*>(q);
                    bf16x4 o;
                    // two s_load_dwordx4 per vector (channels 8 g .. + 3 for lanes 0-31, 8 g + 4 .. + 7 for lanes 32-63), selected per element
                    const f32x4 sc_lo = *reinterpret_cast<cf32x4_p>(scp + 8 * g), sc_hi = *reinterpret_cast<cf32x4_p>(scp + 8 * g + 4);
                    const f32x4 sh_lo = *reinterpret_cast<cf32x4_p>(shp + 8 * g), sh_hi = *reinterpret_cast<cf32x4_p>(shp + 8 * g + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sc = hi ? sc_hi[k] : sc_lo[k];
                        const float sh = hi ? sh_hi[k] : sh_lo[k];
                        o[k] = (__bf16)fmaxf(accA[j][4 * g + k] * sc + sh + (float)rv[k], 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(q) = o;
                }
            }
        }
        // ---- barrier 2: Q[qb] is the t3 tile of this chunk; the GEMM-B weights have landed (only residual chunks are younger); GEMM-A
        //      is done everywhere
        wait_dma_leaving(c == 0 ? (QBUFS - 1) * RI : (c - 1 + QBUFS < NCH ? RI : 0), full_tile);
        lds_barrier();
        if (c + 1 < NCH) issue_wa(c + 1);  // slot A is free

        // ================= t3 row stores + GEMM-B partial sums
        {
            const unsigned char* Q = lds + Q_OFF + qb * QBYTES;
            constexpr int UPR = NC / 8;  // 16-B units per row
#pragma unroll
            for (int pass = 0; pass < (BM * UPR) / NTHR; ++pass) {
                const int idx = pass * NTHR + t;
                const int r = idx / UPR, u = idx - r * UPR;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(Q + (u >> 3) * SLAB + r * 128 + (((u & 7) ^ ((r >> 1) & 7)) << 4));
                const int m = m0 + r;
#ifndef HPE_CHAIN_NO_NT
                if (m < p.M) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p.t3 + (size_t)m * C4 + c * NC + u * 8));
#else
                if (m < p.M) *reinterpret_cast<bf16x8*>(p.t3 + (size_t)m * C4 + c * NC + u * 8) = v;
#endif
            }
        }
#pragma unroll
        for (int sb = 0; sb < KSB; ++sb) {
            const unsigned char* Q = lds + Q_OFF + qb * QBYTES + sb * SLAB;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = 2 * g + hi;
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(Q + a_off + ((lc ^ a_x) << 4));
#pragma unroll
                for (int j = 0; j < NTB; ++j) {
                    const bf16x8 fw = *reinterpret_cast<const bf16x8*>(lds + WB_OFF + sb * WB_SLAB + wb_off[j] + ((lc ^ wb_x[j]) << 4));
                    accB[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fa, accB[j], 0, 0, 0);
                }
            }
        }
    }

    // ================= epilogue-B: u1 = bf16(relu(accB * scale' + shift')) -> LDS (the t2 tile's space, slab layout) -> 16-B row stores
    // (GEMM-A last read the t2 tile before barrier 2 of the last chunk: its space is free)
    {
        unsigned char* U = lds + AT_OFF;
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            const int nl0 = wn * (CP / 2) + j * 32;
            cfloat_p scp = (cfloat_p)(p.scaleB + nl0);
            cfloat_p shp = (cfloat_p)(p.shiftB + nl0);
            unsigned char* Us = U + (nl0 >> 6) * SLAB + a_off;
            const int kc0 = (nl0 & 63) >> 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
                const f32x4 sc_lo = *reinterpret_cast<cf32x4_p>(scp + 8 * g), sc_hi = *reinterpret_cast<cf32x4_p>(scp + 8 * g + 4);
                const f32x4 sh_lo = *reinterpret_cast<cf32x4_p>(shp + 8 * g), sh_hi = *reinterpret_cast<cf32x4_p>(shp + 8 * g + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sc = hi ? sc_hi[k] : sc_lo[k];
                    const float sh = hi ? sh_hi[k] : sh_lo[k];
                    o[k] = (__bf16)fmaxf(accB[j][4 * g + k] * sc + sh, 0.f);
                }
                *reinterpret_cast<bf16x4*>(Us + (((kc0 + g) ^ a_x) << 4) + hi * 8) = o;
            }
        }
    }
    lds_barrier();
    {
        constexpr int UPR = CP / 8;  // 16-B units per row
        const unsigned char* U = lds + AT_OFF;
#pragma unroll
        for (int pass = 0; pass < (BM * UPR) / NTHR; ++pass) {
            const int idx = pass * NTHR + t;
            const int r = idx / UPR, u = idx - r * UPR;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(U + (u >> 3) * SLAB + r * 128 + (((u & 7) ^ ((r >> 1) & 7)) << 4));
            const int m = m0 + r;
            if (m < p.M) *reinterpret_cast<bf16x8*>(p.u1 + (size_t)m * CP + u * 8) = v;
        }
    }
}

template <int C, int CP, int NC, int C2, int BM = 64>
hipError_t launch_chain(const ChainArgs& p, hipStream_t st) {
    const int grid = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL((chain_expand_reduce_bf16_kernel<C, CP, NC, C2, BM>), dim3(grid), dim3(BM * 4), 0, st, p);
    return hipGetLastError();
}

}  // namespace

// C2 = 0: identity block (residual); C2 > 0: conv_block with a stride-1 projection shortcut over C2 input channels (stage 2)
bool hpe_chain_bf16_supported(int C, int C4, int CP, int C2) {
    if (C4 != 4 * C || CP != C) return false;
    return C2 == 0 ? (C == 64 || C == 128 || C == 256) : (C == 64 && C2 == 64);
}

hipError_t hpe_launch_chain_bf16(const ChainArgs& p, int C, int C4, int CP, int C2, hipStream_t st) {
    if (!hpe_chain_bf16_supported(C, C4, CP, C2)) return hipErrorInvalidValue;
    if (p.M <= 0 || !p.t2 || !p.w2c || !p.w2a || !p.t3 || !p.u1 || !p.scaleA || !p.shiftA || !p.scaleB || !p.shiftB) return hipErrorInvalidValue;
    if (C2 == 0 ? !p.res : !p.x2) return hipErrorInvalidValue;
    if (p.ldw2c < C + C2 || p.ldw2a < C4 || (p.ldw2c % 8) != 0 || (p.ldw2a % 8) != 0) return hipErrorInvalidValue;
    if ((((uintptr_t)p.t2 | (uintptr_t)p.res | (uintptr_t)p.x2 | (uintptr_t)p.w2c | (uintptr_t)p.w2a | (uintptr_t)p.t3 | (uintptr_t)p.u1) & 15) != 0)
        return hipErrorInvalidValue;
    if (C2 > 0) return launch_chain<64, 64, 64, 64>(p, st);
    if (C == 64) return launch_chain<64, 64, 128, 0>(p, st);
    if (C == 256) return launch_chain<256, 256, 64, 0, 128>(p, st);
    return launch_chain<128, 128, 64, 0>(p, st);
}

// resident workgroups per CU of the three instantiations (the design needs 2): out[0] C = 64, out[1] C = 128, out[2] conv_block form
hipError_t hpe_chain_bf16_occupancy(int out[3]) {
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], reinterpret_cast<const void*>(chain_expand_reduce_bf16_kernel<64, 64, 128, 0>), 256, 0);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], reinterpret_cast<const void*>(chain_expand_reduce_bf16_kernel<128, 128, 64, 0>), 256, 0);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[2], reinterpret_cast<const void*>(chain_expand_reduce_bf16_kernel<64, 64, 64, 64>), 256, 0);
}
