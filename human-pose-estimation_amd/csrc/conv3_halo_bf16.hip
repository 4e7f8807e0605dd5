// conv3_halo_bf16.hip -- 3x3 / stride 1 / SAME convolution in bf16 with the activation tile AND its halo resident in LDS (round 4).
//
// Why: the implicit-GEMM kernel (conv_gemm_bf16.hip) treats the 9 taps as 9 k-slabs and DMAs the activation rows of a tile into LDS once
// PER TAP: a 128 x 128 tile moves 32 KB into LDS per 2.1 MFLOP (64 FLOP/B).  The LDS-DMA staging path of these kernels delivers
// ~10-11 TB/s chip-wide (~42 GB/s per CU; every stage 3-5 layer of the bf16 encoder measures 8.2-10.7 TB/s of fill -- paced by the issue
// of the DMA instructions), which caps that kernel at 0.6-0.7 PFLOP/s -- 28 % of the bf16 matrix rate -- whatever
// the HBM traffic is.  Here a workgroup owns BM consecutive pixels of the
// flattened [B*H*W] axis and stages, per 64-channel input slab, the BM + 2(W+1) activation rows ONCE; the 9 taps read it at 9 row shifts.
// Out-of-image taps read a zero row instead (one v_cndmask on the address, not on the data).  What still streams per tap is the
// [BN x 64] weight slab: fill per step (BM = 256, BN = 128) = 16 KB weights + 4.4 KB of the next activation image for 4.2 MFLOP = 205 FLOP/B.
// Measured (B = 256, ms per layer, implicit GEMM -> this kernel): 56x56 0.103 -> 0.070, 28x28 0.090 -> 0.063, 14x14 0.084 -> 0.0615,
// 7x7 0.085 -> 0.0605 (profiles/r04/layers_bf16_halo3_ab.txt).
//
// Structure: steps s = (input slab cs, tap t), 9 CS of them; one barrier per step; four 16-deep fragment groups per step.
//   * weights: ring of 4 slabs ([BN rows x 128 B], XOR-swizzled like every slab here) at the bottom of the LDS, W(s + 3) requested in
//     the middle of step s.  9 = 1 mod 4: slab cs starts at ring position cs mod 4; the slab loop is unrolled by 4, so every ring
//     position is a compile-time constant and goes into the immediate offset of the fragment reads
//   * activations: two image buffers (one when Cin = 64, and in the A1 form); the pieces of image cs + 1 are requested one per step
//     during taps 0 .. API-1 of slab cs (A1: all at the slab boundary -- two workgroups per CU cover the bubble)
//   * the step's barrier sits between fragment groups 0 and 1 and is about the NEXT step ("W(s + 1) has landed everywhere, everyone is
//     past step s - 1"): the first groups of step s + 1 are read while step s still multiplies
//   * every wave issues the same number of DMA instructions per step, so "slab s + 1 has landed" is a counted vmcnt wait (loads complete
//     in order) in front of the barrier; no stores inside the loop
//   * fragment reads: inline asm, three register sets, the two groups after the current one in flight behind counted lgkmcnt waits
//     (below: why, and what tools/isa_lint.py checks about it)
//   * accumulators transposed (weight fragment = MFMA operand A): a lane owns one pixel x 4 consecutive channels, so the epilogue writes
//     8-B pieces into an LDS image of the output tile, which leaves as 16-B row stores
// Oracle: tests/test_gpu_parity.py::test_bf16_halo3_* against the fp64 convolution of the bf16-rounded operands and the round-2 kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "hpe_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cfloat_p;
typedef const __attribute__((address_space(4))) f32x4* cf32x4_p;

namespace {

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Fragment reads are inline asm with COUNTED lgkmcnt waits.  hipcc never emits a counted lgkmcnt in a kernel that issues LDS-DMA inside
// the loop (a global_load_lds is a FLAT-encoded instruction that touches LDS; LLVM's waitcnt pass then treats the counter as unordered
// and falls back to lgkmcnt(0) -- checked on a six-line kernel).  lgkmcnt(0) in front of the multiplies of a group also waits for the
// reads of the NEXT group that have just been issued: both waves of a SIMD then sit in LDS latency with an idle matrix pipe (rocprofv3:
// SQ_WAIT_ANY 38 % of the wave cycles, although neither the DMA waits nor the barriers cost anything -- removing both changed the layer
// times by < 3 %).  LDS operations of a wave return in order, so "all but the youngest MT + NT" is exact.  The compiler does not know
// that the asm results arrive late: tools/isa_lint.py replays the lgkmcnt queue over the disassembly and fails the build check if any
// instruction reads a fragment register between its ds_read and the wait that covers it (a register copy inserted by the allocator).
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// the multiplies that follow consume the operands: tying them to the wait keeps every use behind it
#define HPE_LGKM_WAIT4(N, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define HPE_LGKM_WAIT3(N, a, b, c) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c))
#define HPE_LGKM_WAIT12(N, a, b, c, d, e, f, g, h, i, j, k, l) \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i), "+v"(j), "+v"(k), "+v"(l))

// all but the wave's n youngest vector-memory operations are done; n is a compile-time constant after unrolling
__device__ __forceinline__ void wait_dma_leaving(int n) {
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // 0, and anything unforeseen: over-waiting is safe
    }
}

// HW = map height = width, CIN = input channels, BN = output channels per workgroup, BM = pixels per workgroup (BM / 32 waves),
// NSW = weight ring depth
// A1: ONE image buffer also when Cin > 64 (with BN = 64: 72 KB, two workgroups per CU): the next slab's image is requested when the
// current slab is done -- the bubble is the other workgroup's to fill.
template <int HW, int CIN, int BN, int BM, int NSW, bool A1 = false>
__global__ __launch_bounds__(BM * 2, A1 ? 4 : (BM == 256 ? 2 : 1)) void conv3_halo_bf16_kernel(Halo3Args p) {
    constexpr int NW = BM / 32, NTHR = 64 * NW;
    constexpr int CS = CIN / 64;           // 64-channel input slabs
    constexpr int HALO = HW + 1;           // rows in front of / behind the tile that its taps reach
    constexpr int NPIECE = (BM + 2 * HALO + 7) / 8;
    constexpr int API = (NPIECE + NW - 1) / NW;  // activation DMA instructions per wave and image
    constexpr int A_BYTES = API * NW * 1024;
    constexpr int ABUFS = (CS > 1 && !A1) ? 2 : 1;
    constexpr int W_BYTES = BN * 128;      // one weight slab
    constexpr int WI = BN / (8 * NW);      // weight DMA instructions per wave and slab
    // the weight ring sits at the bottom of the LDS: ring buffer + fragment offsets stay below 64 KB, i.e. inside the immediate offset
    // field of ds_read_b128 (one address VGPR per fragment instead of one per fragment and ring position)
    // Cin = 64 (one image, requested whole in the prologue): the zero row lives in the unused tail of the image allocation, whose DMA
    // piece is skipped -- 80.1 KB would be one workgroup per CU, 79.1 KB are two
    constexpr bool ZERO_IN_TAIL = CS == 1 && API * NW > NPIECE;
    constexpr int W_OFF = 0, A_OFF = NSW * W_BYTES + (ZERO_IN_TAIL ? 0 : 256), ZERO_OFF = ZERO_IN_TAIL ? A_OFF + NPIECE * 1024 : NSW * W_BYTES;
    constexpr int LDS_BYTES = A_OFF + ABUFS * A_BYTES;
    constexpr int MT = 2, NT = BN / 64;    // wave tile: 64 pixels x BN / 2 channels
    constexpr int SLAB = BM * 128;         // output staging: [BM rows x 64 channels]
    static_assert(CIN % 64 == 0 && BN % 64 == 0 && BN % (8 * NW) == 0 && (BM == 128 || BM == 256), "geometry");
    static_assert(NSW == 4 && (CS <= 4 || (CS % 4 == 0 && (NT == 2 || A1))), "ring positions are compile-time constants: 9 = 1 mod 4, period 4 slabs");
    static_assert(CS == 1 || A1 || API <= 6, "the pieces of the next image are requested one per step, taps 0 .. API-1, and are older than W(cs + 1, 0)");
    static_assert((BN / 64) * SLAB <= LDS_BYTES && LDS_BYTES <= 160 * 1024 && NSW * W_BYTES <= 65536, "LDS");
    static_assert((BM * BN / 8) % NTHR == 0, "row stores");

    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hi = lane >> 5;
    // consecutive workgroup ids share an XCD's L2 every 8th: give each XCD a contiguous run of tiles (neighbouring pixel tiles share
    // halo rows, the channel tiles of one pixel tile share the whole image)
    const int total = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int q = total >> 3, rr = total & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blockIdx.x >> 3);
    const int mtile = __builtin_amdgcn_readfirstlane(swz / p.n_ntiles);  // (the division runs on the VALU)
    const int n0 = (swz - mtile * p.n_ntiles) * BN;
    const int m0 = mtile * BM;

    if (t < 16) reinterpret_cast<f32x4*>(lds + ZERO_OFF)[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA sources: one wave-instruction = 8 rows x 128 B; lane l -> row l >> 3, 16-B position l & 7 holds logical chunk
    //      (l & 7) ^ ((row >> 1) & 7).  Pieces pi = wave + NW i: (row >> 1) & 7 = (4 pi + (drow >> 1)) & 7 depends on the wave's parity only.
    const int drow = lane >> 3;
    const int r0 = 8 * wave + drow;
    const unsigned swz_a = (((lane & 7) ^ ((r0 >> 1) & 7)) * 8) * 2;
    const unsigned off_w = (unsigned)r0 * (unsigned)(p.ldw * 2) + swz_a;
    const char* Xb = reinterpret_cast<const char*>(p.x);
    const char* Wb = reinterpret_cast<const char*>(p.w) + (size_t)n0 * p.ldw * 2;

    auto issue_a_piece = [&](int cs, int i) {  // piece i of this wave of image cs -> buffer cs % ABUFS
        int g = m0 - HALO + r0 + 8 * NW * i;   // global pixel row (clamped: rows outside the buffer are never read unmasked)
        g = g < 0 ? 0 : (g >= p.M ? p.M - 1 : g);
        // the slab offset stays in SGPRs (opaque): folded into the address it becomes one per-lane 64-bit VGPR address per unrolled
        // (slab, tap, piece) -- 150 VGPRs of constants and spills -- instead of the SGPR-base + 32-bit-offset form of the instruction
        const char* base = Xb + cs * 128;
        asm volatile("" : "+s"(base));
        dma16(base + ((unsigned)g * (CIN * 2) + swz_a), lds + A_OFF + (cs % ABUFS) * A_BYTES + (wave + NW * i) * 1024);
    };
    auto issue_w = [&](int cs, int tap, int buf) {  // weights [n0 .. + BN) x (tap, slab cs) -> ring buffer buf
        const char* base = Wb + (tap * CIN + cs * 64) * 2;
        asm volatile("" : "+s"(base));
#pragma unroll
        for (int i = 0; i < WI; ++i) dma16(base + (size_t)(8 * NW * i) * p.ldw * 2 + off_w, lds + W_OFF + buf * W_BYTES + (wave + NW * i) * 1024);
    };

    // ---- fragment addressing
    const int wm = wave >> 1, wn = wave & 1;
    int arow[MT];       // local image row of the lane's pixel at the centre tap
    unsigned amask[MT];  // 9 bits: tap inside the image
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ml = wm * 64 + i * 32 + (lane & 31);
        arow[i] = ml + HALO;
        int m = m0 + ml;
        if (m >= p.M) m = p.M - 1;
        const int xw = m % HW, yh = (m / HW) % HW;
        unsigned mk = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dh = tap / 3 - 1, dw = tap % 3 - 1;
            if ((unsigned)(yh + dh) < (unsigned)HW && (unsigned)(xw + dw) < (unsigned)HW) mk |= 1u << tap;
        }
        amask[i] = mk;
    }
    int wfrag[NT][4];  // byte offset inside a weight slab of fragment g of channel block j
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * (BN / 2) + j * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) wfrag[j][g] = r * 128 + (((2 * g + hi) ^ ((r >> 1) & 7)) << 4);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragments of one 16-deep group: three register sets, the two groups after the current one are being read from LDS while the current
    // one multiplies (group G = 4 tap + g of a slab uses set G % 3; a slab has 36 groups, so the set is a compile-time constant)
    bf16x8 fa[3][MT], fw[3][NT];
    unsigned aoff[MT], ax[MT];  // image row address / swizzle term of the step whose fragments are read next
    auto tap_addr = [&](int cs, int tap) {
        const int shift = (tap / 3 - 1) * HW + (tap % 3 - 1);
        const int abuf = A_OFF + (cs % ABUFS) * A_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // opaque to the optimiser: these addresses do not depend on the slab, and hoisting the 9 x MT x 4 of them out of the slab loop
            // (or sharing them between unrolled slabs) costs 70+ VGPRs; recomputing them is ~30 VALU per step
            int ar = arow[i];
            asm volatile("" : "+v"(ar));
            const int lr = ar + shift;
            const bool ok = (amask[i] >> tap) & 1u;
            // an out-of-image tap reads the 256 B of zeros at the SAME bank slot its image row would have used (row parity and swizzle
            // term kept): the 16-lane groups of ds_read_b128 stay conflict-free (with one shared zero address 17-24 % of the LDS cycles
            // of these kernels were bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
            aoff[i] = ok ? (unsigned)(abuf + lr * 128) : (unsigned)(ZERO_OFF + ((lr & 1) << 7));
            ax[i] = (unsigned)((lr >> 1) & 7);
        }
    };
    const unsigned lds_base = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) unsigned char*)lds);
    auto load_frags = [&](int set, int wbuf, int g) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[set][i] = lds_read16<0>(lds_base + aoff[i] + (((2 * g + hi) ^ ax[i]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const unsigned a = lds_base + (unsigned)wfrag[j][g];
            switch (wbuf) {  // a constant in every unrolled copy: the ring position goes into the immediate offset
                case 0: fw[set][j] = lds_read16<W_OFF>(a); break;
                case 1: fw[set][j] = lds_read16<W_OFF + W_BYTES>(a); break;
                case 2: fw[set][j] = lds_read16<W_OFF + 2 * W_BYTES>(a); break;
                default: fw[set][j] = lds_read16<W_OFF + 3 * W_BYTES>(a); break;
            }
        }
    };
    static_assert(MT == 2 && (NT == 1 || NT == 2), "a fragment group is MT + NT reads: the counted lgkmcnt waits below");
    auto wait_frags = [&](int set, int groups_in_flight) {  // group `set` has landed; the 2 / 1 / 0 groups read after it may still be in flight
        if constexpr (NT == 2) {
            if (groups_in_flight == 2) HPE_LGKM_WAIT4(8, fa[set][0], fa[set][1], fw[set][0], fw[set][1]);
            else if (groups_in_flight == 1) HPE_LGKM_WAIT4(4, fa[set][0], fa[set][1], fw[set][0], fw[set][1]);
            else HPE_LGKM_WAIT4(0, fa[set][0], fa[set][1], fw[set][0], fw[set][1]);
        } else {
            if (groups_in_flight == 2) HPE_LGKM_WAIT3(6, fa[set][0], fa[set][1], fw[set][0]);
            else if (groups_in_flight == 1) HPE_LGKM_WAIT3(3, fa[set][0], fa[set][1], fw[set][0]);
            else HPE_LGKM_WAIT3(0, fa[set][0], fa[set][1], fw[set][0]);
        }
    };

    // ---- prologue.  Issue ORDER (what the counted waits rely on): image 0 (API), W(0) .. W(2); then in the middle of step s: W(s + 3),
    //      then -- taps 0 .. API-1 of every slab but the last -- one piece of the next image.
#pragma unroll
    for (int i = 0; i < API; ++i)
        if (!ZERO_IN_TAIL || wave + NW * i < NPIECE) issue_a_piece(0, i);  // (older than every weight slab: the counted waits are unaffected)
#pragma unroll
    for (int u = 0; u < NSW - 1; ++u) issue_w(0, u, u);
    wait_dma_leaving((NSW - 2) * WI);  // image 0 and W(0)
    lds_barrier();
    tap_addr(0, 0);
    load_frags(0, 0, 0);
    load_frags(1, 0, 1);

    // One input slab = 9 steps of 4 fragment groups.  The step's barrier sits between groups 0 and 1 and is about the NEXT step: "W(s + 1)
    // has landed everywhere, every wave is past step s - 1" -- so the fragments of step s + 1's first group are read (group 3 of step s)
    // while step s still multiplies, and the matrix pipe does not drain at the step boundary.  R0 = ring position of the slab's first step.
    auto slab_steps = [&](int cs, bool last, auto r0_tag) {
        constexpr int R0 = decltype(r0_tag)::value;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == 1 && !(last && tap == 8)) {
                    // younger than W(s + 1): W(s + 2) and the image pieces requested in the two steps before this one
                    const int na1 = (!A1 && tap - 1 >= 0 && tap - 1 < API) ? 1 : 0;
                    const int na2 = (!A1 && tap - 2 >= 0 && tap - 2 < API) ? 1 : 0;
                    const int steady = WI + na1 + na2;
                    const int tail = tap + 2 < 9 ? WI : 0;  // last slab: no image pieces; W(s + 2) exists while tap + 2 < 9
                    if (CS > 1 && !last) wait_dma_leaving(steady);
                    else wait_dma_leaving(tail);
                    asm volatile("s_barrier" ::: "memory");  // (no lgkmcnt drain: this wave's reads of older steps were waited for before their multiplies)
                    const int t3 = tap + NSW - 1;
                    if (t3 < 9) issue_w(cs, t3, (R0 + t3) % NSW);
                    else if (!last) issue_w(cs + 1, t3 - 9, (R0 + t3) % NSW);
                    if (CS > 1 && !A1 && !last && tap < API) issue_a_piece(cs + 1, tap);
                }
                // read the group two ahead -- (tap, g + 2), or group g - 2 of the next step -- then wait for the current one only
                constexpr bool LOOPED = CS > 4;  // Cin = 512: the slab groups are a run-time loop and `last` a run-time flag
                const int G = 4 * tap + g, cur = G % 3, nxt = (G + 2) % 3;
                if (g < 2) {
                    load_frags(nxt, (R0 + tap) % NSW, g + 2);
                    wait_frags(cur, 2);
                } else if (tap < 8) {
                    if (g == 2) tap_addr(cs, tap + 1);
                    load_frags(nxt, (R0 + tap + 1) % NSW, g - 2);
                    wait_frags(cur, 2);
                } else if (A1) {
                    // one image buffer: nothing of the next slab can be read yet -- the last two groups of the slab drain the pipeline
                    wait_frags(cur, g == 2 ? 1 : 0);
                } else if (LOOPED && R0 == NSW - 1) {
                    // The back edge of the slab-group loop (and, on the last pass, the end).  Nothing may cross it in flight: the register
                    // copies the compiler places on a back edge or a branch would copy fragments that have not landed (tools/isa_lint.py
                    // found exactly that).  So no branch here: the next groups are read unconditionally -- after the last slab those are
                    // reads of valid LDS that nobody uses -- and at the edge all three sets are waited for.
                    if (g == 2) tap_addr(cs + 1, 0);
                    load_frags(nxt, (R0 + tap + 1) % NSW, g - 2);
                    if (g == 2) wait_frags(cur, 2);
                    else HPE_LGKM_WAIT12(0, fa[0][0], fa[0][1], fw[0][0], fw[0][NT - 1], fa[1][0], fa[1][1], fw[1][0], fw[1][NT - 1], fa[2][0], fa[2][1], fw[2][0], fw[2][NT - 1]);
                } else if (last) {  // (a compile-time constant in the unrolled kernels): the last two groups of the kernel
                    wait_frags(cur, g == 2 ? 1 : 0);
                } else {
                    if (g == 2) tap_addr(cs + 1, 0);
                    load_frags(nxt, (R0 + tap + 1) % NSW, g - 2);
                    wait_frags(cur, 2);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
            }
        }
        if (A1 && CS > 1 && !last) {
            // slab boundary with one image buffer: everyone is done reading image cs -> request image cs + 1 -> it has landed everywhere
            // (the wait also covers the weight slabs in flight, which are L2 hits) -> restart the fragment pipeline.  Nothing is in flight
            // when the slab function returns (it may be the body of a run-time loop).
            wait_dma_leaving(2 * WI);  // (at most two weight slabs are in flight here: no wait in effect -- named for tools/isa_lint.py, this barrier publishes no DMA data)
            lds_barrier();
#pragma unroll
            for (int i = 0; i < API; ++i) issue_a_piece(cs + 1, i);
            wait_dma_leaving(0);
            lds_barrier();
            tap_addr(cs + 1, 0);
            load_frags(0, (R0 + 9) % NSW, 0);
            load_frags(1, (R0 + 9) % NSW, 1);
            wait_frags(0, 1);
            wait_frags(1, 0);
        }
    };
    // 9 = 1 mod 4: slab cs starts at ring position cs % 4
    for (int g4 = 0; g4 < (CS + 3) / 4; ++g4) {
        const int c0 = 4 * g4;
        slab_steps(c0, c0 == CS - 1, std::integral_constant<int, 0>{});
        if constexpr (CS > 1) slab_steps(c0 + 1, c0 + 1 == CS - 1, std::integral_constant<int, 1>{});
        if constexpr (CS > 2) {
            slab_steps(c0 + 2, c0 + 2 == CS - 1, std::integral_constant<int, 2>{});
            slab_steps(c0 + 3, c0 + 3 == CS - 1, std::integral_constant<int, 3>{});
        }
    }

    // ---- epilogue: y = bf16(relu(acc * scale + shift)) -> LDS image of the output tile (slab layout) -> 16-B row stores
    wait_dma_leaving(0);  // (the queue is empty here; spelled out for tools/isa_lint.py, which cannot see that the last slab requests nothing)
    lds_barrier();        // every wave is out of the last step: the images and the ring are free
    {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nl0 = wn * (BN / 2) + j * 32;
            cfloat_p scp = (cfloat_p)(p.scale + n0 + nl0);
            cfloat_p shp = (cfloat_p)(p.shift + n0 + nl0);
            const int kc0 = (nl0 & 63) >> 3;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int ml = wm * 64 + i * 32 + (lane & 31);
                unsigned char* Us = lds + (nl0 >> 6) * SLAB + ml * 128;
                const int ux = (ml >> 1) & 7;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sc_lo = *reinterpret_cast<cf32x4_p>(scp + 8 * g), sc_hi = *reinterpret_cast<cf32x4_p>(scp + 8 * g + 4);
                    const f32x4 sh_lo = *reinterpret_cast<cf32x4_p>(shp + 8 * g), sh_hi = *reinterpret_cast<cf32x4_p>(shp + 8 * g + 4);
                    bf16x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sc = hi ? sc_hi[k] : sc_lo[k];
                        const float sh = hi ? sh_hi[k] : sh_lo[k];
                        const float v = acc[i][j][4 * g + k] * sc + sh;
                        o[k] = (__bf16)(p.relu ? fmaxf(v, 0.f) : v);
                    }
                    *reinterpret_cast<bf16x4*>(Us + (((kc0 + g) ^ ux) << 4) + hi * 8) = o;
                }
            }
        }
    }
    lds_barrier();
    {
        constexpr int UPR = BN / 8;  // 16-B units per row
#pragma unroll
        for (int pass = 0; pass < (BM * UPR) / NTHR; ++pass) {
            const int idx = pass * NTHR + t;
            const int r = idx / UPR, u = idx - r * UPR;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(lds + (u >> 3) * SLAB + r * 128 + (((u & 7) ^ ((r >> 1) & 7)) << 4));
            const int m = m0 + r;
            if (m < p.M) *reinterpret_cast<bf16x8*>(p.y + (size_t)m * p.N + n0 + u * 8) = v;
        }
    }
}

template <int HW, int CIN, int BN, int BM, int NSW = 4, bool A1 = false>
hipError_t launch_halo3(Halo3Args p, hipStream_t st) {
    p.n_ntiles = p.N / BN;
    const int grid = ((p.M + BM - 1) / BM) * p.n_ntiles;
    hipLaunchKernelGGL((conv3_halo_bf16_kernel<HW, CIN, BN, BM, NSW, A1>), dim3(grid), dim3(BM * 2), 0, st, p);
    return hipGetLastError();
}

}  // namespace

// the 3x3 layers of ResNet-50 v1: (map, Cin = N) = (56, 64), (28, 128), (14, 256), (7, 512)
bool hpe_halo3_bf16_supported(int HW, int Cin, int N) {
    return (HW == 56 && Cin == 64 && N == 64) || (HW == 28 && Cin == 128 && N == 128) || (HW == 14 && Cin == 256 && N == 256) ||
           (HW == 7 && Cin == 512 && N == 512);
}

hipError_t hpe_launch_halo3_bf16(const Halo3Args& p, int HW, int Cin, hipStream_t st) {
    if (!hpe_halo3_bf16_supported(HW, Cin, p.N)) return hipErrorInvalidValue;
    if (p.M <= 0 || p.M % (HW * HW) != 0 || !p.x || !p.w || !p.y || !p.scale || !p.shift || p.ldw < 9 * Cin || (p.ldw % 8) != 0) return hipErrorInvalidValue;
    if ((((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.y) & 15) != 0) return hipErrorInvalidValue;
    // Map sizes (bit mask as halo3, Halo3Args::two) that run the two-workgroups-per-CU form: 64 output channels per workgroup and ONE
    // image buffer (72 KB).  The caller's default is the 28 x 28 maps only -- with two 64-channel slabs a tile is 18 steps, a quarter of
    // its time is prologue (first image) and epilogue (row stores), and a second resident workgroup fills that: 0.078 -> 0.063 ms per
    // layer.  On the 14 x 14 maps (36 steps) it measures the same, on the 7 x 7 maps (72 steps) 3 % slower: there the reload of the
    // single image buffer at every slab boundary costs what the overlap gains.
    const int two = p.two;
    if (HW == 28 && (two & 4)) return launch_halo3<28, 128, 64, 256, 4, true>(p, st);
    if (HW == 14 && (two & 2)) return launch_halo3<14, 256, 64, 256, 4, true>(p, st);
    if (HW == 7 && (two & 1)) return launch_halo3<7, 512, 64, 256, 4, true>(p, st);
    switch (HW) {
        case 56: return launch_halo3<56, 64, 64, 256>(p, st);
        case 28: return launch_halo3<28, 128, 128, 256>(p, st);
        case 14: return launch_halo3<14, 256, 128, 256>(p, st);
        default: return launch_halo3<7, 512, 128, 256>(p, st);
    }
}
