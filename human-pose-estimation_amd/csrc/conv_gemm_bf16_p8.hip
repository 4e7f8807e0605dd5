// conv_gemm_bf16_p8.hip -- the bf16 implicit-GEMM convolution with a 256 x 256 x 64 workgroup tile and a phase-interleaved main
// loop (round 3).  Same contract as conv_gemm_bf16_dma_kernel (conv_gemm_bf16.hip):
//   Y[m][n] = act((sum_k A[m][k] * Wt[n][k]) * scale[n] + shift[n] + R[m][n]),  A / Wt / R / Y in bf16, sum in fp32.
//
// Why another kernel: the 128 x 128 kernel (4 waves of 64 x 64) needs one barrier per 16 MFMAs and, per 64-deep k-slab and wave,
// 16 KB of fragment reads for 16 x 32 cycles of matrix work -- at two workgroups per CU that is the whole 128 B/clk of the LDS
// pipe for the reads alone, before the LDS-DMA writes: its 3x3 layers sit at 650-720 TF whatever the schedule.
// Here 8 waves (2 x 4) own 128 x 64 each: 24 KB of fragment reads per 32 MFMAs (0.75 B per matrix-core cycle instead of 1.0),
// 128 KB of LDS = two 64-KB k-tile buffers (one workgroup per CU), and the k-tile is cut into FOUR phases = the four 64 x 32
// quadrants of the wave's tile:
//     phase   reads LDS -> registers                multiplies                 issues LDS-DMA (2 instructions per lane)
//     P1      A0 (8 x b128), B0 (4)                 Q(0,0) = A0 x B0           B1 of k-tile t+1
//     P2      B1 (4)                                Q(0,1) = A0 x B1           A1 of k-tile t+1
//     P3      A1 (8; reuses A0's registers)         Q(1,1) = A1 x B1           A0 of k-tile t+2
//     P4      -                                     Q(1,0) = A1 x B0           B0 of k-tile t+2
// (A0 / A1: the wave's upper / lower 64 rows; B0 / B1: its two 32-column blocks; B0 stays in registers from P1 to P4.)
// Every phase is  { ds_reads; 2 x global_load_lds; s_waitcnt vmcnt(8); s_barrier; 8 MFMAs; s_barrier }  and the two wave
// groups (wr = 0 / 1, one wave of each per SIMD) run ONE BARRIER APART: while one group is on the matrix pipe the other
// issues its fragment reads and DMAs.  Half-tiles (128 rows of A or of Wt = 16 KB = 2 DMA rounds of the 512 lanes) are refilled
// two phases after their last fragment read (WAR, with the stagger) and are needed 6 to 9 phases after they were issued: with the
// issue order above "everything issued at least four phases ago has landed" is the one counted wait, vmcnt(8) after the phase's
// own two instructions -- the DMA queue is never drained inside the loop, and a staged half is first read one phase after the
// wait that retires it (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it").
// k-tiles past the end of K are "issued" from the zero page so that the counts stay uniform through the tail.
//
// Row -> wave assignment is chosen so that a half-tile is a whole number of DMA rounds: tile rows 128 i + 64 wr + 32 r + (0..31)
// (i = half, r = 32-row block) and Wt rows 128 j + 32 wc + (0..31).  LDS rows are 128 B with the source-side XOR swizzle of the
// other kernels (chunk c of row r holds logical chunk c ^ ((r >> 1) & 7)): every 16-lane ds_read_b128 group hits 16 distinct
// 16-B slots.
// Split-K (small grids: stage 4 / 5 launches have 98 / 50 tiles): slices store raw accumulators, a fix-up launch adds them in
// a fixed order and runs the epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpe_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#include "bf16_rows.h"

namespace {

constexpr int P8_BM = 256, P8_BN = 256, P8_THREADS = 512;
constexpr int P8_BUF = (P8_BM + P8_BN) * RF;  // floats per k-tile buffer (64 KB)
constexpr int P8_EP = P8_BN + 4;              // epilogue staging pitch (floats)

// ---- epilogue: four passes of 64 rows (both wr halves of one 32-row block) through LDS, rows leave as 16 B (8 bf16) per lane
__device__ __forceinline__ void p8_epilogue(const GemmArgs& p, float* lds, f32x16 (&acc)[4][2], int m0, int n0, int t, int lane, int wr, int wc) {
    const __bf16* __restrict__ R = reinterpret_cast<const __bf16*>(p.res);
    __bf16* __restrict__ Y = reinterpret_cast<__bf16*>(p.y);
    const int col_l = lane & 31;
    const int row_l = 4 * (lane >> 5);
    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 128 * j + 32 * wc + col_l;
        const bool ok = n < p.N;
        sc[j] = ok ? p.scale[n] : 0.f;
        sh[j] = ok ? p.shift[n] : 0.f;
    }
    constexpr int TPR = P8_BN / 8;          // threads per output row (32)
    constexpr int RPP = P8_THREADS / TPR;   // rows per sweep (16)
    const int er = t / TPR;
    const int c8 = (t - er * TPR) * 8;
    const int n = n0 + c8;
    const bool full = (n + 7) < p.N;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = pass >> 1, r = pass & 1;
        __syncthreads();  // the previous pass (or the main loop) is done with the staging area
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = 128 * j + 32 * wc + col_l;
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[(32 * wr + row_l + (e & 3) + 8 * (e >> 2)) * P8_EP + cl] = acc[2 * i + r][j][e] * sc[j] + sh[j];
        }
        __syncthreads();
#pragma unroll
        for (int sweep = 0; sweep < 64 / RPP; ++sweep) {
            const int s = sweep * RPP + er;  // staging row: 32 * wr' + local row
            const int m = m0 + 128 * i + 64 * (s >> 5) + 32 * r + (s & 31);
            if (m >= p.M || n >= p.N) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&lds[s * P8_EP + c8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&lds[s * P8_EP + c8 + 4]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (full) {
                if (R) {
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(R + (size_t)m * p.ldres + n);
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] += (float)rv[u];
                }
                bf16x8 o;
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = (__bf16)(p.relu ? fmaxf(v[u], 0.f) : v[u]);
                *reinterpret_cast<bf16x8*>(Y + (size_t)m * p.ldy + n) = o;
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (n + u < p.N) {
                        float o = v[u];
                        if (R) o += (float)R[(size_t)m * p.ldres + n + u];
                        if (p.relu) o = fmaxf(o, 0.f);
                        Y[(size_t)m * p.ldy + n + u] = (__bf16)o;
                    }
                }
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(P8_THREADS, 1) void conv_gemm_bf16_p8_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * P8_BUF];  // 128 KB: the ONLY LDS object of the kernel

    const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(p.x);
    const __bf16* __restrict__ X2 = reinterpret_cast<const __bf16*>(p.x2);
    const __bf16* __restrict__ W = reinterpret_cast<const __bf16*>(p.w);

    // ---- tile (XCD-aware bijective remap: the N tiles sharing an A row panel run back to back on one XCD) and k slice
    const int tiles = p.n_mtiles * p.n_ntiles;
    const int total = tiles * p.split_k;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int part = swz % p.split_k;
    const int tile = swz / p.split_k;
    const int mtile = tile / p.n_ntiles;
    const int ntile = tile - mtile * p.n_ntiles;
    const int m0 = mtile * P8_BM;
    const int n0 = ntile * P8_BN;
    const int S_all = p.K / BKE;
    const int per = (S_all + p.split_k - 1) / p.split_k;
    const int ks0 = part * per;
    const int ks1 = (ks0 + per < S_all) ? ks0 + per : S_all;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2;
    const int wc = wave & 3;

    // ---- LDS-DMA sources: round d (0..3) of a tile covers rows 64 d + 8 wave + (lane >> 3); the swizzle term of those rows does not
    //      depend on d, so one chunk index serves all rounds
    const int drow = wave * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((drow >> 1) & 7);
    RowB arow[4];
    int arow2[MODE == GEMM_DUAL ? 4 : 1];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        arow[d] = make_row_b<MODE>(p, m0 + 64 * d + drow, lc);
        if (MODE == GEMM_DUAL) arow2[d] = make_row_b<GEMM_STRIDED>(p, m0 + 64 * d + drow, lc).base;
    }
    const __bf16* wsrc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int nr = n0 + 64 * d + drow;
        if (nr >= p.w_rows) nr = p.w_rows - 1;  // tile columns past the packed weights: any valid row (their outputs are never stored)
        wsrc[d] = W + (size_t)nr * p.ldw + lc * 8;
    }
    const int cs_shift = __builtin_ctz((unsigned)(p.cin_slabs > 0 ? p.cin_slabs : 1));  // CONV3: channel slabs per tap (power of two)

    // half-tile h of A for k-tile kt into buffer `buf` (rounds 2h, 2h+1)
    auto issue_a = [&](int h, int kt, int buf) {
        const bool live = kt < ks1;
        int off = 0;
        unsigned tapbit = 1u;
        if (MODE == GEMM_CONV3) {
            const int tap = kt >> cs_shift;
            const int cs = kt - (tap << cs_shift);
            const int kh = tap / 3;
            off = ((kh - 1) * p.Wi + (tap - kh * 3 - 1)) * p.Cin + cs * BKE;
            tapbit = 1u << tap;
        } else {
            off = kt * BKE;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = 2 * h + u;
            const void* src;
            if (MODE == GEMM_CONV3) {
                src = (live && (arow[d].mask & tapbit)) ? static_cast<const void*>(X + (arow[d].base + off)) : static_cast<const void*>(p.zero);
            } else if (MODE == GEMM_DUAL) {
                const void* s1 = kt < p.k1_slabs ? static_cast<const void*>(X + (arow[d].base + kt * BKE))
                                                 : static_cast<const void*>(X2 + (arow2[d] + (kt - p.k1_slabs) * BKE));
                src = live ? s1 : static_cast<const void*>(p.zero);
            } else {
                src = live ? static_cast<const void*>(X + (arow[d].base + off)) : static_cast<const void*>(p.zero);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf * P8_BUF + (64 * d + wave * 8) * RF), 16, 0, 0);
        }
    };
    auto issue_b = [&](int h, int kt, int buf) {
        const bool live = kt < ks1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = 2 * h + u;
            const void* src = live ? static_cast<const void*>(wsrc[d] + kt * BKE) : static_cast<const void*>(p.zero);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf * P8_BUF + (P8_BM + 64 * d + wave * 8) * RF), 16, 0, 0);
        }
    };

    // ---- fragment read addresses (floats): row (lane & 31) of a 32-row block, logical chunk 2 ks + hi, XOR the row's swizzle term
    const int hi = lane >> 5;
    const int fx = (lane >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = (lane & 31) * RF + (((2 * ks + hi) ^ fx) << 2);
    const int a_base = 64 * wr * RF;            // + (128 i + 32 r) * RF
    const int b_base = (P8_BM + 32 * wc) * RF;  // + 128 j * RF

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    bf16x8 fa[2][4], fb0[4], fb1[4];

    auto load_a = [&](int i, int buf) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[r][ks] = *reinterpret_cast<const bf16x8*>(&lds[buf * P8_BUF + a_base + (128 * i + 32 * r) * RF + foff[ks]]);
    };
    auto load_b = [&](bf16x8 (&fb)[4], int j, int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(&lds[buf * P8_BUF + b_base + 128 * j * RF + foff[ks]]);
    };
    auto mma = [&](int i, int j, bf16x8 (&fb)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[2 * i + r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[r][ks], fb[ks], acc[2 * i + r][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // the two barriers of a phase with the counted wait in front of the first one
    auto sync_loaded = [&]() {
        HPE_WAIT_VMCNT(8);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        HPE_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_done = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: k-tile ks0 whole, A0 / B0 of k-tile ks0 + 1 (what phases P3 / P4 of "k-tile -1" would have issued)
    issue_a(0, ks0, 0);
    issue_b(0, ks0, 0);
    issue_b(1, ks0, 0);
    issue_a(1, ks0, 0);
    issue_a(0, ks0 + 1, 1);
    issue_b(0, ks0 + 1, 1);
    HPE_WAIT_VMCNT(8);  // A0, B0 of the first k-tile have landed (this wave's part; the barrier extends it to every wave's)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: the second wave group runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);

    // ---- main loop, two k-tiles per trip so that every LDS offset is a compile-time constant
    for (int kt0 = ks0; kt0 < ks1; kt0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kt = kt0 + u;
            if (kt >= ks1) break;
            const int buf = u;  // ks0-relative parity == u (kt0 - ks0 is even)
            // P1
            load_a(0, buf);
            load_b(fb0, 0, buf);
            issue_b(1, kt + 1, buf ^ 1);
            sync_loaded();
            mma(0, 0, fb0);
            sync_done();
            // P2
            load_b(fb1, 1, buf);
            issue_a(1, kt + 1, buf ^ 1);
            sync_loaded();
            mma(0, 1, fb1);
            sync_done();
            // P3
            load_a(1, buf);
            issue_a(0, kt + 2, buf);
            sync_loaded();
            mma(1, 1, fb1);
            sync_done();
            // P4
            issue_b(0, kt + 2, buf);
            sync_loaded();
            mma(1, 0, fb0);
            sync_done();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the second group's last barrier
    HPE_WAIT_VMCNT(0);  // the zero-page DMAs of the tail have landed before the epilogue reuses the LDS

    if (p.split_k > 1) {
        // split-K slice: raw accumulators -> workspace [tile][slice][block 8][register quad 4][thread 512] (16 B per lane, coalesced)
        f32x4* dst = reinterpret_cast<f32x4*>(p.partial) + (size_t)(tile * p.split_k + part) * (8 * 4 * P8_THREADS) + t;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    f32x4 v = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
                    dst[((i * 2 + j) * 4 + qd) * P8_THREADS] = v;
                }
        return;
    }
    p8_epilogue(p, lds, acc, m0, n0, t, lane, wr, wc);
}

// Reduces the K-slices of every tile (fixed order -> bitwise reproducible) and runs the normal epilogue.
__global__ __launch_bounds__(P8_THREADS, 1) void conv_gemm_bf16_p8_fixup_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[64 * P8_EP];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile = blockIdx.x;
    const int mtile = tile / p.n_ntiles;
    const int ntile = tile - mtile * p.n_ntiles;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.partial) + (size_t)tile * p.split_k * (8 * 4 * P8_THREADS) + t;
    for (int part = 0; part < p.split_k; ++part) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 v = src[(size_t)part * (8 * 4 * P8_THREADS) + ((i * 2 + j) * 4 + qd) * P8_THREADS];
                    acc[i][j][4 * qd] += v.x;
                    acc[i][j][4 * qd + 1] += v.y;
                    acc[i][j][4 * qd + 2] += v.z;
                    acc[i][j][4 * qd + 3] += v.w;
                }
    }
    p8_epilogue(p, lds, acc, mtile * P8_BM, ntile * P8_BN, t, lane, wave >> 2, wave & 3);
}

template <int MODE>
hipError_t launch_p8(GemmArgs& p, hipStream_t st) {
    p.n_mtiles = (p.M + P8_BM - 1) / P8_BM;
    p.n_ntiles = (p.N + P8_BN - 1) / P8_BN;
    const int tiles = p.n_mtiles * p.n_ntiles;
    const int S = p.K / BKE;
    // Split K when the tiles alone leave most of the 256 one-workgroup CUs idle: about one workgroup per CU, >= 4 k-tiles per slice
    int sk = 1;
    if (p.partial && tiles <= 160 && S >= 8) {
        sk = (256 + tiles - 1) / tiles;
        if (sk > S / 4) sk = S / 4;
        if (sk > 8) sk = 8;
        while (sk > 1 && (size_t)tiles * sk * P8_BM * P8_BN > p.partial_floats) --sk;
        // every slice must own at least one k-tile
        while (sk > 1 && ((S + sk - 1) / sk) * (sk - 1) >= S) --sk;
        if (sk < 1) sk = 1;
    }
    p.split_k = sk;
    hipLaunchKernelGGL((conv_gemm_bf16_p8_kernel<MODE>), dim3(tiles * sk), dim3(P8_THREADS), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || sk == 1) return e;
    hipLaunchKernelGGL(conv_gemm_bf16_p8_fixup_kernel, dim3(tiles), dim3(P8_THREADS), 0, st, p);
    return hipGetLastError();
}

}  // namespace

// Host-side contract as hpe_launch_gemm_bf16 (conv_gemm_bf16.hip), plus: N % 8 == 0 rows of 16 B, cin_slabs a power of two.
hipError_t hpe_launch_gemm_bf16_p8(GemmArgs p, int mode, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BKE) != 0 || (p.ldw % 8) != 0 || p.ldw < p.K || p.w_rows < 1) return hipErrorInvalidValue;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift || !p.zero) return hipErrorInvalidValue;
    if ((p.ldy % 8) != 0 || ((uintptr_t)p.y & 15) != 0) return hipErrorInvalidValue;
    if (p.res && ((p.ldres % 8) != 0 || ((uintptr_t)p.res & 15) != 0)) return hipErrorInvalidValue;
    if (((uintptr_t)p.x & 15) != 0 || ((uintptr_t)p.w & 15) != 0 || ((uintptr_t)p.zero & 15) != 0) return hipErrorInvalidValue;
    if (p.partial && ((uintptr_t)p.partial & 15) != 0) return hipErrorInvalidValue;
    switch (mode) {
        case GEMM_DENSE:
            if (p.lda < p.K || (p.lda % 8) != 0) return hipErrorInvalidValue;
            return launch_p8<GEMM_DENSE>(p, st);
        case GEMM_STRIDED:
            if (p.Cin != p.K || (p.Cin % 8) != 0) return hipErrorInvalidValue;
            if ((p.Ho - 1) * p.stride >= p.Hi || (p.Wo - 1) * p.stride >= p.Wi) return hipErrorInvalidValue;
            return launch_p8<GEMM_STRIDED>(p, st);
        case GEMM_CONV3:
            if ((p.Cin % BKE) != 0 || p.K != 9 * p.Cin || p.cin_slabs != p.Cin / BKE || p.Ho != p.Hi || p.Wo != p.Wi) return hipErrorInvalidValue;
            if ((p.cin_slabs & (p.cin_slabs - 1)) != 0) return hipErrorInvalidValue;
            return launch_p8<GEMM_CONV3>(p, st);
        case GEMM_DUAL:
            if (!p.x2 || ((uintptr_t)p.x2 & 15) != 0 || p.k1_slabs < 1 || p.k1_slabs * BKE >= p.K || p.lda < p.k1_slabs * BKE || (p.lda % 8) != 0)
                return hipErrorInvalidValue;
            if (p.Cin != p.K - p.k1_slabs * BKE || (p.Cin % 8) != 0 || p.M != (p.M / (p.Ho * p.Wo)) * p.Ho * p.Wo) return hipErrorInvalidValue;
            if ((p.Ho - 1) * p.stride >= p.Hi || (p.Wo - 1) * p.stride >= p.Wi) return hipErrorInvalidValue;
            return launch_p8<GEMM_DUAL>(p, st);
        default: return hipErrorInvalidValue;
    }
}
