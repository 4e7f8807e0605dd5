// conv_gemm_bf16.hip -- bf16 implicit-GEMM convolution on the gfx950 matrix cores (BASELINE config 4: bf16 encoder,
// fp32 accumulate, fp32 regressor + SMPL).  Same structure as the fp32 LDS-DMA kernel of conv_gemm.hip:
//   Y[m][n] = act((sum_k A[m][k] * Wt[n][k]) * scale[n] + shift[n] + R[m][n]),  A / Wt / R / Y in bf16, sum in fp32.
// A k-slab is 64 bf16 = one 128-B line per row, so the byte geometry of the staging (8 x 16-B chunks per row, 1-KiB
// global_load_lds_dwordx4 pieces, source-side XOR swizzle) is identical to the fp32 kernel; one ds_read_b128 per lane is
// exactly one operand of v_mfma_f32_32x32x16_bf16 (lane l: row l&31, k = 8*(l>>5)..+7 of the 16-deep step).
// At 16x the fp32 matrix rate the encoder is HBM / L2 bound in bf16: everything here is about moving 128-B lines.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpe_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#include "bf16_rows.h"

namespace {

struct SlabB {
    int off, tap, cs;
};

template <int MODE>
__device__ __forceinline__ void slab_advance_b(const GemmArgs& p, SlabB& sp) {
    if (MODE == GEMM_DENSE || MODE == GEMM_STRIDED || MODE == GEMM_DUAL) {
        sp.off += BKE;
    } else if (MODE == GEMM_CONV3) {
        sp.cs += 1;
        sp.off += BKE;
        if (sp.cs == p.cin_slabs) {
            sp.cs = 0;
            sp.tap += 1;
            const int kh = sp.tap / 3;
            sp.off = ((kh - 1) * p.Wi + (sp.tap - kh * 3 - 1)) * p.Cin;
        }
    } else {
        sp.off += 2 * p.Wi * 4;
    }
}

// NS = LDS ring depth: NS - 1 slabs in flight behind counted vmcnt waits and a raw s_barrier (one barrier per slab).  NS = 2 is
// the shipped configuration.  NS = 3 (HPE_NS_BF16=3, kept for the 128x128 and 256x128 tiles) was the experiment "is the slab DMA
// latency exposed at the bf16 matrix rate?" -- it is not: the deeper ring halves the workgroups per CU (96 KB of LDS) and LOSES
// 30-45 % on the 3x3 layers (res4*_branch2b 0.080 -> 0.117 ms; profiles/r02/bf16_ring_depth.txt): block-level overlap of two
// workgroups per CU already hides the DMA, the kernel sits at the ceiling of the one-barrier-per-slab structure.
template <int MODE, int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(64 * WM * WN, (NS > 2 || BM * BN >= 256 * 128) ? (WM * WN) / 4 : (WM * WN) / 2) void conv_gemm_bf16_dma_kernel(GemmArgs p) {
    constexpr int MT = BM / WM / 32;
    constexpr int NT = BN / WN / 32;
    constexpr int NW = WM * WN;
    constexpr int NTHR = 64 * NW;
    constexpr int AP = BM / (8 * NW);
    constexpr int BP = BN / (8 * NW);
    constexpr int EP = BN + 4;
    constexpr int BUF = (BM + BN) * RF;
    constexpr int LDS_FLOATS = (NS * BUF > BM * EP) ? NS * BUF : BM * EP;
    constexpr int NDMA = AP + BP;  // LDS-DMA instructions per wave and slab
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(AP >= 1 && BP >= 1, "tile too small for the wave count");
    static_assert(NS >= 2 && NS <= 4 && (NS - 2) * NDMA <= 63, "ring depth");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(p.x);
    const __bf16* __restrict__ W = reinterpret_cast<const __bf16*>(p.w);
    const __bf16* __restrict__ R = reinterpret_cast<const __bf16*>(p.res);
    __bf16* __restrict__ Y = reinterpret_cast<__bf16*>(p.y);

    const int total = p.n_mtiles * p.n_ntiles;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int mtile = swz / p.n_ntiles;
    const int ntile = swz - mtile * p.n_ntiles;
    const int m0 = mtile * BM;
    const int n0 = ntile * BN;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    const int drow = lane >> 3;
    RowB arow[AP];
    int arow2[MODE == GEMM_DUAL ? AP : 1];  // GEMM_DUAL: the same rows in the second (strided) source
    const __bf16* X2 = reinterpret_cast<const __bf16*>(p.x2);
    const __bf16* wsrc[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int r = (NW * i + wave) * 8 + drow;
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        arow[i] = make_row_b<MODE>(p, m0 + r, lc);
        if (MODE == GEMM_DUAL) arow2[i] = make_row_b<GEMM_STRIDED>(p, m0 + r, lc).base;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int r = (NW * i + wave) * 8 + drow;
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[i] = W + (size_t)(n0 + r) * p.ldw + lc * 8;
    }

    int a_row[MT], b_row[NT], a_x[MT], b_x[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = (wm * MT + i) * 32 + (lane & 31);
        a_row[i] = r * RF;
        a_x[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = (wn * NT + j) * 32 + (lane & 31);
        b_row[j] = (BM + r) * RF;
        b_x[j] = (r >> 1) & 7;
    }
    const int hi = lane >> 5;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Residual rows of the epilogue, fetched NOW (tiles with <= 4 row passes per thread: the expand layers' 128 x 64 tiles): their HBM
    // latency runs beside the operand DMA instead of after the LDS transpose (conv_gemm.hip does the same in fp32).
    constexpr int R_TPR = BN / 8, R_RPP = NTHR / R_TPR, R_NPASS = BM / R_RPP;
    constexpr bool R_PRE = R_NPASS <= 4;
    bf16x8 rpre[R_PRE ? R_NPASS : 1];
    bool r_pre = false;
    if constexpr (R_PRE) {
        const int rr_ = t / R_TPR;
        const int n = n0 + (t - rr_ * R_TPR) * 8;
        r_pre = R != nullptr && p.res_prefetch && (n + 7) < p.N;
        if (r_pre) {
#pragma unroll
            for (int pass = 0; pass < R_NPASS; ++pass) {
                const int m = m0 + pass * R_RPP + rr_;
                rpre[pass] = *reinterpret_cast<const bf16x8*>(R + (size_t)(m < p.M ? m : p.M - 1) * p.ldres + n);
            }
        }
    }

    const int S = p.K / BKE;
    SlabB sp;
    sp.tap = 0;
    sp.cs = 0;
    sp.off = (MODE == GEMM_CONV3) ? (-p.Wi - 1) * p.Cin : 0;

    auto issue_dma = [&](int slab, int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const void* src;
            if (MODE == GEMM_CONV3) {
                const bool ok = (arow[i].mask >> sp.tap) & 1u;
                src = ok ? static_cast<const void*>(X + (arow[i].base + sp.off)) : static_cast<const void*>(p.zero);
            } else if (MODE == GEMM_DUAL) {
                src = slab < p.k1_slabs ? X + (arow[i].base + slab * BKE) : X2 + (arow2[i] + (slab - p.k1_slabs) * BKE);
            } else {
                src = X + (arow[i].base + sp.off);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf + (NW * i + wave) * 8 * RF), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + slab * BKE),
                                             (__attribute__((address_space(3))) void*)(lds + buf + (BM + (NW * i + wave) * 8) * RF), 16, 0,
                                             0);
        }
    };

    // ---- ring of NS slab buffers: slabs s+1 .. s+NS-1 are in flight while slab s is multiplied
    {
        const int npre = (NS - 1 < S) ? NS - 1 : S;
        for (int i = 0; i < npre; ++i) {
            if (i > 0) slab_advance_b<MODE>(p, sp);
            issue_dma(i, i * BUF);
        }
    }
    // The ring position must be a compile-time constant in every copy of the loop body (u below, fully unrolled): with a
    // run-time buffer offset hipcc can neither prove the 16-B alignment of the fragment reads (they decay to ds_read2_b32) nor that
    // they do not alias the LDS-DMA it has just issued (it then waits vmcnt(0) in front of them, which serialises the DMA latency).
    for (int s0 = 0; s0 < S; s0 += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int s = s0 + u;
            if (s >= S) break;
            const int cur = u * BUF;                    // buffer of slab s
            const int nxt = ((u + NS - 1) % NS) * BUF;  // buffer of slab s + NS - 1 (== the one slab s - 1 has just vacated)
            // this wave's part of slab s has landed once at most min(NS-2, S-1-s) younger slabs are outstanding; the barrier
            // extends that to every wave's part, and tells that every wave is done reading the buffer of slab s-1
            const int ahead = (S - 1 - s < NS - 2) ? S - 1 - s : NS - 2;
            if (ahead <= 0) HPE_WAIT_VMCNT(0);
            else if (ahead == 1) HPE_WAIT_VMCNT(NDMA);
            else HPE_WAIT_VMCNT(2 * NDMA);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + NS - 1 < S) {
                slab_advance_b<MODE>(p, sp);
                issue_dma(s + NS - 1, nxt);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x8 fa[MT], fb[NT];
                const int lc = 2 * g + hi;
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&lds[cur + a_row[i] + ((lc ^ a_x[i]) << 2)]);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(&lds[cur + b_row[j] + ((lc ^ b_x[j]) << 2)]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // every wave is out of the last slab before the epilogue reuses the LDS

    // ---- epilogue: fp32 scale/shift, transpose through LDS, rows leave as 16 B (8 bf16) per lane
    {
        const int col_l = lane & 31;
        const int row_l = 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cl = (wn * NT + j) * 32 + col_l;
            const int n = n0 + cl;
            const bool n_ok = n < p.N;
            const float sc = n_ok ? p.scale[n] : 0.f;
            const float sh = n_ok ? p.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int rl = (wm * MT + i) * 32 + row_l;
#pragma unroll
                for (int e = 0; e < 16; ++e) lds[(rl + (e & 3) + 8 * (e >> 2)) * EP + cl] = acc[i][j][e] * sc + sh;
            }
        }
    }
    __syncthreads();
    {
        constexpr int TPR = BN / 8;
        constexpr int RPP = NTHR / TPR;
        const int r = t / TPR;
        const int c8 = (t - r * TPR) * 8;
        const int n = n0 + c8;
        const bool full = (n + 7) < p.N;
#pragma unroll
        for (int pass = 0; pass < BM / RPP; ++pass) {
            const int row = pass * RPP + r;
            const int m = m0 + row;
            if (m >= p.M || n >= p.N) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&lds[row * EP + c8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&lds[row * EP + c8 + 4]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (full) {
                if (R_PRE && r_pre) {
                    const bf16x8 rv = rpre[pass < (R_PRE ? R_NPASS : 1) ? pass : 0];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] += (float)rv[u];
                } else if (R) {
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

template <int MODE, int BM, int BN, int WM, int WN, int NS>
hipError_t launch_cfg_b(GemmArgs& p, hipStream_t st) {
    p.n_mtiles = (p.M + BM - 1) / BM;
    p.n_ntiles = (p.N + BN - 1) / BN;
    static const int res_prefetch = [] {
        const char* e = getenv("HPE_RES_PREFETCH");
        return e ? atoi(e) : 1;
    }();
    p.res_prefetch = res_prefetch;
    hipLaunchKernelGGL((conv_gemm_bf16_dma_kernel<MODE, BM, BN, WM, WN, NS>), dim3(p.n_mtiles * p.n_ntiles), dim3(64 * WM * WN), 0, st, p);
    return hipGetLastError();
}

// ring depth: 2 everywhere; 3 only on request (HPE_NS_BF16=3) for the two tiles it was measured on
template <int MODE, int BM, int BN, int WM, int WN>
hipError_t launch_ns(GemmArgs& p, int ns, hipStream_t st) {
    if (ns >= 3) return launch_cfg_b<MODE, BM, BN, WM, WN, 3>(p, st);
    return launch_cfg_b<MODE, BM, BN, WM, WN, 2>(p, st);
}

template <int MODE>
hipError_t launch_mode_b(GemmArgs& p, int tile, int ns, hipStream_t st) {
    switch (tile) {
        case TILE_128x128: return launch_ns<MODE, 128, 128, 2, 2>(p, ns, st);
        case TILE_128x64: return launch_cfg_b<MODE, 128, 64, 2, 2, 2>(p, st);
        case TILE_64x64: return launch_cfg_b<MODE, 64, 64, 2, 2, 2>(p, st);
        case TILE_64x128: return launch_cfg_b<MODE, 64, 128, 2, 2, 2>(p, st);
        case TILE_128x128_W8: return launch_cfg_b<MODE, 128, 128, 2, 4, 2>(p, st);
        case TILE_128x64_W8: return launch_cfg_b<MODE, 128, 64, 4, 2, 2>(p, st);
        case TILE_256x128_W8: return launch_ns<MODE, 256, 128, 4, 2>(p, ns, st);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------- glue ops in bf16
// img [B,H,W,3] fp32 -> out [B,Hp,Wp,4] bf16 (zero border, 4th channel 0); 8 B per thread
__global__ void pad_input_bf16_kernel(const float* __restrict__ img, uint2* __restrict__ out, int B, int H, int W, int Hp, int Wp) {
    const long total = (long)B * Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const long r = i / Wp;
        const int yp = (int)(r % Hp);
        const int b = (int)(r / Hp);
        const int x = xp - 3, y = yp - 3;
        __bf16 v[4] = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const float* s = img + (((long)b * H + y) * W + x) * 3;
            v[0] = (__bf16)s[0];
            v[1] = (__bf16)s[1];
            v[2] = (__bf16)s[2];
        }
        uint2 o;
        __builtin_memcpy(&o, v, 8);
        out[i] = o;
    }
}

__global__ void maxpool3x3s2_bf16_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ y, int B, int H, int C8) {
    const int Ho = H / 2;
    const long total = (long)B * Ho * Ho * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        long r = i / C8;
        const int wo = (int)(r % Ho);
        r /= Ho;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[u] = 0.f;  // zero padding takes part in the max; inputs are post-ReLU (>= 0)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * ho - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * wo - 1 + dx;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)H) {
                    const bf16x8 v = x[(((long)b * H + yy) * H + xx) * C8 + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) m[u] = fmaxf(m[u], (float)v[u]);
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = (__bf16)m[u];
        y[i] = o;
    }
}

__global__ void avgpool_bf16_kernel(const bf16x8* __restrict__ x, float* __restrict__ y, int B, int HW, int C8, int ldy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C8) return;
    const int c = i % C8;
    const int b = i / C8;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    const bf16x8* p = x + (long)b * HW * C8 + c;
    for (int k = 0; k < HW; ++k) {
        const bf16x8 v = p[(long)k * C8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += (float)v[u];
    }
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int u = 0; u < 8; ++u) y[(long)b * ldy + c * 8 + u] = s[u] * inv;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (__bf16)x[i];
}

__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (float)x[i];
}

inline int grid_for(long total, int block, int cap = 2048) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

hipError_t hpe_launch_gemm_bf16(GemmArgs p, int mode, int tile, int ns, hipStream_t st) {
    if (tile == TILE_P8_256x256) return hpe_launch_gemm_bf16_p8(p, mode, st);
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BKE) != 0 || (p.ldw % 8) != 0 || p.ldw < p.K) return hipErrorInvalidValue;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift || !p.zero) return hipErrorInvalidValue;
    if ((p.ldy % 8) != 0 || ((uintptr_t)p.y & 15) != 0) return hipErrorInvalidValue;
    if (p.res && ((p.ldres % 8) != 0 || ((uintptr_t)p.res & 15) != 0)) return hipErrorInvalidValue;
    if (((uintptr_t)p.x & 15) != 0 || ((uintptr_t)p.w & 15) != 0) return hipErrorInvalidValue;
    const int bn = (tile == TILE_128x128 || tile == TILE_64x128 || tile == TILE_128x128_W8 || tile == TILE_256x128_W8) ? 128 : 64;
    if (((p.N + bn - 1) / bn) * bn > p.w_rows) return hipErrorInvalidValue;
    switch (mode) {
        case GEMM_DENSE:
            if (p.lda < p.K || (p.lda % 8) != 0) return hipErrorInvalidValue;
            return launch_mode_b<GEMM_DENSE>(p, tile, ns, st);
        case GEMM_STRIDED:
            if (p.Cin != p.K || (p.Cin % 8) != 0) return hipErrorInvalidValue;
            if ((p.Ho - 1) * p.stride >= p.Hi || (p.Wo - 1) * p.stride >= p.Wi) return hipErrorInvalidValue;
            return launch_mode_b<GEMM_STRIDED>(p, tile, ns, st);
        case GEMM_CONV3:
            if ((p.Cin % BKE) != 0 || p.K != 9 * p.Cin || p.cin_slabs != p.Cin / BKE || p.Ho != p.Hi || p.Wo != p.Wi)
                return hipErrorInvalidValue;
            return launch_mode_b<GEMM_CONV3>(p, tile, ns, st);
        case GEMM_STEM:
            if (p.K != 4 * BKE || p.Hi < 2 * (p.Ho - 1) + 8 || p.Wi < 2 * (p.Wo - 1) + 8) return hipErrorInvalidValue;
            return launch_mode_b<GEMM_STEM>(p, tile, ns, st);
        case GEMM_DUAL:
            if (!p.x2 || ((uintptr_t)p.x2 & 15) != 0 || p.k1_slabs < 1 || p.k1_slabs * BKE >= p.K || p.lda < p.k1_slabs * BKE || (p.lda % 8) != 0)
                return hipErrorInvalidValue;
            if (p.Cin != p.K - p.k1_slabs * BKE || (p.Cin % 8) != 0 || p.M != (p.M / (p.Ho * p.Wo)) * p.Ho * p.Wo) return hipErrorInvalidValue;
            if ((p.Ho - 1) * p.stride >= p.Hi || (p.Wo - 1) * p.stride >= p.Wi) return hipErrorInvalidValue;
            return launch_mode_b<GEMM_DUAL>(p, tile, ns, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t hpe_launch_f32_to_bf16(const float* x, void* y, long n, hipStream_t st) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, x, reinterpret_cast<__bf16*>(y), n);
    return hipGetLastError();
}

hipError_t hpe_launch_bf16_to_f32(const void* x, float* y, long n, hipStream_t st) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(x), y, n);
    return hipGetLastError();
}

hipError_t hpe_launch_pad_input_bf16(const float* img, void* out, int B, int H, int W, int Hp, int Wp, hipStream_t st) {
    const long total = (long)B * Hp * Wp;
    hipLaunchKernelGGL(pad_input_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, img, reinterpret_cast<uint2*>(out), B, H, W,
                       Hp, Wp);
    return hipGetLastError();
}

hipError_t hpe_launch_maxpool_bf16(const void* x, void* y, int B, int H, int C, hipStream_t st) {
    if ((C % 8) != 0 || (H % 2) != 0) return hipErrorInvalidValue;
    const long total = (long)B * (H / 2) * (H / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, reinterpret_cast<const bf16x8*>(x),
                       reinterpret_cast<bf16x8*>(y), B, H, C / 8);
    return hipGetLastError();
}

hipError_t hpe_launch_avgpool_bf16(const void* x, float* y, int B, int HW, int C, int ldy, hipStream_t st) {
    if ((C % 8) != 0) return hipErrorInvalidValue;
    const int total = B * (C / 8);
    hipLaunchKernelGGL(avgpool_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, st, reinterpret_cast<const bf16x8*>(x), y, B, HW,
                       C / 8, ldy);
    return hipGetLastError();
}
