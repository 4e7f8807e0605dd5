// bf16_rows.h -- per-lane source addressing of the bf16 implicit-GEMM kernels (conv_gemm_bf16.hip, conv_gemm_bf16_p8.hip):
// which 16-byte chunk of which activation row a lane feeds to its global_load_lds, per GEMM mode.
#pragma once
#include <hip/hip_runtime.h>

#include "hpe_internal.h"

#define BKE 64  // bf16 elements per k-slab (128 B)
#define RF 32   // floats per staged LDS row (128 B)

// s_waitcnt vmcnt(n), n <= 63, everything else unconstrained (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
#define HPE_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
// s_waitcnt lgkmcnt(0), vmcnt / expcnt unconstrained
#define HPE_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

namespace {

struct RowB {
    int base;  // element offset (bf16 elements) of this lane's 16-B chunk in slab 0
    unsigned mask;
};

template <int MODE>
__device__ __forceinline__ RowB make_row_b(const GemmArgs& p, int m, int lc) {
    RowB r;
    r.mask = 0x1ffu;
    if (m >= p.M) m = p.M - 1;
    if (MODE == GEMM_DENSE || MODE == GEMM_DUAL) {
        r.base = m * p.lda + lc * 8;
    } else {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw;
        const int rem = m - b * hw;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        if (MODE == GEMM_STRIDED) {
            r.base = ((b * p.Hi + ho * p.stride) * p.Wi + wo * p.stride) * p.Cin + lc * 8;
        } else if (MODE == GEMM_CONV3) {
            r.base = ((b * p.Hi + ho) * p.Wi + wo) * p.Cin + lc * 8;
            unsigned mk = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dh = tap / 3 - 1, dw = tap % 3 - 1;
                if ((unsigned)(ho + dh) < (unsigned)p.Hi && (unsigned)(wo + dw) < (unsigned)p.Wi) mk |= 1u << tap;
            }
            r.mask = mk;
        } else {
            // STEM: padded input [B,Hi,Wi,4] bf16; one slab = kernel rows (2s, 2s+1), each 8 px x 4 ch = 64 B
            r.base = ((b * p.Hi + 2 * ho + (lc >> 2)) * p.Wi + 2 * wo) * 4 + (lc & 3) * 8;
        }
    }
    return r;
}

}  // namespace
