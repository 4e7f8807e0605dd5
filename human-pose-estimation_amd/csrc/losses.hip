// losses.hip -- forward of the two reprojection losses the reference evaluates on the path's outputs
// (BASELINE config 5): kp_reprojection_loss (src/ops.py:35-47) and mesh_reprojection_loss ->
// bidirectional_dist -> find_nearest_neighbors (src/ops.py:60-137).
//
// The reference materialises a [P_i, 6890] fp32 distance matrix per image in a Python loop.  Here the
// nearest-neighbour search is a tiled brute force: four points per lane with running (min, argmin) in
// registers, the other point set streamed through LDS (broadcast ds_read_b128), no distance matrix in memory.
// The squared distance is evaluated in the reference's expanded form ((-2 a.b) + |a|^2) + |b|^2 in fp32 and
// ties keep the lowest index (tf.argmin), so the chosen neighbours follow the reference's choice.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>

#include "hpe_internal.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = sum vis*|gt-pred|, out[1] = #nonzero broadcast weights (2 per visible kp), out[2] = safe ratio
__global__ __launch_bounds__(256) void kp_loss_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int n,
                                                      float* __restrict__ out) {
    __shared__ float red[4];
    float num = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gx = gt[i * 3], gy = gt[i * 3 + 1], vis = gt[i * 3 + 2];
        const float px = pred[i * 2], py = pred[i * 2 + 1];
        num += fabsf(px - gx) * vis + fabsf(py - gy) * vis;
        cnt += (vis != 0.f) ? 2.f : 0.f;
    }
    const float tn = block_sum_256(num, red);
    const float tc = block_sum_256(cnt, red);
    if (threadIdx.x == 0) {
        out[0] = tn;
        out[1] = tc;
        out[2] = tc > 0.f ? tn / tc : 0.f;
    }
}

// ordered compaction of the silhouette pixels of image b: pts[b][i] = (x = col, y = row), row-major order
// (tf.where order; src/trainer.py:291, src/ops.py:123-125).  Each of the 4 waves owns a contiguous quarter of the image:
// pass 1 counts (ballot + popcount), the 4 totals give each wave its output offset, pass 2 writes -- coalesced reads, no
// barrier inside the loops.
__global__ __launch_bounds__(256) void sil_compact_kernel(const float* __restrict__ seg, int HW, int W, float* __restrict__ pts,
                                                          int* __restrict__ counts) {
    __shared__ int wtot[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int quarter = (HW + 3) / 4;
    const int lo = wave * quarter, hi = min(lo + quarter, HW);
    const float* s = seg + (size_t)b * HW;
    int c = 0;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool on = i < hi && s[i] > 0.f;
        c += __popcll(__ballot(on));
    }
    if (lane == 0) wtot[wave] = c;
    __syncthreads();
    int pos = 0;
    for (int w = 0; w < wave; ++w) pos += wtot[w];
    float* o = pts + (size_t)b * HW * 2;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool on = i < hi && s[i] > 0.f;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int k = pos + __popcll(m & ((1ull << lane) - 1ull));
            const int y = i / W;
            *reinterpret_cast<float2*>(&o[2 * k]) = make_float2((float)(i - y * W), (float)y);
        }
        pos += __popcll(m);
    }
    if (t == 0) counts[b] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}

// direction A -> B: every silhouette point a finds its nearest mesh vertex, contributes |a - b*|_1.
// grid (ceil(HW/1024), B); 4 points per lane (register tile), B's P points live in dynamic LDS as (x, y, |b|^2, 0)
// so one broadcast ds_read_b128 serves 4 pair evaluations.
#define NN_PT 4
#define NN_BT 1024  // mesh vertices staged per LDS tile (16 KB) -- small enough for 8 waves/SIMD
__global__ __launch_bounds__(256) void nn_a2b_kernel(const float* __restrict__ pts, const int* __restrict__ counts,
                                                     const float* __restrict__ v2d, int HW, int P, float* __restrict__ partial,
                                                     int nblk) {
    __shared__ __attribute__((aligned(16))) float sB[NN_BT * 4];  // (x, y, |b|^2, 0)
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int cnt = counts[b];
    const int base = blockIdx.x * 256 * NN_PT;
    if (base >= cnt) {
        if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = 0.f;
        return;
    }
    const float* Bp = v2d + (size_t)b * P * 2;
    float ax[NN_PT], ay[NN_PT], aa[NN_PT], best[NN_PT], cx[NN_PT], cy[NN_PT];
#pragma unroll
    for (int u = 0; u < NN_PT; ++u) {
        const int idx = min(base + u * 256 + (int)threadIdx.x, cnt - 1);
        ax[u] = pts[((size_t)b * HW + idx) * 2];
        ay[u] = pts[((size_t)b * HW + idx) * 2 + 1];
        aa[u] = ax[u] * ax[u] + ay[u] * ay[u];
        best[u] = 3.4e38f;
        cx[u] = cy[u] = 0.f;
    }
    for (int p0 = 0; p0 < P; p0 += NN_BT) {
        const int n = min(NN_BT, P - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const float bx = Bp[2 * (p0 + i)], by = Bp[2 * (p0 + i) + 1];
            *reinterpret_cast<float4*>(&sB[4 * i]) = make_float4(bx, by, bx * bx + by * by, 0.f);
        }
        __syncthreads();
#pragma unroll 2
        for (int p = 0; p < n; ++p) {
            const float4 q = *reinterpret_cast<const float4*>(&sB[4 * p]);
#pragma unroll
            for (int u = 0; u < NN_PT; ++u) {
                const float d = (-2.0f * (ax[u] * q.x + ay[u] * q.y) + aa[u]) + q.z;
                const bool lt = d < best[u];  // strict: the lowest index keeps a tie (tf.argmin)
                best[u] = lt ? d : best[u];
                cx[u] = lt ? q.x : cx[u];
                cy[u] = lt ? q.y : cy[u];
            }
        }
    }
    float contrib = 0.f;
#pragma unroll
    for (int u = 0; u < NN_PT; ++u) {
        const int idx = base + u * 256 + (int)threadIdx.x;
        if (idx < cnt) contrib += fabsf(ax[u] - cx[u]) + fabsf(ay[u] - cy[u]);
    }
    const float s = block_sum_256(contrib, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = s;
}

// A -> B on the matrix cores.  The reference forms D = -2 A B^T + |A|^2 + |B|^2 with a matmul (src/ops.py:60-71) and takes
// argmin over B.  |A|^2 is constant per point, so the search minimises  |b|^2 - 2 a.b,  which is exactly one
// v_mfma_f32_32x32x2_f32 per 32 vertices x 32 points: A operand (-2 bx | -2 by), B operand (ax | ay), C operand |b|^2.
// Vertices are the MFMA's M side and points its N side, so a lane owns ONE point (column n = lane & 31) and receives 16
// vertices of it per MFMA in its accumulator registers.  Per MFMA the VALU only reduces those 16 values with v_min3 and
// keeps (best value, id of the 16-vertex group it came from): ~11 instructions per 1024 pairs instead of ~7 per pair in the
// VALU-only kernel above, which ran at 95 % of the fp32 issue rate.  The winning vertex inside the winning group is
// recovered once per point at the end (16 candidates, ascending index, strict <), and groups are visited in ascending
// index with strict <, so equal distances resolve to the lowest index as tf.argmin does.  The two lane halves see
// different vertex subsets of the same point and are merged once at the end.
// In exact arithmetic the choice is the reference's; in fp32 it can differ between candidates whose distances agree to
// rounding (the reference's own expanded form has that noise: entries ~1e5 px^2 carry ~8e-3 px^2 of rounding).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// One element of v_mfma_f32_32x32x2_f32, bit for bit: the matrix core accumulates k = 0, then k = 1, each as a fused
// multiply-add (tools/probes/mfma_f32_rounding.hip: 0 mismatches in 4.2 M pairs; the other order and a single rounding both
// differ in ~30 % of them).  The searches recover their winner by recomputing candidates with this and comparing for equality.
// |b|^2 with two rounded products and a rounded sum (tf.reduce_sum(B * B, 1)); written out so that every kernel that compares
// matrix-core values for equality feeds the same bits whatever -ffp-contract does to `x * x + y * y`
// (__fmul_rn / __fadd_rn are plain operators in this HIP and contract like them)
__device__ __forceinline__ float norm2(float x, float y) {
#pragma clang fp contract(off)
    const float xx = x * x;
    const float yy = y * y;
    return xx + yy;
}
__device__ __forceinline__ float mfma_k2_value(float a0, float b0, float a1, float b1, float c) { return fmaf(a1, b1, fmaf(a0, b0, c)); }
#define NN_PG 8  // point groups of 32 per wave -> 1024 points per 256-thread block
// (the explicit waves-per-SIMD bound makes hipcc keep the MFMA results in VGPRs; without it they land in AGPRs and every value
// costs an extra v_accvgpr_read before the VALU can touch it)
__global__ __launch_bounds__(256, 3) void nn_a2b_mfma_kernel(const float* __restrict__ pts, const int* __restrict__ counts,
                                                          const float* __restrict__ v2d, int HW, int P, float* __restrict__ partial,
                                                          int nblk, const int* __restrict__ only_flagged,
                                                          unsigned long long* __restrict__ mfma_count) {
    if (only_flagged && only_flagged[blockIdx.y] == 0) return;  // image already done by the cell-grid search
    __shared__ __attribute__((aligned(16))) float sX[NN_BT];  // -2 bx
    __shared__ __attribute__((aligned(16))) float sY[NN_BT];  // -2 by
    __shared__ __attribute__((aligned(16))) float sN[NN_BT];  // |b|^2 (+inf for padding vertices: never selected)
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int cnt = counts[b];
    const int base = blockIdx.x * 256 * NN_PT;
    if (base >= cnt) {
        if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = 0.f;
        return;
    }
    const float* Bp = v2d + (size_t)b * P * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    float px[NN_PG], py[NN_PG], best[NN_PG];
    int bgrp[NN_PG];  // id of the best 16-vertex group: 2 * (global 32-vertex group) + half
#pragma unroll
    for (int g = 0; g < NN_PG; ++g) {
        const int idx = min(base + wave * (32 * NN_PG) + g * 32 + l31, cnt - 1);
        px[g] = pts[((size_t)b * HW + idx) * 2];
        py[g] = pts[((size_t)b * HW + idx) * 2 + 1];
        best[g] = __builtin_inff();
        bgrp[g] = hi;
    }
    for (int p0 = 0; p0 < P; p0 += NN_BT) {
        const int n = min(NN_BT, P - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < NN_BT; i += 256) {
            float bx = 0.f, by = 0.f, bn = __builtin_inff();
            if (i < n) {
                bx = Bp[2 * (p0 + i)];
                by = Bp[2 * (p0 + i) + 1];
                bn = norm2(bx, by);
            }
            sX[i] = -2.0f * bx;
            sY[i] = -2.0f * by;
            sN[i] = bn;
        }
        __syncthreads();
        const int ngroups = (n + 31) >> 5;
        for (int vg = 0; vg < ngroups; ++vg) {
            const float a = hi ? sY[vg * 32 + l31] : sX[vg * 32 + l31];
            f32x16_t qz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q = *reinterpret_cast<const float4*>(&sN[vg * 32 + 8 * j + 4 * hi]);
                qz[4 * j] = q.x;
                qz[4 * j + 1] = q.y;
                qz[4 * j + 2] = q.z;
                qz[4 * j + 3] = q.w;
            }
            const int gid = 2 * ((p0 >> 5) + vg) + hi;
            // four MFMAs in flight, then their reductions: the VALU work of one batch covers the MFMA latency of the next
#pragma unroll
            for (int g0 = 0; g0 < NN_PG; g0 += 4) {
                f32x16_t d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, hi ? py[g0 + u] : px[g0 + u], qz, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int g = g0 + u;
                    const float m0 = fminf(fminf(d[u][0], d[u][1]), d[u][2]);
                    const float m1 = fminf(fminf(d[u][3], d[u][4]), d[u][5]);
                    const float m2 = fminf(fminf(d[u][6], d[u][7]), d[u][8]);
                    const float m3 = fminf(fminf(d[u][9], d[u][10]), d[u][11]);
                    const float m4 = fminf(fminf(d[u][12], d[u][13]), d[u][14]);
                    const float m5 = fminf(fminf(m0, m1), d[u][15]);
                    const float m6 = fminf(fminf(m2, m3), m4);
                    // the running best rides in the last min3 of the tree (8 min3 + compare + one select per MFMA instead of 8 + min +
                    // compare + two selects: the VALU chain behind each MFMA is what this loop waits for -- timing ablations: 1.38 ms
                    // per launch as is, 1.13 with a 4-value reduction, 0.90 with none)
                    const float nb = fminf(fminf(m5, m6), best[g]);
                    const bool lt = nb < best[g];  // strict: the first (lowest-index) group keeps a tie
                    best[g] = nb;
                    bgrp[g] = lt ? gid : bgrp[g];
                }
            }
        }
    }
    // diagnostics counter: one atomic per workgroup (its four waves issue the same number of MFMAs)
    if (mfma_count && threadIdx.x == 0) atomicAdd(mfma_count, (unsigned long long)(((P + 31) >> 5) * NN_PG) * 4ull);
    float contrib = 0.f;
#pragma unroll
    for (int g = 0; g < NN_PG; ++g) {
        // each lane half recovers the vertex inside its winning 16-vertex group (ascending index, strict <), then the halves
        // merge by (value, vertex index): the two halves of a 32-vertex group interleave in index
        const int idx = base + wave * (32 * NN_PG) + g * 32 + l31;
        const int v0 = (bgrp[g] >> 1) * 32 + 4 * (bgrp[g] & 1);
        float bd = __builtin_inff(), vx = 0.f, vy = 0.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int vi = v0 + (e & 3) + 8 * (e >> 2);  // ascending with e
            if (vi < P) {
                const float bx = Bp[2 * vi], by = Bp[2 * vi + 1];
                const float dd = mfma_k2_value(-2.0f * bx, px[g], -2.0f * by, py[g], norm2(bx, by));
                if (dd < bd) {
                    bd = dd;
                    bi = vi;
                    vx = bx;
                    vy = by;
                }
            }
        }
        const float ob = __shfl_xor(best[g], 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        const float ox = __shfl_xor(vx, 32, 64), oy = __shfl_xor(vy, 32, 64);
        const bool take = (ob < best[g]) || (ob == best[g] && oi < bi);
        if (hi == 0 && idx < cnt) contrib += fabsf(px[g] - (take ? ox : vx)) + fabsf(py[g] - (take ? oy : vy));
    }
    const float s = block_sum_256(contrib, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = s;
}

// A -> B through a cell grid over the mesh vertices, candidates evaluated on the matrix cores.
// The silhouette points are integer pixels, so an 8x8-pixel tile is one wave, and the vertices that can be the nearest
// neighbour of a pixel of the tile lie in the cells around it.  Per (image, slice) workgroup: counting-sort the image's P
// vertices into CELL x CELL-pixel cells in LDS (vertices outside the image go to the border cells), stage the silhouette
// bitmap, then every wave walks its tiles: skip the tile if no bit is set, else visit the tile's own cells, then ring after
// ring of cells around them.  After ring r every vertex not yet seen is farther from a pixel than the pixel's distance m to
// the edge of the visited block, so a pixel is finished once best + E < m^2 (E bounds the fp32 rounding of the expanded form:
// the comparison is between COMPUTED values, the same ones a full search compares, so the winner of the full search is always
// inside the visited set); the wave stops when all its set pixels are finished.
// Cells of one grid row are contiguous in the sorted array, so a ring is two row segments plus two cells per middle row.  The
// sorted array is cut into fixed chunks of 32 vertices; a range is evaluated chunk-wise (one v_mfma_f32_32x32x2_f32 per chunk
// and 32-pixel half tile, operands as in nn_a2b_mfma_kernel) -- vertices of neighbouring cells that share the chunk ride
// along for free, and a per-wave bitmask keeps a chunk from being evaluated twice.  The winner inside the winning
// 16-vertex group is recovered once per tile and ordered by (distance, vertex index) = tf.argmin's choice.  Chunks that reach
// exactly the best fp32 value again (an ulp is ~1e-3 px^2 at these magnitudes, so that is common) are remembered, up to three
// per pixel, and take part in that recovery; a pixel with more marks its tile, which is then re-evaluated over the same block
// with an explicit (distance, index) order on the VALU (integer vertex coordinates, duplicated vertices).
// Typical meshes: 10-30 chunks per tile instead of all 216; a mesh collapsed into a few cells degenerates to the full search.
// NPG: 32-pixel groups (8 x 4 pixels) per tile, stacked vertically; the tile is 8 x 4 NPG pixels
template <int CELL, int NPG>
__global__ __launch_bounds__(1024) void nn_a2b_grid_kernel(const unsigned long long* __restrict__ bits, const int* __restrict__ counts,
                                                           const float* __restrict__ v2d, int H, int W, int WW, int P, int Gx, int Gy,
                                                           float* __restrict__ partial, int nblk, int nslots, int* __restrict__ full_search,
                                                           int min_cells, unsigned long long* __restrict__ mfma_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char grid_smem[];
    const int NC = Gx * Gy;
    const int Ppad = (P + 31) & ~31;
    float* sX = reinterpret_cast<float*>(grid_smem);  // -2 x, sorted by cell
    float* sY = sX + Ppad;                            // -2 y
    float* sN = sY + Ppad;                            // |b|^2 (+inf in the padding)
    int* sI = reinterpret_cast<int*>(sN + Ppad);      // vertex index
    int* sStart = sI + Ppad;                          // [NC + 1]
    int* sCur = sStart + NC + 1;                      // [NC] histogram, then scatter cursors (+1 pad keeps sbits 8-byte aligned)
    unsigned long long* sbits = reinterpret_cast<unsigned long long*>(sCur + NC + 1);  // [H][WW]
    float* sTile = reinterpret_cast<float*>(sbits + H * WW);                            // [tiles of this workgroup]
    __shared__ int wtot[16];
    __shared__ int s_next;
    __shared__ float red[16];
    __shared__ int s_occupied;
    const int b = blockIdx.y, slice = blockIdx.x, nslice = gridDim.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cnt = counts[b];
#ifdef HPE_A2B_STAMPS  // diagnostics: 100 MHz wall-clock stamps of one workgroup's phases into counter[8 ...] (tools/a2b_phases.py)
#define A2B_STAMP(i) do { if (mfma_count && b == 5 && slice == 0 && t == 0) mfma_count[8 + (i)] = wall_clock64(); } while (0)
#else
#define A2B_STAMP(i) do { } while (0)
#endif
    A2B_STAMP(0);
    if (slice == 0)
        for (int i = nslice + t; i < nslots; i += 1024) partial[(size_t)b * nblk + i] = 0.f;
    if (cnt == 0) {
        if (t == 0) {
            partial[(size_t)b * nblk + slice] = 0.f;
            if (slice == 0) full_search[b] = 0;
        }
        return;
    }
    if (t == 0) s_occupied = s_next = 0;
    for (int i = t; i < NC; i += 1024) sCur[i] = 0;
    for (int i = t; i < H * WW; i += 1024) sbits[i] = bits[(size_t)b * H * WW + i];
    for (int i = P + t; i < Ppad; i += 1024) {
        sX[i] = 0.f;
        sY[i] = 0.f;
        sN[i] = __builtin_inff();
        sI[i] = 0x7fffffff;
    }
    __syncthreads();
    const float* Bp = v2d + (size_t)b * P * 2;
    const float gxm = (float)(Gx - 1), gym = (float)(Gy - 1);
    auto cell_of = [&](float x, float y) {
        // fmaxf(NaN, 0) = 0: a NaN vertex lands in a valid cell and never wins a comparison, as in the full search
        const int cx = (int)fminf(fmaxf(floorf(x * (1.0f / CELL)), 0.f), gxm);
        const int cy = (int)fminf(fmaxf(floorf(y * (1.0f / CELL)), 0.f), gym);
        return cy * Gx + cx;
    };
    A2B_STAMP(1);
    for (int i = t; i < P; i += 1024) atomicAdd(&sCur[cell_of(Bp[2 * i], Bp[2 * i + 1])], 1);
    __syncthreads();
    A2B_STAMP(2);
    {
        // exclusive scan of the histogram: `per` consecutive cells per thread, wave scan, 16 wave totals
        const int per = (NC + 1023) / 1024;
        int loc = 0;
        for (int k = 0; k < per; ++k) {
            const int idx = t * per + k;
            if (idx < NC) loc += sCur[idx];
        }
        {
            int occ = 0;
            for (int k = 0; k < per; ++k) {
                const int idx = t * per + k;
                if (idx < NC && sCur[idx] > 0) ++occ;
            }
            if (occ) atomicAdd(&s_occupied, occ);
        }
        int inc = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(inc, off, 64);
            if (lane >= off) inc += n;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int run = inc - loc;
        for (int w = 0; w < wave; ++w) run += wtot[w];
        for (int k = 0; k < per; ++k) {
            const int idx = t * per + k;
            if (idx < NC) {
                const int c = sCur[idx];
                sStart[idx] = run;
                sCur[idx] = run;
                run += c;
            }
        }
        if (t == 0) sStart[NC] = P;
    }
    __syncthreads();
    // A mesh concentrated in a few cells leaves nothing to prune (every tile ends up visiting most chunks, at a higher cost per
    // chunk than the full search): such an image is left to nn_a2b_mfma_kernel, launched behind this kernel for flagged images.
    A2B_STAMP(3);
    const bool concentrated = s_occupied < min_cells;
    if (t == 0 && slice == 0) full_search[b] = concentrated ? 1 : 0;
    if (concentrated) return;
    // scatter; the order inside a cell is whatever the atomics give, which decides nothing below: the search returns the lowest
    // vertex index among the vertices at the minimal computed distance, and that is a property of the set
    for (int i = t; i < P; i += 1024) {
        const float x = Bp[2 * i], y = Bp[2 * i + 1];
        const int pos = atomicAdd(&sCur[cell_of(x, y)], 1);
        sX[pos] = -2.0f * x;
        sY[pos] = -2.0f * y;
        sN[pos] = norm2(x, y);
        sI[pos] = i;
    }
    __syncthreads();
    A2B_STAMP(4);
    constexpr int TH = 4 * NPG;                  // tile height in pixels
    constexpr int TCX = 8 / CELL, TCY = TH / CELL;  // cells per tile edge
    static_assert(8 % CELL == 0 && TH % CELL == 0, "tiles are whole cells");
    constexpr float NONE = 3.0e38f;
    const int TX = (W + 7) >> 3, TY = (H + TH - 1) / TH, NT = TX * TY;
    const int hi = lane >> 5, l31 = lane & 31;
    const int lx = l31 & 7, ly = l31 >> 3;
    // tiles q * nslice + slice of this workgroup are handed out through an LDS counter (their cost varies with the distance to
    // the mesh); every tile leaves its sum in sTile[q], summed in a fixed order at the end, so the result does not depend on
    // which wave took which tile
    const int nq = (NT - slice + nslice - 1) / nslice;
    auto next_tile = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(&s_next, 1);
        return __shfl(v, 0, 64);
    };
    int wave_mfma = 0;  // MFMAs issued by this wave (diagnostics)
    auto tile_sum = [&](int q) -> float {
        const int tile = q * nslice + slice;
        const int ty = tile / TX, tx = tile - ty * TX;
        float px[NPG], py[NPG], aa[NPG], best[NPG];
        int bchunk[NPG][4];  // [0]: chunk of the best value; [1..3]: later chunks that reached exactly the same value
        int ntie[NPG];
        bool active[NPG];
        bool any_active = false;
#pragma unroll
        for (int g = 0; g < NPG; ++g) {
            const int col = tx * 8 + lx, row = ty * TH + 4 * g + ly;
            active[g] = (col < W && row < H) ? ((sbits[row * WW + (col >> 6)] >> (col & 63)) & 1ull) : false;
            any_active |= active[g];
            px[g] = (float)col;
            py[g] = (float)row;
            aa[g] = px[g] * px[g] + py[g] * py[g];
            best[g] = NONE;
            bchunk[g][0] = bchunk[g][1] = bchunk[g][2] = bchunk[g][3] = 0;
            ntie[g] = 0;
        }
        if (__ballot(any_active) == 0ull) return 0.f;
        bool overflow = false;  // more than three chunks tied with the best: the tile takes the explicit (distance, index) pass
        int vmask = 0;  // lane w holds bits 32w .. 32w+31 of the wave's visited-chunk set
        int nchunk = 0;  // chunks evaluated for this tile (diagnostics counter)
        // vertices [s, e) of the sorted array (wave-uniform), chunk by chunk
        auto scan = [&](int s, int e) {
            for (int k = s >> 5; k <= (e - 1) >> 5; ++k) {
                const int word = __builtin_amdgcn_readlane(vmask, k >> 5);
                if ((word >> (k & 31)) & 1) continue;
                vmask |= (lane == (k >> 5)) ? (1 << (k & 31)) : 0;
                ++nchunk;
                const float a = hi ? sY[32 * k + l31] : sX[32 * k + l31];
                f32x16_t qz;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 q = *reinterpret_cast<const float4*>(&sN[32 * k + 8 * j + 4 * hi]);
                    qz[4 * j] = q.x;
                    qz[4 * j + 1] = q.y;
                    qz[4 * j + 2] = q.z;
                    qz[4 * j + 3] = q.w;
                }
                f32x16_t d[NPG];
#pragma unroll
                for (int g = 0; g < NPG; ++g) d[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, hi ? py[g] : px[g], qz, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NPG; ++g) {
                    const float m0 = fminf(fminf(d[g][0], d[g][1]), d[g][2]);
                    const float m1 = fminf(fminf(d[g][3], d[g][4]), d[g][5]);
                    const float m2 = fminf(fminf(d[g][6], d[g][7]), d[g][8]);
                    const float m3 = fminf(fminf(d[g][9], d[g][10]), d[g][11]);
                    const float m4 = fminf(fminf(d[g][12], d[g][13]), d[g][14]);
                    const float m5 = fminf(fminf(m0, m1), d[g][15]);
                    const float m = fminf(fminf(m2, m3), fminf(m4, m5));
                    // most chunks of the outer rings improve no pixel of the tile: one compare + ballot skips the bookkeeping
                    // (pixels outside the silhouette are never read back: they do not keep a chunk alive)
                    if (!__any(active[g] && m <= best[g])) continue;
                    const bool lt = m < best[g];
                    const bool eq = m == best[g];
                    best[g] = fminf(m, best[g]);
                    bchunk[g][0] = lt ? k : bchunk[g][0];
                    ntie[g] = lt ? 0 : ntie[g];
                    if (__any(eq)) {
                        // equal fp32 values in two chunks (at ~1e4 px^2 an ulp is 1e-3 px^2: not rare); vertex indices decide
                        // at the end, so the chunk is remembered
                        ntie[g] += eq ? 1 : 0;
                        bchunk[g][1] = (eq && ntie[g] == 1) ? k : bchunk[g][1];
                        bchunk[g][2] = (eq && ntie[g] == 2) ? k : bchunk[g][2];
                        bchunk[g][3] = (eq && ntie[g] == 3) ? k : bchunk[g][3];
                    }
                }
            }
        };
        const int X0 = tx * TCX, Y0 = ty * TCY, X1 = min(X0 + TCX - 1, Gx - 1), Y1 = min(Y0 + TCY - 1, Gy - 1);
        int xa, xb, ya, yb;
        for (int r = 0;; ++r) {
            xa = X0 - r, xb = X1 + r, ya = Y0 - r, yb = Y1 + r;
            const int xac = max(xa, 0), xbc = min(xb, Gx - 1), yac = max(ya, 0), ybc = min(yb, Gy - 1);
            // (Round 4 tried having the lanes look the ring's cell ranges up in parallel and hand them out through readlane instead of one
            // LDS -> SGPR round trip per range: 0.688 / 0.528 / 0.323 against 0.688 / 0.527 / 0.317 ms per call on the bench's stage-1 /
            // stretched / evenly spread meshes -- the lookups are not what the search waits for.  Not kept.)
            for (int cy = yac; cy <= ybc; ++cy) {
                auto one = [&](int c0, int c1) {
                    const int s_ = __builtin_amdgcn_readfirstlane(sStart[c0]), e_ = __builtin_amdgcn_readfirstlane(sStart[c1 + 1]);
                    if (e_ > s_) scan(s_, e_);
                };
                if (r == 0 || cy == ya || cy == yb) {
                    one(cy * Gx + xac, cy * Gx + xbc);
                } else {
                    if (xa >= 0) one(cy * Gx + xa, cy * Gx + xa);
                    if (xb < Gx) one(cy * Gx + xb, cy * Gx + xb);
                }
            }
            if (xa <= 0 && xb >= Gx - 1 && ya <= 0 && yb >= Gy - 1) break;  // every cell visited
            bool done = true;
#pragma unroll
            for (int g = 0; g < NPG; ++g) {
                // distance from the pixel to the nearest side of the visited block that still has cells beyond it
                float m = __builtin_inff();
                if (xa > 0) m = fminf(m, px[g] - (float)(xa * CELL));
                if (xb < Gx - 1) m = fminf(m, (float)((xb + 1) * CELL) - px[g]);
                if (ya > 0) m = fminf(m, py[g] - (float)(ya * CELL));
                if (yb < Gy - 1) m = fminf(m, (float)((yb + 1) * CELL) - py[g]);
                const float m2 = m * m;
                const float bm = fminf(best[g], __shfl_xor(best[g], 32, 64));  // the two lane halves see different vertex rows
                // |error of a computed distance| <= ~1e-6 (3 |a|^2 + 2 d^2) for any candidate at true distance d >= m
                done = done && (!active[g] || ((bm + aa[g]) + (3e-6f * aa[g] + 2e-6f * m2) < m2));
            }
            if (__all(done)) break;
        }
        // winner inside the winning 16-vertex groups of each lane half (the best chunk and the chunks tied with it), then the
        // better half; order (distance, vertex index)
        float wx[NPG], wy[NPG];
#pragma unroll
        for (int g = 0; g < NPG; ++g) {
            float vx = 0.f, vy = 0.f;
            int bi = 0x7fffffff;
            const int nt = (best[g] < NONE) ? min(ntie[g], 3) + 1 : 0;
            overflow |= ntie[g] > 3 && active[g];
            for (int a = 0; a < 4; ++a) {
                if (!__any(a < nt)) break;
                if (a < nt) {
                    const int ck = a == 0 ? bchunk[g][0] : (a == 1 ? bchunk[g][1] : (a == 2 ? bchunk[g][2] : bchunk[g][3]));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p0 = 32 * ck + 8 * j + 4 * hi;
                        const float4 qx = *reinterpret_cast<const float4*>(&sX[p0]);
                        const float4 qy = *reinterpret_cast<const float4*>(&sY[p0]);
                        const float4 qn = *reinterpret_cast<const float4*>(&sN[p0]);
                        const int4 qi = *reinterpret_cast<const int4*>(&sI[p0]);
                        const float ex[4] = {qx.x, qx.y, qx.z, qx.w}, ey[4] = {qy.x, qy.y, qy.z, qy.w}, en[4] = {qn.x, qn.y, qn.z, qn.w};
                        const int ei[4] = {qi.x, qi.y, qi.z, qi.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float dd = mfma_k2_value(ex[i], px[g], ey[i], py[g], en[i]);
                            if (dd == best[g] && ei[i] < bi) {
                                bi = ei[i];
                                vx = -0.5f * ex[i];
                                vy = -0.5f * ey[i];
                            }
                        }
                    }
                }
            }
            const float ob = __shfl_xor(best[g], 32, 64);
            const int oi = __shfl_xor(bi, 32, 64);
            const float ox = __shfl_xor(vx, 32, 64), oy = __shfl_xor(vy, 32, 64);
            const bool take = (ob < best[g]) || (ob == best[g] && oi < bi);
            wx[g] = take ? ox : vx;
            wy[g] = take ? oy : vy;
        }
        if (__any(overflow)) {
            // more tied chunks than the lane keeps: the same block again with an explicit (distance, vertex index) order
            float bd[NPG];
            int bi[NPG];
#pragma unroll
            for (int g = 0; g < NPG; ++g) {
                bd[g] = __builtin_inff();
                bi[g] = 0x7fffffff;
                wx[g] = wy[g] = 0.f;
            }
            const int xac = max(xa, 0), xbc = min(xb, Gx - 1), yac = max(ya, 0), ybc = min(yb, Gy - 1);
            for (int cy = yac; cy <= ybc; ++cy) {
                const int s = __builtin_amdgcn_readfirstlane(sStart[cy * Gx + xac]);
                const int e = __builtin_amdgcn_readfirstlane(sStart[cy * Gx + xbc + 1]);
                for (int j = s; j < e; ++j) {
                    const float ex = sX[j], ey = sY[j], en = sN[j];
                    const int ei = sI[j];
#pragma unroll
                    for (int g = 0; g < NPG; ++g) {
                        const float dd = mfma_k2_value(ex, px[g], ey, py[g], en);
                        const bool lt = dd < bd[g] || (dd == bd[g] && ei < bi[g]);
                        bd[g] = lt ? dd : bd[g];
                        bi[g] = lt ? ei : bi[g];
                        wx[g] = lt ? -0.5f * ex : wx[g];
                        wy[g] = lt ? -0.5f * ey : wy[g];
                    }
                }
            }
        }
        float contrib = 0.f;
        if (hi == 0) {
#pragma unroll
            for (int g = 0; g < NPG; ++g)
                if (active[g]) contrib += fabsf(px[g] - wx[g]) + fabsf(py[g] - wy[g]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) contrib += __shfl_xor(contrib, off, 64);
        wave_mfma += nchunk * NPG;
        return contrib;
    };
    for (int q = next_tile(); q < nq; q = next_tile()) {
        const float c = tile_sum(q);
        if (lane == 0) sTile[q] = c;
    }
    // diagnostics counter: ONE atomic per wave.  (Round 3 added one per TILE: ~200 k atomics on one address per launch, which serialise
    // at ~88 per microsecond -- the counted step of bench.py's loss_roofline measured 0.77 ms per search where the uncounted kernel takes
    // 0.35 ms, so `frac` was 2x pessimistic.)
    if (mfma_count && lane == 0 && wave_mfma) atomicAdd(mfma_count, (unsigned long long)wave_mfma);
    A2B_STAMP(5);  // thread 0's wave is out of tiles
    __syncthreads();
    A2B_STAMP(6);  // every wave is
    float acc = 0.f;
    for (int q = t; q < nq; q += 1024) acc += sTile[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += red[w];
        partial[(size_t)b * nblk + slice] = s;
    }
}

// direction B -> A: every mesh vertex finds its nearest silhouette point, contributes ||b - a*||_2.
// grid (ceil(P/1024), B); 4 vertices per lane; A streamed through LDS in tiles of 2048 points (x, y interleaved,
// one broadcast ds_read_b128 = 2 points = 8 pair evaluations).
__global__ __launch_bounds__(256) void nn_b2a_kernel(const float* __restrict__ pts, const int* __restrict__ counts,
                                                     const float* __restrict__ v2d, int HW, int P, float* __restrict__ partial,
                                                     int nblk, int blk_off) {
    __shared__ __attribute__((aligned(16))) float sA[2048 * 2 + 4];
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int cnt = counts[b];
    float bx[NN_PT], by[NN_PT], bb[NN_PT], best[NN_PT], cx[NN_PT], cy[NN_PT];
    bool live[NN_PT];
#pragma unroll
    for (int u = 0; u < NN_PT; ++u) {
        const int p = blockIdx.x * 256 * NN_PT + u * 256 + threadIdx.x;
        live[u] = p < P;
        const int pc = live[u] ? p : P - 1;
        bx[u] = v2d[((size_t)b * P + pc) * 2];
        by[u] = v2d[((size_t)b * P + pc) * 2 + 1];
        bb[u] = bx[u] * bx[u] + by[u] * by[u];
        best[u] = 3.4e38f;
        cx[u] = cy[u] = 0.f;
    }
    const float* Ap = pts + (size_t)b * HW * 2;
    for (int t0 = 0; t0 < cnt; t0 += 2048) {
        const int n = min(2048, cnt - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * n; i += 256) sA[i] = Ap[2 * t0 + i];
        if ((n & 1) && threadIdx.x == 0) {  // pad to an even count with a copy of the last point (never wins a tie: later index)
            sA[2 * n] = Ap[2 * (t0 + n - 1)];
            sA[2 * n + 1] = Ap[2 * (t0 + n - 1) + 1];
        }
        __syncthreads();
        const int n2 = (n + 1) >> 1;
        for (int i = 0; i < n2; ++i) {
            const float4 q = *reinterpret_cast<const float4*>(&sA[4 * i]);
            const float a0 = q.x * q.x + q.y * q.y, a1 = q.z * q.z + q.w * q.w;
#pragma unroll
            for (int u = 0; u < NN_PT; ++u) {
                const float d0 = (-2.0f * (q.x * bx[u] + q.y * by[u]) + a0) + bb[u];
                bool lt = d0 < best[u];
                best[u] = lt ? d0 : best[u];
                cx[u] = lt ? q.x : cx[u];
                cy[u] = lt ? q.y : cy[u];
                const float d1 = (-2.0f * (q.z * bx[u] + q.w * by[u]) + a1) + bb[u];
                lt = d1 < best[u];
                best[u] = lt ? d1 : best[u];
                cx[u] = lt ? q.z : cx[u];
                cy[u] = lt ? q.w : cy[u];
            }
        }
    }
    float contrib = 0.f;
#pragma unroll
    for (int u = 0; u < NN_PT; ++u) {
        if (live[u] && cnt > 0) {
            const float dx = bx[u] - cx[u], dy = by[u] - cy[u];
            contrib += sqrtf(dx * dx + dy * dy);
        }
    }
    const float s = block_sum_256(contrib, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blk_off + blockIdx.x] = s;
}

// silhouette bitmap: bits[b][y][w] bit x%64 of word x/64 is set iff seg[b][y][x] > 0; one wave builds one word with a ballot
__global__ __launch_bounds__(256) void sil_bitmap_kernel(const float* __restrict__ seg, int H, int W, int WW,
                                                         unsigned long long* __restrict__ bits, long nwords) {
    const long word = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (word >= nwords) return;
    const int lane = threadIdx.x & 63;
    const int w = (int)(word % WW);
    const long row = word / WW;  // b * H + y
    const int x = w * 64 + lane;
    const float v = (x < W) ? seg[row * W + x] : 0.f;
    const unsigned long long m = __ballot(v > 0.f);
    if (lane == 0) bits[word] = m;
}

// tf.where order compaction from the bitmap (7 KB per image instead of two passes over 200 KB of floats): a thread owns 4
// consecutive words, a block scan of their popcounts gives its first output slot, then it walks its set bits
__global__ __launch_bounds__(256) void sil_compact_bits_kernel(const unsigned long long* __restrict__ bits, int H, int W, int WW,
                                                               float* __restrict__ pts, int* __restrict__ counts) {
    __shared__ int wtot[4];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nwords = H * WW, per = (nwords + 255) / 256;
    const unsigned long long* src = bits + (size_t)b * nwords;
    int loc = 0;
    for (int k = 0; k < per; ++k) {
        const int w = t * per + k;
        if (w < nwords) loc += __popcll(src[w]);
    }
    int inc = loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(inc, off, 64);
        if (lane >= off) inc += n;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int pos = inc - loc;
    for (int w = 0; w < wave; ++w) pos += wtot[w];
    float* o = pts + (size_t)b * H * W * 2;
    for (int k = 0; k < per; ++k) {
        const int w = t * per + k;
        if (w >= nwords) break;
        unsigned long long m = src[w];
        const int y = w / WW, x0 = (w - y * WW) * 64;
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            *reinterpret_cast<float2*>(&o[2 * pos]) = make_float2((float)(x0 + bit), (float)y);
            ++pos;
        }
    }
    if (t == 0) counts[b] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}

// direction B -> A on the pixel grid: the silhouette points are integer pixels, so for one vertex only TWO pixels per image
// row can be its nearest neighbour (the closest set bit on either side of its x), and rows farther than the best distance
// found so far cannot win.  Rows are visited outwards from the vertex' own row; the candidate distance is still the
// reference's expanded fp32 form and ties are resolved to the lowest (y, x) = lowest tf.where index, so the chosen pixel is
// the one tf.argmin picks.  ~2*sqrt(d) rows x 2 candidates per vertex instead of P_i (~12k) candidates.
#define NN_MAXWW 8
// Round 4: two things cut the walk (0.29 -> see DESIGN.md ms per stage at B = 256), neither changes which pixel wins:
//  * a per-row record built once per workgroup from the bitmap -- first and last set column and whether the row is ONE run of set
//    pixels (a convex silhouette: every row) -- answers "nearest set bit at or left of x0 / right of x0" with two clamps instead of
//    two word-by-word scans with 64-bit clz / ffs (rows with several runs keep the scans); empty rows cost one LDS read;
//  * the walk starts at the vertex' row clamped into the silhouette's row range [ymin, ymax] and never leaves that range: the rows it
//    skips are empty.  The visiting order does not matter: try_pixel's tie rule is explicit and a row is pruned only by its own
//    vertical distance against the best value so far, which is monotone along each direction;
//  * rows are visited in aligned blocks of 8 with a record per block (smallest first column, largest last column): a block whose
//    bounding rectangle is farther from the vertex than the best distance so far is skipped whole.  A vertex beside the silhouette
//    used to scan every row within its horizontal distance; the rows above and below where the shape has narrowed away from it now
//    cost one rectangle test per 8.  (The rectangle bound is not monotone along the walk -- a farther block may be wider -- so only
//    the vertical distance ends a direction.)
__global__ __launch_bounds__(256) void nn_b2a_rows_kernel(const unsigned long long* __restrict__ bits, const int* __restrict__ counts,
                                                          const float* __restrict__ v2d, int H, int W, int WW, int P,
                                                          float* __restrict__ partial, int nblk, int blk_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sbits[];  // [H][WW], then rowinfo[H]
    __shared__ float red[4];
    __shared__ int s_ymin, s_ymax;
    unsigned* rowinfo = reinterpret_cast<unsigned*>(sbits + H * WW);  // xl | xr << 10 | kind << 20 (0 empty, 1 one run, 2 several runs)
    unsigned* blkinfo = rowinfo + H;                                  // per aligned block of 8 rows: min xl | max xr << 10 | nonempty << 20
    const int b = blockIdx.y;
    const int cnt = counts[b];
    if (threadIdx.x == 0) {
        s_ymin = H;
        s_ymax = -1;
    }
    for (int i = threadIdx.x; i < H * WW; i += 256) sbits[i] = bits[(size_t)b * H * WW + i];
    __syncthreads();
    for (int y = threadIdx.x; y < H; y += 256) {
        int xl = -1, xr = -1, pc = 0;
        for (int i = 0; i < WW; ++i) {
            const unsigned long long m = sbits[y * WW + i];
            if (m) {
                if (xl < 0) xl = i * 64 + __ffsll((long long)m) - 1;
                xr = i * 64 + 63 - __clzll(m);
                pc += __popcll(m);
            }
        }
        unsigned info = 0;
        if (pc > 0) {
            info = (unsigned)xl | ((unsigned)xr << 10) | ((pc == xr - xl + 1 ? 1u : 2u) << 20);
            atomicMin(&s_ymin, y);
            atomicMax(&s_ymax, y);
        }
        rowinfo[y] = info;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < (H + 7) / 8; j += 256) {
        unsigned lo = 1023u, hi_ = 0u, any = 0u;
        for (int y = 8 * j; y < min(8 * j + 8, H); ++y) {
            const unsigned info = rowinfo[y];
            if (info >> 20) {
                lo = min(lo, info & 1023u);
                hi_ = max(hi_, (info >> 10) & 1023u);
                any = 1u;
            }
        }
        blkinfo[j] = lo | (hi_ << 10) | (any << 20);
    }
    __syncthreads();
    const int ymin = s_ymin, ymax = s_ymax;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float contrib = 0.f;
    if (p < P && cnt > 0 && ymax >= ymin) {
        const float bx = v2d[((size_t)b * P + p) * 2], by = v2d[((size_t)b * P + p) * 2 + 1];
        const float bb = bx * bx + by * by;
        const int x0 = min(max((int)rintf(bx), 0), W - 1);
        const int y0 = min(max((int)rintf(by), 0), H - 1);
        float best = 3.4e38f;
        int cx = 0, cy = 0;
        auto try_pixel = [&](int x, int y) {
            const float ax = (float)x, ay = (float)y;
            const float d = (-2.0f * (ax * bx + ay * by) + (ax * ax + ay * ay)) + bb;
            if (d < best || (d == best && (y < cy || (y == cy && x < cx)))) {
                best = d;
                cx = x;
                cy = y;
            }
        };
        auto scan_row = [&](int y) {
            const unsigned info = rowinfo[y];
            const unsigned kind = info >> 20;
            if (kind == 0) return;
            if (kind == 1) {
                const int xl = (int)(info & 1023u), xr = (int)((info >> 10) & 1023u);
                if (xl <= x0) try_pixel(min(x0, xr), y);      // nearest set bit at or left of x0
                if (xr > x0) try_pixel(max(x0 + 1, xl), y);   // nearest set bit right of x0
                return;
            }
            const unsigned long long* row = sbits + y * WW;
            // nearest set bit at or left of x0
            for (int i = x0 >> 6; i >= 0; --i) {
                unsigned long long m = row[i];
                if (i == (x0 >> 6)) m &= (~0ull >> (63 - (x0 & 63)));
                if (m) {
                    try_pixel(i * 64 + 63 - __clzll(m), y);
                    break;
                }
            }
            // nearest set bit right of x0
            const int x1 = x0 + 1;
            for (int i = x1 >> 6; i < WW; ++i) {
                unsigned long long m = row[i];
                if (i == (x1 >> 6)) m &= (~0ull << (x1 & 63));
                if (m) {
                    try_pixel(i * 64 + __ffsll((long long)m) - 1, y);
                    break;
                }
            }
        };
        const int ys = min(max(y0, ymin), ymax);
        scan_row(ys);
        int yd = ys + 1, yu = ys - 1;
        bool down = yd <= ymax, up = yu >= ymin;
        // squared distance from the vertex to the block's bounding rectangle [min xl, max xr] x [rows of the segment]; dyv = vertical
        // distance to the segment's nearest row (<= 0: the vertex is level with or inside the segment's rows)
        auto block_far = [&](unsigned binfo, float dyv) {
            if (!(binfo >> 20)) return true;
            const float xl = (float)(binfo & 1023u), xr = (float)((binfo >> 10) & 1023u);
            const float dxr = fmaxf(fmaxf(xl - bx, bx - xr), 0.f), dyr = fmaxf(dyv, 0.f);
            return dxr * dxr + dyr * dyr > best + 0.5f;  // the expanded-form rounding error of a candidate is << 0.5 at these magnitudes
        };
        while (down || up) {
            if (down) {
                // rows yd .. yend: the rest of yd's aligned block of 8.  A row farther than the best distance cannot win (and neither can
                // any row below it): the direction ends; dy <= 0 only while the walk is still above the vertex (no pruning there).
                const int yend = min((yd | 7), ymax);
                const float dy0 = (float)yd - by;
                if (dy0 > 0.f && dy0 * dy0 > best + 0.25f) {
                    down = false;
                } else {
                    if (!block_far(blkinfo[yd >> 3], dy0)) {
                        for (int y = yd; y <= yend; ++y) {
                            const float dy = (float)y - by;
                            if (dy > 0.f && dy * dy > best + 0.25f) {
                                down = false;
                                break;
                            }
                            scan_row(y);
                        }
                    }
                    yd = yend + 1;
                    if (yd > ymax) down = false;
                }
            }
            if (up) {
                const int yend = max((yu & ~7), ymin);
                const float dy0 = by - (float)yu;
                if (dy0 > 0.f && dy0 * dy0 > best + 0.25f) {
                    up = false;
                } else {
                    if (!block_far(blkinfo[yu >> 3], dy0)) {
                        for (int y = yu; y >= yend; --y) {
                            const float dy = by - (float)y;
                            if (dy > 0.f && dy * dy > best + 0.25f) {
                                up = false;
                                break;
                            }
                            scan_row(y);
                        }
                    }
                    yu = yend - 1;
                    if (yu < ymin) up = false;
                }
            }
        }
        const float dx = bx - (float)cx, dy = by - (float)cy;
        contrib = sqrtf(dx * dx + dy * dy);
    }
    const float s = block_sum_256(contrib, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blk_off + blockIdx.x] = s;
}

// out[0] = sum_b ( sum of image b's partials ) / (3 + P), images added in index order (src/ops.py:129-136).
// A wave sums one image's partials with shuffles (no barrier); the per-image values are then added sequentially in
// image order by one thread, so the result does not depend on the launch geometry.
__global__ __launch_bounds__(1024) void mesh_loss_finish_kernel(const float* __restrict__ partial, int B, int nblk, int nused, int P,
                                                                float* __restrict__ out) {
    extern __shared__ float per_image[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B; b += 16) {  // 16 waves: 16 images in flight (4 waves took 39 us for 256 images, all of it load latency)
        float v = 0.f;
        for (int i = lane; i < nused; i += 64) v += partial[(size_t)b * nblk + i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) per_image[b] = v / (float)(3 + P);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int b = 0; b < B; ++b) total += per_image[b];
        out[0] = total;
    }
}

}  // namespace

#define A2B_GRID_MAX_LDS (156 * 1024)
static size_t a2b_grid_lds_bytes(int H, int W, int WW, int P, int Gx, int Gy) {
    return (size_t)((P + 31) & ~31) * 16 + (size_t)(2 * Gx * Gy + 2) * 4 + (size_t)H * WW * 8 + (size_t)((W + 7) / 8) * ((H + 7) / 8) * 4;
}

typedef void (*A2bGridKernel)(const unsigned long long*, const int*, const float*, int, int, int, int, int, int, float*, int, int, int*, int,
                              unsigned long long*);
// (cell edge, 32-pixel groups per tile) variants; [0] is the default
static const struct {
    int cell, npg;
    A2bGridKernel fn;
} a2b_grid_variants[] = {{8, 2, nn_a2b_grid_kernel<8, 2>}, {4, 2, nn_a2b_grid_kernel<4, 2>}};

hipError_t hpe_losses_init_device() {
    for (const auto& v : a2b_grid_variants) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, A2B_GRID_MAX_LDS);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t hpe_launch_kp_loss(const float* gt, const float* pred, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(kp_loss_kernel, dim3(1), dim3(256), 0, st, gt, pred, n, out);
    return hipGetLastError();
}

size_t hpe_mesh_loss_ws_floats(int B, int H, int W, int P) {
    const int HW = H * W;
    const int nblk = (HW + 1023) / 1024 + (P + 255) / 256;
    const size_t bitmap_floats = (size_t)B * H * ((W + 63) / 64) * 2;
    return (size_t)B * HW * 2 + (size_t)B * nblk + (size_t)B + 64 + bitmap_floats + 16 + (size_t)B;
}

// Workspace layout shared by the two halves of the mesh loss
struct MeshWs {
    float* pts;
    float* partial;
    int* counts;
    unsigned long long* bits;
    int* full_search;  // [B] per image: 1 = the pixel -> vertex search of this image is done by the full search
    int nA, nB, nblk, WW;
    bool grid_path;
};

static MeshWs mesh_ws_layout(float* ws, int B, int H, int W, int P) {
    MeshWs m;
    const int HW = H * W;
    m.WW = (W + 63) / 64;
    m.grid_path = m.WW <= NN_MAXWW && (size_t)H * m.WW * 8 <= 64 * 1024;
    m.nA = (HW + 1023) / 1024;
    m.nB = m.grid_path ? (P + 255) / 256 : (P + 1023) / 1024;
    m.nblk = (HW + 1023) / 1024 + (P + 255) / 256;  // workspace pitch (>= nA + nB)
    m.pts = ws;
    m.partial = m.pts + (size_t)B * HW * 2;
    m.counts = reinterpret_cast<int*>(m.partial + (size_t)B * m.nblk);
    // 8-byte aligned bitmap after the counts
    size_t off = (size_t)B * HW * 2 + (size_t)B * m.nblk + (size_t)B + 2;
    off = (off + 1) & ~(size_t)1;
    m.bits = reinterpret_cast<unsigned long long*>(ws + off);
    m.full_search = reinterpret_cast<int*>(ws + off + (size_t)B * H * m.WW * 2);
    return m;
}

// Step-invariant half: the ground-truth silhouette does not change between the IEF stages of one step (src/trainer.py:285-296
// evaluates the loss of every stage against the same seg_gts), so its compaction (tf.where order) and its bitmap are built once.
hipError_t hpe_launch_mesh_loss_prepare(const float* seg, int B, int H, int W, int P, float* ws, hipStream_t st) {
    const MeshWs m = mesh_ws_layout(ws, B, H, W, P);
    const int HW = H * W;
    if (!m.grid_path) {
        hipLaunchKernelGGL(sil_compact_kernel, dim3(B), dim3(256), 0, st, seg, HW, W, m.pts, m.counts);
        return hipGetLastError();
    }
    const long nwords = (long)B * H * m.WW;
    hipLaunchKernelGGL(sil_bitmap_kernel, dim3((unsigned)((nwords + 3) / 4)), dim3(256), 0, st, seg, H, W, m.WW, m.bits, nwords);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sil_compact_bits_kernel, dim3(B), dim3(256), 0, st, (const unsigned long long*)m.bits, H, W, m.WW, m.pts, m.counts);
    return hipGetLastError();
}

// Per-stage half: both nearest-neighbour searches against the prepared silhouette + the reduction.
int hpe_mesh_a2b_mode_from_env() {
    // HPE_MESH_A2B: "grid" (default: cell-grid search), "mfma" / "valu" (the full searches, for A/B comparisons)
    const char* e = getenv("HPE_MESH_A2B");
    return !e ? 0 : (e[0] == 'v' ? 1 : (e[0] == 'm' ? 2 : 0));
}

hipError_t hpe_launch_mesh_loss_search(const float* v2d, int B, int H, int W, int P, float* ws, float* out, hipStream_t st,
                                       hipEvent_t ev_a2b0, hipEvent_t ev_a2b1, int a2b_mode, unsigned long long* counter) {
    const MeshWs m = mesh_ws_layout(ws, B, H, W, P);
    const int HW = H * W;
    // images whose vertices occupy fewer cells than this go to the full search (0: never)
    static const int a2b_min_cells = [] {
        const char* e = getenv("HPE_MESH_A2B_MINCELLS");
        return e ? atoi(e) : 40;
    }();
    static const int a2b_slices = [] {
        const char* e = getenv("HPE_MESH_A2B_SLICES");  // workgroups per image of the grid search (0: about 512 / B)
        return e ? atoi(e) : 0;
    }();
    if (ev_a2b0) (void)hipEventRecord(ev_a2b0, st);
    static const int a2b_variant = [] {
        // cell edge 8 (default) / 4 pixels.  Measured on the config-5 inputs (tools/mesh_loss_bench.py, ms per loss call, B = 256):
        // 8 px cells 0.83 / 0.66 / 0.41, 4 px cells 0.99 / 0.76 / 0.39, 8 x 16-pixel tiles (4 groups per wave) 0.90 / 0.70 / 0.43
        const char* c = getenv("HPE_MESH_A2B_CELL");
        const int cell = c ? atoi(c) : 8;
        for (int i = 0; i < 2; ++i)
            if (a2b_grid_variants[i].cell == cell) return i;
        return 0;
    }();
    const int cell = a2b_grid_variants[a2b_variant].cell;
    const int Gx = (W + cell - 1) / cell, Gy = (H + cell - 1) / cell;
    const size_t grid_lds = a2b_grid_lds_bytes(H, W, m.WW, P, Gx, Gy);
    if (a2b_mode == 0 && m.grid_path && grid_lds <= A2B_GRID_MAX_LDS) {
        int nslice = a2b_slices > 0 ? a2b_slices : (512 + B - 1) / B;
        nslice = std::max(1, std::min(nslice, m.nA));
        const int min_cells = a2b_min_cells * (64 / (cell * cell));  // the knob is in 8 x 8-pixel cells
        hipLaunchKernelGGL(a2b_grid_variants[a2b_variant].fn, dim3(nslice, B), dim3(1024), grid_lds, st, m.bits, m.counts, v2d, H, W, m.WW, P,
                           Gx, Gy, m.partial, m.nblk, m.nA, m.full_search, min_cells, counter);
        hipError_t eg = hipGetLastError();
        if (eg != hipSuccess) return eg;
        hipLaunchKernelGGL(nn_a2b_mfma_kernel, dim3(m.nA, B), dim3(256), 0, st, m.pts, m.counts, v2d, HW, P, m.partial, m.nblk,
                               (const int*)m.full_search, counter ? counter + 1 : nullptr);
    } else if (a2b_mode == 1)
        hipLaunchKernelGGL(nn_a2b_kernel, dim3(m.nA, B), dim3(256), 0, st, m.pts, m.counts, v2d, HW, P, m.partial, m.nblk);
    else
        hipLaunchKernelGGL(nn_a2b_mfma_kernel, dim3(m.nA, B), dim3(256), 0, st, m.pts, m.counts, v2d, HW, P, m.partial, m.nblk,
                           (const int*)nullptr, counter ? counter + 1 : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ev_a2b1) (void)hipEventRecord(ev_a2b1, st);
    if (m.grid_path) {
        hipLaunchKernelGGL(nn_b2a_rows_kernel, dim3(m.nB, B), dim3(256), (size_t)H * m.WW * 8 + (size_t)(H + (H + 7) / 8) * 4, st, m.bits, m.counts, v2d, H, W, m.WW, P,
                           m.partial, m.nblk, m.nA);
    } else {
        hipLaunchKernelGGL(nn_b2a_kernel, dim3(m.nB, B), dim3(256), 0, st, m.pts, m.counts, v2d, HW, P, m.partial, m.nblk, m.nA);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // zero-fill is not needed: every partial slot in [0, nA + nB) is written; finish sums exactly those
    hipLaunchKernelGGL(mesh_loss_finish_kernel, dim3(1), dim3(1024), (size_t)B * sizeof(float), st, m.partial, B, m.nblk, m.nA + m.nB, P, out);
    return hipGetLastError();
}

hipError_t hpe_launch_mesh_loss(const float* seg, const float* v2d, int B, int H, int W, int P, float* ws, float* out,
                                hipStream_t st, int a2b_mode, unsigned long long* counter) {
    hipError_t e = hpe_launch_mesh_loss_prepare(seg, B, H, W, P, ws, st);
    if (e != hipSuccess) return e;
    return hpe_launch_mesh_loss_search(v2d, B, H, W, P, ws, out, st, nullptr, nullptr, a2b_mode, counter);
}
