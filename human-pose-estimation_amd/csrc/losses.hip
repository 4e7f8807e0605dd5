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
#define NN_PG 8  // point groups of 32 per wave -> 1024 points per 256-thread block
// (the explicit waves-per-SIMD bound makes hipcc keep the MFMA results in VGPRs; without it they land in AGPRs and every value
// costs an extra v_accvgpr_read before the VALU can touch it)
__global__ __launch_bounds__(256, 3) void nn_a2b_mfma_kernel(const float* __restrict__ pts, const int* __restrict__ counts,
                                                          const float* __restrict__ v2d, int HW, int P, float* __restrict__ partial,
                                                          int nblk) {
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
                bn = bx * bx + by * by;
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
    float contrib = 0.f;
#pragma unroll
    for (int g = 0; g < NN_PG; ++g) {
        // merge the halves (lower value, then lower group id), then recover the vertex inside the winning 16-vertex group
        const float ob = __shfl_xor(best[g], 32, 64);
        const int og = __shfl_xor(bgrp[g], 32, 64);
        const bool take = (ob < best[g]) || (ob == best[g] && og < bgrp[g]);
        const int grp = take ? og : bgrp[g];
        const int idx = base + wave * (32 * NN_PG) + g * 32 + l31;
        if (hi == 0 && idx < cnt) {
            const int v0 = (grp >> 1) * 32 + 4 * (grp & 1);
            float bd = __builtin_inff(), vx = 0.f, vy = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int vi = v0 + (e & 3) + 8 * (e >> 2);  // ascending with e
                if (vi < P) {
                    const float bx = Bp[2 * vi], by = Bp[2 * vi + 1];
                    const float dd = fmaf(-2.0f * bx, px[g], fmaf(-2.0f * by, py[g], bx * bx + by * by));
                    if (dd < bd) {
                        bd = dd;
                        vx = bx;
                        vy = by;
                    }
                }
            }
            contrib += fabsf(px[g] - vx) + fabsf(py[g] - vy);
        }
    }
    const float s = block_sum_256(contrib, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = s;
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

// direction B -> A on the pixel grid: the silhouette points are integer pixels, so for one vertex only TWO pixels per image
// row can be its nearest neighbour (the closest set bit on either side of its x), and rows farther than the best distance
// found so far cannot win.  Rows are visited outwards from the vertex' own row; the candidate distance is still the
// reference's expanded fp32 form and ties are resolved to the lowest (y, x) = lowest tf.where index, so the chosen pixel is
// the one tf.argmin picks.  ~2*sqrt(d) rows x 2 candidates per vertex instead of P_i (~12k) candidates.
#define NN_MAXWW 8
__global__ __launch_bounds__(256) void nn_b2a_rows_kernel(const unsigned long long* __restrict__ bits, const int* __restrict__ counts,
                                                          const float* __restrict__ v2d, int H, int W, int WW, int P,
                                                          float* __restrict__ partial, int nblk, int blk_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sbits[];  // [H][WW]
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int cnt = counts[b];
    for (int i = threadIdx.x; i < H * WW; i += 256) sbits[i] = bits[(size_t)b * H * WW + i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    float contrib = 0.f;
    if (p < P && cnt > 0) {
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
        scan_row(y0);
        bool up = true, down = true;
        for (int k = 1; (up || down) && k < H; ++k) {
            if (down) {
                const int y = y0 + k;
                const float dy = (float)y - by;
                if (y >= H || dy * dy > best + 0.25f) down = false;  // expanded-form rounding error is << 0.25 here
                else scan_row(y);
            }
            if (up) {
                const int y = y0 - k;
                const float dy = by - (float)y;
                if (y < 0 || dy * dy > best + 0.25f) up = false;
                else scan_row(y);
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

hipError_t hpe_launch_kp_loss(const float* gt, const float* pred, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(kp_loss_kernel, dim3(1), dim3(256), 0, st, gt, pred, n, out);
    return hipGetLastError();
}

size_t hpe_mesh_loss_ws_floats(int B, int H, int W, int P) {
    const int HW = H * W;
    const int nblk = (HW + 1023) / 1024 + (P + 255) / 256;
    const size_t bitmap_floats = (size_t)B * H * ((W + 63) / 64) * 2;
    return (size_t)B * HW * 2 + (size_t)B * nblk + (size_t)B + 64 + bitmap_floats + 16;
}

// Workspace layout shared by the two halves of the mesh loss
struct MeshWs {
    float* pts;
    float* partial;
    int* counts;
    unsigned long long* bits;
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
    return m;
}

// Step-invariant half: the ground-truth silhouette does not change between the IEF stages of one step (src/trainer.py:285-296
// evaluates the loss of every stage against the same seg_gts), so its compaction (tf.where order) and its bitmap are built once.
hipError_t hpe_launch_mesh_loss_prepare(const float* seg, int B, int H, int W, int P, float* ws, hipStream_t st) {
    const MeshWs m = mesh_ws_layout(ws, B, H, W, P);
    const int HW = H * W;
    hipLaunchKernelGGL(sil_compact_kernel, dim3(B), dim3(256), 0, st, seg, HW, W, m.pts, m.counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !m.grid_path) return e;
    const long nwords = (long)B * H * m.WW;
    hipLaunchKernelGGL(sil_bitmap_kernel, dim3((unsigned)((nwords + 3) / 4)), dim3(256), 0, st, seg, H, W, m.WW, m.bits, nwords);
    return hipGetLastError();
}

// Per-stage half: both nearest-neighbour searches against the prepared silhouette + the reduction.
hipError_t hpe_launch_mesh_loss_search(const float* v2d, int B, int H, int W, int P, float* ws, float* out, hipStream_t st,
                                       hipEvent_t ev_a2b0, hipEvent_t ev_a2b1) {
    const MeshWs m = mesh_ws_layout(ws, B, H, W, P);
    const int HW = H * W;
    static const int a2b_valu = [] {
        const char* e = getenv("HPE_MESH_A2B");  // "valu": the VALU-only search (A/B comparisons)
        return (e && e[0] == 'v') ? 1 : 0;
    }();
    if (ev_a2b0) (void)hipEventRecord(ev_a2b0, st);
    if (a2b_valu)
        hipLaunchKernelGGL(nn_a2b_kernel, dim3(m.nA, B), dim3(256), 0, st, m.pts, m.counts, v2d, HW, P, m.partial, m.nblk);
    else
        hipLaunchKernelGGL(nn_a2b_mfma_kernel, dim3(m.nA, B), dim3(256), 0, st, m.pts, m.counts, v2d, HW, P, m.partial, m.nblk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ev_a2b1) (void)hipEventRecord(ev_a2b1, st);
    if (m.grid_path) {
        hipLaunchKernelGGL(nn_b2a_rows_kernel, dim3(m.nB, B), dim3(256), (size_t)H * m.WW * 8, st, m.bits, m.counts, v2d, H, W, m.WW, P,
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
                                hipStream_t st) {
    hipError_t e = hpe_launch_mesh_loss_prepare(seg, B, H, W, P, ws, st);
    if (e != hipSuccess) return e;
    return hpe_launch_mesh_loss_search(v2d, B, H, W, P, ws, out, st, nullptr, nullptr);
}
