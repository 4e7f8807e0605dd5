// encoder_ops.hip -- the HBM-bound glue ops of the ResNet-50 v1 encoder (reference: src/models.py:35-41 ->
// keras.applications.ResNet50): conv1_pad ZeroPadding2D(3) (+ channel pad 3->4 so that one 7-tap kernel
// row is 8 px * 4 ch = one contiguous 32-float GEMM slab), pool1_pad ZeroPadding2D(1) + MaxPooling2D(3,2),
// and the final GlobalAveragePooling2D.  All are 16 B/lane coalesced NHWC streams.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpe_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// img [B,H,W,3] -> out [B,Hp,Wp,4]; out(b, y+3, x+3, c<3) = img(b,y,x,c); everything else 0.
__global__ void pad_input_kernel(const float* __restrict__ img, f32x4* __restrict__ out, int B, int H, int W, int Hp, int Wp) {
    const long total = (long)B * Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const long r = i / Wp;
        const int yp = (int)(r % Hp);
        const int b = (int)(r / Hp);
        const int x = xp - 3, y = yp - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const float* s = img + (((long)b * H + y) * W + x) * 3;
            v.x = s[0];
            v.y = s[1];
            v.z = s[2];
        }
        out[i] = v;
    }
}

// x [B,H,H,C] (post-ReLU conv1) -> y [B,H/2,H/2,C]: zero pad 1 (the pad value takes part in the max,
// exactly as ZeroPadding2D + 'valid' MaxPooling2D does), 3x3 window, stride 2.
__global__ void maxpool3x3s2_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int B, int H, int C4) {
    const int Ho = H / 2;
    const long total = (long)B * Ho * Ho * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int wo = (int)(r % Ho);
        r /= Ho;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        f32x4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * ho - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * wo - 1 + dx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)H) v = x[(((long)b * H + yy) * H + xx) * C4 + c];
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
            }
        }
        y[i] = m;
    }
}

// x [B,HW,C] -> y [B, ldy] (first C columns): mean over HW.
__global__ void avgpool_kernel(const f32x4* __restrict__ x, float* __restrict__ y, int B, int HW, int C4, int ldy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C4) return;
    const int c = i % C4;
    const int b = i / C4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = x + (long)b * HW * C4 + c;
    for (int k = 0; k < HW; ++k) s += p[(long)k * C4];
    const float inv = 1.0f / (float)HW;
    f32x4 o = s * inv;
    *reinterpret_cast<f32x4*>(y + (long)b * ldy + c * 4) = o;
}

// The same for small batches (the latency path): 8 threads share a channel quad, each adds every 8th pixel (independent loads in
// flight instead of a chain of HW dependent ones: 13 -> 4 us for one image), shuffle reduction in the fixed order of the xor tree.
__global__ __launch_bounds__(256) void avgpool_small_kernel(const f32x4* __restrict__ x, float* __restrict__ y, int B, int HW, int C4, int ldy) {
    const int q = blockIdx.x * 32 + (threadIdx.x >> 3);  // (image, channel quad)
    const int kg = threadIdx.x & 7;
    const bool ok = q < B * C4;
    const int c = ok ? q % C4 : 0;
    const int b = ok ? q / C4 : 0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = x + (long)b * HW * C4 + c;
#pragma unroll 8
    for (int k = kg; k < HW; k += 8) s += p[(long)k * C4];
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        s.x += __shfl_xor(s.x, off, 64);
        s.y += __shfl_xor(s.y, off, 64);
        s.z += __shfl_xor(s.z, off, 64);
        s.w += __shfl_xor(s.w, off, 64);
    }
    if (ok && kg == 0) *reinterpret_cast<f32x4*>(y + (long)b * ldy + c * 4) = s * (1.0f / (float)HW);
}

// Dense layer for very small batches (M <= 4; from M = 8 on the split-K GEMM + fix-up pair is faster again): y[m][n] = act((x[m,:] . Wt[n,:]) * scale[n] + shift[n] + res[m][n]).
// One wave per output column n (a contiguous K-float row of the packed weights, read once, 16 B per lane), all M rows of x at
// once, wave-shuffle reduction.  ONE launch where the implicit-GEMM kernel needs a split-K launch plus a fix-up launch to occupy the
// chip at M <= 8: the single-frame path is a chain of dependent ~8-us launches, so launches are what it pays for
// (RegressionNetwork, reference: src/models.py:60-74; predictor loop src/predictor.py:129-133).
template <int MM>
__global__ __launch_bounds__(256) void dense_gemv_kernel(const float* __restrict__ x, int lda, const float* __restrict__ wt, int K, int N,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ res, int ldres, int relu, float* __restrict__ y, int ldy, int M) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    const float* w = wt + (size_t)n * K;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m < M) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * lda + k);
                acc[m] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        float v = acc[m];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && m < M) {
            v = v * scale[n] + shift[n];
            if (res) v += res[(size_t)m * ldres + n];
            if (relu) v = fmaxf(v, 0.f);
            y[(size_t)m * ldy + n] = v;
        }
    }
}

// theta[b][0..84] = mean[0..84]  (tf.tile(mean_var, [B,1]); reference: src/predictor.py:126)
__global__ void tile_theta_kernel(const float* __restrict__ mean, float* __restrict__ theta, int B, int ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 85) return;
    const int b = i / 85, k = i - b * 85;
    theta[(long)b * ld + k] = mean[k];
}

__global__ void copy_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int B, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    const int b = i / n, k = i - b * n;
    dst[(long)b * ldd + k] = src[(long)b * lds + k];
}

inline int grid_for(long total, int block, int cap = 256 * 8) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

hipError_t hpe_launch_pad_input(const float* img, float* out, int B, int H, int W, int Hp, int Wp, hipStream_t st) {
    const long total = (long)B * Hp * Wp;
    hipLaunchKernelGGL(pad_input_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, img, reinterpret_cast<f32x4*>(out), B, H,
                       W, Hp, Wp);
    return hipGetLastError();
}

hipError_t hpe_launch_maxpool(const float* x, float* y, int B, int H, int C, hipStream_t st) {
    if ((C % 4) != 0 || (H % 2) != 0) return hipErrorInvalidValue;
    const long total = (long)B * (H / 2) * (H / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x),
                       reinterpret_cast<f32x4*>(y), B, H, C / 4);
    return hipGetLastError();
}

hipError_t hpe_launch_avgpool(const float* x, float* y, int B, int HW, int C, int ldy, hipStream_t st) {
    if ((C % 4) != 0 || (ldy % 4) != 0) return hipErrorInvalidValue;
    const int total = B * (C / 4);
    if (B <= 16) {
        hipLaunchKernelGGL(avgpool_small_kernel, dim3((total + 31) / 32), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), y, B, HW, C / 4, ldy);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(avgpool_kernel, dim3((total + 255) / 256), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), y, B, HW,
                       C / 4, ldy);
    return hipGetLastError();
}

hipError_t hpe_launch_dense_gemv(const float* x, int lda, int M, int K, const float* wt, int N, const float* scale, const float* shift, const float* res,
                                 int ldres, int relu, float* y, int ldy, hipStream_t st) {
    if (M < 1 || M > 4 || (K % 4) != 0 || (lda % 4) != 0 || N < 1 || !x || !wt || !y || !scale || !shift) return hipErrorInvalidValue;
    if (((uintptr_t)x & 15) != 0 || ((uintptr_t)wt & 15) != 0) return hipErrorInvalidValue;
    const dim3 grid((N + 3) / 4), block(256);
    if (M <= 2)
        hipLaunchKernelGGL(dense_gemv_kernel<2>, grid, block, 0, st, x, lda, wt, K, N, scale, shift, res, ldres, relu, y, ldy, M);
    else
        hipLaunchKernelGGL(dense_gemv_kernel<4>, grid, block, 0, st, x, lda, wt, K, N, scale, shift, res, ldres, relu, y, ldy, M);
    return hipGetLastError();
}

hipError_t hpe_launch_tile_theta(const float* mean85, float* theta, int B, int ld, hipStream_t st) {
    hipLaunchKernelGGL(tile_theta_kernel, dim3((B * 85 + 255) / 256), dim3(256), 0, st, mean85, theta, B, ld);
    return hipGetLastError();
}

hipError_t hpe_launch_copy_theta(const float* src, int lds, float* dst, int ldd, int B, int n, hipStream_t st) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3((B * n + 255) / 256), dim3(256), 0, st, src, lds, dst, ldd, B, n);
    return hipGetLastError();
}
