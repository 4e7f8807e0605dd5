// prepost.hip -- the steps immediately before and after the hot path, on the GPU:
//   * preprocess: the webcam demo's preprocess_image (reference: preview.py:18-35) -> scale_and_crop / resize_img
//     (reference: src/util/image.py:7-39): resize so that max(H,W) == 224 (cv2.resize, bilinear), edge-pad by 112,
//     crop 224x224 around the scaled image centre, map uint8 to [-1,1] with 2*(x/255 - 0.5).  One fused HBM-bound
//     pass: no resized / padded intermediate exists.  The bilinear arithmetic restates OpenCV's 8-bit path
//     (INTER_LINEAR: half-pixel centres, 11-bit fixed-point coefficients, (((b0*(S0>>4))>>16)+((b1*(S1>>4))>>16)+2)>>2).
//     PARITY UNPINNED: cv2 is not installed here, so this is checked against a NumPy restatement of the same
//     published algorithm only (oracle/prepost_oracle.py).
//   * get_original (reference: src/util/renderer.py:260-283): camera conversion, vertex shift, keypoint un-crop.
#include <hip/hip_runtime.h>
#include <math.h>

#include "hpe_internal.h"

namespace {

struct ResizeAxis {
    int x0, x1;  // source taps (clamped)
    int a0, a1;  // 11-bit coefficients, a0 + a1 == 2048
};

// OpenCV resize(INTER_LINEAR) index/coefficient rule for one destination coordinate
__device__ __forceinline__ ResizeAxis axis_coef(int d, int dsize, int ssize) {
    const double scale = (double)ssize / (double)dsize;
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) {
        fx = 0.f;
        sx = 0;
    }
    if (sx >= ssize - 1) {
        fx = 0.f;
        sx = ssize - 1;
    }
    ResizeAxis r;
    r.x0 = sx;
    r.x1 = min(sx + 1, ssize - 1);
    const int c1 = (int)rintf(fx * 2048.f);  // saturate_cast<short>(fx * INTER_RESIZE_COEF_SCALE)
    r.a1 = c1;
    r.a0 = (int)rintf((1.f - fx) * 2048.f);
    return r;
}

// one output pixel of one frame (the body shared by the single-frame and the batched kernel)
__device__ __forceinline__ void preprocess_pixel(const unsigned char* __restrict__ img, int H, int W, int C, int newH, int newW, int start_x,
                                                 int start_y, int margin, float* __restrict__ out, int S, int i) {
    const int oy = i / S, ox = i - oy * S;
    // coordinate in the edge-padded scaled image -> clamp back into the scaled image (np.pad mode='edge')
    int sy = start_y + oy - margin, sx = start_x + ox - margin;
    sy = min(max(sy, 0), newH - 1);
    sx = min(max(sx, 0), newW - 1);
    float v[3];
    if (newH == H && newW == W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)img[((size_t)sy * W + sx) * C + c];
    } else {
        const ResizeAxis ax = axis_coef(sx, newW, W);
        const ResizeAxis ay = axis_coef(sy, newH, H);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s00 = img[((size_t)ay.x0 * W + ax.x0) * C + c], s01 = img[((size_t)ay.x0 * W + ax.x1) * C + c];
            const int s10 = img[((size_t)ay.x1 * W + ax.x0) * C + c], s11 = img[((size_t)ay.x1 * W + ax.x1) * C + c];
            const int S0 = s00 * ax.a0 + s01 * ax.a1;  // horizontal pass, 11 fractional bits
            const int S1 = s10 * ax.a0 + s11 * ax.a1;
            const int d = (((ay.a0 * (S0 >> 4)) >> 16) + ((ay.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
            v[c] = (float)min(max(d, 0), 255);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)i * 3 + c] = 2.0f * ((v[c] / 255.0f) - 0.5f);  // preview.py:33
}

// out [224,224,3] float; one thread per output pixel
__global__ void preprocess_u8_kernel(const unsigned char* __restrict__ img, int H, int W, int C, int newH, int newW, int start_x,
                                     int start_y, int margin, float* __restrict__ out, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    preprocess_pixel(img, H, W, C, newH, newW, start_x, start_y, margin, out, S, i);
}

// batch: grid (ceil(S*S / 256), B); frame geometry from the per-image table, or `uni` for a stream of equal frames
__global__ void preprocess_u8_batch_kernel(const unsigned char* __restrict__ img, const PreprocFrame* __restrict__ table, PreprocFrame uni,
                                           int C, int margin, float* __restrict__ out, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    const int b = blockIdx.y;
    PreprocFrame f = uni;
    if (table)
        f = table[b];
    else
        f.offset = (long long)b * uni.H * uni.W * C;
    preprocess_pixel(img + f.offset, f.H, f.W, C, f.newH, f.newW, f.start_x, f.start_y, margin, out + (size_t)b * S * S * 3, S, i);
}

// vert_shifted[b,v,:] = verts[b,v,:] + [tx, ty, tz],  tz = flength / (0.5 * img_size * s)   (renderer.py:266-271)
__global__ void shift_verts_kernel(const float* __restrict__ verts, const float* __restrict__ cam, int B, int P, float flength,
                                   float img_size, float* __restrict__ out) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / P);
        const float s = cam[b * 3], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
        const float tz = flength / (0.5f * img_size * s);
        out[i * 3] = verts[i * 3] + tx;
        out[i * 3 + 1] = verts[i * 3 + 1] + ty;
        out[i * 3 + 2] = verts[i * 3 + 2] + tz;
    }
}

}  // namespace

hipError_t hpe_launch_preprocess_u8(const unsigned char* img, int H, int W, int C, int newH, int newW, int start_x, int start_y,
                                    int margin, float* out, int S, hipStream_t st) {
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((S * S + 255) / 256), dim3(256), 0, st, img, H, W, C, newH, newW, start_x, start_y,
                       margin, out, S);
    return hipGetLastError();
}

hipError_t hpe_launch_preprocess_u8_batch(const unsigned char* img, const PreprocFrame* table_dev, PreprocFrame uni, int B, int C, int margin,
                                          float* out, int S, hipStream_t st) {
    hipLaunchKernelGGL(preprocess_u8_batch_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0, st, img, table_dev, uni, C, margin, out, S);
    return hipGetLastError();
}

hipError_t hpe_launch_shift_verts(const float* verts, const float* cam, int B, int P, float flength, float img_size, float* out,
                                  hipStream_t st) {
    long g = ((long)B * P + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(shift_verts_kernel, dim3((int)g), dim3(256), 0, st, verts, cam, B, P, flength, img_size, out);
    return hipGetLastError();
}
