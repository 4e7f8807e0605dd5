// stem_fused.hip -- the stem of the Keras ResNet-50 v1 the reference instantiates (src/models.py:35-41 ->
// keras_applications resnet50.py: conv1_pad ZeroPadding2D(3) -> conv1 7x7/2 (+bias) -> bn_conv1 -> ReLU -> pool1_pad
// ZeroPadding2D(1) -> MaxPooling2D(3, strides 2)) as ONE kernel: [B,224,224,3] fp32 images in, [B,56,56,64] out.
// Nothing of the 112x112x64 conv1 map (0.82 GB at B = 256 in fp32) ever reaches HBM, and there is no separate pad pass.
//
// A workgroup (8 waves) owns one strip of an image: R pooled rows = 2R (+1 halo) conv rows.  It stages the 4R+7 padded
// input rows it needs in LDS once (zero borders written here: that IS conv1_pad) and then runs an implicit GEMM whose A
// operand is read straight from that LDS image -- no per-slab global traffic at all -- against weights held in registers:
//   fp32:  v_mfma_f32_16x16x4_f32 (M = 112 = 7 x 16 conv pixels of a row: no padding rows, K packed 3 channels tight).
//          k enumeration: kh-major, 22 slots per kernel row = {1 lead float (weight 0), 7 px x 3 ch}, so that every lane's
//          2-float fragment (ds_read_b64) stays inside one input row and is 8-B aligned: 7 x 22 = 154 -> 160 = 20 chunks of 8.
//          91.9 % of the issued MACs are real ones (the im2col GEMM this replaces: K padded 147 -> 224 = 65.6 %).
//   bf16:  v_mfma_f32_16x16x32_bf16; the LDS image is bf16 with the channel padded 3 -> 4, so one kernel row of a pixel pair
//          is one aligned ds_read_b128 and one MFMA k-step is one kernel row (8 px x 4 ch = 32 k); the matrix pipe is idle
//          most of the time anyway -- this variant is bound by the image read and the LDS traffic.
// A workgroup has 8 waves: wave w owns output channels 16 (w & 3) .. + 15 and the conv pixels 0-63 (w < 4: 4 blocks of 16) or
// 64-111 (w >= 4: 3 blocks) of a conv row pair; the two waves that share a SIMD cover each other's LDS latency (one wave per SIMD
// ran the fp32 loop at 66 % of the matrix rate).  The conv rows 2py and 2py+1 of pooled row py sit in the same lanes / register slots as row 2py-1 kept from the previous
// iteration, so the vertical 3-max is register-wise; the maximum goes through one LDS buffer for the horizontal 3-max
// (stride 2) and leaves as full NHWC rows.  ReLU output is >= 0, so pool1_pad's zeros never win and are not materialised.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpe_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int IMG = 224;        // input side
constexpr int CONV = 112;       // conv1 output side
constexpr int POOL = 56;        // pooled output side
constexpr int NCH = 64;         // conv1 output channels
constexpr int PITCH_F = 696;    // fp32 image row in LDS: 3 lead floats + 230 px x 3 ch + 3 tail floats (multiple of 4)
constexpr int PITCH_B = 464;    // bf16 image row in LDS, in floats: 232 px x 4 ch x 2 B = 1856 B
constexpr int VP = 68;          // V buffer row pitch in floats (64 channels + 4: the 4 lane groups of a store hit 4 bank sets)
constexpr int KCH = 20;         // fp32: chunks of 8 k slots (160 >= 154)

struct StemArgs {
    const float* img;    // [B,224,224,3] fp32
    const void* w;       // fp32: [64][160] floats in the k enumeration above;  bf16: [64][7][32] bf16 (kh, then 8 px x 4 ch)
    const float* scale;  // [64] folded BN
    const float* shift;
    void* y;             // [B,56,56,64] fp32 or bf16
    int B, R, strips;    // pooled rows per strip, strips per image (R * strips == 56)
};

// horizontal 3-max (stride 2, left pad = 0 which never wins after ReLU) of the V buffer -> one pooled NHWC row
template <bool BF16>
__device__ __forceinline__ void pool_store(const float* sV, void* y, int b, int py, int t) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = t + 512 * it;
        if (idx < POOL * 16) {
            const int px = idx >> 4;
            const int c = (idx & 15) * 4;
            f32x4 m = *reinterpret_cast<const f32x4*>(sV + (2 * px) * VP + c);
            const f32x4 r = *reinterpret_cast<const f32x4*>(sV + (2 * px + 1) * VP + c);
            m.x = fmaxf(m.x, r.x);
            m.y = fmaxf(m.y, r.y);
            m.z = fmaxf(m.z, r.z);
            m.w = fmaxf(m.w, r.w);
            if (px > 0) {
                const f32x4 l = *reinterpret_cast<const f32x4*>(sV + (2 * px - 1) * VP + c);
                m.x = fmaxf(m.x, l.x);
                m.y = fmaxf(m.y, l.y);
                m.z = fmaxf(m.z, l.z);
                m.w = fmaxf(m.w, l.w);
            }
            const size_t o = (((size_t)b * POOL + py) * POOL + px) * NCH + c;
            if (BF16) {
                bf16x4 v;
                v[0] = (__bf16)m.x;
                v[1] = (__bf16)m.y;
                v[2] = (__bf16)m.z;
                v[3] = (__bf16)m.w;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + o) = v;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + o) = m;
            }
        }
    }
}

// BN + ReLU of the row pair in acc, vertical 3-max with the row kept from the previous iteration, V -> LDS.
// PAIR == false: only acc[1] holds a conv row (the halo row 2 r0 - 1): it just becomes `prev`.
template <bool PAIR, int RB0, int NRB>
__device__ __forceinline__ void bn_relu_vmax(f32x4 (&acc)[2][NRB], f32x4 (&prev)[NRB], float sc, float sh, float* sV, int lane, int wave) {
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v1 = fmaxf(acc[1][rb][i] * sc + sh, 0.f);
            if (PAIR) {
                const float v0 = fmaxf(acc[0][rb][i] * sc + sh, 0.f);
                const float v = fmaxf(fmaxf(prev[rb][i], v0), v1);
                sV[(16 * (RB0 + rb) + 4 * g + i) * VP + 16 * (wave & 3) + m] = v;  // C layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + i
            }
            prev[rb][i] = v1;
        }
    }
}

// --------------------------------------------------------------------------------------------------------- fp32
// main part of a wave: conv pixels 16 RB0 .. 16 (RB0 + NRB) - 1 of every conv row, channels 16 (wave & 3) .. + 15
template <int RB0, int NRB>
__device__ __forceinline__ void stem_rows_f32(const StemArgs& p, const float* sIn, float* sV, int b, int r0, int t, int lane, int wave) {
    const int m = lane & 15, g = lane >> 4;
    const int ch = 16 * (wave & 3) + m;
    // ---- weights of this wave's 16 channels into registers: lane (col = m, k group g) holds slots 8p + 2g, 8p + 2g + 1
    f32x2 wb[KCH];
    {
        const float* wrow = reinterpret_cast<const float*>(p.w) + (size_t)ch * (8 * KCH) + 2 * g;
#pragma unroll
        for (int pc = 0; pc < KCH; ++pc) wb[pc] = *reinterpret_cast<const f32x2*>(wrow + 8 * pc);
    }
    const float sc = p.scale[ch], sh = p.shift[ch];
    // per-lane offset of k chunk pc inside the image: slot k' = 8 pc + 2 g -> (kh = k' / 22, j = k' % 22) -> kh * PITCH + j;
    // slots >= 154 carry zero weights: they re-read slot 0 (finite data) instead of running past the staged rows
    int off[KCH];
#pragma unroll
    for (int pc = 0; pc < KCH; ++pc) {
        const int k = 8 * pc + 2 * g;
        const int kh = k / 22;
        off[pc] = (k < 154) ? kh * PITCH_F + (k - 22 * kh) : 0;
    }
    // pixel wo = 16 rb + m of a conv row: window slot j is float 2 + 6 wo + j of the staged row (lead = 3, slot 0 = float before)
    const float* abase = sIn + 2 + 6 * (16 * RB0 + m);

    f32x4 acc[2][NRB], prev[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) prev[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // conv row hl (local: conv row 2 r0 - 1 + hl) reads staged rows 2 hl .. 2 hl + 6
    auto conv_rows = [&](int hl0, bool both) {
#pragma unroll
        for (int cr = 0; cr < 2; ++cr)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[cr][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* a0 = abase + (2 * (hl0 + 1)) * PITCH_F;  // staged row of conv row hl0 + 1 (cr = 1); cr = 0 is 2 rows up
#pragma unroll
        for (int pc = 0; pc < KCH; ++pc) {
            const float* ap = a0 + off[pc];
#pragma unroll
            for (int cr = 0; cr < 2; ++cr) {
                if (cr == 0 && !both) continue;
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const f32x2 a = *reinterpret_cast<const f32x2*>(ap + (cr - 1) * (2 * PITCH_F) + rb * 96);
                    acc[cr][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wb[pc].x, acc[cr][rb], 0, 0, 0);
                    acc[cr][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wb[pc].y, acc[cr][rb], 0, 0, 0);
                }
            }
        }
    };

    if (r0 > 0) {  // halo: conv row 2 r0 - 1 (local 0) into acc[1] (for r0 == 0 it is pool1_pad's zero row)
        conv_rows(-1, false);
        bn_relu_vmax<false, RB0, NRB>(acc, prev, sc, sh, sV, lane, wave);
    }
    for (int pyl = 0; pyl < p.R; ++pyl) {
        conv_rows(2 * pyl + 1, true);  // local rows 2 pyl + 1, 2 pyl + 2 = conv rows 2 py, 2 py + 1
        bn_relu_vmax<true, RB0, NRB>(acc, prev, sc, sh, sV, lane, wave);
        __syncthreads();
        pool_store<false>(sV, p.y, b, r0 + pyl, t);
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 2) void stem_fused_f32_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = 4 * p.R + 7;
    float* sIn = lds;
    float* sV = lds + rows * PITCH_F;

    const int b = blockIdx.x / p.strips;
    const int strip = blockIdx.x - b * p.strips;
    const int r0 = strip * p.R;  // first pooled row of the strip
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- stage the padded input rows: staged row s = padded row 4 r0 - 2 + s = raw row 4 r0 - 5 + s; float4 slot q4 of a row
    //      holds floats 4 q4 .. 4 q4 + 3, raw float4 q (pixels 4q/3 ...) lands in slot q + 3 (3 lead floats + 3 pad pixels = 12)
    {
        const int total = rows * (PITCH_F / 4);
        const f32x4* img4 = reinterpret_cast<const f32x4*>(p.img) + (size_t)b * IMG * (IMG * 3 / 4);
#pragma unroll 8
        for (int idx = t; idx < total; idx += 512) {
            const int s = idx / (PITCH_F / 4);
            const int q4 = idx - s * (PITCH_F / 4);
            const int raw = 4 * r0 - 5 + s;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)raw < (unsigned)IMG && q4 >= 3 && q4 < 3 + IMG * 3 / 4) v = img4[raw * (IMG * 3 / 4) + (q4 - 3)];
            *reinterpret_cast<f32x4*>(sIn + s * PITCH_F + 4 * q4) = v;
        }
    }
    __syncthreads();
    if (wave < 4)
        stem_rows_f32<0, 4>(p, sIn, sV, b, r0, t, lane, wave);
    else
        stem_rows_f32<4, 3>(p, sIn, sV, b, r0, t, lane, wave);
}

// --------------------------------------------------------------------------------------------------------- bf16
template <int RB0, int NRB>
__device__ __forceinline__ void stem_rows_bf16(const StemArgs& p, const float* sIn, float* sV, int b, int r0, int t, int lane, int wave) {
    const int m = lane & 15, g = lane >> 4;
    const int ch = 16 * (wave & 3) + m;
    // ---- weights: lane (col = m, pixel pair g) holds for each kernel row kh the 8 values of pixels 2g, 2g + 1 (4 ch each)
    bf16x8 wb[7];
    {
        const bf16x8* wrow = reinterpret_cast<const bf16x8*>(p.w) + (size_t)ch * 28 + g;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) wb[kh] = wrow[4 * kh];
    }
    const float sc = p.scale[ch], sh = p.shift[ch];
    // pixel wo = 16 rb + m, pixel pair g: 16 B at padded column 2 wo + 2 g of the staged row
    const float* abase = sIn + 4 * (16 * RB0 + m + g);

    f32x4 acc[2][NRB], prev[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) prev[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto conv_rows = [&](int hl0, bool both) {
#pragma unroll
        for (int cr = 0; cr < 2; ++cr)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[cr][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* a0 = abase + (2 * (hl0 + 1)) * PITCH_B;  // staged row of conv row hl0 + 1 (cr = 1)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
            for (int cr = 0; cr < 2; ++cr) {
                if (cr == 0 && !both) continue;
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a0 + (kh + 2 * (cr - 1)) * PITCH_B + rb * 64);
                    acc[cr][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wb[kh], acc[cr][rb], 0, 0, 0);
                }
            }
        }
    };

    if (r0 > 0) {
        conv_rows(-1, false);
        bn_relu_vmax<false, RB0, NRB>(acc, prev, sc, sh, sV, lane, wave);
    }
    for (int pyl = 0; pyl < p.R; ++pyl) {
        conv_rows(2 * pyl + 1, true);
        bn_relu_vmax<true, RB0, NRB>(acc, prev, sc, sh, sV, lane, wave);
        __syncthreads();
        pool_store<true>(sV, p.y, b, r0 + pyl, t);
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 2) void stem_fused_bf16_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = 4 * p.R + 7;
    float* sIn = lds;  // bf16 image [rows][232 px][4 ch]
    float* sV = lds + rows * PITCH_B;

    const int b = blockIdx.x / p.strips;
    const int strip = blockIdx.x - b * p.strips;
    const int r0 = strip * p.R;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- zero the image (borders + rows outside the picture), then copy: 4 raw pixels (3 aligned float4) -> 4 x 8 B
    {
        const int total = rows * (PITCH_B / 4);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int idx = t; idx < total; idx += 512) reinterpret_cast<f32x4*>(sIn)[idx] = z;
    }
    __syncthreads();
    {
        const int total = rows * (IMG / 4);
        const f32x4* img4 = reinterpret_cast<const f32x4*>(p.img) + (size_t)b * IMG * (IMG * 3 / 4);
#pragma unroll 4
        for (int idx = t; idx < total; idx += 512) {
            const int s = idx / (IMG / 4);
            const int u = idx - s * (IMG / 4);
            const int raw = 4 * r0 - 5 + s;
            if ((unsigned)raw < (unsigned)IMG) {
                const f32x4* src = img4 + raw * (IMG * 3 / 4) + 3 * u;
                const f32x4 v0 = src[0], v1 = src[1], v2 = src[2];
                const float px[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
                // raw pixel 4u + i = padded column 4u + 3 + i, 8 B per pixel
                __bf16* dst = reinterpret_cast<__bf16*>(sIn) + ((size_t)s * 232 + 4 * u + 3) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf16x4 o;
                    o[0] = (__bf16)px[3 * i];
                    o[1] = (__bf16)px[3 * i + 1];
                    o[2] = (__bf16)px[3 * i + 2];
                    o[3] = (__bf16)0.f;
                    *reinterpret_cast<bf16x4*>(dst + 4 * i) = o;
                }
            }
        }
    }
    __syncthreads();
    if (wave < 4)
        stem_rows_bf16<0, 4>(p, sIn, sV, b, r0, t, lane, wave);
    else
        stem_rows_bf16<4, 3>(p, sIn, sV, b, r0, t, lane, wave);
}

}  // namespace

size_t hpe_stem_fused_lds_bytes(int R, int bf16) { return ((size_t)(4 * R + 7) * (bf16 ? PITCH_B : PITCH_F) + (size_t)CONV * VP) * sizeof(float); }

// per-device attribute (dynamic LDS above 64 KB); call with the target device current
hipError_t hpe_stem_fused_init_device() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)hpe_stem_fused_lds_bytes(8, 0));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)hpe_stem_fused_lds_bytes(8, 1));
}

// pooled rows per strip: 8 (7 strips per image, halo recompute 1/16) once the grid fills the chip several times over,
// smaller strips for small batches (more workgroups, more halo)
int hpe_stem_fused_pick_rows(int B) {
    if (B >= 96) return 8;
    if (B >= 24) return 4;
    if (B >= 6) return 2;
    return 1;
}

hipError_t hpe_launch_stem_fused(const float* img, const void* w, const float* scale, const float* shift, void* y, int B, int R, int bf16,
                                 hipStream_t st) {
    if (!img || !w || !scale || !shift || !y || B < 1 || R < 1 || R > 8 || (POOL % R) != 0) return hipErrorInvalidValue;
    if (((uintptr_t)img & 15) != 0 || ((uintptr_t)y & 15) != 0 || ((uintptr_t)w & 15) != 0) return hipErrorInvalidValue;
    StemArgs p{};
    p.img = img;
    p.w = w;
    p.scale = scale;
    p.shift = shift;
    p.y = y;
    p.B = B;
    p.R = R;
    p.strips = POOL / R;
    const size_t ldsb = hpe_stem_fused_lds_bytes(R, bf16);
    if (bf16)
        hipLaunchKernelGGL(stem_fused_bf16_kernel, dim3(B * p.strips), dim3(512), ldsb, st, p);
    else
        hipLaunchKernelGGL(stem_fused_f32_kernel, dim3(B * p.strips), dim3(512), ldsb, st, p);
    return hipGetLastError();
}
