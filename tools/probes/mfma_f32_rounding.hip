// How does v_mfma_f32_32x32x2_f32 round?  Compares D = C + A0*B0 + A1*B1 from the matrix core with the two fused
// multiply-add orders on the VALU, over random operands of the magnitudes the mesh loss uses.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f32_rounding.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* ax, const float* ay, const float* px, const float* py, const float* c, int* counts) {
    const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
    const int base = blockIdx.x * 32;
    const float a = hi ? ay[base + l31] : ax[base + l31];
    const float b = hi ? py[base + l31] : px[base + l31];
    f32x16 cz;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) cz[4 * j + i] = c[base + 8 * j + 4 * hi + i];
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, cz, 0, 0, 0);
    int e_xy = 0, e_yx = 0, e_un = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            const int r = base + 8 * j + 4 * hi + i;  // vertex row
            const float X = ax[r], Y = ay[r], C = c[r], PX = px[base + l31], PY = py[base + l31];
            const float xy = fmaf(Y, PY, fmaf(X, PX, C));  // k = 0 first
            const float yx = fmaf(X, PX, fmaf(Y, PY, C));  // k = 1 first
            const float un = (float)((double)C + (double)X * PX + (double)Y * PY);  // single rounding
            e_xy += d[4 * j + i] != xy;
            e_yx += d[4 * j + i] != yx;
            e_un += d[4 * j + i] != un;
        }
    atomicAdd(&counts[0], e_xy);
    atomicAdd(&counts[1], e_yx);
    atomicAdd(&counts[2], e_un);
}
int main() {
    const int NB = 4096, N = NB * 32;
    float *h[5], *d[5];
    for (int k = 0; k < 5; ++k) h[k] = (float*)malloc(N * 4);
    srand(1);
    for (int i = 0; i < N; ++i) {
        const float bx = 224.f * rand() / RAND_MAX, by = 224.f * rand() / RAND_MAX;
        h[0][i] = -2.f * bx;
        h[1][i] = -2.f * by;
        h[2][i] = (float)(rand() % 224);
        h[3][i] = (float)(rand() % 224);
        h[4][i] = bx * bx + by * by;
    }
    for (int k = 0; k < 5; ++k) {
        hipMalloc(&d[k], N * 4);
        hipMemcpy(d[k], h[k], N * 4, hipMemcpyHostToDevice);
    }
    int* dc;
    hipMalloc(&dc, 12);
    hipMemset(dc, 0, 12);
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, 0, d[0], d[1], d[2], d[3], d[4], dc);
    int hc[3];
    hipMemcpy(hc, dc, 12, hipMemcpyDeviceToHost);
    printf("pairs %d  mismatches: fma(k0 then k1) %d   fma(k1 then k0) %d   single rounding %d\n", NB * 32 * 32, hc[0], hc[1], hc[2]);
    return 0;
}
