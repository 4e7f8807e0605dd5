// What does ONE dependent phase cost on this box?  The single-frame forward is a chain of ~95 dependent launches (53 layers, 40 split-K
// fix-ups, pools): this probe measures the floor of such a chain with nothing in the kernels, with the smallest dependent piece of work
// (one load -> store round trip per workgroup, one LDS-DMA slab chain) and with a grid-wide barrier inside one persistent launch instead
// of a kernel boundary -- the numbers DESIGN.md's single-frame section prices the "persistent kernel per bottleneck block" with.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_floor.hip -o gpurun_out/launch_floor && gpurun_out/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                         \
        }                                                                     \
    } while (0)

__global__ void k_empty() {}

// one dependent global round trip per workgroup: y = x + 1 over 256 floats per workgroup (x was written by the previous launch)
__global__ void k_touch(const float* __restrict__ x, float* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    y[i] = x[i] + 1.0f;
}

// a split-K slice as the GEMM kernel runs it: `slabs` dependent steps of {16 KB global -> LDS by LDS-DMA, wait, barrier, read back}
__global__ __launch_bounds__(256) void k_slabs(const float* __restrict__ x, float* __restrict__ y, int slabs) {
    __shared__ __attribute__((aligned(16))) float lds[2][4096];
    const int t = threadIdx.x, wave = t >> 6;
    const float* src = x + (size_t)blockIdx.x * 4096;
    float acc = 0.f;
    for (int s = 0; s < slabs; ++s) {
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((s & 7) * 256 * 4096) + (i * 4 + wave) * 256 + (t & 63) * 4),
                                             (__attribute__((address_space(3))) void*)(&lds[s & 1][(i * 4 + wave) * 256]), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += lds[s & 1][(t * 13) & 4095];
    }
    y[blockIdx.x * 256 + t] = acc;
}

// the same bytes with every slab requested up front (one wait, one barrier): what a split-K slice costs when its slab loads do not
// depend on each other.  Dynamic LDS: slabs x 16 KB.
__global__ __launch_bounds__(256) void k_slabs_upfront(const float* __restrict__ x, float* __restrict__ y, int slabs, int stride_slabs) {
    extern __shared__ __attribute__((aligned(16))) float dl[];
    const int t = threadIdx.x, wave = t >> 6;
    const float* src = x + (size_t)blockIdx.x * 4096;
    for (int s = 0; s < slabs; ++s)
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)(s % stride_slabs) * 256 * 4096) + (i * 4 + wave) * 256 + (t & 63) * 4),
                                             (__attribute__((address_space(3))) void*)(&dl[s * 4096 + (i * 4 + wave) * 256]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float acc = 0.f;
    for (int s = 0; s < slabs; ++s) acc += dl[s * 4096 + ((t * 13) & 4095)];
    y[blockIdx.x * 256 + t] = acc;
}

// persistent launch: `phases` phases separated by a grid-wide barrier (monotonic counter, relaxed agent-scope polling, bounded wait);
// one workgroup per CU, every workgroup does the k_touch work in each phase
__global__ __launch_bounds__(256) void k_persistent(float* __restrict__ a, float* __restrict__ b, unsigned* __restrict__ counter, int phases, unsigned* err) {
    const int nwg = gridDim.x;
    float* x = a;
    float* y = b;
    for (int ph = 0; ph < phases; ++ph) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        const float v = __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(y + i, v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(ph + 1) * nwg;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 200000) {  // bounded: a workgroup that is not resident must not hang the box
                    *err = 1;
                    break;
                }
            }
        }
        __syncthreads();
        float* tmp = x;
        x = y;
        y = tmp;
    }
}

int main() {
    int ncu = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    float *a, *b;
    unsigned* cnt;
    const size_t n = (size_t)8 * 256 * 4096 + 1024 * 256;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
    CK(hipMalloc(&cnt, 8));
    CK(hipMemset(a, 0, n * 4));
    CK(hipMemset(b, 0, n * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int N = 400;
    auto timeit = [&](const char* name, auto&& body, int reps) -> int {
        for (int w = 0; w < 2; ++w) {  // warm-up
            body();
            if (hipStreamSynchronize(st) != hipSuccess) return 1;
        }
        double best = 1e30;
        for (int r = 0; r < reps; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            body();
            if (hipStreamSynchronize(st) != hipSuccess) return 1;
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        printf("%-72s %8.2f us per phase (%d phases, best of %d)\n", name, best / N, N, reps);
        return 0;
    };
    for (int wgs : {64, 256, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "empty kernel, %d workgroups x 256 threads, eager launches", wgs);
        if (timeit(nm, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, st); }, 5)) return 1;
    }
    for (int wgs : {64, 256}) {
        char nm[128];
        snprintf(nm, sizeof nm, "y = x + 1 (one dependent global round trip), %d workgroups, eager", wgs);
        if (timeit(nm, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b); }, 5)) return 1;
    }
    for (int slabs : {1, 4, 8}) {
        char nm[128];
        snprintf(nm, sizeof nm, "split-K slice skeleton: %d dependent 16 KB LDS-DMA slabs + barrier, 256 workgroups", slabs);
        if (timeit(nm, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_slabs, dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, slabs); }, 5)) return 1;
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_slabs_upfront), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
    for (int slabs : {4, 8}) {
        char nm[128];
        snprintf(nm, sizeof nm, "the same %d slabs requested up front, one wait (working set 8 x 4 MB per buffer)", slabs);
        if (timeit(nm, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_slabs_upfront, dim3(256), dim3(256), slabs * 16384, st, (i & 1) ? b : a, (i & 1) ? a : b, slabs, 8); }, 5)) return 1;
    }
    {
        // the same chain replayed from a hipGraph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        if (timeit("y = x + 1, 256 workgroups, hipGraph replay", [&] { (void)hipGraphLaunch(ge, st); }, 5)) return 1;
    }
    {
        unsigned* err = cnt + 1;
        if (timeit("persistent launch: y = x + 1 + grid barrier (counter, relaxed poll), 1 WG per CU",
                   [&] {
                       (void)hipMemsetAsync(cnt, 0, 8, st);
                       hipLaunchKernelGGL(k_persistent, dim3(ncu), dim3(256), 0, st, a, b, cnt, N, err);
                   },
                   5))
            return 1;
        unsigned h[2];
        CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
        printf("  (grid barrier timeouts: %u)\n", h[1]);
    }
    return 0;
}
