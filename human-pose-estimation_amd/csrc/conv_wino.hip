// conv_wino.hip -- 3x3 / stride 1 / SAME convolutions of the ResNet-50 bottlenecks (res*_branch2b, reference:
// src/models.py:39 -> keras_applications resnet50 conv_block / identity_block) as Winograd F(2x2, 3x3) in fp32:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, summed over input channels
//
// 16 multiplies per tile and channel pair instead of 36: the MFMA work of these 16 layers (48 % of the encoder's
// FLOPs) shrinks 2.25x (1.72x on the 7x7 maps, whose 4x4 tile grid covers 8x8).  Arithmetic stays fp32 end to end.
//
// Two variants:
//  (a) 14x14 and 7x7 maps -- two kernels per layer:
//   wino_input_kernel   x [B,H,W,C] NHWC -> V, HBM bound (reads x once, writes 4x its size).  V is stored in the exact
//                       byte image the GEMM kernel's LDS staging wants, so that each k-slab of a workgroup is one
//                       contiguous 32 KB block:   V[tile block of 64][slab of 8 ch][comp 16][half 2][tile 64][4 ch]
//   wino_gemm_kernel    16 independent GEMMs  M_c[tile][cout] = sum_ch V_c[tile][ch] * U_c[cout][ch]  for a 64-tile x
//                       64-cout block, all 16 components in one workgroup (8 waves x 2 components x four 32x32 MFMA
//                       blocks = 128 accumulator VGPRs per lane), LDS-DMA double buffering (2 x 64 KB, two slabs in flight,
//                       the barrier in the middle of a slab's MFMA work), then the output
//                       transform A^T M A through LDS, BN scale/shift, ReLU and 16 B/lane NHWC stores.
//                       U = G g G^T is precomputed on the host in the same blocked layout.
//  (b) 56x56 and 28x28 maps -- wino_fused_kernel: the same GEMM with the input transform done inside, from activations the
//      producing 1x1 convolution wrote channel-slab major (see the comment at the kernel); no V in memory.
//
// Why (a) does not read NHWC activations directly: a slab can only hold 8 channels of 64 tiles x 16 components (LDS), i.e.
// 32 B of every 128 B activation line per pass, and the 1024 lines a workgroup touches per slab do not survive in the
// 32 KB L1 until the next slab -> 4x L2 traffic.  The blocked V removes the problem at the price of one extra HBM round
// trip that runs concurrently with another batch chunk's MFMA-bound GEMM (chunk streams); the slab-major producer of (b)
// removes it without that round trip, where whole tile rows of the batch fill a workgroup.
#include "hpe_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WT = 64;             // tiles per workgroup
constexpr int WN_ = 64;            // output channels per workgroup
constexpr int WK = 8;              // channels per slab
constexpr int OPER = 16 * 2 * 64 * 4;  // floats of one operand (V or U) per slab = 8192 (32 KB)
constexpr int WINO_LDS_BYTES = 2 * 2 * OPER * (int)sizeof(float);  // wino_gemm_kernel: 2 x (V slab + U slab) = 128 KB

// ------------------------------------------------------------------------------------------------ input transform
// one thread = (tile, 4 channels); a wave = 8 tiles x 32 channels (full 128-B lines on the read side, 128-B
// segments on the write side)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C,
                                                         int TW, int TT, int T, int Tpad, int lda) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int lane = gid & 63;
    const int wv = gid >> 6;
    const int ncb = C >> 5;
    const int tg = wv / ncb;
    const int cb = wv - tg * ncb;
    const int t = tg * 8 + (lane >> 3);
    if (t >= Tpad) return;
    const int c = cb * 32 + (lane & 7) * 4;
    const int S = C >> 3;
    float* dst = V + ((((size_t)(t >> 6) * S + (c >> 3)) * 16) * 2 + ((c >> 2) & 1)) * 256 + (t & 63) * 4;
    if (t >= T) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4*>(dst + k * 512) = z;
        return;
    }
    const int b = t / TT;
    const int rem = t - b * TT;
    const int ty = rem / TW;
    const int tx = rem - ty * TW;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int iy = y0 + a;
        const bool oky = iy >= 0 && iy < H;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = x0 + e;
            const bool ok = oky && ix >= 0 && ix < W;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            d[a][e] = ok ? *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + iy) * W + ix) * lda + c) : z;
        }
    }
    // B^T d (rows), then (.) B (columns);  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    f32x4 r[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[0][e] = d[0][e] - d[2][e];
        r[1][e] = d[1][e] + d[2][e];
        r[2][e] = d[2][e] - d[1][e];
        r[3][e] = d[1][e] - d[3][e];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        *reinterpret_cast<f32x4*>(dst + (a * 4 + 0) * 512) = r[a][0] - r[a][2];
        *reinterpret_cast<f32x4*>(dst + (a * 4 + 1) * 512) = r[a][1] + r[a][2];
        *reinterpret_cast<f32x4*>(dst + (a * 4 + 2) * 512) = r[a][2] - r[a][1];
        *reinterpret_cast<f32x4*>(dst + (a * 4 + 3) * 512) = r[a][1] - r[a][3];
    }
}

// ------------------------------------------------------------------------------------------------ 16-component GEMM + output transform
struct WinoArgs {
    const float* V;
    const float* U;
    const float* scale;
    const float* shift;
    float* y;
    int H, W, N;      // output map, output channels
    int TW, TT, T;    // tiles per row / per image / total
    int S;            // k-slabs (C / 8)
    int n_tb, n_nt;   // tile blocks, cout blocks
    int ldy, relu;
    // stream-K only
    float* ws;            // [workgroups][128 accumulator floats x 512 threads]
    unsigned* flags;      // [workgroups], holds the epoch of the launch that parked data
    unsigned epoch;
    unsigned* err;        // device error word of the ctx: bit 0 = a stream-K wait timed out (hpe_device_status reports it)
};

// main loop over k-slabs [k0, k1) of work item (tb, nt); accumulators are added to (caller zeroes them)
__device__ __forceinline__ void wino_mainloop(const WinoArgs& p, float* lds, int tb, int nt, int k0, int k1, f32x16 (&acc)[2][2][2], int wave,
                                              int lane) {
    const int hi = lane >> 5;
    const float* vsrc = p.V + ((size_t)tb * p.S + k0) * OPER + wave * 1024 + lane * 4;
    const float* usrc = p.U + ((size_t)nt * p.S + k0) * OPER + wave * 1024 + lane * 4;

    auto issue = [&](int s, int buf) {
        float* dstv = lds + buf * (2 * OPER) + wave * 1024;
        const float* sv = vsrc + (size_t)s * OPER;
        const float* su = usrc + (size_t)s * OPER;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sv + i * 256),
                                             (__attribute__((address_space(3))) void*)(dstv + i * 256), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(su + i * 256),
                                             (__attribute__((address_space(3))) void*)(dstv + OPER + i * 256), 16, 0, 0);
    };

    // fragment offsets (floats) inside an operand slab: [comp][half][row][4]
    const int frag = ((2 * wave) * 2 + hi) * 256 + (lane & 31) * 4;
    const int n = k1 - k0;

    // Two slabs in flight.  Per slab s: [MFMAs of component 0] [barrier] [DMA of slab s+2 into the buffer slab s just vacated]
    // [MFMAs of component 1].  The barrier sits in the middle of the MFMA work, after this wave has pulled both components'
    // fragments of slab s into registers, so (a) every wave is done reading buffer s&1 when the DMA of slab s+2 overwrites it
    // and (b) that DMA has a whole slab of MFMA time (16 + 16 MFMAs) to land before barrier s+1 waits for it.
    issue(0, 0);
    if (n > 1) issue(1, 1);
    if (n > 1)
        __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8): slab 0 landed, slab 1 (the 8 newest DMA instructions) may be in flight
    else
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("" ::: "memory");  // no LDS access of the loop may be scheduled above this barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int s = 0; s < n; ++s) {
        const int cur = (s & 1) * (2 * OPER);
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[c][i] = *reinterpret_cast<const f32x4*>(&lds[cur + frag + c * 512 + i * 128]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[c][j] = *reinterpret_cast<const f32x4*>(&lds[cur + OPER + frag + c * 512 + j * 128]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][i][ks], fb[0][j][ks], acc[0][i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // my part of slab s+1 has landed: written out, not left to __syncthreads() (which guarantees lgkmcnt(0) -- my fragments of slab s
        // are in registers -- but adds a vmcnt wait only behind some DMA patterns; tools/isa_lint.py checks the ISA)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        if (s + 2 < n) issue(s + 2, s & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][i][ks], fb[1][j][ks], acc[1][i][j], 0, 0, 0);
    }
    __syncthreads();  // all waves out of the last slab before the epilogue reuses the LDS
}

// epilogue: M[comp][tile][cout] of one 32-tile half through LDS, output transform, BN, ReLU, NHWC stores.
// Entered with all waves past the main loop's last barrier (LDS free); leaves with a barrier pending only for readers of
// the second half, so the caller must __syncthreads() before the next LDS write.
__device__ __forceinline__ void wino_epilogue(const WinoArgs& p, float* lds, int tb, int nt, f32x16 (&acc)[2][2][2], int t, int wave, int lane) {
    const int hi = lane >> 5;
    const int n0 = nt * WN_;
    const int em = t >> 4;          // tile within the half
    const int eq = (t & 15) * 4;    // cout quad
    const int n = n0 + eq;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    lds[((2 * wave + c) * 32 + m) * 64 + 32 * j + (lane & 31)] = acc[c][i][j][e];
                }
        __syncthreads();
        const int tg = tb * WT + 32 * i + em;
        if (tg < p.T) {
            f32x4 M[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) M[c] = *reinterpret_cast<const f32x4*>(&lds[(c * 32 + em) * 64 + eq]);
            // A^T = [1 1 1 0; 0 1 -1 -1]
            f32x4 u0[4], u1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                u0[v] = M[v] + M[4 + v] + M[8 + v];
                u1[v] = M[4 + v] - M[8 + v] - M[12 + v];
            }
            f32x4 o[2][2];
            o[0][0] = u0[0] + u0[1] + u0[2];
            o[0][1] = u0[1] - u0[2] - u0[3];
            o[1][0] = u1[0] + u1[1] + u1[2];
            o[1][1] = u1[1] - u1[2] - u1[3];
            const int b = tg / p.TT;
            const int rem = tg - b * p.TT;
            const int ty = rem / p.TW;
            const int tx = rem - ty * p.TW;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oy = 2 * ty + dy, ox = 2 * tx + dx;
                    if (oy < p.H && ox < p.W) {
                        f32x4 v = o[dy][dx] * sc + sh;
                        if (p.relu) {
                            v.x = fmaxf(v.x, 0.f);
                            v.y = fmaxf(v.y, 0.f);
                            v.z = fmaxf(v.z, 0.f);
                            v.w = fmaxf(v.w, 0.f);
                        }
                        *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * p.ldy + n) = v;
                    }
                }
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2][2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][i][j][e] = 0.f;
}

// one work item (64 tiles x 64 couts, all k-slabs) per workgroup: launches that cannot fill the persistent grid
__global__ __launch_bounds__(512, 2) void wino_gemm_kernel(WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x (V slab + U slab) = 128 KB; reused by the epilogue

    const int total = p.n_tb * p.n_nt;
    const int bid = blockIdx.x;
    // XCD-aware bijective remap: consecutive logical ids (the cout blocks of one tile block) share an XCD and its L2
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tb = lid / p.n_nt;
    const int nt = lid - tb * p.n_nt;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    f32x16 acc[2][2][2];
    zero_acc(acc);
    wino_mainloop(p, lds, tb, nt, 0, p.S, acc, wave, lane);
    wino_epilogue(p, lds, tb, nt, acc, t, wave, lane);
}

// Persistent stream-K variant: one workgroup per CU.  Workgroups form teams of n_nt members (member j computes cout block j,
// so a team streams each V block once through the XCD's L2 while its members read it concurrently); a team walks a
// contiguous range of (tile block, k-slab) units of equal length for every team, so there is no partial last round of
// workgroups.  A tile block cut by a range boundary is computed in two parts: the next team does slabs [k, S) as the FIRST
// thing in its life and parks the raw accumulators in `ws`; this team reaches slabs [0, k) LAST, adds the parked part and
// runs the epilogue -- the flag it polls was normally set long before.  The XCD L2s are not coherent with each other, so
// the parked accumulators and the flag move with relaxed agent-scope atomics (sc1 accesses, coherent by themselves).
__global__ __launch_bounds__(512, 2) void wino_gemm_streamk_kernel(WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    const int nwg = gridDim.x;                   // multiple of 8 * n_nt
    const int per_xcd = nwg >> 3;
    const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // teams are contiguous inside an XCD
    const int team = lid / p.n_nt;
    const int nt = lid - team * p.n_nt;
    const int n_teams = nwg / p.n_nt;
    const long units = (long)p.n_tb * p.S;
    long u = units * team / n_teams;
    const long u_end = units * (team + 1) / n_teams;

    // A team's range is [tail part of a cut tile block][whole tile blocks ...][head part of a cut tile block]; the three kinds
    // are separate code regions (one generic loop body makes the register allocator spill the 128 accumulators).
    const int S = p.S;
    int tb = (int)(u / S);
    const int k_first = (int)(u - (long)tb * S);
    if (k_first > 0) {
        // tail part [k_first, S) of a tile block owned by the previous team: park the accumulators, publish
        f32x16 acc[2][2][2];
        zero_acc(acc);
        wino_mainloop(p, lds, tb, nt, k_first, S, acc, wave, lane);
        // parked data moves with relaxed AGENT-scope atomics (8 B each: sc1 stores / loads that are coherent at device level by
        // themselves).  A release/acquire fence pair would do it too, but on a multi-XCD part it writes back / invalidates the
        // whole L2 of the XCD -- measured +0.1 ms per layer because every workgroup's V/U reuse is lost with it.
        unsigned long long* my_ws = reinterpret_cast<unsigned long long*>(p.ws) + (size_t)lid * (64 * 512) + t;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qd = 0; qd < 8; ++qd) {
                        const unsigned long long v = (unsigned long long)__float_as_uint(acc[c][i][j][2 * qd]) |
                                                     ((unsigned long long)__float_as_uint(acc[c][i][j][2 * qd + 1]) << 32);
                        __hip_atomic_store(&my_ws[(((c * 2 + i) * 2 + j) * 8 + qd) * 512], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
        // every wave's parked stores must be acknowledged before the counter / flag moves: written out, because hipcc emits no
        // vmcnt wait for a workgroup-scope release fence or __syncthreads() here (checked in the ISA)
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __syncthreads();
        if (t == 0) __hip_atomic_store(p.flags + lid, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u += S - k_first;
        ++tb;
    }
    const int tb_whole_end = (int)(u_end / S);  // whole tile blocks: [tb, tb_whole_end)
    for (; tb < tb_whole_end; ++tb) {
        f32x16 acc[2][2][2];
        zero_acc(acc);
        wino_mainloop(p, lds, tb, nt, 0, S, acc, wave, lane);
        wino_epilogue(p, lds, tb, nt, acc, t, wave, lane);
        __syncthreads();  // epilogue readers done before the next item's DMA lands in LDS
    }
    const int k_last = (int)(u_end - (long)tb_whole_end * S);
    if (k_last > 0) {
        // head part [0, k_last): the next team's member nt parked slabs [k_last, S) at the very start of its life
        f32x16 acc[2][2][2];
        zero_acc(acc);
        wino_mainloop(p, lds, tb_whole_end, nt, 0, k_last, acc, wave, lane);
        const int partner = lid + p.n_nt;
        if (t == 0) {
            // bounded (~1 s): every workgroup reaches its publish before anything it waits for, so the bound is only hit when
            // the partner is not resident (CUs held by other streams / a CU mask) or on a logic error.  It keeps the device from
            // hanging; the output of this tile is then WRONG, which is recorded in the ctx's error word (hpe_device_status).
            bool ok = false;
            for (int spin = 0; spin < (1 << 22); ++spin) {
                if (__hip_atomic_load(p.flags + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.epoch) {
                    ok = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok && p.err) atomicOr(p.err, 1u);
        }
        __syncthreads();
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p.ws) + (size_t)partner * (64 * 512) + t;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("" ::: "memory");  // at most one 32x32 block (8 x 8 B per lane) of parked data in flight
#pragma unroll
                    for (int qd = 0; qd < 8; ++qd) {
                        const unsigned long long v = __hip_atomic_load(&src[(((c * 2 + i) * 2 + j) * 8 + qd) * 512], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        acc[c][i][j][2 * qd] += __uint_as_float((unsigned)(v & 0xFFFFFFFFull));
                        acc[c][i][j][2 * qd + 1] += __uint_as_float((unsigned)(v >> 32));
                    }
                }
        // consumed: clear the flag, so that a replay of this very launch (hipGraph: same epoch baked in) waits for fresh data
        if (t == 0) __hip_atomic_store(p.flags + partner, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wino_epilogue(p, lds, tb_whole_end, nt, acc, t, wave, lane);
    }
}

// ------------------------------------------------------------------------------------------------ fused input transform (56x56 and 28x28 maps)
// For the large maps the blocked V costs more HBM traffic than the layer is worth (C = 64: 1.6 GB per layer), so V never
// leaves the chip here.  The producing 1x1 convolution writes its output channel-slab major, Xs[slab of 8 ch][pixel][8]
// (GemmArgs::y_slab8), which makes the 8-channel slab of a row of pixels contiguous.  A workgroup owns R consecutive
// tile rows of the batch (R * TW <= 64 tiles: 2 x 28 on the 56x56 maps, 4 x 14 on 28x28; the batch is one tall stack of
// tile rows, so every workgroup is full).  Per slab:
//   * the 4 input rows of each of its tile rows arrive by LDS-DMA as 16-B chunks in the layout
//     raw[tile row][input row 4][half 2][pixel parity 2][pixel / 2][4 ch]  (zero page for the halo) -- ~15 KB;
//   * thread (tile = lane, wave = (half, xi)) reads 2 rows x 4 pixels from it (lanes hit consecutive 16-B slots:
//     conflict-free), forms row xi of B^T d B for its 4 channels and writes the 4 components into the V slab image the
//     MFMA loop reads -- exactly the bytes wino_input_kernel would have written to HBM;
//   * the MFMA work on slab s overlaps the transform of slab s + 1 and the DMAs of U(s + 1) and raw(s + 2).
// LDS: V 2 x 32 KB + U 2 x 32 KB + raw 2 x 15 KB = 158 KB of the 160 KB.
struct WinoFusedArgs {
    const float* Xs;     // [S][M][8], M = B * H * W
    const float* U;
    const float* scale;
    const float* shift;
    const float* zero;   // >= 16 B of zeros
    float* y;            // NHWC [M][ldy]
    int H, W, TW, TH;    // map, tiles per row / per column
    int R;               // tile rows per workgroup
    int NG;              // tile rows in the batch (B * TH)
    int n_blk, n_nt, S;
    int NC;              // raw chunks per slab = R * 4 * 2 * 2 * (TW + 1)
    long M;
    int ldy, relu;
};

constexpr int RAWF = 960 * 4;  // floats per raw buffer (960 chunks >= 2*4*4*29 = 928 and 4*4*4*15 = 960)
constexpr int FUSED_LDS_FLOATS = 4 * OPER + 2 * RAWF;

__global__ __launch_bounds__(512, 2) void wino_fused_kernel(WinoFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rawb = lds + 4 * OPER;

    const int total = p.n_blk * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = total >> 3, rr = total & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int blk = lid / p.n_nt;
    const int nt = lid - blk * p.n_nt;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hi = lane >> 5;
    const int TW1 = p.TW + 1;

    // ---- raw DMA sources: chunk slot c = (i * 8 + wave) * 64 + lane, i = 0, 1 -> (tile row, input row, half, parity, idx)
    long rsrc[2];  // float offset inside a slab of Xs, or -1 for the zero page
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (i * 8 + wave) * 64 + lane;
        long off = -1;
        if (c < p.NC) {
            const int idx = c % TW1;
            int r = c / TW1;
            const int par = r & 1;
            r >>= 1;
            const int h = r & 1;
            r >>= 1;
            const int a = r & 3;
            const int trl = r >> 2;
            const int g = blk * p.R + trl;
            if (g < p.NG) {
                const int b = g / p.TH;
                const int ty = g - b * p.TH;
                const int iy = 2 * ty - 1 + a;
                const int px = 2 * idx + par - 1;
                if (iy >= 0 && iy < p.H && px >= 0 && px < p.W) off = ((long)(b * p.H + iy) * p.W + px) * 8 + h * 4;
            }
        }
        rsrc[i] = off;
    }
    const int n_raw_instr = (p.NC + 63) >> 6;  // wave-instructions per slab (<= 15)
    const float* usrc = p.U + (size_t)nt * p.S * OPER + wave * 1024 + lane * 4;

    auto issue_raw = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i * 8 + wave < n_raw_instr) {
                const float* src = rsrc[i] >= 0 ? p.Xs + (size_t)s * p.M * 8 + rsrc[i] : p.zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(rawb + buf * RAWF + (i * 8 + wave) * 256), 16, 0, 0);
            }
        }
    };
    auto issue_u = [&](int s, int buf) {
        float* dstu = lds + buf * (2 * OPER) + OPER + wave * 1024;
        const float* su = usrc + (size_t)s * OPER;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(su + i * 256),
                                             (__attribute__((address_space(3))) void*)(dstu + i * 256), 16, 0, 0);
    };

    // ---- transform role: tile = lane, wave = (half, xi)
    const int th = wave & 1, xi = wave >> 1;
    const int n_tiles = p.R * p.TW;
    const bool t_live = lane < n_tiles;
    const int t_trl = t_live ? lane / p.TW : 0;
    const int t_tx = t_live ? lane - t_trl * p.TW : 0;
    // row xi of B^T d uses input rows (ra, rb): xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = (xi == 0) ? 0 : (xi == 2 ? 2 : 1);
    const int rb = (xi == 0) ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sgn = (xi == 1) ? 1.f : -1.f;
    // raw chunk (floats) of input row a, pixel 2 tx - 1 + e: parity e & 1, idx tx + (e >> 1)
    const int raw_a = (((t_trl * 4 + ra) * 2 + th) * 2) * TW1 + t_tx;
    const int raw_b = (((t_trl * 4 + rb) * 2 + th) * 2) * TW1 + t_tx;
    auto transform = [&](int rbuf, int vbuf) {
        if (!t_live) return;
        const float* rw = rawb + rbuf * RAWF;
        f32x4 r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = ((e & 1) * TW1 + (e >> 1)) * 4;
            const f32x4 da = *reinterpret_cast<const f32x4*>(rw + raw_a * 4 + o);
            const f32x4 db = *reinterpret_cast<const f32x4*>(rw + raw_b * 4 + o);
            r[e] = da + sgn * db;
        }
        float* v = lds + vbuf * (2 * OPER) + (((xi * 4) * 2 + th) * 64 + lane) * 4;
        *reinterpret_cast<f32x4*>(v + 0 * 512) = r[0] - r[2];
        *reinterpret_cast<f32x4*>(v + 1 * 512) = r[1] + r[2];
        *reinterpret_cast<f32x4*>(v + 2 * 512) = r[2] - r[1];
        *reinterpret_cast<f32x4*>(v + 3 * 512) = r[1] - r[3];
    };

    f32x16 acc[2][2][2];
    zero_acc(acc);
    const int frag = ((2 * wave) * 2 + hi) * 256 + (lane & 31) * 4;
    const int S = p.S;

    issue_raw(0, 0);
    if (S > 1) issue_raw(1, 1);
    issue_u(0, 0);
    // rows of V that belong to no tile (n_tiles < 64) are never written by the transform: clear them once in both buffers so
    // that the MFMAs do not chew on stale LDS contents (their outputs are masked anyway)
    if (n_tiles < 64) {
        const int dead = 64 - n_tiles;  // rows n_tiles .. 63 of each of the 32 (component, half) planes, 2 buffers
        for (int i = t; i < 2 * 32 * dead; i += 512) {
            const int row = n_tiles + i % dead;
            const int plane = (i / dead) & 31;
            const int buf = i / (dead * 32);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(lds + buf * (2 * OPER) + (plane * 64 + row) * 4) = z;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70 | 4);  // vmcnt(4): everything but the 4 newest DMAs (U(0), issued last) has landed: raw(0), raw(1)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    transform(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), written out (see wino_gemm_loop): raw(1), U(0) landed
    __syncthreads();                     // ... and V(0) written by every wave
    for (int s = 0; s < S; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < S) issue_u(s + 1, nxt);
        if (s + 2 < S) issue_raw(s + 2, cur);
        const int cb = cur * (2 * OPER);
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[c][i] = *reinterpret_cast<const f32x4*>(&lds[cb + frag + c * 512 + i * 128]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[c][j] = *reinterpret_cast<const f32x4*>(&lds[cb + OPER + frag + c * 512 + j * 128]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][i][ks], fb[0][j][ks], acc[0][i][j], 0, 0, 0);
        if (s + 1 < S) transform(nxt, nxt);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][i][ks], fb[1][j][ks], acc[1][i][j], 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), written out: U(s + 1) and raw(s + 2) of this wave have landed
        __syncthreads();
    }

    // ---- epilogue (as wino_epilogue, with this kernel's tile -> pixel mapping)
    const int n0 = nt * WN_;
    const int em = t >> 4;
    const int eq = (t & 15) * 4;
    const int n = n0 + eq;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    lds[((2 * wave + c) * 32 + m) * 64 + 32 * j + (lane & 31)] = acc[c][i][j][e];
                }
        __syncthreads();
        const int lt = 32 * i + em;
        const int trl = lt / p.TW;
        const int tx = lt - trl * p.TW;
        const int g = blk * p.R + trl;
        if (lt < n_tiles && g < p.NG) {
            f32x4 Mv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) Mv[c] = *reinterpret_cast<const f32x4*>(&lds[(c * 32 + em) * 64 + eq]);
            f32x4 u0[4], u1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                u0[v] = Mv[v] + Mv[4 + v] + Mv[8 + v];
                u1[v] = Mv[4 + v] - Mv[8 + v] - Mv[12 + v];
            }
            f32x4 o[2][2];
            o[0][0] = u0[0] + u0[1] + u0[2];
            o[0][1] = u0[1] - u0[2] - u0[3];
            o[1][0] = u1[0] + u1[1] + u1[2];
            o[1][1] = u1[1] - u1[2] - u1[3];
            const int b = g / p.TH;
            const int ty = g - b * p.TH;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oy = 2 * ty + dy, ox = 2 * tx + dx;
                    f32x4 v = o[dy][dx] * sc + sh;
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * p.ldy + n) = v;
                }
        }
    }
}

// NHWC -> channel-slab major (only the debug entry point needs it: in the network the producing convolution writes Xs itself)
__global__ __launch_bounds__(256) void nhwc_to_slab8_kernel(const float* __restrict__ x, float* __restrict__ xs, long M, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one float4
    const long total = M * C / 4;
    if (i >= total) return;
    const long m = i / (C / 4);
    const int c = (int)(i - m * (C / 4)) * 4;
    *reinterpret_cast<f32x4*>(xs + ((size_t)(c >> 3) * M + m) * 8 + (c & 7)) = *reinterpret_cast<const f32x4*>(x + m * C + c);
}

}  // namespace

// hipFuncSetAttribute applies to the CURRENT device: hpe_finalize calls this once per ctx under its device guard
hipError_t hpe_wino_init_device() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm_streamk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               FUSED_LDS_FLOATS * (int)sizeof(float));
}

size_t hpe_wino_v_floats(int B, int H, int W, int C) {
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const size_t T = (size_t)B * TH * TW;
    return ((T + 63) / 64) * 64 * 16 * (size_t)C;
}

hipError_t hpe_launch_wino_conv3(const float* x, int lda, const float* U, const float* scale, const float* shift, float* y, int ldy,
                                 int B, int H, int W, int C, int N, int relu, float* V, const WinoStreamK* sk, hipStream_t st) {
    if (C % 32 != 0 || N % 64 != 0 || lda % 4 != 0 || ldy % 4 != 0 || B < 1 || H < 1 || W < 1) return hipErrorInvalidValue;
    constexpr int LDS_BYTES = WINO_LDS_BYTES;
    const int TH = (H + 1) / 2, TW = (W + 1) / 2, TT = TH * TW;
    const long Tl = (long)B * TT;
    if (Tl > (1L << 30)) return hipErrorInvalidValue;
    const int T = (int)Tl;
    const int Tpad = (T + 63) / 64 * 64;
    {
        const long waves = (long)(Tpad / 8) * (C / 32);
        const int blocks = (int)((waves + 3) / 4);
        hipLaunchKernelGGL(wino_input_kernel, dim3(blocks), dim3(256), 0, st, x, V, H, W, C, TW, TT, T, Tpad, lda);
    }
    WinoArgs p{};
    p.V = V;
    p.U = U;
    p.scale = scale;
    p.shift = shift;
    p.y = y;
    p.H = H;
    p.W = W;
    p.N = N;
    p.TW = TW;
    p.TT = TT;
    p.T = T;
    p.S = C / 8;
    p.n_tb = Tpad / 64;
    p.n_nt = N / 64;
    p.ldy = ldy;
    p.relu = relu;
    // persistent stream-K grid: one workgroup per CU in teams of n_nt, every team at least one whole tile block
    if (sk && sk->ws && sk->flags && sk->n_wg >= 8 && sk->n_wg % (8 * p.n_nt) == 0 && p.n_tb >= sk->n_wg / p.n_nt) {
        p.ws = sk->ws;
        p.flags = sk->flags;
        p.epoch = sk->epoch;
        p.err = sk->err;
        hipLaunchKernelGGL(wino_gemm_streamk_kernel, dim3(sk->n_wg), dim3(512), LDS_BYTES, st, p);
    } else {
        hipLaunchKernelGGL(wino_gemm_kernel, dim3(p.n_tb * p.n_nt), dim3(512), LDS_BYTES, st, p);
    }
    return hipGetLastError();
}

// fused variant for even maps with W / 2 <= 64 tiles per row: xs is channel-slab major [C/8][B*H*W][8]
hipError_t hpe_launch_wino_fused_conv3(const float* xs, const float* U, const float* scale, const float* shift, const float* zero16, float* y,
                                       int ldy, int B, int H, int W, int C, int N, int relu, hipStream_t st) {
    if (C % 8 != 0 || N % 64 != 0 || ldy % 4 != 0 || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || !zero16) return hipErrorInvalidValue;
    const int TW = W / 2, TH = H / 2;
    if (TW > 64) return hipErrorInvalidValue;
    int R = 64 / TW;
    if (R > 960 / (16 * (TW + 1))) R = 960 / (16 * (TW + 1));  // raw buffer: 960 chunks
    if (R > TH * B) R = TH * B;
    const int NC = R * 4 * 2 * 2 * (TW + 1);
    if (R < 1 || NC > 960) return hipErrorInvalidValue;
    constexpr int LDS_BYTES = FUSED_LDS_FLOATS * (int)sizeof(float);
    WinoFusedArgs p{};
    p.Xs = xs;
    p.U = U;
    p.scale = scale;
    p.shift = shift;
    p.zero = zero16;
    p.y = y;
    p.H = H;
    p.W = W;
    p.TW = TW;
    p.TH = TH;
    p.R = R;
    p.NG = B * TH;
    p.n_blk = (p.NG + R - 1) / R;
    p.n_nt = N / 64;
    p.S = C / 8;
    p.NC = NC;
    p.M = (long)B * H * W;
    p.ldy = ldy;
    p.relu = relu;
    hipLaunchKernelGGL(wino_fused_kernel, dim3(p.n_blk * p.n_nt), dim3(512), LDS_BYTES, st, p);
    return hipGetLastError();
}

int hpe_wino_fused_items(int B, int H, int W, int N) {
    const int TW = W / 2, TH = H / 2;
    if ((H & 1) || (W & 1) || TW > 64 || TW < 1) return 0;
    int R = 64 / TW;
    if (R > 960 / (16 * (TW + 1))) R = 960 / (16 * (TW + 1));
    if (R < 1) return 0;
    return ((B * TH + R - 1) / R) * (N / 64);
}

hipError_t hpe_launch_nhwc_to_slab8(const float* x, float* xs, long M, int C, hipStream_t st) {
    if (C % 8 != 0 || M < 1) return hipErrorInvalidValue;
    const long total = M * C / 4;
    hipLaunchKernelGGL(nhwc_to_slab8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, xs, M, C);
    return hipGetLastError();
}
