// smpl.hip -- SMPL mesh generator + orthographic reprojection for gfx950.
//
// Reference semantics: SMPL.__call__ (src/tf_smpl/batch_smpl.py:88-160), batch_rodrigues / batch_skew /
// batch_global_rigid_transformation (src/tf_smpl/batch_lbs.py:15-64, 91-152), batch_orth_proj_idrot /
// reproject_vertices (src/tf_smpl/projection.py:23-56).
//
// Three kernels per IEF stage (all VALU; no MFMA here by design -- see DESIGN.md):
//   smpl_pose_kernel    one wave per image: batched Rodrigues (24 joints on 24 lanes), J = Jbasis . [1|beta],
//                       forward kinematics by tree level with the 24 joint matrices staged in LDS,
//                       relative transforms A, pose feature (R - I).
//   smpl_skin_kernel    (vertex tile 256) x (image tile 8): shape blend + pose blend + linear blend skinning
//                       with the per-image operands (beta, pose feature, 24 x 3x4 A) wave-uniform, so they come
//                       through the scalar cache into SGPRs while the per-vertex bases stream 12 B/lane coalesced.
//   joint_regress_kernel  the 6890 -> K joint regressor: per-thread strided partial sums, wavefront
//                       shuffle reduction, cross-wave LDS reduction; epilogue fuses batch_orth_proj_idrot.
//                       The same kernel, run once at load time on (v_template, shapedirs[k]) gives the
//                       24-joint basis Jbasis (J is linear in beta: J = Jreg^T (v_template + S beta)).
#include <hip/hip_runtime.h>

#include "../../include/hpe.h"
#include "hpe_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define V SMPL_V
#define V3 (SMPL_V * 3)
#define IT SMPL_IMG_TILE

namespace {

// ---------------------------------------------------------------------------------------------
// one wave (64 threads) per image
__global__ __launch_bounds__(64) void smpl_pose_kernel(SmplDev d, const float* __restrict__ theta, int ldtheta, int B, int Bpad,
                                                       float* __restrict__ pfT, float* __restrict__ betaT,
                                                       float* __restrict__ Aout, float* __restrict__ cams_ws,
                                                       float* __restrict__ Rs_out, float* __restrict__ Jt_out,
                                                       float* __restrict__ cams_out, float* __restrict__ theta_out) {
    __shared__ float sR[24][9];
    __shared__ float sJ[24][3];
    __shared__ float sG[24][12];
    __shared__ float sTh[85];
    const int b = blockIdx.x;
    const int j = threadIdx.x;
    const bool live = b < B;

    for (int k = j; k < 85; k += 64) {
        const float v = live ? theta[(size_t)b * ldtheta + k] : 0.f;
        sTh[k] = v;
        if (live && theta_out) theta_out[(size_t)b * 85 + k] = v;
    }
    __syncthreads();
    if (j < 4) {
        const float v = j < 3 ? sTh[j] : 0.f;
        cams_ws[b * 4 + j] = v;
        if (live && cams_out && j < 3) cams_out[b * 3 + j] = v;
    }
    if (j < 10) betaT[j * Bpad + b] = sTh[75 + j];

    if (j < 24) {
        // batch_rodrigues (batch_lbs.py:42-64): angle = ||theta + 1e-8||, r = theta / angle
        const float x = sTh[3 + 3 * j], y = sTh[4 + 3 * j], z = sTh[5 + 3 * j];
        const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = x / angle, ry = y / angle, rz = z / angle;
        const float c = cosf(angle), s = sinf(angle), oc = 1.0f - c;
        float R[9];
        // cos * I + (1 - cos) * r r^T + sin * skew(r);  skew = [[0,-z,y],[z,0,-x],[-y,x,0]] (batch_lbs.py:24-36)
        R[0] = c + oc * (rx * rx);
        R[1] = oc * (rx * ry) + s * (-rz);
        R[2] = oc * (rx * rz) + s * ry;
        R[3] = oc * (ry * rx) + s * rz;
        R[4] = c + oc * (ry * ry);
        R[5] = oc * (ry * rz) + s * (-rx);
        R[6] = oc * (rz * rx) + s * (-ry);
        R[7] = oc * (rz * ry) + s * rx;
        R[8] = c + oc * (rz * rz);
#pragma unroll
        for (int e = 0; e < 9; ++e) sR[j][e] = R[e];
        if (live && Rs_out) {
#pragma unroll
            for (int e = 0; e < 9; ++e) Rs_out[((size_t)b * 24 + j) * 9 + e] = R[e];
        }
        // pose_feature = (Rs[:,1:] - I).reshape(207)  (batch_smpl.py:126-127), stored transposed [207][Bpad]
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e) pfT[((j - 1) * 9 + e) * Bpad + b] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
        // J = Jreg^T v_shaped  ==  Jbasis[0] + sum_k beta_k Jbasis[1+k]   (batch_smpl.py:110-118)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) acc += sTh[75 + k] * d.j_basis[((1 + k) * 24 + j) * 3 + cc];
            sJ[j][cc] = acc + d.j_basis[j * 3 + cc];
        }
    }
    __syncthreads();

    // batch_global_rigid_transformation (batch_lbs.py:128-135), level by level of the kinematic tree
    const int par = (j < 24) ? d.parents[j] : -1;
    const int dep = (j < 24) ? d.depth[j] : -1;
    for (int lvl = 0; lvl <= d.max_depth; ++lvl) {
        if (j < 24 && dep == lvl) {
            if (par < 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    sG[j][r * 4 + 0] = sR[j][r * 3 + 0];
                    sG[j][r * 4 + 1] = sR[j][r * 3 + 1];
                    sG[j][r * 4 + 2] = sR[j][r * 3 + 2];
                    sG[j][r * 4 + 3] = sJ[j][r];
                }
            } else {
                const float tx = sJ[j][0] - sJ[par][0], ty = sJ[j][1] - sJ[par][1], tz = sJ[j][2] - sJ[par][2];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float p0 = sG[par][r * 4 + 0], p1 = sG[par][r * 4 + 1], p2 = sG[par][r * 4 + 2], p3 = sG[par][r * 4 + 3];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) sG[j][r * 4 + cc] = p0 * sR[j][cc] + p1 * sR[j][3 + cc] + p2 * sR[j][6 + cc];
                    sG[j][r * 4 + 3] = p0 * tx + p1 * ty + p2 * tz + p3;
                }
            }
        }
        __syncthreads();
    }
    if (j < 24) {
        // A = G - [0 | G.R * J]  (batch_lbs.py:146-150); J_transformed = G[:, :3, 3] (:140)
        float* a = Aout + ((size_t)b * 24 + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float g0 = sG[j][r * 4 + 0], g1 = sG[j][r * 4 + 1], g2 = sG[j][r * 4 + 2], g3 = sG[j][r * 4 + 3];
            a[r * 4 + 0] = g0;
            a[r * 4 + 1] = g1;
            a[r * 4 + 2] = g2;
            a[r * 4 + 3] = g3 - (g0 * sJ[j][0] + g1 * sJ[j][1] + g2 * sJ[j][2]);
            if (live && Jt_out) Jt_out[((size_t)b * 24 + j) * 3 + r] = g3;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// grid (ceil(V/256), Bpad/IT); thread = one vertex x IT images
template <bool W2D>
__global__ __launch_bounds__(256) void smpl_skin_kernel(SmplDev d, const float* __restrict__ pfT, const float* __restrict__ betaT,
                                                        const float* __restrict__ A, const float* __restrict__ cams, int Bpad,
                                                        int B, float* __restrict__ verts, float* __restrict__ verts2d,
                                                        float im_w, float im_h) {
    const int vraw = blockIdx.x * 256 + threadIdx.x;
    const bool vok = vraw < V;
    const int v = vok ? vraw : V - 1;
    const int img0 = blockIdx.y * IT;

    // 1. v_shaped = beta . shapedirs + v_template   (batch_smpl.py:110-112)
    float vs[IT][3];
#pragma unroll
    for (int i = 0; i < IT; ++i) vs[i][0] = vs[i][1] = vs[i][2] = 0.f;
#pragma unroll 2
    for (int k = 0; k < 10; ++k) {
        const float* sd = d.shapedirs + (size_t)k * V3 + 3 * v;
        const float s0 = sd[0], s1 = sd[1], s2 = sd[2];
        const float* bt = betaT + k * Bpad + img0;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const float bb = bt[i];
            vs[i][0] += bb * s0;
            vs[i][1] += bb * s1;
            vs[i][2] += bb * s2;
        }
    }
    {
        const float t0 = d.v_template[3 * v], t1 = d.v_template[3 * v + 1], t2 = d.v_template[3 * v + 2];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            vs[i][0] += t0;
            vs[i][1] += t1;
            vs[i][2] += t2;
        }
    }
    // 3. v_posed = pose_feature . posedirs + v_shaped   (batch_smpl.py:130-132)
    float vp[IT][3];
#pragma unroll
    for (int i = 0; i < IT; ++i) vp[i][0] = vp[i][1] = vp[i][2] = 0.f;
    // 6 basis rows per batch: 18 vertex-basis loads and 6 scalar loads (48 SGPRs of pose feature) in flight per wave -- with 3 the
    // 72 FMAs of a batch did not cover the scalar-load latency (66-70 us per launch against 21 us of FMA issue time)
#pragma unroll 6
    for (int k = 0; k < 207; ++k) {
        const float* pd = d.posedirs + (size_t)k * V3 + 3 * v;
        const float p0 = pd[0], p1 = pd[1], p2 = pd[2];
        const float* pf = pfT + k * Bpad + img0;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const float f = pf[i];
            vp[i][0] += f * p0;
            vp[i][1] += f * p1;
            vp[i][2] += f * p2;
        }
    }
    // 5. skinning: T = W . A ; verts = (T . [v_posed; 1])[:3]   (batch_smpl.py:139-149)
    float w[24];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(d.weights + (size_t)v * 24);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const f32x4 t = wp[q];
            w[4 * q] = t.x;
            w[4 * q + 1] = t.y;
            w[4 * q + 2] = t.z;
            w[4 * q + 3] = t.w;
        }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const float* Ai = A + (size_t)(img0 + i) * 288;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 24; ++jj) {
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] += w[jj] * Ai[jj * 12 + e];
        }
        const float px = vp[i][0] + vs[i][0], py = vp[i][1] + vs[i][1], pz = vp[i][2] + vs[i][2];
        const float ox = T[0] * px + T[1] * py + T[2] * pz + T[3];
        const float oy = T[4] * px + T[5] * py + T[6] * pz + T[7];
        const float oz = T[8] * px + T[9] * py + T[10] * pz + T[11];
        if (vok && (img0 + i) < B) {
            float* o = verts + ((size_t)(img0 + i) * V + v) * 3;
            o[0] = ox;
            o[1] = oy;
            o[2] = oz;
            if (W2D) {
                // reproject_vertices (projection.py:45-56): (s*(x+t) + 1) * 0.5 * im_size
                const float* cm = cams + (img0 + i) * 4;
                float* o2 = verts2d + ((size_t)(img0 + i) * V + v) * 2;
                o2[0] = ((cm[0] * (ox + cm[1])) + 1.0f) * 0.5f * im_w;
                o2[1] = ((cm[0] * (oy + cm[2])) + 1.0f) * 0.5f * im_h;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Small batches (B <= 8, the single-frame latency path): one wave = 64 vertices of ONE image, grid (ceil(V/64), B) = 108 workgroups
// per image instead of 27 per 8 images, the 207-term pose blend unrolled 23 deep (69 independent loads in flight per lane: nine
// round trips instead of thirty-five), and the keypoint regressor's partial sums over the wave's 64 vertices taken right here from
// the registers that hold the vertices (wave-shuffle reduction) -- smpl_kp_finish_kernel adds the 108 partials per image in a fixed
// order.  Same arithmetic per vertex as smpl_skin_kernel (summation order of the joint regressor differs: 64-vertex groups).
template <bool W2D>
__global__ __launch_bounds__(64) void smpl_skin_small_kernel(SmplDev d, const float* __restrict__ pfT, const float* __restrict__ betaT,
                                                             const float* __restrict__ A, const float* __restrict__ cams, int Bpad, int B,
                                                             float* __restrict__ verts, float* __restrict__ verts2d, float im_w, float im_h,
                                                             float* __restrict__ kp_part, int n_tiles) {
    const int lane = threadIdx.x;
    const int vraw = blockIdx.x * 64 + lane;
    const bool vok = vraw < V;
    const int v = vok ? vraw : V - 1;
    const int img = blockIdx.y;
    float vs0 = d.v_template[3 * v], vs1 = d.v_template[3 * v + 1], vs2 = d.v_template[3 * v + 2];
    // 1. v_shaped = beta . shapedirs + v_template   (batch_smpl.py:110-112); same order of additions as the large-batch kernel:
    //    the ten blend terms first, the template last
    {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float* sd = d.shapedirs + (size_t)k * V3 + 3 * v;
            const float bb = betaT[k * Bpad + img];
            a0 += bb * sd[0];
            a1 += bb * sd[1];
            a2 += bb * sd[2];
        }
        vs0 = a0 + vs0;
        vs1 = a1 + vs1;
        vs2 = a2 + vs2;
    }
    // 3. v_posed = pose_feature . posedirs + v_shaped   (batch_smpl.py:130-132)
    float vp0 = 0.f, vp1 = 0.f, vp2 = 0.f;
#pragma unroll 23
    for (int k = 0; k < 207; ++k) {
        const float* pd = d.posedirs + (size_t)k * V3 + 3 * v;
        const float f = pfT[k * Bpad + img];
        vp0 += f * pd[0];
        vp1 += f * pd[1];
        vp2 += f * pd[2];
    }
    // 5. skinning: T = W . A ; verts = (T . [v_posed; 1])[:3]   (batch_smpl.py:139-149)
    float w[24];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(d.weights + (size_t)v * 24);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const f32x4 t = wp[q];
            w[4 * q] = t.x;
            w[4 * q + 1] = t.y;
            w[4 * q + 2] = t.z;
            w[4 * q + 3] = t.w;
        }
    }
    const float* Ai = A + (size_t)img * 288;
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 24; ++jj) {
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += w[jj] * Ai[jj * 12 + e];
    }
    const float px = vp0 + vs0, py = vp1 + vs1, pz = vp2 + vs2;
    const float ox = T[0] * px + T[1] * py + T[2] * pz + T[3];
    const float oy = T[4] * px + T[5] * py + T[6] * pz + T[7];
    const float oz = T[8] * px + T[9] * py + T[10] * pz + T[11];
    if (vok && verts) {
        float* o = verts + ((size_t)img * V + v) * 3;
        o[0] = ox;
        o[1] = oy;
        o[2] = oz;
    }
    if (W2D && vok) {
        const float* cm = cams + img * 4;
        float* o2 = verts2d + ((size_t)img * V + v) * 2;
        o2[0] = ((cm[0] * (ox + cm[1])) + 1.0f) * 0.5f * im_w;
        o2[1] = ((cm[0] * (oy + cm[2])) + 1.0f) * 0.5f * im_h;
    }
    if (kp_part) {
        // keypoint regressor (batch_smpl.py:152-155), partial over this wave's 64 vertices
        const float x0 = vok ? ox : 0.f, x1 = vok ? oy : 0.f, x2 = vok ? oz : 0.f;
        const f32x4* rp = reinterpret_cast<const f32x4*>(d.kp_reg + (size_t)v * 24);
        float* out = kp_part + ((size_t)img * n_tiles + blockIdx.x) * 72;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const f32x4 r = rp[q];
            const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s0 = x0 * rr[u], s1 = x1 * rr[u], s2 = x2 * rr[u];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    s0 += __shfl_xor(s0, off, 64);
                    s1 += __shfl_xor(s1, off, 64);
                    s2 += __shfl_xor(s2, off, 64);
                }
                if (lane == 0) {
                    out[(4 * q + u) * 3 + 0] = s0;
                    out[(4 * q + u) * 3 + 1] = s1;
                    out[(4 * q + u) * 3 + 2] = s2;
                }
            }
        }
    }
}

// joints[n][k][c] = sum over the n_tiles partials (fixed order: four interleaved groups, then the groups); optional fused
// batch_orth_proj_idrot.  288 threads = 72 sums x 4 groups, so that the loads of a sum are 27 independent ones, not 108 in a chain.
__global__ __launch_bounds__(288) void smpl_kp_finish_kernel(const float* __restrict__ kp_part, int n_tiles, int K, float* __restrict__ out,
                                                             const float* __restrict__ cams, float* __restrict__ kp2d) {
    __shared__ float grp[4][72];
    __shared__ float fin[72];
    const int n = blockIdx.x;
    const int t = threadIdx.x;
    const int e = t % 72, g = t / 72;
    {
        const float* p = kp_part + (size_t)n * n_tiles * 72 + e;
        float s = 0.f;
#pragma unroll 9
        for (int i = g; i < n_tiles; i += 4) s += p[(size_t)i * 72];
        grp[g][e] = s;
    }
    __syncthreads();
    if (t < 72) {
        const float s = (grp[0][t] + grp[1][t]) + (grp[2][t] + grp[3][t]);
        fin[t] = s;
        if (t < K * 3 && out) out[(size_t)n * K * 3 + t] = s;
    }
    __syncthreads();
    if (kp2d && t < K * 2) {
        const int k = t >> 1, c = t & 1;
        const float* cm = cams + n * 4;
        kp2d[(size_t)n * K * 2 + t] = cm[0] * (fin[k * 3 + c] + cm[1 + c]);  // projection.py:27-33
    }
}

// ---------------------------------------------------------------------------------------------
// out[n][k][c] = sum_v X[n][v][c] * reg[v][k], k < K <= 24 (reg rows are 24 floats, zero padded).
// one 256-thread workgroup per n.  Optional fused batch_orth_proj_idrot: kp2d[n][k][0:2] = s*(out[..,:2] + t).
__global__ __launch_bounds__(256) void joint_regress_kernel(const float* __restrict__ X, const float* __restrict__ reg, int K,
                                                            float* __restrict__ out, const float* __restrict__ cams,
                                                            float* __restrict__ kp2d) {
    __shared__ float part[4][72];
    __shared__ float fin[72];
    const int n = blockIdx.x;
    const int t = threadIdx.x;
    float acc[24][3];
#pragma unroll
    for (int k = 0; k < 24; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
    const float* Xn = X + (size_t)n * V3;
    // one workgroup per mesh = 4 waves per CU: nothing hides a load but the loads of the same thread -> 4 vertices (28 loads) in flight
#pragma unroll 4
    for (int v = t; v < V; v += 256) {
        const float x0 = Xn[3 * v], x1 = Xn[3 * v + 1], x2 = Xn[3 * v + 2];
        const f32x4* rp = reinterpret_cast<const f32x4*>(reg + (size_t)v * 24);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const f32x4 r = rp[q];
            const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[4 * q + u][0] += x0 * rr[u];
                acc[4 * q + u][1] += x1 * rr[u];
                acc[4 * q + u][2] += x2 * rr[u];
            }
        }
    }
    // wavefront shuffle reduction (64 lanes), then 4 waves through LDS
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int k = 0; k < 24; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = acc[k][c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) part[wave][k * 3 + c] = s;
        }
    __syncthreads();
    if (t < 72) {
        const float s = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        fin[t] = s;
        if (t < K * 3 && out) out[(size_t)n * K * 3 + t] = s;
    }
    __syncthreads();
    if (kp2d && t < K * 2) {
        const int k = t >> 1, c = t & 1;
        const float* cm = cams + n * 4;
        kp2d[(size_t)n * K * 2 + t] = cm[0] * (fin[k * 3 + c] + cm[1 + c]);  // projection.py:27-33
    }
}

// out[b][p][0:2] = cam[b][0] * (X[b][p][0:2] + cam[b][1:3]);  pixels: (out + 1) * 0.5 * (sx, sy)
__global__ void orth_proj_kernel(const float* __restrict__ X, const float* __restrict__ cam, int B, int P, float sx, float sy,
                                 int pixels, float* __restrict__ out) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / P);
        const float s = cam[b * 3], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
        float u = s * (X[i * 3] + tx), w = s * (X[i * 3 + 1] + ty);
        if (pixels) {
            u = (u + 1.0f) * 0.5f * sx;
            w = (w + 1.0f) * 0.5f * sy;
        }
        out[i * 2] = u;
        out[i * 2 + 1] = w;
    }
}

}  // namespace

hipError_t hpe_launch_joint_regress(const float* X, const float* reg, int n, int K, float* out, const float* cams, float* kp2d,
                                    hipStream_t st) {
    if (K < 1 || K > 24 || n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(joint_regress_kernel, dim3(n), dim3(256), 0, st, X, reg, K, out, cams, kp2d);
    return hipGetLastError();
}

hipError_t hpe_launch_orth_proj(const float* X, const float* cam, int B, int P, float sx, float sy, int pixels, float* out,
                                hipStream_t st) {
    const long total = (long)B * P;
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(orth_proj_kernel, dim3((int)g), dim3(256), 0, st, X, cam, B, P, sx, sy, pixels, out);
    return hipGetLastError();
}

hipError_t hpe_launch_smpl(const SmplDev& d, const SmplWork& w, const float* theta, int ldtheta, int B, const HpeOutputs* o,
                           hipStream_t st) {
    const int Bpad = ((B + IT - 1) / IT) * IT;
    if (Bpad > w.Bpad) return hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_pose_kernel, dim3(Bpad), dim3(64), 0, st, d, theta, ldtheta, B, w.Bpad, w.pfT, w.betaT, w.A, w.cams,
                       o->Rs, o->J_transformed, o->cams, o->theta);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const bool need_verts = o->verts || o->joints || o->kp2d || o->verts2d;
    if (!need_verts) return hipSuccess;
    if (B <= SMPL_SMALL_B && w.kp_part) {
        // latency path: 108 one-wave workgroups per image, keypoint partials from the skin kernel's registers, one finish launch
        const int n_tiles = (V + 63) / 64;
        const bool want_kp = o->joints || o->kp2d;
        dim3 g(n_tiles, B);
        if (o->verts2d)
            hipLaunchKernelGGL(smpl_skin_small_kernel<true>, g, dim3(64), 0, st, d, w.pfT, w.betaT, w.A, w.cams, w.Bpad, B, o->verts, o->verts2d,
                               (float)HPE_IMG_SIZE, (float)HPE_IMG_SIZE, want_kp ? w.kp_part : nullptr, n_tiles);
        else
            hipLaunchKernelGGL(smpl_skin_small_kernel<false>, g, dim3(64), 0, st, d, w.pfT, w.betaT, w.A, w.cams, w.Bpad, B, o->verts,
                               (float*)nullptr, 0.f, 0.f, want_kp ? w.kp_part : nullptr, n_tiles);
        e = hipGetLastError();
        if (e != hipSuccess || !want_kp) return e;
        hipLaunchKernelGGL(smpl_kp_finish_kernel, dim3(B), dim3(288), 0, st, w.kp_part, n_tiles, d.num_kp, o->joints, w.cams, o->kp2d);
        return hipGetLastError();
    }
    float* verts = o->verts ? o->verts : w.verts_tmp;
    dim3 grid((V + 255) / 256, Bpad / IT);
    if (o->verts2d)
        hipLaunchKernelGGL(smpl_skin_kernel<true>, grid, dim3(256), 0, st, d, w.pfT, w.betaT, w.A, w.cams, w.Bpad, B, verts,
                           o->verts2d, (float)HPE_IMG_SIZE, (float)HPE_IMG_SIZE);
    else
        hipLaunchKernelGGL(smpl_skin_kernel<false>, grid, dim3(256), 0, st, d, w.pfT, w.betaT, w.A, w.cams, w.Bpad, B, verts,
                           (float*)nullptr, 0.f, 0.f);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (o->joints || o->kp2d) {
        e = hpe_launch_joint_regress(verts, d.kp_reg, B, d.num_kp, o->joints, w.cams, o->kp2d, st);
    }
    return e;
}
