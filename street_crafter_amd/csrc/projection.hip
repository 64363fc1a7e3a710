// a1: fully_fused_projection forward / backward for gfx950.
// Replaces gsplat.rendering.fully_fused_projection as called at
// street_gaussian/models/street_gaussian_renderer.py:219-232 (semantics: SURVEY.md A.1).
//
// HBM-bound streaming kernel: 40 B in / 32 B out per Gaussian, one lane per (camera, gaussian).
// The forward feeds INTEGER decisions downstream (radii -> tile rectangles, depth bits -> sort
// key), so it is compiled without FMA contraction and mirrors oracle/gsplat_oracle.py's op
// order exactly: its outputs are bit-identical to the oracle's, not merely close.
#include "projection_common.h"

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void projection_fwd_kernel(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, int N, int width, int height,
    float eps2d, float near_plane, float far_plane, float radius_clip, ProjOpt opt,
    int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ depths,
    float* __restrict__ conics, float* __restrict__ comps) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int cam = blockIdx.y;
    if (n >= N) return;
    const Cam c = load_cam(viewmats + cam * 16, Ks + cam * 9);
    const size_t o = (size_t)cam * N + n;

    const ProjOut p = project_one(c, means, quats, scales, n, width, height, eps2d, near_plane, far_plane,
                                  radius_clip, opt);
    radii[o] = p.rad_i;
    *reinterpret_cast<float2*>(means2d + o * 2) = make_float2(p.m2x, p.m2y);
    depths[o] = p.depth;
    conics[o * 3 + 0] = p.con0;
    conics[o * 3 + 1] = p.con1;
    conics[o * 3 + 2] = p.con2;
    if (comps) comps[o] = p.comp;
}

}  // namespace

#pragma clang fp contract(fast)

namespace {

// ---- backward -----------------------------------------------------------------------------
// Recomputes the forward intermediates per (camera, gaussian) and chains the VJPs of
// SURVEY.md A.1 steps 1-5 (radius / cull are non-differentiable).  Gradients over cameras are
// summed with atomics only when C > 1.
__global__ __launch_bounds__(256) void projection_bwd_kernel(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, int C, int N, int width,
    int height, float eps2d, ProjOpt opt, const int32_t* __restrict__ radii, const float* __restrict__ conics,
    const float* __restrict__ comps, const float* __restrict__ v_means2d,
    const float* __restrict__ v_depths, const float* __restrict__ v_conics,
    const float* __restrict__ v_comps, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float gm[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f};

    const float mx = means[n * 3 + 0], my = means[n * 3 + 1], mz = means[n * 3 + 2];
    const float4 q4 = *reinterpret_cast<const float4*>(quats + (size_t)n * 4);
    const float s0 = scales[n * 3 + 0], s1 = scales[n * 3 + 1], s2 = scales[n * 3 + 2];
    // normalised quaternion + rotation
    const float qn2 = q4.y * q4.y + q4.z * q4.z + q4.w * q4.w + q4.x * q4.x;
    const float inv = 1.0f / sqrtf(qn2);
    const float qw = q4.x * inv, qx = q4.y * inv, qy = q4.z * inv, qz = q4.w * inv;
    float R[3][3];
    R[0][0] = 1.f - 2.f * (qy * qy + qz * qz); R[0][1] = 2.f * (qx * qy - qw * qz); R[0][2] = 2.f * (qx * qz + qw * qy);
    R[1][0] = 2.f * (qx * qy + qw * qz); R[1][1] = 1.f - 2.f * (qx * qx + qz * qz); R[1][2] = 2.f * (qy * qz - qw * qx);
    R[2][0] = 2.f * (qx * qz - qw * qy); R[2][1] = 2.f * (qy * qz + qw * qx); R[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
    const float sc[3] = {s0, s1, s2};
    float M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * sc[j];
    float S[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S[i][j] = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];

    float vS[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // dL/dSigma (world)

    for (int cam = 0; cam < C; ++cam) {
        const size_t o = (size_t)cam * N + n;
        if (radii[o] <= 0) continue;
        const float* V = viewmats + cam * 16;
        const float* K = Ks + cam * 9;
        float Wm[3][3] = {{V[0], V[1], V[2]}, {V[4], V[5], V[6]}, {V[8], V[9], V[10]}};
        const float fx = K[0], fy = K[4];
        const float x = Wm[0][0] * mx + Wm[0][1] * my + Wm[0][2] * mz + V[3];
        const float y = Wm[1][0] * mx + Wm[1][1] * my + Wm[1][2] * mz + V[7];
        const float z = Wm[2][0] * mx + Wm[2][1] * my + Wm[2][2] * mz + V[11];
        // Sigma_c = W S W^T
        float T[3][3], Sc[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) T[i][j] = Wm[i][0] * S[0][j] + Wm[i][1] * S[1][j] + Wm[i][2] * S[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Sc[i][j] = T[i][0] * Wm[j][0] + T[i][1] * Wm[j][1] + T[i][2] * Wm[j][2];

        const ProjLim lim = proj_limits(opt, fx, fy, K[2], K[5], width, height);
        const float rz = 1.f / z, rz2 = rz * rz;
        const float xr = x * rz, yr = y * rz;
        const bool clx = (xr < -lim.xn) || (xr > lim.xp);
        const bool cly = (yr < -lim.yn) || (yr > lim.yp);
        const float tx = z * fminf(lim.xp, fmaxf(-lim.xn, xr));
        const float ty = z * fminf(lim.yp, fmaxf(-lim.yn, yr));
        // J = [[ja,0,jb],[0,jc,jd]]
        const float ja = fx * rz, jb = -fx * tx * rz2, jc = fy * rz, jd = -fy * ty * rz2;
        // cov2d = J Sc J^T
        const float u0 = ja * Sc[0][0] + jb * Sc[2][0], u1 = ja * Sc[0][1] + jb * Sc[2][1], u2 = ja * Sc[0][2] + jb * Sc[2][2];
        const float w0 = jc * Sc[1][0] + jd * Sc[2][0], w1 = jc * Sc[1][1] + jd * Sc[2][1], w2 = jc * Sc[1][2] + jd * Sc[2][2];
        const float a = u0 * ja + u2 * jb, b = u1 * jc + u2 * jd, cc = w1 * jc + w2 * jd;
        const float a1 = a + eps2d, c1 = cc + eps2d;
        const float det1 = a1 * c1 - b * b;

        // ---- VJP: conics -> blurred cov2d.  conic = inv([[a1,b],[b,c1]]);  v_cov = -X^-1 V X^-1
        const float i00 = conics[o * 3 + 0], i01 = conics[o * 3 + 1], i11 = conics[o * 3 + 2];
        const float g0 = v_conics[o * 3 + 0], g1 = v_conics[o * 3 + 1] * 0.5f, g2 = v_conics[o * 3 + 2];
        // P = Xinv * G
        const float p00 = i00 * g0 + i01 * g1, p01 = i00 * g1 + i01 * g2;
        const float p10 = i01 * g0 + i11 * g1, p11 = i01 * g1 + i11 * g2;
        float va = -(p00 * i00 + p01 * i01);
        float vb = -((p00 * i01 + p01 * i11) + (p10 * i00 + p11 * i01));  // grad wrt the single b
        float vc = -(p10 * i01 + p11 * i11);
        // ---- VJP: compensation = sqrt(max(0, det0/det1))
        if (v_comps && comps) {
            const float comp = comps[o];
            const float vcomp = v_comps[o];
            if (comp > 0.f) {
                const float inv_det1 = 1.f / det1;
                const float one_m = 1.f - comp * comp;
                const float k = 0.5f * vcomp / comp * inv_det1;  // d comp / d ratio * (1/det1)
                // d ratio/da = (c - ratio*c1)/det1 ; d ratio/dc = (a - ratio*a1)/det1 ; d ratio/db = -2b(1-ratio)/det1
                const float ratio = comp * comp;
                va += k * (cc - ratio * c1);
                vc += k * (a - ratio * a1);
                vb += k * (-2.f * b * one_m);
            }
        }
        // ---- VJP: cov2d = J Sc J^T  ->  v_Sc = J^T Vc J ;  v_J = Vc J Sc^T + Vc^T J Sc
        const float h = 0.5f * vb;  // symmetric split of the b gradient
        // Vc = [[va,h],[h,vc]];  J rows: r0=(ja,0,jb) r1=(0,jc,jd)
        float vSc[3][3];
        {
            const float J0[3] = {ja, 0.f, jb}, J1[3] = {0.f, jc, jd};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    vSc[i][j] = J0[i] * (va * J0[j] + h * J1[j]) + J1[i] * (h * J0[j] + vc * J1[j]);
        }
        // v_J = 2 * Vc * J * Sc  (Sc symmetric)
        // (J Sc) rows are (u0,u1,u2) and (w0,w1,w2)
        const float vJ00 = 2.f * (va * u0 + h * w0);
        const float vJ02 = 2.f * (va * u2 + h * w2);
        const float vJ11 = 2.f * (h * u1 + vc * w1);
        const float vJ12 = 2.f * (h * u2 + vc * w2);

        // ---- VJP: means2d & depth & J -> camera-space mean
        const float vm2x = v_means2d[o * 2 + 0], vm2y = v_means2d[o * 2 + 1];
        float vx = fx * rz * vm2x;
        float vy = fy * rz * vm2y;
        float vz = -(fx * x * vm2x + fy * y * vm2y) * rz2 + v_depths[o];
        // ja = fx/z ; jc = fy/z
        vz += -fx * rz2 * vJ00 - fy * rz2 * vJ11;
        // jb = -fx*tx/z^2 with tx = x (unclamped) or z*lim*sign (clamped)
        const float rz3 = rz2 * rz;
        if (!clx) {
            vx += -fx * rz2 * vJ02;
            vz += 2.f * fx * tx * rz3 * vJ02;
        } else {
            // tx = z*k  ->  jb = -fx*k/z  ->  d/dz = fx*k/z^2 = fx*tx/z^3
            vz += fx * tx * rz3 * vJ02;
        }
        if (!cly) {
            vy += -fy * rz2 * vJ12;
            vz += 2.f * fy * ty * rz3 * vJ12;
        } else {
            vz += fy * ty * rz3 * vJ12;
        }
        // camera -> world mean
        gm[0] += Wm[0][0] * vx + Wm[1][0] * vy + Wm[2][0] * vz;
        gm[1] += Wm[0][1] * vx + Wm[1][1] * vy + Wm[2][1] * vz;
        gm[2] += Wm[0][2] * vx + Wm[1][2] * vy + Wm[2][2] * vz;
        // Sc = W S W^T -> v_S += W^T vSc W
        float Q[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Q[i][j] = vSc[i][0] * Wm[0][j] + vSc[i][1] * Wm[1][j] + vSc[i][2] * Wm[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vS[i][j] += Wm[0][i] * Q[0][j] + Wm[1][i] * Q[1][j] + Wm[2][i] * Q[2][j];
    }

    // ---- Sigma = M M^T -> v_M = (vS + vS^T) M ;  M = R diag(s)
    float vM[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            vM[i][j] = (vS[i][0] + vS[0][i]) * M[0][j] + (vS[i][1] + vS[1][i]) * M[1][j] + (vS[i][2] + vS[2][i]) * M[2][j];
    float vR[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gs[j] = R[0][j] * vM[0][j] + R[1][j] * vM[1][j] + R[2][j] * vM[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i) vR[i][j] = vM[i][j] * sc[j];
    }
    // rotation -> normalised quaternion (w,x,y,z)
    const float vqw = 2.f * (qx * (vR[2][1] - vR[1][2]) + qy * (vR[0][2] - vR[2][0]) + qz * (vR[1][0] - vR[0][1]));
    const float vqx = 2.f * (-2.f * qx * (vR[1][1] + vR[2][2]) + qy * (vR[1][0] + vR[0][1]) + qz * (vR[2][0] + vR[0][2]) + qw * (vR[2][1] - vR[1][2]));
    const float vqy = 2.f * (qx * (vR[1][0] + vR[0][1]) - 2.f * qy * (vR[0][0] + vR[2][2]) + qz * (vR[2][1] + vR[1][2]) + qw * (vR[0][2] - vR[2][0]));
    const float vqz = 2.f * (qx * (vR[2][0] + vR[0][2]) + qy * (vR[2][1] + vR[1][2]) - 2.f * qz * (vR[0][0] + vR[1][1]) + qw * (vR[1][0] - vR[0][1]));
    // through normalisation: v_q = (v_qn - (v_qn . qn) qn) / |q|
    const float dotp = vqw * qw + vqx * qx + vqy * qy + vqz * qz;
    gq[0] = (vqw - dotp * qw) * inv;
    gq[1] = (vqx - dotp * qx) * inv;
    gq[2] = (vqy - dotp * qy) * inv;
    gq[3] = (vqz - dotp * qz) * inv;

    v_means[n * 3 + 0] = gm[0]; v_means[n * 3 + 1] = gm[1]; v_means[n * 3 + 2] = gm[2];
    *reinterpret_cast<float4*>(v_quats + (size_t)n * 4) = make_float4(gq[0], gq[1], gq[2], gq[3]);
    v_scales[n * 3 + 0] = gs[0]; v_scales[n * 3 + 1] = gs[1]; v_scales[n * 3 + 2] = gs[2];
}

}  // namespace

extern "C" int sc_projection_fwd(const float* means, const float* quats, const float* scales,
                                 const float* viewmats, const float* Ks, int C, int N, int width,
                                 int height, float eps2d, float near_plane, float far_plane,
                                 float radius_clip, int32_t* radii, float* means2d, float* depths,
                                 float* conics, float* compensations, sc_stream_t stream) {
    if (C < 0 || N < 0 || width <= 0 || height <= 0) return SC_EINVAL;
    if (C == 0 || N == 0) return SC_OK;
    if (!means || !quats || !scales || !viewmats || !Ks || !radii || !means2d || !depths || !conics)
        return SC_EINVAL;
    if (C > 65535) return SC_EINVAL;
    dim3 grid((N + 255) / 256, C);
    hipLaunchKernelGGL(projection_fwd_kernel, grid, dim3(256), 0, sc_s(stream), means, quats, scales,
                       viewmats, Ks, N, width, height, eps2d, near_plane, far_plane, radius_clip, sc_proj_opt(),
                       radii, means2d, depths, conics, compensations);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_projection_bwd(const float* means, const float* quats, const float* scales,
                                 const float* viewmats, const float* Ks, int C, int N, int width,
                                 int height, float eps2d, const int32_t* radii, const float* conics,
                                 const float* compensations, const float* v_means2d,
                                 const float* v_depths, const float* v_conics,
                                 const float* v_compensations, float* v_means, float* v_quats,
                                 float* v_scales, sc_stream_t stream) {
    if (C < 0 || N < 0 || width <= 0 || height <= 0) return SC_EINVAL;
    if (N == 0) return SC_OK;
    if (!means || !quats || !scales || !viewmats || !Ks || !radii || !conics || !v_means2d ||
        !v_depths || !v_conics || !v_means || !v_quats || !v_scales)
        return SC_EINVAL;
    hipLaunchKernelGGL(projection_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, sc_s(stream),
                       means, quats, scales, viewmats, Ks, C, N, width, height, eps2d, sc_proj_opt(), radii, conics,
                       compensations, v_means2d, v_depths, v_conics, v_compensations, v_means,
                       v_quats, v_scales);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
