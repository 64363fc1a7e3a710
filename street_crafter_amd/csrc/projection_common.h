// The per-(camera, Gaussian) projection arithmetic of SURVEY.md A.1, shared by projection_fwd_kernel
// (projection.hip) and the fused projection + SH kernel (fused_fwd.hip).  Compiled without FMA
// contraction, in oracle/gsplat_oracle.py's op order: bit-identical to the oracle wherever it is used.
#pragma once
#include "sc_common.h"

#pragma clang fp contract(off)

struct ProjOpt { int clamp; float radius_floor; };
extern int g_sc_proj_clamp, g_sc_radius_floor;       // sc_set_option "proj_clamp" / "radius_floor" (capi.hip)

namespace {

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

struct Cam {
    float W00, W01, W02, tx, W10, W11, W12, ty, W20, W21, W22, tz;
    float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ V, const float* __restrict__ K) {
    Cam c;
    c.W00 = V[0]; c.W01 = V[1]; c.W02 = V[2];  c.tx = V[3];
    c.W10 = V[4]; c.W11 = V[5]; c.W12 = V[6];  c.ty = V[7];
    c.W20 = V[8]; c.W21 = V[9]; c.W22 = V[10]; c.tz = V[11];
    c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
    return c;
}

// The two constants of SURVEY A.1 that differ between upstream gsplat versions (the reference installs an unpinned fork,
// README.md:35), selectable at run time: sc_set_option("proj_clamp") 0 = x/z, y/z clamped to +-1.3 tan(fov/2) (v1.0-1.3,
// default), 1 = to [-(cx/fx + 0.3 tan), (W - cx)/fx + 0.3 tan] (v1.4+); sc_set_option("radius_floor") 0 = 0.01 (v1.x,
// default), 1 = 0.1 (the original Inria rasterizer).  oracle/gsplat_oracle.py takes the same two arguments.
static inline ProjOpt sc_proj_opt() { return ProjOpt{g_sc_proj_clamp, g_sc_radius_floor ? 0.1f : 0.01f}; }

struct ProjLim { float xp, xn, yp, yn; };       // tan-space limits: x/z in [-xn, xp], y/z in [-yn, yp]
__device__ __forceinline__ ProjLim proj_limits(const ProjOpt& opt, float fx, float fy, float cx, float cy, int width, int height) {
    const float tanx = 0.5f * (float)width / fx;
    const float tany = 0.5f * (float)height / fy;
    ProjLim l;
    if (opt.clamp == 0) {
        l.xp = l.xn = 1.3f * tanx;
        l.yp = l.yn = 1.3f * tany;
    } else {
        l.xp = ((float)width - cx) / fx + 0.3f * tanx;
        l.xn = cx / fx + 0.3f * tanx;
        l.yp = ((float)height - cy) / fy + 0.3f * tany;
        l.yn = cy / fy + 0.3f * tany;
    }
    return l;
}

struct ProjOut {
    int rad_i;
    float m2x, m2y, con0, con1, con2, comp, depth;
};

__device__ __forceinline__ ProjOut project_one(const Cam& c, const float* __restrict__ means,
                                               const float* __restrict__ quats,
                                               const float* __restrict__ scales, int n, int width, int height,
                                               float eps2d, float near_plane, float far_plane,
                                               float radius_clip, const ProjOpt& opt) {
    const float mx = means[n * 3 + 0], my = means[n * 3 + 1], mz = means[n * 3 + 2];
    const float x = dot3(c.W00, mx, c.W01, my, c.W02, mz) + c.tx;
    const float y = dot3(c.W10, mx, c.W11, my, c.W12, mz) + c.ty;
    const float z = dot3(c.W20, mx, c.W21, my, c.W22, mz) + c.tz;

    bool valid = !((z < near_plane) || (z > far_plane));
    int rad_i = 0;
    float m2x = 0.f, m2y = 0.f, con0 = 0.f, con1 = 0.f, con2 = 0.f, comp = 0.f, depth = 0.f;
    if (valid) {
        const float4 q = *reinterpret_cast<const float4*>(quats + (size_t)n * 4);
        float qw = q.x, qx = q.y, qy = q.z, qz = q.w;
        const float n2 = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
        const float inv = 1.0f / sqrtf(n2);
        qw *= inv; qx *= inv; qy *= inv; qz *= inv;
        const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
        const float R00 = 1.0f - 2.0f * (y2 + z2), R01 = 2.0f * (xy - wz), R02 = 2.0f * (xz + wy);
        const float R10 = 2.0f * (xy + wz), R11 = 1.0f - 2.0f * (x2 + z2), R12 = 2.0f * (yz - wx);
        const float R20 = 2.0f * (xz - wy), R21 = 2.0f * (yz + wx), R22 = 1.0f - 2.0f * (x2 + y2);
        const float s0 = scales[n * 3 + 0], s1 = scales[n * 3 + 1], s2 = scales[n * 3 + 2];
        const float M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
        const float M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
        const float M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
        const float S00 = dot3(M00, M00, M01, M01, M02, M02);
        const float S01 = dot3(M00, M10, M01, M11, M02, M12);
        const float S02 = dot3(M00, M20, M01, M21, M02, M22);
        const float S11 = dot3(M10, M10, M11, M11, M12, M12);
        const float S12 = dot3(M10, M20, M11, M21, M12, M22);
        const float S22 = dot3(M20, M20, M21, M21, M22, M22);
        // T = W * Sigma
        const float T00 = dot3(c.W00, S00, c.W01, S01, c.W02, S02);
        const float T01 = dot3(c.W00, S01, c.W01, S11, c.W02, S12);
        const float T02 = dot3(c.W00, S02, c.W01, S12, c.W02, S22);
        const float T10 = dot3(c.W10, S00, c.W11, S01, c.W12, S02);
        const float T11 = dot3(c.W10, S01, c.W11, S11, c.W12, S12);
        const float T12 = dot3(c.W10, S02, c.W11, S12, c.W12, S22);
        const float T20 = dot3(c.W20, S00, c.W21, S01, c.W22, S02);
        const float T21 = dot3(c.W20, S01, c.W21, S11, c.W22, S12);
        const float T22 = dot3(c.W20, S02, c.W21, S12, c.W22, S22);
        // Sigma_c = T * W^T
        const float c00 = dot3(T00, c.W00, T01, c.W01, T02, c.W02);
        const float c01 = dot3(T00, c.W10, T01, c.W11, T02, c.W12);
        const float c02 = dot3(T00, c.W20, T01, c.W21, T02, c.W22);
        const float c11 = dot3(T10, c.W10, T11, c.W11, T12, c.W12);
        const float c12 = dot3(T10, c.W20, T11, c.W21, T12, c.W22);
        const float c22 = dot3(T20, c.W20, T21, c.W21, T22, c.W22);

        const ProjLim lim = proj_limits(opt, c.fx, c.fy, c.cx, c.cy, width, height);
        const float rz = 1.0f / z;
        const float rz2 = rz * rz;
        const float tx = z * fminf(lim.xp, fmaxf(-lim.xn, x * rz));
        const float ty = z * fminf(lim.yp, fmaxf(-lim.yn, y * rz));
        const float ja = c.fx * rz;
        const float jb = ((-c.fx) * tx) * rz2;
        const float jc = c.fy * rz;
        const float jd = ((-c.fy) * ty) * rz2;
        const float u0 = ja * c00 + jb * c02;
        const float u1 = ja * c01 + jb * c12;
        const float u2 = ja * c02 + jb * c22;
        const float v1 = jc * c11 + jd * c12;
        const float v2 = jc * c12 + jd * c22;
        const float a = u0 * ja + u2 * jb;
        const float b = u1 * jc + u2 * jd;
        const float cc = v1 * jc + v2 * jd;
        m2x = (c.fx * x) * rz + c.cx;
        m2y = (c.fy * y) * rz + c.cy;

        const float det0 = a * cc - b * b;
        const float a1 = a + eps2d;
        const float c1 = cc + eps2d;
        const float det1 = a1 * c1 - b * b;
        comp = sqrtf(fmaxf(0.0f, det0 / det1));
        valid = det1 > 0.0f;  // false for NaN too
        con0 = c1 / det1;
        con1 = (-b) / det1;
        con2 = a1 / det1;
        const float bb = 0.5f * (a1 + c1);
        const float lam = bb + sqrtf(fmaxf(opt.radius_floor, bb * bb - det1));
        const float radius = ceilf(3.0f * sqrtf(lam));
        valid = valid && (radius > radius_clip);  // false for NaN
        const float Wf = (float)width, Hf = (float)height;
        valid = valid && !((m2x + radius <= 0.0f) || (m2x - radius >= Wf) ||
                           (m2y + radius <= 0.0f) || (m2y - radius >= Hf));
        rad_i = (int)fminf(radius, 2147483520.0f);
        depth = z;
    }
    if (!valid) {
        rad_i = 0; m2x = m2y = con0 = con1 = con2 = comp = depth = 0.f;
    }
    ProjOut r;
    r.rad_i = rad_i; r.m2x = m2x; r.m2y = m2y; r.con0 = con0; r.con1 = con1; r.con2 = con2;
    r.comp = comp; r.depth = depth;
    return r;
}

}  // namespace

#pragma clang fp contract(fast)
