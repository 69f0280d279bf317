// Rasterizer stage 8: per-Gaussian chain rule from the blending gradients
// (pixel position, conic, opacity, colour) back to means3D, scales, rotations,
// SH coefficients and the means2D gradient carrier.  One thread per Gaussian;
// HBM-bound (reads 59 input floats + two 48-byte records, writes 62 floats).
#include "fdgs_common.h"

__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(fdgs_preprocess_bwd_args a) {
    __shared__ float sh_tile[256 * FDGS_SH_TILE_STRIDE];
    const int first = blockIdx.x * 256;
    const int rows = min(256, a.num_points - first);
    const bool inb = (int)threadIdx.x < rows;
    const int i = first + (inb ? (int)threadIdx.x : 0);
    const bool use_sh = a.colors_precomp == nullptr, use_cov = a.cov3D_precomp != nullptr;
    if (use_sh) {                                   // SH coefficients in; the same rows carry dL/dSH out
        sh_tile_copy<true>(sh_tile, const_cast<float*>(a.shs) + (size_t)first * 48, rows);
        __syncthreads();
    }
    const float4* __restrict__ geom4 = reinterpret_cast<const float4*>(a.geom) + 3 * (size_t)i;
    const float4* __restrict__ dg4 = reinterpret_cast<const float4*>(a.dL_dgeom) + 3 * (size_t)i;
    const float4 g2 = geom4[2];
    const int radius = fdgs_unpack_radius(__float_as_int(g2.z));
    const int clampmask = fdgs_unpack_clampmask(__float_as_int(g2.z));

    float dmean[3] = {0.f, 0.f, 0.f}, dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dop = 0.f, dm2x = 0.f, dm2y = 0.f;
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dcolor[3] = {0.f, 0.f, 0.f};
    float* dsh = sh_tile + threadIdx.x * FDGS_SH_TILE_STRIDE;      // this thread's row: read as SH, overwritten with dL/dSH

    if (radius <= 0) {
        if (use_sh) {
#pragma unroll
            for (int k = 0; k < 48; ++k) dsh[k] = 0.f;
        }
    } else {
        const float4 d0 = dg4[0], d1 = dg4[1], d2 = dg4[2];
        if (a.clear_dL_dgeom && inb) {               // only splats with a radius are ever accumulated into
            float4* z = const_cast<float4*>(dg4);
            z[0] = z[1] = z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float g_px = FDGS_DGEOM_XY * d0.x, g_py = FDGS_DGEOM_XY * d0.y;
        const float gA = FDGS_DGEOM_AC * d0.z, gB = FDGS_DGEOM_B * d0.w, gC = FDGS_DGEOM_AC * d1.x;
        dop = d1.y;
        float gcol[3] = {d1.z, d1.w, d2.x};
        const int W = a.image_width, H = a.image_height;
        const float* __restrict__ V = a.viewmatrix;
        const float* __restrict__ P = a.projmatrix;
        const float x = a.means3D[3 * i + 0], y = a.means3D[3 * i + 1], z = a.means3D[3 * i + 2];

        // ---- colour: SH backward (clamped channels pass no gradient), or straight to colors_precomp ----
        if (!use_sh) {
            dcolor[0] = gcol[0]; dcolor[1] = gcol[1]; dcolor[2] = gcol[2];
        } else {
        const float ddx = x - a.campos[0], ddy = y - a.campos[1], ddz = z - a.campos[2];
        const float nrm = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
        const float inv_n = 1.0f / nrm;
        const float ux = ddx * inv_n, uy = ddy * inv_n, uz = ddz * inv_n;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (clampmask & (1 << c)) gcol[c] = 0.f;
        const float* sh = dsh;
        float basis[16], bdx[16], bdy[16], bdz[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { basis[k] = 0.f; bdx[k] = 0.f; bdy[k] = 0.f; bdz[k] = 0.f; }
        basis[0] = SH_C0;
        if (a.sh_degree > 0) {
            basis[1] = -SH_C1 * uy; bdy[1] = -SH_C1;
            basis[2] = SH_C1 * uz;  bdz[2] = SH_C1;
            basis[3] = -SH_C1 * ux; bdx[3] = -SH_C1;
            if (a.sh_degree > 1) {
                const float xx = ux * ux, yy = uy * uy, zz = uz * uz, xy = ux * uy, yz = uy * uz, xz = ux * uz;
                basis[4] = SH_C2_0 * xy; bdx[4] = SH_C2_0 * uy; bdy[4] = SH_C2_0 * ux;
                basis[5] = SH_C2_1 * yz; bdy[5] = SH_C2_1 * uz; bdz[5] = SH_C2_1 * uy;
                basis[6] = SH_C2_2 * (2.0f * zz - xx - yy);
                bdx[6] = -2.0f * SH_C2_2 * ux; bdy[6] = -2.0f * SH_C2_2 * uy; bdz[6] = 4.0f * SH_C2_2 * uz;
                basis[7] = SH_C2_3 * xz; bdx[7] = SH_C2_3 * uz; bdz[7] = SH_C2_3 * ux;
                basis[8] = SH_C2_4 * (xx - yy); bdx[8] = 2.0f * SH_C2_4 * ux; bdy[8] = -2.0f * SH_C2_4 * uy;
                if (a.sh_degree > 2) {
                    basis[9] = SH_C3_0 * uy * (3.0f * xx - yy);
                    bdx[9] = SH_C3_0 * 6.0f * xy; bdy[9] = SH_C3_0 * (3.0f * xx - 3.0f * yy);
                    basis[10] = SH_C3_1 * xy * uz;
                    bdx[10] = SH_C3_1 * yz; bdy[10] = SH_C3_1 * xz; bdz[10] = SH_C3_1 * xy;
                    basis[11] = SH_C3_2 * uy * (4.0f * zz - xx - yy);
                    bdx[11] = SH_C3_2 * (-2.0f * xy); bdy[11] = SH_C3_2 * (4.0f * zz - xx - 3.0f * yy);
                    bdz[11] = SH_C3_2 * 8.0f * yz;
                    basis[12] = SH_C3_3 * uz * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                    bdx[12] = SH_C3_3 * (-6.0f * xz); bdy[12] = SH_C3_3 * (-6.0f * yz);
                    bdz[12] = SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                    basis[13] = SH_C3_4 * ux * (4.0f * zz - xx - yy);
                    bdx[13] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); bdy[13] = SH_C3_4 * (-2.0f * xy);
                    bdz[13] = SH_C3_4 * 8.0f * xz;
                    basis[14] = SH_C3_5 * uz * (xx - yy);
                    bdx[14] = SH_C3_5 * 2.0f * xz; bdy[14] = SH_C3_5 * (-2.0f * yz); bdz[14] = SH_C3_5 * (xx - yy);
                    basis[15] = SH_C3_6 * ux * (xx - 3.0f * yy);
                    bdx[15] = SH_C3_6 * (3.0f * xx - 3.0f * yy); bdy[15] = SH_C3_6 * (-6.0f * xy);
                }
            }
        }
        float gux = 0.f, guy = 0.f, guz = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dot += sh[3 * k + c] * gcol[c];
                dsh[3 * k + c] = basis[k] * gcol[c];
            }
            gux += bdx[k] * dot; guy += bdy[k] * dot; guz += bdz[k] * dot;
        }
        // through dir = d/|d|
        const float udot = ux * gux + uy * guy + uz * guz;
        dmean[0] += (gux - ux * udot) * inv_n;
        dmean[1] += (guy - uy * udot) * inv_n;
        dmean[2] += (guz - uz * udot) * inv_n;
        }

        // ---- recompute the forward geometry ----
        const float pvx = V[0] * x + V[4] * y + V[8] * z + V[12];
        const float pvy = V[1] * x + V[5] * y + V[9] * z + V[13];
        const float pvz = V[2] * x + V[6] * y + V[10] * z + V[14];
        const float mod = a.scale_modifier;
        float sx = 0.f, sy = 0.f, sz = 0.f, qr = 1.f, qx = 0.f, qy = 0.f, qz = 0.f, qinv = 1.f;
        float R00 = 0.f, R01 = 0.f, R02 = 0.f, R10 = 0.f, R11 = 0.f, R12 = 0.f, R20 = 0.f, R21 = 0.f, R22 = 0.f;
        float M00 = 0.f, M01 = 0.f, M02 = 0.f, M10 = 0.f, M11 = 0.f, M12 = 0.f, M20 = 0.f, M21 = 0.f, M22 = 0.f;
        float c00, c01, c02, c11, c12, c22;
        if (use_cov) {
            const float* __restrict__ cv = a.cov3D_precomp + (size_t)i * 6;
            c00 = cv[0]; c01 = cv[1]; c02 = cv[2]; c11 = cv[3]; c12 = cv[4]; c22 = cv[5];
        } else {
            sx = a.scales[3 * i + 0]; sy = a.scales[3 * i + 1]; sz = a.scales[3 * i + 2];
            if (a.activations & FDGS_ACT_SCALE_EXP) { sx = expf(sx); sy = expf(sy); sz = expf(sz); }
            sx *= mod; sy *= mod; sz *= mod;
            qr = a.rotations[4 * i + 0]; qx = a.rotations[4 * i + 1]; qy = a.rotations[4 * i + 2]; qz = a.rotations[4 * i + 3];
            if (a.activations & FDGS_ACT_ROTATION_NORMALIZE) {
                qinv = 1.0f / fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
                qr *= qinv; qx *= qinv; qy *= qinv; qz *= qinv;
            }
            R00 = 1.0f - 2.0f * (qy * qy + qz * qz); R01 = 2.0f * (qx * qy - qr * qz); R02 = 2.0f * (qx * qz + qr * qy);
            R10 = 2.0f * (qx * qy + qr * qz); R11 = 1.0f - 2.0f * (qx * qx + qz * qz); R12 = 2.0f * (qy * qz - qr * qx);
            R20 = 2.0f * (qx * qz - qr * qy); R21 = 2.0f * (qy * qz + qr * qx); R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
            M00 = R00 * sx; M01 = R01 * sy; M02 = R02 * sz;
            M10 = R10 * sx; M11 = R11 * sy; M12 = R12 * sz;
            M20 = R20 * sx; M21 = R21 * sy; M22 = R22 * sz;
            c00 = M00 * M00 + M01 * M01 + M02 * M02; c01 = M00 * M10 + M01 * M11 + M02 * M12;
            c02 = M00 * M20 + M01 * M21 + M02 * M22; c11 = M10 * M10 + M11 * M11 + M12 * M12;
            c12 = M10 * M20 + M11 * M21 + M12 * M22; c22 = M20 * M20 + M21 * M21 + M22 * M22;
        }

        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float fx = (float)W / (2.0f * a.tanfovx), fy = (float)H / (2.0f * a.tanfovy);
        const float tz = pvz, itz = 1.0f / tz, itz2 = itz * itz;
        const float rxz = pvx * itz, ryz = pvy * itz;
        const bool clx = (rxz < -limx) || (rxz > limx), cly = (ryz < -limy) || (ryz > limy);
        const float lx = fminf(limx, fmaxf(-limx, rxz)), ly = fminf(limy, fmaxf(-limy, ryz));
        const float tx = lx * tz, ty = ly * tz;
        const float J00 = fx * itz, J02 = -(fx * tx) * itz2, J11 = fy * itz, J12 = -(fy * ty) * itz2;
        const float T00 = J00 * V[0] + J02 * V[2], T01 = J00 * V[4] + J02 * V[6], T02 = J00 * V[8] + J02 * V[10];
        const float T10 = J11 * V[1] + J12 * V[2], T11 = J11 * V[5] + J12 * V[6], T12 = J11 * V[9] + J12 * V[10];
        const float U00 = T00 * c00 + T01 * c01 + T02 * c02, U01 = T00 * c01 + T01 * c11 + T02 * c12;
        const float U02 = T00 * c02 + T01 * c12 + T02 * c22;
        const float U10 = T10 * c00 + T11 * c01 + T12 * c02, U11 = T10 * c01 + T11 * c11 + T12 * c12;
        const float U12 = T10 * c02 + T11 * c12 + T12 * c22;
        const float ca = U00 * T00 + U01 * T01 + U02 * T02 + FDGS_COV_DILATION;
        const float cb = U00 * T10 + U01 * T11 + U02 * T12;
        const float cc = U10 * T10 + U11 * T11 + U12 * T12 + FDGS_COV_DILATION;
        const float det = ca * cc - cb * cb;
        const float idet2 = 1.0f / (det * det);

        // ---- conic (A,B,C) = (c,-b,a)/det  ->  (a,b,c) ----
        const float ga = (-cc * cc * gA + cb * cc * gB - cb * cb * gC) * idet2;
        const float gb = (2.0f * cb * cc * gA - (ca * cc + cb * cb) * gB + 2.0f * ca * cb * gC) * idet2;
        const float gc = (-cb * cb * gA + ca * cb * gB - ca * ca * gC) * idet2;

        // ---- cov2D = T Sigma T^T ----
        // unique Sigma entries
        const float gS00 = ga * T00 * T00 + gb * T00 * T10 + gc * T10 * T10;
        const float gS11 = ga * T01 * T01 + gb * T01 * T11 + gc * T11 * T11;
        const float gS22 = ga * T02 * T02 + gb * T02 * T12 + gc * T12 * T12;
        const float gS01 = 2.0f * ga * T00 * T01 + gb * (T00 * T11 + T01 * T10) + 2.0f * gc * T10 * T11;
        const float gS02 = 2.0f * ga * T00 * T02 + gb * (T00 * T12 + T02 * T10) + 2.0f * gc * T10 * T12;
        const float gS12 = 2.0f * ga * T01 * T02 + gb * (T01 * T12 + T02 * T11) + 2.0f * gc * T11 * T12;
        const float gT00 = 2.0f * ga * U00 + gb * U10, gT01 = 2.0f * ga * U01 + gb * U11, gT02 = 2.0f * ga * U02 + gb * U12;
        const float gT10 = gb * U00 + 2.0f * gc * U10, gT11 = gb * U01 + 2.0f * gc * U11, gT12 = gb * U02 + 2.0f * gc * U12;

        // ---- T = J Wr ----
        const float gJ00 = gT00 * V[0] + gT01 * V[4] + gT02 * V[8];
        const float gJ02 = gT00 * V[2] + gT01 * V[6] + gT02 * V[10];
        const float gJ11 = gT10 * V[1] + gT11 * V[5] + gT12 * V[9];
        const float gJ12 = gT10 * V[2] + gT11 * V[6] + gT12 * V[10];
        // J00 = fx/tz, J11 = fy/tz, J02 = -fx*tx'/tz^2 with tx' = clamp(pvx/tz)*tz
        const float itz3 = itz2 * itz;
        float gpvx = clx ? 0.f : (-fx * itz2) * gJ02;
        float gpvy = cly ? 0.f : (-fy * itz2) * gJ12;
        float gpvz = -fx * itz2 * gJ00 - fy * itz2 * gJ11;
        gpvz += (clx ? (fx * lx * itz2) : (2.0f * fx * pvx * itz3)) * gJ02;
        gpvz += (cly ? (fy * ly * itz2) : (2.0f * fy * pvy * itz3)) * gJ12;
        dmean[0] += gpvx * V[0] + gpvy * V[1] + gpvz * V[2];
        dmean[1] += gpvx * V[4] + gpvy * V[5] + gpvz * V[6];
        dmean[2] += gpvx * V[8] + gpvy * V[9] + gpvz * V[10];

        if (use_cov) {
            dcov[0] = gS00; dcov[1] = gS01; dcov[2] = gS02; dcov[3] = gS11; dcov[4] = gS12; dcov[5] = gS22;
        } else {
            // ---- Sigma = M M^T, M = R diag(s) ----
            // symmetric gradient matrix Gs: diagonal as is, off-diagonal halves
            const float h01 = 0.5f * gS01, h02 = 0.5f * gS02, h12 = 0.5f * gS12;
            const float gM00 = 2.0f * (gS00 * M00 + h01 * M10 + h02 * M20);
            const float gM01 = 2.0f * (gS00 * M01 + h01 * M11 + h02 * M21);
            const float gM02 = 2.0f * (gS00 * M02 + h01 * M12 + h02 * M22);
            const float gM10 = 2.0f * (h01 * M00 + gS11 * M10 + h12 * M20);
            const float gM11 = 2.0f * (h01 * M01 + gS11 * M11 + h12 * M21);
            const float gM12 = 2.0f * (h01 * M02 + gS11 * M12 + h12 * M22);
            const float gM20 = 2.0f * (h02 * M00 + h12 * M10 + gS22 * M20);
            const float gM21 = 2.0f * (h02 * M01 + h12 * M11 + gS22 * M21);
            const float gM22 = 2.0f * (h02 * M02 + h12 * M12 + gS22 * M22);
            dscale[0] = mod * (gM00 * R00 + gM10 * R10 + gM20 * R20);
            dscale[1] = mod * (gM01 * R01 + gM11 * R11 + gM21 * R21);
            dscale[2] = mod * (gM02 * R02 + gM12 * R12 + gM22 * R22);
            const float r00 = gM00 * sx, r01 = gM01 * sy, r02 = gM02 * sz;
            const float r10 = gM10 * sx, r11 = gM11 * sy, r12 = gM12 * sz;
            const float r20 = gM20 * sx, r21 = gM21 * sy, r22 = gM22 * sz;
            drot[0] = 2.0f * (-qz * r01 + qy * r02 + qz * r10 - qx * r12 - qy * r20 + qx * r21);
            drot[1] = 2.0f * (qy * r01 + qz * r02 + qy * r10 - 2.0f * qx * r11 - qr * r12 + qz * r20 + qr * r21 - 2.0f * qx * r22);
            drot[2] = 2.0f * (-2.0f * qy * r00 + qx * r01 + qr * r02 + qx * r10 + qz * r12 - qr * r20 + qz * r21 - 2.0f * qy * r22);
            drot[3] = 2.0f * (-2.0f * qz * r00 - qr * r01 + qx * r02 + qr * r10 - 2.0f * qz * r11 + qy * r12 + qx * r20 + qy * r21);
            // chain through the activations: d exp = exp, d (q/|q|) = (I - q^ q^T) / |q|
            if (a.activations & FDGS_ACT_SCALE_EXP) {       // mod * exp(raw) = sx
                dscale[0] *= sx / mod; dscale[1] *= sy / mod; dscale[2] *= sz / mod;
            }
            if (a.activations & FDGS_ACT_ROTATION_NORMALIZE) {
                const float along = qinv > 0.5e12f ? 0.0f : (qr * drot[0] + qx * drot[1] + qy * drot[2] + qz * drot[3]);
                drot[0] = (drot[0] - qr * along) * qinv; drot[1] = (drot[1] - qx * along) * qinv;
                drot[2] = (drot[2] - qy * along) * qinv; drot[3] = (drot[3] - qz * along) * qinv;
            }
        }
        if (a.activations & FDGS_ACT_OPACITY_SIGMOID) {     // geom keeps the activated opacity
            const float o = geom4[1].y;
            dop *= o * (1.0f - o);
        }

        // ---- pixel position through the perspective projection ----
        dm2x = 0.5f * (float)W * g_px;     // dL/d(ndc x)
        dm2y = 0.5f * (float)H * g_py;
        const float phx = P[0] * x + P[4] * y + P[8] * z + P[12];
        const float phy = P[1] * x + P[5] * y + P[9] * z + P[13];
        const float phw = P[3] * x + P[7] * y + P[11] * z + P[15];
        const float p_w = 1.0f / (phw + 1e-7f);
        const float gphx = dm2x * p_w, gphy = dm2y * p_w;
        const float gphw = -(dm2x * phx + dm2y * phy) * p_w * p_w;
        dmean[0] += gphx * P[0] + gphy * P[1] + gphw * P[3];
        dmean[1] += gphx * P[4] + gphy * P[5] + gphw * P[7];
        dmean[2] += gphx * P[8] + gphy * P[9] + gphw * P[11];
    }
    if (use_sh) {
        __syncthreads();
        sh_tile_copy<false>(sh_tile, a.dL_dshs + (size_t)first * 48, rows);
    }
    if (!inb) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.dL_dmeans3D[3 * i + k] = dmean[k];
    if (use_cov) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * (size_t)i + k] = dcov[k];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dL_dscales[3 * i + k] = dscale[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) a.dL_drotations[4 * i + k] = drot[k];
    }
    if (!use_sh) {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dL_dcolors[3 * i + k] = dcolor[k];
    }
    a.dL_dopacities[i] = dop;
    a.dL_dmeans2D[3 * i + 0] = dm2x;
    a.dL_dmeans2D[3 * i + 1] = dm2y;
    a.dL_dmeans2D[3 * i + 2] = 0.f;
}

extern "C" int fdgs_raster_preprocess_bwd(const fdgs_preprocess_bwd_args* a, void* stream) {
    if (!a || a->num_points < 0 || a->image_width <= 0 || a->image_height <= 0) return FDGS_EINVAL;
    if (a->sh_degree < 0 || a->sh_degree > 3) return FDGS_EINVAL;
    if (a->num_points == 0) return FDGS_OK;
    if (!a->viewmatrix || !a->projmatrix || !a->campos || !a->means3D || !a->geom || !a->dL_dgeom ||
        !a->dL_dmeans3D || !a->dL_dmeans2D || !a->dL_dopacities)
        return FDGS_EINVAL;
    if (a->colors_precomp ? !a->dL_dcolors : (!a->shs || !a->dL_dshs)) return FDGS_EINVAL;
    if (a->cov3D_precomp ? !a->dL_dcov3D : (!a->scales || !a->rotations || !a->dL_dscales || !a->dL_drotations))
        return FDGS_EINVAL;
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((a->num_points + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       *a);
    FDGS_CHECK_LAUNCH();
    return FDGS_OK;
}
