// clrs_assemble_w4.hip.h -- k_cluster_assemble_w4: the register-resident Schur assembly of k_cluster_assemble_w3 for LARGER simple blocks, gfx950.
//
// Same kind of cluster as k_cluster_assemble_w3 (every low-rank block "simple": one sub-block, rank-1 symmetric terms v_u v_u^T, one term per
// constraint, U = P and the same constraint order in every block; dense blocks 1 x 1) with block sides up to 32 and up to 16 NU <= 64 constraints:
// the univariate / multivariate sums-of-squares shapes beyond the Cohn-Elkies ones (PolyOpt 2d = 40: n = 21, P = 41), which until round 5 went
// through the LDS-staged general kernel at 4-5 % of the HBM roof.  src/solver.jl:1121-1212 per block, as in clrs_assemble_w3.hip.h:
//     T_Y = Y V,  G_Y = V^T T_Y,  Z = L^-1 V,  G_X = Z^T Z,  S += (lambda lambda^T) o G_X o G_Y
// on v_mfma_f64_16x16x4_f64 tiles, one wave per run of clusters, nothing synchronises.  What is new against w3:
//   * two row tiles (rows 0-15, 16-31) and eight k-steps: an accumulator tile (a, t) read as "k-step 4 a + reg" is the operand with contraction
//     index 16 a + 4 reg + l4 -- the same identity, tile by tile;
//   * L^-1 of a 32-row factor L = [A 0; C B] is never formed: Z_top = A^-1 V_top, Z_bot = B^-1 (V_bot - C Z_top).  A^-1 and B^-1 come from the
//     16-row DPP substitution of w3 on the two diagonal blocks; V_bot is an accumulator as it stands in operand order, C an A operand, Z_top an
//     accumulator read as B operand: three MFMA passes of four k-steps per column tile, no transposition anywhere;
//   * k-steps beyond ceil(n / 4) and the second row tile of blocks with n <= 16 are skipped (uniform branches);
//   * one wave per SIMD (launch bounds 256 x 1), no look-ahead across blocks (see PF below); S_j leaves through LDS in whole rows.
// NU = 3 (P <= 48) or 4 (P <= 64).
#pragma once
#include "clrs_assemble_w3.hip.h"

namespace clrs {

// Two workgroups (two waves per SIMD) per compute unit while a block's registers allow it (P <= 48: 256 registers per lane): one wave's loads and
// stores run beneath the other's matrix instructions.
#define W4_WGS_PER_CU(NU) ((NU) <= 3 ? 2 : 1)

struct W4Tables {
    const double *Xc, *Y;      // iterates (xy layout): Cholesky factors of the X blocks, Y blocks
    double *S, *AY;            // outputs
    const double *vop;         // per block NU x 512 doubles: vop[((t*4 + p)*64 + lane)*2 + e] = V[4 (2 p + e) + (lane >> 4), 16 t + (lane & 15)], zero padded
    const double *lam;         // [lam_off + u]: lambda of vector u; [dlam_off + u]: entries of the 1 x 1 dense blocks, in vector order
    const int *pmap;           // [pmap_off + u]: constraint of vector u
    const int *ay;             // [W3Block::ay_base + u]: position of the term of vector u in the A_Y output
    const W3Dense *dense;
};

// (No look-ahead across blocks.  Tried: the next block's operands in flight in a second register set during the arithmetic of the current one, as in w3 --
// 161 us against 120 us on 8192 blocks of PolyOpt 2d = 40 (n = 21, P = 41): two operand sets beside one block's accumulators leave the register allocator
// 928 bytes of scratch per lane and a stream of v_accvgpr moves, which cost more than the ~2 us of load latency per block they hide.)
template <int NU>
__global__ __launch_bounds__(256, W4_WGS_PER_CU(NU)) void k_cluster_assemble_w4(const int *__restrict__ cluster_blk0, const W3Block *__restrict__ blocks, const W4Tables tb,
                                                                                int nclusters, int nblocks, int s_rows) {
    constexpr int LD = 18;                                       // doubles per row of the staged L rows (as in w3)
    constexpr int NT = NU * (NU + 1) / 2;                        // lower tiles of a U x U matrix
    constexpr int LDS_S = 16 * NU + 2;                           // leading dimension of the staged S_j (even: 16-byte aligned pairs)
    const int PER_WAVE = 16 * LD + 16 + s_rows * LDS_S;          // L rows | diagonal of L | S_j (s_rows = the largest P of the launch: W4_LDS_BYTES)
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    double *Lt = lds_all + wave * PER_WAVE, *Ld = Lt + 16 * LD, *Ss = Ld + 16;
    const int c0 = (int)((long long)gw * nclusters / nw), c1 = (int)((long long)(gw + 1) * nclusters / nw);
    if (c0 >= c1) return;
    const int bbeg = cluster_blk0[c0], bend = (c1 < nclusters) ? cluster_blk0[c1] : nblocks;
    const int dreg = (l15 - l4) >> 2;                             // lanes with l15 = 4 dreg + l4 hold a diagonal entry of a diagonal tile
    const bool on_diag = l15 >= l4 && ((l15 - l4) & 3) == 0;

    v4d_f sacc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) sacc[t] = (v4d_f){0.0, 0.0, 0.0, 0.0};

    // inverse of one 16 x 16 diagonal block of L (rows / columns r0 .. r0 + 15) in A-operand order: x[q] = W[l15, 4 q + l4]
    auto diag_inverse = [&](const double (&lt)[4], int rows, double (&x)[4]) {
        {
            double dsel = lt[0];
            dsel = (dreg == 1) ? lt[1] : dsel;
            dsel = (dreg == 2) ? lt[2] : dsel;
            dsel = (dreg == 3) ? lt[3] : dsel;
            if (on_diag) Ld[l15] = dsel;
        }
        wave_sync();
        const double dg = Ld[l15];
        double di = __builtin_amdgcn_rcp(dg);
        di = __builtin_fma(__builtin_fma(-dg, di, 1.0), di, di);
        di = __builtin_fma(__builtin_fma(-dg, di, 1.0), di, di);
        di = (l15 < rows) ? di : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) Lt[l15 * LD + 4 * q + l4] = (4 * q + l4 < l15 && l15 < rows) ? -(lt[q] * di) : 0.0;
        wave_sync();
        double lr[16];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const v2d_f t2 = *(const v2d_f *)(Lt + l15 * LD + 2 * p);
            lr[2 * p] = t2[0];
            lr[2 * p + 1] = t2[1];
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (l15 == 4 * q + l4) ? di : 0.0;
        w3_substitute(x, lr);
    };

    for (int bi = bbeg; bi < bend; bi++) {
        const W3Block k = blocks[bi];
        const int n = k.n, U = k.U;
        const int nq = (n + 3) >> 2;                              // k-steps that carry rows of this block
        const bool two = n > 16;
        const double *Lg = tb.Xc + k.xyoff, *Yg = tb.Y + k.xyoff, *Vg = tb.vop + (long long)k.vop_off * 512;
        // ---- operands ----
        double y[2][8], lr0[4], lr1[8], v[NU][8], lam[NU];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int row = 16 * a + l15, col = 4 * q + l4;
                const bool ok = row < n && col < n;
                const double t = Yg[ok ? row + n * col : 0];
                y[a][q] = ok ? t : 0.0;
            }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = l15, col = 4 * q + l4;
            const bool ok = row < n && col <= row;
            const double t = Lg[ok ? row + n * col : 0];
            lr0[q] = ok ? t : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int row = 16 + l15, col = 4 * q + l4;
            const bool ok = two && row < n && col <= row;
            const double t = Lg[ok ? row + n * col : 0];
            lr1[q] = ok ? t : 0.0;
        }
#pragma unroll
        for (int t = 0; t < NU; t++)
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const v2d_f w = ((const v2d_f *)Vg)[(t * 4 + p) * 64 + lane];
                v[t][2 * p] = w[0];
                v[t][2 * p + 1] = w[1];
            }
#pragma unroll
        for (int t = 0; t < NU; t++) {
            const bool ok = 16 * t + l15 < U;
            const double tl = tb.lam[k.lam_off + (ok ? 16 * t + l15 : 0)];
            lam[t] = ok ? tl : 0.0;
        }
        // ---- T_Y = Y V: accumulator (a, t)[reg] = T_Y[16 a + 4 reg + l4, 16 t + l15] ----
        v4d_f ty[2][NU];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < NU; t++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
                if (a == 0 || two) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (q < nq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y[a][q], v[t][q], acc, 0, 0, 0);
                }
                ty[a][t] = acc;
            }
        // ---- the inverses of the two diagonal blocks of L (VALU, beneath the MFMAs) ----
        double x0[4], x1[4];
        diag_inverse(lr0, n < 16 ? n : 16, x0);
        {
            double lt1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) lt1[q] = lr1[4 + q];
            if (two) diag_inverse(lt1, n - 16, x1);
            else {
#pragma unroll
                for (int q = 0; q < 4; q++) x1[q] = 0.0;
            }
        }
        // ---- G_Y = V^T T_Y, lower tiles: entry (16 ti + 4 reg + l4, 16 tj + l15) ----
        v4d_f gy[NT];
#pragma unroll
        for (int ti = 0; ti < NU; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (q < nq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti][q], ty[q >> 2][tj][q & 3], acc, 0, 0, 0);
                gy[ti * (ti + 1) / 2 + tj] = acc;
            }
        // A_Y: the diagonal of G_Y (src/solver.jl:1152-1170)
#pragma unroll
        for (int t = 0; t < NU; t++) {
            const v4d_f g = gy[t * (t + 1) / 2 + t];
            double dv = g[0];
            dv = (dreg == 1) ? g[1] : dv;
            dv = (dreg == 2) ? g[2] : dv;
            dv = (dreg == 3) ? g[3] : dv;
            if (on_diag && 16 * t + l15 < U) tb.AY[tb.ay[k.ay_base + 16 * t + l15]] = dv;
        }
        // ---- Z D = (L^-1 V) diag(lambda): Z_top = A^-1 V_top; Z_bot = B^-1 (V_bot - C Z_top) ----
        v4d_f z[2][NU];
#pragma unroll
        for (int t = 0; t < NU; t++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], v[t][q], acc, 0, 0, 0);
            z[0][t] = acc;
        }
#pragma unroll
        for (int t = 0; t < NU; t++) {
            v4d_f zb = {0.0, 0.0, 0.0, 0.0};
            if (two) {
                v4d_f r = {v[t][4], v[t][5], v[t][6], v[t][7]};                  // V_bot in accumulator order = its operand order
#pragma unroll
                for (int q = 0; q < 4; q++) r = __builtin_amdgcn_mfma_f64_16x16x4f64(-lr1[q], z[0][t][q], r, 0, 0, 0);      // - C Z_top
#pragma unroll
                for (int q = 0; q < 4; q++) zb = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], r[q], zb, 0, 0, 0);
            }
            z[1][t] = zb;
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < NU; t++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) z[a][t][reg] *= lam[t];
        // ---- D G_X D = (Z D)^T (Z D) tile by tile, times G_Y, into S ----
#pragma unroll
        for (int ti = 0; ti < NU; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (q < nq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(z[q >> 2][ti][q & 3], z[q >> 2][tj][q & 3], acc, 0, 0, 0);
                const int T = ti * (ti + 1) / 2 + tj;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) sacc[T][reg] = __builtin_fma(acc[reg], gy[T][reg], sacc[T][reg]);
            }
        if (!k.last) continue;
        // ---- 1 x 1 dense blocks: S[p_u, p_v] += a_u a_v Y / X, one rank-1 MFMA step per tile (k = 0 carries the data) ----
        for (int e = 0; e < k.ndense; e++) {
            const W3Dense de = tb.dense[k.dense0 + e];
            const double dy = tb.Y[de.xyoff], dl = tb.Xc[de.xyoff];
            double a[NU], ar[NU];
            const double xx = dl * dl;
            double rx = __builtin_amdgcn_rcp(xx);
            rx = __builtin_fma(__builtin_fma(-xx, rx, 1.0), rx, rx);
            rx = __builtin_fma(__builtin_fma(-xx, rx, 1.0), rx, rx);
            const double ratio = dy * rx;
#pragma unroll
            for (int t = 0; t < NU; t++) {
                const bool ok = 16 * t + l15 < U;
                const double ta = tb.lam[de.lam_off + (ok ? 16 * t + l15 : 0)];
                a[t] = (ok && l4 == 0) ? ta : 0.0;
                ar[t] = a[t] * ratio;
            }
#pragma unroll
            for (int ti = 0; ti < NU; ti++)
#pragma unroll
                for (int tj = 0; tj <= ti; tj++) sacc[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[ti], a[tj], sacc[ti * (ti + 1) / 2 + tj], 0, 0, 0);
        }
        // ---- S_j: entries u >= v are computed; both (u, v) and (v, u) are written (symmetric!, src/tools.jl:43-57) ----
        const int P = U;
        double *Sg = tb.S + k.S_off;
        // through LDS, so that S_j (P x P contiguous) leaves in whole consecutive pieces: the mirrored half written from the registers is one 8-byte
        // store per cache line (24 store instructions of 64 lines each: 43 -> 27 us for 2048 blocks of PolyOpt 2d = 40)
#pragma unroll
        for (int ti = 0; ti < NU; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int u = 16 * ti + 4 * reg + l4, vv = 16 * tj + l15;
                    if (u >= vv && u < P) {
                        const double sv = sacc[ti * (ti + 1) / 2 + tj][reg];
                        const int pu = k.pmap_identity ? u : tb.pmap[k.pmap_off + u], pv = k.pmap_identity ? vv : tb.pmap[k.pmap_off + vv];
                        Ss[pv + pu * LDS_S] = sv;
                        Ss[pu + pv * LDS_S] = sv;
                    }
                }
        wave_sync();
        {
            const int dr = 64 / P, dc = 64 % P;                 // element m = lane + 64 i of S_j is (r, c) = (m / P, m % P): stepped, not divided
            int r = lane / P, cc = lane % P;
            for (int m = lane; m < P * P; m += 64) {
                Sg[m] = Ss[r * LDS_S + cc];
                cc += dc; r += dr;
                if (cc >= P) { cc -= P; r++; }
            }
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < NT; t++) sacc[t] = (v4d_f){0.0, 0.0, 0.0, 0.0};
    }
}

#define W4_LDS_BYTES(NU, s_rows) ((size_t)4 * (16 * 18 + 16 + (s_rows) * (16 * (NU) + 2)) * sizeof(double))

}  // namespace clrs
