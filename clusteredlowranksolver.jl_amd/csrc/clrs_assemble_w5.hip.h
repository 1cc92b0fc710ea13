// clrs_assemble_w5.hip.h -- k_cluster_assemble_w5: register-resident fp64 Schur assembly of 2 x 2 blocks of 16 x 16 sub-blocks, gfx950.
//
// The matrix-valued constraints of Nsphere_packing (examples/Nsphere_packing.jl; BASELINE config 3): a PSD block of side 32 = 2 x 16 whose
// constraint matrices are E_rs (x) v_u v_u^T -- the SAME U <= 32 sample vectors v_u (16 entries) in the sub-blocks (0,0), (1,1) and, symmetrised
// as the two terms (0,1) + (1,0), in the off-diagonal pair: 3 U constraints, 4 U terms.  The general kernel walks this as a 32 x 64 problem with
// a 64 x 64 pairing matrix through LDS (0.03 of the HBM roof); here the structure is used instead.  With Z_A = L^-1[:, 0:16] V, Z_B = L^-1[:, 16:32] V
// and T_A = Y[:, 0:16] V, T_B = Y[:, 16:32] V (src/solver.jl:1121-1143 sub-block by sub-block) the four U x U blocks of each pairing matrix are
//     AA_x = Z_A^T Z_A, AB_x = Z_A^T Z_B, BA_x = Z_B^T Z_A, BB_x = Z_B^T Z_B        AA_y = V^T T_A[0:16], AB_y = V^T T_B[0:16], BA_y = V^T T_A[16:32], BB_y = V^T T_B[16:32]
// (bpX[s,r] = V^T X^-1_sr V, bpY[s,r] = V^T Y_sr V), and the sum over the term pairs of src/solver.jl:1176-1212 collapses, pair-block by pair-block
// of S (pairs in the order 00, 01, 11; every product elementwise in [u, v]; lambda_i[u] lambda_j[v] in front), to
//     (00,00) AA_x AA_y      (01,00) AA_x BA_y + BA_x AA_y      (11,00) BA_x BA_y
//     (01,01) AA_x BB_y + BB_x AA_y + AB_x BA_y + BA_x AB_y      (11,01) BA_x BB_y + BB_x BA_y      (11,11) BB_x BB_y .
// Everything is v_mfma_f64_16x16x4_f64 on operands that never leave the registers, as in clrs_assemble_w3 / _w4.hip.h (same lane maps, same DPP
// substitution for the two 16 x 16 diagonal blocks of L; Z_A's lower half is B^-1 (- C Z_A,top), Z_B's upper half is zero): 208 MFMAs per block,
// S_j accumulates IN MEMORY across the cluster's blocks (one triangle: the first block stores, the others add -- the 21 tiles of a 6 x 6 tile grid as register accumulators beside the operands exceed the 512 registers of a lane), the cluster's last block writes the mirrored entries as well.  A_Y (src/solver.jl:1152-1170) is the diagonals
// of AA_y, AB_y, BA_y, BB_y.  One wave per run of clusters, nothing synchronises.
#pragma once
#include <type_traits>
#include "clrs_assemble_w4.hip.h"

namespace clrs {

struct W5Tables {
    const double *Xc, *Y;      // iterates (xy layout): Cholesky factors of the X blocks, Y blocks
    double *S, *AY;            // outputs
    const double *vop;         // per block 512 doubles, w3's operand order: vop[((t*2 + p)*64 + lane)*2 + e] = V[4 (2 p + e) + (lane >> 4), 16 t + (lane & 15)], zero padded
    const double *lam;         // [lam_off + i U + u]: lambda of the constraint (pair i, vector u), i = 0 (0,0), 1 (0,1)+(1,0), 2 (1,1)
    const int *ay;             // [ay_base + (2 r + s) 32 + u]: position of the term (r, s) of vector u in the A_Y output
};

__global__ __launch_bounds__(256, 1) void k_cluster_assemble_w5(const int *__restrict__ cluster_blk0, const W3Block *__restrict__ blocks, const W5Tables tb,
                                                                int nclusters, int nblocks) {
    constexpr int LD = 18;
    constexpr int PER_WAVE = 16 * LD + 16 + 96;                 // L rows | diagonal of L | lambda of the block's constraints
    __shared__ __attribute__((aligned(16))) double lds_all[4 * PER_WAVE];
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    double *Lt = lds_all + wave * PER_WAVE, *Ld = Lt + 16 * LD, *Lm = Ld + 16;
    const int c0 = (int)((long long)gw * nclusters / nw), c1 = (int)((long long)(gw + 1) * nclusters / nw);
    if (c0 >= c1) return;
    const int bbeg = cluster_blk0[c0], bend = (c1 < nclusters) ? cluster_blk0[c1] : nblocks;
    const int dreg = (l15 - l4) >> 2;
    const bool on_diag = l15 >= l4 && ((l15 - l4) & 3) == 0;

    // S tiles of the cluster: tile (2 i + ti, 2 j + tj) of the 6 x 6 grid, pairs i >= j; diagonal pair-blocks keep ti >= tj only
    // index: D(i)[ti >= tj] -> 3 each (i = 0, 1, 2), O(i > j)[ti][tj] -> 4 each: (1,0), (2,0), (2,1)
    auto diag_inverse = [&](const double (&lt)[4], double (&x)[4]) {
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
#pragma unroll
        for (int q = 0; q < 4; q++) Lt[l15 * LD + 4 * q + l4] = (4 * q + l4 < l15) ? -(lt[q] * di) : 0.0;
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
        const int U = k.U;
        constexpr int n = 32;
        const double *Lg = tb.Xc + k.xyoff, *Yg = tb.Y + k.xyoff, *Vg = tb.vop + (long long)k.vop_off * 512;
        // ---- operands ----
        double y[2][8], lr0[4], lr1[8], v[2][4];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int q = 0; q < 8; q++) y[a][q] = Yg[(16 * a + l15) + n * (4 * q + l4)];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = l15, col = 4 * q + l4;
            const double t = Lg[row + n * col];
            lr0[q] = col <= row ? t : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int row = 16 + l15, col = 4 * q + l4;
            const double t = Lg[row + n * col];
            lr1[q] = col <= row ? t : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {                               // 16 bytes per lane: k-steps 2p and 2p + 1 of tile t, c = 2t + p
            const v2d_f w = ((const v2d_f *)Vg)[c * 64 + lane];
            v[c >> 1][2 * (c & 1)] = w[0];
            v[c >> 1][2 * (c & 1) + 1] = w[1];
        }
        // lambda of (pair i, vector u) into LDS (read back per tile position: by row u = 16 t + 4 reg + l4 and by column u = 16 t + l15)
        for (int e = lane; e < 96; e += 64) {
            const int i = e >> 5, u = e & 31;
            const double tl = tb.lam[k.lam_off + i * U + (u < U ? u : 0)];
            Lm[e] = u < U ? tl : 0.0;
        }
        // ---- T_A = Y[:, 0:16] V, T_B = Y[:, 16:32] V: accumulator (a, t)[reg] = T[16 a + 4 reg + l4, 16 t + l15] ----
        v4d_f TA[2][2], TB[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                v4d_f accA = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    accA = __builtin_amdgcn_mfma_f64_16x16x4f64(y[a][q], v[t][q], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f64_16x16x4f64(y[a][4 + q], v[t][q], accB, 0, 0, 0);
                }
                TA[a][t] = accA;
                TB[a][t] = accB;
            }
        // ---- the inverses of the two diagonal blocks of L ----
        double x0[4], x1[4];
        diag_inverse(lr0, x0);
        {
            double lt1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) lt1[q] = lr1[4 + q];
            diag_inverse(lt1, x1);
        }
        // ---- Z_A = L^-1[:, 0:16] V: top = A^-1 V, bottom = B^-1 (- C top);  Z_B = L^-1[:, 16:32] V: top = 0, bottom = B^-1 V ----
        v4d_f ZAt[2], ZAb[2], ZBb[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0}, r = {0.0, 0.0, 0.0, 0.0}, zb = {0.0, 0.0, 0.0, 0.0}, zbb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], v[t][q], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) r = __builtin_amdgcn_mfma_f64_16x16x4f64(-lr1[q], acc[q], r, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) zb = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], r[q], zb, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) zbb = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], v[t][q], zbb, 0, 0, 0);
            ZAt[t] = acc; ZAb[t] = zb; ZBb[t] = zbb;
        }
        double ayv[2][4];
        auto tiles = [&](auto first_c, auto last_c) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        // ---- per [u, v] tile position (ti, tj): the four blocks of both pairing matrices, A_Y from the diagonal tiles, S ----
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj < 2; tj++) {
                auto gram4 = [&](const v4d_f (&L)[2], const v4d_f (&R)[2], v4d_f acc) {        // sum over four k-steps of L[ti]^T R[tj]
#pragma unroll
                    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L[ti][q], R[tj][q], acc, 0, 0, 0);
                    return acc;
                };
                auto vgram4 = [&](const v4d_f &R) {                                             // V[:, tile ti]^T R
                    v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[ti][q], R[q], acc, 0, 0, 0);
                    return acc;
                };
                const v4d_f zero4 = {0.0, 0.0, 0.0, 0.0};
                const v4d_f AAy = vgram4(TA[0][tj]), ABy = vgram4(TB[0][tj]), BAy = vgram4(TA[1][tj]), BBy = vgram4(TB[1][tj]);
                const v4d_f AAx = gram4(ZAb, ZAb, gram4(ZAt, ZAt, zero4)), ABx = gram4(ZAb, ZBb, zero4), BAx = gram4(ZBb, ZAb, zero4), BBx = gram4(ZBb, ZBb, zero4);
                if (ti == tj) {                                     // A_Y of the terms (r, s) of vector 16 t + l15: the diagonals (src/solver.jl:1152-1170), stored behind the loop
#pragma unroll
                    for (int rs = 0; rs < 4; rs++) {
                        const v4d_f &g = rs == 0 ? AAy : rs == 1 ? ABy : rs == 2 ? BAy : BBy;
                        double d0 = g[0];
                        d0 = (dreg == 1) ? g[1] : d0;
                        d0 = (dreg == 2) ? g[2] : d0;
                        d0 = (dreg == 3) ? g[3] : d0;
                        ayv[ti][rs] = d0;
                    }
                }
                double lamr[3][4], lamc[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    lamc[i] = Lm[32 * i + 16 * tj + l15];
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) lamr[i][reg] = Lm[32 * i + 16 * ti + 4 * reg + l4];
                }
                const int T4 = 2 * ti + tj;                         // tile of an off-diagonal pair-block
                const int T3 = ti + tj;                             // tile (ti >= tj) of a diagonal pair-block: (0,0) -> 0, (1,0) -> 1, (1,1) -> 2
                {
                    double *Sg2 = tb.S + k.S_off;
                    const int P2 = 3 * U;
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const double c00 = lamr[0][reg] * lamc[0], c10 = lamr[1][reg] * lamc[0], c20 = lamr[2][reg] * lamc[0];
                        const double c11 = lamr[1][reg] * lamc[1], c21 = lamr[2][reg] * lamc[1], c22 = lamr[2][reg] * lamc[2];
                        const int u = 16 * ti + 4 * reg + l4, vv = 16 * tj + l15;
                        const bool okk = u < U && vv < U;
                        const double s10 = c10 * __builtin_fma(AAx[reg], BAy[reg], BAx[reg] * AAy[reg]);
                        const double s20 = c20 * (BAx[reg] * BAy[reg]);
                        const double s21 = c21 * __builtin_fma(BAx[reg], BBy[reg], BBx[reg] * BAy[reg]);
                        const double s00 = c00 * (AAx[reg] * AAy[reg]);
                        const double s11 = c11 * __builtin_fma(AAx[reg], BBy[reg], __builtin_fma(BBx[reg], AAy[reg], __builtin_fma(ABx[reg], BAy[reg], BAx[reg] * ABy[reg])));
                        const double s22 = c22 * (BBx[reg] * BBy[reg]);
                        if (okk) {
                            double *row = Sg2 + (long long)u * P2 + vv;
                            // S_j accumulates in memory, lower triangle (row (i, u), column (j, v), i U + u >= j U + v): the cluster's first block stores, the others add
                            // the cluster's last block also writes the mirrored entry (symmetric!, src/tools.jl:43-57): one 8-byte store per cache line, but nothing waits for it
                            double *col = Sg2 + (long long)vv * P2 + u;                  // entry ((0, v), (0, u)); + j U P2 + i U for ((j, v), (i, u))
                            auto upd = [&](long long off, long long moff, double val) {
                                const double nv = FIRST ? val : row[off] + val;
                                row[off] = nv;
                                if (LAST) col[moff] = nv;
                            };
                            upd((long long)U * P2, U, s10); upd(2LL * U * P2, 2 * U, s20); upd(2LL * U * P2 + U, (long long)U * P2 + 2 * U, s21);
                            if (u >= vv) { upd(0, 0, s00); upd((long long)U * P2 + U, (long long)U * P2 + U, s11); upd(2LL * U * P2 + 2 * U, 2LL * U * P2 + 2 * U, s22); }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                  // one tile position at a time: the eight pairing tiles of all four in flight at once do not fit the registers
            }
        };
        if (k.ndense) {                                 // (W3Block::ndense: 1 on the first block of a cluster)
            if (k.last) tiles(std::true_type{}, std::true_type{});
            else tiles(std::true_type{}, std::false_type{});
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the earlier block's stores have reached L2 before this block reads them back
            if (k.last) tiles(std::false_type{}, std::true_type{});
            else tiles(std::false_type{}, std::false_type{});
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (on_diag && 16 * t + l15 < U) {                      // (every vector has its four terms: the host takes no other block)
                const int *ap = tb.ay + k.ay_base + 16 * t + l15;
                tb.AY[ap[0]] = ayv[t][0]; tb.AY[ap[32]] = ayv[t][1]; tb.AY[ap[64]] = ayv[t][2]; tb.AY[ap[96]] = ayv[t][3];
            }
    }
}

}  // namespace clrs
