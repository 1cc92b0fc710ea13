// clrs_assemble_w3.hip.h -- k_cluster_assemble_w3: register-resident Schur assembly, one wave per run of clusters, gfx950.
//
// Same eligibility as k_cluster_assemble_w2 (clusters whose low-rank blocks are "simple": one sub-block, n <= 16, rank-1
// symmetric terms, one term per constraint, the same constraint order in every block, U = P <= 32; dense blocks 1 x 1) --
// the Cohn-Elkies / Delsarte / univariate-SOS shapes -- but the matrices never pass through LDS and nothing synchronises.
//
// The observation that makes this possible: for v_mfma_f64_16x16x4_f64 the A operand (lane (l15, l4) holds A[i = l15][k = l4])
// and the B operand (lane holds B[k = l4][j = l15]) have the SAME lane <-> (free index, contraction index) map, and an
// accumulator (lane holds D[4 reg + l4][l15], reg = 0..3) read as "k-step reg" is again an operand with contraction index
// 4 reg + l4 and free index l15.  So the whole chain of src/solver.jl:1121-1212
//     T_Y = Y V          (A = Y rows,    B = V)          ->  accumulators = operands with contraction index = row of T_Y
//     G_Y = V^T T_Y      (A = V,         B = T_Y acc)
//     Z   = L^-1 V       (A = L^-1 rows, B = V)          ->  accumulators = operands with contraction index = row of Z
//     G_X = Z^T Z        (A = Z acc,     B = Z acc)      ( = V^T X^-1 V, no explicit inverse of X: src/solver.jl:1117)
//     S  += (lambda lambda^T) o G_X o G_Y                (same lanes hold the same entry of G_X and G_Y)
// runs from registers to registers.  V is static data: clrs_ctx_create stores it in MFMA-operand order
// (vop[(t*4+q)*64 + lane] = V[4q + l4, 16t + l15], zero padded), so every load of it is one fully coalesced 512-byte
// wave access; Y (column-major n x n, n = 16) is in operand order as it stands.  L^-1 comes from the DPP substitution on the
// identity (clrs_wave.hip.h) with the structural zeros skipped, and runs on the VALU underneath the T_Y / G_Y MFMAs; the
// row of L_X each lane needs for it is the only thing staged through LDS (2.3 KB per wave, written and read by the same
// wave: in order, no barrier).  lambda is folded into Z (Z D), so the accumulation into S is one FMA per entry.
// 1 x 1 dense blocks (Y / X scalars) are rank-1 MFMA steps into the same accumulators when their cluster ends.
//
// Work distribution: each wave owns a CONTIGUOUS range of clusters, i.e. a contiguous range of the block table, and walks
// it with the loads of the next block (and the descriptor of the one after) in flight during the arithmetic of the
// current one -- across cluster boundaries too, so only a wave's very first block waits for memory.
#pragma once
#include "clrs_fused.hip.h"

namespace clrs {

struct W3Block {               // one simple low-rank block (n <= 16, U <= 32) of the flat block sequence
    long long xyoff;           // offset of the block in the X/Y layout
    long long S_off;           // last block of a cluster: offset of S_j in the S layout
    long long dxyoff;          // last block, ndense >= 1: X/Y offset of the cluster's first 1 x 1 dense block
    int n, U;                  // block side; U = P of the cluster
    int last;                  // 1: S_j is stored after this block
    int vop_off;               // W3Tables::vop: the vectors in MFMA-operand order, in units of 512 doubles
    int lam_off;               // W3Tables::lam: [U] lambda of vector u
    int ay_base;               // the term of vector u is at ay_base + u of the A_Y output (clusters whose terms are not consecutive in
                               // A_Y stay on k_cluster_assemble_w2: a table lookup here would be a vector load on a rarely taken path,
                               // and the compiler then drains vmcnt on the common path too before reusing that load's register)
    int pmap_off;              // W3Tables::pmap: [U] constraint of vector u (a permutation of 0..P-1)
    int ndense, dense0;        // the cluster's 1 x 1 dense blocks: W3Tables::dense[dense0 .. dense0 + ndense)
    int dlam_off;              // W3Tables::lam: [U] matrix entry of the first dense block for the constraint of vector u
    int pmap_identity;         // 1: pmap[u] == u (the usual case): the S store needs no table
    int pad[3];
};
struct W3Dense {
    long long xyoff;
    int lam_off, pad;
};
struct W3Tables {
    const double *Xc, *Y;      // iterates (xy layout): Cholesky factors of the X blocks, Y blocks
    double *S, *AY;            // outputs: S layout, A_Y per term
    const double *vop;         // vectors in MFMA-operand order, two k-steps per 16-byte lane element:
                               // vop[((t*2+p)*64 + lane)*2 + e] = V[4(2p+e) + (lane >> 4), 16t + (lane & 15)], zero padded
    const double *lam;         // [lam_off + u]: lambda of the term of vector u; [dlam_off + u]: entries of the 1 x 1 dense blocks
    const int *pmap;
    const W3Dense *dense;
};

struct W3Pre {                 // what one PSD block needs from memory, as it arrives in registers
    double y[4];               // Y[l15, 4q + l4]                       (A operand of T_Y = Y V)
    double v[8];               // V[4q + l4, 16t + l15] at [t*4 + q]    (B operand of T_Y and Z, A operand of G_Y)
    double lt[4];              // L[l15, 4q + l4]
    double lam[2];             // lambda of vector 16t + l15
    double da[2], dy, dl;      // the cluster's first 1 x 1 dense block (last block of a cluster only; harmless reads otherwise):
                               // its entry for the constraint of vector 16t + l15, Y, chol X
};

// Forward substitution on four right-hand-side groups inside a wave, one instruction per elimination:
//     x_q[i] += x_q[K] * m[K]      for the lanes i of every 16-lane row, x_q[K] read from lane K of the row by DPP
// (v_fmac_f64_dpp with row_newbcast -- the only DPP control the 64-bit ALU has).  m[K] = -L[i,K] / L[i,i] for K < i and 0
// otherwise (row-scaled, so the system is unit lower triangular and no step multiplies by a reciprocal), the right-hand
// side is D^-1: the result is L^-1.  Lane (l15, l4) carries columns 4q + l4 of the right-hand side in x_q: column c is zero
// above row c, so chain q starts at step 4q: 36 eliminations instead of 60.  The chains are independent of each other; they
// are interleaved so that an instruction reads a register written at least three instructions earlier (a DPP read needs two
// wait states after the VALU write); the tail of the longest chain is padded with s_nop.
__device__ __forceinline__ void w3_substitute(double (&x)[4], const double (&m)[16]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %14 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %15 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %16 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %3, %16 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %14 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %15 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %3, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %16 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %18 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %3, %18 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %0, %14 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %18 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f64_dpp %0, %0, %15 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %16 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %18 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]),
          "v"(m[12]), "v"(m[13]), "v"(m[14]));
}

typedef double v2d_f __attribute__((ext_vector_type(2)));

// Diagnostic builds (-DCLRS_W3_STAMPS): wave 0 of workgroup 0 accumulates the shader cycles (s_memtime) it spends in each phase
// of a block; read back with clrs_debug_w3_stamps.  The product build contains none of this.
#ifdef CLRS_W3_STAMPS
__device__ unsigned long long g_w3_stamps[16];
#define W3_STAMP(i)                                                                  \
    do {                                                                             \
        if (gw == 0) {                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();              \
            st_acc[i] += t_ - st_prev;                                               \
            st_prev = t_;                                                            \
        }                                                                            \
    } while (0)
#else
#define W3_STAMP(i) do { } while (0)
#endif

template <bool FULL>           // FULL: every block has n == 16 and U == P == 32 exactly
__global__ __launch_bounds__(256, 2) void k_cluster_assemble_w3(const int *__restrict__ cluster_blk0, const W3Block *__restrict__ blocks,
                                                                const W3Tables tb, int nclusters, int nblocks) {
    constexpr int LD = 18;                                       // doubles per row of the staged L: conflict-free ds_read_b128
    constexpr int LDS_S = 34;                                    // leading dimension of the staged S_j (even: 16-byte aligned pairs)
    constexpr int PER_WAVE = 16 * LD + 16 + 32 + 32 * LDS_S;     // L rows | diagonal of L | A_Y of the block | S_j
    __shared__ __attribute__((aligned(16))) double lds_all[4 * PER_WAVE];
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    double *Lt = lds_all + wave * PER_WAVE, *Ld = Lt + 16 * LD, *Ay = Ld + 16, *Ss = Ay + 32;
    // contiguous cluster range of this wave
    const int c0 = (int)((long long)gw * nclusters / nw), c1 = (int)((long long)(gw + 1) * nclusters / nw);
    if (c0 >= c1) return;
    const int bbeg = cluster_blk0[c0], bend = (c1 < nclusters) ? cluster_blk0[c1] : nblocks;

    auto issue = [&](const W3Block &kb, W3Pre &r) {
        const double *Lg = tb.Xc + kb.xyoff, *Yg = tb.Y + kb.xyoff, *Vg = tb.vop + (long long)kb.vop_off * 512;
        const int n = FULL ? 16 : kb.n, U = FULL ? 32 : kb.U;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool ok = FULL || (l15 < n && 4 * q + l4 < n);
            const int idx = ok ? l15 + n * (4 * q + l4) : 0;
            const double ty = Yg[idx], tl = Lg[idx];
            r.y[q] = ok ? ty : 0.0;
            r.lt[q] = ok ? tl : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {                               // 16 bytes per lane: k-steps 2p and 2p + 1 of tile t, c = 2t + p
            const v2d_f w = ((const v2d_f *)Vg)[c * 64 + lane];
            r.v[2 * c] = w[0];
            r.v[2 * c + 1] = w[1];
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const bool ok = FULL || 16 * t + l15 < U;
            const double tl = tb.lam[kb.lam_off + (ok ? 16 * t + l15 : 0)];
            r.lam[t] = ok ? tl : 0.0;
        }
        // first 1 x 1 dense block of the cluster, fetched with the cluster's last block.  These loads are issued for EVERY block (from
        // addresses that are valid anyway when there is nothing to fetch): the number of vector-memory operations per block must not
        // depend on the block, or the s_waitcnt vmcnt(N) the compiler places before the first use of this block's data would have to
        // assume the smaller count and so also wait for the loads of the block after it.
        {
            const bool has = kb.last && kb.ndense > 0;
            const long long dxy = has ? kb.dxyoff : kb.xyoff;
            const int dlo = has ? kb.dlam_off : kb.lam_off;
            // (vector loads on purpose: a scalar load would come back through lgkmcnt, which every LDS wait of the block drains)
            r.dy = tb.Y[dxy];
            r.dl = tb.Xc[dxy];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const bool ok = FULL || 16 * t + l15 < U;
                const double ta = tb.lam[dlo + (ok ? 16 * t + l15 : 0)];
                r.da[t] = ok ? ta : 0.0;
            }
        }
    };

    v4d_f sacc[3];
#pragma unroll
    for (int t = 0; t < 3; t++) sacc[t] = (v4d_f){0.0, 0.0, 0.0, 0.0};
#ifdef CLRS_W3_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
    const int dreg = (l15 - l4) >> 2;                             // lanes with l15 = 4 dreg + l4 hold a diagonal entry of a diagonal tile
    const bool on_diag = l15 >= l4 && ((l15 - l4) & 3) == 0;

    // one block: consume `cur` (its loads were issued one block ago), start the loads of block bi + 1 into `nxt`
    auto process = [&](int bi, const W3Block &k, const W3Block &kn, W3Pre &cur, W3Pre &nxt) {
        W3_STAMP(0);
        issue(kn, nxt);          // unconditional (the last block of the range re-reads itself): see the note on the dense loads in issue()
        // ---- the diagonal of L_X: the lane that holds L[l15,l15] in lt[] hands it to its row through LDS ----
        {
            double dsel = cur.lt[0];
            dsel = (dreg == 1) ? cur.lt[1] : dsel;
            dsel = (dreg == 2) ? cur.lt[2] : dsel;
            dsel = (dreg == 3) ? cur.lt[3] : dsel;
            if (on_diag) Ld[l15] = dsel;
        }
        wave_sync();             // LDS hand-offs between lanes of this wave: in order in hardware, this only pins the compiler's order
        const double dg = Ld[l15];
        // ---- m = -D^-1 strict_lower(L_X) through LDS: lane (l15, *) reads row l15 ----
        double di = __builtin_amdgcn_rcp(dg);                       // 1 / L[l15,l15]: v_rcp_f64 + two Newton steps
        di = __builtin_fma(__builtin_fma(-dg, di, 1.0), di, di);
        di = __builtin_fma(__builtin_fma(-dg, di, 1.0), di, di);
        if (!FULL) di = (l15 < k.n) ? di : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) Lt[l15 * LD + 4 * q + l4] = (4 * q + l4 < l15) ? -(cur.lt[q] * di) : 0.0;
        wave_sync();
        double lr[16];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const v2d_f t2 = *(const v2d_f *)(Lt + l15 * LD + 2 * p);
            lr[2 * p] = t2[0];
            lr[2 * p + 1] = t2[1];
        }
        wave_sync();
        W3_STAMP(1);
        // ---- T_Y = Y V: two tiles, accumulator reg = row 4 reg + l4 of T_Y, column 16 t + l15 ----
        v4d_f ty[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.y[q], cur.v[t * 4 + q], acc, 0, 0, 0);
            ty[t] = acc;
        }
#ifdef CLRS_W3_STAMPS
        asm volatile("" :: "v"(ty[0][0]), "v"(ty[1][3]));
#endif
        W3_STAMP(2);
        // ---- L^-1 = (D^-1 L)^-1 D^-1 by substitution: lane (l15, l4) gets W[l15, 4q + l4], q = 0..3 ----
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (l15 == 4 * q + l4) ? di : 0.0;
        w3_substitute(x, lr);
        W3_STAMP(3);
        // ---- G_Y = V^T T_Y, lower tiles: entry (16 ti + 4 reg + l4, 16 tj + l15) ----
        v4d_f gy[3];
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.v[ti * 4 + q], ty[tj][q], acc, 0, 0, 0);
                gy[ti * (ti + 1) / 2 + tj] = acc;
            }
#ifdef CLRS_W3_STAMPS
        asm volatile("" :: "v"(gy[0][0]), "v"(gy[2][3]));
#endif
        W3_STAMP(4);
        // A_Y: the diagonal of G_Y (src/solver.jl:1152-1170): one lane-select, one store per diagonal tile
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const v4d_f g = gy[t * (t + 1) / 2 + t];
            double dv = g[0];
            dv = (dreg == 1) ? g[1] : dv;
            dv = (dreg == 2) ? g[2] : dv;
            dv = (dreg == 3) ? g[3] : dv;
            if (on_diag && (FULL || 16 * t + l15 < k.U)) tb.AY[k.ay_base + 16 * t + l15] = dv;
        }
        W3_STAMP(5);
        // ---- Z D = (L^-1 V) diag(lambda) ----
        v4d_f z[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[q], cur.v[t * 4 + q], acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) acc[reg] *= cur.lam[t];
            z[t] = acc;
        }
#ifdef CLRS_W3_STAMPS
        asm volatile("" :: "v"(z[0][0]), "v"(z[1][3]));
#endif
        W3_STAMP(6);
        // ---- D G_X D = (Z D)^T (Z D) tile by tile, times G_Y, into S ----
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
            for (int tj = 0; tj <= ti; tj++) {
                v4d_f acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(z[ti][q], z[tj][q], acc, 0, 0, 0);
                const int T = ti * (ti + 1) / 2 + tj;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) sacc[T][reg] = __builtin_fma(acc[reg], gy[T][reg], sacc[T][reg]);
            }
#ifdef CLRS_W3_STAMPS
        asm volatile("" :: "v"(sacc[0][0]), "v"(sacc[2][3]));
#endif
        W3_STAMP(7);
        if (k.last) {
            // ---- 1 x 1 dense blocks: S[p_u, p_v] += a_u a_v Y / X, one rank-1 MFMA step per tile (k = 0 carries the data) ----
            double da[2] = {cur.da[0], cur.da[1]}, dy = cur.dy, dl = cur.dl;
            for (int e = 0; e < k.ndense; e++) {
                if (e > 0) {                                        // further dense blocks of the cluster: loaded here (rare)
                    const W3Dense de = tb.dense[k.dense0 + e];
                    dy = tb.Y[de.xyoff];
                    dl = tb.Xc[de.xyoff];
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const bool ok = FULL || 16 * t + l15 < k.U;
                        const double ta = tb.lam[de.lam_off + (ok ? 16 * t + l15 : 0)];
                        da[t] = ok ? ta : 0.0;
                    }
                }
                const double xx = dl * dl;                          // Y / X with X = L^2: reciprocal + two Newton steps (an IEEE division is ~30 VALU)
                double rx = __builtin_amdgcn_rcp(xx);
                rx = __builtin_fma(__builtin_fma(-xx, rx, 1.0), rx, rx);
                rx = __builtin_fma(__builtin_fma(-xx, rx, 1.0), rx, rx);
                const double ratio = dy * rx;
                double a[2], ar[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    a[t] = (l4 == 0) ? da[t] : 0.0;
                    ar[t] = a[t] * ratio;
                }
                sacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[0], a[0], sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[1], a[0], sacc[1], 0, 0, 0);
                sacc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[1], a[1], sacc[2], 0, 0, 0);
            }
            // the constraint of vector u: u itself unless the cluster carries a permutation (then read here: rare, waits for memory)
            int pm_v[2], pm_u[8];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                pm_v[t] = 16 * t + l15;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) pm_u[t * 4 + reg] = 16 * t + 4 * reg + l4;
            }
            if (!k.pmap_identity) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    pm_v[t] = tb.pmap[k.pmap_off + ((FULL || pm_v[t] < k.U) ? pm_v[t] : 0)];
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) pm_u[t * 4 + reg] = tb.pmap[k.pmap_off + ((FULL || pm_u[t * 4 + reg] < k.U) ? pm_u[t * 4 + reg] : 0)];
                }
            }
            // ---- S_j: entries u >= v are computed; both (u, v) and (v, u) are written (symmetric!, src/tools.jl:43-57) ----
            const int P = FULL ? 32 : k.U;
            double *Sg = tb.S + k.S_off;
            // through LDS, so that S_j leaves as whole consecutive 512-byte / 1-KB pieces (S_j is P x P contiguous)
            if (k.pmap_identity) {
                // (u, v) -> v + u LDS_S = [l15 + l4 LDS_S] + constant, mirror [l4 + l15 LDS_S] + constant: two lane offsets, immediates
                double *s_uv = Ss + l15 + l4 * LDS_S, *s_vu = Ss + l4 + l15 * LDS_S;
#pragma unroll
                for (int ti = 0; ti < 2; ti++)
#pragma unroll
                    for (int tj = 0; tj <= ti; tj++)
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int u = 16 * ti + 4 * reg + l4, v = 16 * tj + l15;
                            if (u >= v && (FULL || u < P)) {
                                const double sv = sacc[ti * (ti + 1) / 2 + tj][reg];
                                s_uv[16 * tj + (16 * ti + 4 * reg) * LDS_S] = sv;
                                s_vu[16 * ti + 4 * reg + 16 * tj * LDS_S] = sv;
                            }
                        }
            } else {
#pragma unroll
                for (int ti = 0; ti < 2; ti++)
#pragma unroll
                    for (int tj = 0; tj <= ti; tj++)
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int u = 16 * ti + 4 * reg + l4, v = 16 * tj + l15;
                            if (u >= v && (FULL || u < P)) {
                                const double sv = sacc[ti * (ti + 1) / 2 + tj][reg];
                                const int pu = pm_u[ti * 4 + reg], pv = pm_v[tj];
                                Ss[pv + pu * LDS_S] = sv;
                                Ss[pu + pv * LDS_S] = sv;
                            }
                        }
            }
            wave_sync();
            if (FULL) {
#pragma unroll
                for (int i = 0; i < 8; i++) {                       // 16 bytes per lane, 1 KB per store
                    const int m = 128 * i + 2 * lane;
                    const v2d_f w = *(const v2d_f *)(Ss + (m >> 5) * LDS_S + (m & 31));
                    __builtin_nontemporal_store(w, (v2d_f *)(Sg + m));
                }
            } else {
                for (int m = lane; m < P * P; m += 64) Sg[m] = Ss[(m / P) * LDS_S + (m % P)];
            }
            wave_sync();
#pragma unroll
            for (int t = 0; t < 3; t++) sacc[t] = (v4d_f){0.0, 0.0, 0.0, 0.0};
            W3_STAMP(8);
        }
    };

    W3Pre pa, pb;
    // descriptors: k0 = current, k1 = next (its loads are issued at the start of the current block), k2 = the one after
    // (its scalar loads are in flight during the current block)
    W3Block k0 = blocks[bbeg], k1 = blocks[bbeg + 1 < bend ? bbeg + 1 : bbeg], k2;
    issue(k0, pa);
    for (int bi = bbeg;;) {
        k2 = blocks[bi + 2 < bend ? bi + 2 : bi];
        process(bi, k0, k1, pa, pb);
        if (++bi >= bend) break;
        k0 = k1; k1 = k2;
        k2 = blocks[bi + 2 < bend ? bi + 2 : bi];
        process(bi, k0, k1, pb, pa);
        if (++bi >= bend) break;
        k0 = k1; k1 = k2;
    }
#ifdef CLRS_W3_STAMPS
    if (gw == 0 && lane == 0) {
        for (int i = 0; i < 15; i++) g_w3_stamps[i] = st_acc[i];
        g_w3_stamps[15] = (unsigned long long)(bend - bbeg);
    }
#endif
}

}  // namespace clrs
