// clrs_mw_exact.hip.h -- the pairing matrices of a low-rank block through EXACT slice products on the fp64 matrix cores.
//
// The multi-word contractions of the Schur assembly (the four GEMMs of src/solver.jl:1121-1147: T = Y V, Z = X^-1/2 V, GY = V^T T,
// GX = Z^T Z) cost 345 fp64 lane instructions per 5-limb multiply-add in k_mw_zt / k_mw_gram: 30 flops of limb products, the rest
// are the error-free transformations that keep the ADDITIONS exact (profiles/r02/m_pmc_mw_assembly_sq_counters.csv).  Here the
// operands are cut into slices whose products and k-sums are exact in an fp64 accumulator by construction, so the additions need no
// transformation at all and the inner loop is v_mfma_f64_16x16x4:
//
//   x_ik = 2^e_i  sum_s d_s(i,k) 2^-(s+1)B ,   d_s integers, |d_s| <= 2^(B-1) + 1,   e_i one exponent per ROW of the left operand
//   y_kj = 2^f_j  sum_t d'_t(k,j) 2^-(t+1)B ,                                       f_j one exponent per COLUMN of the right operand
//   sum_k x_ik y_kj = 2^(e_i + f_j)  sum_o 2^-(o+2)B  [ sum_{s+t=o} sum_k d_s(i,k) d'_t(k,j) ]        (orders o >= S dropped)
//
// With B = 23 the bracket is an integer below (o+1) k 2^(2B-2) <= 2^52 for k <= 32 and o < 16: every MFMA accumulation is exact, in any
// order.  S = ceil((52 K + 16) / B) slices (12 at K = 5: 276 bits below the row's largest entry).  Dropping the orders o >= S and cutting
// every entry at 2^-SB relative to its row / column maximum are the only roundings: a normwise error of k 2^-(SB-9) relative to
// max|row| max|column|, which is what the backward-error analysis of a GEMM asks for (measured on the trajectory iterates of
// cohnelkies(8,15), mu from 1e20 to 2e-16: 2^-273 of max|GY|, against 2^-262 for the 5-limb expansions).  An entry far below its row's
// maximum keeps fewer bits of its own -- unlike the expansions, which round every product relative to itself.
//
// The digits are stored as fp32 (exact: 23 bits and a sign), half the LDS of doubles, converted by one v_cvt_f64_f32 per operand.
// One workgroup of four waves per PSD block; every slice array has the layout [slice][k][col] (k = contraction index, col = the free
// index, contiguous): the MFMA A operand (lane (l15, l4) holds A[i = l15][k = l4]) and B operand (B[k = l4][j = l15]) read it alike.
// The two conversions -- K limbs -> digits, order sums -> K limbs -- are in clrs_mw_slices.h (host + device).
//   phase A  slice Y and Xi = chol(X)^-1 by rows (dynamic); the slices of V by columns are static data (context creation)
//   phase B  T = Y V and Z = Xi V: one 16 x 16 output tile per wave and turn; recombine the order sums to K limbs, slice by columns
//   phase D  GY = V^T T, GX = Z^T Z, lower tiles only, recombined to K limbs and written mirrored like k_mw_gram does
// The per-term pairings A_Y and the accumulation into S_j stay with k_mw_saccum.
#ifndef CLRS_MW_EXACT_HIP_H
#define CLRS_MW_EXACT_HIP_H

#include "clrs_mw_kernels.hip.h"

#include "clrs_mw_slices.h"       // MWS_BETA, mws_slices, mws_exponent, mws_slice, mws_recombine_orders (host + device; tests/mw_host checks them against mpmath)

#define MWS_NT 256

namespace mwk {

typedef double v4d_mw __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;

// register `reg` of the S order accumulators of a tile -> K limbs (clrs_mw_slices.h)
template <int K, int S>
__device__ __forceinline__ mwa::mw<K> mws_recombine(const v4d_mw (&acc)[S], int reg, int escale) {
    double a[S];
#pragma unroll
    for (int o = 0; o < S; o++) a[o] = acc[o][reg];
    return mws_recombine_orders<K, S>(a, escale);
}

// one 16 x 16 output tile: acc[o] += sum over the pairs s + t = o, s < SA, t < SB, and the k-steps, of A_s^T-style products
//   A operand: As[s][k][ca + l15], B operand: Bs[t][k][cb + l15]; strides in floats: slice strides sa / sb, row (k) strides ra / rb; ca, cb: the
//   tile's first column as this lane's k-rows hold it (mws_tilecol)
// SA, SB are compile-time: every digit of a k-step is in registers before its first MFMA and the MFMA sequence has no branch in it.
template <int S, int SA, int SB>
__device__ __forceinline__ void mws_tile(v4d_mw (&acc)[S], const lds_f *As, int sa, int ra, int ca, const lds_f *Bs, int sb, int rb, int cb, int ksteps, int l15, int l4) {
#pragma unroll
    for (int o = 0; o < S; o++) acc[o] = (v4d_mw){0.0, 0.0, 0.0, 0.0};
    const lds_f *ap = As + l4 * ra + ca + l15, *bp = Bs + l4 * rb + cb + l15;
    for (int ks = 0; ks < ksteps; ks++) {
        float af[SA], bf[SB];
#pragma unroll
        for (int s = 0; s < SA; s++) af[s] = ap[s * sa + ks * 4 * ra];
#pragma unroll
        for (int t = 0; t < SB; t++) bf[t] = bp[t * sb + ks * 4 * rb];
        double bv[SB];
#pragma unroll
        for (int t = 0; t < SB; t++) bv[t] = (double)bf[t];
#pragma unroll
        for (int s = 0; s < SA; s++) {
            const double av = (double)af[s];
#pragma unroll
            for (int t = 0; t < SB; t++)
                if (s + t < S) acc[s + t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], acc[s + t], 0, 0, 0);
        }
    }
}
// the static operand (V) has SV <= S slices that are not all zero: instantiations for SV rounded up to S/2, 3S/4, S (mws_sv_class)
template <int S, bool V_IS_A>
__device__ __forceinline__ void mws_tile_v(v4d_mw (&acc)[S], int SV, const lds_f *As, int sa, int ra, int ca, const lds_f *Bs, int sb, int rb, int cb, int ksteps, int l15, int l4) {
    constexpr int S1 = (S + 1) / 2, S2 = (3 * S + 3) / 4;
    if (V_IS_A) {
        if (SV <= S1) mws_tile<S, S1, S>(acc, As, sa, ra, ca, Bs, sb, rb, cb, ksteps, l15, l4);
        else if (SV <= S2) mws_tile<S, S2, S>(acc, As, sa, ra, ca, Bs, sb, rb, cb, ksteps, l15, l4);
        else mws_tile<S, S, S>(acc, As, sa, ra, ca, Bs, sb, rb, cb, ksteps, l15, l4);
    } else {
        if (SV <= S1) mws_tile<S, S, S1>(acc, As, sa, ra, ca, Bs, sb, rb, cb, ksteps, l15, l4);
        else if (SV <= S2) mws_tile<S, S, S2>(acc, As, sa, ra, ca, Bs, sb, rb, cb, ksteps, l15, l4);
        else mws_tile<S, S, S>(acc, As, sa, ra, ca, Bs, sb, rb, cb, ksteps, l15, l4);
    }
}

}  // namespace mwk

// static slices of V of every eligible block: digits [S][np][mws_rowstride(Up)] in the column order of mws_col (np = n rounded up to 4,
// Up = U rounded up to 16), column exponents [Up]
struct MwsDev {
    const float *Vs;         // digits, per block at vs_off
    const int *Vexp;         // per block at ve_off
    const long long *vs_off; // [NB] (-1: the block is not eligible)
    const int *ve_off;       // [NB]
    const int *sv;           // [NB] slices of V that are not all zero
    unsigned long long *stamps;   // diagnostic: wall_clock64 at the phase boundaries of wave 0 of workgroup 0 (or null)
};

// Layout of a slice array [slice][k][col] with c16 columns (a multiple of 16).  A wave reads four consecutive k-rows of one 16-column tile at
// once; the four rows must fall into different quarters of the 64 banks.  An odd number of tiles per row does that by itself; two tiles per row
// (32 columns, the named shapes) by exchanging the two tiles in the rows with bit 1 of k set; other even counts by one tile of padding.
// (Half the LDS of the padded form of the first version: two workgroups per compute unit instead of one.)
__host__ __device__ __forceinline__ int mws_rowstride(int c16) { const int m = c16 / 16; return ((m & 1) || m == 2) ? c16 : c16 + 16; }
__host__ __device__ __forceinline__ int mws_col(int k, int col, int c16) { return c16 == 32 ? col ^ (((k >> 1) & 1) << 4) : col; }
__device__ __forceinline__ int mws_tilecol(int tile, int l4, int c16) { return c16 == 32 ? (tile ^ (l4 >> 1)) * 16 : tile * 16; }     // k = 4 ks + l4
// slices of V kept in LDS: the class of mws_tile_v its count of non-zero slices falls into
__host__ __device__ __forceinline__ int mws_sv_class(int S, int SV) { const int S1 = (S + 1) / 2, S2 = (3 * S + 3) / 4; return SV <= S1 ? S1 : SV <= S2 ? S2 : S; }
// LDS of one workgroup in floats: V slices, the Y / Xi slices (later: the T slices), the Z slices, + exponents
static inline size_t mws_lds_bytes(int S, int SVc, int n, int U) {
    const int np = (n + 3) & ~3, n16 = (n + 15) & ~15, U16 = (U + 15) & ~15;
    const size_t vsl = (size_t)SVc * np * mws_rowstride(U16), yx = (size_t)2 * S * np * mws_rowstride(n16), tz = (size_t)S * np * mws_rowstride(U16);
    return (vsl + (yx > tz ? yx : tz) + tz) * sizeof(float) + (size_t)(2 * n16 + 3 * U16 + 8 * U16) * sizeof(int);
}

// TURNS = 1: every block of the launch has at most four T / Z tiles (n <= 16 with U <= 32, the named shapes), one per wave; 2: up to eight
template <int K, int DK, int TURNS>
__global__ __launch_bounds__(MWS_NT, 2) void k_mws_pair(const MwDev q, const MwsDev w, const double *__restrict__ Y) {
    using namespace mwk;
    constexpr int S = mws_slices(K);
    // workgroups beyond the low-rank blocks (the launch adds them when every dense block is 1 x 1): the dense blocks' tables, beside the pairing
    // matrices instead of in a launch of their own behind them
    if ((int)blockIdx.x >= q.nlr) { mw_dense_1x1<K, DK>(q, q.blk[q.dn_list[blockIdx.x - q.nlr]], Y, threadIdx.x); return; }
    const int b = q.lr_list[blockIdx.x];
    if (w.vs_off[b] < 0) return;
    const MwBlk &k = q.blk[b];
    const int n = k.n, U = k.U, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int np = (n + 3) & ~3, n16 = (n + 15) & ~15, U16 = (U + 15) & ~15, ksteps = np / 4;
    const int rV = mws_rowstride(U16), rY = mws_rowstride(n16), sV = np * rV, sY = np * rY;
    const int SV = w.sv[b], SVc = mws_sv_class(S, SV);
    extern __shared__ __attribute__((aligned(16))) float mws_lds[];
#ifdef CLRS_MW_STAMPS            // diagnostic builds only (scripts/mw_pmc.sh, scripts/dense_stamps.py): the product carries no run-time probe
    const bool stamp = w.stamps && blockIdx.x == 0 && tid == 0;
    int nst = 0;
#define MWS_STAMP() do { if (stamp) w.stamps[nst++] = wall_clock64(); } while (0)
#else
#define MWS_STAMP() do { } while (0)
#endif
    MWS_STAMP();
    lds_f *Vsl = (lds_f *)mws_lds;                              // [SVc][np][rV]
    const size_t yx = (size_t)2 * S * sY, tz = (size_t)S * sV;
    lds_f *Ysl = Vsl + (size_t)SVc * sV, *Xsl = Ysl + (size_t)S * sY;      // [S][np][rY] each; the T slices overlay them later
    lds_f *Tsl = Ysl, *Zsl = Ysl + (yx > tz ? yx : tz);         // [S][np][rV] each
    int *eY = (int *)(Zsl + tz), *eX = eY + n16, *fV = eX + n16, *fT = fV + U16, *fZ = fT + U16, *part = fZ + U16;      // part: [2][4][U16] column maxima per row tile
    // ---- phase A: V slices (static), row exponents of Y and Xi, their slices ----
    {
        typedef float v4f_mw __attribute__((ext_vector_type(4)));
        const v4f_mw *gv = (const v4f_mw *)(w.Vs + w.vs_off[b]);
        v4f_mw __attribute__((address_space(3))) *lv = (v4f_mw __attribute__((address_space(3))) *)Vsl;
        const int nq = SVc * sV / 4, nv = SV * sV / 4;          // sV is a multiple of 4 (rows of U16 + 16 floats)
        for (int e0 = 0; e0 < nq; e0 += 4 * MWS_NT) {          // four 16-byte loads in flight per thread and pass
            v4f_mw v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int e = e0 + tid + u * MWS_NT; v[u] = e < nv ? gv[e] : (v4f_mw){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int e = e0 + tid + u * MWS_NT; if (e < nq) lv[e] = v[u]; }
        }
        for (int u = tid; u < U16; u += MWS_NT) fV[u] = w.Vexp[w.ve_off[b] + u];
        if (n != n16 || np != n) {                             // padding rows / columns of the Y and Xi slices are zero digits
            for (int e = tid; e < 2 * S * sY; e += MWS_NT) Ysl[e] = 0.0f;
        }
        for (int i = tid; i < 2 * n16; i += MWS_NT) eY[i] = 0;                    // (eY and eX are adjacent)
    }
    __syncthreads();
    MWS_STAMP();
    const double *Yg = Y + k.xyoff, *Xig = q.Xi + k.xyoff;
    // row exponents: every thread takes the heads of its entries (coalesced loads) to an integer maximum per row in LDS
    for (int e = tid; e < 2 * n * n; e += MWS_NT) {
        const int which = e / (n * n), ee = e % (n * n), i = ee % n, c = ee / n;
        if (which == 1 && c > i) continue;
        const double h = (which == 0 ? Yg : Xig)[i + (long)c * n];
        if (h != 0.0) atomicMax(&(which == 0 ? eY : eX)[i], mws_exponent(h) + 4096);      // biased: the initial 0 is below every real exponent
    }
    __syncthreads();
    for (int i = tid; i < 2 * n16; i += MWS_NT) eY[i] = eY[i] == 0 ? 0 : eY[i] - 4096;
    __syncthreads();
    for (int e = tid; e < 2 * n * n; e += MWS_NT) {
        const int which = e / (n * n), ee = e % (n * n), i = ee % n, c = ee / n;        // entry (row i, column c): A operand [k = c][col = i]
        lds_f *dst = (which == 0 ? Ysl : Xsl) + c * rY + mws_col(c, i, n16);
        if (which == 1 && c > i) {                           // Xi is lower triangular
#pragma unroll
            for (int sl = 0; sl < S; sl++) dst[sl * sY] = 0.0f;
            continue;
        }
        const mw<K> x = ldx<K>(which == 0 ? Yg : Xig, q.xylen, i + (long)c * n);
        mws_slice<K, S>(x, (which == 0 ? eY : eX)[i], [&](int sl, float d) { dst[sl * sY] = d; });
    }
    __syncthreads();
    MWS_STAMP();
    // ---- phase B: T = Y V, Z = Xi V; tasks = (which, row tile, column tile), one per wave and turn ----
    const int ntn = n16 / 16, ntu = U16 / 16, ntask = 2 * ntn * ntu;
    v4d_mw acc[S];
    // the results stay in registers across the barrier that frees the Y / Xi slices: at most two turns are supported (ntask <= 8)
    mw<K> res[TURNS][4];
    int tsk[TURNS];
#pragma unroll
    for (int turn = 0; turn < TURNS; turn++) tsk[turn] = -1;
#pragma unroll
    for (int turn = 0; turn < TURNS; turn++) {
        const int task = wave + 4 * turn;
        if (task >= ntask) break;
        tsk[turn] = task;
        const int which = task / (ntn * ntu), ti = (task / ntu) % ntn, tj = task % ntu;
        mws_tile_v<S, false>(acc, SV, which == 0 ? Ysl : Xsl, sY, rY, mws_tilecol(ti, l4, n16), Vsl, sV, rV, mws_tilecol(tj, l4, U16), ksteps, l15, l4);
        MWS_STAMP();
        int cmax = -100000;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int i = ti * 16 + 4 * reg + l4, j = tj * 16 + l15;
            res[turn][reg] = mws_recombine<K, S>(acc, reg, (which == 0 ? eY : eX)[min(i, n16 - 1)] + fV[j]);
            if (i < n && j < U && res[turn][reg].l[0] != 0.0) cmax = max(cmax, mws_exponent(res[turn][reg].l[0]));
        }
        cmax = max(cmax, __shfl_xor(cmax, 16, 64));
        cmax = max(cmax, __shfl_xor(cmax, 32, 64));
        if (l4 == 0) part[(which * 4 + ti) * U16 + tj * 16 + l15] = cmax;
    }
    MWS_STAMP();
    __syncthreads();                                            // every wave is done with the Y / Xi slices; the partial column maxima are written
    for (int u = tid; u < 2 * U16; u += MWS_NT) {
        const int which = u / U16, j = u % U16;
        int m = -100000;
        for (int ti = 0; ti < ntn; ti++) m = max(m, part[(which * 4 + ti) * U16 + j]);
        (which == 0 ? fT : fZ)[j] = m == -100000 ? 0 : m;
    }
    if (n != np || U != U16) {                                  // padding rows / columns of the T and Z slices
        for (int e = tid; e < (int)(2 * tz); e += MWS_NT) (e < (int)tz ? Tsl[e] : Zsl[e - tz]) = 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int turn = 0; turn < TURNS; turn++) {
        const int task = tsk[turn];
        if (task < 0) break;
        const int which = task / (ntn * ntu), ti = (task / ntu) % ntn, tj = task % ntu;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int i = ti * 16 + 4 * reg + l4, j = tj * 16 + l15;        // entry (row i, column j) of T / Z: operand [k = i][col = j]
            if (i >= n || j >= U) continue;
            lds_f *dst = (which == 0 ? Tsl : Zsl) + i * rV + mws_col(i, j, U16);
            mws_slice<K, S>(res[turn][reg], (which == 0 ? fT : fZ)[j], [&](int s, float d) { dst[s * sV] = d; });
        }
    }
    __syncthreads();
    MWS_STAMP();
    // ---- phase D: GX = Z^T Z, GY = V^T T, lower tiles; GX tiles first (they are the longer tasks: all S slices on both sides) ----
    // (six tiles on four waves at the named shapes: waves 0, 1 get two.  Workgroups of odd index deal the tiles in the opposite wave order, so
    // that two workgroups that share a compute unit do not leave the same SIMDs idle.)
    const int ntri = ntu * (ntu + 1) / 2;
    for (int task = (blockIdx.x & 1) ? 3 - wave : wave; task < 2 * ntri; task += 4) {
        const int which = task / ntri;                      // 0: GX, 1: GY
        int ti, tj;
        tri_index(task % ntri, ti, tj);                     // ti >= tj
        if (which == 0) mws_tile<S, S, S>(acc, Zsl, sV, rV, mws_tilecol(ti, l4, U16), Zsl, sV, rV, mws_tilecol(tj, l4, U16), ksteps, l15, l4);
        else mws_tile_v<S, true>(acc, SV, Vsl, sV, rV, mws_tilecol(ti, l4, U16), Tsl, sV, rV, mws_tilecol(tj, l4, U16), ksteps, l15, l4);
        MWS_STAMP();
        double *G = (which == 0 ? q.GX : q.GY) + k.g_off;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int a = ti * 16 + 4 * reg + l4, c = tj * 16 + l15;        // entry (a, c), a >= c kept
            if (a >= U || c >= U || c > a) continue;
            const mw<K> v = mws_recombine<K, S>(acc, reg, which == 0 ? fZ[a] + fZ[c] : fV[a] + fT[c]);
            stx<K>(G, q.glen, a + (long)c * U, v);
            stx<K>(G, q.glen, c + (long)a * U, v);
        }
        MWS_STAMP();
    }
#undef MWS_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Pairing matrices of blocks of ANY size (many unique vectors: the three-point bound has blocks of side 54 with 408 of them) through the same
// exact slice products: GX = Z^T Z and GY = V^T T are U x U Gram products of n-row operands, 83 000 entries of 54-term K-limb dot products for
// such a block in k_mw_gram.  Here k_mw_zt leaves Z and T as K-limb matrices as before, k_mwx_slice cuts them into digits by columns ([slice][k][col]
// in global memory, one exponent per column; the digits of V are static), and k_mwx_gram gives every wave one 16 x 16 tile of the lower triangle:
// operands straight from memory (L2), 32 rows of k per exact accumulation, the order sums pushed into one K-limb accumulator per entry across the
// chunks of k.  Blocks with sub-blocks (m > 1) and the small shapes of k_mws_pair are not taken.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct MwxDev {
    const float *Vd;         // static digits of V, [S][np][U16] per block at d_off (np = n rounded up to 4, U16 = U rounded up to 16; padding zero)
    float *Zd, *Td;          // digits of Z and T of the current assembly, same shape and offsets
    const int *eV;           // column exponents, [U16] per block at e_off
    int *eZ, *eT;
    const long long *d_off;  // [NB], -1: not taken
    const int *e_off;        // [NB]
    const int *sv;           // [NB] slices of V that are not all zero
};

// digits and column exponents of Z (blockIdx.y = 0) and T (1) of one block: sixteen columns per workgroup
template <int K>
__global__ __launch_bounds__(MWS_NT) void k_mwx_slice(const MwDev q, const MwxDev w) {
    using namespace mwk;
    constexpr int S = mws_slices(K);
    const int b = q.lr_list[blockIdx.z];
    if (w.d_off[b] < 0) return;
    const MwBlk &k = q.blk[b];
    const int n = k.n, U = k.U, tid = threadIdx.x, np = (n + 3) & ~3, U16 = (U + 15) & ~15;
    const int c0 = blockIdx.x * 16;
    if (c0 >= U16) return;
    const int nc = min(16, U - c0);                         // live columns of this chunk (<= 0: padding only)
    const double *src = (blockIdx.y == 0 ? q.Z : q.Tm) + k.z_off;
    float *D = (blockIdx.y == 0 ? w.Zd : w.Td) + w.d_off[b];
    int *E = (blockIdx.y == 0 ? w.eZ : w.eT) + w.e_off[b];
    __shared__ int emax[16];
    if (tid < 16) emax[tid] = 0;
    __syncthreads();
    for (int e = tid; e < n * nc; e += MWS_NT) {            // heads, coalesced along the columns' rows
        const int i = e % n, cl = e / n;
        const double h = src[i + (long)(c0 + cl) * n];
        if (h != 0.0) atomicMax(&emax[cl], mws_exponent(h) + 4096);
    }
    __syncthreads();
    if (tid < 16) {
        const int ev = emax[tid] == 0 ? 0 : emax[tid] - 4096;
        emax[tid] = ev;
        E[c0 + tid] = ev;
    }
    __syncthreads();
    const long sD = (long)np * U16;
    for (int e = tid; e < n * nc; e += MWS_NT) {            // digits: the column runs fastest, so that the sixteen floats of a (slice, row) are one store
        const int cl = e % nc, i = e / nc;
        const mw<K> x = ldx<K>(src, q.zlen, i + (long)(c0 + cl) * n);
        float *dst = D + (long)i * U16 + c0 + cl;
        mws_slice<K, S>(x, emax[cl], [&](int sl, float d) { dst[sl * sD] = d; });
    }
}

namespace mwk {
// acc[o] += the pairs s + t = o of the k-steps ks0 .. ks0 + nks - 1, operands in global memory: A[s][k][ca + l15], B[t][k][cb + l15]
template <int S, int SA, int SB>
__device__ __forceinline__ void mwx_tile(v4d_mw (&acc)[S], const float *__restrict__ Ag, long sa, int ra, const float *__restrict__ Bg, long sb, int rb, int ks0, int nks, int l15,
                                         int l4) {
    const float *ap = Ag + (long)(4 * ks0 + l4) * ra + l15, *bp = Bg + (long)(4 * ks0 + l4) * rb + l15;
    float af[SA], bf[SB];
#pragma unroll
    for (int s = 0; s < SA; s++) af[s] = ap[s * sa];
#pragma unroll
    for (int t = 0; t < SB; t++) bf[t] = bp[t * sb];
    for (int ks = 0; ks < nks; ks++) {
        double av[SA], bv[SB];
#pragma unroll
        for (int s = 0; s < SA; s++) av[s] = (double)af[s];
#pragma unroll
        for (int t = 0; t < SB; t++) bv[t] = (double)bf[t];
        if (ks + 1 < nks) {                                 // the next k-step's digits are on their way while this one's products run
            ap += 4 * ra; bp += 4 * rb;
#pragma unroll
            for (int s = 0; s < SA; s++) af[s] = ap[s * sa];
#pragma unroll
            for (int t = 0; t < SB; t++) bf[t] = bp[t * sb];
        }
#pragma unroll
        for (int s = 0; s < SA; s++)
#pragma unroll
            for (int t = 0; t < SB; t++)
                if (s + t < S) acc[s + t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[t], acc[s + t], 0, 0, 0);
    }
}

}  // namespace mwk

// one 16 x 16 tile (ta >= tb) of GX (blockIdx.y = 0) or GY (1) per wave
template <int K>
__global__ __launch_bounds__(MWS_NT, 2) void k_mwx_gram(const MwDev q, const MwxDev w) {
    using namespace mwk;
    constexpr int S = mws_slices(K), S1 = (S + 1) / 2, S2 = (3 * S + 3) / 4;
    const int b = q.lr_list[blockIdx.z];
    if (w.d_off[b] < 0) return;
    const MwBlk &k = q.blk[b];
    const int n = k.n, U = k.U, np = (n + 3) & ~3, U16 = (U + 15) & ~15, nt = U16 / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int task = blockIdx.x * (MWS_NT / 64) + wave;
    if (task >= nt * (nt + 1) / 2) return;                  // (uniform over the wave; no barrier below)
    int ta, tb;
    tri_index(task, ta, tb);                                // ta >= tb
    const bool gy = blockIdx.y == 1;
    const long sD = (long)np * U16, off = w.d_off[b];
    const float *Lg = (gy ? w.Vd : (const float *)w.Zd) + off + ta * 16, *Rg = (gy ? (const float *)w.Td : (const float *)w.Zd) + off + tb * 16;
    const int *eL = (gy ? w.eV : (const int *)w.eZ) + w.e_off[b], *eR = (gy ? (const int *)w.eT : (const int *)w.eZ) + w.e_off[b];
    const int SV = w.sv[b], ksteps = np / 4;
    double bins[4][mws_bins(S)];                           // (clrs_mw_slices.h: the chunks' order sums add up exactly in the bins)
#pragma unroll
    for (int reg = 0; reg < 4; reg++) mws_bins_zero<S>(bins[reg]);
    v4d_mw acc[S];
    for (int ks0 = 0; ks0 < ksteps; ks0 += 8) {             // 32 rows of k: the order sums stay below 2^53
        const int nks = min(8, ksteps - ks0);
#pragma unroll
        for (int o = 0; o < S; o++) acc[o] = (v4d_mw){0.0, 0.0, 0.0, 0.0};
        if (!gy) mwx_tile<S, S, S>(acc, Lg, sD, U16, Rg, sD, U16, ks0, nks, l15, l4);
        else if (SV <= S1) mwx_tile<S, S1, S>(acc, Lg, sD, U16, Rg, sD, U16, ks0, nks, l15, l4);
        else if (SV <= S2) mwx_tile<S, S2, S>(acc, Lg, sD, U16, Rg, sD, U16, ks0, nks, l15, l4);
        else mwx_tile<S, S, S>(acc, Lg, sD, U16, Rg, sD, U16, ks0, nks, l15, l4);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            double a[S];
#pragma unroll
            for (int o = 0; o < S; o++) a[o] = acc[o][reg];
            mws_bins_add<S>(bins[reg], a);
        }
    }
    double *G = (gy ? q.GY : q.GX) + k.g_off;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int a = ta * 16 + 4 * reg + l4, c = tb * 16 + l15;        // entry (a, c), a >= c kept and mirrored
        if (a >= U || c >= U || c > a) continue;
        const mw<K> v = mws_bins_result<K, S>(bins[reg], eL[a] + eR[c]);
        stx<K>(G, q.glen, a + (long)c * U, v);
        stx<K>(G, q.glen, c + (long)a * U, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// T_e = X^-1 (A_e Y) = Xi^T (Xi (A_e Y)) of the dense blocks with 16 < n <= 32 (SDPA x64: 512 matrices of 32 x 32) through the same exact slice
// products: three chained 32 x 32 x 32 products per matrix, 2 x 2 output tiles = the four waves of one workgroup, every operand as digits in LDS
// ([slice][k][32 columns], tile-exchange layout).  Two regions of S 32 x 32 digit arrays: X holds the left operand of the product at hand (the
// static digits of A_e, then Xi^T by its rows' exponents, then Xi by its columns'), R the right one (Y by columns, then the K-limb result of the
// previous product cut again by columns).  The expansion form of the same (k_mw_dense_tp) runs at twice the fp64 pipe's bound; this one does a
// fifth of its work.  Limb counts up to 6 (the two regions need 2 S 4 KB of LDS).
// ---------------------------------------------------------------------------------------------------------------------------------------
struct MwdDev {
    const float *Ad;         // static digits of the dense matrices: [S1][32][32] per matrix (S1 = (S + 1) / 2 slices: 106-bit data), matrix e of block b at a_off[b] + e S1 1024
    const int *eA;           // column exponents, [32] per matrix at e_off[b] + 32 e
    const long long *a_off;  // [NB], -1: not taken
    const int *e_off;        // [NB]
    const int *tasks;        // [2 ntasks]: (index into dn_list, matrix) of every matrix of every block taken -- a dense list: with one workgroup per compute unit
                             // (98 KB of LDS) a grid over (block, most matrices of a block) ran four rounds of workgroups, the last two mostly empty
    unsigned long long *stamps;   // diagnostic (clrs_mw_debug_exact_stamps): wall_clock64 at the phase boundaries of wave 0 of the first workgroup, or null
};
template <int K, int DK>
__global__ __launch_bounds__(MWS_NT) void k_mwx_dense(const MwDev q, const MwdDev w, const double *__restrict__ Y) {
    using namespace mwk;
    constexpr int S = mws_slices(K), S1 = (S + 1) / 2, sN = 32 * 32;
    const int b = q.dn_list[w.tasks[2 * blockIdx.x]], e = w.tasks[2 * blockIdx.x + 1];
    const MwBlk &k = q.blk[b];
    const int n = k.n, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int np = (n + 3) & ~3, ksteps = np / 4;
    extern __shared__ __attribute__((aligned(16))) float mwd_lds[];
    lds_f *RX = (lds_f *)mwd_lds, *RR = RX + (size_t)S * sN;
    int *eL = (int *)(RR + (size_t)S * sN), *eR = eL + 32, *part = eR + 32;      // part: [2][32] column maxima per row tile
    const double *Yg = Y + k.xyoff, *Xig = q.Xi + k.xyoff;
    const long nn = (long)n * n;
#ifdef CLRS_MW_STAMPS
    const bool stamp = w.stamps && blockIdx.x == 0 && tid == 0;
    int nst = 0;
#define MWD_STAMP() do { if (stamp) w.stamps[nst++] = wall_clock64(); } while (0)
#else
#define MWD_STAMP() do { } while (0)
#endif
    MWD_STAMP();
    // ---- operands of product 1: A_e (static digits, by columns; A_e is symmetric) and Y by columns ----
    {
        typedef float v4f_mw __attribute__((ext_vector_type(4)));
        const v4f_mw *ga = (const v4f_mw *)(w.Ad + w.a_off[b] + (long)e * S1 * sN);
        v4f_mw __attribute__((address_space(3))) *lx = (v4f_mw __attribute__((address_space(3))) *)RX;
        for (int o = tid; o < S1 * sN / 4; o += MWS_NT) lx[o] = ga[o];
        if (n != 32) for (int o = tid; o < S * sN; o += MWS_NT) RR[o] = 0.0f;      // padding rows / columns are zero digits
        if (tid < 32) { eL[tid] = w.eA[w.e_off[b] + 32 * e + tid]; eR[tid] = 0; }
    }
    __syncthreads();
    for (int o = tid; o < n * n; o += MWS_NT) {
        const double h = Yg[o];
        if (h != 0.0) atomicMax(&eR[o / n], mws_exponent(h) + 4096);
    }
    __syncthreads();
    if (tid < 32) eR[tid] = eR[tid] == 0 ? 0 : eR[tid] - 4096;
    __syncthreads();
    for (int o = tid; o < n * n; o += MWS_NT) {
        const int kk = o % n, c = o / n;                  // entry (row kk, column c): operand [k = kk][col = c]
        lds_f *dst = RR + kk * 32 + mws_col(kk, c, 32);
        mws_slice<K, S>(ldx<K>(Yg, q.xylen, kk + (long)c * n), eR[c], [&](int sl, float d) { dst[sl * sN] = d; });
    }
    __syncthreads();
    const int ti = wave & 1, tj = wave >> 1;              // output tile: rows 16 ti .., columns 16 tj ..
    v4d_mw acc[S];
    mw<K> res[4];
    // the result of a product as the right operand of the next one: column maxima over both row tiles, then digits by columns into RR
    auto publish = [&](const int *ex) {
        int cmax = -100000;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int i = ti * 16 + 4 * reg + l4, c = tj * 16 + l15;
            res[reg] = mws_recombine<K, S>(acc, reg, ex[i] + eR[c]);
            if (i < n && c < n && res[reg].l[0] != 0.0) cmax = max(cmax, mws_exponent(res[reg].l[0]));
        }
        cmax = max(cmax, __shfl_xor(cmax, 16, 64));
        cmax = max(cmax, __shfl_xor(cmax, 32, 64));
        if (l4 == 0) part[ti * 32 + tj * 16 + l15] = cmax;
        __syncthreads();                                   // every wave is done with both regions; the partial maxima are written
        if (tid < 32) { const int m2 = max(part[tid], part[32 + tid]); eR[tid] = m2 == -100000 ? 0 : m2; }
        __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int i = ti * 16 + 4 * reg + l4, c = tj * 16 + l15;
            if (i >= n || c >= n) continue;
            lds_f *dst = RR + i * 32 + mws_col(i, c, 32);
            mws_slice<K, S>(res[reg], eR[c], [&](int sl, float d) { dst[sl * sN] = d; });
        }
    };
    // digits of Xi into RX as the left operand [k][col = i]: transposed = 1: element Xi[i, k] (product Xi M), exponents by rows of Xi;
    // transposed = 0: element Xi[k, i] (product Xi^T M), exponents by columns.  Xi is lower triangular.
    auto load_xi = [&](bool transposed) {
        if (tid < 32) eL[tid] = 0;
        if (n != 32) for (int o = tid; o < S * sN; o += MWS_NT) RX[o] = 0.0f;
        __syncthreads();
        for (int o = tid; o < n * n; o += MWS_NT) {
            const int r = o % n, c = o / n;               // Xi[r, c], r >= c
            if (c > r) continue;
            const double h = Xig[o];
            if (h != 0.0) atomicMax(&eL[transposed ? r : c], mws_exponent(h) + 4096);
        }
        __syncthreads();
        if (tid < 32) eL[tid] = eL[tid] == 0 ? 0 : eL[tid] - 4096;
        __syncthreads();
        for (int o = tid; o < n * n; o += MWS_NT) {
            const int r = o % n, c = o / n;
            const int kk = transposed ? c : r, col = transposed ? r : c;
            lds_f *dst = RX + kk * 32 + mws_col(kk, col, 32);
            if (c > r) {
#pragma unroll
                for (int sl = 0; sl < S; sl++) dst[sl * sN] = 0.0f;
                continue;
            }
            mws_slice<K, S>(ldx<K>(Xig, q.xylen, o), eL[col], [&](int sl, float d) { dst[sl * sN] = d; });
        }
    };
    MWD_STAMP();
    // ---- product 1: M1 = A_e Y ----
    mws_tile<S, S1, S>(acc, RX, sN, 32, mws_tilecol(ti, l4, 32), RR, sN, 32, mws_tilecol(tj, l4, 32), ksteps, l15, l4);
    MWD_STAMP();
    publish(eL);
    MWD_STAMP();
    load_xi(true);
    __syncthreads();
    MWD_STAMP();
    // ---- product 2: M2 = Xi M1 (k <= i: the first row tile needs the first four k-steps only) ----
    mws_tile<S, S, S>(acc, RX, sN, 32, mws_tilecol(ti, l4, 32), RR, sN, 32, mws_tilecol(tj, l4, 32), ti == 0 ? min(ksteps, 4) : ksteps, l15, l4);
    MWD_STAMP();
    publish(eL);
    MWD_STAMP();
    load_xi(false);
    __syncthreads();
    MWD_STAMP();
    // ---- product 3: T_e = Xi^T M2 (k >= i: the second row tile starts at k = 16) ----
    {
        const int ks0 = ti == 0 ? 0 : 4;
        if (ks0 < ksteps) mws_tile<S, S, S>(acc, RX + ks0 * 4 * 32, sN, 32, mws_tilecol(ti, l4, 32), RR + ks0 * 4 * 32, sN, 32, mws_tilecol(tj, l4, 32), ksteps - ks0, l15, l4);
        else {
#pragma unroll
            for (int o = 0; o < S; o++) acc[o] = (v4d_mw){0.0, 0.0, 0.0, 0.0};
        }
    }
    MWD_STAMP();
    double *W = q.W + k.w_off;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int i = ti * 16 + 4 * reg + l4, c = tj * 16 + l15;
        if (i >= n || c >= n) continue;
        stx<K>(W, q.wlen, (long)e * nn + i + (long)c * n, mws_recombine<K, S>(acc, reg, eL[i] + eR[c]));
    }
    MWD_STAMP();
#undef MWD_STAMP
}

#endif
