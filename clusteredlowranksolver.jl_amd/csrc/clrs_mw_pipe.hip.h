// clrs_mw_pipe.hip.h -- Cholesky AND inverse factor of one small matrix (n <= 32: a cluster's S_j, or Q) as a PIPELINE of workgroups, gfx950.
//
// The elimination of wg_potrf (clrs_mw_kernels.hip.h: fraction-free, [M | I], one barrier per pivot) is a chain of n dependent pivot steps, and in
// one workgroup a step costs 1.0 us + 0.65 us per wave-round of entries: the 32 x 32 matrices of the named problem start with nine wave-rounds on
// four SIMDs (3.3 us per pivot) and average 2.5 us (k_mw_factor 93 us, k_mw_potrf_q 80 us: DESIGN.md section 5.5).  The trailing update cannot be
// split over compute units without exchanging pivot columns -- but the exchange only ever runs ONE WAY when the matrix is split by COLUMN BLOCKS:
//
//   stage g (one workgroup, 256 entry threads with one entry each IN REGISTERS, and four loader waves) owns columns [8 g, 8 g + 8) of the lower triangle.  Pivot column k is
//   final once pivots 0 .. k-1 have been applied to it; its owner publishes it the moment that is so.  A stage first applies the pivot columns of
//   the stages before it, as they arrive (each is one step of at most 256 entry updates: ONE wave per SIMD), then runs its own eight pivots.  Nothing
//   ever flows back, so a stage never waits for a later one: the chain is n steps of the cheapest kind (1.4-1.6 us) plus one hand-off per stage.
//   The columns of W = [I] (the inverse factor) need every pivot column and nothing else: four more workgroups consume the same stream, each with
//   a quarter of W's columns, again one entry per thread.
//
// Hand-off: a pivot column is <= 32 K-limb numbers = 1.3 KB at K = 5.  It travels as 8-byte {32 data bits, 32-bit tag} granules written with
// agent-scope relaxed (sc1, write-through) stores and read with sc1 loads by ONE loader wave per consumer, which retries a granule until its tag
// (launch epoch, pivot) matches -- MI355X_MICROARCH.md, "handoff-1to1": no flag, no fence, no ordering between granules is needed because every granule
// carries its own validity; 0.8-1.0 us per hop, hidden behind the consumer's current step (the loader fetches column k + 1 while step k runs).
// A consumer that reads a non-positive pivot stops exactly like its producer did (approx_cholesky!'s failure test, src/tools.jl:92-95); a loader
// that never sees its tag gives up after a bounded number of polls and reports the failure instead of hanging.
//
// Measured (cohnelkies(8,15), 5 limbs, scripts/pipe_stamps.py, profiles/r04): a stage's own steps 1.45 us, steps that apply an incoming column 1.9 us, a
// hop 4 us (not the 1 us of an idle hand-off: the column is asked for before it exists and found by polling); k_mw_factor 93 -> 85 us, k_mw_potrf_q
// 80 -> 83 us (its launch also carries the first products of the next solve): the default (clrs_mw_options.pipeline = 1) pipelined the clusters only.
// Round 5: that hop was the consumers' own look-ahead (MWP_AHEAD below): with the columns asked for in the step that needs them k_mw_factor_pipe is
// 53 us at 4 limbs and the pipeline wins for Q as well (the default from two stages on).
//
// The tail (scripts/pipe_stamps.py at the end of round 4): a consumer that runs less than MWP_NL steps behind its producer sends for every column before it
// exists and finds it at its turn only -- one sc1 round trip (2-3.5 us) on its step instead of the step's arithmetic.  The workgroups of W catch up with the
// last stage, and two of the four take 3-3.5 us for each of the last four columns: the kernel ends at 84-86 us where the other two are done at 78.  Tried
// against it, none kept: one workgroup per compute unit (86 KB of LDS asked for: no change -- it is not sharing); at a miss the rows at once before the
// sentinel (no change); every lane polling its own granules for the last four columns (no change: the loads themselves take that long); every loader asking
// again once per step for what came back stale (cures the tail of the workgroups it hits, but all cadences go from 1.9 to 2.0 us: 0.526 against 0.522 ms
// per iteration, Nsphere_packing N = 3 2.54 against 2.51); the same in the last eight steps of the W workgroups only (they end at 77-81 us instead of
// 84-86, and the iteration does not move: 0.521-0.522 ms, 2.50 against 2.51).
//
// The arithmetic per entry and pivot is wg_potrf's, in the same order: the factor, its reciprocal diagonal and the inverse are bit for bit those of
// the one-workgroup kernels (tests/test_mw_parity.py::test_pipelined_factorisation_is_bit_identical).
#ifndef CLRS_MW_PIPE_HIP_H
#define CLRS_MW_PIPE_HIP_H

#define MWP_N 32             // largest matrix side
#define MWP_W 8              // columns per stage
#define MWP_WW 4             // workgroups that share the columns of W
#define MWP_ET 256           // entry threads per workgroup: MWP_W columns x MWP_N rows
#ifndef MWP_AHEAD
#define MWP_AHEAD 0          // steps between the loads of a pivot column and its delivery (< MWP_NL).  A workgroup settles this many steps (plus the publisher's
                             // delay and one round trip) behind the stage it reads from: a column asked for before it exists comes back stale, and the poll and
                             // the second round trip of that case (1.4 us) sit on the step until the workgroup has fallen far enough behind to find its columns at
                             // the first try.  With the loads MWP_NL = 4 steps ahead (round 4) that was 4.6-4.9 us per stage boundary (scripts/pipe_substamps.py).
#endif
#define MWP_NL 4             // LOADER waves: each fetches every MWP_NL-th pivot column, MWP_NL steps ahead of the entry waves (the first one also publishes this stage's own)
#define MWP_NT (MWP_ET + 64 * MWP_NL)
#define MWP_SPIN_LIMIT (1 << 22)
#define MWP_GRANULES(K) ((K) * MWP_N * 2)                 // granules of one published pivot column
#define MWP_PC_WORDS(K) ((long)MWP_N * MWP_GRANULES(K))    // ... of one matrix (MWP_N pivots)
// The same pipeline for matrices of 33 .. 64 rows (round 5: template parameter NR = 32 or 64 rows per column; MWP_N above is the 32 of the blocked path's
// diagonal blocks and of the first form): 8 NR entry threads + four loader waves per workgroup, NR / 8 stages + NR / 8 workgroups for W.
#define MWP_N64 64
#define MWP_NT64 (MWP_W * MWP_N64 + 64 * MWP_NL)
#define MWP_GRANULES_N(K, NR) ((K) * (NR) * 2)
#define MWP_PC_WORDS_N(K, NR) ((long)(NR) * MWP_GRANULES_N(K, NR))

struct MwPipeMat {           // one matrix of a pipelined factorisation
    double *keep;            // null, or where a copy of the input goes (same layout as `in`)
    const double *in;        // input: planar n x n (leading dimension in_ld), lower triangle read; in_slots > 1: the sum of that many arrays, in order
    long inplane, in_stride;
    int in_slots, n;
    int in_ld, l_ld, inv_ld, pad;      // leading dimensions of in (and keep), L, Inv: n for a matrix of its own, more for a diagonal block of a larger one (k_mw_bp_diag_pipe)
    double *L, *rd, *Inv;    // outputs: factor (strict upper triangle zeroed), reciprocal diagonal, inverse factor (upper triangle zeroed); L may be `in` (every entry
                             // is read by the one thread that owns it, before anything is written)
    long lplane, rdplane, invplane;
    unsigned long long *pc;  // MWP_PC_WORDS granules: the published pivot columns
    int fail_code;           // atomicMin'ed into info[0] at a non-positive pivot
    unsigned long long *stamps;   // diagnostic builds (-DCLRS_MW_STAMPS): [roles][40] time stamps, or null
    unsigned long long *sub = nullptr;      // diagnostic builds: [8 roles][32 steps][4 who][4 points] stamps inside the steps, or null
};

namespace mwk {

__device__ __forceinline__ void mwp_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long mwp_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// publish entry (i, k) of pivot column k: 2 K granules
template <int K>
__device__ __forceinline__ void mwp_publish(unsigned long long *pcol, unsigned tag, int i, const mw<K> &v) {
#pragma unroll
    for (int l = 0; l < K; l++) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v.l[l]), t = (unsigned long long)tag << 32;
        mwp_store(pcol + ((long)l * MWP_N + i) * 2, t | (b & 0xffffffffull));
        mwp_store(pcol + ((long)l * MWP_N + i) * 2 + 1, t | (b >> 32));
    }
}
// the loader wave (64 lanes) publishes rows r0 .. n-1 of a pivot column held in LDS (planar, plane MWP_N) -- and, in the granules of row 0 (which no
// column k >= 1 uses), the running product s_k of the scaled pivots (sk: limb l at sk[l splane]; null for column 0, whose s_0 = 1): the workgroups behind
// this one read s_k there instead of repeating the chain of K-limb products s_(k+1) = s_k dh_k step by step (0.9 us on the first loader wave of every step
// that applies an incoming column, 0.2 us of it on the step's barrier: scripts/pipe_substamps.py).  The same products in the same order: the same bits.
template <int K, int NR>
__device__ __forceinline__ void mwp_publish_column(unsigned long long *pcol, unsigned tag, int r0, int n, const lds_d *buf, int lane, const lds_d *sk = nullptr, int splane = 0) {
    const int rows = n - r0, ks = sk ? K : 0, cnt = rows * K + ks;
    const unsigned long long t = (unsigned long long)tag << 32;
    for (int e = lane; e < cnt; e += 64) {                       // s_k first: the sentinel (limb K - 1 of row n - 1) stays the last granule stored
        const int e2 = e - ks;
        const int l = e < ks ? e : e2 / rows, i = e < ks ? 0 : r0 + e2 % rows;
        const unsigned long long b = (unsigned long long)__double_as_longlong(e < ks ? (double)sk[(long)l * splane] : (double)buf[(long)l * NR + i]);
        unsigned long long *g = pcol + ((long)l * NR + i) * 2;
        const unsigned long long g0 = t | (b & 0xffffffffull), g1 = t | (b >> 32);
#ifdef MWP_PLAIN_PUBLISH
        *(volatile unsigned long long *)g = g0;
        *(volatile unsigned long long *)(g + 1) = g1;
#else
        mwp_store(g, g0);
        mwp_store(g + 1, g1);
#endif
    }
#ifdef MWP_PLAIN_PUBLISH
    // Tried and NOT used (-DMWP_PLAIN_PUBLISH): plain stores, which leave the lines in this XCD's L2 where the sc1 loads of a consumer on the same XCD would
    // find them (the kernels' block map puts the workgroups of a matrix on one XCD), and one agent-scope write-back of the L2 behind them for consumers
    // elsewhere.  The wait for the stores and the write-back hold the publishing wave for ~2 us, and with it the barrier of every one of this stage's own
    // steps: 3.3 us per step instead of 1.45, kernel 103 us instead of 86 (measured).  Plain stores ON TOP of the sc1 stores: the same.
    asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1" ::: "memory");
#endif
}
// A loader wave (64 lanes) fetches row `prow` (the pivot) and rows r0 .. n-1 of a pivot column in two halves: `issue` sends every load of the column
// at once, `complete` -- a step later -- looks at the tags, polls the granules that were not there yet one by one and writes the numbers to LDS
// (planar, plane MWP_N); it returns false when a granule never arrived.  A hop is ~2.5 us (measured, scripts/pipe_stamps.py: the producer's sc1
// stores drop the line from its L2, the consumer's sc1 loads go to memory; 4-5 us when the first poll comes too early), the producers publish a column
// every 1.5 us: with ONE column in flight a consumer ran at 2.6 us per step, with two at 2.1.  So MWP_NL loader waves take turns: wave p fetches the
// columns j = p (mod MWP_NL), sends for column j + MWP_NL the moment it has delivered column j, and so has MWP_NL steps per column.
template <int K, int NR>
struct MwpFetch {
    static constexpr int R = (K * NR + 63) / 64;               // passes of the wave over the K NR numbers of a whole column
    unsigned long long a[R], b[R];
    int off[R];                                                 // (l MWP_N + i) of this lane's numbers: the same for every column (no index arithmetic per step)
    bool want[R];                                               // of the column in flight: the rows this workgroup reads (the pivot, and rows r0 ..)
    const unsigned long long *col;
    __device__ __forceinline__ void init(int n, int lane) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int t = lane + 64 * r, l = t / NR, i = t % NR;
            off[r] = (l < K && i < n) ? l * NR + i : -1;
            want[r] = false; a[r] = b[r] = 0;
        }
        col = nullptr;
    }
    __device__ __forceinline__ void issue(const unsigned long long *pcol, int prow, int r0) {      // (columns >= 1: row 0 carries the running product s_k)
        col = pcol;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = off[r] & (NR - 1);
            want[r] = off[r] >= 0 && (i == prow || i >= r0 || (i == 0 && prow > 0));
            if (want[r]) {
                const unsigned long long *g = pcol + (long)off[r] * 2;
                a[r] = mwp_load(g);
                b[r] = mwp_load(g + 1);
            }
        }
    }
    // `sentinel`: the granule the publisher stores LAST (limb K - 1 of row n - 1, second half).  A column that was not there when `issue` ran is not
    // polled granule by granule -- the waves of the workgroups behind one producer would hammer the lines its stores are queued at (measured: the
    // producer's own steps 2.5 us instead of 1.45, in both forms: whole columns and only the rows read) -- but by ONE address per wave; when its tag
    // is there, the rows are loaded again, and any granule that still is not is polled like that again.
    __device__ __forceinline__ bool complete(unsigned tag, lds_d *buf, const unsigned long long *sentinel) {
        int spins = 0;
        for (;;) {
            bool mine = true;
#pragma unroll
            for (int r = 0; r < R; r++) mine = mine && (!want[r] || ((unsigned)(a[r] >> 32) == tag && (unsigned)(b[r] >> 32) == tag));
            if (__all(mine)) break;
            // (asking for the rows themselves again instead of the sentinel first -- one round trip less once the column is there -- moves nothing: 0.4157
            // against 0.4156 ms per iteration on cohnelkies(8,15), 2.061 against 2.062 on Nsphere_packing N = 3)
            while ((unsigned)(mwp_load(sentinel) >> 32) != tag && spins < MWP_SPIN_LIMIT) {
                __builtin_amdgcn_s_sleep(2);
                spins++;
            }
            if (++spins >= MWP_SPIN_LIMIT) break;
#pragma unroll
            for (int r = 0; r < R; r++)
                if (want[r]) {
                    const unsigned long long *g = col + (long)off[r] * 2;
                    a[r] = mwp_load(g); b[r] = mwp_load(g + 1);
                }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (want[r]) buf[off[r]] = __longlong_as_double((long long)((a[r] & 0xffffffffull) | (b[r] << 32)));
        return spins < MWP_SPIN_LIMIT;
    }
};

// The barrier of a pivot step: LDS traffic only.  __syncthreads() also waits for the vector-memory counter, i.e. for the write-through (sc1) stores of
// the pivot column this step has just published -- a fabric round trip on the dependent chain of every pivot (measured: 3 us per step instead of 1.5).
// The published granules need no ordering (each carries its tag); what the next step reads comes from LDS.
__device__ __forceinline__ void mwp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one elimination step on one entry: v <- (dh v - (ci ph)(cj ph)) with dh = d 2^-ex, ph = 2^(-ex/2)  (wg_potrf's step)
template <int K>
__device__ __forceinline__ mw<K> mwp_step(const mw<K> &dh, double ph, const mw<K> &v, const mw<K> &ci, const mw<K> &cj) {
    acc<K> s;
    acc_zero<K>(s);
    acc_fma<K, K, K>(s, dh, v);
    acc_fma<K, K, K>(s, mul_pow2<K>(ci, ph), mul_pow2<K>(cj, ph), -1.0);
    return acc_result<K>(s);
}

// LDS of one workgroup: two pivot-column buffers, the products s_k of the scaled pivots, the pivots, the post-processing factors, W's row k
template <int K, int NR>
struct MwpLds {                                      // (no arrays of pointers: indexing one with k & 1 would put it in scratch memory, a round trip per pivot)
    lds_d *col0, *us, *dd, *fs, *rs, *wrow0;
    int *flag;
    __device__ __forceinline__ MwpLds(lds_d *base) {
        col0 = base;
        us = col0 + 2L * K * NR; dd = us + (long)K * (NR + 1); fs = dd + (long)K * NR; rs = fs + (long)K * NR;
        wrow0 = rs + (long)K * NR;
        flag = (int *)(wrow0 + 2L * K * MWP_W);
    }
    __device__ __forceinline__ lds_d *col(int k) const { return col0 + (long)(k & 1) * K * NR; }
    __device__ __forceinline__ lds_d *wrow(int k) const { return wrow0 + (long)(k & 1) * K * MWP_W; }
};
// LDS a workgroup asks for.  (Asking for more than half of a compute unit's 160 KB, so that no two workgroups share a compute unit, was tried with the
// plain-store hand-offs below and changed nothing: their 3.5 us steps came from the stores, not from sharing.)
#define MWP_LDS_ALONE (MWP_LDS_DOUBLES(10) * sizeof(double))
#define MWP_LDS_DOUBLES(K) ((K) * (6 * MWP_N + 1 + 2 * MWP_W) + 2)
#define MWP_LDS_ALONE64 (((10) * (6 * MWP_N64 + 1 + 2 * MWP_W) + 2) * sizeof(double))

// role < stages: stage `role` of the elimination of M; role >= stages: workgroup role - stages of the MWP_WW that form W.  Returns false at a
// non-positive pivot (or a hand-off that never arrived); every workgroup of the matrix then stops at the same pivot.
// KS: limbs of the arrays in memory (input, keep, L, rd, Inv); K <= KS: limbs of the elimination (MwDev::kf).  K < KS: the input is truncated to K limbs
// (the copy `keep` carries all KS), the results are stored with their upper KS - K planes zero.
template <int KS, int K, int NR = MWP_N>
__device__ __forceinline__ bool mwp_run(const MwPipeMat &m, int role, unsigned epoch, int *info, int tid) {
    const int n = m.n, stages = (n + MWP_W - 1) / MWP_W;
    constexpr int ET = MWP_W * NR, WW = NR / MWP_W;             // entry threads (one entry each: MWP_W columns x NR rows); workgroups that share W's columns
    MwpLds<K, NR> S(MW_LDS);
    const bool loader = tid >= ET;                                           // the last four waves: hand-offs and the running product s_k
    const int lw = (tid - ET) >> 6, lane = (tid - ET) & 63;                  // (loader waves only)
    const int cc = (tid / NR) & 7, i = tid % NR;
    const bool is_w = role >= stages;
    const int c0 = is_w ? 0 : role * MWP_W, c1 = is_w ? n : min(c0 + MWP_W, n);
    const int c = is_w ? (role - stages) + WW * cc : c0 + cc;                // my column
    const bool live = !loader && c < c1 && i >= c && i < n;
    const int nfetch = is_w ? n : c0;                                        // columns 0 .. nfetch - 1 come from other workgroups
    mw<K> v = zero<K>();
    if (!is_w && live) {
        mw<KS> vin;
        if (m.in_slots > 1) {
            acc<KS> s;
            acc_zero<KS>(s);
            for (int r = 0; r < m.in_slots; r++) acc_add<KS, KS>(s, ldx<KS>(m.in + (long)r * m.in_stride, m.inplane, i + (long)c * m.in_ld));
            vin = acc_result<KS>(s);
        } else vin = ldx<KS>(m.in, m.inplane, i + (long)c * m.in_ld);
        if (m.keep) {                                                        // the matrix as it came in (both triangles), for the residuals of the refined solve
            stx<KS>(m.keep, m.inplane, i + (long)c * m.in_ld, vin);
            stx<KS>(m.keep, m.inplane, c + (long)i * m.in_ld, vin);
        }
        v = cvt<K, KS>(vin);
    }
    if (tid == ET) { stx<K>(S.us, NR + 1, 0, from_double<K>(1.0)); *S.flag = 1; }
    MwpFetch<K, NR> F;
    F.init(n, lane);
    const unsigned tag0 = epoch << (NR > 32 ? 6 : 5);                          // tag = (launch epoch, pivot): 26 bits of epoch, 5 or 6 of pivot
    // column 0: stage 0 owns it, everybody else fetches it; columns 1 .. MWP_NL are on their way when step 0 begins
    if (!is_w && c0 == 0) {
        if (live && c == 0) stx<K>(S.col(0), NR, i, v);
    } else if (loader && lw == 0) {
        F.issue(m.pc, 0, is_w ? 0 : c0);
        if (!F.complete(tag0, S.col(0), m.pc + ((long)(K - 1) * NR + n - 1) * 2 + 1)) *S.flag = 0;
    }
    if (loader) {                                                            // columns 1 .. MWP_AHEAD: asked for now, by the waves that will deliver them
        const int j = lw == 0 ? MWP_NL : lw;
        if (j <= MWP_AHEAD && j < nfetch) F.issue(m.pc + (long)j * MWP_GRANULES_N(K, NR), j, is_w ? j : c0);
    }
    __syncthreads();
    const int kend = is_w ? n : c1;                                          // pivots this workgroup looks at: 0 .. kend - 1
#ifdef CLRS_MW_STAMPS            // diagnostic builds: wall clock (100 MHz) of thread 0 at the top of every step, per role, for the matrix with m.stamps != null
    unsigned long long *stamps = m.stamps ? m.stamps + (long)role * 40 : nullptr;
    if (stamps && tid == 0) stamps[39] = wall_clock64();
    // ... and inside the steps (g_mwp_sub[role][step][who][point], scripts/pipe_substamps.py): who 0 / 1 = first lane of entry wave 0 / of the last entry wave,
    // 2 = first loader wave, 3 = the loader wave whose turn the step is; points: top, pivot read and scaled, arithmetic / hand-off done, behind the barrier
    const int sub_who = !m.sub || role >= 8 ? -1 : tid == 0 ? 0 : tid == ET - 64 ? 1 : tid == ET ? 2 : -1;
#define MWP_SUB(k_, who_, pt_) do { if ((who_) >= 0 && (k_) < 32) m.sub[((role * 32 + (k_)) * 4 + (who_)) * 4 + (pt_)] = wall_clock64(); } while (0)
#else
#define MWP_SUB(k_, who_, pt_) do { } while (0)
#endif
    for (int k = 0; k < kend; k++) {
#ifdef CLRS_MW_STAMPS
        if (stamps && tid == 0) stamps[k] = wall_clock64();
#endif
        lds_d *cur = S.col(k), *nxt = S.col(k + 1);
#ifdef CLRS_MW_STAMPS
        const int sub_turn = (m.sub && role < 8 && loader && lane == 0 && (k + 1) % MWP_NL == lw) ? 3 : -1;
#endif
        MWP_SUB(k, sub_who, 0); MWP_SUB(k, sub_turn, 0);
        if (MWP_AHEAD == 0 && loader && (k + 1) % MWP_NL == lw && k + 1 < nfetch) F.issue(m.pc + (long)(k + 1) * MWP_GRANULES_N(K, NR), k + 1, is_w ? k + 1 : c0);
        const mw<K> d = ldx<K>(cur, NR, k);
        if (!(*S.flag) || !(d.l[0] > 0.0)) {                                 // uniform: every thread reads the same words
            // the owner of a column with a non-positive pivot publishes it all the same: the workgroups behind it read the pivot there and stop at the same
            // step -- without it they would poll for the column until their bound (1.3 s: found on the 16-cluster weak-scaling instance, whose solve ends
            // with a failed factorisation)
            if (loader && lw == 0 && !is_w && k >= c0) mwp_publish_column<K, NR>(m.pc + (long)k * MWP_GRANULES_N(K, NR), tag0 | (unsigned)k, k, n, cur, lane, k ? S.us + k : nullptr, NR + 1);
            if (tid == 0) atomicMin(info, m.fail_code);
            return false;
        }
        if (is_w && tid == ET) stx<K>(S.dd, NR, k, d);
        // a column this stage owns goes to the stages and the W workgroups behind it from HERE, out of LDS, by a loader wave: a store of the entry waves
        // would put its wait for the write-through on the dependent chain (the compiler guards the stored registers with s_waitcnt vmcnt)
        if (loader && lw == 0 && !is_w && k >= c0) mwp_publish_column<K, NR>(m.pc + (long)k * MWP_GRANULES_N(K, NR), tag0 | (unsigned)k, k, n, cur, lane, k ? S.us + k : nullptr, NR + 1);
        // s_k of a column that came from another workgroup: it travelled in the granules of row 0 (kept for the post-processing)
        if (tid == ET && k >= 1 && k < nfetch) stx<K>(S.us, NR + 1, k, ldx<K>(cur, NR, 0));
        if (k + 1 >= kend) break;
        double p1, ph;
        pivot_scale(d.l[0], p1, ph);
        const mw<K> dh = mul_pow2<K>(d, p1);
        MWP_SUB(k, sub_who, 1); MWP_SUB(k, sub_turn, 1);
        if (loader) {
            if ((k + 1) % MWP_NL == lw) {                                    // my turn: deliver column k + 1 (sent for MWP_AHEAD steps ago)
                if (k + 1 < nfetch && !F.complete(tag0 | (unsigned)(k + 1), nxt, m.pc + (long)(k + 1) * MWP_GRANULES_N(K, NR) + ((long)(K - 1) * NR + n - 1) * 2 + 1)) *S.flag = 0;
            }
            if (MWP_AHEAD > 0 && (k + 1 + MWP_AHEAD) % MWP_NL == lw) {       // ... and the wave that delivers column k + 1 + MWP_AHEAD sends for it (MWP_AHEAD = 0: at the top of the step)
                const int j = k + 1 + MWP_AHEAD;
                if (j < nfetch) F.issue(m.pc + (long)j * MWP_GRANULES_N(K, NR), j, is_w ? j : c0);
            }
            // s_(k+1) = s_k dh_k: one K-limb product per step, off the chain, on the first loader wave: it shares its SIMD with entry wave 0, which in a
            // stage's own steps is the first to run out of live columns (moving it to another SIMD made those steps 1.9-2.9 us instead of 1.45: measured)
            // (only where column k + 1 is this stage's own: the s_k of the others arrive with their columns)
            if (tid == ET && k + 1 >= nfetch) stx<K>(S.us, NR + 1, k + 1, mul<K>(k >= 1 && k < nfetch ? ldx<K>(cur, NR, 0) : ldx<K>(S.us, NR + 1, k), dh));
        } else if (!is_w) {
            if (live && c > k) v = mwp_step<K>(dh, ph, v, ldx<K>(cur, NR, i), ldx<K>(cur, NR, c));
            if (k + 1 >= c0 && live && c == k + 1) stx<K>(nxt, NR, i, v);      // my stage's column k + 1 is final now: to this stage's next step
        } else {
            // W: entries (i, c) with c <= k < i; row k of W comes from the thread that owns (k, c), through LDS; w_kk = s_k
            if (live && c <= k && i > k) {
                const mw<K> wk = c == k ? (k == 0 ? ldx<K>(S.us, NR + 1, 0) : ldx<K>(cur, NR, 0)) : ldx<K>(S.wrow(k), MWP_W, cc);
                v = mwp_step<K>(dh, ph, v, ldx<K>(cur, NR, i), wk);
            }
            if (live && i == k + 1 && c <= k) stx<K>(S.wrow(k + 1), MWP_W, cc, v);
        }
        MWP_SUB(k, sub_who, 2); MWP_SUB(k, sub_turn, 2);
        mwp_barrier();
        MWP_SUB(k, sub_who, 3); MWP_SUB(k, sub_turn, 3);
    }
    __syncthreads();
    // post-processing (wg_potrf's): f_k = 1 / sqrt(s_k d~_k); L_kk = d~_k f_k, 1 / L_kk = f_k s_k; L_ik = a~_ik f_k; (L^-1)_ij = W_ij f_i
    if (!is_w) {
        if (live && i == c) {
            const mw<K> sk = ldx<K>(S.us, NR + 1, c), f = rsqrt<K>(mul<K>(sk, v)), r = mul<K>(f, sk);
            stx<K>(S.fs, NR, cc, f);
            stx<KS>(m.rd, m.rdplane, c, cvt<KS, K>(r));
        }
        __syncthreads();
        if (!loader && c < c1 && i < n) stx<KS>(m.L, m.lplane, i + (long)c * m.l_ld, i >= c ? cvt<KS, K>(mul<K>(v, ldx<K>(S.fs, NR, cc))) : zero<KS>());
    } else {
        if (tid < n) {                                                       // (every W workgroup needs every f_i: n reciprocal square roots side by side)
            const mw<K> sk = ldx<K>(S.us, NR + 1, tid), dt = ldx<K>(S.dd, NR, tid), f = rsqrt<K>(mul<K>(sk, dt));
            stx<K>(S.fs, NR, tid, f);
            stx<K>(S.rs, NR, tid, mul<K>(f, sk));
        }
        __syncthreads();
        if (!loader && c < n && i < n) stx<KS>(m.Inv, m.invplane, i + (long)c * m.inv_ld, cvt<KS, K>(i > c ? mul<K>(v, ldx<K>(S.fs, NR, i)) : i == c ? ldx<K>(S.rs, NR, i) : zero<K>()));
    }
#ifdef CLRS_MW_STAMPS
    if (stamps && tid == 0) stamps[38] = wall_clock64();
#endif
    return true;
}

}  // namespace mwk

// Block index -> (matrix, role): blocks are dealt to the 8 XCDs round-robin (observed, not promised: MI355X_MICROARCH.md), so the 8 workgroups of a matrix
// (up to 4 stages + 4 for W) get block indices that agree modulo 8 -- matrix j of a group of 8 matrices has the blocks 64 (j / 8) + (j % 8) + 8 role.
// Earlier stages have lower block indices.  Placement is a speed matter only: the hand-off is correct anywhere.
#define MWP_ROLES 8
__device__ __forceinline__ void mwp_block_map(int b, int &matrix, int &role) {
    matrix = (b & 7) + 8 * (b >> 6);
    role = (b >> 3) & 7;
}
__host__ __device__ static inline int mwp_blocks(int matrices) { return 64 * ((matrices + 7) / 8); }

// L_j = chol(S_j) and L_j^-1 of every cluster with P <= MWP_N: mwp_blocks(J) workgroups
template <int K>
__global__ __launch_bounds__(MWP_NT) void k_mw_factor_pipe(const MwDev q, unsigned epoch) {
    using namespace mwk;
    mw_mark(q);
    int j, role;
    mwp_block_map(blockIdx.x, j, role);
    if (j >= q.J) return;
    const MwClu &c = q.clu[j];
    const int P = c.P, stages = (P + MWP_W - 1) / MWP_W;
    if (role >= stages + MWP_WW) return;
    MwPipeMat m;
    m.in = q.S + c.Soff; m.inplane = q.Slen; m.in_stride = 0; m.in_slots = 1; m.n = P; m.keep = q.S0 + c.Soff;
    m.in_ld = m.l_ld = m.inv_ld = P;
    m.L = q.S + c.Soff; m.lplane = q.Slen; m.rd = q.srd + c.coff; m.rdplane = q.xlen; m.Inv = q.Si + c.Soff; m.invplane = q.Slen;
    m.pc = q.pipe_pc + (long)j * MWP_PC_WORDS(K);
    m.fail_code = j + 1;
    m.stamps = j == 0 && q.pipe_stamps ? q.pipe_stamps : nullptr;
    m.sub = m.stamps ? m.stamps + 16 * 40 : nullptr;
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mwp_run<K, mw_kf_of(K)>(m, role, epoch, &q.info[0], threadIdx.x); return; } }
    mwp_run<K, K>(m, role, epoch, &q.info[0], threadIdx.x);
}

// The same for clusters of 33 .. 64 rows (MWP_NT64 threads: 8 x 64 entry threads + four loader waves; up to eight stages and eight workgroups for W per
// matrix: sixteen roles, blocks 128 (j / 8) + (j % 8) + 8 role).  Bit for bit the factors of the one-workgroup kernel, like the 32-row form.
__device__ __forceinline__ void mwp_block_map64(int b, int &matrix, int &role) {
    matrix = (b & 7) + 8 * (b >> 7);
    role = (b >> 3) & 15;
}
__host__ __device__ static inline int mwp_blocks64(int matrices) { return 128 * ((matrices + 7) / 8); }
template <int K>
__global__ __launch_bounds__(MWP_NT64) void k_mw_factor_pipe64(const MwDev q, unsigned epoch) {
    using namespace mwk;
    if constexpr (K > 6) return;                        // (never launched beyond six limbs: twelve waves of that many registers do not fit a compute unit)
    else {
    mw_mark(q);
    int j, role;
    mwp_block_map64(blockIdx.x, j, role);
    if (j >= q.J) return;
    const MwClu &c = q.clu[j];
    const int P = c.P, stages = (P + MWP_W - 1) / MWP_W;
    if (role >= stages + MWP_N64 / MWP_W) return;
    MwPipeMat m;
    m.in = q.S + c.Soff; m.inplane = q.Slen; m.in_stride = 0; m.in_slots = 1; m.n = P; m.keep = q.S0 + c.Soff;
    m.in_ld = m.l_ld = m.inv_ld = P;
    m.L = q.S + c.Soff; m.lplane = q.Slen; m.rd = q.srd + c.coff; m.rdplane = q.xlen; m.Inv = q.Si + c.Soff; m.invplane = q.Slen;
    m.pc = q.pipe_pc + (long)j * MWP_PC_WORDS_N(K, MWP_N64);
    m.fail_code = j + 1;
    m.stamps = nullptr;
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mwp_run<K, mw_kf_of(K), MWP_N64>(m, role, epoch, &q.info[0], threadIdx.x); return; } }
    mwp_run<K, K, MWP_N64>(m, role, epoch, &q.info[0], threadIdx.x);
    }
}

// The Cholesky of the X blocks (and, beside it, of a second block-diagonal matrix Y whose inverse factors alone are kept: the step length's) through the same
// pipelines -- k_mw_potrf_x's work for contexts whose every block carries its explicit inverse factor (MwBlk::inv == 1): the 32-row form for blocks of at
// most 32 rows (its 512 threads: the other 256 of this launch leave at once), the 64-row form beyond.  What k_mw_potrf_x leaves besides -- the scaled
// triangles Xf / Xb of the substitution paths -- has no reader in such a context.  Bit for bit the factors and inverses of the one-workgroup kernel.
// Block index -> (matrix, role) as in k_mw_factor_pipe64 (sixteen roles per matrix); matrices NB .. 2 NB - 1 are the blocks of Y.
template <int K>
__global__ __launch_bounds__(MWP_NT64) void k_mw_potrf_x_pipe(const MwDev q, const double *__restrict__ X, double *__restrict__ Xc, const double *__restrict__ Y2, double *__restrict__ Yi,
                                                              int *__restrict__ yfail, double *__restrict__ scrL, double *__restrict__ scrRd, int *__restrict__ scrInfo,
                                                              unsigned long long *__restrict__ pcx, unsigned epoch) {
    using namespace mwk;
    if constexpr (K > 6) return;
    else {
    mw_mark(q);
    int mi, role;
    mwp_block_map64(blockIdx.x, mi, role);
    const int nmat = Y2 ? 2 * q.NB : q.NB;
    if (mi >= nmat) return;
    const bool second = mi >= q.NB;
    const int b = second ? mi - q.NB : mi;
    const MwBlk &k = q.blk[b];
    const int n = k.n;
    const bool big = n > MWP_N;
    const int stages = (n + MWP_W - 1) / MWP_W;
    if (role >= stages + (big ? MWP_N64 / MWP_W : MWP_WW)) return;
    if (!big && threadIdx.x >= MWP_NT) return;
    MwPipeMat m;
    m.in = (second ? Y2 : X) + k.xyoff; m.inplane = q.xylen; m.in_stride = 0; m.in_slots = 1; m.n = n; m.keep = nullptr;
    m.in_ld = m.l_ld = m.inv_ld = n;
    m.L = (second ? scrL : Xc) + k.xyoff; m.lplane = q.xylen;
    m.rd = (second ? scrRd : q.xrd) + k.rd_off; m.rdplane = q.xrdlen;
    m.Inv = (second ? Yi : q.Xi) + k.xyoff; m.invplane = q.xylen;
    m.pc = pcx + (long)mi * MWP_PC_WORDS_N(K, MWP_N64);
    m.fail_code = b + 1;
    m.stamps = nullptr;
    int *info = second ? scrInfo + b : &q.info[1];
    const bool ok = big ? mwp_run<K, K, MWP_N64>(m, role, epoch, info, threadIdx.x) : mwp_run<K, K, MWP_N>(m, role, epoch, info, threadIdx.x);
    if (second && role == 0 && threadIdx.x == 0) yfail[b] = ok ? 0 : 1;
    }
}

// L_Q = chol(Q), Q = the sum of the ranks' partial sums, and L_Q^-1: the blocks 0, 8, 16, ... of the first 64 (one XCD); the blocks from 64 on carry the
// first product pair of the next solve (as in k_mw_potrf_q), one cluster each
template <int K>
__global__ __launch_bounds__(MWP_NT) void k_mw_potrf_q_pipe(const MwDev q, unsigned epoch, const double *__restrict__ fwd_rhs, const int *__restrict__ wait_word, int wait_value) {
    using namespace mwk;
    if (blockIdx.x >= 64) {
        if (wait_word) mw_wait_word(wait_word, wait_value, &q.info[0], q.J + 1);
        if ((int)blockIdx.x >= 64 + q.J) {           // the second right-hand side (MwDev::ride2_rhs)
            MwDev q2 = q;
            q2.t = q.ride2_t; q2.u = q.ride2_u;
            mw_solve_fwd_cluster<K>(q2, blockIdx.x - 64 - q.J, q.ride2_rhs);
        } else mw_solve_fwd_cluster<K>(q, blockIdx.x - 64, fwd_rhs);
        return;
    }
    int mtx, role;
    mwp_block_map(blockIdx.x, mtx, role);
    const int N = q.N;
    if (mtx != 0 || role >= (N + MWP_W - 1) / MWP_W + MWP_WW) return;
    if (q.info[0] != MW_INFO_NONE) return;                          // a cluster failed: the reference throws before reaching Q
    MwPipeMat m;
    m.in = q.Qg; m.inplane = (long)N * N; m.in_stride = (long)K * N * N; m.in_slots = q.world; m.n = N; m.keep = nullptr;
    m.in_ld = m.l_ld = m.inv_ld = N;
    m.L = q.Q; m.lplane = (long)N * N; m.rd = q.qrd; m.rdplane = N; m.Inv = q.Qi; m.invplane = (long)N * N;
    m.pc = q.pipe_pc + (long)q.pipe_q * MWP_PC_WORDS(K);
    m.fail_code = q.J + 1;
    m.stamps = q.pipe_stamps ? q.pipe_stamps + 8 * 40 : nullptr;
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mwp_run<K, mw_kf_of(K)>(m, role, epoch, &q.info[0], threadIdx.x); return; } }
    mwp_run<K, K>(m, role, epoch, &q.info[0], threadIdx.x);
}

// The diagonal block of one block column of the blocked factorisation (k_mw_bp_diag's work: Cholesky of the MW_PB x MW_PB block at (j0, j0) of every matrix
// of the list and the inverse of its factor, both in place in the matrices' M / Mi) as the same pipeline: a 32-column block is four stages of eight columns
// plus four workgroups for the inverse instead of four workgroups that each repeat the whole elimination -- the diagonal blocks are the serial part of the
// blocked path (13 of them per iteration on Nsphere_packing(8,15,[1/2,1/2,1/2]): 40 % of its time).  Bit for bit the results of k_mw_bp_diag.
// A matrix whose earlier block column failed is eliminated all the same (its workgroups cannot agree on a status word that changes while they run): it fails
// again, with the same code.  Blocks beyond mwp_blocks(nm): first the clusters that fit in LDS (k_mw_bp_diag's ride; `ride` of them, MW_INV_WG workgroups
// each), then block row `inv_row` of the inverse factors (>= 1, or 0 for none: the blocks (inv_row, i), i < inv_row, of L^-1 -- k_mw_bp_inv_row's work; the row's
// own diagonal block was finished by an earlier launch), inv_row x MW_PB / MW_BP_IC workgroups per matrix: what was a chain of launches behind the last block
// column (one per block distance: 11 launches, 200 us per iteration on Nsphere_packing(8,15,[1/2,1/2,1/2])) runs beside the diagonal blocks, on other compute units.
template <int K>
__global__ __launch_bounds__(MWP_NT) void k_mw_bp_diag_pipe(const MwDev q, const MwBp *__restrict__ ms, int nm, int j0, unsigned epoch, int ride, int inv_row) {
    using namespace mwk;
    static_assert(MWP_NT == MW_PT && MWP_N == MW_PB_OF(K), "the ride of the LDS clusters and the panel width assume the blocked path's shapes");
    mw_mark(q);
    const int nbp = mwp_blocks(nm);
    if ((int)blockIdx.x >= nbp) {
        int r = (int)blockIdx.x - nbp;
        if (r < ride * MW_INV_WG) { mw_factor_cluster<K>(q, r / MW_INV_WG, r % MW_INV_WG, MW_INV_WG); return; }
        r -= ride * MW_INV_WG;
        constexpr int CB = MW_PB_OF(K) / MW_BP_IC;                   // column workgroups per block
        const int per = inv_row * CB;
        if (per == 0 || r >= per * nm) return;                       // (the launch has no such blocks)
        mw_bp_inv_block<K>(q, ms[r / per], (r % per) / CB, inv_row, r % CB);
        return;
    }
    int mtx, role;
    mwp_block_map(blockIdx.x, mtx, role);
    if (mtx >= nm) return;
    const MwBp b = ms[mtx];
    if (j0 >= b.n) return;
    const int nb = min(MWP_N, b.n - j0);
    if (role >= (nb + MWP_W - 1) / MWP_W + MWP_WW) return;
    const long at = j0 + (long)j0 * b.ld;
    MwPipeMat m;
    m.in = b.M + at; m.inplane = b.plane; m.in_stride = 0; m.in_slots = 1; m.n = nb; m.keep = nullptr;
    m.in_ld = m.l_ld = m.inv_ld = b.ld;
    m.L = b.M + at; m.lplane = b.plane; m.rd = b.rd + j0; m.rdplane = b.rdplane; m.Inv = b.Mi + at; m.invplane = b.plane;
    m.pc = q.pipe_pc + (long)(q.pipe_bp + b.slot) * MWP_PC_WORDS(K);
    m.fail_code = b.code;
    m.stamps = nullptr;
    if constexpr (mw_kf_of(K) < K) { if (q.kf < K) { mwp_run<K, mw_kf_of(K)>(m, role, epoch, &q.info[b.which], threadIdx.x); return; } }
    mwp_run<K, K>(m, role, epoch, &q.info[b.which], threadIdx.x);
}

#endif
