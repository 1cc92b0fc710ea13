/*
 * clrs_hip.h -- C ABI of the MI355X-native interior-point hot path for clustered low-rank SDPs.
 *
 * This is the drop-in boundary for the ONE path of nanleij/ClusteredLowRankSolver.jl (v2.1.0) that
 * this repository accelerates: per-iteration Schur-complement assembly from low-rank constraint
 * matrices and the block-Cholesky factor/solve of the cluster-block-diagonal normal equations.
 * The reference has no FFI seam for this path (it is plain Julia mutating preallocated Arb buffers);
 * each entry point below names the reference function it replaces (paths relative to the reference
 * repository).  INTEGRATION.md shows the Julia `ccall` shim a maintainer would add.
 *
 * Conventions
 *   - plain C: pointers, sizes, int return codes; no C++/torch types.
 *   - all matrices are COLUMN-MAJOR fp64 (Julia `Matrix{Float64}` layout).
 *   - block-diagonal iterates X, Y (and the Cholesky factors of X) are passed as ONE array: the
 *     blocks (j,l) in the order of `clrs_sdp_desc`, each n x n column-major, concatenated
 *     ("xy layout", length sum n^2).
 *   - S / L_j are passed as one array: per cluster P_j x P_j column-major, concatenated ("S layout").
 *   - x-like vectors (rhs_x, dx) concatenate the clusters (length sum P_j).
 *   - indices are 0-based; constraint indices are cluster-local (the reference's `cs_map`,
 *     src/solver.jl:156-167, is applied by the caller).
 *   - one context per GPU / per process; a context is not thread-safe (like the reference's shared
 *     scratch `tempX`, `part_r`).
 *   - return value: 0 ok; >0 factorisation failure (see clrs_schur_factor); <0 an error from
 *     `clrs_strerror`.
 *   - host entry points take host pointers and synchronise before returning; `_dev` entry points
 *     take device pointers, enqueue on the context's stream and do NOT synchronise.
 */
#ifndef CLRS_HIP_H
#define CLRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLRS_OK 0
#define CLRS_ERR_INVALID (-1)      /* malformed description / argument */
#define CLRS_ERR_HIP (-2)          /* a HIP runtime call failed (message via clrs_last_error) */
#define CLRS_ERR_NO_DEVICE (-3)    /* no usable gfx950 device */
#define CLRS_ERR_STATE (-4)        /* call order violated (e.g. solve before factor) */

/*
 * Numeric description of a ClusteredLowRankSDP restricted to what the hot path reads:
 * replaces the traversal of `sdp.A[j][l][r,s][p]`, `sdp.B[j]` (src/interface.jl:807-819) done by
 * precompute_matrices_bilinear_pairings (src/solver.jl:985-1059).
 *
 * Blocks are listed cluster by cluster.  A low-rank block (kind 0) is an m x m grid of delta x delta
 * sub-blocks; each term t is one rank-1 piece lambda * vs * ws^T of A[j][l][r,s][p]
 * (LowRankMat, src/interface.jl:759-763).  The convention A[r,s][p] = A[s,r][p]^T
 * (src/solver.jl:1009) is required: for every term (p,r,s,rank) the term (p,s,r,rank) must exist.
 * A dense ("high rank") block (kind 1) has m == 1 and lists its matrices A[j][l][1,1][p].
 */
typedef struct clrs_sdp_desc {
    int32_t n_clusters;            /* J */
    int32_t n_free;                /* N, number of free variables (columns of B) */
    const int32_t *cluster_P;      /* [J] constraints per cluster */
    const double *B;               /* per cluster P_j x N column-major, concatenated */
    int32_t n_blocks;              /* total number of PSD blocks (j,l) */
    const int32_t *block_cluster;  /* [NB] cluster of each block (non-decreasing) */
    const int32_t *block_m;        /* [NB] sub-blocks per side */
    const int32_t *block_delta;    /* [NB] sub-block size; block side n = m*delta */
    const int32_t *block_kind;     /* [NB] 0 = low rank, 1 = dense */
    const int64_t *term_ptr;       /* [NB+1] CSR: terms of block b are term_ptr[b]..term_ptr[b+1]-1 */
    const int32_t *term_p;         /* [T] constraint index in the cluster */
    const int32_t *term_r;         /* [T] sub-block row */
    const int32_t *term_s;         /* [T] sub-block column */
    const int32_t *term_rank;      /* [T] index of the rank-1 piece inside A[r,s][p] */
    const double *term_lambda;     /* [T] */
    const int64_t *term_vec_ptr;   /* [T+1] offsets of the term's vectors in term_vs / term_ws */
    const double *term_vs;         /* delta doubles per term */
    const double *term_ws;         /* delta doubles per term */
    const int64_t *dense_ptr;      /* [NB+1] CSR over dense entries */
    const int32_t *dense_p;        /* [D] constraint index in the cluster */
    const int64_t *dense_A_ptr;    /* [D+1] offsets into dense_A */
    const double *dense_A;         /* n x n column-major per entry */
} clrs_sdp_desc;

typedef struct clrs_ctx clrs_ctx;

/* Sizes of the flat layouts, for buffer allocation by the caller. */
typedef struct clrs_dims {
    int64_t xy_len;    /* sum n^2 over blocks */
    int64_t x_len;     /* sum P_j */
    int64_t S_len;     /* sum P_j^2 */
    int64_t n_terms;   /* T */
    int32_t n_free;    /* N */
    int32_t n_clusters;
    int32_t n_blocks;
    int32_t reserved;
} clrs_dims;

/* Create a context on HIP device `device`: de-duplicates the sampled vectors, builds the gather
 * tables and the launch plan, uploads all static data and allocates every device buffer.
 * Replaces: precompute_matrices_bilinear_pairings (src/solver.jl:985-1059), the preallocation
 * block src/solver.jl:298-317 and ThreadingInfo (src/threadinginfo.jl:59-102). */
int clrs_ctx_create(const clrs_sdp_desc *desc, int device, clrs_ctx **out);
void clrs_ctx_destroy(clrs_ctx *ctx);
int clrs_get_dims(const clrs_ctx *ctx, clrs_dims *dims);

/* Number of unique right / left vectors of sub-block row r of block b after de-duplication
 * (sizes of rightvecs[j][l][r] / leftvecs[j][l][r], src/solver.jl:1018-1051). */
int clrs_get_unique_counts(const clrs_ctx *ctx, int32_t block, int32_t r, int32_t *n_right, int32_t *n_left);

/* Cluster sharding across GPUs (one process per GPU) is done by the caller: each rank creates its
 * context from the sub-description holding only ITS clusters (all N free variables), calls
 * clrs_schur_factor_local_dev, sums the partial Q (clrs_q_buffer_dev) over ranks with one RCCL
 * all-reduce and calls clrs_schur_factor_finish_dev; the solve is split the same way around the
 * all-reduce of the partial u = LinvB^T t (clrs_u_buffer_dev).  clrs_schur_solve_fwd_dev may be called right after
 * clrs_schur_factor_local_dev (it needs only L_j and LinvB_j), and the u buffer lies directly behind the Q buffer
 * (clrs_u_buffer_dev == clrs_q_buffer_dev + N * N): the exchange of Q can be deferred to the first solve after a
 * factorisation and merged with that solve's exchange of u into ONE all-reduce of N * N + N doubles, followed by
 * clrs_schur_factor_finish_dev and clrs_schur_solve_bwd_dev (clrs_amd.sharded.ShardedSchur does this). */

/* Lower Cholesky factors of all X blocks: Xchol_blk = chol(X_blk).
 * Replaces: the approx_cholesky!(X_inv_blk, X_blk) loop, src/solver.jl:388-399.
 * Returns 0, or b+1 for the first block whose pivot is not positive. */
int clrs_cholesky_blocks(clrs_ctx *ctx, const double *X, double *Xchol);

/* Schur complement assembly S[j][p,q] = sum_l <A_p, X^-1 A_q Y> from the Cholesky factors of X
 * (xy layout) and Y.  Optionally returns S (S layout, full symmetric) and the per-term bilinear
 * pairings w^T Y v (A_Y, length T, in term order).
 * Replaces: compute_S_integrated! (src/solver.jl:1062-1226). */
int clrs_schur_assemble(clrs_ctx *ctx, const double *Xchol, const double *Y, double *S_out, double *AY_out);

/* S_j = L_j L_j^T, LinvB_j = L_j^-1 B_j, Q = sum_j LinvB_j^T LinvB_j, Q = L_Q L_Q^T.
 * Replaces: steps 3-4 of compute_T_decomposition! (src/solver.jl:1244-1279).
 * Returns 0; j+1 if S_j is not positive definite ("S was not decomposed succesfully in block j",
 * :1249); n_clusters+1 if Q is not ("Q was not decomposed correctly", :1277). */
int clrs_schur_factor(clrs_ctx *ctx);

/* Copies of the factors for inspection / for the caller's buffers `S` (holding L_j), `LinvB`, `Q`
 * (holding L_Q) of compute_T_decomposition!.  Any pointer may be NULL.
 * L: S layout (strict upper triangles zero); LinvB: per cluster P_j x N column-major, concatenated;
 * LQ: N x N. */
int clrs_get_factor(clrs_ctx *ctx, double *L, double *LinvB, double *LQ);

/* Solve [S -B; B^T 0] (dx; dy) = (rhs_x; rhs_y) with the current factorisation.
 * Replaces: the "solve system" stage of compute_search_direction! (src/solver.jl:1527-1582). */
int clrs_schur_solve(clrs_ctx *ctx, const double *rhs_x, const double *rhs_y, double *dx, double *dy);

/* --- device-pointer / sharded variants (enqueue on the context stream, no synchronisation) --- */
int clrs_cholesky_blocks_dev(clrs_ctx *ctx, const double *d_X, double *d_Xchol);
int clrs_sync_status_cholesky(clrs_ctx *ctx);   /* blocks; status of the last clrs_cholesky_blocks_dev: 0 or block b+1 */
int clrs_schur_assemble_dev(clrs_ctx *ctx, const double *d_Xchol, const double *d_Y);
int clrs_schur_factor_dev(clrs_ctx *ctx);                     /* whole factorisation on one GPU (no exchange, no sync) */
int clrs_schur_solve_dev(clrs_ctx *ctx, const double *d_rhs_x, const double *d_rhs_y, double *d_dx, double *d_dy);
int clrs_schur_factor_local_dev(clrs_ctx *ctx);               /* up to the local partial Q */
double *clrs_q_buffer_dev(clrs_ctx *ctx);                     /* N x N device buffer holding Q (partial, then total) */
int clrs_schur_factor_finish_dev(clrs_ctx *ctx);              /* Cholesky of the (summed) Q */
int clrs_schur_solve_fwd_dev(clrs_ctx *ctx, const double *d_rhs_x);   /* t = L^-1 rhs_x ; u = LinvB^T t (partial) */
double *clrs_u_buffer_dev(clrs_ctx *ctx);                     /* N doubles: the partial u, to be summed over ranks */
int clrs_schur_solve_bwd_dev(clrs_ctx *ctx, const double *d_rhs_y, double *d_dx, double *d_dy);
double *clrs_S_buffer_dev(clrs_ctx *ctx);                     /* S layout; holds S after assemble, L_j after factor */
double *clrs_AY_buffer_dev(clrs_ctx *ctx);                    /* [T] */
/* Blocks until the stream is idle and returns the factorisation status of the last factor call
 * (same codes as clrs_schur_factor). */
int clrs_sync_status(clrs_ctx *ctx);
void *clrs_stream(clrs_ctx *ctx);                             /* hipStream_t of the context */

/* --- device-resident interior-point iteration around the path -------------------------
 * The whole loop body of solvesdp (src/solver.jl:348-589) with x, y, X, Y and every intermediate in HBM: residuals
 * (:863-918, 961-983), search directions (:1474-1616, with the solves of this library), step lengths (:1620-1693) and the
 * update (:485-495).  The host sequences kernels and reads one record per iteration.  Available when every PSD block
 * fits in LDS (n <= ~48); otherwise clrs_ipm_create returns CLRS_ERR_INVALID and the caller keeps its own loop. */
typedef struct clrs_ipm_data {
    const double *C;      /* objective matrices, xy layout (sdp.C) */
    const double *c;      /* right-hand sides, x layout (sdp.c) */
    const double *b;      /* objective of the free variables [N] (sdp.b) */
    int32_t maximize;     /* sdp.maximize */
    int32_t reserved;
    double constant;      /* objective constant */
} clrs_ipm_data;
typedef struct clrs_ipm_params {   /* keyword arguments of solvesdp, src/solver.jl:100-127 */
    double beta_infeasible, beta_feasible, gamma;
    double dual_error_threshold, primal_error_threshold, max_complementary_gap, step_length_threshold;
    int32_t safe_step;
    int32_t corrector_only;   /* the reference's `correctoronly` keyword (src/solver.jl:121, 370-374, 945): mu_p = mu in the predictor's residual and no "optimal" termination
                               * (the loop ends on need_dual_feasible / need_primal_feasible, errors or max_iterations only).  clrs_mw_ipm_*; the fp64 loop refuses it */
} clrs_ipm_params;
typedef struct clrs_ipm_record {   /* one row of the reference's iteration table (:566-582) + status */
    int32_t iter, pd_feas, error_code, factor_status, cholesky_status;
    int32_t refine_bits;   /* clrs_mw_ipm_*: bits the FIRST pass of the corrector's refined solve was good to, -log2(max|correction| / max|solution|) over dx and dy
                            * (0: no refinement step ran).  With factors in fewer limbs (clrs_mw_options.factor_limbs) the library returns to all limbs when this
                            * falls below one limb plus a margin */
    double mu, d_obj, p_obj, gap, dual_error, primal_error, alpha_d, alpha_p, beta_c, max_P, max_p, max_d;
} clrs_ipm_record;
int clrs_ipm_create(clrs_ctx *ctx, const clrs_ipm_data *data);
int clrs_ipm_set_params(clrs_ctx *ctx, const clrs_ipm_params *params);
int clrs_ipm_init(clrs_ctx *ctx, double omega_p, double omega_d);            /* x = y = 0, X = omega_p I, Y = omega_d I */
int clrs_ipm_iterate(clrs_ctx *ctx, clrs_ipm_record *out);                   /* one iteration; error_code 0 / 1 / 3 / 4 as docs/src/solving.md:64-70 */
int clrs_ipm_get(clrs_ctx *ctx, double *x, double *y, double *X, double *Y); /* copy the iterate to the host (any pointer may be NULL) */
int clrs_ipm_debug(clrs_ctx *ctx, double out[32]);                           /* diagnostic builds only: kernel phase stamps */

/* Timings of the last assemble/factor calls in seconds, measured with HIP events on the context
 * stream: t[0..4] = schur, cholS, LinvB, Q, cholQ -- the 5-way split compute_T_decomposition!
 * returns (src/solver.jl:1282-1286); t[5] = last solve.  Enabled by clrs_set_timing(ctx, 1). */
int clrs_set_timing(clrs_ctx *ctx, int enabled);
int clrs_get_timings(clrs_ctx *ctx, double t[6]);

/* Algorithmic work of one Schur assembly (SURVEY.md section 8d formulas, with the de-duplicated
 * vector counts): bytes and flops; and of one factor + one solve. */
int clrs_get_counters(const clrs_ctx *ctx, double *assemble_bytes, double *assemble_flops,
                      double *factor_flops, double *solve_flops);

/* Run every later call on the caller's hipStream_t instead of the context's own stream (e.g. the
 * stream a collective library orders itself against).  The caller keeps ownership of the stream. */
int clrs_set_stream(clrs_ctx *ctx, void *hip_stream);

/* Per-kernel timing with HIP events recorded on the context stream around every launch of kernel
 * `kind` (-1: every kind, -2: off; resets the accumulators).  Only in eager (non-graph) mode.
 * clrs_get_kernel_times synchronises, fills seconds[k] / launches[k] for k < max_kinds and returns
 * the number of kinds; clrs_kernel_name(k) names them (the names rocprofv3 --kernel-trace shows). */
int clrs_set_kernel_timing(clrs_ctx *ctx, int kind);
int clrs_get_kernel_times(clrs_ctx *ctx, int max_kinds, double *seconds, int64_t *launches);
const char *clrs_kernel_name(int kind);

/* Process-wide knobs, read when a context is created.  "fused_assemble" (default 1): clusters whose blocks fit
 * in one CU's LDS are assembled by the fused per-cluster kernel; 0 forces the staged grouped-GEMM path.
 * clrs_fused_clusters returns how many clusters of the context the fused kernel takes. */
int clrs_config_set(const char *key, int value);
int clrs_fused_clusters(const clrs_ctx *ctx);
/* "wave_assemble" (default 1): among those, clusters made only of simple rank-1 blocks with n <= 16 and small dense
 * blocks are assembled with one wave per PSD block (k_cluster_assemble_w1); clrs_wave_clusters counts them. */
int clrs_wave_clusters(const clrs_ctx *ctx);
/* "wave2_assemble" (default 1 = automatic; 2 = always; 0 = never): of those, clusters whose low-rank blocks all use the same
 * constraint order (and whose dense blocks are 1 x 1) are assembled by ONE wave per cluster with S in registers.
 * "wave3_assemble" (default 1): with U = P <= 32 that wave is k_cluster_assemble_w3 -- MFMA accumulators reused as operands,
 * nothing but the row of chol(X) and S_j on its way out touches LDS, each wave walks a contiguous run of clusters with the next
 * block's loads in flight -- and "automatic" means always (it is also the fastest form for two clusters); otherwise
 * (U <= 64, or "wave3_assemble" = 0) it is the LDS-staged k_cluster_assemble_w2 and "automatic" means >= 64 clusters.
 * clrs_wave2_clusters counts the clusters either kernel takes. */
int clrs_wave2_clusters(const clrs_ctx *ctx);
/* "wave4_assemble" (default 1): clusters of the same kind (every low-rank block simple, U = P, one constraint order; dense blocks 1 x 1) with a block
 * of 17 .. 32 rows or 33 .. 64 constraints -- out of reach of k_cluster_assemble_w3 -- take k_cluster_assemble_w4 (csrc/clrs_assemble_w4.hip.h: two
 * row tiles, three or four column tiles, L^-1 V by blocks without forming L^-1); 0 leaves them on the general LDS-staged kernel.
 * clrs_wave4_clusters counts them (they are not in clrs_wave_clusters / clrs_wave2_clusters). */
int clrs_wave4_clusters(const clrs_ctx *ctx);
/* "wave5_assemble" (default 1): clusters whose PSD blocks are all 2 x 2 blocks of 16 x 16 sub-blocks carrying E_rs (x) v v^T on the same <= 32 sample vectors in
 * (0,0), (1,1) and the symmetrised off-diagonal pair (the matrix-valued constraints of Nsphere_packing) take k_cluster_assemble_w5
 * (csrc/clrs_assemble_w5.hip.h); clrs_wave5_clusters counts them. */
int clrs_wave5_clusters(const clrs_ctx *ctx);
/* "factor_small" (default 1): a context with ONE cluster (P, N <= 64) runs clrs_schur_factor as one launch of k_factor_small
 * (S_j and B_j staged in one trip, Q never leaves LDS before it is factored); 2 = also for 2-4 clusters, one wave per cluster
 * (slower than the workgroup-per-cluster kernels on the named problems: kept for measurement); 0 = never.
 * "split_blocks" (default 1): when at most 32 clusters take the general fused assembly (k_cluster_assemble), every PSD block
 * of a cluster gets a workgroup of its own (groups of blocks beyond 32 per cluster) that writes its contribution to S_j as a
 * slab; k_sum_S_slabs adds the slabs in block order, which reproduces the one-workgroup accumulation bit for bit; 0 = one
 * workgroup per cluster.
 * "solve_small_max" (default 32768): the one-workgroup solve stage (k_solve_small / k_solve_small2) is used while the operands
 * of a solve (L_j, LinvB, L_Q) stay below this many doubles; beyond, one workgroup per cluster in three launches.
 * "ipm_wmfma" (default 3): the device-resident interior-point loop (clrs_ipm_*) forms, for low-rank blocks that are large enough
 * and whose operands fit in LDS, (bit 0) sum_i a_i A_i as one MFMA contraction over the block's terms and (bit 1) Z V for the
 * per-term pairings as one MFMA product; 0 = per-entry loops.
 * "solve_small2" (default 1): contexts with <= 8 clusters whose factors fit in 150 KB of LDS run the solve stage as ONE launch
 * of k_solve_small2 (every operand staged in one trip to memory, one wave per triangular solve); 0 keeps k_solve_small.
 * "dense_wave" (default 1): dense blocks with n <= 32 beyond the LDS-resident kernel form X^-1 A Y with one wave per matrix
 * (k_trtri32 + k_dense_T32); 0 = two substitution launches and a batched GEMM.
 * "factor_aug" (default 1): a context whose clusters are all beyond one 64-wide block, with 1 .. 512 free variables, factors [S_j .; B_j^T 0] in one
 * blocked factorisation that stops before the corner: L^-1 B comes out as the panels of the appended block row, -Q as its Schur
 * complement (k_chol_pack, k_chol_level, k_chol_unpack); 0 = Cholesky of S, the substitution for L^-1 B and the Gram product apart.
 * "pairing_tri" (default 1): staged low-rank blocks whose left and right vectors coincide (W = V, one sub-block) compute only the
 * lower tiles of the symmetric pairing matrices V^T X^-1 V and V^T Y V, and staged dense blocks only the lower tiles of
 * <A_i, X^-1 A_k Y>; k_schur_gather mirrors its reads; 0 = full matrices.
 * "potrf_levels" (default 1): the staged Cholesky of a matrix beyond one 64-wide block runs ONE launch per block column
 * (k_chol_level); 0 = k_potrf_diag + k_trsm_diag + k_gemm_f64_t per block column.
 * "trsm_blockinv" (default 1): staged triangular solves with n > 512 go through inverted 512 x 512 diagonal blocks (k_trtri_diag
 * and GEMMs, n / 512 levels); 0 = 64-wide substitutions (n / 64 levels of k_trsm_diag + k_gemm_f64_t). */
/* Diagnostic builds (-DCLRS_FUSED_STAMPS) only: s_memtime stamps of the phases of workgroup 0 of the fused kernel. */
int clrs_debug_stamps(clrs_ctx *ctx, uint64_t out[64]);

/* Capture the per-iteration launch sequences into hipGraphs (1) or launch kernels one by one (0).  The default (null) stream cannot
 * be captured: with a caller's stream (clrs_set_stream) graph mode needs a stream of its own; calls fail with CLRS_ERR_STATE otherwise. */
int clrs_set_graph_mode(clrs_ctx *ctx, int enabled);

/* Name of the kernel that dominates the assembly for this context and the number of launches of one
 * assemble / factor / solve call (for profiling). */
int clrs_plan_info(const clrs_ctx *ctx, int32_t *n_launch_assemble, int32_t *n_launch_factor, int32_t *n_launch_solve);


/* =====================================================================================================================
 * The same path at the reference's working precision: multi-word fp64 ("limbs").
 *
 * The reference computes in Arb midpoints at `prec` bits (default 256: src/solver.jl:73,103; Cholesky on midpoints
 * src/tools.jl:59-107; products Arblib.approx_mul!, src/solver.jl:1125-1143) because the Schur complements of its
 * headline problems are not positive definite to fp64 accuracy (cohnelkies(8,15) fails at the first iterate in fp64 and
 * at 113 bits).  A clrs_mw_ctx runs the identical stages with every number an unevaluated sum of
 * `limbs` doubles (limbs = 2..6, 8, 10: ~104 / 157 / 209 / 262 / 315 / 420 / 525 bits); limbs = 5 covers the reference's default precision
 * (256), 6 its test at prec = 300, 10 the prec = 512 of its tutorial and example generators.
 *
 * Array format ("planar limbs"): an array of logical length len is limbs * len doubles, limb l of element i at
 * [l * len + i]; value_i = sum_l a[l * len + i], limb 0 the value rounded to fp64, |limb l+1| <= ulp(limb l).
 * Every layout of the fp64 API (xy layout, S layout, x-like vectors) is used unchanged for the logical array.
 * The problem description is the same clrs_sdp_desc (fp64 data: the sampled vectors, lambda, B, dense A_p are exact
 * doubles; the iterates and everything computed from them carry `limbs` words).
 * Return codes and failure semantics are those of the fp64 entry points they mirror.
 * ===================================================================================================================== */
typedef struct clrs_mw_ctx clrs_mw_ctx;

/* Replaces precompute_matrices_bilinear_pairings (src/solver.jl:985-1059) + the preallocation :298-317 for a solve at
 * `limbs` words per number. */
int clrs_mw_create(const clrs_sdp_desc *desc, int device, int limbs, clrs_mw_ctx **out);
/* The same with problem data beyond fp64: every `const double *` array of `desc` (B, term_lambda, term_vs, term_ws, dense_A)
 * is planar with `data_limbs` planes (1 or 2; plane length = the array's fp64 length).  The reference holds the sampled
 * problem at `prec` bits as well (convert_to_prec, src/interface.jl:1078-1112), and its headline problems need it: the
 * cohnelkies(8,15) SDP whose data are rounded to fp64 is a different, dual-infeasible problem. */
int clrs_mw_create_ex(const clrs_sdp_desc *desc, int data_limbs, int device, int limbs, clrs_mw_ctx **out);
/* The same with per-context choices instead of the process-wide clrs_config_set knobs (a field < 0, or opts == NULL: the knob's value):
 * exact_products 0 / 1 / 2 = pairing matrices through exact slice products on the matrix cores never / from 256 eligible blocks on / always;
 * refine 0 / 1 / 2 = iterative refinement of the solve stage off / one step (default) / one step with the correction in fewer limbs. */
typedef struct clrs_mw_options {
    int32_t exact_products;
    int32_t refine;
    int32_t pipeline;        /* factorisations of small matrices as a pipeline of workgroups: 0 never / 1 (default) the clusters' S_j of at most 32 rows, of 48 .. 64
                              * rows (the 64-row form; clrs_config_set("mw_pipeline64", 0) switches that form off) and Q from 9 rows on / 2 every S_j and Q of at
                              * most 64 / 32 rows, and at 8 and 10 limbs as well */
    int32_t refine_predictor; /* clrs_mw_ipm_*: 0 (default) the predictor's solve is one pass of products, the corrector's is refined; 1 both are refined */
    int32_t factor_limbs;    /* Mixed-precision iterative refinement: limbs of the FACTOR stage (L_j, L_j^-1, L^-1 B, Q, L_Q, L_Q^-1) and of the inverse-factor products of
                              * the solve stage; the residuals of the refinement step, S_j itself and the solution carry all `limbs`.  0 (default) = automatic: inside
                              * clrs_mw_ipm_* limbs - 1 for limbs = 5, 6 on the LDS-resident paths, while the measured first-pass accuracy (clrs_ipm_record.refine_bits)
                              * stays above one limb plus a margin, then all limbs; the stand-alone entry points (clrs_mw_schur_factor / _solve) use all limbs.
                              * `limbs` = never reduce; limbs - 1 (5, 6 limbs) = the reduced count in the stand-alone entry points as well */
    int32_t matmul_limbs;    /* The reference's `matmul_prec` keyword (src/solver.jl:125, 304, 312-313, 1125-1143): limbs of the products that form the pairing matrices
                              * (part_r = Y V, X^-1 V and bilinear_pairings = W^T part_r; here T = Y V, Z = chol(X)^-1 V, Z^T Z, V^T T) -- S_j is accumulated from them in all
                              * `limbs`, as the reference does at `prec`.  0 (default) = `limbs`; a smaller count is rounded up to the next on offer (from limbs / 2 up).
                              * With fewer limbs than `limbs` the exact slice products (exact_products) are off. */
    int32_t reserved[2];
} clrs_mw_options;
int clrs_mw_create_opts(const clrs_sdp_desc *desc, int data_limbs, int device, int limbs, const clrs_mw_options *opts, clrs_mw_ctx **out);
void clrs_mw_destroy(clrs_mw_ctx *ctx);
int clrs_mw_limbs(const clrs_mw_ctx *ctx);
int clrs_mw_get_dims(const clrs_mw_ctx *ctx, clrs_dims *dims);          /* logical lengths; dims->reserved = limbs */
/* number of unique sampled vectors of block b over all its sub-blocks (the union of rightvecs / leftvecs) */
int clrs_mw_get_unique_count(const clrs_mw_ctx *ctx, int32_t block, int32_t *n_unique);

/* approx_cholesky!(X_inv_blk, X_blk) for every block (src/solver.jl:388-399); 0 or b+1 */
int clrs_mw_cholesky_blocks(clrs_mw_ctx *ctx, const double *X, double *Xchol);
/* compute_S_integrated! (src/solver.jl:1062-1226) */
int clrs_mw_schur_assemble(clrs_mw_ctx *ctx, const double *Xchol, const double *Y, double *S_out, double *AY_out);
/* steps 3-4 of compute_T_decomposition! (src/solver.jl:1244-1279); 0, j+1 or n_clusters+1 */
int clrs_mw_schur_factor(clrs_mw_ctx *ctx);
int clrs_mw_get_factor(clrs_mw_ctx *ctx, double *L, double *LinvB, double *LQ);
int clrs_mw_debug_exact_stamps(clrs_mw_ctx *ctx, unsigned long long *out /* [16] */);   /* diagnostic: first call arms, later calls read the phase stamps of k_mws_pair */
int clrs_mw_debug_pipe_stamps(clrs_mw_ctx *ctx, unsigned long long *out /* [16 * 40] */);   /* diagnostic builds (-DCLRS_MW_STAMPS) only: DESIGN.md section 5.5 */
int clrs_mw_get_S(clrs_mw_ctx *ctx, double *S_out, double *AY_out);   /* S_j and A_Y of the last (device-pointer) assembly -> host; NULL pointers are skipped */
/* the solve stage of compute_search_direction! (src/solver.jl:1527-1582) */
int clrs_mw_schur_solve(clrs_mw_ctx *ctx, const double *rhs_x, const double *rhs_y, double *dx, double *dy);

/* device-pointer variants: planar device arrays, enqueue on the context stream, no synchronisation.
 * clrs_mw_schur_assemble_dev uses the reciprocal Cholesky diagonals the last clrs_mw_cholesky_blocks_dev left in the
 * context; a caller that brings factors of X from elsewhere calls clrs_mw_set_xchol_dev first. */
int clrs_mw_cholesky_blocks_dev(clrs_mw_ctx *ctx, const double *d_X, double *d_Xchol);
int clrs_mw_sync_status_cholesky(clrs_mw_ctx *ctx);
int clrs_mw_set_xchol_dev(clrs_mw_ctx *ctx, const double *d_Xchol);
int clrs_mw_schur_assemble_dev(clrs_mw_ctx *ctx, const double *d_Xchol, const double *d_Y);
int clrs_mw_schur_factor_dev(clrs_mw_ctx *ctx);
int clrs_mw_sync_status(clrs_mw_ctx *ctx);
int clrs_mw_schur_solve_dev(clrs_mw_ctx *ctx, const double *d_rhs_x, const double *d_rhs_y, double *d_dx, double *d_dy);
double *clrs_mw_S_buffer_dev(clrs_mw_ctx *ctx);      /* planar S layout: S after assemble, L_j after factor */
double *clrs_mw_AY_buffer_dev(clrs_mw_ctx *ctx);     /* planar [T] */
void *clrs_mw_stream(clrs_mw_ctx *ctx);
int clrs_mw_set_stream(clrs_mw_ctx *ctx, void *hip_stream);

/* Cluster sharding (one process per GPU): a context created from the description of ITS clusters only (all N
 * free variables) is told its place with clrs_mw_set_shard.  The two sums over all clusters, Q = sum_j LinvB_j^T LinvB_j
 * (src/solver.jl:1268-1269) and u = sum_j LinvB_j^T t_j (:1550-1553), are formed from per-rank partial sums that are ALL-GATHERED
 * (limb planes cannot be all-reduced: a sum of limbs is not the limbs of the sum) and added in rank order by every rank, so all
 * ranks hold bit-identical Q, L_Q, dy.  Either the caller moves the slots (split-phase entry points; any transport), or the
 * library does it with RCCL on the context stream after clrs_mw_comm_init, in which case clrs_mw_schur_factor_dev and
 * clrs_mw_schur_solve_dev are collective calls: one all-gather of limbs*N*N doubles per factorisation, one of limbs*N per solve. */
int clrs_comm_unique_id(void *id128);                                     /* ncclGetUniqueId: 128 bytes, to be distributed by the caller */
int clrs_mw_set_shard(clrs_mw_ctx *ctx, int rank, int world);
int clrs_mw_comm_init(clrs_mw_ctx *ctx, const void *id128, int rank, int world);   /* clrs_mw_set_shard + ncclCommInitRank */
int clrs_mw_comm_destroy(clrs_mw_ctx *ctx);
/* Diagnostic (collective over the communicator): microseconds per all-gather of the three message sizes of one sharded iteration -- us[0] partial Q,
 * us[1] partial u, us[2] one scalar record -- from HIP events around `reps` exchanges back to back on the context's stream; info = rank, world, backend
 * (0 none, 1 RCCL, 2 in-process group).  bench.py reports them beside the N-GPU rate. */
int clrs_mw_comm_probe(clrs_mw_ctx *ctx, int reps, double us[3], int info[3]);
/* The interior-point iteration of a sharded context (clrs_mw_ipm_*) exchanges on two streams: a second communicator (a second unique
 * id, same rank and world) serves its side stream. */
int clrs_mw_comm_init_side(clrs_mw_ctx *ctx, const void *id128);
/* In-process stand-in for the communicators: `world` contexts of ONE process on ONE device, each driven by its own host thread (a
 * collective call blocks until every rank of the group has issued it).  For tests on a single GPU (RCCL refuses two ranks on one
 * device); the exchange is the same all-gather into rank-ordered slots. */
typedef struct clrs_mw_local_group clrs_mw_local_group;
int clrs_mw_local_group_create(int world, int device, clrs_mw_local_group **out);
void clrs_mw_local_group_destroy(clrs_mw_local_group *group);
int clrs_mw_comm_init_local(clrs_mw_ctx *ctx, clrs_mw_local_group *group, int rank);
int clrs_mw_schur_factor_local_dev(clrs_mw_ctx *ctx);                     /* L_j, LinvB_j, partial Q into slot `rank` */
double *clrs_mw_q_gather_dev(clrs_mw_ctx *ctx);                           /* [world][limbs * N * N] */
int clrs_mw_schur_factor_finish_dev(clrs_mw_ctx *ctx);                    /* Q = sum of the slots, Cholesky of Q */
int clrs_mw_schur_solve_fwd_dev(clrs_mw_ctx *ctx, const double *d_rhs_x);  /* t = L^-1 rhs_x, partial u into slot `rank` */
double *clrs_mw_u_gather_dev(clrs_mw_ctx *ctx);                           /* [world][limbs * N] */
int clrs_mw_schur_solve_bwd_dev(clrs_mw_ctx *ctx, const double *d_rhs_y, double *d_dx, double *d_dy);
/* The solve stage multiplies with explicit inverse factors and then takes one step of iterative refinement against the assembled S_j and B,
 * which gives (dx, dy) the backward error of the reference's substitutions (src/solver.jl:1538, 1557, 1567-1572) -- clrs_mw_schur_solve[_dev] do
 * both (clrs_mw_options.refine, or clrs_config_set("mw_refine", 0 / 1 / 2) before the context is created: no step / one step, the default / one step
 * with the correction in fewer limbs -- cheaper, and as good while twice the bits the products lose fit in those limbs: DESIGN.md section 5.5).
 * Split-phase callers: clrs_mw_schur_solve_bwd_dev leaves this rank's partial u' of the correction in slot `rank` of the u gather buffer; after ONE
 * MORE exchange of that buffer this call adds the correction to dx, dy (without it they are the plain products' solution). */
int clrs_mw_schur_solve_refine_dev(clrs_mw_ctx *ctx, const double *d_rhs_y, double *d_dx, double *d_dy);

/* HIP-event timings of the last calls, seconds: t[0] schur, t[1] cholS + LinvB (one kernel), t[2] 0, t[3] Q, t[4] cholQ,
 * t[5] last solve (the split compute_T_decomposition! returns, src/solver.jl:1282-1286). */
int clrs_mw_set_timing(clrs_mw_ctx *ctx, int enabled);
int clrs_mw_get_timings(clrs_mw_ctx *ctx, double t[6]);
/* algorithmic multi-word multiply-adds of one assembly / factorisation / solve */
int clrs_mw_get_counters(const clrs_mw_ctx *ctx, double *assemble_muladds, double *factor_muladds, double *solve_muladds);

/* The interior-point iteration around the path at the same precision (the loop body of solvesdp, src/solver.jl:348-589, as
 * clrs_ipm_* above but with every iterate and intermediate in `limbs` words; the step-length eigenvalue alone is taken in
 * fp64, as the reference's Float64 Lanczos does, src/solver.jl:1659).  C, c, b are fp64; x, y, X, Y of clrs_mw_ipm_get / _set
 * are planar limbs.  clrs_mw_ipm_create may be called again on the same context with other objective data.
 * Available when every PSD block fits in LDS at this limb count. */
int clrs_mw_ipm_create(clrs_mw_ctx *ctx, const clrs_ipm_data *data);
int clrs_mw_ipm_create_ex(clrs_mw_ctx *ctx, const clrs_ipm_data *data, int data_limbs);   /* C, c, b planar with data_limbs planes */
int clrs_mw_ipm_set_params(clrs_mw_ctx *ctx, const clrs_ipm_params *params);
int clrs_mw_ipm_init(clrs_mw_ctx *ctx, double omega_p, double omega_d);
int clrs_mw_ipm_set(clrs_mw_ctx *ctx, const double *x, const double *y, const double *X, const double *Y);   /* warm start, src/solver.jl:202-239 */
int clrs_mw_ipm_iterate(clrs_mw_ctx *ctx, clrs_ipm_record *out);
int clrs_mw_ipm_get(clrs_mw_ctx *ctx, double *x, double *y, double *X, double *Y);
int clrs_mw_ipm_objectives(clrs_mw_ctx *ctx, double *out /* [3 * limbs]: d_obj, p_obj, gap */);
/* Cluster-sharded iteration (one context per rank, each created from its own clusters and ALL free variables): the scalars that are sums,
 * maxima or minima over all clusters -- mu (src/solver.jl:369), the errors (:441-442), p = +-b - B^T x (:899-916), beta_c (:429), the step
 * lengths (:1684-1686), the objectives (:793-804) -- travel as one small record per rank and stage through all-gathers the library issues
 * itself and are reduced in rank order by every rank: x, X, Y stay sharded, y and every scalar are bit-identical on all ranks.  This call
 * supplies what a rank cannot know: the row count of X over all clusters, the cluster count of the whole problem, and the global numbers
 * (0-based) of its clusters / PSD blocks for failure codes (NULL: local numbers). */
int clrs_mw_ipm_set_global(clrs_mw_ctx *ctx, int rows_of_X_global, int clusters_global, const int32_t *cluster_ids, const int32_t *block_ids);
/* The whole loop with its termination test (src/solver.jl:348-589, 921-950) in one call: at most max_iterations iterations from the
 * current iterate, enqueued one ahead of the record the host waits for (the device evaluates the same test and freezes the iterate once
 * it holds, so the device never idles between iterations).  records[i] (i < max_records) = table row of the i-th iteration of this call;
 * *n_iter = iterations run; *error_code = 0, 1 / 3 / 4 as clrs_mw_ipm_iterate reports them, or 2 = max_iterations reached (:362-366). */
typedef struct clrs_ipm_stop {
    double duality_gap_threshold;                 /* with the error thresholds of clrs_ipm_params */
    int32_t need_dual_feasible, need_primal_feasible, max_iterations, reserved;
} clrs_ipm_stop;
int clrs_mw_ipm_solve(clrs_mw_ctx *ctx, const clrs_ipm_stop *stop, clrs_ipm_record *records, int max_records, int *n_iter, int *error_code);
/* The same loop with a callback per record: `on_record(record, user)` runs on the calling thread as the host reads the record of an iteration (the device
 * is one iteration further by then) -- the live iteration table of `verbose = true` (src/solver.jl:566-582) without giving up the one-call loop.  The
 * callback must not call into this context.  on_record and records may be NULL. */
typedef void (*clrs_ipm_record_fn)(const clrs_ipm_record *record, void *user);
int clrs_mw_ipm_solve_cb(clrs_mw_ctx *ctx, const clrs_ipm_stop *stop, clrs_ipm_record_fn on_record, void *user, clrs_ipm_record *records, int max_records, int *n_iter, int *error_code);
/* error_code of clrs_mw_ipm_iterate / clrs_mw_ipm_solve: 0; 1 / 3 / 4 as the reference's (docs/src/solving.md:64-70); 2 = max_iterations; and 5 = a wait
 * between the two streams of the iteration ran past its wall-clock bound (30 s: a hang, e.g. under a profiler that serialises kernels -- there, or to rule
 * it out, clrs_config_set("mw_stream_words", 0) / CLRS_MW_STREAM_WORDS=0 makes new contexts synchronise through events only); the iterate is not moved. */

const char *clrs_strerror(int code);
const char *clrs_last_error(void);
void clrs_set_last_error(const char *msg);   /* used by the library's own translation units */
const char *clrs_version(void);

/* Test hook: C = alpha * op(A) op(B) + beta * C through the grouped fp64 MFMA GEMM kernel
 * (host pointers).  Used by tests/ to check the kernel in isolation. */
int clrs_test_gemm(int device, int ta, int tb, int M, int N, int K, double alpha, const double *A, int lda,
                   const double *B, int ldb, double beta, double *C, int ldc);
/* Test hooks for the blocked dense kernels (host pointers, in place): lower Cholesky of an n x n
 * matrix (returns 0 or 1 on a non-positive pivot), and B <- L^-1 B / L^-T B. */
int clrs_test_potrf(int device, int n, double *A, int lda);
int clrs_test_trsm(int device, int trans, int n, int nrhs, const double *L, int ldl, double *B, int ldb);

/* Measurement hook: average duration (us, HIP events, `reps` back-to-back launches) of a pure streaming kernel that reads
 * `read_bytes` (two streams) and writes `write_bytes` (one stream, non-temporal) of freshly allocated device memory: the rate a
 * bandwidth-bound kernel of that footprint and read : write mix can reach on this device.  bench.py reports the assembly kernel's
 * HBM traffic per second beside it (`roofline.copy_roof`). */
int clrs_test_stream(int device, long long read_bytes, long long write_bytes, int reps, double *avg_us);

#ifdef __cplusplus
}
#endif
#endif /* CLRS_HIP_H */
