// clrs_hip.hip -- host side of the C ABI (include/clrs_hip.h): context creation (vector
// de-duplication, gather tables, launch plan), and the per-iteration drivers that replay the plan.
//
// Design (MI355X-first, not a translation of the reference's j -> l -> r -> s serial walk):
//   * every step of the path is ONE grouped launch over all PSD blocks / clusters, driven by
//     device-resident descriptor tables that are built once at context creation;
//   * blocked Cholesky / triangular solves are level-synchronous across matrices: panel k of every
//     matrix is processed by the same launch;
//   * the explicit inverse of X (reference "method 3", src/solver.jl:1107-1119) is avoided:
//     V^T X^-1 V = (L^-1 V)^T (L^-1 V), one triangular solve + one MFMA contraction;
//   * each S entry is produced by exactly one thread (deterministic, no atomics);
//   * the launch sequences are static, so they can be captured into hipGraphs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/clrs_hip.h"
#include "clrs_kernels.hip.h"
#include "clrs_fused.hip.h"
#include "clrs_assemble_w3.hip.h"
#include "clrs_assemble_w4.hip.h"
#include "clrs_assemble_w5.hip.h"
#include "clrs_solve_small.hip.h"
#include "clrs_factor_small.hip.h"
#include "clrs_ipm.hip.h"

using namespace clrs;
typedef long long i64;

static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
extern "C" void clrs_set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }   // for the other translation units of the library (clrs_mw.hip)
#define HIPCHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(CLRS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

static const int INFO_NONE = 0x7f7f7f7f;

// process-wide knobs read at context creation (clrs_config_set)
static int g_cfg_fused_assemble = 1;
static int g_cfg_split_blocks = 1;      // fused general assembly: one workgroup per PSD block (partial S_j slabs, summed in block order) when the launch is far from filling the chip
static int g_cfg_fused_factor = 1;
static int g_cfg_wave_assemble = 1;
static int g_cfg_wave2_assemble = 1;
static int g_cfg_wave5_assemble = 1;   // k_cluster_assemble_w5: the register-resident form for 2 x 2 blocks of 16 x 16 sub-blocks with shared sample vectors (Nsphere_packing); 0: the general kernels
static int g_cfg_wave4_assemble = 1;   // k_cluster_assemble_w4: the register-resident form for simple blocks of up to 32 rows and up to 64 constraints; 0: the general kernels
static int g_cfg_wave3_assemble = 1;   // register-resident form of the cluster-per-wave assembly (U <= 32); 0: k_cluster_assemble_w2
static int g_cfg_dense_block = 1;
static int g_cfg_trsm_blockinv = 1;  // triangular solves with n > TRSM_IB through inverted diagonal blocks (plan_trsm_blockinv)
static int g_cfg_potrf_levels = 1;   // Cholesky of matrices beyond one block: one launch per block column (plan_potrf_levels)
static int g_cfg_pairing_tri = 1;    // staged low-rank blocks with W = V: lower triangles of the pairing matrices only
static int g_cfg_factor_aug = 1;     // clusters beyond one block, <= 512 free variables: L_j, L_j^-1 B_j and Q from one factorisation of [S_j .; B_j^T 0] each
static int g_cfg_dense_wave = 1;      // dense blocks with n <= 32 beyond k_dense_block: one wave per (block, matrix), k_dense_T32
static int g_cfg_factor_small = 1;      // factor + Q in one launch of one workgroup for <= 4 small clusters (0: k_cluster_factor + k_small_potrf)
static int g_cfg_solve_small_max = 32768;     // one-workgroup solve stage only up to this many doubles of operands (beyond: one workgroup per cluster, three launches)
static int g_cfg_ipm_wmfma = 3;         // device interior-point loop, bit 0: sum_i a_i A_i of a low-rank block as an MFMA contraction over its terms; bit 1: Z V by MFMA (0: per-entry loops)
int g_cfg_mw_stream_words = 1;        // multi-word interior-point iteration: its two streams synchronise through words that kernels store and await (1) or through events only (0: for counter-collection runs, whose profiler serialises kernels -- a kernel that waits for one that cannot start sits in its bounded polls); env CLRS_MW_STREAM_WORDS overrides
int g_cfg_mw_pipeline64 = 1;          // multi-word path: clusters of 33 .. 64 rows through the 64-row form of the pipelined factorisation (k_mw_factor_pipe64); 0: the one-workgroup kernels
int g_cfg_mw_zt_small_maxn = 64;       // multi-word path: T = Y V, Z = chol(X)^-1 V with two columns per workgroup and eight lanes per entry (instead of eight columns, two lanes) for blocks of at most this many rows
int g_cfg_mw_pipeline_x = 1;           // multi-word path: the Cholesky of the X (and Y) blocks through the pipelines of workgroups (k_mw_potrf_x_pipe) where every block carries its inverse factor; 0: k_mw_potrf_x
int g_cfg_mw_pipeline_x_min = 24;      // ... from this many rows of the largest block on (pipeline = 2: always)
int g_cfg_mw_sharded_factor_limbs = 1; // multi-word loop, sharded contexts: the reduced factor limbs of the mixed-precision refinement there too (the measured first-pass accuracy is gathered with the step lengths); 0: all limbs
int g_cfg_mw_pipeline = 1;            // multi-word path: Cholesky + inverse factor of matrices <= 32 rows as a pipeline of workgroups (clrs_mw_pipe.hip.h); 0: one workgroup chain per matrix
int g_cfg_mw_refine_predictor = 0;    // multi-word interior-point iteration: 1 = the PREDICTOR's solve takes the refinement step too (default: the corrector's only -- the predictor's direction sets beta_c and the second-order term, nothing that moves the iterate)
int g_cfg_mw_refine = 1;              // multi-word path: one step of iterative refinement of the solve stage (k_mw_refine); 0: products with the inverse factors only; 2: the correction in fewer limbs
int g_cfg_mw_affine_corrector = 1;    // multi-word interior-point iteration: the corrector's right-hand side as rhs0 + mu_c tau, with beta_c / mu_c formed on the side stream beside Z0 and its traces (clrs_mw_ipm_host.inc); 0: the corrector waits for mu_c
int g_cfg_mw_factor_limbs = 0;        // multi-word path: limbs of the factor stage and of the solve stage's products (mixed-precision refinement, clrs_mw_kernels.hip.h::mw_kf_of): 0 = automatic (reduced inside clrs_mw_ipm_* while the measured contraction allows, all limbs in the stand-alone entry points); a limb count = that count everywhere
int g_cfg_mw_exact_products = 1;      // multi-word path: pairing matrices through exact slice products on the matrix cores (k_mws_pair): 0 never, 1 when >= 256 blocks are eligible (one per compute unit: below, the chip is not full and the latency-oriented kernels win), 2 always
static int g_cfg_solve_small2 = 1;     // one-workgroup solve stage with all loads up front and single-wave triangular solves (0: k_solve_small)
static const int LDS_BUDGET_DOUBLES = 20000;   // of the 20480 doubles (160 KiB) a workgroup may claim

// ------------------------------------------------------------------------------------------------
// launch plan
// ------------------------------------------------------------------------------------------------
enum StepKind { STEP_MEMCPY, STEP_GEMM, STEP_TRSM, STEP_POTRF, STEP_GATHER_S, STEP_GATHER_SCALAR, STEP_SUB, STEP_MEMSET_INFO, STEP_ZERO_UPPER, STEP_FUSED_ASSEMBLE, STEP_SMALL_POTRF, STEP_CLUSTER_FACTOR, STEP_GEMV_T, STEP_CSOLVE_FWD, STEP_Q_SOLVE, STEP_CSOLVE_BWD, STEP_SUM_SLABS, STEP_ASSEMBLE_W1, STEP_GRAM_SMALL, STEP_ASSEMBLE_W2, STEP_SOLVE_SMALL, STEP_DENSE_BLOCK, STEP_ASSEMBLE_W3, STEP_FACTOR_SMALL, STEP_SUM_S_SLABS, STEP_TRTRI32, STEP_DENSE_T32, STEP_TRTRI_DIAG, STEP_COPY2D, STEP_CHOL_LEVEL, STEP_CHOL_PACK, STEP_CHOL_UNPACK, STEP_ASSEMBLE_W4, STEP_ASSEMBLE_W5, STEP_NKINDS };
static const char *const STEP_NAMES[STEP_NKINDS] = {"hipMemcpyAsync(D2D)", "k_gemm_f64_t", "k_trsm_diag", "k_potrf_diag", "k_schur_gather",
                                                    "k_gather_scalar", "k_sub", "hipMemsetAsync", "k_zero_upper", "k_cluster_assemble", "k_small_potrf", "k_cluster_factor", "k_gemv_t",
                                                    "k_cluster_solve_fwd", "k_q_solve", "k_cluster_solve_bwd", "k_sum_slabs", "k_cluster_assemble_w1", "k_gram_small", "k_cluster_assemble_w2", "k_solve_small", "k_dense_block", "k_cluster_assemble_w3", "k_factor_small", "k_sum_S_slabs", "k_trtri32", "k_dense_T32", "k_trtri_diag", "k_copy2d", "k_chol_level", "k_chol_pack", "k_chol_unpack", "k_cluster_assemble_w4", "k_cluster_assemble_w5"};
static const int KT_MAX_EVENTS = 8192;   // event pairs kept between two clrs_get_kernel_times calls

struct Step {
    StepKind kind;
    int grid = 0;
    void *d0 = nullptr, *d1 = nullptr, *d2 = nullptr;  // device descriptor tables
    void *dst = nullptr;
    const void *src = nullptr;
    size_t bytes = 0;
    i64 n = 0;
    int nmax = 0;
    int aux0 = 0, aux1 = 0;   // kind-specific (STEP_ASSEMBLE_W3: grid size, number of blocks)
    bool s_inverse = false;   // forms the inverted diagonal blocks of the factors L_j of S: skipped while those of the current factorisation exist
};

struct Plan {
    std::vector<Step> steps;
    hipGraphExec_t graph = nullptr;
    const void *captured[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // bindings baked into `graph`
    int launches() const { return (int)steps.size(); }
};

struct DeviceArena {  // bump allocator over one hipMalloc (static tables + work buffers)
    char *base = nullptr;
    size_t size = 0, used = 0;
};

struct TrsmJob {
    const double *L;
    int ldl, n;
    double *B;
    int ldb, nrhs;
};
struct PotrfJob {
    double *A;
    int lda, n, code;
    int stop = 0;     // > 0: factor the first `stop` block columns only (the trailing updates still reach the rest: a Schur complement)
};

struct BlockInfo {
    int j = 0, m = 1, delta = 1, n = 1, kind = 0;
    i64 xyoff = 0, t0 = 0, t1 = 0, d0 = 0, d1 = 0;
    std::vector<int> UR, UL, offR, offL;
    int URt = 0, ULt = 0;
    bool sym = false;
    bool sd_tri = false;      // dense block: only the lower tiles of the symmetric Sd = <A_i, X^-1 A_k Y> are formed (staged Gram GEMM)
    // offsets (in doubles) inside the work / static arenas
    i64 zr_off = -1, zl_off = -1, ty_off = -1, g_off = -1;  // ZR/ZL in the "solve" arena; TY; GX,GY
    int cnt = 0;
    i64 w_off = -1, tt_off = -1, sd_off = -1;
    int *d_tptr = nullptr;   // low rank: [P+1] CSR over the cluster's constraints into the sorted term arrays; dense: [cnt] constraint index
    bool fused = false;      // handled by k_cluster_assemble
};

struct IpmState;

struct clrs_ctx {
    int device = 0;
    int *h_info = nullptr;                             // pinned host copy of the two status words
    // pinned staging arena of the host-pointer entry points: a copy to or from pageable memory goes through the runtime's own
    // staging with a synchronisation per call (10-20 us for a few KB); one memcpy into pinned memory and an async copy do not
    char *pin = nullptr;
    size_t pin_cap = 0, pin_used = 0, pin_demand = 0;
    struct PinOut { void *user; const void *pinned; size_t bytes; };
    std::vector<PinOut> pin_out;
    IpmState *ipm = nullptr;                           // device-resident interior-point iteration (clrs_ipm_*), created on demand
    std::vector<int> h_term_p, h_dense_p;              // host copies of the description arrays the IPM tables are built from
    std::vector<double> h_term_lambda;
    int *d_ayL = nullptr, *d_ayR = nullptr;            // per original term: left / right expanded vector index of its pairing
    hipStream_t stream = nullptr;
    bool own_stream = true;
    // per-kernel HIP-event timing (clrs_set_kernel_timing): -2 off, -1 every step kind, k >= 0 only kind k
    int kt_kind = -2;
    std::vector<hipEvent_t> kt_ev;
    std::vector<int> kt_kinds;
    double kt_total[STEP_NKINDS] = {0};
    long long kt_count[STEP_NKINDS] = {0};
    int J = 0, N = 0, NB = 0;
    i64 T = 0, D = 0;
    std::vector<int> P;
    std::vector<i64> coff, Soff;
    i64 xylen = 0, xlen = 0, Slen = 0;
    std::vector<BlockInfo> blk;
    std::vector<void *> allocs;  // every hipMalloc, for destroy
    // device buffers
    std::vector<CholLevelJob> chol_jobs;           // jobs of the single-matrix k_chol_level launches (passed by value)
    std::map<const double *, std::pair<double *, double *>> s_inv;   // factor L_j of S -> its inverted diagonal blocks and their scratch (shared by the solve plans)
    std::vector<std::pair<i64, double *>> trsm_y_pool;               // scratch copies of the right-hand sides of plan_trsm_blockinv, by job position: the plans of a
                                                                    // context run one after the other on one stream, so the forward and the backward solve of a triangle share them
    bool s_inv_valid = false;                      // ... formed since the last factorisation (eager mode skips forming them again)
    std::vector<CholAugDesc> chol_aug;             // augmented factorisations [S; B^T] (k_chol_pack / k_chol_unpack, passed by value)
    double *d_Xc = nullptr, *d_Y = nullptr;        // inputs (xy layout)
    double *d_static = nullptr, *d_work = nullptr; // [Vexp | Astack] and [Z | W] (identical layouts: one memcpy)
    i64 solve_arena_len = 0;
    double *d_TY = nullptr, *d_G = nullptr, *d_TT = nullptr, *d_Sd = nullptr;
    double *d_S = nullptr, *d_B = nullptr, *d_LB = nullptr, *d_Q = nullptr;
    double *d_t = nullptr, *d_u = nullptr, *d_dy = nullptr, *d_rhsy = nullptr, *d_dx = nullptr, *d_rhsx = nullptr;
    double *d_AY = nullptr;
    i64 *d_ayidx = nullptr;
    int *d_info = nullptr;
    double *d_X = nullptr;  // scratch for clrs_cholesky_blocks
    Plan p_assemble, p_cholS, p_linvB, p_Q, p_cholQ, p_fwd, p_bwd, p_cholX;
    FTables ftables = {};
    W3Tables w3tables = {};
    W4Tables w4tables = {};
    W5Tables w5tables = {};
    CSolve8 solve8 = {};          // host copy of the (<= 8) CSolve descriptors: kernel argument of k_solve_small2
    struct SolveSmall2 {          // its staging job table, per phase: 0 = whole stage, 1 / 2 = before / after the exchange of u (sharded path)
        StageJobs jobs = {};
        int rx_job[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ry_job[2] = {-1, -1};     // [first, end) job ranges of rhs_x[j] / rhs_y
        bool ok = false;
    } ss2[3];
    Plan p_fwd_small, p_bwd_small;     // one-launch forms of p_fwd / p_bwd (k_solve_small2, phases 1 and 2)
    Plan p_factor_small;               // k_factor_small: the whole of clrs_schur_factor in one launch
    FSmallArgs fsmall = {};
    StageJobs fsmall_jobs = {};
    // caller-provided device pointers of the current call, read by the fused kernels at launch (no staging copies)
    const double *bind_X = nullptr, *bind_rhsx = nullptr, *bind_rhsy = nullptr;
    double *bind_Xchol = nullptr, *bind_dx = nullptr, *bind_dy = nullptr;
    double *d_dinvS = nullptr, *d_dinvQ = nullptr, *d_Qslabs = nullptr;
    bool q_slabs = false;     // k_cluster_factor leaves per-cluster partial Q slabs
    Plan p_cholQ_slabs, p_solve_all;
    bool fused_fs = false, fused_q = false, fused_x = false, all_assemble_fused = false;
    bool factored = false, assembled = false, local_factored = false;
    bool timing = false, graph_mode = false;
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double times[6] = {0, 0, 0, 0, 0, 0};
    bool times_pending = false, solve_time_pending = false;
    double cnt_bytes = 0, cnt_flops = 0, cnt_factor_flops = 0, cnt_solve_flops = 0;
    std::vector<char> cluster_fused;         // per cluster: assembled by the fused kernel
    int n_fused_clusters = 0, n_wave_clusters = 0, n_wave2_clusters = 0, n_wave4_clusters = 0, n_wave5_clusters = 0;
    std::vector<int> host_UR, host_UL;       // flattened per (block, r) for clrs_get_unique_counts
    std::vector<i64> host_U_off;
};

template <class T>
static int upload(clrs_ctx *c, const std::vector<T> &h, T **d) {
    *d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIPCHECK(hipMalloc((void **)d, bytes));
    c->allocs.push_back(*d);
    if (!h.empty()) HIPCHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
static int dmalloc(clrs_ctx *c, double **d, i64 n) {
    size_t bytes = (size_t)std::max<i64>(n, 1) * sizeof(double);
    HIPCHECK(hipMalloc((void **)d, bytes));
    c->allocs.push_back(*d);
    HIPCHECK(hipMemset(*d, 0, bytes));
    return 0;
}

// ---- stage builders -----------------------------------------------------------------------------
// 128 x 128 or 64 x 64 tiles for one product: whichever finishes its last round of workgroups first, in units of one 64 x 64 tile on a
// CU of its own (scripts/gemm_probe.py).  Small tiles: 1024 resident (four per CU), the matrix pipe at 0.63 of its peak; a tile on
// the edge of C skips the MFMAs outside C and, dispatched first, hides behind the whole tiles on its CU (about 0.3 of a tile).
// Large tiles: 512 resident (two per CU, four units each) at 0.72, a lone one on a CU at 0.60; an edge tile is bound by the latency
// of its loads and takes as long as a whole one.  4096 x 4096: two rounds of large tiles; 4097 x 4097 (1089 large tiles: a third
// round) and 2049 x 2049 (289: 33 CUs with two) are faster with small ones.
static bool gemm_prefers_large_tiles(const GemmDesc &d) {
    if (d.M < 256 || d.N < 256) return false;
    const long long batch = std::max(1, d.pad0);
    auto count = [&](long long tm, long long tn) { return (d.lower_only ? tm * (tn + 1) / 2 : tm * tn) * batch; };
    const long long all64 = count((d.M + 63) / 64, (d.N + 63) / 64), full64 = count(d.M / 64, d.N / 64), all128 = count((d.M + 127) / 128, (d.N + 127) / 128);
    const double small = ((double)((full64 + 1023) / 1024) * 4.0 + 0.3 * 4.0 * (double)(all64 - full64) / 1024.0) / 0.63;
    const double large = all128 > 256 ? (double)((all128 + 511) / 512) * 8.0 / 0.72 : 4.0 / 0.60;
    return large < small;
}
static int add_gemm_stage(clrs_ctx *c, Plan &pl, const std::vector<GemmDesc> &descs) {
    // tile size per product: gemm_prefers_large_tiles
    for (int var = 11; var >= 0; var--) {                      // one launch per (tile size / offset width, transposes): template parameters of the kernel
        const int cls = var >> 2, vta = (var >> 1) & 1, vtb = var & 1;          // class 0: 64 x 64 tiles, 1: 128 x 128, 2: 64 x 64 with 64-bit offsets
        const int big = cls == 1;
        const int BM = big ? 128 : GEMM_BM, BN = big ? 128 : GEMM_BN;
        std::vector<GemmDesc> ds;
        std::vector<GemmTile> tiles;
        for (const GemmDesc &d : descs) {
            if (d.M <= 0 || d.N <= 0) continue;
            // the kernel addresses a tile's operands by 32-bit byte offsets from the tile's origin (gemm_offsets_reach): a product whose
            // leading dimensions are beyond the reach of the large tile takes the small one, beyond that the 64-bit instantiation
            int want = gemm_prefers_large_tiles(d) ? 1 : 0;
            if (want == 1 && !gemm_offsets_reach(d, 128)) want = 0;
            if (want == 0 && !gemm_offsets_reach(d, 64)) want = 2;
            if (want != cls || (d.ta != 0) != (vta == 1) || (d.tb != 0) != (vtb == 1)) continue;
            int id = (int)ds.size();
            ds.push_back(d);
            int tm = (d.M + BM - 1) / BM, tn = (d.N + BN - 1) / BN;
            int batch = std::max(1, d.pad0);   // batch count is encoded by the caller through pad0 (>=1)
            // tiles on the edge of C first: they skip most of their MFMAs and are bound by the latency of their loads, which costs nothing
            // while whole tiles share their CUs -- and a full tile's time at the tail of the launch when they come last
            const bool ragged_m = d.M % BM != 0, ragged_n = d.N % BN != 0;
            for (int edge = 1; edge >= 0; edge--)
                for (int b = 0; b < batch; b++)
                    for (int j = 0; j < tn; j++)
                        for (int i = 0; i < tm; i++) {
                            if (d.lower_only && (i + 1) * BM <= j * BN) continue;
                            const bool is_edge = (ragged_m && i == tm - 1) || (ragged_n && j == tn - 1);
                            if (is_edge == (edge == 1)) tiles.push_back(GemmTile{id, b, i, j});
                        }
        }
        if (tiles.empty()) continue;
        Step s;
        s.kind = STEP_GEMM;
        s.grid = (int)tiles.size();
        s.nmax = cls; s.aux0 = vta * 2 + vtb;
        GemmDesc *dd; GemmTile *dt;
        int rc;
        if ((rc = upload(c, ds, &dd))) return rc;
        if ((rc = upload(c, tiles, &dt))) return rc;
        s.d0 = dd; s.d1 = dt;
        pl.steps.push_back(s);
    }
    return 0;
}

static GemmDesc mk_gemm(int ta, int tb, int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb,
                        double beta, double *C, int ldc, int batch = 1, i64 sA = 0, i64 sB = 0, i64 sC = 0, int lower_only = 0) {
    GemmDesc d;
    d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.ta = ta; d.tb = tb; d.lower_only = lower_only; d.pad0 = batch; d.alpha = alpha; d.beta = beta;
    d.sA = sA; d.sB = sB; d.sC = sC;
    return d;
}

static int add_trsm_stage(clrs_ctx *c, Plan &pl, const std::vector<TrsmDesc> &descs) {
    std::vector<TrsmDesc> ds;
    std::vector<TrsmWork> work;
    for (const TrsmDesc &d : descs) {
        if (d.n <= 0 || d.nvec <= 0) continue;
        int id = (int)ds.size();
        ds.push_back(d);
        for (int ch = 0; ch * 64 < d.nvec; ch++) work.push_back(TrsmWork{id, ch});
    }
    if (work.empty()) return 0;
    Step s;
    s.kind = STEP_TRSM;
    s.grid = (int)work.size();
    TrsmDesc *dd; TrsmWork *dw;
    int rc;
    if ((rc = upload(c, ds, &dd))) return rc;
    if ((rc = upload(c, work, &dw))) return rc;
    s.d0 = dd; s.d1 = dw;
    pl.steps.push_back(s);
    return 0;
}

// Large triangles (n > TRSM_IB): the same solves through explicit inverses of the TRSM_IB x TRSM_IB diagonal blocks of L.
//   1. every 64 x 64 leaf of L inverted, all leaves of all problems in one launch (k_trtri_diag);
//   2. pairs of inverses joined, log2(TRSM_IB / 64) levels of two GEMM stages:  T = B A^-1,  off-diagonal block = -C^-1 T;
//   3. per TRSM_IB rows:  Y_k = op(Inv_k) B_k  (into a scratch copy of B),  B_rest -= op(L_rest,k) Y_k;  then Y copied back over B.
// n / TRSM_IB levels of two launches instead of n / 64: the level-synchronous chain of plan_trsm_chain is what bounds the staged regime
// (profiles/r02/g_staged_polyopt2048*), not its flops.  The products with the triangular inverse blocks run as full GEMMs
// (2 TRSM_IB n nrhs flops on top of the n^2 nrhs of the substitution).
// share: the triangles are the factors of S, solved with several times per factorisation -- their inverse blocks live in the context
// (one buffer per factor for all the solve plans) and the steps that form them are marked, so that only the first solve after a
// factorisation runs them.
static int plan_trsm_blockinv(clrs_ctx *c, Plan &pl, const std::vector<TrsmJob> &jobs, int trans, bool share) {
    constexpr int IB = TRSM_IB;
    int rc, maxob = 0;
    struct Aux { double *inv, *T, *Y; int nob; };
    std::vector<Aux> aux(jobs.size());
    std::vector<TrtriDesc> leaves;
    for (size_t q = 0; q < jobs.size(); q++) {
        const TrsmJob &j = jobs[q];
        Aux &a = aux[q];
        a.nob = (j.n + IB - 1) / IB;
        maxob = std::max(maxob, a.nob);
        auto hit = share ? c->s_inv.find(j.L) : c->s_inv.end();
        if (hit != c->s_inv.end()) { a.inv = hit->second.first; a.T = hit->second.second; }
        else {
            if ((rc = dmalloc(c, &a.inv, (i64)a.nob * IB * IB))) return rc;      // zeroed: the strict upper triangles stay zero
            if ((rc = dmalloc(c, &a.T, (i64)j.n * (IB / 2)))) return rc;
            if (share) c->s_inv[j.L] = {a.inv, a.T};
        }
        {   // the q-th job of every plan uses the q-th scratch of the context (jobs of ONE plan run side by side and need one each)
            const i64 need = (i64)j.n * j.nrhs;
            if (c->trsm_y_pool.size() <= q) c->trsm_y_pool.resize(q + 1, {0, nullptr});
            if (c->trsm_y_pool[q].first < need) {
                if ((rc = dmalloc(c, &c->trsm_y_pool[q].second, need))) return rc;
                c->trsm_y_pool[q].first = need;
            }
            a.Y = c->trsm_y_pool[q].second;
        }
        for (int o = 0; o < a.nob; o++) {
            const int r0 = o * IB, nb = std::min(IB, j.n - r0);
            for (int l0 = 0; l0 < nb; l0 += TRSM_NB) {
                TrtriDesc d;
                d.L = j.L + (r0 + l0) + (i64)(r0 + l0) * j.ldl; d.ldl = j.ldl; d.n = std::min(TRSM_NB, nb - l0); d.pad = 0;
                d.out = a.inv + (i64)o * IB * IB + l0 + (i64)l0 * IB; d.ldo = IB;
                leaves.push_back(d);
            }
        }
    }
    {
        Step s;
        s.kind = STEP_TRTRI_DIAG; s.grid = (int)leaves.size();
        TrtriDesc *dl;
        if ((rc = upload(c, leaves, &dl))) return rc;
        s.d0 = dl; s.s_inverse = share;
        pl.steps.push_back(s);
        HIPCHECK(hipFuncSetAttribute((const void *)k_trtri_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trtri_diag_lds_bytes()));
    }
    for (int b = TRSM_NB; b < IB; b *= 2) {
        std::vector<GemmDesc> ga, gb;
        for (size_t q = 0; q < jobs.size(); q++) {
            const TrsmJob &j = jobs[q];
            const Aux &a = aux[q];
            i64 toff = 0;
            for (int o = 0; o < a.nob; o++) {
                const int r0 = o * IB, nb = std::min(IB, j.n - r0);
                double *inv = a.inv + (i64)o * IB * IB;
                for (int lo = 0; lo + b < nb; lo += 2 * b) {
                    const int mid = lo + b, hi = std::min(lo + 2 * b, nb), rows = hi - mid;
                    double *T = a.T + toff;
                    toff += (i64)rows * b;
                    ga.push_back(mk_gemm(0, 0, rows, b, b, 1.0, j.L + (r0 + mid) + (i64)(r0 + lo) * j.ldl, j.ldl, inv + lo + (i64)lo * IB, IB, 0.0, T, rows));
                    gb.push_back(mk_gemm(0, 0, rows, b, rows, -1.0, inv + mid + (i64)mid * IB, IB, T, rows, 0.0, inv + mid + (i64)lo * IB, IB));
                }
            }
        }
        const size_t first = pl.steps.size();
        if ((rc = add_gemm_stage(c, pl, ga))) return rc;
        if ((rc = add_gemm_stage(c, pl, gb))) return rc;
        for (size_t i = first; i < pl.steps.size(); i++) pl.steps[i].s_inverse = share;
    }
    for (int step = 0; step < maxob; step++) {
        std::vector<GemmDesc> gy, gu;
        for (size_t q = 0; q < jobs.size(); q++) {
            const TrsmJob &j = jobs[q];
            const Aux &a = aux[q];
            if (step >= a.nob) continue;
            const int k = trans ? a.nob - 1 - step : step, r0 = k * IB, nk = std::min(IB, j.n - r0);
            const double *inv = a.inv + (i64)k * IB * IB;
            gy.push_back(mk_gemm(trans ? 1 : 0, 0, nk, j.nrhs, nk, 1.0, inv, IB, j.B + r0, j.ldb, 0.0, a.Y + r0, j.n));
            if (!trans) {
                const int rem = j.n - (r0 + nk);
                if (rem > 0) gu.push_back(mk_gemm(0, 0, rem, j.nrhs, nk, -1.0, j.L + (r0 + nk) + (i64)r0 * j.ldl, j.ldl, a.Y + r0, j.n, 1.0, j.B + r0 + nk, j.ldb));
            } else if (r0 > 0)
                gu.push_back(mk_gemm(1, 0, r0, j.nrhs, nk, -1.0, j.L + r0, j.ldl, a.Y + r0, j.n, 1.0, j.B, j.ldb));
        }
        if ((rc = add_gemm_stage(c, pl, gy))) return rc;
        if ((rc = add_gemm_stage(c, pl, gu))) return rc;
    }
    std::vector<Copy2dDesc> cp;
    i64 most = 0;
    for (size_t q = 0; q < jobs.size(); q++) {
        const TrsmJob &j = jobs[q];
        cp.push_back(Copy2dDesc{aux[q].Y, j.B, j.n, j.ldb, j.n, j.nrhs});
        most = std::max(most, (i64)j.n * j.nrhs);
    }
    Step s;
    s.kind = STEP_COPY2D; s.grid = (int)cp.size(); s.aux0 = (int)std::min<i64>((most + 1023) / 1024, 4096);
    Copy2dDesc *dc;
    if ((rc = upload(c, cp, &dc))) return rc;
    s.d0 = dc;
    pl.steps.push_back(s);
    return 0;
}

// B <- L^-1 B (trans = 0) or L^-T B (trans = 1) for a list of independent problems, blocked by
// TRSM_NB, level-synchronous over the problems.
static int plan_trsm_chain(clrs_ctx *c, Plan &pl, const std::vector<TrsmJob> &jobs, int trans);
static int plan_trsm(clrs_ctx *c, Plan &pl, const std::vector<TrsmJob> &jobs, int trans, bool factors_of_S = false) {
    std::vector<TrsmJob> chain, big;
    for (const TrsmJob &j : jobs) {
        if (j.n <= 0 || j.nrhs <= 0) continue;
        const bool inv = g_cfg_trsm_blockinv && j.n > TRSM_IB && (i64)j.n * j.nrhs <= (i64)1 << 31;       // scratch copy of B: 16 GB at most
        (inv ? big : chain).push_back(j);
    }
    int rc;
    if (!chain.empty() && (rc = plan_trsm_chain(c, pl, chain, trans))) return rc;
    if (!big.empty() && (rc = plan_trsm_blockinv(c, pl, big, trans, factors_of_S))) return rc;
    return 0;
}
static int plan_trsm_chain(clrs_ctx *c, Plan &pl, const std::vector<TrsmJob> &jobs, int trans) {
    int maxp = 0;
    for (const TrsmJob &j : jobs) maxp = std::max(maxp, (j.n + TRSM_NB - 1) / TRSM_NB);
    int rc;
    for (int step = 0; step < maxp; step++) {
        std::vector<TrsmDesc> td;
        std::vector<GemmDesc> gd;
        for (const TrsmJob &j : jobs) {
            if (j.n <= 0 || j.nrhs <= 0) continue;
            int np = (j.n + TRSM_NB - 1) / TRSM_NB;
            if (step >= np) continue;
            int k = trans ? np - 1 - step : step;
            int r0 = k * TRSM_NB, nk = std::min(TRSM_NB, j.n - r0);
            TrsmDesc d;
            d.L = j.L + r0 + (i64)r0 * j.ldl; d.ldl = j.ldl; d.n = nk; d.B = j.B + r0; d.nvec = j.nrhs; d.trans = trans;
            d.es = 1; d.vs = j.ldb;
            td.push_back(d);
            if (!trans) {
                int rem = j.n - (r0 + nk);
                if (rem > 0)  // B[r0+nk:, :] -= L[r0+nk:, r0:r0+nk] B[r0:r0+nk, :]
                    gd.push_back(mk_gemm(0, 0, rem, j.nrhs, nk, -1.0, j.L + (r0 + nk) + (i64)r0 * j.ldl, j.ldl, j.B + r0, j.ldb, 1.0,
                                         j.B + r0 + nk, j.ldb));
            } else {
                if (r0 > 0)  // B[0:r0, :] -= L[r0:r0+nk, 0:r0]^T B[r0:r0+nk, :]
                    gd.push_back(mk_gemm(1, 0, r0, j.nrhs, nk, -1.0, j.L + r0, j.ldl, j.B + r0, j.ldb, 1.0, j.B, j.ldb));
            }
        }
        if ((rc = add_trsm_stage(c, pl, td))) return rc;
        if ((rc = add_gemm_stage(c, pl, gd))) return rc;
    }
    return 0;
}

// Matrices beyond one block: one launch per block column (k_chol_level) instead of the three of plan_potrf_chain, then the diagonal
// factors copied from their side buffer into the matrices.
static int plan_potrf_levels(clrs_ctx *c, Plan &pl, const std::vector<PotrfJob> &jobs, int *info) {
    int rc, maxp = 0;
    std::vector<double *> dbuf(jobs.size());
    auto stop_of = [](const PotrfJob &j) { const int np = (j.n + 63) / 64; return j.stop > 0 ? std::min(j.stop, np) : np; };
    for (size_t q = 0; q < jobs.size(); q++) {
        const int np = (jobs[q].n + 63) / 64;
        maxp = std::max(maxp, np);
        if ((rc = dmalloc(c, &dbuf[q], (i64)np * 4096))) return rc;
    }
    HIPCHECK(hipFuncSetAttribute((const void *)k_chol_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_level_lds_bytes()));
    // launch lvl: block column lvl final; column workgroups for block column lvl + 1 (while it is to be factored), trailing tiles beyond
    for (int lvl = -1; lvl <= maxp - 2; lvl++) {
        std::vector<CholLevelJob> cj;
        std::vector<CholLevelWork> col, bulk;
        for (size_t q = 0; q < jobs.size(); q++) {
            const PotrfJob &j = jobs[q];
            const int np = (j.n + 63) / 64, cb = lvl + 1, stop = stop_of(j);
            if (cb >= np || cb > stop) continue;
            const bool closing = cb == stop;                       // nothing left to factor: the trailing update of block column lvl alone
            const int id = (int)cj.size(), bulk0 = closing ? cb : cb + 1;
            cj.push_back(CholLevelJob{j.A, dbuf[q], j.lda, j.n, lvl, j.code, bulk0, 0});
            if (!closing)
                for (int i = cb; i < np; i++) col.push_back(CholLevelWork{id, 0, i, 0});
            const int rb = bulk0 * 64;
            if (lvl >= 0 && rb < j.n) {
                const int nt = (j.n - rb + 127) / 128;
                for (int tj = 0; tj < nt; tj++)
                    for (int ti = tj; ti < nt; ti++) bulk.push_back(CholLevelWork{id, 1, ti, tj});
            }
        }
        if (col.empty() && bulk.empty()) continue;
        const int ncol = (int)col.size();
        col.insert(col.end(), bulk.begin(), bulk.end());       // the column workgroups (the critical path) are dispatched first
        Step s;
        s.kind = STEP_CHOL_LEVEL; s.grid = (int)col.size(); s.dst = info ? info : c->d_info;
        if (cj.size() == 1) {                                  // the kernel derives the same order from blockIdx.x
            c->chol_jobs.push_back(cj[0]);
            s.aux1 = 1; s.aux0 = ncol; s.n = (i64)c->chol_jobs.size() - 1;
        } else {
            CholLevelJob *dj; CholLevelWork *dw;
            if ((rc = upload(c, cj, &dj))) return rc;
            if ((rc = upload(c, col, &dw))) return rc;
            s.d0 = dj; s.d1 = dw;
        }
        pl.steps.push_back(s);
    }
    std::vector<Copy2dDesc> cp;
    for (size_t q = 0; q < jobs.size(); q++) {
        const PotrfJob &j = jobs[q];
        const int stop = stop_of(j);
        for (int r0 = 0, b = 0; r0 < j.n && b < stop; r0 += 64, b++) {
            const int m = std::min(64, j.n - r0);
            cp.push_back(Copy2dDesc{dbuf[q] + (i64)b * 4096, j.A + r0 + (i64)r0 * j.lda, 64, j.lda, m, m});
        }
    }
    Step s;
    s.kind = STEP_COPY2D; s.grid = (int)cp.size(); s.aux0 = 4;
    Copy2dDesc *dc;
    if ((rc = upload(c, cp, &dc))) return rc;
    s.d0 = dc;
    pl.steps.push_back(s);
    return 0;
}

// in-place lower Cholesky of a list of independent matrices, blocked by POTRF_NB, level-synchronous.
static int plan_potrf_chain(clrs_ctx *c, Plan &pl, const std::vector<PotrfJob> &jobs, int *info);
static int plan_potrf(clrs_ctx *c, Plan &pl, const std::vector<PotrfJob> &jobs, int *info = nullptr) {
    std::vector<PotrfJob> chain, big;
    for (const PotrfJob &j : jobs) ((g_cfg_potrf_levels && j.n > POTRF_NB) ? big : chain).push_back(j);
    int rc;
    if (!chain.empty() && (rc = plan_potrf_chain(c, pl, chain, info))) return rc;
    if (!big.empty() && (rc = plan_potrf_levels(c, pl, big, info))) return rc;
    return 0;
}
static int plan_potrf_chain(clrs_ctx *c, Plan &pl, const std::vector<PotrfJob> &jobs, int *info) {
    int maxp = 0;
    for (const PotrfJob &j : jobs) maxp = std::max(maxp, (j.n + POTRF_NB - 1) / POTRF_NB);
    int rc;
    for (int k = 0; k < maxp; k++) {
        std::vector<PotrfDesc> pd;
        std::vector<TrsmDesc> td;
        std::vector<GemmDesc> gd;
        for (const PotrfJob &j : jobs) {
            int r0 = k * POTRF_NB;
            if (r0 >= j.n) continue;
            int nk = std::min(POTRF_NB, j.n - r0), rem = j.n - r0 - nk;
            double *Akk = j.A + r0 + (i64)r0 * j.lda;
            pd.push_back(PotrfDesc{Akk, j.lda, nk, j.code, 0});
            if (rem > 0) {
                double *Pn = j.A + (r0 + nk) + (i64)r0 * j.lda;  // panel below the diagonal block
                TrsmDesc d;                                       // rows of the panel: x L_kk^T = b
                d.L = Akk; d.ldl = j.lda; d.n = nk; d.B = Pn; d.nvec = rem; d.trans = 0; d.es = j.lda; d.vs = 1;
                td.push_back(d);
                double *A22 = j.A + (r0 + nk) + (i64)(r0 + nk) * j.lda;
                gd.push_back(mk_gemm(0, 1, rem, rem, nk, -1.0, Pn, j.lda, Pn, j.lda, 1.0, A22, j.lda, 1, 0, 0, 0, 1));
            }
        }
        if (pd.empty()) continue;
        Step s;
        s.kind = STEP_POTRF;
        s.grid = (int)pd.size();
        s.dst = info ? info : c->d_info;
        PotrfDesc *dd;
        if ((rc = upload(c, pd, &dd))) return rc;
        s.d0 = dd;
        pl.steps.push_back(s);
        if ((rc = add_trsm_stage(c, pl, td))) return rc;
        if ((rc = add_gemm_stage(c, pl, gd))) return rc;
    }
    return 0;
}

static void add_memcpy(Plan &pl, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return;
    Step s;
    s.kind = STEP_MEMCPY;
    s.dst = dst; s.src = src; s.bytes = bytes;
    pl.steps.push_back(s);
}

// ---- plan execution -----------------------------------------------------------------------------
static int run_steps(clrs_ctx *c, const Plan &pl) {
    hipStream_t st = c->stream;
    for (const Step &s : pl.steps) {
        if (s.s_inverse && c->s_inv_valid && !c->graph_mode) continue;       // (a captured graph replays all of its steps)
        const bool timed = !c->graph_mode && (c->kt_kind == -1 || c->kt_kind == (int)s.kind) && (int)c->kt_kinds.size() < KT_MAX_EVENTS;
        if (timed) {
            const size_t need = 2 * (c->kt_kinds.size() + 1);
            while (c->kt_ev.size() < need) {
                hipEvent_t e;
                HIPCHECK(hipEventCreate(&e));
                c->kt_ev.push_back(e);
            }
            HIPCHECK(hipEventRecord(c->kt_ev[need - 2], st));
        }
        switch (s.kind) {
            case STEP_MEMCPY:
                HIPCHECK(hipMemcpyAsync(s.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, st));
                break;
            case STEP_GEMM:
#define CLRS_GEMM_LAUNCH(BMN, TA, TB) \
    hipLaunchKernelGGL((k_gemm_f64_t<BMN, BMN, TA, TB>), dim3(s.grid), dim3(256), gemm_lds_bytes(BMN, BMN), st, (const GemmDesc *)s.d0, (const GemmTile *)s.d1)
#define CLRS_GEMM_LAUNCH_WIDE(TA, TB) \
    hipLaunchKernelGGL((k_gemm_f64_t<64, 64, TA, TB, unsigned long long>), dim3(s.grid), dim3(256), gemm_lds_bytes(64, 64), st, (const GemmDesc *)s.d0, (const GemmTile *)s.d1)
                switch (s.nmax * 4 + s.aux0) {
                    case 8: CLRS_GEMM_LAUNCH_WIDE(0, 0); break;
                    case 9: CLRS_GEMM_LAUNCH_WIDE(0, 1); break;
                    case 10: CLRS_GEMM_LAUNCH_WIDE(1, 0); break;
                    case 11: CLRS_GEMM_LAUNCH_WIDE(1, 1); break;
                    case 0: CLRS_GEMM_LAUNCH(64, 0, 0); break;
                    case 1: CLRS_GEMM_LAUNCH(64, 0, 1); break;
                    case 2: CLRS_GEMM_LAUNCH(64, 1, 0); break;
                    case 3: CLRS_GEMM_LAUNCH(64, 1, 1); break;
                    case 4: CLRS_GEMM_LAUNCH(128, 0, 0); break;
                    case 5: CLRS_GEMM_LAUNCH(128, 0, 1); break;
                    case 6: CLRS_GEMM_LAUNCH(128, 1, 0); break;
                    default: CLRS_GEMM_LAUNCH(128, 1, 1); break;
                }
#undef CLRS_GEMM_LAUNCH
#undef CLRS_GEMM_LAUNCH_WIDE
                break;
            case STEP_TRSM:
                hipLaunchKernelGGL(k_trsm_diag, dim3(s.grid), dim3(256), 0, st, (const TrsmDesc *)s.d0, (const TrsmWork *)s.d1);
                break;
            case STEP_POTRF:
                hipLaunchKernelGGL(k_potrf_diag, dim3(s.grid), dim3(256), 0, st, (const PotrfDesc *)s.d0, (int *)s.dst);
                break;
            case STEP_GATHER_S:
                if (s.aux0 >= 4)       // some cluster has four blocks or more: four lanes per entry walk them
                    hipLaunchKernelGGL(k_schur_gather<4>, dim3(s.grid), dim3(1024), 0, st, (const SClusterDesc *)s.d0, (const SBlockDesc *)s.d1, (const STile *)s.d2);
                else
                    hipLaunchKernelGGL(k_schur_gather<1>, dim3(s.grid), dim3(256), 0, st, (const SClusterDesc *)s.d0, (const SBlockDesc *)s.d1, (const STile *)s.d2);
                break;
            case STEP_GATHER_SCALAR:
                hipLaunchKernelGGL(k_gather_scalar, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, st, (double *)s.dst, (const double *)s.src,
                                   (const i64 *)s.d0, s.n);
                break;
            case STEP_SUB:
                hipLaunchKernelGGL(k_sub, dim3((unsigned)((s.n + 255) / 256)), dim3(256), 0, st, (double *)s.dst, (const double *)s.src,
                                   (const double *)s.d0, (int)s.n);
                break;
            case STEP_MEMSET_INFO:
                HIPCHECK(hipMemsetAsync(s.dst ? s.dst : (void *)c->d_info, 0x7f, sizeof(int), st));
                break;
            case STEP_FUSED_ASSEMBLE:
                hipLaunchKernelGGL(k_cluster_assemble, dim3(s.grid), dim3(256), s.bytes, st, (const FCluster *)s.d0, (const FBlock *)s.d1, *(const FTables *)s.src);
                break;
            case STEP_DENSE_BLOCK:
                hipLaunchKernelGGL(k_dense_block, dim3(s.grid), dim3(256), s.bytes, st, (const DBlock *)s.d0, *(const FTables *)s.src);
                break;
            case STEP_TRTRI32:
                hipLaunchKernelGGL(k_trtri32, dim3(s.grid), dim3(64), 0, st, (const DenseTBlock *)s.d0);
                break;
            case STEP_DENSE_T32:
                hipLaunchKernelGGL(k_dense_T32, dim3(s.grid), dim3(64 * DT32_WAVES), dense_T32_lds_bytes(), st, (const DenseTBlock *)s.d0, (const DenseTPair *)s.d1);
                break;
            case STEP_TRTRI_DIAG:
                hipLaunchKernelGGL(k_trtri_diag, dim3(s.grid), dim3(256), trtri_diag_lds_bytes(), st, (const TrtriDesc *)s.d0);
                break;
            case STEP_CHOL_LEVEL:
                if (s.aux1 == 1) {       // one matrix: job by value (kept in the context), no tables
                    hipLaunchKernelGGL(k_chol_level, dim3(s.grid), dim3(CL_NT), chol_level_lds_bytes(), st, c->chol_jobs[(size_t)s.n], s.aux0,
                                       (const CholLevelJob *)nullptr, (const CholLevelWork *)nullptr, (int *)s.dst);
                } else
                    hipLaunchKernelGGL(k_chol_level, dim3(s.grid), dim3(CL_NT), chol_level_lds_bytes(), st, CholLevelJob{}, 0, (const CholLevelJob *)s.d0,
                                       (const CholLevelWork *)s.d1, (int *)s.dst);
                break;
            case STEP_CHOL_PACK:
                hipLaunchKernelGGL(k_chol_pack, dim3(s.grid), dim3(256), 0, st, c->chol_aug[(size_t)s.n]);
                break;
            case STEP_CHOL_UNPACK:
                hipLaunchKernelGGL(k_chol_unpack, dim3(s.grid), dim3(256), 0, st, c->chol_aug[(size_t)s.n]);
                break;
            case STEP_COPY2D:
                hipLaunchKernelGGL(k_copy2d, dim3(s.aux0, s.grid), dim3(256), 0, st, (const Copy2dDesc *)s.d0);
                break;
            case STEP_SOLVE_SMALL:
                if (s.aux0 >= 2) {
                    const int ph = s.aux0 - 2;
                    clrs_ctx::SolveSmall2 &q = c->ss2[ph];
                    // right-hand sides bound for this call: piece i of a vector starts 16 x (its destination offset - the first piece's) in
                    // (a pure zero-fill piece -- no valid rows -- reads and ignores the first entry)
                    auto piece_off = [&](int t, int first) -> long long {
                        const StageJob &jb = q.jobs.j[t];
                        return ((jb.meta >> 16) & 31) ? (long long)(jb.dst_ldd & 0xffff) - (long long)(q.jobs.j[first].dst_ldd & 0xffff) : 0;
                    };
                    for (int j = 0; j < c->J; j++)
                        for (int t = q.rx_job[j]; t < q.rx_job[8 + j]; t++) q.jobs.j[t].src = c->bind_rhsx + c->solve8.d[j].off + piece_off(t, q.rx_job[j]);
                    for (int t = q.ry_job[0]; t >= 0 && t < q.ry_job[1]; t++) q.jobs.j[t].src = c->bind_rhsy + piece_off(t, q.ry_job[0]);
#define CLRS_SS2(NJ, PH) hipLaunchKernelGGL((k_solve_small2<NJ, PH>), dim3(1), dim3(256), s.bytes, st, c->solve8, q.jobs, c->J, c->N, (int)c->xlen, c->bind_dx, c->bind_dy, c->d_t, c->d_u)
#define CLRS_SS2_NJ(PH)                                  \
    do {                                                 \
        if (q.jobs.n <= 8) CLRS_SS2(8, PH);              \
        else if (q.jobs.n <= 16) CLRS_SS2(16, PH);       \
        else if (q.jobs.n <= 24) CLRS_SS2(24, PH);       \
        else if (q.jobs.n <= 32) CLRS_SS2(32, PH);       \
        else CLRS_SS2(48, PH);                           \
    } while (0)
                    if (ph == 0) CLRS_SS2_NJ(0);
                    else if (ph == 1) CLRS_SS2_NJ(1);
                    else CLRS_SS2_NJ(2);
#undef CLRS_SS2_NJ
#undef CLRS_SS2
                    break;
                }
                hipLaunchKernelGGL(k_solve_small, dim3(1), dim3(256), s.bytes, st, (const CSolve *)s.d0, c->J, (const double *)c->d_Q, (const double *)c->d_dinvQ, c->N,
                                   (int)c->xlen, c->bind_rhsx, c->bind_rhsy, (const double *)c->d_LB, c->bind_dx, c->bind_dy, (int)s.n);
                break;
            case STEP_ASSEMBLE_W2: {
                const FTables *tb = (const FTables *)s.src;
                const int ncl = (int)s.n;
                const bool full = s.grid != 0;      // every low-rank block is exactly 16 x (16 UT)
                if (s.nmax <= 2 && full)
                    hipLaunchKernelGGL((k_cluster_assemble_w2<2, true>), dim3((ncl + 3) / 4), dim3(256), s.bytes, st, (const W2Cluster *)s.d0, (const W2Block *)s.d1, *tb, ncl);
                else if (s.nmax <= 2)
                    hipLaunchKernelGGL((k_cluster_assemble_w2<2, false>), dim3((ncl + 3) / 4), dim3(256), s.bytes, st, (const W2Cluster *)s.d0, (const W2Block *)s.d1, *tb, ncl);
                else
                    hipLaunchKernelGGL((k_cluster_assemble_w2<4, false>), dim3((ncl + 3) / 4), dim3(256), s.bytes, st, (const W2Cluster *)s.d0, (const W2Block *)s.d1, *tb, ncl);
                break;
            }
            case STEP_FACTOR_SMALL: {
#define CLRS_FS(NJ) hipLaunchKernelGGL(k_factor_small<NJ>, dim3(1), dim3(256), s.bytes, st, c->fsmall, c->fsmall_jobs, c->d_info)
                if (c->fsmall_jobs.n <= 8) CLRS_FS(8);
                else if (c->fsmall_jobs.n <= 16) CLRS_FS(16);
                else if (c->fsmall_jobs.n <= 24) CLRS_FS(24);
                else if (c->fsmall_jobs.n <= 32) CLRS_FS(32);
                else CLRS_FS(48);
#undef CLRS_FS
                break;
            }
            case STEP_ASSEMBLE_W3: {
                W3Tables tb = *(const W3Tables *)s.src;
                tb.Xc = c->ftables.Xc; tb.Y = c->ftables.Y;      // the iterates bound for this call (the caller's buffers or the context's)
                if (s.grid != 0)
                    hipLaunchKernelGGL((k_cluster_assemble_w3<true>), dim3(s.aux0), dim3(256), 0, st, (const int *)s.d0, (const W3Block *)s.d1, tb, (int)s.n, s.aux1);
                else
                    hipLaunchKernelGGL((k_cluster_assemble_w3<false>), dim3(s.aux0), dim3(256), 0, st, (const int *)s.d0, (const W3Block *)s.d1, tb, (int)s.n, s.aux1);
                break;
            }
            case STEP_ASSEMBLE_W4: {
                W4Tables tb = *(const W4Tables *)s.src;
                tb.Xc = c->ftables.Xc; tb.Y = c->ftables.Y;      // the iterates bound for this call
                if (s.nmax <= 3)
                    hipLaunchKernelGGL((k_cluster_assemble_w4<3>), dim3(s.aux0), dim3(256), s.bytes, st, (const int *)s.d0, (const W3Block *)s.d1, tb, (int)s.n, s.aux1, s.grid);
                else
                    hipLaunchKernelGGL((k_cluster_assemble_w4<4>), dim3(s.aux0), dim3(256), s.bytes, st, (const int *)s.d0, (const W3Block *)s.d1, tb, (int)s.n, s.aux1, s.grid);
                break;
            }
            case STEP_ASSEMBLE_W5: {
                W5Tables tb = *(const W5Tables *)s.src;
                tb.Xc = c->ftables.Xc; tb.Y = c->ftables.Y;      // the iterates bound for this call
                hipLaunchKernelGGL(k_cluster_assemble_w5, dim3(s.aux0), dim3(256), 0, st, (const int *)s.d0, (const W3Block *)s.d1, tb, (int)s.n, s.aux1);
                break;
            }
            case STEP_ASSEMBLE_W1: {
                const FTables *tb = (const FTables *)s.src;
                const int nw = (int)s.n, mb = (int)(intptr_t)s.dst;
                if (s.nmax <= 2)
                    hipLaunchKernelGGL(k_cluster_assemble_w1<2>, dim3(s.grid), dim3(64 * nw), s.bytes, st, (const WCluster *)s.d0, (const WBlock *)s.d1, *tb, nw, mb);
                else if (s.nmax <= 4)
                    hipLaunchKernelGGL(k_cluster_assemble_w1<4>, dim3(s.grid), dim3(64 * nw), s.bytes, st, (const WCluster *)s.d0, (const WBlock *)s.d1, *tb, nw, mb);
                else
                    hipLaunchKernelGGL(k_cluster_assemble_w1<8>, dim3(s.grid), dim3(64 * nw), s.bytes, st, (const WCluster *)s.d0, (const WBlock *)s.d1, *tb, nw, mb);
                break;
            }
            case STEP_SMALL_POTRF:
                if (s.n == 0)
                    hipLaunchKernelGGL(k_small_potrf, dim3(s.grid), dim3(256), s.bytes, st, (const SmallPotrf *)s.d0, c->bind_X, c->bind_Xchol, (int *)s.dst);
                else if (s.n == 1)
                    hipLaunchKernelGGL(k_small_potrf, dim3(s.grid), dim3(256), s.bytes, st, (const SmallPotrf *)s.d0, (const double *)c->d_Q, c->d_Q, (int *)s.dst);
                else   // Q = sum of the per-cluster slabs, factored in the same launch
                    hipLaunchKernelGGL(k_small_potrf, dim3(s.grid), dim3(256), s.bytes, st, (const SmallPotrf *)s.d0, (const double *)c->d_Qslabs, c->d_Q, (int *)s.dst);
                break;
            case STEP_GRAM_SMALL:
                hipLaunchKernelGGL(k_gram_small, dim3(c->N * c->N), dim3(256), 0, st, (const double *)c->d_LB, (int)c->xlen, (int)c->xlen, c->N, c->d_Q);
                break;
            case STEP_SUM_S_SLABS:
                hipLaunchKernelGGL(k_sum_S_slabs, dim3((unsigned)((s.n + 1023) / 1024), s.grid), dim3(256), 0, st, (const SSlabSum *)s.d0);
                break;
            case STEP_SUM_SLABS:
                hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)(((i64)c->N * c->N + 255) / 256)), dim3(256), 0, st, (const double *)c->d_Qslabs, (i64)c->N * c->N, c->J,
                                   (i64)c->N * c->N, c->d_Q);
                break;
            case STEP_CLUSTER_FACTOR:
                hipLaunchKernelGGL(k_cluster_factor, dim3(s.grid), dim3(256), s.bytes, st, (const CFactor *)s.d0, (int *)s.dst);
                break;
            case STEP_GEMV_T:
                hipLaunchKernelGGL(k_gemv_t, dim3((c->N + 3) / 4), dim3(256), 0, st, (const double *)c->d_LB, (int)c->xlen, (int)c->xlen, c->N, (const double *)c->d_t, c->d_u);
                break;
            case STEP_CSOLVE_FWD:
                hipLaunchKernelGGL(k_cluster_solve_fwd, dim3(s.grid), dim3(256), s.bytes, st, (const CSolve *)s.d0, c->bind_rhsx, c->d_t);
                break;
            case STEP_Q_SOLVE:
                hipLaunchKernelGGL(k_q_solve, dim3(1), dim3(256), s.bytes, st, (const double *)c->d_Q, (const double *)c->d_dinvQ, c->N, c->bind_rhsy, c->d_u, c->bind_dy,
                                   s.n ? (const double *)c->d_LB : (const double *)nullptr, (int)c->xlen, (int)c->xlen, (const double *)c->d_t);
                break;
            case STEP_CSOLVE_BWD:
                hipLaunchKernelGGL(k_cluster_solve_bwd, dim3(s.grid), dim3(256), s.bytes, st, (const CSolve *)s.d0, (const double *)c->bind_dy, (const double *)c->d_t, c->bind_dx);
                break;
            case STEP_ZERO_UPPER:
                hipLaunchKernelGGL(k_zero_upper, dim3((unsigned)((s.n + 255) / 256), (unsigned)s.grid), dim3(256), 0, st, (const PotrfDesc *)s.d0);
                break;
            default:
                break;
        }
        if (timed) {
            HIPCHECK(hipEventRecord(c->kt_ev[2 * c->kt_kinds.size() + 1], st));
            c->kt_kinds.push_back((int)s.kind);
        }
    }
    HIPCHECK(hipGetLastError());
    return 0;
}

static int run_plan(clrs_ctx *c, Plan &pl) {
    if (pl.steps.empty()) return 0;
    if (!c->graph_mode) return run_steps(c, pl);
    if (!c->stream) return fail(CLRS_ERR_STATE, "graph mode cannot capture the default (null) stream: pass a stream of its own to clrs_set_stream");
    const void *now[8] = {c->bind_X, c->bind_Xchol, c->bind_rhsx, c->bind_rhsy, c->bind_dx, c->bind_dy, c->ftables.Xc, c->ftables.Y};
    if (pl.graph && std::memcmp(now, pl.captured, sizeof(now)) != 0) {   // the graph bakes the caller's pointers: re-capture
        hipGraphExecDestroy(pl.graph);
        pl.graph = nullptr;
    }
    if (!pl.graph) {
        std::memcpy(pl.captured, now, sizeof(now));
        hipGraph_t g;
        HIPCHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int rc = run_steps(c, pl);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) return rc;
        if (e != hipSuccess) return fail(CLRS_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        HIPCHECK(hipGraphInstantiate(&pl.graph, g, nullptr, nullptr, 0));
        HIPCHECK(hipGraphDestroy(g));
    }
    HIPCHECK(hipGraphLaunch(pl.graph, c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// context creation
// ------------------------------------------------------------------------------------------------
static bool vec_eq(const double *a, const double *b, int n) {   // exact equality, like the reference's == (src/tools.jl:134)
    for (int i = 0; i < n; i++)
        if (a[i] != b[i]) return false;
    return true;
}

extern "C" int clrs_ctx_create(const clrs_sdp_desc *d, int device, clrs_ctx **out) {
    if (!d || !out) return fail(CLRS_ERR_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(CLRS_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(CLRS_ERR_NO_DEVICE, "device index out of range");
    HIPCHECK(hipSetDevice(device));
    clrs_ctx *c = new clrs_ctx();
    c->device = device;
    int rc = 0;
#define CK(x) do { rc = (x); if (rc) { clrs_ctx_destroy(c); return rc; } } while (0)
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { clrs_ctx_destroy(c); return fail(CLRS_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    HIPCK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCK(hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128)));
    HIPCK(hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128)));
    HIPCK(hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128)));
    HIPCK(hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128)));
    for (int i = 0; i < 10; i++) HIPCK(hipEventCreate(&c->ev[i]));
    c->J = d->n_clusters; c->N = d->n_free; c->NB = d->n_blocks;
    if (c->J < 0 || c->N < 0 || c->NB < 0) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "negative sizes"); }
    const int J = c->J, N = c->N, NB = c->NB;
    c->P.assign(d->cluster_P, d->cluster_P + J);
    c->coff.assign(J + 1, 0); c->Soff.assign(J + 1, 0);
    for (int j = 0; j < J; j++) {
        if (c->P[j] <= 0) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "cluster with no constraints"); }
        c->coff[j + 1] = c->coff[j] + c->P[j];
        c->Soff[j + 1] = c->Soff[j] + (i64)c->P[j] * c->P[j];
    }
    c->xlen = c->coff[J]; c->Slen = c->Soff[J];
    c->T = d->term_ptr[NB]; c->D = d->dense_ptr[NB];
    const i64 T = c->T;

    // ---- blocks, de-duplication of the sampled vectors (src/solver.jl:985-1059) ----
    c->blk.resize(NB);
    std::vector<int> ridx(T), lidx(T);
    std::vector<i64> partner(T, -1);
    std::vector<std::vector<std::vector<i64>>> uniqR(NB), uniqL(NB);
    i64 xyoff = 0;
    int prevj = 0;
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        k.j = d->block_cluster[b]; k.m = d->block_m[b]; k.delta = d->block_delta[b]; k.kind = d->block_kind[b];
        k.n = k.m * k.delta;
        if (k.j < prevj || k.j >= J || k.m <= 0 || k.delta <= 0) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "bad block description (clusters must be listed in order)"); }
        prevj = k.j;
        k.xyoff = xyoff; xyoff += (i64)k.n * k.n;
        k.t0 = d->term_ptr[b]; k.t1 = d->term_ptr[b + 1]; k.d0 = d->dense_ptr[b]; k.d1 = d->dense_ptr[b + 1];
        c->host_U_off.push_back((i64)c->host_UR.size());
        if (k.kind != 0) {
            if (k.m != 1) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "dense blocks need m == 1"); }
            k.cnt = (int)(k.d1 - k.d0);
            for (i64 e = k.d0; e < k.d1; e++)
                if (d->dense_p[e] < 0 || d->dense_p[e] >= c->P[k.j]) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "dense constraint index out of range"); }
            continue;
        }
        const int m = k.m, dl = k.delta;
        k.UR.assign(m, 0); k.UL.assign(m, 0); k.offR.assign(m + 1, 0); k.offL.assign(m + 1, 0);
        uniqR[b].resize(m); uniqL[b].resize(m);
        for (i64 t = k.t0; t < k.t1; t++) {
            int r = d->term_r[t], s = d->term_s[t], p = d->term_p[t];
            if (r < 0 || r >= m || s < 0 || s >= m || p < 0 || p >= c->P[k.j] || d->term_vec_ptr[t + 1] - d->term_vec_ptr[t] != dl) {
                clrs_ctx_destroy(c);
                return fail(CLRS_ERR_INVALID, "bad term description");
            }
            const double *v = d->term_vs + d->term_vec_ptr[t], *w = d->term_ws + d->term_vec_ptr[t];
            std::vector<i64> &uR = uniqR[b][r], &uL = uniqL[b][r];
            int f = -1;
            for (size_t u = 0; u < uR.size(); u++) if (vec_eq(d->term_vs + d->term_vec_ptr[uR[u]], v, dl)) { f = (int)u; break; }
            if (f < 0) { f = (int)uR.size(); uR.push_back(t); }
            ridx[t] = f;
            f = -1;
            for (size_t u = 0; u < uL.size(); u++) if (vec_eq(d->term_ws + d->term_vec_ptr[uL[u]], w, dl)) { f = (int)u; break; }
            if (f < 0) { f = (int)uL.size(); uL.push_back(t); }
            lidx[t] = f;
        }
        {   // transposed partner (p, s, r, rank) of every term; required by the convention src/solver.jl:1009
            std::map<std::array<int, 4>, i64> index;
            for (i64 t = k.t0; t < k.t1; t++) index[{d->term_p[t], d->term_r[t], d->term_s[t], d->term_rank[t]}] = t;
            for (i64 t = k.t0; t < k.t1; t++) {
                auto it = index.find({d->term_p[t], d->term_s[t], d->term_r[t], d->term_rank[t]});
                if (it == index.end()) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "term without transposed partner: A[r,s][p] must equal A[s,r][p]^T"); }
                partner[t] = it->second;
            }
        }
        k.sym = true;
        for (int r = 0; r < m; r++) {
            k.UR[r] = (int)uniqR[b][r].size(); k.UL[r] = (int)uniqL[b][r].size();
            k.offR[r + 1] = k.offR[r] + k.UR[r]; k.offL[r + 1] = k.offL[r] + k.UL[r];
            c->host_UR.push_back(k.UR[r]); c->host_UL.push_back(k.UL[r]);
            if (k.UR[r] != k.UL[r]) k.sym = false;
            else
                for (int u = 0; u < k.UR[r] && k.sym; u++)
                    if (!vec_eq(d->term_vs + d->term_vec_ptr[uniqR[b][r][u]], d->term_ws + d->term_vec_ptr[uniqL[b][r][u]], dl)) k.sym = false;
        }
        k.URt = k.offR[m]; k.ULt = k.offL[m];
    }
    c->xylen = xyoff;

    // ---- arena layout ----
    // "solve" arena: per low-rank block ZR (n x URt) [+ ZL (n x ULt) if not symmetric]; per dense block W (n x n*cnt).
    // the static arena has the identical layout and holds the expanded vectors / the dense A stacks.
    i64 so = 0, tyo = 0, go = 0, tto = 0, sdo = 0;
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        if (k.kind == 0) {
            k.zr_off = so; so += (i64)k.n * k.URt;
            if (k.sym) k.zl_off = k.zr_off; else { k.zl_off = so; so += (i64)k.n * k.ULt; }
            k.ty_off = tyo; tyo += (i64)k.n * k.URt;
            k.g_off = go; go += 2 * (i64)k.ULt * k.URt;
        } else {
            k.w_off = so; so += (i64)k.n * k.n * k.cnt;
            k.tt_off = tto; tto += (i64)k.n * k.n * k.cnt;
            k.sd_off = sdo; sdo += (i64)k.cnt * k.cnt;
        }
    }
    c->solve_arena_len = so;
    std::vector<double> h_static((size_t)std::max<i64>(so, 1), 0.0);
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        if (k.kind == 0) {
            const int dl = k.delta, n = k.n;
            for (int r = 0; r < k.m; r++) {
                for (int u = 0; u < k.UR[r]; u++) {
                    const double *v = d->term_vs + d->term_vec_ptr[uniqR[b][r][u]];
                    double *dst = h_static.data() + k.zr_off + (i64)(k.offR[r] + u) * n + (i64)r * dl;
                    std::memcpy(dst, v, sizeof(double) * dl);
                }
                if (!k.sym)
                    for (int u = 0; u < k.UL[r]; u++) {
                        const double *w = d->term_ws + d->term_vec_ptr[uniqL[b][r][u]];
                        double *dst = h_static.data() + k.zl_off + (i64)(k.offL[r] + u) * n + (i64)r * dl;
                        std::memcpy(dst, w, sizeof(double) * dl);
                    }
            }
        } else {
            for (i64 e = k.d0; e < k.d1; e++) {
                if (d->dense_A_ptr[e + 1] - d->dense_A_ptr[e] != (i64)k.n * k.n) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "dense matrix has the wrong size"); }
                {   // the dense branch forms <A_i, X^-1 A_k Y> from lower tiles and transposed stores (k_dense_T32, pairing_tri): A_p must be
                    // symmetric, as the reference's constructor makes it (src/interface.jl:1010-1017 symmetrises a non-symmetric entry)
                    const double *Ae = d->dense_A + d->dense_A_ptr[e];
                    for (int cc = 0; cc < k.n; cc++)
                        for (int rr = cc + 1; rr < k.n; rr++)
                            if (Ae[rr + (i64)cc * k.n] != Ae[cc + (i64)rr * k.n]) { clrs_ctx_destroy(c); return fail(CLRS_ERR_INVALID, "dense constraint matrices must be symmetric"); }
                }
                std::memcpy(h_static.data() + k.w_off + (e - k.d0) * (i64)k.n * k.n, d->dense_A + d->dense_A_ptr[e], sizeof(double) * k.n * k.n);
            }
        }
    }
    CK(upload(c, h_static, &c->d_static));
    CK(dmalloc(c, &c->d_work, so));
    CK(dmalloc(c, &c->d_Xc, c->xylen)); CK(dmalloc(c, &c->d_Y, c->xylen)); CK(dmalloc(c, &c->d_X, c->xylen));
    CK(dmalloc(c, &c->d_TY, tyo)); CK(dmalloc(c, &c->d_G, go)); CK(dmalloc(c, &c->d_TT, tto)); CK(dmalloc(c, &c->d_Sd, sdo));
    CK(dmalloc(c, &c->d_S, c->Slen));
    CK(dmalloc(c, &c->d_LB, c->xlen * (i64)N)); CK(dmalloc(c, &c->d_Q, (i64)N * N + N));
    c->d_u = c->d_Q + (i64)N * N;      // u directly behind Q: a sharded caller sums both partials over the ranks with ONE collective
    CK(dmalloc(c, &c->d_t, c->xlen)); CK(dmalloc(c, &c->d_dy, N)); CK(dmalloc(c, &c->d_rhsy, N));
    CK(dmalloc(c, &c->d_dx, c->xlen)); CK(dmalloc(c, &c->d_rhsx, c->xlen));
    CK(dmalloc(c, &c->d_AY, T));
    {   // B stacked: rows = all constraints (cluster after cluster), columns = free variables; ld = xlen
        std::vector<double> hB((size_t)std::max<i64>(c->xlen * (i64)N, 1), 0.0);
        i64 boff = 0;
        for (int j = 0; j < J; j++) {
            for (int col = 0; col < N; col++)
                for (int r = 0; r < c->P[j]; r++) hB[(size_t)(c->coff[j] + r + (i64)col * c->xlen)] = d->B[boff + r + (i64)col * c->P[j]];
            boff += (i64)c->P[j] * N;
        }
        CK(upload(c, hB, &c->d_B));
    }
    {
        int *di;
        std::vector<int> hi(2, INFO_NONE);   // [0]: S_j / Q factorisations, [1]: Cholesky of the X blocks
        CK(upload(c, hi, &di));
        c->d_info = di;
    }

    // ---- per-term gather tables ----
    std::vector<int> h_tL(T), h_tR(T);
    std::vector<double> h_tlam(d->term_lambda, d->term_lambda + T);
    std::vector<i64> h_ayidx(T);
    // the kernel wants the terms of a block sorted by p (CSR over p); build a permutation per block
    std::vector<i64> perm(T);
    std::vector<std::vector<int>> h_tptr(NB);
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        if (k.kind != 0) continue;
        std::vector<i64> idx;
        for (i64 t = k.t0; t < k.t1; t++) idx.push_back(t);
        std::stable_sort(idx.begin(), idx.end(), [&](i64 a, i64 bb) { return d->term_p[a] < d->term_p[bb]; });
        const int Pj = c->P[k.j];
        h_tptr[b].assign(Pj + 1, 0);
        for (size_t q = 0; q < idx.size(); q++) {
            i64 t = idx[q];
            perm[k.t0 + q] = t;
            h_tptr[b][d->term_p[t] + 1]++;
        }
        for (int p = 0; p < Pj; p++) h_tptr[b][p + 1] += h_tptr[b][p];
        for (i64 t = k.t0; t < k.t1; t++) {
            int r = d->term_r[t], s = d->term_s[t];
            // A_Y[t] = bpY[r,s][pointers_left[r][(s,p,k)], pointers_right[s][(r,p,k)]]  (src/solver.jl:1163)
            i64 gl = k.offL[r] + lidx[t], gr = k.offR[s] + ridx[partner[t]];
            if (g_cfg_pairing_tri && k.sym && k.m == 1 && k.ULt == k.URt && gl < gr) std::swap(gl, gr);   // only the lower triangle of the symmetric GY is formed
            h_ayidx[t] = k.g_off + (i64)k.ULt * k.URt + gl + gr * k.ULt;
        }
    }
    std::vector<int> s_tL(T), s_tR(T);
    std::vector<double> s_tlam(T);
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        if (k.kind != 0) continue;
        for (i64 q = k.t0; q < k.t1; q++) {
            i64 t = perm[q];
            s_tL[q] = k.offL[d->term_s[t]] + lidx[partner[t]];   // pointers_left[s][(r,p,k)]
            s_tR[q] = k.offR[d->term_r[t]] + ridx[t];            // pointers_right[r][(s,p,k)]
            s_tlam[q] = d->term_lambda[t];
        }
    }
    int *d_tL, *d_tR; double *d_tlam;
    CK(upload(c, s_tL, &d_tL)); CK(upload(c, s_tR, &d_tR)); CK(upload(c, s_tlam, &d_tlam));
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        if (k.kind == 0) {
            if (k.t1 == k.t0) continue;
            for (int &v : h_tptr[b]) v += (int)k.t0;   // absolute positions in the sorted term arrays
            CK(upload(c, h_tptr[b], &k.d_tptr));
        } else if (k.cnt > 0) {
            std::vector<int> dp(d->dense_p + k.d0, d->dense_p + k.d1);
            CK(upload(c, dp, &k.d_tptr));
        }
    }

    // ---- which clusters does the fused kernel take?  (everything of the cluster must fit in LDS) ----
    c->cluster_fused.assign(J, 0);
    // ---- k_cluster_assemble_w5: clusters whose PSD blocks are all 2 x 2 blocks of 16 x 16 sub-blocks with constraint matrices E_rs (x) v_u v_u^T on the SAME
    //      U <= 32 vectors in (0,0), (1,1) and as the two terms (0,1) + (1,0): 3 U constraints, every (pair, u) once (Nsphere_packing; clrs_assemble_w5.hip.h) ----
    std::vector<W3Block> w5b;
    std::vector<int> w5_cl_blk0, w5_ay;
    std::vector<double> w5_lam, w5_vop;
    if (g_cfg_fused_assemble && g_cfg_wave_assemble && g_cfg_wave5_assemble) {
        int b = 0;
        for (int j = 0; j < J; j++) {
            const int b_first = b;
            int b_end = b;
            while (b_end < NB && c->blk[b_end].j == j) b_end++;
            b = b_end;
            const int Pj = c->P[j];
            if (Pj < 3 || Pj % 3 != 0 || Pj > 96) continue;
            const int U = Pj / 3;
            bool ok = true;
            int nlr = 0;
            std::vector<int> pm0;
            struct Blk5 { std::vector<int> ay; std::vector<double> lam; };
            std::vector<Blk5> b5;
            for (int bb = b_first; bb < b_end && ok; bb++) {
                const BlockInfo &k = c->blk[bb];
                if (k.kind != 0) { if (k.cnt > 0) ok = false; continue; }
                const int Tn = (int)(k.t1 - k.t0);
                if (Tn == 0) continue;
                if (k.m != 2 || k.delta != 16 || !k.sym || Tn != 4 * U || (int)k.UR.size() != 2 || k.UR[0] != U || k.UR[1] != U) { ok = false; break; }
                // the same vectors in both sub-blocks
                const double *V = h_static.data() + k.zr_off;
                for (int u = 0; u < U && ok; u++)
                    for (int i2 = 0; i2 < 16; i2++)
                        if (V[i2 + (i64)(k.offR[0] + u) * k.n] != V[16 + i2 + (i64)(k.offR[1] + u) * k.n]) { ok = false; break; }
                if (!ok) break;
                Blk5 e;
                e.ay.assign(4 * 32, -1);
                e.lam.assign(3 * U, 0.0);
                std::vector<int> pm(3 * U, -1), seen_terms(4 * U, 0);
                for (i64 q = k.t0; q < k.t1 && ok; q++) {
                    const i64 t = perm[q];
                    const int r = d->term_r[t], s2 = d->term_s[t];
                    if (r < 0 || r > 1 || s2 < 0 || s2 > 1) { ok = false; break; }
                    const int uR = s_tR[q] - k.offR[r], uL = s_tL[q] - k.offL[s2];
                    if (uR != uL || uR < 0 || uR >= U) { ok = false; break; }
                    const int pair = r + s2;                              // 0: (0,0); 1: (0,1) and (1,0); 2: (1,1)
                    const int vi = pair * U + uR, ti2 = (2 * r + s2) * U + uR;
                    if (seen_terms[ti2]++) { ok = false; break; }
                    const int pc = d->term_p[t];
                    const double lm = d->term_lambda[t];
                    if (pm[vi] < 0) { pm[vi] = pc; e.lam[vi] = lm; }
                    else if (pm[vi] != pc || e.lam[vi] != lm) { ok = false; break; }      // the two terms of an off-diagonal pair: one constraint, one lambda
                    e.ay[(2 * r + s2) * 32 + uR] = (int)t;
                }
                if (!ok) break;
                for (int i2 = 0; i2 < 4 * U; i2++) if (seen_terms[i2] != 1) ok = false;
                std::vector<int> seen(pm);
                std::sort(seen.begin(), seen.end());
                for (int i2 = 0; i2 < 3 * U && ok; i2++) if (seen[i2] != i2) ok = false;      // a permutation of the cluster's constraints
                if (!ok) break;
                if (pm0.empty()) pm0 = pm;
                else if (pm != pm0) { ok = false; break; }
                b5.push_back(e);
                nlr++;
            }
            if (!ok || nlr == 0) continue;
            bool ident = true;
            for (int u = 0; u < 3 * U; u++) if (pm0[u] != u) ident = false;
            if (!ident) continue;                          // (constraints in pair-major vector order only: a table lookup per stored entry costs the kernel its registers)
            w5_cl_blk0.push_back((int)w5b.size());
            int ilr = 0;
            for (int bb = b_first; bb < b_end; bb++) {
                const BlockInfo &k = c->blk[bb];
                if (k.kind != 0 || k.t1 == k.t0) continue;
                W3Block k5;
                std::memset(&k5, 0, sizeof(k5));
                k5.xyoff = k.xyoff; k5.n = k.n; k5.U = U; k5.pmap_identity = 1;
                k5.ndense = ilr == 0 ? 1 : 0;              // (first block of its cluster: it stores S_j, the others add to it)
                k5.S_off = c->Soff[j];                     // (every block of the cluster writes S_j: it accumulates in memory)
                k5.lam_off = (int)w5_lam.size();
                w5_lam.insert(w5_lam.end(), b5[ilr].lam.begin(), b5[ilr].lam.end());
                k5.ay_base = (int)w5_ay.size();
                w5_ay.insert(w5_ay.end(), b5[ilr].ay.begin(), b5[ilr].ay.end());
                ilr++;
                k5.vop_off = (int)(w5_vop.size() / 512);
                w5_vop.resize(w5_vop.size() + 512, 0.0);
                double *dst = w5_vop.data() + (size_t)k5.vop_off * 512;
                const double *V = h_static.data() + k.zr_off;
                for (int t = 0; t < 2; t++)
                    for (int q = 0; q < 4; q++)
                        for (int ln = 0; ln < 64; ln++) {
                            const int row = 4 * q + (ln >> 4), col = 16 * t + (ln & 15);
                            if (col < U) dst[((t * 2 + (q >> 1)) * 64 + ln) * 2 + (q & 1)] = V[row + (i64)(k.offR[0] + col) * k.n];
                        }
                w5b.push_back(k5);
            }
            W3Block &kl = w5b.back();
            kl.last = 1; kl.S_off = c->Soff[j];
            c->cluster_fused[j] = 5;
            for (int bb = b_first; bb < b_end; bb++) c->blk[bb].fused = true;
        }
    }
    c->n_wave5_clusters = (int)w5_cl_blk0.size();
    // ---- k_cluster_assemble_w4: clusters whose low-rank blocks are all "simple" (one sub-block, rank-1 symmetric terms, one term per constraint, U = P, the
    //      same constraint order in every block) with n <= 32 and P <= 64, dense blocks 1 x 1, and at least one block beyond the reach of
    //      k_cluster_assemble_w3 (n > 16 or P > 32): the register-resident kernel of clrs_assemble_w4.hip.h ----
    std::vector<W3Block> w4b;
    std::vector<W3Dense> w4d;
    std::vector<int> w4_cl_blk0, w4_pmap, w4_ay;
    std::vector<double> w4_lam, w4_vop;
    int w4_nu = 0;
    if (g_cfg_fused_assemble && g_cfg_wave_assemble && g_cfg_wave4_assemble) {
        int b = 0;
        for (int j = 0; j < J; j++) {
            const int b_first = b;
            int b_end = b;
            while (b_end < NB && c->blk[b_end].j == j) b_end++;
            b = b_end;
            const int Pj = c->P[j];
            if (Pj < 1 || Pj > 64 || c->cluster_fused[j] != 0) continue;
            bool ok = true, beyond_w3 = Pj > 32;
            int nlr = 0;
            std::vector<int> pm0;
            std::vector<std::vector<int>> ays;
            std::vector<std::vector<double>> lams;
            for (int bb = b_first; bb < b_end && ok; bb++) {
                const BlockInfo &k = c->blk[bb];
                if (k.kind == 0) {
                    const int Tn = (int)(k.t1 - k.t0);
                    if (Tn == 0) continue;
                    if (k.m != 1 || k.n > 32 || !k.sym || k.URt != Tn || Tn != Pj) { ok = false; break; }
                    std::vector<int> pm(Tn, -1), ay(Tn, -1);
                    std::vector<double> lam(Tn, 0.0);
                    for (i64 q = k.t0; q < k.t1 && ok; q++) {
                        const i64 t = perm[q];
                        const int u = s_tR[q];
                        if (s_tL[q] != u || u < 0 || u >= Tn || pm[u] >= 0) { ok = false; break; }
                        pm[u] = d->term_p[t]; lam[u] = d->term_lambda[t]; ay[u] = (int)t;
                    }
                    if (!ok) break;
                    std::vector<int> seen(pm);
                    std::sort(seen.begin(), seen.end());
                    for (int i2 = 0; i2 < Tn; i2++) if (seen[i2] != i2) ok = false;      // a permutation of the cluster's constraints
                    if (!ok) break;
                    if (pm0.empty()) pm0 = pm;
                    else if (pm != pm0) { ok = false; break; }
                    if (k.n > 16) beyond_w3 = true;
                    ays.push_back(ay); lams.push_back(lam);
                    nlr++;
                } else if (k.cnt > 0 && k.n != 1) ok = false;
            }
            if (!ok || nlr == 0 || !beyond_w3) continue;
            std::vector<int> inv(Pj, -1);
            for (int u = 0; u < Pj; u++) inv[pm0[u]] = u;
            const int nu = (Pj + 15) / 16 <= 3 ? 3 : 4;
            w4_nu = std::max(w4_nu, nu);
            w4_cl_blk0.push_back((int)w4b.size());
            const int pm_off = (int)w4_pmap.size();
            w4_pmap.insert(w4_pmap.end(), pm0.begin(), pm0.end());
            bool ident = true;
            for (int u = 0; u < Pj; u++) if (pm0[u] != u) ident = false;
            const int dense0 = (int)w4d.size();
            size_t last_lr = 0;
            int ilr = 0;
            for (int bb = b_first; bb < b_end; bb++) {
                const BlockInfo &k = c->blk[bb];
                if (k.kind != 0) {
                    if (k.cnt == 0) continue;
                    W3Dense de; de.xyoff = k.xyoff; de.lam_off = (int)w4_lam.size(); de.pad = 0;
                    std::vector<double> a(Pj, 0.0);
                    for (i64 e = k.d0; e < k.d1; e++) a[inv[d->dense_p[e]]] = d->dense_A[d->dense_A_ptr[e]];
                    w4_lam.insert(w4_lam.end(), a.begin(), a.end());
                    w4d.push_back(de);
                    continue;
                }
                if (k.t1 == k.t0) continue;
                W3Block k4;
                std::memset(&k4, 0, sizeof(k4));
                k4.xyoff = k.xyoff; k4.n = k.n; k4.U = Pj; k4.pmap_off = pm_off; k4.pmap_identity = ident ? 1 : 0;
                k4.lam_off = (int)w4_lam.size();
                w4_lam.insert(w4_lam.end(), lams[ilr].begin(), lams[ilr].end());
                k4.ay_base = (int)w4_ay.size();
                w4_ay.insert(w4_ay.end(), ays[ilr].begin(), ays[ilr].end());
                ilr++;
                k4.vop_off = (int)(w4_vop.size() / 512);
                w4_vop.resize(w4_vop.size() + (size_t)4 * 512, 0.0);      // (four column tiles for every block: the launch's NU is the largest of its clusters')
                double *dst = w4_vop.data() + (size_t)k4.vop_off * 512;
                const double *V = h_static.data() + k.zr_off;
                for (int t = 0; t < nu; t++)
                    for (int q = 0; q < 8; q++)
                        for (int ln = 0; ln < 64; ln++) {
                            const int row = 4 * q + (ln >> 4), col = 16 * t + (ln & 15);
                            if (row < k.n && col < Pj) dst[((t * 4 + (q >> 1)) * 64 + ln) * 2 + (q & 1)] = V[row + (i64)col * k.n];
                        }
                last_lr = w4b.size();
                w4b.push_back(k4);
            }
            W3Block &kl = w4b[last_lr];
            kl.last = 1; kl.S_off = c->Soff[j];
            kl.ndense = (int)w4d.size() - dense0; kl.dense0 = dense0;
            c->cluster_fused[j] = 4;
            for (int bb = b_first; bb < b_end; bb++) c->blk[bb].fused = true;
        }
    }
    c->n_wave4_clusters = (int)w4_cl_blk0.size();
    // ---- wave-per-block assembly (k_cluster_assemble_w1): clusters made of "simple" rank-1 blocks with n <= 16 and small dense blocks ----
    std::vector<WCluster> wcl;
    std::vector<WBlock> wbl;
    std::vector<int> w_pmap, w_ay;
    std::vector<double> w_lam;
    std::vector<size_t> w_ioff;       // per WBlock: offset of its pmap / lam / ay entries in the packed arrays
    std::vector<size_t> w_cl_first;   // per WCluster: its first WBlock
    std::vector<W2Cluster> w2cl;      // cluster-per-wave kernel
    std::vector<W2Block> w2bl;
    std::vector<int> w2_pmap, w2_ay;
    std::vector<double> w2_lam;
    std::vector<size_t> w2_cl_pm, w2_boff, w2_bl_cl;
    int w2_ut = 0;
    int w_ut = 0, w_nwaves = 8, w_maxblocks = 0;
    size_t w_cluster_doubles = 0;
    if (g_cfg_fused_assemble && g_cfg_wave_assemble) {
        int b = 0;
        for (int j = 0; j < J; j++) {
            const int b_first = b;
            bool ok = c->cluster_fused[j] == 0;      // (not taken by k_cluster_assemble_w4)
            int work = 0, ut = 1, nblk = 0;
            std::vector<WBlock> mine;
            std::vector<size_t> mine_off;
            const size_t keep_pm = w_pmap.size(), keep_lam = w_lam.size(), keep_ay = w_ay.size();
            for (; b < NB && c->blk[b].j == j; b++) {
                BlockInfo &k = c->blk[b];
                if (!ok) continue;
                WBlock wb;
                std::memset(&wb, 0, sizeof(wb));
                wb.kind = k.kind; wb.n = k.n; wb.xyoff = k.xyoff;
                if (k.kind == 0) {
                    const int Tn = (int)(k.t1 - k.t0);
                    if (Tn == 0) continue;
                    if (k.m != 1 || k.n > 16 || !k.sym || k.URt != Tn || Tn > 128) { ok = false; continue; }
                    std::vector<int> pm(Tn, -1), ay(Tn, -1);
                    std::vector<double> lam(Tn, 0.0);
                    for (i64 q = k.t0; q < k.t1 && ok; q++) {
                        const i64 t = perm[q];
                        const int u = s_tR[q];
                        if (s_tL[q] != u || u < 0 || u >= Tn || pm[u] >= 0) { ok = false; break; }
                        pm[u] = d->term_p[t]; lam[u] = d->term_lambda[t]; ay[u] = (int)t;
                    }
                    if (ok) {   // one term per constraint
                        std::vector<int> seen(pm);
                        std::sort(seen.begin(), seen.end());
                        for (int i2 = 1; i2 < Tn; i2++) if (seen[i2] == seen[i2 - 1]) ok = false;
                    }
                    if (!ok) continue;
                    wb.U = Tn; wb.v_off = k.zr_off;
                    mine_off.push_back(w_pmap.size());
                    w_pmap.insert(w_pmap.end(), pm.begin(), pm.end());
                    w_ay.insert(w_ay.end(), ay.begin(), ay.end());
                    w_lam.resize(w_pmap.size(), 0.0);
                    std::copy(lam.begin(), lam.end(), w_lam.end() - Tn);
                    ut = std::max(ut, (Tn + 15) / 16);
                } else {
                    if (k.cnt == 0) continue;
                    const int need = 2 * k.n * k.n + 3 * k.cnt * k.n * k.n;
                    if (k.n > 16 || need > 4096) { ok = false; continue; }
                    wb.U = k.cnt; wb.v_off = k.w_off;
                    mine_off.push_back(w_pmap.size());
                    for (i64 e = k.d0; e < k.d1; e++) { w_pmap.push_back(d->dense_p[e]); w_ay.push_back(0); }
                    w_lam.resize(w_pmap.size(), 0.0);
                    work = std::max(work, need);
                }
                mine.push_back(wb);
                nblk++;
            }
            const int utr = ut <= 2 ? 2 : ut <= 4 ? 4 : 8;
            work = std::max(work, 2 * 17 * 16 * utr);
            const i64 per_wave = (i64)c->P[j] * (c->P[j] | 1) + work;     // slab (odd leading dimension) + work area
            if (!ok || mine.empty() || per_wave > LDS_BUDGET_DOUBLES) {
                w_pmap.resize(keep_pm); w_lam.resize(keep_lam); w_ay.resize(keep_ay);
                continue;
            }
            // cluster-per-wave kernel: every low-rank block uses the same constraint order (U = P) and the dense blocks are 1 x 1
            // ("wave2_assemble" = 1: automatic, 2: always, 0: never.  Automatic: always when the register-resident kernel
            // k_cluster_assemble_w3 applies (U <= 32) -- it is also the fastest form for a handful of clusters, 8 us against 12.5 us of
            // the wave-per-block kernel on cohnelkies(8,15) -- and from 64 clusters on for the LDS-staged k_cluster_assemble_w2,
            // which walks the blocks of a cluster one after the other and needs enough clusters to fill the chip)
            if ((g_cfg_wave2_assemble == 2 || (g_cfg_wave2_assemble == 1 && (J >= 64 || (g_cfg_wave3_assemble && utr <= 2)))) && utr <= 4) {
                const int Pj = c->P[j];
                bool w2 = true;
                const int *pm0 = nullptr;
                for (size_t i2 = 0; i2 < mine.size() && w2; i2++) {
                    const WBlock &wb = mine[i2];
                    if (wb.kind == 0) {
                        const int *pm = w_pmap.data() + mine_off[i2];
                        if (wb.U != Pj) w2 = false;
                        else if (!pm0) pm0 = pm;
                        else if (std::memcmp(pm, pm0, sizeof(int) * Pj) != 0) w2 = false;
                    } else if (wb.n != 1) w2 = false;
                }
                if (w2 && pm0) {
                    std::vector<int> inv(Pj, -1);
                    for (int u = 0; u < Pj; u++) inv[pm0[u]] = u;
                    W2Cluster wc2;
                    std::memset(&wc2, 0, sizeof(wc2));
                    wc2.S = c->d_S + c->Soff[j]; wc2.P = Pj; wc2.nblk = (int)mine.size(); wc2.blk0 = (i64)w2bl.size();
                    w2_cl_pm.push_back(w2_pmap.size());
                    w2_pmap.insert(w2_pmap.end(), pm0, pm0 + Pj);
                    int bsel = b_first;
                    for (size_t i2 = 0; i2 < mine.size(); i2++) {
                        const WBlock &wb = mine[i2];
                        W2Block q2;
                        std::memset(&q2, 0, sizeof(q2));
                        q2.kind = wb.kind; q2.n = wb.n; q2.xyoff = wb.xyoff; q2.v_off = wb.v_off;
                        w2_boff.push_back(w2_lam.size());
                        w2_bl_cl.push_back(w2cl.size());
                        if (wb.kind == 0) {
                            w2_lam.insert(w2_lam.end(), w_lam.begin() + mine_off[i2], w_lam.begin() + mine_off[i2] + Pj);
                            w2_ay.insert(w2_ay.end(), w_ay.begin() + mine_off[i2], w_ay.begin() + mine_off[i2] + Pj);
                        } else {
                            // locate the BlockInfo of this dense block to read its 1 x 1 matrices
                            while (!(c->blk[bsel].kind != 0 && c->blk[bsel].xyoff == wb.xyoff)) bsel++;
                            const BlockInfo &kb = c->blk[bsel];
                            std::vector<double> a(Pj, 0.0);
                            for (i64 e = kb.d0; e < kb.d1; e++) a[inv[d->dense_p[e]]] = d->dense_A[d->dense_A_ptr[e]];
                            w2_lam.insert(w2_lam.end(), a.begin(), a.end());
                            w2_ay.insert(w2_ay.end(), Pj, 0);
                        }
                        w2bl.push_back(q2);
                    }
                    w2cl.push_back(wc2);
                    w2_ut = std::max(w2_ut, utr);
                    c->cluster_fused[j] = 3;
                    for (int bb = b_first; bb < b; bb++) c->blk[bb].fused = true;
                    w_pmap.resize(keep_pm); w_lam.resize(keep_lam); w_ay.resize(keep_ay);
                    continue;
                }
            }
            WCluster wc;
            std::memset(&wc, 0, sizeof(wc));
            wc.S = c->d_S + c->Soff[j]; wc.P = c->P[j]; wc.nblk = (int)mine.size(); wc.work_doubles = work;
            w_cl_first.push_back(wbl.size());
            for (size_t i2 = 0; i2 < mine.size(); i2++) { wbl.push_back(mine[i2]); w_ioff.push_back(mine_off[i2]); }
            wcl.push_back(wc);
            c->cluster_fused[j] = 2;
            for (int bb = b_first; bb < b; bb++) c->blk[bb].fused = true;
            w_ut = std::max(w_ut, utr);
            w_nwaves = std::min(w_nwaves, (int)(LDS_BUDGET_DOUBLES / per_wave));
            w_maxblocks = std::max(w_maxblocks, nblk);
            w_cluster_doubles = std::max(w_cluster_doubles, (size_t)per_wave);
        }
        w_nwaves = std::max(1, std::min(w_nwaves, std::max(w_maxblocks, 1)));
        for (WCluster &wc : wcl) wc.work_doubles = (int)(w_cluster_doubles - (size_t)wc.P * (wc.P | 1));   // one slab + work stride for all clusters
        // a cluster whose own P^2 + work is smaller still gets the common stride; re-check the budget with it
        if ((i64)w_nwaves * (i64)w_cluster_doubles > LDS_BUDGET_DOUBLES) w_nwaves = std::max(1, (int)(LDS_BUDGET_DOUBLES / (i64)w_cluster_doubles));
    }
    c->n_wave_clusters = (int)wcl.size();
    c->n_wave2_clusters = (int)w2cl.size();
    std::vector<FCluster> fcl;
    std::vector<int> fcl_cluster;
    std::vector<SSlabSum> fsum;
    std::vector<FBlock> fbl;
    int fused_nmax = 0;
    size_t fused_lds = 0;
    if (g_cfg_fused_assemble) {
        int b = 0;
        for (int j = 0; j < J; j++) {
            const int b_first = b;
            int szL = 0, szY = 0, szV = 0, szTY = 0, szZL = 0, szG = 0, szTab = 0, nmax = 0;
            bool ok = c->cluster_fused[j] == 0;      // not already taken by the wave-per-block kernel
            std::vector<FBlock> mine;
            for (; b < NB && c->blk[b].j == j; b++) {
                BlockInfo &k = c->blk[b];
                if (!ok) continue;
                FBlock fb;
                std::memset(&fb, 0, sizeof(fb));
                fb.kind = k.kind; fb.n = k.n; fb.xyoff = k.xyoff;
                const int n = k.n, n16 = (n + 15) & ~15;
                if (k.kind == 0) {
                    if (k.t1 == k.t0) continue;
                    if (n > 64) { ok = false; continue; }
                    const int UR16 = (k.URt + 15) & ~15, UL16 = (k.ULt + 15) & ~15, Tn = (int)(k.t1 - k.t0);
                    fb.URt = k.URt; fb.ULt = k.ULt; fb.sym = k.sym ? 1 : 0; fb.T = Tn;
                    fb.ldn = n16 + 2;                    // >= ceil16(n) rows (zero padded), ldn % 4 == 2: conflict-free ds_read_b64 of the MFMA operands
                    fb.ldg = k.ULt | 1;
                    fb.v_off = k.zr_off; fb.w_off = k.zl_off; fb.t0 = k.t0; fb.tptr = k.d_tptr;
                    szL = std::max(szL, fb.ldn * n16); szY = std::max(szY, fb.ldn * n16);
                    szV = std::max(szV, fb.ldn * UR16); szTY = std::max(szTY, fb.ldn * UR16);
                    if (!k.sym) szZL = std::max(szZL, fb.ldn * UL16);
                    szG = std::max(szG, fb.ldg * k.URt);
                    szTab = std::max(szTab, ((c->P[j] + 1 + 2 * Tn + 1) & ~1) / 2 + Tn);
                    nmax = std::max(nmax, n);
                } else {
                    if (k.cnt == 0) continue;
                    if (n > 16) { ok = false; continue; }
                    fb.T = k.cnt; fb.ldn = n | 1; fb.v_off = k.w_off; fb.tptr = k.d_tptr;
                    const int st = k.cnt * n * n;
                    szL = std::max(szL, fb.ldn * n); szY = std::max(szY, fb.ldn * n);
                    szV = std::max(szV, st); szTY = std::max(szTY, st); szG = std::max(szG, st);
                    szTab = std::max(szTab, (k.cnt + 1) / 2);
                }
                mine.push_back(fb);
            }
            if (!ok || mine.empty()) continue;
            FCluster fc;
            std::memset(&fc, 0, sizeof(fc));
            fc.S = c->d_S + c->Soff[j]; fc.P = c->P[j];
            auto even = [](int v) { return (v + 1) & ~1; };   // keep every region 16-byte aligned
            int o = 0;
            fc.oL = o; o += even(szL); fc.oY = o; o += even(szY); fc.oV = o; o += even(szV); fc.oTY = o; o += even(szTY);
            fc.oZL = o; o += even(szZL); fc.oGX = o; o += even(szG); fc.oGY = o; o += even(szG); fc.oTab = o; o += even(szTab);
            if (o > LDS_BUDGET_DOUBLES) continue;
            const int PP = c->P[j] * c->P[j];
            fc.oS = o;
            if (o + PP <= LDS_BUDGET_DOUBLES) { fc.s_in_lds = 1; o += even(PP); }
            fc.lds_doubles = o;
            fc.b0 = (int)fbl.size();
            for (const FBlock &fb : mine) fbl.push_back(fb);
            fc.b1 = (int)fbl.size();
            fcl.push_back(fc);
            fcl_cluster.push_back(j);
            c->cluster_fused[j] = 1;
            for (int bb = b_first; bb < b; bb++) c->blk[bb].fused = true;
            fused_nmax = std::max(fused_nmax, nmax);
            fused_lds = std::max(fused_lds, (size_t)o * sizeof(double));
        }
    }
    c->n_fused_clusters = (int)fcl.size() + (int)wcl.size() + (int)w2cl.size() + c->n_wave4_clusters + c->n_wave5_clusters;
    // Few clusters with several blocks each: the blocks of a cluster are independent until their contributions meet in S_j.  One
    // workgroup per block (groups of blocks beyond 32 per cluster) writes its contribution as a P x P slab, k_sum_S_slabs adds
    // the slabs in block order -- the same additions in the same order as the one-workgroup form, so S_j is bit-identical.
    if (g_cfg_split_blocks && !fcl.empty() && fcl.size() <= 32) {
        std::vector<FCluster> split;
        std::vector<int> split_sum;                // per entry of `split`: its sum descriptor, or -1
        i64 slab_doubles = 0;
        for (size_t i = 0; i < fcl.size(); i++) {
            const FCluster &fc = fcl[i];
            const int nb = fc.b1 - fc.b0, groups = std::min(nb, 32);
            if (nb < 2) { split.push_back(fc); split_sum.push_back(-1); continue; }
            SSlabSum ss;
            ss.out = fc.S; ss.slabs = nullptr; ss.len = (i64)fc.P * fc.P; ss.nslabs = groups; ss.pad = (int)slab_doubles;      // offset for now (fits: few small clusters)
            for (int g = 0; g < groups; g++) {
                FCluster part = fc;
                part.b0 = fc.b0 + (int)((i64)g * nb / groups);
                part.b1 = fc.b0 + (int)((i64)(g + 1) * nb / groups);
                part.S = nullptr;                       // patched below: slab g of this cluster
                split.push_back(part);
                split_sum.push_back((int)fsum.size());
            }
            slab_doubles += ss.len * groups;
            fsum.push_back(ss);
        }
        if (!fsum.empty()) {
            double *dslabs;
            CK(dmalloc(c, &dslabs, slab_doubles));
            std::vector<int> used(fsum.size(), 0);
            for (size_t i = 0; i < split.size(); i++)
                if (split_sum[i] >= 0) {
                    const int si = split_sum[i];
                    split[i].S = dslabs + fsum[si].pad + fsum[si].len * used[si]++;
                }
            for (SSlabSum &ss : fsum) { ss.slabs = dslabs + ss.pad; ss.pad = 0; }
            fcl.swap(split);
        }
    }
    for (int b = 0; b < NB; b++)
        if (c->blk[b].fused && c->blk[b].kind == 0)
            for (i64 t = c->blk[b].t0; t < c->blk[b].t1; t++) h_ayidx[t] = -1;   // written by the fused kernel
    CK(upload(c, h_ayidx, &c->d_ayidx));
    std::vector<int> h_ayL(T), h_ayR(T);
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        if (k.kind != 0) continue;
        for (i64 t = k.t0; t < k.t1; t++) {
            h_ayL[t] = k.offL[d->term_r[t]] + lidx[t];
            h_ayR[t] = k.offR[d->term_s[t]] + ridx[partner[t]];
        }
    }
    int *d_ayL, *d_ayR;
    CK(upload(c, h_ayL, &d_ayL)); CK(upload(c, h_ayR, &d_ayR));
    c->d_ayL = d_ayL; c->d_ayR = d_ayR;
    c->h_term_p.assign(d->term_p, d->term_p + T); c->h_term_lambda.assign(d->term_lambda, d->term_lambda + T);
    c->h_dense_p.assign(d->dense_p, d->dense_p + c->D);

    // =============================================================================================
    // plan: assemble
    // =============================================================================================
    {
        Plan &pl = c->p_assemble;
        bool need_copy = false;
        for (int b = 0; b < NB; b++) need_copy = need_copy || !c->blk[b].fused;
        c->all_assemble_fused = !need_copy;
        // which of the non-fused blocks still need the work arena (a copy of the static arena that the staged solves overwrite)?
        bool need_work_arena = false;
        std::vector<DBlock> dblocks;
        size_t dense_lds = 0;
        const size_t copy_step_index = pl.steps.size();
        if (need_copy) add_memcpy(pl, c->d_work, c->d_static, sizeof(double) * (size_t)so);
        std::vector<TrsmJob> fwd, bwd;
        std::vector<DenseTBlock> dtb;                       // dense blocks taken by k_dense_T32
        std::vector<DenseTPair> dtp;
        std::vector<GemmDesc> g1, g2;
        bool any_general = false;
        for (int b = 0; b < NB; b++) {
            BlockInfo &k = c->blk[b];
            const double *Lx = c->d_Xc + k.xyoff, *Yb = c->d_Y + k.xyoff;
            const int n = k.n, dl = k.delta;
            if (k.fused) continue;
            any_general = true;
            if (k.kind == 0) {
                need_work_arena = true;
                double *ZR = c->d_work + k.zr_off, *ZL = c->d_work + k.zl_off, *TY = c->d_TY + k.ty_off;
                double *GX = c->d_G + k.g_off, *GY = GX + (i64)k.ULt * k.URt;
                const double *VR = c->d_static + k.zr_off, *WL = c->d_static + k.zl_off;
                for (int r = 0; r < k.m; r++) {
                    const int r0 = r * dl;
                    // columns of sub-block row r are zero above row r0: solve with the trailing triangle only
                    if (k.UR[r] > 0) fwd.push_back(TrsmJob{Lx + r0 + (i64)r0 * n, n, n - r0, ZR + r0 + (i64)k.offR[r] * n, n, k.UR[r]});
                    if (!k.sym && k.UL[r] > 0) fwd.push_back(TrsmJob{Lx + r0 + (i64)r0 * n, n, n - r0, ZL + r0 + (i64)k.offL[r] * n, n, k.UL[r]});
                    // T_Y[:, cols r] = Y[:, r-block] V_r            (src/solver.jl:1125)
                    if (k.UR[r] > 0)
                        g1.push_back(mk_gemm(0, 0, n, k.UR[r], dl, 1.0, Yb + (i64)r0 * n, n, VR + r0 + (i64)k.offR[r] * n, n, 0.0, TY + (i64)k.offR[r] * n, n));
                }
                // W = V in one sub-block: GX = V^T X^-1 V and GY = V^T Y V are symmetric -- lower tiles only, the gather mirrors its reads
                const int tri = g_cfg_pairing_tri && k.sym && k.m == 1 && k.ULt == k.URt ? 1 : 0;
                for (int s = 0; s < k.m; s++)  // GY[rows s, :] = W_s^T T_Y[s-block, :]   (src/solver.jl:1131)
                    if (k.UL[s] > 0 && k.URt > 0)
                        g2.push_back(mk_gemm(1, 0, k.UL[s], k.URt, dl, 1.0, WL + s * dl + (i64)k.offL[s] * n, n, TY + s * dl, n, 0.0, GY + k.offL[s], k.ULt, 1, 0, 0, 0, tri));
                // GX = ZL^T ZR = W^T X^-1 V       (replaces src/solver.jl:1117,1137-1143)
                if (k.ULt > 0 && k.URt > 0) g2.push_back(mk_gemm(1, 0, k.ULt, k.URt, n, 1.0, ZL, n, ZR, n, 0.0, GX, k.ULt, 1, 0, 0, 0, tri));
            } else if (k.cnt > 0) {
                {   // dense block that fits in LDS: one fused launch for all such blocks
                    const size_t n16 = (n + 15) & ~15, msz = (n16 + 2) * n16, need = 2 * msz + n16 + 2 * (size_t)k.cnt * msz + 1024;
                    if (g_cfg_dense_block && g_cfg_fused_assemble && n <= 64 && need <= (size_t)LDS_BUDGET_DOUBLES) {
                        DBlock db;
                        db.n = n; db.cnt = k.cnt; db.xyoff = k.xyoff; db.a_off = k.w_off; db.Sd = c->d_Sd + k.sd_off;
                        dblocks.push_back(db);
                        dense_lds = std::max(dense_lds, need * sizeof(double));
                        continue;
                    }
                }
                if (g_cfg_dense_wave && n <= 32) {      // T_e = X^-1 A_e Y by one wave per matrix (k_dense_T32), then the same Gram GEMM
                    double *TT = c->d_TT + k.tt_off, *Sd = c->d_Sd + k.sd_off;
                    const double *Ast = c->d_static + k.w_off;
                    const int bi = (int)dtb.size();
                    dtb.push_back(DenseTBlock{Lx, Yb, Ast, nullptr, TT, n, k.cnt});
                    for (int e0 = 0; e0 < k.cnt; e0 += DT32_WAVES * DT32_ITER) dtp.push_back(DenseTPair{bi, e0});
                    k.sd_tri = g_cfg_pairing_tri != 0;      // <A_i, X^-1 A_k Y> = tr(A_i X^-1 A_k Y) is symmetric in (i, k): lower tiles, mirrored reads
                    g2.push_back(mk_gemm(1, 0, k.cnt, k.cnt, n * n, 1.0, Ast, n * n, TT, n * n, 0.0, Sd, k.cnt, 1, 0, 0, 0, k.sd_tri ? 1 : 0));           // <A_i, T_k>   (:1102)
                    continue;
                }
                need_work_arena = true;
                double *W = c->d_work + k.w_off, *TT = c->d_TT + k.tt_off, *Sd = c->d_Sd + k.sd_off;
                const double *Ast = c->d_static + k.w_off;
                fwd.push_back(TrsmJob{Lx, n, n, W, n, n * k.cnt});      // X^-1 A_p for all p  (src/solver.jl:1095)
                bwd.push_back(TrsmJob{Lx, n, n, W, n, n * k.cnt});
                g1.push_back(mk_gemm(0, 0, n, n, n, 1.0, W, n, Yb, n, 0.0, TT, n, k.cnt, (i64)n * n, 0, (i64)n * n));  // (X^-1 A_p) Y  (:1097)
                k.sd_tri = g_cfg_pairing_tri != 0;
                g2.push_back(mk_gemm(1, 0, k.cnt, k.cnt, n * n, 1.0, Ast, n * n, TT, n * n, 0.0, Sd, k.cnt, 1, 0, 0, 0, k.sd_tri ? 1 : 0));           // <A_i, T_k>   (:1102)
            }
        }
        if (need_copy && !need_work_arena) pl.steps.erase(pl.steps.begin() + copy_step_index);   // nothing left that overwrites it
        CK(plan_trsm(c, pl, fwd, 0));
        CK(plan_trsm(c, pl, bwd, 1));
        CK(add_gemm_stage(c, pl, g1));
        if (!dtb.empty()) {
            double *linv = nullptr;
            CK(dmalloc(c, &linv, (i64)dtb.size() * 1024));
            for (size_t i = 0; i < dtb.size(); i++) dtb[i].Linv = linv + i * 1024;
            DenseTBlock *dblk; DenseTPair *dpr;
            CK(upload(c, dtb, &dblk));
            CK(upload(c, dtp, &dpr));
            Step s1;
            s1.kind = STEP_TRTRI32; s1.grid = (int)dtb.size(); s1.d0 = dblk;
            pl.steps.push_back(s1);
            Step s2;
            s2.kind = STEP_DENSE_T32; s2.grid = (int)dtp.size(); s2.d0 = dblk; s2.d1 = dpr;
            pl.steps.push_back(s2);
            HIPCK(hipFuncSetAttribute((const void *)k_dense_T32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dense_T32_lds_bytes()));
        }
        CK(add_gemm_stage(c, pl, g2));
        if (!dblocks.empty()) {
            DBlock *ddb;
            CK(upload(c, dblocks, &ddb));
            c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y; c->ftables.stat = c->d_static;
            Step s;
            s.kind = STEP_DENSE_BLOCK; s.grid = (int)dblocks.size(); s.d0 = ddb; s.src = &c->ftables; s.bytes = dense_lds;
            pl.steps.push_back(s);
            if (dense_lds > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_dense_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dense_lds));
        }
        // gather
        std::vector<SClusterDesc> cl(J);
        std::vector<SBlockDesc> bl;
        std::vector<STile> tiles;
        int b = 0;
        for (int j = 0; j < J; j++) {
            cl[j].S = c->d_S + c->Soff[j]; cl[j].P = c->P[j]; cl[j].b0 = (int)bl.size(); cl[j].pad = 0;
            for (; b < NB && c->blk[b].j == j; b++) {
                BlockInfo &k = c->blk[b];
                SBlockDesc sd;
                std::memset(&sd, 0, sizeof(sd));
                sd.kind = k.kind;
                if (k.fused) continue;
                if (k.kind == 0) {
                    if (k.t1 == k.t0) continue;
                    sd.ldg = k.ULt;
                    sd.tri = g_cfg_pairing_tri && k.sym && k.m == 1 && k.ULt == k.URt ? 1 : 0;
                    sd.GX = c->d_G + k.g_off; sd.GY = sd.GX + (i64)k.ULt * k.URt;
                    sd.tptr = k.d_tptr; sd.tL = d_tL; sd.tR = d_tR; sd.tlam = d_tlam;
                } else {
                    if (k.cnt == 0) continue;
                    sd.cnt = k.cnt; sd.Sd = c->d_Sd + k.sd_off; sd.tri = k.sd_tri ? 1 : 0;
                    std::vector<int> inv(c->P[j], -1);
                    for (i64 e = k.d0; e < k.d1; e++) inv[d->dense_p[e]] = (int)(e - k.d0);
                    int *dinv;
                    CK(upload(c, inv, &dinv));
                    sd.inv = dinv;
                }
                bl.push_back(sd);
            }
            cl[j].b1 = (int)bl.size();
            if (c->cluster_fused[j]) continue;
            int nt = (c->P[j] + 15) / 16;
            for (int tj = 0; tj < nt; tj++)
                for (int ti = 0; ti <= tj; ti++) tiles.push_back(STile{j, ti, tj, 0});
        }
        if (!tiles.empty()) {
            Step s;
            s.kind = STEP_GATHER_S;
            s.grid = (int)tiles.size();
            for (const SClusterDesc &cd : cl) s.aux0 = std::max(s.aux0, cd.b1 - cd.b0);      // most blocks in one cluster
            SClusterDesc *dcl; SBlockDesc *dbl; STile *dt;
            CK(upload(c, cl, &dcl)); CK(upload(c, bl, &dbl)); CK(upload(c, tiles, &dt));
            s.d0 = dcl; s.d1 = dbl; s.d2 = dt;
            pl.steps.push_back(s);
        }
        if (T > 0 && any_general) {
            Step s;
            s.kind = STEP_GATHER_SCALAR;
            s.dst = c->d_AY; s.src = c->d_G; s.d0 = c->d_ayidx; s.n = T;
            pl.steps.push_back(s);
        }
        if (!w5b.empty()) {
            int *dcb, *day; W3Block *db5; double *dvop, *dlam;
            CK(upload(c, w5_cl_blk0, &dcb)); CK(upload(c, w5b, &db5)); CK(upload(c, w5_vop, &dvop));
            CK(upload(c, w5_ay, &day)); CK(upload(c, w5_lam, &dlam));
            c->w5tables.Xc = c->d_Xc; c->w5tables.Y = c->d_Y; c->w5tables.S = c->d_S; c->w5tables.AY = c->d_AY;
            c->w5tables.vop = dvop; c->w5tables.lam = dlam; c->w5tables.ay = day;
            c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y;
            int cus = 256, dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            Step s;
            s.kind = STEP_ASSEMBLE_W5;
            s.d0 = dcb; s.d1 = db5; s.src = &c->w5tables; s.n = (i64)w5_cl_blk0.size();
            s.aux0 = (int)std::min<i64>(((i64)w5_cl_blk0.size() + 3) / 4, (i64)cus);
            s.aux1 = (int)w5b.size();
            pl.steps.push_back(s);
        }
        if (!w4b.empty()) {
            if (w4d.empty()) { W3Dense de; std::memset(&de, 0, sizeof(de)); w4d.push_back(de); }
            if (w4_ay.empty()) w4_ay.push_back(0);
            int *dcb, *dpm, *day; W3Block *db4; W3Dense *dd4; double *dvop, *dlam;
            CK(upload(c, w4_cl_blk0, &dcb)); CK(upload(c, w4b, &db4)); CK(upload(c, w4d, &dd4)); CK(upload(c, w4_vop, &dvop));
            CK(upload(c, w4_pmap, &dpm)); CK(upload(c, w4_ay, &day)); CK(upload(c, w4_lam, &dlam));
            c->w4tables.Xc = c->d_Xc; c->w4tables.Y = c->d_Y; c->w4tables.S = c->d_S; c->w4tables.AY = c->d_AY;
            c->w4tables.vop = dvop; c->w4tables.lam = dlam; c->w4tables.pmap = dpm; c->w4tables.ay = day; c->w4tables.dense = dd4;
            c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y;
            int cus = 256, dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            Step s;
            s.kind = STEP_ASSEMBLE_W4;
            s.d0 = dcb; s.d1 = db4; s.src = &c->w4tables; s.n = (i64)w4_cl_blk0.size(); s.nmax = w4_nu;
            int maxP = 1;
            for (const W3Block &kb : w4b) maxP = std::max(maxP, kb.U);
            s.grid = maxP;                                                                   // rows of the staged S_j
            s.bytes = W4_LDS_BYTES(w4_nu <= 3 ? 3 : 4, maxP);
            s.aux0 = (int)std::min<i64>(((i64)w4_cl_blk0.size() + 3) / 4, (i64)cus * W4_WGS_PER_CU(w4_nu <= 3 ? 3 : 4));      // as many workgroups as are resident at once
            s.aux1 = (int)w4b.size();
            pl.steps.push_back(s);
            if (w4_nu <= 3) HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble_w4<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
            else HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble_w4<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
        }
        if (!w2cl.empty()) {
            int *dpm, *day; double *dlam;
            CK(upload(c, w2_pmap, &dpm)); CK(upload(c, w2_ay, &day)); CK(upload(c, w2_lam, &dlam));
            for (size_t i2 = 0; i2 < w2bl.size(); i2++) { w2bl[i2].lam = dlam + w2_boff[i2]; w2bl[i2].ay = day + w2_boff[i2]; }
            for (size_t i2 = 0; i2 < w2cl.size(); i2++) w2cl[i2].pmap = dpm + w2_cl_pm[i2];
            W2Cluster *dwc; W2Block *dwb;
            CK(upload(c, w2cl, &dwc)); CK(upload(c, w2bl, &dwb));
            c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y; c->ftables.stat = c->d_static; c->ftables.AY = c->d_AY;
            Step s;
            s.kind = STEP_ASSEMBLE_W2;
            bool use_w3 = g_cfg_wave3_assemble && w2_ut <= 2;
            // k_cluster_assemble_w3 takes the A_Y position of vector u as ay_base + u (terms of a block consecutive in A_Y, in vector order)
            for (size_t i2 = 0; i2 < w2bl.size() && use_w3; i2++)
                if (w2bl[i2].kind == 0)
                    for (int u = 0; u < w2cl[w2_bl_cl[i2]].P; u++)
                        if (w2_ay[w2_boff[i2] + u] != w2_ay[w2_boff[i2]] + u) { use_w3 = false; break; }
            s.d0 = dwc; s.d1 = dwb; s.src = &c->ftables; s.n = (i64)w2cl.size(); s.nmax = w2_ut;
            bool full = w2_ut <= 2;
            for (size_t i2 = 0; i2 < w2bl.size() && full; i2++) if (w2bl[i2].kind == 0 && w2bl[i2].n != 16) full = false;
            for (size_t i2 = 0; i2 < w2cl.size() && full; i2++) if (w2cl[i2].P != 16 * w2_ut) full = false;
            s.grid = full ? 1 : 0;
            s.bytes = (size_t)4 * 2 * 17 * 16 * w2_ut * sizeof(double);
            if (use_w3) {
                // k_cluster_assemble_w3: flat sequence of the low-rank blocks, vectors in MFMA-operand order, the 1 x 1 dense
                // blocks attached to the last low-rank block of their cluster
                std::vector<W3Block> b3;
                std::vector<W3Dense> d3;
                std::vector<int> cl_blk0;
                std::vector<double> h_vop;
                for (size_t ci = 0; ci < w2cl.size(); ci++) {
                    const W2Cluster &wc = w2cl[ci];
                    cl_blk0.push_back((int)b3.size());
                    const int dense0 = (int)d3.size();
                    size_t last_lr = 0;
                    for (int i2 = 0; i2 < wc.nblk; i2++) {
                        const size_t bi2 = (size_t)wc.blk0 + i2;
                        const W2Block &q2 = w2bl[bi2];
                        if (q2.kind != 0) { W3Dense de; de.xyoff = q2.xyoff; de.lam_off = (int)w2_boff[bi2]; de.pad = 0; d3.push_back(de); continue; }
                        W3Block k3;
                        std::memset(&k3, 0, sizeof(k3));
                        k3.xyoff = q2.xyoff; k3.n = q2.n; k3.U = wc.P; k3.lam_off = (int)w2_boff[bi2]; k3.pmap_off = (int)w2_cl_pm[ci];
                        k3.ay_base = w2_ay[w2_boff[bi2]];
                        k3.vop_off = (int)(h_vop.size() / 512);
                        h_vop.resize(h_vop.size() + 512, 0.0);
                        double *dst = h_vop.data() + (size_t)k3.vop_off * 512;
                        const double *V = h_static.data() + q2.v_off;
                        for (int t = 0; t < 2; t++)
                            for (int q = 0; q < 4; q++)
                                for (int ln = 0; ln < 64; ln++) {
                                    const int row = 4 * q + (ln >> 4), col = 16 * t + (ln & 15);
                                    if (row < q2.n && col < wc.P) dst[((t * 2 + (q >> 1)) * 64 + ln) * 2 + (q & 1)] = V[row + (i64)col * q2.n];
                                }
                        last_lr = b3.size();
                        b3.push_back(k3);
                    }
                    W3Block &kl = b3[last_lr];
                    kl.last = 1; kl.S_off = (i64)(wc.S - c->d_S);
                    kl.pmap_identity = 1;
                    for (int u = 0; u < wc.P; u++) if (w2_pmap[w2_cl_pm[ci] + u] != u) kl.pmap_identity = 0;
                    kl.ndense = (int)d3.size() - dense0; kl.dense0 = dense0;
                    if (kl.ndense > 0) { kl.dxyoff = d3[dense0].xyoff; kl.dlam_off = d3[dense0].lam_off; }
                }
                if (d3.empty()) { W3Dense de; std::memset(&de, 0, sizeof(de)); d3.push_back(de); }
                int *dcb; W3Block *db3; W3Dense *dd3; double *dvop;
                CK(upload(c, cl_blk0, &dcb)); CK(upload(c, b3, &db3)); CK(upload(c, d3, &dd3)); CK(upload(c, h_vop, &dvop));
                c->w3tables.Xc = c->d_Xc; c->w3tables.Y = c->d_Y; c->w3tables.S = c->d_S; c->w3tables.AY = c->d_AY;
                c->w3tables.vop = dvop; c->w3tables.lam = dlam; c->w3tables.pmap = dpm; c->w3tables.dense = dd3;
                // persistent form: as many workgroups as are resident at once, each wave walks a contiguous range of clusters
                int per_cu = 2, cus = 256, dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
                if (full) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cluster_assemble_w3<true>, 256, 0) != hipSuccess) per_cu = 2; }
                else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cluster_assemble_w3<false>, 256, 0) != hipSuccess) per_cu = 2;
                if (const char *e = std::getenv("CLRS_W3_WGS_PER_CU")) per_cu = std::max(1, std::atoi(e));     // diagnostic: waves per SIMD
                if (std::getenv("CLRS_DEBUG")) std::fprintf(stderr, "[clrs] k_cluster_assemble_w3<%s>: %zu clusters, %zu blocks, %d workgroups per CU x %d CUs\n", full ? "full" : "general", w2cl.size(), b3.size(), per_cu, cus);
                s.kind = STEP_ASSEMBLE_W3;
                s.d0 = dcb; s.d1 = db3; s.src = &c->w3tables; s.bytes = 0;
                s.aux0 = (int)std::min<i64>(((i64)w2cl.size() + 3) / 4, (i64)std::max(per_cu, 1) * cus);
                s.aux1 = (int)b3.size();
            }
            pl.steps.push_back(s);
            if (s.bytes > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble_w2<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
        }
        if (!wcl.empty()) {
            int *dpm, *day; double *dlam;
            CK(upload(c, w_pmap, &dpm)); CK(upload(c, w_ay, &day)); CK(upload(c, w_lam, &dlam));
            for (size_t i2 = 0; i2 < wbl.size(); i2++) { wbl[i2].pmap = dpm + w_ioff[i2]; wbl[i2].ay = day + w_ioff[i2]; wbl[i2].lam = dlam + w_ioff[i2]; }
            // block table with a fixed number of slots per cluster (the kernel fetches slot `wave` without knowing the cluster yet)
            std::vector<WBlock> table(wcl.size() * (size_t)w_maxblocks);
            std::memset(table.data(), 0, table.size() * sizeof(WBlock));
            for (size_t ci = 0; ci < wcl.size(); ci++)
                for (int sl = 0; sl < w_maxblocks; sl++)
                    table[ci * w_maxblocks + sl] = wbl[w_cl_first[ci] + (sl < wcl[ci].nblk ? sl : 0)];
            WCluster *dwc; WBlock *dwb;
            CK(upload(c, wcl, &dwc)); CK(upload(c, table, &dwb));
            c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y; c->ftables.stat = c->d_static; c->ftables.AY = c->d_AY;
            if (!c->ftables.stamps) {
                std::vector<unsigned long long> z(64, 0ull);
                unsigned long long *ds;
                CK(upload(c, z, &ds));
                c->ftables.stamps = ds;
            }
            Step s;
            s.kind = STEP_ASSEMBLE_W1;
            s.grid = (int)wcl.size(); s.d0 = dwc; s.d1 = dwb; s.src = &c->ftables; s.n = w_nwaves; s.nmax = w_ut; s.dst = (void *)(intptr_t)w_maxblocks;
            s.bytes = (size_t)w_nwaves * w_cluster_doubles * sizeof(double);
            pl.steps.push_back(s);
            if (s.bytes > 64 * 1024) {
                if (w_ut <= 2) HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble_w1<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
                else if (w_ut <= 4) HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble_w1<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
                else HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble_w1<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
            }
        }
        if (!fcl.empty()) {
            FCluster *dfc; FBlock *dfb;
            CK(upload(c, fcl, &dfc)); CK(upload(c, fbl, &dfb));
            c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y; c->ftables.stat = c->d_static;
            c->ftables.tL = d_tL; c->ftables.tR = d_tR; c->ftables.tlam = d_tlam;
            c->ftables.ayL = d_ayL; c->ftables.ayR = d_ayR; c->ftables.AY = c->d_AY;
            if (!c->ftables.stamps) {
                std::vector<unsigned long long> z(64, 0ull);
                unsigned long long *ds;
                CK(upload(c, z, &ds));
                c->ftables.stamps = ds;
            }
            Step s;
            s.kind = STEP_FUSED_ASSEMBLE;
            s.grid = (int)fcl.size(); s.d0 = dfc; s.d1 = dfb; s.src = &c->ftables; s.bytes = fused_lds; s.nmax = fused_nmax;
            pl.steps.push_back(s);
            if (fused_lds > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_cluster_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
            if (!fsum.empty()) {
                SSlabSum *dss;
                CK(upload(c, fsum, &dss));
                Step s2;
                s2.kind = STEP_SUM_S_SLABS; s2.grid = (int)fsum.size(); s2.d0 = dss; s2.n = 0;
                for (const SSlabSum &ss : fsum) s2.n = std::max<i64>(s2.n, ss.len);
                pl.steps.push_back(s2);
            }
        }
    }
    // =============================================================================================
    // plan: factor (src/solver.jl:1244-1279), split like the reference's timings
    // =============================================================================================
    auto lds_square = [](int n) { const int n16 = (n + 15) & ~15; return (n16 + 2) * n16 + n16; };   // matrix + dinv, in doubles
    {
        int maxP = 0, maxn = 0;
        for (int j = 0; j < J; j++) maxP = std::max(maxP, c->P[j]);
        for (int b = 0; b < NB; b++) maxn = std::max(maxn, c->blk[b].n);
        c->fused_fs = g_cfg_fused_factor && J > 0 && maxP <= 128;
        c->fused_q = g_cfg_fused_factor && N > 0 && N <= 128;
        c->fused_x = g_cfg_fused_factor && NB > 0 && maxn <= 128;
        CK(dmalloc(c, &c->d_dinvS, c->xlen)); CK(dmalloc(c, &c->d_dinvQ, N));
    }
    bool q_from_factor = false;
    {
        if (c->fused_fs) {
            std::vector<CFactor> cf(J);
            size_t lds = 0;
            c->q_slabs = c->fused_q && J <= 4096;
            for (int j = 0; j < J; j++) {
                const int P = c->P[j], P16 = (P + 15) & ~15, lda = P16 + 2;
                CFactor &f = cf[j];
                f.S = c->d_S + c->Soff[j]; f.B = c->d_B + c->coff[j]; f.LB = c->d_LB + c->coff[j]; f.dinv = c->d_dinvS + c->coff[j];
                f.P = P; f.N = N; f.ldb = (int)c->xlen; f.code = j + 1;
                const int room = (LDS_BUDGET_DOUBLES - lds_square(P)) / lda;      // columns of B that fit beside L
                f.nc = std::max(1, std::min(std::max(N, 1), room));
                if (f.nc < N || room < ((N + 15) & ~15)) c->q_slabs = false;      // the Gram matrix needs all of LinvB_j (+ tile padding) resident
                lds = std::max(lds, (size_t)(lds_square(P) + lda * (N > 0 ? std::max(f.nc, c->q_slabs ? ((N + 15) & ~15) : 0) : 0)) * sizeof(double));
            }
            if (c->q_slabs) {
                CK(dmalloc(c, &c->d_Qslabs, (i64)J * N * N));
                for (int j = 0; j < J; j++) cf[j].Qslab = c->d_Qslabs + (i64)j * N * N;
            } else
                for (int j = 0; j < J; j++) cf[j].Qslab = nullptr;
            CFactor *dcf;
            CK(upload(c, cf, &dcf));
            Step s;
            s.kind = STEP_CLUSTER_FACTOR; s.grid = J; s.d0 = dcf; s.dst = c->d_info; s.bytes = lds;
            c->p_cholS.steps.push_back(s);      // Cholesky of S_j and L_j^-1 B_j in one launch: the LinvB timing slot stays 0
            if (lds > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_cluster_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        } else if (g_cfg_factor_aug && g_cfg_potrf_levels && J >= 1 && N > 0 && N <= 512 && *std::min_element(c->P.begin(), c->P.end()) > POTRF_NB) {
            // large clusters, a few free variables: L_j, L_j^-1 B_j and the cluster's share of Q from ONE blocked factorisation of
            // [S_j .; B_j^T 0] each (k_chol_pack), all clusters in the same launches; several clusters: the shares summed in cluster order
            if (J > 1) CK(dmalloc(c, &c->d_Qslabs, (i64)J * N * N));
            Step ms; ms.kind = STEP_MEMSET_INFO;
            c->p_cholS.steps.push_back(ms);
            std::vector<PotrfJob> pj;
            std::vector<Step> unpack;
            for (int j = 0; j < J; j++) {
                const int P = c->P[j], P64 = (P + 63) & ~63, na = P64 + N;
                CholAugDesc ad;
                std::memset(&ad, 0, sizeof(ad));
                CK(dmalloc(c, &ad.Aug, (i64)na * na));
                ad.S = c->d_S + c->Soff[j]; ad.B = c->d_B + c->coff[j]; ad.LB = c->d_LB + c->coff[j];
                ad.Q = J > 1 ? c->d_Qslabs + (i64)j * N * N : c->d_Q;
                ad.P = P; ad.P64 = P64; ad.N = N; ad.ldb = (int)c->xlen;
                c->chol_aug.push_back(ad);
                Step ps;
                ps.kind = STEP_CHOL_PACK; ps.n = (i64)c->chol_aug.size() - 1; ps.grid = (int)std::min<i64>(((i64)na * na + 255) / 256, 8192);
                c->p_cholS.steps.push_back(ps);
                PotrfJob job{ad.Aug, na, na, j + 1};
                job.stop = P64 / 64;
                pj.push_back(job);
                Step us = ps;
                us.kind = STEP_CHOL_UNPACK; us.grid = (int)std::min<i64>(((i64)P * P + (i64)P * N + (i64)N * N + 255) / 256, 8192);
                unpack.push_back(us);
            }
            CK(plan_potrf(c, c->p_cholS, pj));
            for (const Step &us : unpack) c->p_cholS.steps.push_back(us);
            q_from_factor = true;
        } else {
            std::vector<PotrfJob> pj;
            std::vector<TrsmJob> tj;
            Step ms; ms.kind = STEP_MEMSET_INFO;
            c->p_cholS.steps.push_back(ms);
            for (int j = 0; j < J; j++) {
                pj.push_back(PotrfJob{c->d_S + c->Soff[j], c->P[j], c->P[j], j + 1});
                if (N > 0) tj.push_back(TrsmJob{c->d_S + c->Soff[j], c->P[j], c->P[j], c->d_LB + c->coff[j], (int)c->xlen, N});
            }
            CK(plan_potrf(c, c->p_cholS, pj));
            if (N > 0) {
                add_memcpy(c->p_linvB, c->d_LB, c->d_B, sizeof(double) * (size_t)(c->xlen * N));
                CK(plan_trsm(c, c->p_linvB, tj, 0));
            }
        }
        if (N > 0) {
            if (q_from_factor) {
                // Q came out of the factorisation of S (k_chol_unpack); several clusters: their shares are added up in cluster order
                if (J > 1) {
                    Step s;
                    s.kind = STEP_SUM_SLABS;
                    c->p_Q.steps.push_back(s);
                }
            } else if (c->q_slabs) {
                Step s;
                s.kind = STEP_SUM_SLABS;
                c->p_Q.steps.push_back(s);
            } else if (N <= 16 && c->xlen >= 512) {
                Step s;
                s.kind = STEP_GRAM_SMALL;   // few free variables, many constraints: one reduction per entry of Q
                c->p_Q.steps.push_back(s);
            } else {
                std::vector<GemmDesc> gq;   // Q = LB^T LB  (vcat + matmul, src/solver.jl:1268-1269)
                gq.push_back(mk_gemm(1, 0, N, N, (int)c->xlen, 1.0, c->d_LB, (int)c->xlen, c->d_LB, (int)c->xlen, 0.0, c->d_Q, N));
                CK(add_gemm_stage(c, c->p_Q, gq));
            }
            if (c->fused_q) {
                std::vector<SmallPotrf> sp(1);
                std::memset(&sp[0], 0, sizeof(SmallPotrf));
                sp[0].in_off = 0; sp[0].out_off = 0; sp[0].dinv = c->d_dinvQ; sp[0].n = N; sp[0].ldin = N; sp[0].ldout = N; sp[0].code = J + 1;
                sp[0].nslabs = 1; sp[0].slab_stride = 0;
                SmallPotrf *dsp;
                CK(upload(c, sp, &dsp));
                Step s;
                s.kind = STEP_SMALL_POTRF; s.grid = 1; s.d0 = dsp; s.dst = c->d_info; s.n = 1; s.bytes = (size_t)lds_square(N) * sizeof(double);
                c->p_cholQ.steps.push_back(s);
                if (c->q_slabs) {   // single-GPU path: sum the slabs while loading Q, no separate reduction launch
                    sp[0].nslabs = J; sp[0].slab_stride = (i64)N * N;
                    CK(upload(c, sp, &dsp));
                    s.d0 = dsp; s.n = 2;
                    c->p_cholQ_slabs.steps.push_back(s);
                }
                if (s.bytes > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_small_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_square(128) * sizeof(double))));
            } else {
                std::vector<PotrfJob> qj;
                qj.push_back(PotrfJob{c->d_Q, N, N, J + 1});
                CK(plan_potrf(c, c->p_cholQ, qj));
            }
        }
    }
    // ONE small cluster: the whole factorisation stage in one launch of one workgroup (measured: 1.6-1.9 us per step saved on
    // polyopt 2d = 40 and delsarte(3,10)).  "factor_small" = 2 also takes 2-4 clusters, one wave per cluster -- slower than
    // k_cluster_factor + k_small_potrf on cohnelkies(8,15) (23.4 us against 12.4 + 8.0): a single wave per 32 x 32 cluster is
    // latency bound on its own dependent chains, the workgroup-per-cluster kernels overlap four waves on them.
    if (g_cfg_factor_small && c->fused_fs && c->fused_q && N > 0 && (J == 1 || (g_cfg_factor_small == 2 && J <= 4)) && !c->p_cholQ_slabs.steps.empty()) {
        bool ok = N <= 64;
        for (int j = 0; j < J; j++) ok = ok && c->P[j] <= 64;
        const size_t need = factor_small_lds_doubles(c->P.data(), J, N) * sizeof(double);
        ok = ok && need <= 150 * 1024;
        if (ok) {
            std::memset(&c->fsmall, 0, sizeof(c->fsmall));
            const double *Ssrc[4], *Bsrc[4];
            for (int j = 0; j < J; j++) {
                c->fsmall.c[j].S = c->d_S + c->Soff[j]; c->fsmall.c[j].LB = c->d_LB + c->coff[j]; c->fsmall.c[j].dinv = c->d_dinvS + c->coff[j];
                c->fsmall.c[j].P = c->P[j]; c->fsmall.c[j].code = j + 1;
                Ssrc[j] = c->d_S + c->Soff[j]; Bsrc[j] = c->d_B + c->coff[j];
            }
            c->fsmall.Q = c->d_Q; c->fsmall.dinvQ = c->d_dinvQ; c->fsmall.J = J; c->fsmall.N = N; c->fsmall.ldb = (int)c->xlen; c->fsmall.codeQ = J + 1;
            ok = factor_small_jobs(c->fsmall_jobs, c->fsmall, Ssrc, Bsrc);
        }
        if (ok) {
            Step s;
            s.kind = STEP_FACTOR_SMALL; s.bytes = need;
            c->p_factor_small.steps.push_back(s);
            if (need > 64 * 1024) {
#define CLRS_FS_ATTR(NJ) HIPCK(hipFuncSetAttribute((const void *)k_factor_small<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
                CLRS_FS_ATTR(8) CLRS_FS_ATTR(16) CLRS_FS_ATTR(24) CLRS_FS_ATTR(32) CLRS_FS_ATTR(48)
#undef CLRS_FS_ATTR
            }
        }
    }
    // =============================================================================================
    // plan: solve (src/solver.jl:1527-1582)
    // =============================================================================================
    {
        std::vector<TrsmJob> tj;
        for (int j = 0; j < J; j++) tj.push_back(TrsmJob{c->d_S + c->Soff[j], c->P[j], c->P[j], c->d_t + c->coff[j], (int)c->xlen, 1});
        CSolve *dcs = nullptr;
        size_t lds_cs = 0;
        if (c->fused_fs) {
            std::vector<CSolve> cs(J);
            for (int j = 0; j < J; j++) {
                cs[j].L = c->d_S + c->Soff[j]; cs[j].dinv = c->d_dinvS + c->coff[j]; cs[j].LB = c->d_LB + c->coff[j];
                cs[j].off = c->coff[j]; cs[j].P = c->P[j]; cs[j].N = N; cs[j].ldb = (int)c->xlen; cs[j].pad = 0;
                lds_cs = std::max(lds_cs, (size_t)(lds_square(c->P[j]) + ((c->P[j] + 15) & ~15) + 2) * sizeof(double));
            }
            CK(upload(c, cs, &dcs));
            if (lds_cs > 64 * 1024) {
                HIPCK(hipFuncSetAttribute((const void *)k_cluster_solve_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cs));
                HIPCK(hipFuncSetAttribute((const void *)k_cluster_solve_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cs));
            }
            Step s;
            s.kind = STEP_CSOLVE_FWD; s.grid = J; s.d0 = dcs; s.bytes = lds_cs;
            c->p_fwd.steps.push_back(s);                                                        // t_j = L_j^-1 rhs_x[j]   (:1538)
        } else {
            add_memcpy(c->p_fwd, c->d_t, c->d_rhsx, sizeof(double) * (size_t)c->xlen);
            CK(plan_trsm(c, c->p_fwd, tj, 0, true));
        }
        if (N > 0) {
            if (c->fused_fs) {
                Step s;
                s.kind = STEP_GEMV_T;                                                           // u = LB^T t             (:1546)
                c->p_fwd.steps.push_back(s);
            } else {
                std::vector<GemmDesc> g;
                g.push_back(mk_gemm(1, 0, N, 1, (int)c->xlen, 1.0, c->d_LB, (int)c->xlen, c->d_t, (int)c->xlen, 0.0, c->d_u, N));
                CK(add_gemm_stage(c, c->p_fwd, g));
            }
            if (c->fused_q) {
                Step s;
                s.kind = STEP_Q_SOLVE; s.bytes = (size_t)(lds_square(N) + ((N + 15) & ~15) + 2) * sizeof(double);   // dy = Q^-1 (rhs_y - u)  (:1550-1558)
                c->p_bwd.steps.push_back(s);
                if (s.bytes > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_q_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
            } else {
                Step s;
                s.kind = STEP_SUB; s.dst = c->d_dy; s.src = c->d_rhsy; s.d0 = c->d_u; s.n = N;
                c->p_bwd.steps.push_back(s);
                std::vector<TrsmJob> qj;
                qj.push_back(TrsmJob{c->d_Q, N, N, c->d_dy, N, 1});
                CK(plan_trsm(c, c->p_bwd, qj, 0));
                CK(plan_trsm(c, c->p_bwd, qj, 1));
            }
            if (!c->fused_fs) {
                std::vector<GemmDesc> g2;                                                       // t += LB dy             (:1568-1569)
                g2.push_back(mk_gemm(0, 0, (int)c->xlen, 1, N, 1.0, c->d_LB, (int)c->xlen, c->d_dy, N, 1.0, c->d_t, (int)c->xlen));
                CK(add_gemm_stage(c, c->p_bwd, g2));
            }
        }
        if (c->fused_fs) {
            Step s;
            s.kind = STEP_CSOLVE_BWD; s.grid = J; s.d0 = dcs; s.bytes = lds_cs;                 // dx_j = L_j^-T (t_j + LB_j dy)  (:1566-1573)
            c->p_bwd.steps.push_back(s);
        } else {
            CK(plan_trsm(c, c->p_bwd, tj, 1, true));
        }
    }
    // a handful of small clusters: the whole solve stage in one workgroup, one launch
    i64 solve_doubles = (i64)N * N + c->xlen * N;                     // operands of one solve: L_j, LinvB, L_Q
    for (int j = 0; j < J; j++) solve_doubles += (i64)c->P[j] * c->P[j];
    if (c->fused_fs && (c->fused_q || N == 0) && J <= 8 && c->xlen <= 1024 && solve_doubles <= g_cfg_solve_small_max) {
        int maxP16 = (N + 15) & ~15;
        for (int j = 0; j < J; j++) maxP16 = std::max(maxP16, (c->P[j] + 15) & ~15);
        Step s = c->p_fwd.steps[0];                                   // reuses the CSolve table
        s.kind = STEP_SOLVE_SMALL; s.n = maxP16;
        s.bytes = (size_t)((maxP16 + 2) * maxP16 + 2 * maxP16 + 2 + ((c->xlen + 15) & ~15) + 2 * ((N + 15) & ~15) + 16) * sizeof(double);
        if (s.bytes > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_solve_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
        const size_t need2 = solve_small2_lds_doubles(c->P.data(), J, N, c->xlen) * sizeof(double);
        bool fits2 = g_cfg_solve_small2 && need2 <= 150 * 1024;
        if (fits2) {
            HIPCK(hipMemcpy(c->solve8.d, s.d0, sizeof(CSolve) * J, hipMemcpyDeviceToHost));
            for (int ph = 0; ph < 3 && fits2; ph++) {
                clrs_ctx::SolveSmall2 &q = c->ss2[ph];
                q.ok = solve_small2_jobs(q.jobs, c->solve8.d, J, c->d_Q, c->d_dinvQ, N, c->xlen, c->d_LB, q.rx_job, q.ry_job, ph, c->d_t, c->d_u);
                fits2 = q.ok;
            }
        }
        if (fits2) {                                                // everything resident at once: the latency-first variant
            Step s1 = s, s2 = s;
            s.aux0 = 2; s.bytes = need2;
            s1.aux0 = 3; s1.bytes = need2; s2.aux0 = 4; s2.bytes = need2;
            c->p_fwd_small.steps.push_back(s1);
            c->p_bwd_small.steps.push_back(s2);
            if (s.bytes > 64 * 1024) {
#define CLRS_SS2_ATTR(NJ)                                                                                                                   \
    HIPCK(hipFuncSetAttribute((const void *)k_solve_small2<NJ, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));               \
    HIPCK(hipFuncSetAttribute((const void *)k_solve_small2<NJ, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));               \
    HIPCK(hipFuncSetAttribute((const void *)k_solve_small2<NJ, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.bytes));
                CLRS_SS2_ATTR(8) CLRS_SS2_ATTR(16) CLRS_SS2_ATTR(24) CLRS_SS2_ATTR(32) CLRS_SS2_ATTR(48)
#undef CLRS_SS2_ATTR
            }
        }
        c->p_solve_all.steps.push_back(s);
    } else
    // single-GPU solve in three launches: the u = LinvB^T t product moves into the Q-solve kernel
    if (c->fused_fs && c->fused_q && c->xlen <= 4096) {
        c->p_solve_all.steps.push_back(c->p_fwd.steps[0]);            // k_cluster_solve_fwd
        Step q = c->p_bwd.steps[0];                                   // k_q_solve
        q.n = 1;
        c->p_solve_all.steps.push_back(q);
        c->p_solve_all.steps.push_back(c->p_bwd.steps.back());        // k_cluster_solve_bwd
    }
    // plan: Cholesky of X blocks (src/solver.jl:388-399)
    {
        if (c->fused_x) {
            std::vector<SmallPotrf> sp(NB);
            size_t lds = 0;
            for (int b = 0; b < NB; b++) {
                std::memset(&sp[b], 0, sizeof(SmallPotrf));
                sp[b].in_off = c->blk[b].xyoff; sp[b].out_off = c->blk[b].xyoff; sp[b].dinv = nullptr;
                sp[b].n = c->blk[b].n; sp[b].ldin = c->blk[b].n; sp[b].ldout = c->blk[b].n; sp[b].code = b + 1; sp[b].nslabs = 1;
                lds = std::max(lds, (size_t)lds_square(c->blk[b].n) * sizeof(double));
            }
            SmallPotrf *dsp;
            CK(upload(c, sp, &dsp));
            Step s;
            s.kind = STEP_SMALL_POTRF; s.grid = NB; s.d0 = dsp; s.dst = c->d_info + 1; s.n = 0; s.bytes = lds;
            c->p_cholX.steps.push_back(s);
            if (lds > 64 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_small_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_square(128) * sizeof(double))));
        } else {
            std::vector<PotrfJob> pj;
            Step ms; ms.kind = STEP_MEMSET_INFO; ms.dst = c->d_info + 1;
            c->p_cholX.steps.push_back(ms);
            for (int b = 0; b < NB; b++) pj.push_back(PotrfJob{c->d_X + c->blk[b].xyoff, c->blk[b].n, c->blk[b].n, b + 1});
            CK(plan_potrf(c, c->p_cholX, pj, c->d_info + 1));
            if (NB > 0) {   // strict upper triangles -> 0, the output format of approx_cholesky! (src/tools.jl:100-105)
                std::vector<PotrfDesc> zd;
                i64 maxnn = 0;
                for (int b = 0; b < NB; b++) {
                    zd.push_back(PotrfDesc{c->d_X + c->blk[b].xyoff, c->blk[b].n, c->blk[b].n, 0, 0});
                    maxnn = std::max(maxnn, (i64)c->blk[b].n * c->blk[b].n);
                }
                PotrfDesc *dz;
                CK(upload(c, zd, &dz));
                Step zs;
                zs.kind = STEP_ZERO_UPPER; zs.grid = NB; zs.d0 = dz; zs.n = maxnn;
                c->p_cholX.steps.push_back(zs);
            }
        }
    }

    // ---- algorithmic work counters ----
    for (int b = 0; b < NB; b++) {
        BlockInfo &k = c->blk[b];
        double n = k.n;
        if (k.kind == 0) {
            double U = 0.5 * (k.URt + k.ULt), m = k.m, dl = k.delta, nt = (double)(k.t1 - k.t0);
            c->cnt_bytes += 8.0 * (2 * n * n + dl * U);
            c->cnt_flops += (m == 1) ? 2.0 * (2 * n * n * U + 2 * n * U * U) + U * U : 2.0 * m * (2 * n * dl * U / m + 2 * U * dl * U / m) + nt * nt;
        } else {
            double Pc = k.cnt;
            c->cnt_bytes += 8.0 * (Pc * n * n + 2 * n * n);
            c->cnt_flops += Pc * 6 * n * n * n + Pc * Pc * n * n;
        }
    }
    for (int j = 0; j < J; j++) {
        double Pj = c->P[j];
        c->cnt_bytes += 8.0 * Pj * Pj;
        c->cnt_factor_flops += Pj * Pj * Pj / 3 + Pj * Pj * N + 2.0 * N * N * Pj;
        c->cnt_solve_flops += 2 * Pj * Pj + 4 * Pj * N;
    }
    c->cnt_factor_flops += (double)N * N * N / 3;
    c->cnt_solve_flops += 2.0 * N * N;
    HIPCK(hipStreamSynchronize(c->stream));
    *out = c;
    return 0;
#undef CK
#undef HIPCK
}

static void ipm_free(clrs_ctx *c);

extern "C" void clrs_ctx_destroy(clrs_ctx *c) {
    if (!c) return;
    ipm_free(c);
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    Plan *plans[] = {&c->p_assemble, &c->p_cholS, &c->p_linvB, &c->p_Q, &c->p_cholQ, &c->p_fwd, &c->p_bwd, &c->p_cholX, &c->p_cholQ_slabs, &c->p_solve_all, &c->p_fwd_small, &c->p_bwd_small, &c->p_factor_small};
    for (Plan *p : plans)
        if (p->graph) hipGraphExecDestroy(p->graph);
    for (void *p : c->allocs) hipFree(p);
    for (int i = 0; i < 10; i++)
        if (c->ev[i]) hipEventDestroy(c->ev[i]);
    for (hipEvent_t e : c->kt_ev) hipEventDestroy(e);
    if (c->h_info) (void)hipHostFree(c->h_info);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int clrs_get_dims(const clrs_ctx *c, clrs_dims *o) {
    if (!c || !o) return fail(CLRS_ERR_INVALID, "null argument");
    o->xy_len = c->xylen; o->x_len = c->xlen; o->S_len = c->Slen; o->n_terms = c->T; o->n_free = c->N;
    o->n_clusters = c->J; o->n_blocks = c->NB; o->reserved = 0;
    return 0;
}

extern "C" int clrs_get_unique_counts(const clrs_ctx *c, int32_t b, int32_t r, int32_t *nr, int32_t *nl) {
    if (!c || b < 0 || b >= c->NB) return fail(CLRS_ERR_INVALID, "block out of range");
    const BlockInfo &k = c->blk[b];
    if (k.kind != 0 || r < 0 || r >= k.m) return fail(CLRS_ERR_INVALID, "not a low-rank sub-block row");
    if (nr) *nr = k.UR[r];
    if (nl) *nl = k.UL[r];
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-iteration drivers
// ------------------------------------------------------------------------------------------------
static size_t PIN_LIMIT = (size_t)64 << 20;            // beyond this the caller's buffers are copied directly (bandwidth, not latency, matters there); "pin_limit" (tests)
// start of a host-pointer call: the arena is empty again; it grows here (never while pointers are handed out) to what the
// previous call asked for in total
static void pin_reset(clrs_ctx *c) {
    c->pin_used = 0;
    c->pin_out.clear();
    if (c->pin_demand > c->pin_cap && c->pin_demand <= PIN_LIMIT) {
        const size_t cap = std::min(PIN_LIMIT, std::max<size_t>((size_t)1 << 20, 2 * c->pin_demand));
        if (c->pin) { (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_cap = 0; }
        if (hipHostMalloc((void **)&c->pin, cap, hipHostMallocDefault) == hipSuccess) c->pin_cap = cap;
        else c->pin = nullptr;
    }
    c->pin_demand = 0;
}
static void *pin_take(clrs_ctx *c, size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    c->pin_demand += bytes;
    if (c->pin_used + bytes > c->pin_cap) return nullptr;      // the caller copies directly; the next call finds a larger arena
    void *p = c->pin + c->pin_used;
    c->pin_used += bytes;
    return p;
}
// host -> device through the arena (falls back to the direct copy when the arena cannot hold it)
static int h2d_staged(clrs_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return 0;
    void *p = pin_take(c, bytes);
    if (p) { std::memcpy(p, src, bytes); src = p; }
    HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}
// device -> host: lands in the arena, handed to the caller by d2h_finish after the stream has been synchronised
static int d2h_staged(clrs_ctx *c, void *user, const void *src, size_t bytes) {
    if (bytes == 0) return 0;
    void *p = pin_take(c, bytes);
    HIPCHECK(hipMemcpyAsync(p ? p : user, src, bytes, hipMemcpyDeviceToHost, c->stream));
    if (p) c->pin_out.push_back(clrs_ctx::PinOut{user, p, bytes});
    return 0;
}
static int d2h_finish(clrs_ctx *c) {
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (const clrs_ctx::PinOut &o : c->pin_out) std::memcpy(o.user, o.pinned, o.bytes);
    c->pin_out.clear();
    return 0;
}

static int read_info(clrs_ctx *c, int *status, int which = 0) {
    if (!c->h_info) HIPCHECK(hipHostMalloc((void **)&c->h_info, 2 * sizeof(int), hipHostMallocDefault));   // pinned: no staging copy behind the 4-byte read
    c->h_info[which] = INFO_NONE;
    HIPCHECK(hipMemcpyAsync(c->h_info + which, c->d_info + which, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const int h = c->h_info[which];
    *status = (h == INFO_NONE) ? 0 : h;
    if (h != INFO_NONE) HIPCHECK(hipMemsetAsync(c->d_info + which, 0x7f, sizeof(int), c->stream));   // re-arm: the status is "since the last query"
    return 0;
}

static int collect_times(clrs_ctx *c) {
    if (c->times_pending) {
        for (int i = 0; i < 5; i++) {
            float ms = 0;
            int a = (i == 0) ? 0 : i + 1, b = (i == 0) ? 1 : i + 2;   // ev0-1: schur; ev2..7: cholS,LinvB,Q,cholQ boundaries
            if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) == hipSuccess) c->times[i] = ms * 1e-3;
        }
        c->times_pending = false;
    }
    if (c->solve_time_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[7], c->ev[8]) == hipSuccess) c->times[5] = ms * 1e-3;
        c->solve_time_pending = false;
    }
    return 0;
}

extern "C" int clrs_schur_assemble_dev(clrs_ctx *c, const double *d_Xchol, const double *d_Y) {
    if (!c || !d_Xchol || !d_Y) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(c->device));
    if (c->all_assemble_fused) {   // the fused kernel reads the caller's buffers directly
        c->ftables.Xc = d_Xchol; c->ftables.Y = d_Y;
    } else {
        c->ftables.Xc = c->d_Xc; c->ftables.Y = c->d_Y;
        if (d_Xchol != c->d_Xc) HIPCHECK(hipMemcpyAsync(c->d_Xc, d_Xchol, sizeof(double) * c->xylen, hipMemcpyDeviceToDevice, c->stream));
        if (d_Y != c->d_Y) HIPCHECK(hipMemcpyAsync(c->d_Y, d_Y, sizeof(double) * c->xylen, hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[0], c->stream));
    int rc = run_plan(c, c->p_assemble);
    if (rc) return rc;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[1], c->stream));
    c->assembled = true; c->factored = false; c->local_factored = false; c->s_inv_valid = false;
    return 0;
}

extern "C" int clrs_schur_assemble(clrs_ctx *c, const double *Xchol, const double *Y, double *S_out, double *AY_out) {
    if (!c || !Xchol || !Y) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(c->device));
    int rc;
    pin_reset(c);
    if ((rc = h2d_staged(c, c->d_Xc, Xchol, sizeof(double) * c->xylen))) return rc;
    if ((rc = h2d_staged(c, c->d_Y, Y, sizeof(double) * c->xylen))) return rc;
    if ((rc = clrs_schur_assemble_dev(c, c->d_Xc, c->d_Y))) return rc;
    if (S_out && (rc = d2h_staged(c, S_out, c->d_S, sizeof(double) * c->Slen))) return rc;
    if (AY_out && c->T > 0 && (rc = d2h_staged(c, AY_out, c->d_AY, sizeof(double) * c->T))) return rc;
    return d2h_finish(c);
}

extern "C" int clrs_schur_factor_local_dev(clrs_ctx *c) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    if (!c->assembled) return fail(CLRS_ERR_STATE, "clrs_schur_factor called before clrs_schur_assemble");
    HIPCHECK(hipSetDevice(c->device));
    int rc;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[2], c->stream));
    c->s_inv_valid = false;
    if ((rc = run_plan(c, c->p_cholS))) return rc;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[3], c->stream));
    if ((rc = run_plan(c, c->p_linvB))) return rc;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[4], c->stream));
    if ((rc = run_plan(c, c->p_Q))) return rc;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[5], c->stream));
    c->assembled = false;  // S now holds L
    c->factored = false;
    c->local_factored = true;      // L_j and LinvB_j are ready: clrs_schur_solve_fwd_dev may run before Q is summed and factored
    return 0;
}

extern "C" int clrs_schur_factor_finish_dev(clrs_ctx *c) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = run_plan(c, c->p_cholQ))) return rc;
    if (c->timing) { HIPCHECK(hipEventRecord(c->ev[6], c->stream)); c->times_pending = true; }
    c->factored = true;
    return 0;
}

extern "C" int clrs_sync_status(clrs_ctx *c) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    int st = 0;
    int rc = read_info(c, &st);
    if (rc) return rc;
    if (c->timing) collect_times(c);
    return st;
}

extern "C" int clrs_sync_status_cholesky(clrs_ctx *c) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    int st = 0;
    int rc = read_info(c, &st, 1);
    return rc ? rc : st;
}

extern "C" int clrs_schur_factor_dev(clrs_ctx *c) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    if (c->p_cholQ_slabs.steps.empty()) {   // no fused single-GPU shortcut: the two split phases back to back
        int rc = clrs_schur_factor_local_dev(c);
        return rc ? rc : clrs_schur_factor_finish_dev(c);
    }
    if (!c->assembled) return fail(CLRS_ERR_STATE, "clrs_schur_factor called before clrs_schur_assemble");
    HIPCHECK(hipSetDevice(c->device));
    int rc;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[2], c->stream));
    if (!c->p_factor_small.steps.empty()) {                           // everything in one launch (k_factor_small)
        if ((rc = run_plan(c, c->p_factor_small))) return rc;
        if (c->timing) { HIPCHECK(hipEventRecord(c->ev[3], c->stream)); HIPCHECK(hipEventRecord(c->ev[4], c->stream)); HIPCHECK(hipEventRecord(c->ev[5], c->stream)); }
    } else {
    c->s_inv_valid = false;
    if ((rc = run_plan(c, c->p_cholS))) return rc;                    // chol S_j, LinvB_j and the Q slabs: one launch
    if (c->timing) { HIPCHECK(hipEventRecord(c->ev[3], c->stream)); HIPCHECK(hipEventRecord(c->ev[4], c->stream)); HIPCHECK(hipEventRecord(c->ev[5], c->stream)); }
    if ((rc = run_plan(c, c->p_cholQ_slabs))) return rc;              // Q = sum of slabs, chol Q: one launch
    }
    if (c->timing) { HIPCHECK(hipEventRecord(c->ev[6], c->stream)); c->times_pending = true; }
    c->assembled = false;
    c->factored = true;
    return 0;
}

extern "C" int clrs_schur_factor(clrs_ctx *c) {
    int rc = clrs_schur_factor_dev(c);
    if (rc) return rc;
    return clrs_sync_status(c);
}

extern "C" int clrs_get_factor(clrs_ctx *c, double *L, double *LinvB, double *LQ) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    if (!c->factored) return fail(CLRS_ERR_STATE, "no factorisation available");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (L) {
        std::vector<double> h((size_t)c->Slen);
        HIPCHECK(hipMemcpy(h.data(), c->d_S, sizeof(double) * c->Slen, hipMemcpyDeviceToHost));
        for (int j = 0; j < c->J; j++) {
            const int P = c->P[j];
            double *Sj = h.data() + c->Soff[j];
            for (int col = 0; col < P; col++)
                for (int r = 0; r < col; r++) Sj[r + (i64)col * P] = 0.0;   // tools.jl:100-105
        }
        std::memcpy(L, h.data(), sizeof(double) * c->Slen);
    }
    if (LinvB && c->N > 0) {
        std::vector<double> h((size_t)(c->xlen * c->N));
        HIPCHECK(hipMemcpy(h.data(), c->d_LB, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
        i64 off = 0;
        for (int j = 0; j < c->J; j++) {
            for (int col = 0; col < c->N; col++)
                for (int r = 0; r < c->P[j]; r++) LinvB[off + r + (i64)col * c->P[j]] = h[(size_t)(c->coff[j] + r + (i64)col * c->xlen)];
            off += (i64)c->P[j] * c->N;
        }
    }
    if (LQ && c->N > 0) {
        HIPCHECK(hipMemcpy(LQ, c->d_Q, sizeof(double) * c->N * c->N, hipMemcpyDeviceToHost));
        for (int col = 0; col < c->N; col++)
            for (int r = 0; r < col; r++) LQ[r + (i64)col * c->N] = 0.0;
    }
    return 0;
}

extern "C" int clrs_schur_solve_fwd_dev(clrs_ctx *c, const double *d_rhs_x) {
    if (!c || !d_rhs_x) return fail(CLRS_ERR_INVALID, "null argument");
    if (!c->factored && !c->local_factored) return fail(CLRS_ERR_STATE, "clrs_schur_solve called before clrs_schur_factor");
    HIPCHECK(hipSetDevice(c->device));
    if (c->fused_fs) c->bind_rhsx = d_rhs_x;
    else if (d_rhs_x != c->d_rhsx) HIPCHECK(hipMemcpyAsync(c->d_rhsx, d_rhs_x, sizeof(double) * c->xlen, hipMemcpyDeviceToDevice, c->stream));
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[7], c->stream));
    const int rc = run_plan(c, c->p_fwd_small.steps.empty() ? c->p_fwd : c->p_fwd_small);
    if (!rc && !c->s_inv.empty()) c->s_inv_valid = true;
    return rc;
}

extern "C" int clrs_schur_solve_bwd_dev(clrs_ctx *c, const double *d_rhs_y, double *d_dx, double *d_dy) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    if (!c->factored) return fail(CLRS_ERR_STATE, "clrs_schur_solve called before clrs_schur_factor");
    HIPCHECK(hipSetDevice(c->device));
    if (c->N > 0 && !d_rhs_y) return fail(CLRS_ERR_INVALID, "rhs_y is required when there are free variables");
    // the fused kernels read / write the caller's buffers; the staged ones work on the context's own
    const bool q_direct = c->fused_q && c->N > 0, x_direct = c->fused_fs;
    c->bind_rhsy = q_direct ? d_rhs_y : c->d_rhsy;
    c->bind_dy = (c->fused_q && d_dy) ? d_dy : c->d_dy;
    c->bind_dx = (x_direct && d_dx) ? d_dx : c->d_dx;
    if (c->N > 0 && !q_direct && d_rhs_y != c->d_rhsy)
        HIPCHECK(hipMemcpyAsync(c->d_rhsy, d_rhs_y, sizeof(double) * c->N, hipMemcpyDeviceToDevice, c->stream));
    int rc = run_plan(c, c->p_bwd_small.steps.empty() ? c->p_bwd : c->p_bwd_small);
    if (rc) return rc;
    if (!c->s_inv.empty()) c->s_inv_valid = true;
    if (c->timing) { HIPCHECK(hipEventRecord(c->ev[8], c->stream)); c->solve_time_pending = true; }
    if (!x_direct && d_dx && d_dx != c->d_t) HIPCHECK(hipMemcpyAsync(d_dx, c->d_t, sizeof(double) * c->xlen, hipMemcpyDeviceToDevice, c->stream));
    if (!c->fused_q && d_dy && c->N > 0 && d_dy != c->d_dy) HIPCHECK(hipMemcpyAsync(d_dy, c->d_dy, sizeof(double) * c->N, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int clrs_schur_solve_dev(clrs_ctx *c, const double *d_rhs_x, const double *d_rhs_y, double *d_dx, double *d_dy) {
    if (!c || !d_rhs_x || !d_dx) return fail(CLRS_ERR_INVALID, "null argument");
    if (c->p_solve_all.steps.empty()) {
        int rc = clrs_schur_solve_fwd_dev(c, d_rhs_x);
        return rc ? rc : clrs_schur_solve_bwd_dev(c, d_rhs_y, d_dx, d_dy);
    }
    if (!c->factored) return fail(CLRS_ERR_STATE, "clrs_schur_solve called before clrs_schur_factor");
    if (c->N > 0 && (!d_rhs_y || !d_dy)) return fail(CLRS_ERR_INVALID, "rhs_y / dy are required when there are free variables");
    HIPCHECK(hipSetDevice(c->device));
    c->bind_rhsx = d_rhs_x; c->bind_rhsy = d_rhs_y; c->bind_dx = d_dx; c->bind_dy = d_dy;
    if (c->timing) HIPCHECK(hipEventRecord(c->ev[7], c->stream));
    int rc = run_plan(c, c->p_solve_all);
    if (rc) return rc;
    if (c->timing) { HIPCHECK(hipEventRecord(c->ev[8], c->stream)); c->solve_time_pending = true; }
    return 0;
}

extern "C" int clrs_schur_solve(clrs_ctx *c, const double *rhs_x, const double *rhs_y, double *dx, double *dy) {
    if (!c || !rhs_x || !dx) return fail(CLRS_ERR_INVALID, "null argument");
    if (!c->factored) return fail(CLRS_ERR_STATE, "clrs_schur_solve called before clrs_schur_factor");
    HIPCHECK(hipSetDevice(c->device));
    int rc;
    pin_reset(c);
    if ((rc = h2d_staged(c, c->d_rhsx, rhs_x, sizeof(double) * c->xlen))) return rc;
    if (c->N > 0) {
        if (!rhs_y || !dy) return fail(CLRS_ERR_INVALID, "null argument");
        if ((rc = h2d_staged(c, c->d_rhsy, rhs_y, sizeof(double) * c->N))) return rc;
    }
    if ((rc = clrs_schur_solve_dev(c, c->d_rhsx, c->d_rhsy, c->d_dx, c->d_dy))) return rc;
    if ((rc = d2h_staged(c, dx, c->d_dx, sizeof(double) * c->xlen))) return rc;
    if (c->N > 0 && (rc = d2h_staged(c, dy, c->d_dy, sizeof(double) * c->N))) return rc;
    return d2h_finish(c);
}

extern "C" int clrs_cholesky_blocks_dev(clrs_ctx *c, const double *d_X, double *d_Xchol) {
    if (!c || !d_X || !d_Xchol) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(c->device));
    if (c->fused_x) {              // one launch, caller's buffers
        c->bind_X = d_X; c->bind_Xchol = d_Xchol;
        return run_plan(c, c->p_cholX);
    }
    if (d_X != c->d_X) HIPCHECK(hipMemcpyAsync(c->d_X, d_X, sizeof(double) * c->xylen, hipMemcpyDeviceToDevice, c->stream));
    int rc = run_plan(c, c->p_cholX);
    if (rc) return rc;
    if (d_Xchol != c->d_X) HIPCHECK(hipMemcpyAsync(d_Xchol, c->d_X, sizeof(double) * c->xylen, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int clrs_cholesky_blocks(clrs_ctx *c, const double *X, double *Xchol) {
    if (!c || !X || !Xchol) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(c->device));
    int rc;
    pin_reset(c);
    if ((rc = h2d_staged(c, c->d_X, X, sizeof(double) * c->xylen))) return rc;
    if ((rc = clrs_cholesky_blocks_dev(c, c->d_X, c->d_X))) return rc;
    if ((rc = d2h_staged(c, Xchol, c->d_X, sizeof(double) * c->xylen))) return rc;
    int st = 0;
    if ((rc = read_info(c, &st, 1))) return rc;          // synchronises the stream: the factors have landed as well
    if ((rc = d2h_finish(c))) return rc;
    return st;
}

extern "C" double *clrs_q_buffer_dev(clrs_ctx *c) { return c ? c->d_Q : nullptr; }
extern "C" double *clrs_u_buffer_dev(clrs_ctx *c) { return c ? c->d_u : nullptr; }
extern "C" double *clrs_S_buffer_dev(clrs_ctx *c) { return c ? c->d_S : nullptr; }
extern "C" double *clrs_AY_buffer_dev(clrs_ctx *c) { return c ? c->d_AY : nullptr; }
extern "C" void *clrs_stream(clrs_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int clrs_set_timing(clrs_ctx *c, int enabled) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    c->timing = enabled != 0;
    return 0;
}
extern "C" int clrs_get_timings(clrs_ctx *c, double t[6]) {
    if (!c || !t) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipStreamSynchronize(c->stream));
    collect_times(c);
    for (int i = 0; i < 6; i++) t[i] = c->times[i];
    return 0;
}
extern "C" int clrs_get_counters(const clrs_ctx *c, double *ab, double *af, double *ff, double *sf) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    if (ab) *ab = c->cnt_bytes;
    if (af) *af = c->cnt_flops;
    if (ff) *ff = c->cnt_factor_flops;
    if (sf) *sf = c->cnt_solve_flops;
    return 0;
}
extern "C" int clrs_set_graph_mode(clrs_ctx *c, int enabled) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    c->graph_mode = enabled != 0;
    return 0;
}
extern "C" int clrs_plan_info(const clrs_ctx *c, int32_t *na, int32_t *nf, int32_t *ns) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    if (na) *na = c->p_assemble.launches();
    if (nf) *nf = c->p_cholS.launches() + c->p_linvB.launches() + c->p_Q.launches() + c->p_cholQ.launches();
    if (ns) *ns = c->p_fwd.launches() + c->p_bwd.launches();
    return 0;
}

// ---- external stream / per-kernel timing -----------------------------------------------------------
extern "C" int clrs_set_stream(clrs_ctx *c, void *stream) {
    if (!c) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    Plan *plans[] = {&c->p_assemble, &c->p_cholS, &c->p_linvB, &c->p_Q, &c->p_cholQ, &c->p_fwd, &c->p_bwd, &c->p_cholX, &c->p_cholQ_slabs, &c->p_solve_all, &c->p_fwd_small, &c->p_bwd_small, &c->p_factor_small};
    for (Plan *p : plans)
        if (p->graph) { hipGraphExecDestroy(p->graph); p->graph = nullptr; }   // graphs are re-captured on the new stream
    if (c->own_stream) HIPCHECK(hipStreamDestroy(c->stream));
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
    return 0;
}

static int kt_collect(clrs_ctx *c) {
    if (c->kt_kinds.empty()) return 0;
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->kt_kinds.size(); i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->kt_ev[2 * i], c->kt_ev[2 * i + 1]) == hipSuccess) {
            c->kt_total[c->kt_kinds[i]] += ms * 1e-3;
            c->kt_count[c->kt_kinds[i]]++;
        }
    }
    c->kt_kinds.clear();
    return 0;
}

extern "C" int clrs_set_kernel_timing(clrs_ctx *c, int kind) {
    if (!c || kind < -2 || kind >= STEP_NKINDS) return fail(CLRS_ERR_INVALID, "bad kernel kind");
    int rc = kt_collect(c);
    if (rc) return rc;
    c->kt_kind = kind;
    for (int k = 0; k < STEP_NKINDS; k++) { c->kt_total[k] = 0; c->kt_count[k] = 0; }
    return 0;
}

extern "C" int clrs_get_kernel_times(clrs_ctx *c, int max_kinds, double *seconds, int64_t *launches) {
    if (!c || !seconds || !launches) return fail(CLRS_ERR_INVALID, "null argument");
    int rc = kt_collect(c);
    if (rc) return rc;
    for (int k = 0; k < max_kinds && k < STEP_NKINDS; k++) { seconds[k] = c->kt_total[k]; launches[k] = c->kt_count[k]; }
    return STEP_NKINDS;
}

extern "C" int clrs_config_set(const char *key, int value) {
    if (!key) return fail(CLRS_ERR_INVALID, "null argument");
    if (!std::strcmp(key, "fused_assemble")) { g_cfg_fused_assemble = value; return 0; }
    if (!std::strcmp(key, "split_blocks")) { g_cfg_split_blocks = value; return 0; }
    if (!std::strcmp(key, "fused_factor")) { g_cfg_fused_factor = value; return 0; }
    if (!std::strcmp(key, "wave_assemble")) { g_cfg_wave_assemble = value; return 0; }
    if (!std::strcmp(key, "wave2_assemble")) { g_cfg_wave2_assemble = value; return 0; }
    if (!std::strcmp(key, "wave3_assemble")) { g_cfg_wave3_assemble = value; return 0; }
    if (!std::strcmp(key, "wave4_assemble")) { g_cfg_wave4_assemble = value; return 0; }
    if (!std::strcmp(key, "wave5_assemble")) { g_cfg_wave5_assemble = value; return 0; }
    if (!std::strcmp(key, "dense_block")) { g_cfg_dense_block = value; return 0; }
    if (!std::strcmp(key, "dense_wave")) { g_cfg_dense_wave = value; return 0; }
    if (!std::strcmp(key, "factor_aug")) { g_cfg_factor_aug = value; return 0; }
    if (!std::strcmp(key, "pairing_tri")) { g_cfg_pairing_tri = value; return 0; }
    if (!std::strcmp(key, "potrf_levels")) { g_cfg_potrf_levels = value; return 0; }
    if (!std::strcmp(key, "trsm_blockinv")) { g_cfg_trsm_blockinv = value; return 0; }
    if (!std::strcmp(key, "solve_small2")) { g_cfg_solve_small2 = value; return 0; }
    if (!std::strcmp(key, "factor_small")) { g_cfg_factor_small = value; return 0; }
    if (!std::strcmp(key, "pin_limit")) { PIN_LIMIT = (size_t)std::max(value, 0); return 0; }
    if (!std::strcmp(key, "ipm_wmfma")) { g_cfg_ipm_wmfma = value; return 0; }
    if (!std::strcmp(key, "solve_small_max")) { g_cfg_solve_small_max = value; return 0; }
    if (!std::strcmp(key, "mw_exact_products")) { g_cfg_mw_exact_products = value; return 0; }
    if (!std::strcmp(key, "mw_refine")) { g_cfg_mw_refine = value; return 0; }
    if (!std::strcmp(key, "mw_factor_limbs")) { g_cfg_mw_factor_limbs = value; return 0; }
    if (!std::strcmp(key, "mw_affine_corrector")) { g_cfg_mw_affine_corrector = value; return 0; }
    if (!std::strcmp(key, "mw_pipeline")) { g_cfg_mw_pipeline = value; return 0; }
    if (!std::strcmp(key, "mw_pipeline64")) { g_cfg_mw_pipeline64 = value; return 0; }
    if (!std::strcmp(key, "mw_zt_small_maxn")) { g_cfg_mw_zt_small_maxn = value; return 0; }
    if (!std::strcmp(key, "mw_pipeline_x")) { g_cfg_mw_pipeline_x = value; return 0; }
    if (!std::strcmp(key, "mw_pipeline_x_min")) { g_cfg_mw_pipeline_x_min = value; return 0; }
    if (!std::strcmp(key, "mw_sharded_factor_limbs")) { g_cfg_mw_sharded_factor_limbs = value; return 0; }
    if (!std::strcmp(key, "mw_stream_words")) { g_cfg_mw_stream_words = value; return 0; }
    if (!std::strcmp(key, "mw_refine_predictor")) { g_cfg_mw_refine_predictor = value; return 0; }
    return fail(CLRS_ERR_INVALID, std::string("unknown configuration key ") + key);
}

#ifdef CLRS_W3_STAMPS
extern "C" int clrs_debug_cf_stamps(clrs_ctx *c, uint64_t out[16]) {
    if (!c || !out) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cf_stamps), 16 * sizeof(uint64_t)));
    return 0;
}
extern "C" int clrs_debug_ss2_stamps(clrs_ctx *c, uint64_t out[16]) {
    if (!c || !out) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ss2_stamps), 16 * sizeof(uint64_t)));
    return 0;
}
extern "C" int clrs_debug_w3_stamps(clrs_ctx *c, uint64_t out[16]) {
    if (!c || !out) return fail(CLRS_ERR_INVALID, "null argument");
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w3_stamps), 16 * sizeof(uint64_t)));
    return 0;
}
#endif

extern "C" int clrs_debug_stamps(clrs_ctx *c, uint64_t out[64]) {
    if (!c || !out) return fail(CLRS_ERR_INVALID, "null argument");
    if (!c->ftables.stamps) { std::memset(out, 0, 64 * sizeof(uint64_t)); return 0; }
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipMemcpy(out, c->ftables.stamps, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int clrs_fused_clusters(const clrs_ctx *c) { return c ? c->n_fused_clusters : 0; }
extern "C" int clrs_wave_clusters(const clrs_ctx *c) { return c ? c->n_wave_clusters + c->n_wave2_clusters : 0; }
extern "C" int clrs_wave2_clusters(const clrs_ctx *c) { return c ? c->n_wave2_clusters : 0; }
extern "C" int clrs_wave4_clusters(const clrs_ctx *c) { return c ? c->n_wave4_clusters : 0; }
extern "C" int clrs_wave5_clusters(const clrs_ctx *c) { return c ? c->n_wave5_clusters : 0; }

extern "C" const char *clrs_kernel_name(int kind) { return (kind >= 0 && kind < STEP_NKINDS) ? STEP_NAMES[kind] : ""; }

extern "C" const char *clrs_strerror(int code) {
    switch (code) {
        case CLRS_OK: return "ok";
        case CLRS_ERR_INVALID: return "invalid argument or malformed SDP description";
        case CLRS_ERR_HIP: return "HIP runtime error";
        case CLRS_ERR_NO_DEVICE: return "no usable HIP device";
        case CLRS_ERR_STATE: return "call order violated";
        default: return code > 0 ? "factorisation failure (non-positive pivot)" : "unknown error";
    }
}
extern "C" const char *clrs_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *clrs_version(void) { return "clrs-hip 0.1.0 (gfx950)"; }

#include "clrs_ipm_host.inc"

// ------------------------------------------------------------------------------------------------
// test hooks
// ------------------------------------------------------------------------------------------------
static clrs_ctx *mini_ctx(int device) {
    clrs_ctx *c = new clrs_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    (void)hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128));
    (void)hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128));
    (void)hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128));
    (void)hipFuncSetAttribute((const void *)k_gemm_f64_t<128, 128, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes(128, 128));
    std::vector<int> hi(1, INFO_NONE);
    int *di;
    if (upload(c, hi, &di)) { clrs_ctx_destroy(c); return nullptr; }
    c->d_info = di;
    return c;
}

extern "C" int clrs_test_gemm(int device, int ta, int tb, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                              int ldb, double beta, double *C, int ldc) {
    clrs_ctx *c = mini_ctx(device);
    if (!c) return fail(CLRS_ERR_NO_DEVICE, "no device");
    int rc = 0;
    i64 na = (i64)lda * (ta ? M : K), nb = (i64)ldb * (tb ? K : N), nc = (i64)ldc * N;
    std::vector<double> hA(A, A + na), hB(B, B + nb), hC(C, C + nc);
    double *dA, *dB, *dC;
    if ((rc = upload(c, hA, &dA)) || (rc = upload(c, hB, &dB)) || (rc = upload(c, hC, &dC))) { clrs_ctx_destroy(c); return rc; }
    Plan pl;
    std::vector<GemmDesc> g;
    g.push_back(mk_gemm(ta, tb, M, N, K, alpha, dA, lda, dB, ldb, beta, dC, ldc));
    if (!(rc = add_gemm_stage(c, pl, g)) && !(rc = run_steps(c, pl))) {
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(C, dC, sizeof(double) * nc, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(CLRS_ERR_HIP, "copy back failed");
    }
    clrs_ctx_destroy(c);
    return rc;
}

extern "C" int clrs_test_potrf(int device, int n, double *A, int lda) {
    clrs_ctx *c = mini_ctx(device);
    if (!c) return fail(CLRS_ERR_NO_DEVICE, "no device");
    int rc = 0, st = 0;
    std::vector<double> hA(A, A + (i64)lda * n);
    double *dA;
    if ((rc = upload(c, hA, &dA))) { clrs_ctx_destroy(c); return rc; }
    Plan pl;
    std::vector<PotrfJob> pj;
    pj.push_back(PotrfJob{dA, lda, n, 1});
    if (!(rc = plan_potrf(c, pl, pj)) && !(rc = run_steps(c, pl)) && !(rc = read_info(c, &st))) {
        if (hipMemcpy(A, dA, sizeof(double) * (i64)lda * n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CLRS_ERR_HIP, "copy back failed");
    }
    clrs_ctx_destroy(c);
    return rc ? rc : st;
}

// ---- measurement hook: what a pure streaming kernel reaches at a given footprint and read : write mix -------------------------
// (bench.py reports the assembly kernel's HBM traffic per second beside this rate: the 8 TB/s of the data sheet is not reachable by
// any kernel -- beyond the 256 MiB Infinity Cache a plain copy runs at 5.0-5.8 TB/s on this part)
typedef double v2d_stream __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_stream_probe(const v2d_stream *__restrict__ a, const v2d_stream *__restrict__ b, v2d_stream *__restrict__ w,
                                                      long long nr, long long nw) {
    const long long stride = (long long)gridDim.x * 256, n = nr > nw ? nr : nw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
        v2d_stream x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long k = i + u * stride;
            x[u] = y[u] = (v2d_stream){0.0, 0.0};
            if (k < nr) { x[u] = a[k]; y[u] = b[k]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long k = i + u * stride;
            if (k < nw) __builtin_nontemporal_store(x[u] + y[u], w + k);
        }
    }
}

extern "C" int clrs_test_stream(int device, long long read_bytes, long long write_bytes, int reps, double *avg_us) {
    if (read_bytes < 32 || write_bytes < 16 || reps < 1 || !avg_us) return fail(CLRS_ERR_INVALID, "clrs_test_stream: bad arguments");
    if (hipSetDevice(device) != hipSuccess) return fail(CLRS_ERR_NO_DEVICE, "no device");
    const long long nr = read_bytes / 32, nw = write_bytes / 16;        // 16-byte elements: two read streams of nr, one write stream of nw
    v2d_stream *a = nullptr, *b = nullptr, *w = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    int rc = 0;
    auto ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && !rc) rc = fail(CLRS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); return e == hipSuccess; };
    if (ok(hipMalloc(&a, nr * 16), "hipMalloc") && ok(hipMalloc(&b, nr * 16), "hipMalloc") && ok(hipMalloc(&w, nw * 16), "hipMalloc") &&
        ok(hipMemset(a, 0, nr * 16), "hipMemset") && ok(hipMemset(b, 0, nr * 16), "hipMemset") && ok(hipMemset(w, 0, nw * 16), "hipMemset") &&
        ok(hipStreamSynchronize(nullptr), "hipStreamSynchronize") &&      // the fills (null stream) are done before the probe's own stream starts
        ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate") && ok(hipEventCreate(&e0), "hipEventCreate") && ok(hipEventCreate(&e1), "hipEventCreate")) {
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
        const int grid = 2 * cus;
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_stream_probe, dim3(grid), dim3(256), 0, st, a, b, w, nr, nw);
        ok(hipEventRecord(e0, st), "hipEventRecord");
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_stream_probe, dim3(grid), dim3(256), 0, st, a, b, w, nr, nw);
        ok(hipEventRecord(e1, st), "hipEventRecord");
        ok(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0.f;
        if (ok(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) *avg_us = 1e3 * (double)ms / reps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (w) (void)hipFree(w);
    return rc;
}

extern "C" int clrs_test_trsm(int device, int trans, int n, int nrhs, const double *L, int ldl, double *B, int ldb) {
    clrs_ctx *c = mini_ctx(device);
    if (!c) return fail(CLRS_ERR_NO_DEVICE, "no device");
    int rc = 0;
    std::vector<double> hL(L, L + (i64)ldl * n), hB(B, B + (i64)ldb * nrhs);
    double *dL, *dB;
    if ((rc = upload(c, hL, &dL)) || (rc = upload(c, hB, &dB))) { clrs_ctx_destroy(c); return rc; }
    Plan pl;
    std::vector<TrsmJob> tj;
    tj.push_back(TrsmJob{dL, ldl, n, dB, ldb, nrhs});
    if (!(rc = plan_trsm(c, pl, tj, trans)) && !(rc = run_steps(c, pl))) {
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(B, dB, sizeof(double) * (i64)ldb * nrhs, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(CLRS_ERR_HIP, "copy back failed");
    }
    clrs_ctx_destroy(c);
    return rc;
}
