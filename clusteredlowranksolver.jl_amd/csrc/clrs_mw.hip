// clrs_mw.hip -- the hot path at the reference's working precision: multi-word fp64 (K limbs), host side + C ABI
// (include/clrs_hip.h, clrs_mw_*).  Compiled with -ffp-contract=off (clrs_mw_arith.h) and linked into libclrs_hip.so.
//
// A context of its own (clrs_mw_ctx), created from the same clrs_sdp_desc as the fp64 context: the de-duplication of
// the sampled vectors (precompute_matrices_bilinear_pairings, src/solver.jl:985-1059) is redone here into ONE table of
// expanded unique vectors per PSD block -- a vector of sub-block r is stored with its delta entries at rows
// r*delta.. and zeros elsewhere, duplicates removed by exact equality as the reference does (src/tools.jl:128-145) -- so
// that both pairing matrices of a block are plain symmetric products V^T X^-1 V and V^T Y V, and the reference's
// pointers_left / pointers_right dictionaries become two integers per term.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/clrs_hip.h"
#include "clrs_mw_kernels.hip.h"
#include "clrs_mw_pipe.hip.h"
#include "clrs_mw_exact.hip.h"
#include "clrs_mw_ipm.hip.h"
#include "clrs_mw_inst.h"
#ifdef MW_SPLIT_UNITS        // the kernels of these limb counts are compiled in units of their own (clrs_mw_inst.hip)
MW_KERNELS_ALL(extern template, 4)
MW_KERNELS_ALL(extern template, 5)
MW_KERNELS_ALL(extern template, 6)
MW_KERNELS_ALL(extern template, 8)
MW_KERNELS_ALL(extern template, 10)
#endif

typedef long long i64;

extern int g_cfg_mw_stream_words;                        // clrs_hip.hip, clrs_config_set("mw_stream_words", 0 / 1); env CLRS_MW_STREAM_WORDS
extern int g_cfg_mw_pipeline64;                          // clrs_hip.hip, clrs_config_set("mw_pipeline64", 0 / 1): the 64-row form for clusters of 33 .. 64 rows
extern int g_cfg_mw_zt_small_maxn;                       // clrs_hip.hip, clrs_config_set("mw_zt_small_maxn", rows): k_mw_zt with two columns per workgroup and eight lanes per entry up to this block side
extern int g_cfg_mw_pipeline_x, g_cfg_mw_pipeline_x_min; // clrs_hip.hip, clrs_config_set("mw_pipeline_x", 0 / 1), ("mw_pipeline_x_min", rows): the Cholesky of the X blocks through the pipelines
extern int g_cfg_mw_sharded_factor_limbs;                // clrs_hip.hip, clrs_config_set("mw_sharded_factor_limbs", 0 / 1)
extern int g_cfg_mw_pipeline;                            // clrs_hip.hip, clrs_config_set("mw_pipeline", 0 / 1): read at context creation
extern int g_cfg_mw_refine_predictor;                    // clrs_hip.hip, clrs_config_set("mw_refine_predictor", 0 / 1)
extern int g_cfg_mw_refine;                              // clrs_hip.hip, clrs_config_set("mw_refine", 0 / 1): read at context creation
extern int g_cfg_mw_exact_products;                      // clrs_hip.hip, clrs_config_set("mw_exact_products", 0 / 1 / 2): read at context creation
extern int g_cfg_mw_affine_corrector;                    // clrs_hip.hip, clrs_config_set("mw_affine_corrector", 0 / 1): read by clrs_mw_ipm_create
extern int g_cfg_mw_factor_limbs;                        // clrs_hip.hip, clrs_config_set("mw_factor_limbs", 0 / limbs): read at context creation
extern "C" void clrs_set_last_error(const char *msg);   // clrs_hip.hip: the library keeps one thread-local message

static int mw_fail(int code, const std::string &msg) {
    clrs_set_last_error(msg.c_str());
    return code;
}
#define MWCHECK(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return mw_fail(CLRS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// RCCL is bound at run time (dlopen): a process that already holds a copy (PyTorch ships one) keeps using that copy, and the
// library loads on machines without RCCL.
typedef struct { char internal[128]; } mw_nccl_id;
struct MwNccl {
    void *lib = nullptr;
    int (*GetUniqueId)(mw_nccl_id *) = nullptr;
    int (*CommInitRank)(void **, int, mw_nccl_id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static MwNccl g_nccl;
static int mw_nccl_load() {
    if (g_nccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // a copy the process already has
    for (const char *n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!h) return mw_fail(CLRS_ERR_INVALID, std::string("RCCL is not available: ") + dlerror());
    g_nccl.GetUniqueId = (int (*)(mw_nccl_id *))dlsym(h, "ncclGetUniqueId");
    g_nccl.CommInitRank = (int (*)(void **, int, mw_nccl_id, int))dlsym(h, "ncclCommInitRank");
    g_nccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_nccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    g_nccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.CommDestroy || !g_nccl.AllGather) return mw_fail(CLRS_ERR_INVALID, "RCCL symbols missing");
    g_nccl.lib = h;
    return 0;
}
#define NCCLCHECK(expr)                                                                                                          \
    do {                                                                                                                         \
        int r_ = (expr);                                                                                                         \
        if (r_ != 0) return mw_fail(CLRS_ERR_HIP, std::string(#expr) + ": " + (g_nccl.GetErrorString ? g_nccl.GetErrorString(r_) : "RCCL error")); \
    } while (0)
static const int MW_NCCL_FLOAT64 = 8;      // ncclFloat64 / ncclDouble

// In-process stand-in for a communicator: `world` contexts of ONE process that share one device, driven by one host thread each
// (the two-shards-on-one-GPU tests; a box has one GPU, and RCCL refuses two ranks on one device).  An all-gather is: every rank
// records "my slot is written" on its stream, the host threads meet, every rank copies the peers' slots device to device behind the
// peers' events, records "I have read", the host threads meet again and every rank orders its stream behind the peers' reads (a
// slot is rewritten every iteration).  Two channels, like the two RCCL communicators of a context: one per stream that exchanges.
struct clrs_mw_local_group {
    int world = 0;
    int attached = 0;                   // contexts that point to this group (clrs_mw_comm_init_local / clrs_mw_comm_destroy, clrs_mw_destroy)
    std::mutex m;
    std::condition_variable cv;
    struct Chan {
        int arrived = 0;
        long gen = 0;
        std::vector<double *> base;
        std::vector<hipEvent_t> ready, copied;
    } ch[2];
    void barrier(int k) {
        std::unique_lock<std::mutex> lk(m);
        Chan &c = ch[k];
        const long g = c.gen;
        if (++c.arrived == world) { c.arrived = 0; c.gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return c.gen != g; });
    }
};

static const size_t MW_LDS_MAX = 160 * 1024 - 2048;     // bytes of LDS one workgroup may claim on gfx950 (margin for the runtime)

struct clrs_mw_ctx {
    int device = 0, K = 4, DK = 1;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    MwDev d = {};
    std::vector<MwBlk> blk;
    std::vector<MwClu> clu;
    std::vector<void *> allocs;
    int maxU = 0, maxP = 0, maxn = 0, maxn_dense = 0;
    bool all_inv = false;               // every low-rank block has its inverse factor (k.inv != 0)
    bool xinv_valid = false;            // Xi holds the inverses of the current Cholesky factors (they come from k_mw_potrf_x, not from the caller)
    bool lds_x = false, lds_q = false, lds_zt_L = false, dense_two = false;
    int nw_factor = 1;                  // workgroups per cluster in k_mw_factor (they share out the columns of the inverse factor)
    int maxcnt = 0;
    double *vz = nullptr;               // 2 N numbers of scratch of the solve stage over many workgroups (k_mw_solve_wide)
    bool wide_solve = false;            // some cluster or Q has more than 64 rows: the products of the solve stage are launches of their own
    const double *ride_fwd = nullptr;   // set by the iteration around clrs_mw_schur_factor_finish_dev: rhs_x of the solve that follows
    const int *ride_wait = nullptr;     // ... and, with it, the word its workgroups wait for inside the launch (>= ride_wait_value) instead of an event in front of it
    int ride_wait_value = 0;
    bool fwd_rode = false;              // ... and its first product pair (k_mw_solve_fwd's work) was done on the launch of k_mw_potrf_q
    int n_one_term = 0, n_many_term = 0;   // clusters whose S_j goes through k_mw_saccum_one / through the general k_mw_saccum
    int sa_lanes = MW_SA_W;             // lanes per entry of k_mw_saccum: 1, 2 or 4 by the largest block count of a cluster
    int maxTb = 0;                      // most low-rank terms in one PSD block
    size_t sm_x = 0, sm_zt = 0, sm_dense = 0, sm_factor = 0, sm_q = 0, sm_fwd = 0, sm_mid = 0, sm_bwd = 0;
    int *h_info = nullptr;               // pinned
    double *d_Xin = nullptr, *d_Xc = nullptr, *d_Y = nullptr, *d_rx = nullptr, *d_ry = nullptr, *d_dx = nullptr, *d_dy = nullptr;   // staging of the host-pointer entry points
    bool assembled = false, factored = false;
    bool timing = false;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double times[6] = {0, 0, 0, 0, 0, 0};
    double cnt_mul = 0;                  // multi-word multiply-adds of one assembly (algorithmic)
    double cnt_factor = 0, cnt_solve = 0;
    struct MwIpm *ipm = nullptr;
    size_t sm_bp_diag = 0, sm_bp_panel = 0, sm_bp_inv = 0;   // LDS of the blocked factorisation (k_mw_bp_*)
    std::vector<MwBp> bp_S, bp_Q;       // the matrices of the blocked path: the clusters beyond LDS, and Q (when it is)
    const MwBp *d_bp = nullptr;         // both lists on the device, bp_S first
    bool any_lds_cluster = false;
    void *comm = nullptr;                // ncclComm_t when the library does the exchanges itself (clrs_mw_comm_init): the context's stream
    void *comm_side = nullptr;           // a second communicator for the exchanges of the iteration's side stream (clrs_mw_comm_init_side)
    clrs_mw_local_group *lgroup = nullptr;   // or: the in-process group (clrs_mw_comm_init_local)
    bool local_factored = false, fwd_done = false;
    const double *split_rhs_x = nullptr;   // right-hand side of the last clrs_mw_schur_solve_fwd_dev (the refinement's residual needs it)
    bool refine_ready = false;             // clrs_mw_schur_solve_bwd_dev left the first half of a refinement step behind
    MwdDev mwd = {};                     // static digits of the dense matrices of the blocks k_mwx_dense takes
    int mwd_blocks = 0, mwd_tasks = 0;
    size_t sm_mwd = 0;
    MwxDev mwx = {};                     // digits of V (static), Z, T of the blocks whose pairing matrices go through k_mwx_slice / k_mwx_gram
    int mwx_blocks = 0, mwx_maxU16 = 0;
    MwsDev mws = {};                     // static V slices of the blocks the exact-product kernel takes
    int mws_blocks = 0;                  // how many
    int mws_turns = 1;                   // 2: some eligible block has more than four T / Z tiles
    size_t sm_mws = 0;
    bool pipe_bp = false;                // the diagonal blocks of the blocked factorisation as pipelines (k_mw_bp_diag_pipe)
    bool pipe_S64 = false;               // ... the clusters' S_j of 33 .. 64 rows through the 64-row form of the pipeline (k_mw_factor_pipe64; limbs <= 6)
    bool pipe_X = false;                 // ... the Cholesky of the X blocks (and of the Y blocks beside it) through the pipelines (k_mw_potrf_x_pipe): every block with its explicit inverse factor, <= 64 rows
    unsigned long long *pipe_pcx = nullptr;    // its hand-off granules: [2 NB][MWP_PC_WORDS_N(K, 64)]
    double *xpipe_L = nullptr, *xpipe_rd = nullptr;   // where the factors / reciprocal diagonals of the Y blocks go (only their inverses are kept)
    int *xpipe_info = nullptr;
    bool pipe_S = false, pipe_Q = false;  // the factorisations of the clusters / of Q as pipelines of workgroups (clrs_mw_pipe.hip.h): every matrix <= 32 rows, few clusters
    unsigned long long *pipe_pc64 = nullptr;   // hand-off granules of k_mw_factor_pipe64: [J][MWP_PC_WORDS_N(K, 64)]
    int pipe_pcQ = 0;                    // index of Q's hand-off region in pipe_pc
    unsigned pipe_epoch = 0;             // launch counter: the tag of the hand-off granules
    bool stream_words = true;            // the interior-point iteration synchronises its two streams through words (clrs_mw_ipm_host.inc) where it can; false: events only
    bool refine_skip_next = false;       // the interior-point iteration's PREDICTOR solve: one pass (set by clrs_mw_ipm_host.inc for the next clrs_mw_schur_solve_dev only)
    int refine_predictor = 0;            // clrs_mw_options.refine_predictor: 1 = the predictor's solve is refined like every other
    int refine = 1;                      // iterative refinement of the solve stage (clrs_mw_options / clrs_config_set("mw_refine")): 0 off, 1 one step with the correction in all K limbs, 2 ... in mw_kc(K) limbs
    bool ipm_arms_info = false;          // inside the device-resident iteration the status words are re-armed by a kernel, not by a memset per call
    // Mixed-precision refinement (MwDev::kf, mw_kf_of): kf_low = the reduced limb count this context may run its factor stage and the products of its solve
    // stage in (K: it may not -- limb counts without a reduced form, refinement off or in fewer limbs, blocked / row-parallel paths); kf_entry = what the
    // stand-alone entry points use (clrs_mw_options.factor_limbs: 0 = K there and kf_low, adaptively, inside clrs_mw_ipm_*; explicit: that count everywhere)
    int kf_low = 0, kf_entry = 0;
};

template <class T>
static int mw_upload(clrs_mw_ctx *c, const std::vector<T> &h, const T **d) {
    T *p = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    MWCHECK(hipMalloc((void **)&p, bytes));
    c->allocs.push_back(p);
    if (!h.empty()) MWCHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = p;
    return 0;
}
static int mw_dmalloc(clrs_mw_ctx *c, double **d, i64 n) {
    size_t bytes = (size_t)std::max<i64>(n, 1) * sizeof(double);
    MWCHECK(hipMalloc((void **)d, bytes));
    c->allocs.push_back(*d);
    MWCHECK(hipMemset(*d, 0, bytes));
    return 0;
}

// dispatch over the limb count K of the computed numbers and DK of the problem data (1: fp64, 2: double-double)
#define MW_CASE(Kc, Dc, ...) if (c_K_ == Kc && c_D_ == Dc) { constexpr int KK = Kc; constexpr int DD = Dc; __VA_ARGS__; } else
#define MW_DISPATCH(ctx, ...)                                                                                       \
    do {                                                                                                            \
        const int c_K_ = (ctx)->K, c_D_ = (ctx)->DK;                                                                \
        MW_CASE(2, 1, __VA_ARGS__) MW_CASE(2, 2, __VA_ARGS__) MW_CASE(3, 1, __VA_ARGS__) MW_CASE(3, 2, __VA_ARGS__)  \
        MW_CASE(4, 1, __VA_ARGS__) MW_CASE(4, 2, __VA_ARGS__) MW_CASE(5, 1, __VA_ARGS__) MW_CASE(5, 2, __VA_ARGS__)  \
        MW_CASE(6, 1, __VA_ARGS__) MW_CASE(6, 2, __VA_ARGS__) MW_CASE(8, 1, __VA_ARGS__) MW_CASE(8, 2, __VA_ARGS__)  \
        MW_CASE(10, 1, __VA_ARGS__) MW_CASE(10, 2, __VA_ARGS__)                                                      \
        MW_CASE(3, 3, __VA_ARGS__) MW_CASE(4, 4, __VA_ARGS__) MW_CASE(5, 5, __VA_ARGS__) MW_CASE(6, 6, __VA_ARGS__)  \
        MW_CASE(8, 8, __VA_ARGS__) MW_CASE(10, 10, __VA_ARGS__)                                                      \
        return mw_fail(CLRS_ERR_INVALID, "limbs must be 2..6, 8 or 10 and data limbs 1, 2 or the limbs");           \
    } while (0)

template <class F>
static int mw_set_lds(F kernel, size_t bytes) {
    if (bytes > 48 * 1024) MWCHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

extern "C" void clrs_mw_destroy(clrs_mw_ctx *c);
static void mw_ipm_free(clrs_mw_ctx *c);
static int mw_launch_xrd(clrs_mw_ctx *c, const double *d_Xc);

extern "C" int clrs_mw_create_opts(const clrs_sdp_desc *d, int data_limbs, int device, int limbs, const clrs_mw_options *opts, clrs_mw_ctx **out);
extern "C" int clrs_mw_create(const clrs_sdp_desc *d, int device, int limbs, clrs_mw_ctx **out) {
    return clrs_mw_create_opts(d, 1, device, limbs, nullptr, out);
}
extern "C" int clrs_mw_create_ex(const clrs_sdp_desc *d, int data_limbs, int device, int limbs, clrs_mw_ctx **out) {
    return clrs_mw_create_opts(d, data_limbs, device, limbs, nullptr, out);
}

// opts: per-context choices (a field < 0, or opts == NULL: the process-wide default of clrs_config_set) -- contexts created side by side from
// several threads do not share a knob this way
extern "C" int clrs_mw_create_opts(const clrs_sdp_desc *d, int data_limbs, int device, int limbs, const clrs_mw_options *opts, clrs_mw_ctx **out) {
    if (!d || !out) return mw_fail(CLRS_ERR_INVALID, "null argument");
    int cfg_exact = opts && opts->exact_products >= 0 ? opts->exact_products : g_cfg_mw_exact_products;
    const int cfg_refine = opts && opts->refine >= 0 ? opts->refine : g_cfg_mw_refine;
    const int cfg_pipe = opts && opts->pipeline >= 0 ? opts->pipeline : g_cfg_mw_pipeline;
    const int cfg_refine_pred = opts && opts->refine_predictor >= 0 ? opts->refine_predictor : g_cfg_mw_refine_predictor;
    const int cfg_factor_limbs = opts && opts->factor_limbs >= 0 ? opts->factor_limbs : g_cfg_mw_factor_limbs;
    // matmul_prec of the reference (src/solver.jl:125): limbs of the pairing products, rounded UP to the next count on offer (mw_km_ok); 0 = the context's limbs
    int cfg_km = opts && opts->matmul_limbs > 0 ? opts->matmul_limbs : limbs;
    if (cfg_km > limbs) cfg_km = limbs;
    while (cfg_km < limbs && !mw_km_ok(limbs, cfg_km)) cfg_km++;
    if (data_limbs > 2) cfg_exact = 0;                     // (the static digits of the exact slice products are cut from two data limbs)
    if (cfg_km < limbs) cfg_exact = 0;                     // (the exact slice products have one slice count per limb count: the expansion kernels take the reduced products)
    if (cfg_exact > 2 || cfg_refine > 2) return mw_fail(CLRS_ERR_INVALID, "clrs_mw_options: exact_products and refine are 0, 1 or 2 (or < 0 for the default)");
    if (limbs < 2 || limbs > 10 || limbs == 7 || limbs == 9) return mw_fail(CLRS_ERR_INVALID, "limbs must be 2..6, 8 or 10");
    if (data_limbs < 1 || data_limbs > limbs || (data_limbs > 2 && data_limbs != limbs))
        return mw_fail(CLRS_ERR_INVALID, "data limbs must be 1, 2 or the context's limbs (the problem data at the working precision: src/interface.jl:1078-1112)");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) return mw_fail(CLRS_ERR_NO_DEVICE, "no usable HIP device");
    MWCHECK(hipSetDevice(device));
    clrs_mw_ctx *c = new clrs_mw_ctx();
    c->device = device;
    c->K = limbs;
    c->DK = data_limbs;
    const int K = limbs, DK = data_limbs;
    const int J = d->n_clusters, N = d->n_free, NB = d->n_blocks;
    if (J <= 0 || N < 0 || NB < 0) { delete c; return mw_fail(CLRS_ERR_INVALID, "bad sizes"); }
    int rc = 0;
#define MW_BAIL(code, msg) do { clrs_mw_destroy(c); return mw_fail(code, msg); } while (0)
#define MW_TRY(call) do { if ((rc = (call))) { clrs_mw_destroy(c); return rc; } } while (0)
    {   // The context's stream at the HIGHEST priority, the side stream of the interior-point iteration (clrs_mw_ipm_host.inc) at the default one: the runtime
        // maps streams onto a few hardware queues PER PRIORITY (GPU_MAX_HW_QUEUES, 4 by default), and two streams of one iteration that land on one queue run
        // its packets in submission order -- measured: the second context of a process, whose side stream shared a queue, 0.58 ms per iteration against
        // 0.45 (profiles/r05).  Different priorities never share; and the main stream carries the longest chain of the iteration.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipStreamCreateWithPriority failed");
    }
    if (hipHostMalloc((void **)&c->h_info, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipHostMalloc failed");
    // ---- clusters ----
    c->clu.resize(J);
    i64 xlen = 0, Slen = 0;
    for (int j = 0; j < J; j++) {
        MwClu &q = c->clu[j];
        q.P = d->cluster_P[j];
        if (q.P <= 0) MW_BAIL(CLRS_ERR_INVALID, "cluster without constraints");
        q.coff = xlen; q.Soff = Slen; q.b0 = NB; q.b1 = 0;
        xlen += q.P; Slen += (i64)q.P * q.P;
        c->maxP = std::max(c->maxP, q.P);
    }
    // ---- blocks, unique expanded vectors, term tables ----
    c->blk.resize(NB);
    const i64 T = NB ? d->term_ptr[NB] : 0, D = NB ? d->dense_ptr[NB] : 0;
    // planes of the description's data arrays (data_limbs planes each)
    const i64 vec_plane = T ? d->term_vec_ptr[T] : 0, dA_plane = D ? d->dense_A_ptr[D] : 0;
    std::vector<double> hV;          // limb 0 while the tables are built; the other limbs follow below
    std::vector<int> hvrow, st_a(std::max<i64>(T, 1)), st_b(std::max<i64>(T, 1)), htptr, ay_a(std::max<i64>(T, 1), 0), ay_b(std::max<i64>(T, 1), 0),
        ay_blk(std::max<i64>(T, 1), -1), hdmap, lr_list, dn_list;
    std::vector<std::tuple<i64, int, int>> drow_pairs;       // (stacked row, block, entry) of every dense matrix
    std::vector<double> st_lam((size_t)std::max<i64>(T, 1) * DK), hdA;
    std::vector<std::vector<double>> hVl(DK), hdAl(DK);     // per limb
    // for the interior-point iteration around the path (clrs_mw_ipm.hip.h), sorted term order: original term, vs at sub-block r / ws at
    // sub-block s (compute_weighted_A!, src/solver.jl:1433-1459), ws at r / vs at s (trace_A, :1334-1341), and the flags s <= r, r != s
    std::vector<int> st_orig(std::max<i64>(T, 1)), st_war(std::max<i64>(T, 1)), st_wac(std::max<i64>(T, 1)), st_trl(std::max<i64>(T, 1)),
        st_trd(std::max<i64>(T, 1)), st_flag(std::max<i64>(T, 1)), st_p(std::max<i64>(T, 1));
    i64 xyoff = 0, rdoff = 0, zoff = 0, goff = 0, sdoff = 0, woff = 0;
    double cnt_mul = 0;
    for (int b = 0; b < NB; b++) {
        MwBlk &k = c->blk[b];
        std::memset(&k, 0, sizeof(k));
        k.j = d->block_cluster[b];
        if (k.j < 0 || k.j >= J || (b > 0 && k.j < c->blk[b - 1].j)) MW_BAIL(CLRS_ERR_INVALID, "block_cluster must be non-decreasing and within range");
        const int m = d->block_m[b];
        k.delta = d->block_delta[b];
        k.n = m * k.delta;
        k.kind = d->block_kind[b];
        k.m = m;
        k.P = c->clu[k.j].P;
        if (k.n <= 0 || (k.kind != 0 && m != 1)) MW_BAIL(CLRS_ERR_INVALID, "bad block shape");
        k.xyoff = xyoff; xyoff += (i64)k.n * k.n;
        k.rd_off = rdoff; rdoff += k.n;
        c->clu[k.j].b0 = std::min(c->clu[k.j].b0, b);
        c->clu[k.j].b1 = std::max(c->clu[k.j].b1, b + 1);
        c->maxn = std::max(c->maxn, k.n);
        if (k.kind != 0) c->maxn_dense = std::max(c->maxn_dense, k.n);
        const int P = k.P, n = k.n, dl = k.delta;
        if (k.kind == 0) {
            lr_list.push_back(b);
            const i64 t0 = d->term_ptr[b], t1 = d->term_ptr[b + 1];
            k.t0 = t0;
            c->maxTb = std::max(c->maxTb, (int)(t1 - t0));
            // unique expanded vectors: (sub-block, delta values), exact equality, first occurrence wins
            std::vector<std::pair<int, const double *>> uniq;      // (sub-block, pointer to limb 0 of the vector inside term_vs / term_ws)
            auto find_or_add = [&](int r, const double *v) -> int {
                for (size_t u = 0; u < uniq.size(); u++) {
                    if (uniq[u].first != r) continue;
                    bool eq = true;
                    for (int l = 0; l < DK && eq; l++)
                        for (int i = 0; i < dl && eq; i++) eq = uniq[u].second[(i64)l * vec_plane + i] == v[(i64)l * vec_plane + i];
                    if (eq) return (int)u;
                }
                uniq.push_back({r, v});
                return (int)uniq.size() - 1;
            };
            std::map<std::tuple<int, int, int, int>, i64> index;
            for (i64 t = t0; t < t1; t++) {
                if (d->term_p[t] < 0 || d->term_p[t] >= P || d->term_r[t] < 0 || d->term_r[t] >= m || d->term_s[t] < 0 || d->term_s[t] >= m)
                    MW_BAIL(CLRS_ERR_INVALID, "term index out of range");
                if (d->term_vec_ptr[t + 1] - d->term_vec_ptr[t] != dl) MW_BAIL(CLRS_ERR_INVALID, "term vectors must have delta entries");
                index[std::make_tuple(d->term_p[t], d->term_r[t], d->term_s[t], d->term_rank[t])] = t;
            }
            // R(t): vs of the term at sub-block r; Lself(t): ws of the term at sub-block r   (rightvecs[r] / leftvecs[r], src/solver.jl:1011, 1032)
            std::vector<int> Rt(t1 - t0), Ls(t1 - t0), Cs(t1 - t0), Ds(t1 - t0);
            std::vector<i64> partner(t1 - t0);
            for (i64 t = t0; t < t1; t++) {
                Rt[t - t0] = find_or_add(d->term_r[t], d->term_vs + d->term_vec_ptr[t]);
                Ls[t - t0] = find_or_add(d->term_r[t], d->term_ws + d->term_vec_ptr[t]);
                Cs[t - t0] = find_or_add(d->term_s[t], d->term_ws + d->term_vec_ptr[t]);
                Ds[t - t0] = find_or_add(d->term_s[t], d->term_vs + d->term_vec_ptr[t]);
                auto it = index.find(std::make_tuple(d->term_p[t], d->term_s[t], d->term_r[t], d->term_rank[t]));
                if (it == index.end()) MW_BAIL(CLRS_ERR_INVALID, "term without transposed partner: A[r,s][p] must equal A[s,r][p]^T");
                partner[t - t0] = it->second;
            }
            k.U = (int)uniq.size();
            c->maxU = std::max(c->maxU, k.U);
            k.v_off = (i64)hVl[0].size();
            k.vrow_off = (i64)hvrow.size();
            for (int l = 0; l < DK; l++) hVl[l].resize(hVl[l].size() + (size_t)n * k.U, 0.0);
            for (int u = 0; u < k.U; u++) {
                hvrow.push_back(uniq[u].first * dl);
                for (int l = 0; l < DK; l++)
                    for (int i = 0; i < dl; i++) hVl[l][k.v_off + (i64)u * n + uniq[u].first * dl + i] = uniq[u].second[(i64)l * vec_plane + i];
            }
            k.z_off = zoff; zoff += (i64)n * k.U;
            k.g_off = goff; goff += (i64)k.U * k.U;
            // terms sorted by constraint (stable), CSR over p
            std::vector<i64> order(t1 - t0);
            for (i64 t = t0; t < t1; t++) order[t - t0] = t;
            std::stable_sort(order.begin(), order.end(), [&](i64 a, i64 b2) { return d->term_p[a] < d->term_p[b2]; });
            k.tptr_off = (i64)htptr.size();
            htptr.resize(htptr.size() + P + 1, 0);
            int *tp = htptr.data() + k.tptr_off;
            for (i64 i = 0; i < t1 - t0; i++) tp[d->term_p[order[i]] + 1]++;
            tp[0] = (int)t0;
            for (int p = 0; p < P; p++) tp[p + 1] += tp[p];
            for (i64 i = 0; i < t1 - t0; i++) {
                const i64 t = order[i];
                st_a[t0 + i] = Ls[partner[t - t0] - t0];      // pointers_left[s][(r,p,k)] = ws of A[s,r][p]
                st_b[t0 + i] = Rt[t - t0];                    // pointers_right[r][(s,p,k)] = vs of A[r,s][p]
                for (int l = 0; l < DK; l++) st_lam[(size_t)l * std::max<i64>(T, 1) + t0 + i] = d->term_lambda[(i64)l * T + t];
                st_orig[t0 + i] = (int)t;
                st_p[t0 + i] = d->term_p[t];
                st_war[t0 + i] = Rt[t - t0]; st_wac[t0 + i] = Cs[t - t0];
                st_trl[t0 + i] = Ls[t - t0]; st_trd[t0 + i] = Ds[t - t0];
                st_flag[t0 + i] = (d->term_s[t] <= d->term_r[t] ? 1 : 0) | (d->term_s[t] != d->term_r[t] ? 2 : 0);
            }
            for (i64 t = t0; t < t1; t++) {                    // A_Y[r,s][idx] = bpY[r,s][left_r(s,p,k), right_s(r,p,k)]  (src/solver.jl:1162)
                ay_blk[t] = b;
                ay_a[t] = Ls[t - t0];
                ay_b[t] = Rt[partner[t - t0] - t0];
            }
            // algorithmic multi-word multiply-adds of the assembly of this block: T = Y V, Z = L^-1 V, GX, GY (lower triangles), S
            cnt_mul += (double)n * dl * k.U + 0.5 * (double)n * n * k.U + 0.5 * (double)k.U * k.U * (n + dl);
            for (int p = 0; p < P; p++)
                for (int q2 = p; q2 < P; q2++) cnt_mul += (double)(tp[p + 1] - tp[p]) * (tp[q2 + 1] - tp[q2]);
        } else {
            dn_list.push_back(b);
            const i64 d0 = d->dense_ptr[b], d1 = d->dense_ptr[b + 1];
            k.cnt = (int)(d1 - d0);
            k.d0 = d0;
            k.a_off = (i64)hdAl[0].size();
            k.dmap_off = (i64)hdmap.size();
            hdmap.resize(hdmap.size() + P, -1);
            for (i64 e = d0; e < d1; e++) {
                const int p = d->dense_p[e];
                if (p < 0 || p >= P) MW_BAIL(CLRS_ERR_INVALID, "dense constraint index out of range");
                if (d->dense_A_ptr[e + 1] - d->dense_A_ptr[e] != (i64)n * n) MW_BAIL(CLRS_ERR_INVALID, "dense matrix must have n*n entries");
                for (int l = 0; l < DK; l++) {          // symmetric, as the reference's constructor makes them (src/interface.jl:1010-1017)
                    const double *Ae = d->dense_A + (i64)l * dA_plane + d->dense_A_ptr[e];
                    for (int cc = 0; cc < n; cc++)
                        for (int rr = cc + 1; rr < n; rr++)
                            if (Ae[rr + (i64)cc * n] != Ae[cc + (i64)rr * n]) MW_BAIL(CLRS_ERR_INVALID, "dense constraint matrices must be symmetric");
                }
                hdmap[k.dmap_off + p] = (int)(e - d0);
                drow_pairs.push_back(std::make_tuple(c->clu[k.j].coff + p, b, (int)(e - d0)));
                for (int l = 0; l < DK; l++)
                    hdAl[l].insert(hdAl[l].end(), d->dense_A + (i64)l * dA_plane + d->dense_A_ptr[e], d->dense_A + (i64)l * dA_plane + d->dense_A_ptr[e + 1]);
            }
            k.sd_off = sdoff; sdoff += (i64)k.cnt * k.cnt;
            k.w_off = woff; woff += (i64)k.cnt * n * n;
            cnt_mul += (double)k.cnt * (2.0 * n * n * n + 0.5 * (double)k.cnt * n * n);
        }
    }
    for (int j = 0; j < J; j++)
        if (c->clu[j].b0 > c->clu[j].b1) { c->clu[j].b0 = c->clu[j].b1 = 0; }
    {
        int mostb = 1;
        for (int j = 0; j < J; j++) mostb = std::max(mostb, c->clu[j].b1 - c->clu[j].b0);
        c->sa_lanes = mostb >= 3 ? 4 : mostb;
        // (one lane per entry when the launch fills the chip anyway was tried at 2048 clusters: 30 M wave instructions instead of 90 M, and
        // 390 us instead of 245: the chains of dependent loads of a cluster's blocks, one after the other, cost more than the idle lanes)
        // clusters with at most four blocks and at most one low-rank term per (constraint, block): k_mw_saccum_one
        for (int j = 0; j < J; j++) {
            MwClu &cl = c->clu[j];
            bool one = J >= 32 && cl.b1 - cl.b0 >= 1 && cl.b1 - cl.b0 <= 4;      // (with a few clusters the launch is latency bound and a lane per block wins: 6.9 against 9.1 us on the named problem)
            for (int b = cl.b0; b < cl.b1 && one; b++) {
                const MwBlk &k = c->blk[b];
                if (k.kind != 0) continue;
                const int *tp = htptr.data() + k.tptr_off;
                for (int p = 0; p < cl.P && one; p++) one = tp[p + 1] - tp[p] <= 1;
            }
            cl.one_term = one ? 1 : 0;
            (one ? c->n_one_term : c->n_many_term)++;
        }
    }
    c->cnt_mul = cnt_mul;
    for (int j = 0; j < J; j++) {
        const double P = c->clu[j].P;
        c->cnt_factor += P * P * P / 6.0 + 0.5 * P * P * N + 0.5 * P * (double)N * N;
        c->cnt_solve += P * P + 2.0 * P * N;
    }
    c->cnt_factor += (double)N * N * N / 6.0;
    c->cnt_solve += (double)N * N;
    // ---- LDS plans ----
    const size_t lim = MW_LDS_MAX / sizeof(double);
    {
        size_t nn = (size_t)c->maxn * c->maxn * K;
        size_t bcw = MW_POTRF_SCR(K, (size_t)c->maxn);       // scratch of wg_potrf
        c->lds_x = nn + bcw <= lim;
        size_t xneed = (c->lds_x ? nn : 0) + bcw;
        for (auto &k : c->blk) {                            // X blocks whose factor and its inverse fit side by side: Xi is formed
            const size_t one = (size_t)k.n * k.n * K + MW_POTRF_SCR(K, (size_t)k.n) + (size_t)K * k.n;   // + a reciprocal diagonal (the Y workgroups of the iteration)
            const size_t two = one + (size_t)MW_TRI(k.n) * K;                // + the packed inverse factor
            k.inv = !c->lds_x ? 0 : two <= lim ? 1 : one <= lim ? 2 : 0;       // the inverse in LDS beside the factor, or in place in memory
            if (k.inv) xneed = std::max(xneed, k.inv == 1 ? two : one);
        }
        c->all_inv = !c->blk.empty();
        for (auto &k : c->blk) c->all_inv = c->all_inv && (k.kind != 0 || k.inv != 0);
        c->sm_x = xneed * 8;
        size_t zt = (size_t)c->maxn * MW_CT * K;
        c->lds_zt_L = zt + nn <= lim;
        c->sm_zt = (zt + (c->lds_zt_L ? nn : 0)) * 8;
        if (zt > lim) MW_BAIL(CLRS_ERR_INVALID, "PSD block too large for the multi-word kernels");
        size_t maxnd = 0;
        for (auto &k : c->blk) if (k.kind != 0 && k.n > 1) maxnd = std::max(maxnd, (size_t)k.n);
        if (maxnd * maxnd * K > lim) MW_BAIL(CLRS_ERR_INVALID, "dense block too large for the multi-word kernels");
        c->dense_two = 2 * maxnd * maxnd * K <= lim;        // room for the two buffers of the product form of X^-1 A
        c->sm_dense = (c->dense_two ? 2 : 1) * maxnd * maxnd * K * 8;
        size_t fmax = 0;
        for (auto &q : c->clu) {                        // S_j and the inverse of its factor side by side in LDS, or the blocked path
            const size_t need = ((size_t)q.P * q.P + (size_t)MW_TRI(q.P)) * K + MW_POTRF_SCR(K, (size_t)q.P);
            q.lds = need <= lim ? 1 : 0;
            if (q.lds) fmax = std::max(fmax, need);
        }
        c->sm_factor = std::max<size_t>(fmax, 1) * 8;
        c->nw_factor = std::max(1, std::min(MW_INV_WG, 256 / std::max(J, 1)));     // only while the clusters leave compute units idle
        c->sm_fwd = 2 * (size_t)c->maxP * K * 8;
        c->sm_bwd = 3 * (size_t)c->maxP * K * 8;              // (three vectors: the refinement's first half rides on the backward launch)
        const size_t qn = (size_t)N * N * K;
        const size_t qneed = qn + (size_t)MW_TRI(N) * K + MW_POTRF_SCR(K, (size_t)N);
        c->lds_q = qneed <= lim;
        c->sm_q = (c->lds_q ? qneed : 1) * 8;
        c->sm_mid = 2 * (size_t)std::max(N, 1) * K * 8;
        if (c->sm_fwd > MW_LDS_MAX || c->sm_mid > MW_LDS_MAX) MW_BAIL(CLRS_ERR_INVALID, "cluster too large for the multi-word solve kernels");
    }
    MW_DISPATCH(c, {
        MW_TRY(mw_set_lds(k_mw_potrf_x<KK>, c->sm_x)); MW_TRY(mw_set_lds(k_mw_zt<KK, DD>, c->sm_zt)); MW_TRY(mw_set_lds((k_mw_dense_t<KK, DD>), c->sm_dense));
        // (only what can be launched: Q beyond LDS never rides k_mw_potrf_q, and systems whose dy and three cluster vectors exceed LDS take the
        // row-parallel solve (k_mw_solve_wide) -- sharded, they are refused at the launch, not here)
        constexpr int KC = mw_kc(KK);
        const size_t bw = std::min(c->sm_mid + c->sm_bwd, MW_LDS_MAX);
        MW_TRY(mw_set_lds(k_mw_factor<KK>, c->sm_factor)); MW_TRY(mw_set_lds(k_mw_potrf_q<KK>, c->lds_q ? std::max(c->sm_q, c->sm_fwd) : c->sm_fwd));
        MW_TRY(mw_set_lds(k_mw_solve_fwd<KK>, c->sm_fwd)); MW_TRY(mw_set_lds((k_mw_solve_mid<KK, KK>), c->sm_mid)); MW_TRY(mw_set_lds((k_mw_solve_mid<KK, KC>), c->sm_mid));
        MW_TRY(mw_set_lds((k_mw_solve_bwd<KK, KK, DD, 0>), bw)); MW_TRY(mw_set_lds((k_mw_solve_bwd<KK, KK, DD, 1>), bw)); MW_TRY(mw_set_lds((k_mw_solve_bwd<KK, KK, DD, 2>), bw));
        MW_TRY(mw_set_lds((k_mw_solve_bwd<KK, KC, DD, 1>), bw)); MW_TRY(mw_set_lds((k_mw_solve_bwd<KK, KC, DD, 2>), bw));
    });
    // ---- exact-product path (clrs_mw_exact.hip.h): static slices of V of the eligible blocks ----
    std::vector<long long> mws_off((size_t)std::max(NB, 1), -1);
    {
        const int S = mws_slices(K);
        std::vector<float> hVs;
        std::vector<int> hVe, hve_off((size_t)std::max(NB, 1), 0), hsv((size_t)std::max(NB, 1), 0);
        for (int b = 0; b < NB; b++) {
            const MwBlk &k = c->blk[b];
            if (k.kind != 0 || !k.inv || k.n > 32) continue;
            const int n = k.n, U = k.U, np = (n + 3) & ~3, n16 = (n + 15) & ~15, U16 = (U + 15) & ~15, rV = mws_rowstride(U16);
            if ((n16 / 16) * (U16 / 16) > 4 || mws_lds_bytes(S, mws_sv_class(S, 1), n, U) > MW_LDS_MAX) continue;
            if (2 * (n16 / 16) * (U16 / 16) > 4) c->mws_turns = 2;
            mws_off[b] = (long long)hVs.size();
            hve_off[b] = (int)hVe.size();
            hVs.resize(hVs.size() + (size_t)S * np * rV, 0.0f);
            hVe.resize(hVe.size() + U16, 0);
            float *dst = hVs.data() + mws_off[b];
            int sv = 0;
            for (int u = 0; u < U; u++) {
                double mx = 0;
                for (int i = 0; i < n; i++) mx = std::max(mx, std::fabs(hVl[0][k.v_off + (i64)u * n + i]));
                const int e = mwk::mws_exponent(mx);
                hVe[hve_off[b] + u] = e;
                for (int i = 0; i < n; i++) {
                    mwa::mw<2> x;
                    x.l[0] = hVl[0][k.v_off + (i64)u * n + i];
                    x.l[1] = DK > 1 ? hVl[1][k.v_off + (i64)u * n + i] : 0.0;
                    double r0 = std::ldexp(x.l[0], -e), r1 = std::ldexp(x.l[1], -e);
                    for (int s = 0; s < S; s++) {
                        const double g = std::ldexp(1.0, -(s + 1) * MWS_BETA), C = 0x1.8p52 * g;
                        volatile double tv = r0 + C;               // (no contraction or reassociation of the rounding trick on the host)
                        const double t = tv - C;
                        const float dgt = (float)(t * std::ldexp(1.0, (s + 1) * MWS_BETA));
                        dst[(size_t)s * np * rV + (size_t)i * rV + mws_col(i, u, U16)] = dgt;      // [s][k = row i of V][col = u]
                        if (dgt != 0.0f) sv = std::max(sv, s + 1);
                        r0 -= t;
                        double sm, er;
                        mwa::two_sum(r0, r1, sm, er);
                        r0 = sm; r1 = er;
                    }
                }
            }
            if (mws_lds_bytes(S, mws_sv_class(S, sv), n, U) > MW_LDS_MAX) {      // (the test above assumed the fewest slices of V)
                hVs.resize((size_t)mws_off[b]); hVe.resize((size_t)hve_off[b]);
                mws_off[b] = -1;
                continue;
            }
            hsv[b] = sv;
            c->mws_blocks++;
            c->sm_mws = std::max(c->sm_mws, mws_lds_bytes(S, mws_sv_class(S, sv), n, U));
        }
        const bool on = cfg_exact == 2 ? c->mws_blocks > 0 : (cfg_exact == 1 && c->mws_blocks >= 256);
        if (!on) { c->mws_blocks = 0; std::fill(mws_off.begin(), mws_off.end(), -1); }
        else {
            MW_TRY(mw_upload(c, hVs, &c->mws.Vs)); MW_TRY(mw_upload(c, hVe, &c->mws.Vexp));
            MW_TRY(mw_upload(c, hve_off, &c->mws.ve_off)); MW_TRY(mw_upload(c, hsv, &c->mws.sv));
            MW_DISPATCH(c, { if constexpr (DD <= 2) { MW_TRY(mw_set_lds((k_mws_pair<KK, DD, 1>), c->sm_mws)); MW_TRY(mw_set_lds((k_mws_pair<KK, DD, 2>), c->sm_mws)); } });
        }
        MW_TRY(mw_upload(c, mws_off, &c->mws.vs_off));
    }
    // ---- pairing matrices of blocks of any size through digits in global memory (k_mwx_slice, k_mwx_gram): blocks without sub-blocks that
    // k_mws_pair does not take and that have enough unique vectors for their U x U Gram products to matter ----
    std::vector<long long> mwx_off((size_t)std::max(NB, 1), -1);
    if (cfg_exact != 0) {
        const int S = mws_slices(K);
        std::vector<float> hVd;
        std::vector<int> heV, he_off((size_t)std::max(NB, 1), 0), hsv((size_t)std::max(NB, 1), 0);
        for (int b = 0; b < NB; b++) {
            const MwBlk &k = c->blk[b];
            if (k.kind != 0 || k.m != 1 || mws_off[b] >= 0) continue;
            if (k.U < (cfg_exact == 2 ? 16 : 64) || k.n < 8) continue;
            const int n = k.n, U = k.U, np = (n + 3) & ~3, U16 = (U + 15) & ~15;
            mwx_off[b] = (long long)hVd.size();
            he_off[b] = (int)heV.size();
            hVd.resize(hVd.size() + (size_t)S * np * U16, 0.0f);
            heV.resize(heV.size() + U16, 0);
            float *dst = hVd.data() + mwx_off[b];
            int sv = 0;
            for (int u = 0; u < U; u++) {
                double mx = 0;
                for (int i = 0; i < n; i++) mx = std::max(mx, std::fabs(hVl[0][k.v_off + (i64)u * n + i]));
                const int e = mwk::mws_exponent(mx);
                heV[he_off[b] + u] = e;
                for (int i = 0; i < n; i++) {
                    double r0 = std::ldexp(hVl[0][k.v_off + (i64)u * n + i], -e), r1 = DK > 1 ? std::ldexp(hVl[1][k.v_off + (i64)u * n + i], -e) : 0.0;
                    for (int s2 = 0; s2 < S; s2++) {
                        const double g = std::ldexp(1.0, -(s2 + 1) * MWS_BETA), C = 0x1.8p52 * g;
                        volatile double tv = r0 + C;
                        const double t = tv - C;
                        const float dgt = (float)(t * std::ldexp(1.0, (s2 + 1) * MWS_BETA));
                        dst[(size_t)s2 * np * U16 + (size_t)i * U16 + u] = dgt;
                        if (dgt != 0.0f) sv = std::max(sv, s2 + 1);
                        r0 -= t;
                        double sm, er;
                        mwa::two_sum(r0, r1, sm, er);
                        r0 = sm; r1 = er;
                    }
                }
            }
            hsv[b] = sv;
            c->mwx_blocks++;
            c->mwx_maxU16 = std::max(c->mwx_maxU16, U16);
        }
        if (c->mwx_blocks > 0) {
            MW_TRY(mw_upload(c, hVd, &c->mwx.Vd)); MW_TRY(mw_upload(c, heV, &c->mwx.eV));
            MW_TRY(mw_upload(c, he_off, &c->mwx.e_off)); MW_TRY(mw_upload(c, hsv, &c->mwx.sv));
            double *zd = nullptr, *td = nullptr, *ez = nullptr, *et = nullptr;      // (the allocator counts doubles)
            MW_TRY(mw_dmalloc(c, &zd, (i64)(hVd.size() + 1) / 2)); MW_TRY(mw_dmalloc(c, &td, (i64)(hVd.size() + 1) / 2));
            MW_TRY(mw_dmalloc(c, &ez, (i64)(heV.size() + 1) / 2)); MW_TRY(mw_dmalloc(c, &et, (i64)(heV.size() + 1) / 2));
            c->mwx.Zd = (float *)zd; c->mwx.Td = (float *)td; c->mwx.eZ = (int *)ez; c->mwx.eT = (int *)et;
            MWCHECK(hipMemset(zd, 0, hVd.size() * sizeof(float))); MWCHECK(hipMemset(td, 0, hVd.size() * sizeof(float)));      // the padding rows and columns stay zero
            MWCHECK(hipMemset(ez, 0, heV.size() * sizeof(int))); MWCHECK(hipMemset(et, 0, heV.size() * sizeof(int)));
        }
        MW_TRY(mw_upload(c, mwx_off, &c->mwx.d_off));
    } else {
        MW_TRY(mw_upload(c, mwx_off, &c->mwx.d_off));
    }
    // ---- dense blocks with 16 < n <= 32 and inverse factors: X^-1 (A_e Y) through exact slice products (k_mwx_dense), static digits of the A_e ----
    std::vector<long long> mwd_off((size_t)std::max(NB, 1), -1);
    if (cfg_exact != 0 && K <= 6) {
        const int S = mws_slices(K), S1 = (S + 1) / 2, sN = 32 * 32;
        std::vector<float> hAd;
        std::vector<int> heA, he_off((size_t)std::max(NB, 1), 0);
        for (int b = 0; b < NB; b++) {
            const MwBlk &k = c->blk[b];
            if (k.kind == 0 || !k.inv || k.n <= 16 || k.n > 32 || k.cnt <= 0) continue;
            const int n = k.n;
            const size_t base = hAd.size(), ebase = heA.size();
            hAd.resize(base + (size_t)k.cnt * S1 * sN, 0.0f);
            heA.resize(ebase + (size_t)k.cnt * 32, 0);
            bool fits = true;
            for (int e = 0; e < k.cnt && fits; e++) {
                float *dst = hAd.data() + base + (size_t)e * S1 * sN;
                const i64 a0 = k.a_off + (i64)e * n * n;
                for (int col = 0; col < n; col++) {
                    double mx = 0;
                    for (int kk = 0; kk < n; kk++) mx = std::max(mx, std::fabs(hdAl[0][a0 + kk + (i64)col * n]));
                    const int ex = mwk::mws_exponent(mx);
                    heA[ebase + (size_t)e * 32 + col] = ex;
                    for (int kk = 0; kk < n; kk++) {
                        double r0 = std::ldexp(hdAl[0][a0 + kk + (i64)col * n], -ex), r1 = DK > 1 ? std::ldexp(hdAl[1][a0 + kk + (i64)col * n], -ex) : 0.0;
                        for (int s2 = 0; s2 < S; s2++) {
                            const double g = std::ldexp(1.0, -(s2 + 1) * MWS_BETA), C = 0x1.8p52 * g;
                            volatile double tv = r0 + C;
                            const double t = tv - C;
                            const float dgt = (float)(t * std::ldexp(1.0, (s2 + 1) * MWS_BETA));
                            if (s2 < S1) dst[(size_t)s2 * sN + (size_t)kk * 32 + mws_col(kk, col, 32)] = dgt;
                            else if (dgt != 0.0f) fits = false;      // data beyond the slices kept: leave the block to the expansion kernels
                            r0 -= t;
                            double sm, er;
                            mwa::two_sum(r0, r1, sm, er);
                            r0 = sm; r1 = er;
                        }
                    }
                }
            }
            if (!fits) { hAd.resize(base); heA.resize(ebase); continue; }
            mwd_off[b] = (long long)base;
            he_off[b] = (int)ebase;
            c->mwd_blocks++;
        }
        if (c->mwd_blocks > 0) {
            std::vector<int> tasks;
            for (size_t di = 0; di < dn_list.size(); di++) {
                const int b = dn_list[di];
                if (mwd_off[b] < 0) continue;
                for (int e = 0; e < c->blk[b].cnt; e++) { tasks.push_back((int)di); tasks.push_back(e); }
            }
            c->mwd_tasks = (int)tasks.size() / 2;
            MW_TRY(mw_upload(c, tasks, &c->mwd.tasks));
            MW_TRY(mw_upload(c, hAd, &c->mwd.Ad)); MW_TRY(mw_upload(c, heA, &c->mwd.eA)); MW_TRY(mw_upload(c, he_off, &c->mwd.e_off));
            c->sm_mwd = (size_t)2 * S * sN * sizeof(float) + 128 * sizeof(int);
            MW_DISPATCH(c, { if constexpr (DD <= 2) { MW_TRY(mw_set_lds((k_mwx_dense<KK, DD>), c->sm_mwd)); } });
        }
    }
    MW_TRY(mw_upload(c, mwd_off, &c->mwd.a_off));
    // ---- upload ----
    MwDev &q = c->d;
    q.mws_off = c->mws.vs_off; q.mws_on = c->mws_blocks > 0 ? 1 : 0;
    q.mwx_off = c->mwx.d_off; q.mwx_on = c->mwx_blocks > 0 ? 1 : 0;
    q.mwd_off = c->mwd.a_off; q.mwd_on = 0;            // (switched on per assembly: the kernel needs the inverse factors of this context's chol X)
    q.J = J; q.N = N; q.NB = NB; q.nlr = (int)lr_list.size(); q.ndn = (int)dn_list.size();
    q.xylen = xyoff; q.xlen = xlen; q.Slen = Slen; q.T = T; q.xrdlen = rdoff;
    q.zlen = std::max<i64>(zoff, 1); q.glen = std::max<i64>(goff, 1); q.wlen = std::max<i64>(woff, 1); q.sdlen = std::max<i64>(sdoff, 1);
    std::vector<int> hdense_p(d->dense_p, d->dense_p + D);
    // B arrives per cluster (P_j x N column-major, concatenated); the kernels read one stacked xlen x N matrix
    const i64 Bp = xlen * (i64)N;
    std::vector<double> hBs((size_t)std::max<i64>(Bp, 1) * DK, 0.0);
    for (int l = 0; l < DK; l++) {
        i64 off = 0;
        for (int j = 0; j < J; j++) {
            const int P = c->clu[j].P;
            for (int a = 0; a < N; a++)
                for (int r = 0; r < P; r++) hBs[(size_t)l * std::max<i64>(Bp, 1) + c->clu[j].coff + r + (i64)a * xlen] = d->B[(i64)l * Bp + off + r + (i64)a * P];
            off += (i64)P * N;
        }
    }
    q.Vp = std::max<i64>((i64)hVl[0].size(), 1); q.dAp = std::max<i64>((i64)hdAl[0].size(), 1); q.lamp = std::max<i64>(T, 1); q.Bp = std::max<i64>(Bp, 1);
    hV.assign((size_t)q.Vp * DK, 0.0);
    hdA.assign((size_t)q.dAp * DK, 0.0);
    for (int l = 0; l < DK; l++) {
        std::copy(hVl[l].begin(), hVl[l].end(), hV.begin() + (size_t)l * q.Vp);
        std::copy(hdAl[l].begin(), hdAl[l].end(), hdA.begin() + (size_t)l * q.dAp);
    }
    MW_TRY(mw_upload(c, c->blk, &q.blk)); MW_TRY(mw_upload(c, c->clu, &q.clu));
    MW_TRY(mw_upload(c, lr_list, &q.lr_list)); MW_TRY(mw_upload(c, dn_list, &q.dn_list));
    MW_TRY(mw_upload(c, hV, &q.V)); MW_TRY(mw_upload(c, hvrow, &q.vrow));
    MW_TRY(mw_upload(c, st_a, &q.st_a)); MW_TRY(mw_upload(c, st_b, &q.st_b)); MW_TRY(mw_upload(c, st_lam, &q.st_lam));
    MW_TRY(mw_upload(c, htptr, &q.tptr));
    MW_TRY(mw_upload(c, st_orig, &q.st_orig)); MW_TRY(mw_upload(c, st_p, &q.st_p)); MW_TRY(mw_upload(c, st_war, &q.st_war)); MW_TRY(mw_upload(c, st_wac, &q.st_wac));
    MW_TRY(mw_upload(c, st_trl, &q.st_trl)); MW_TRY(mw_upload(c, st_trd, &q.st_trd)); MW_TRY(mw_upload(c, st_flag, &q.st_flag));
    MW_TRY(mw_upload(c, ay_a, &q.ay_a)); MW_TRY(mw_upload(c, ay_b, &q.ay_b)); MW_TRY(mw_upload(c, ay_blk, &q.ay_blk));
    MW_TRY(mw_upload(c, hdA, &q.dA)); MW_TRY(mw_upload(c, hdmap, &q.dmap)); MW_TRY(mw_upload(c, hdense_p, &q.dense_p));
    {
        std::sort(drow_pairs.begin(), drow_pairs.end());
        std::vector<int> hptr((size_t)xlen + 1, 0), hblk, hen;
        for (auto &t : drow_pairs) hptr[(size_t)std::get<0>(t) + 1]++;
        for (i64 g = 0; g < xlen; g++) hptr[(size_t)g + 1] += hptr[(size_t)g];
        for (auto &t : drow_pairs) { hblk.push_back(std::get<1>(t)); hen.push_back(std::get<2>(t)); }
        MW_TRY(mw_upload(c, hptr, &q.drow_ptr)); MW_TRY(mw_upload(c, hblk, &q.drow_blk)); MW_TRY(mw_upload(c, hen, &q.drow_en));
        q.dn_big = 0;
        for (auto &k : c->blk) if (k.kind != 0) { c->maxcnt = std::max(c->maxcnt, k.cnt); if (k.n > 1) q.dn_big = 1; }
        q.maxcnt = c->maxcnt;
    }
    MW_TRY(mw_upload(c, hBs, &q.B));
    MW_TRY(mw_dmalloc(c, &q.Z, q.zlen * K)); MW_TRY(mw_dmalloc(c, &q.Tm, q.zlen * K));
    MW_TRY(mw_dmalloc(c, &q.GX, q.glen * K)); MW_TRY(mw_dmalloc(c, &q.GY, q.glen * K));
    MW_TRY(mw_dmalloc(c, &q.W, q.wlen * K)); MW_TRY(mw_dmalloc(c, &q.Sd, q.sdlen * K));
    MW_TRY(mw_dmalloc(c, &q.S, Slen * K)); MW_TRY(mw_dmalloc(c, &q.LB, xlen * (i64)N * K)); MW_TRY(mw_dmalloc(c, &q.Q, (i64)N * N * K));
    MW_TRY(mw_dmalloc(c, &q.Si, Slen * K)); MW_TRY(mw_dmalloc(c, &q.Qi, (i64)N * N * K));

    MW_TRY(mw_dmalloc(c, &q.Xf, xyoff * K)); MW_TRY(mw_dmalloc(c, &q.Xb, xyoff * K)); MW_TRY(mw_dmalloc(c, &q.Xi, xyoff * K));
    MW_TRY(mw_dmalloc(c, &q.xrd, rdoff * K)); MW_TRY(mw_dmalloc(c, &q.srd, xlen * K)); MW_TRY(mw_dmalloc(c, &q.qrd, (i64)N * K));
    MW_TRY(mw_dmalloc(c, &q.t, xlen * K)); MW_TRY(mw_dmalloc(c, &q.u, (i64)J * N * K)); MW_TRY(mw_dmalloc(c, &q.AY, T * K));
    q.AX = nullptr;                      // (the interior-point iteration asks for it: clrs_mw_ipm_create)
    q.aff_mu = q.aff_rhs = q.aff_t = q.aff_u = nullptr; q.aff_wait = nullptr; q.ride2_rhs = nullptr; q.ride2_t = q.ride2_u = nullptr;
    MW_TRY(mw_dmalloc(c, &c->vz, 2 * (i64)N * K));
    MW_TRY(mw_dmalloc(c, &q.S0, Slen * K)); MW_TRY(mw_dmalloc(c, &q.ub, (i64)J * N * K));
    MW_TRY(mw_dmalloc(c, &q.rx2, xlen * K)); MW_TRY(mw_dmalloc(c, &q.dx2, xlen * K));
    MW_TRY(mw_dmalloc(c, &q.u2, (i64)N * K)); MW_TRY(mw_dmalloc(c, &q.dy2, (i64)N * K));
    q.uadd = nullptr;
    c->refine = cfg_refine;
    c->refine_predictor = cfg_refine_pred;
    {
        const char *e = std::getenv("CLRS_MW_STREAM_WORDS");
        c->stream_words = e && *e ? std::atoi(e) != 0 : g_cfg_mw_stream_words != 0;
    }
    c->wide_solve = c->maxP > 64 || N > 64;
    MW_TRY(mw_dmalloc(c, &c->d_Xin, xyoff * K)); MW_TRY(mw_dmalloc(c, &c->d_Xc, xyoff * K)); MW_TRY(mw_dmalloc(c, &c->d_Y, xyoff * K));
    MW_TRY(mw_dmalloc(c, &c->d_rx, xlen * K)); MW_TRY(mw_dmalloc(c, &c->d_dx, xlen * K));
    MW_TRY(mw_dmalloc(c, &c->d_ry, (i64)N * K)); MW_TRY(mw_dmalloc(c, &c->d_dy, (i64)N * K));
    {
        int *info = nullptr;
        MWCHECK(hipMalloc((void **)&info, 2 * sizeof(int)));
        c->allocs.push_back(info);
        q.info = info;
        int init[2] = {MW_INFO_NONE, MW_INFO_NONE};
        MWCHECK(hipMemcpy(info, init, sizeof(init), hipMemcpyHostToDevice));
        int *pc = nullptr;
        MWCHECK(hipMalloc((void **)&pc, (size_t)(2 * J + 3) * sizeof(int)));
        c->allocs.push_back(pc);
        MWCHECK(hipMemset(pc, 0, (size_t)(2 * J + 3) * sizeof(int)));
        q.pcnt = pc;
    }
    {   // matrices of the blocked path (k_mw_bp_*)
        for (int j = 0; j < J; j++) {
            const MwClu &cl = c->clu[j];
            if (cl.lds) { c->any_lds_cluster = true; continue; }
            c->bp_S.push_back(MwBp{q.S + cl.Soff, q.Si + cl.Soff, q.srd + cl.coff, q.Slen, q.xlen, cl.P, cl.P, j + 1, 0, (int)c->bp_S.size(), 0});
        }
        if (N > 0) c->bp_Q.push_back(MwBp{q.Q, q.Qi, q.qrd, (i64)N * N, (i64)N, N, N, J + 1, 0, (int)c->bp_S.size(), 0});
        std::vector<MwBp> all = c->bp_S;
        all.insert(all.end(), c->bp_Q.begin(), c->bp_Q.end());
        MW_TRY(mw_upload(c, all, &c->d_bp));
    }
    {   // pipelined factorisations (clrs_mw_pipe.hip.h): matrices of at most 32 rows, while stages + W workgroups of every matrix can be resident side by side
        bool small = c->maxP <= MWP_N;
        for (auto &cl : c->clu) small = small && cl.lds;
        c->pipe_S = cfg_pipe != 0 && (K <= 6 || cfg_pipe >= 2) && small && (i64)J * ((MWP_N / MWP_W) + MWP_WW) <= 256;      // (8, 10 limbs: measured slower than one workgroup, 2.42 against 2.34 ms per iteration: opt-in)
        // (Q: with the columns asked for four steps ahead the pipeline was slower than the one-workgroup kernel on the named problem, 83 against 80 us; with
        // MWP_AHEAD = 0 it is faster -- whole iterations 0.4197 -> 0.4165 ms on cohnelkies(8,15), 0.4113 -> 0.4057 on delsarte(3,10,1/2): on from two stages,
        // and with pipeline = 2 for every Q of at most MWP_N rows)
        c->pipe_Q = (cfg_pipe >= 2 || (cfg_pipe == 1 && K <= 6 && N > MWP_W)) && N > 0 && N <= MWP_N;
        bool lds_all = !c->clu.empty();
        for (auto &cl : c->clu) lds_all = lds_all && cl.lds;
        // (measured at 5 limbs, factor stage in 4: PolyOpt 2d = 40, P = 41 -- six stages, five hops -- 0.478 -> 0.487 ms per iteration; the tested three-point instance,
        // P = 50, 0.648 -> 0.638: by default from 48 rows on, with pipeline = 2 for every cluster of 33 .. 64 rows)
        c->pipe_S64 = cfg_pipe != 0 && K <= 6 && lds_all && c->maxP > MWP_N && c->maxP <= MWP_N64 && (i64)J * 16 <= 256 && g_cfg_mw_pipeline64 != 0 &&
                      (cfg_pipe >= 2 || c->maxP >= 48);
        // the diagonal blocks of the blocked factorisation (k_mw_bp_diag_pipe): the same pipeline per 32-column block of every matrix beyond LDS
        const size_t nbp = c->bp_S.size() + c->bp_Q.size();
        const bool any_bp = !c->bp_S.empty() || (N > 0 && !c->lds_q);
        // (at every limb count: 8 and 10 limbs gain 2-4 % per iteration as well -- scripts/blocked_pipe_limbs.py -- unlike the clusters that fit in LDS)
        c->pipe_bp = cfg_pipe != 0 && any_bp && (i64)std::max<size_t>(c->bp_S.size(), 1) * ((MWP_N / MWP_W) + MWP_WW) <= 256;
        q.pipe_pc = nullptr;
        q.pipe_stamps = nullptr;
        q.pipe_bp = 0;
        {   // the Cholesky of the X blocks through the pipelines: every block carries its inverse factor in LDS form (inv == 1) and has at most 64 rows
            // (measured at 5 limbs, whole iterations: blocks of 32 and 48 rows -- Nsphere_packing N = 2 1.146 -> 1.122 ms, N = 3 2.059 -> 2.027; blocks of at most 21 rows:
            // nothing either way; 64 blocks of 32 rows -- 2 x 64 x 8 working workgroups for 256 compute units -- 1.69 -> 1.87: by default from 24 rows of the largest
            // block on and while every working workgroup of the launch has a compute unit of its own)
            bool all_inv = c->d.NB > 0 && c->lds_x;
            int maxn_x = 0;
            i64 working = 0;
            for (auto &kb : c->blk) {
                all_inv = all_inv && kb.inv == 1;
                maxn_x = std::max(maxn_x, kb.n);
                working += 2 * ((kb.n + MWP_W - 1) / MWP_W + (kb.n > MWP_N ? MWP_N64 / MWP_W : MWP_WW));
            }
            c->pipe_X = cfg_pipe != 0 && K <= 6 && all_inv && maxn_x <= MWP_N64 && g_cfg_mw_pipeline_x != 0 &&
                        (cfg_pipe >= 2 || (maxn_x >= g_cfg_mw_pipeline_x_min && working <= 256));
            if (c->pipe_X) {
                double *pcx = nullptr, *ti = nullptr;
                MW_TRY(mw_dmalloc(c, &pcx, (i64)2 * c->d.NB * MWP_PC_WORDS_N(K, MWP_N64)));
                c->pipe_pcx = (unsigned long long *)pcx;
                MW_TRY(mw_dmalloc(c, &c->xpipe_L, c->d.xylen * K));
                MW_TRY(mw_dmalloc(c, &c->xpipe_rd, c->d.xrdlen * K));
                MW_TRY(mw_dmalloc(c, &ti, c->d.NB + 1));
                c->xpipe_info = (int *)ti;
                MW_DISPATCH(c, { if constexpr (KK <= 6) { MW_TRY(mw_set_lds(k_mw_potrf_x_pipe<KK>, MWP_LDS_ALONE64)); } });
            }
        }
        if (c->pipe_S64) {                                    // (its own hand-off region: 64-row columns; no other pipeline runs in such a context's factor stage but Q's / the blocked path's below)
            const size_t words64 = (size_t)J * MWP_PC_WORDS_N(K, MWP_N64);
            unsigned long long *pc64 = nullptr;
            if (hipMalloc((void **)&pc64, words64 * sizeof(unsigned long long)) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipMalloc failed");
            c->allocs.push_back(pc64);
            if (hipMemset(pc64, 0xff, words64 * sizeof(unsigned long long)) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipMemset failed");
            c->pipe_pc64 = pc64;
            MW_DISPATCH(c, { if constexpr (KK <= 6) { MW_TRY(mw_set_lds(k_mw_factor_pipe64<KK>, MWP_LDS_ALONE64)); } });
        }
        if (c->pipe_S || c->pipe_Q || c->pipe_bp) {
            const size_t own = (size_t)(c->pipe_S ? J : 0) + 1;
            const size_t words = (own + (c->pipe_bp ? nbp : 0)) * MWP_PC_WORDS(K);
            unsigned long long *pc = nullptr;
            if (hipMalloc((void **)&pc, words * sizeof(unsigned long long)) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipMalloc failed");
            c->allocs.push_back(pc);
            if (hipMemset(pc, 0xff, words * sizeof(unsigned long long)) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipMemset failed");      // no launch epoch has this tag
            q.pipe_pc = pc;
            c->pipe_pcQ = c->pipe_S ? J : 0;
            q.pipe_q = c->pipe_pcQ;
            q.pipe_bp = (int)own;
            MW_DISPATCH(c, {
                MW_TRY(mw_set_lds(k_mw_factor_pipe<KK>, MWP_LDS_ALONE));
                MW_TRY(mw_set_lds(k_mw_potrf_q_pipe<KK>, std::max<size_t>(MWP_LDS_ALONE, c->sm_fwd)));
                MW_TRY(mw_set_lds(k_mw_bp_diag_pipe<KK>, std::max<size_t>(MWP_LDS_ALONE, c->sm_factor)));
            });
        }
    }
    const int MW_PB = MW_PB_OF(K);
    c->sm_bp_diag = ((size_t)MW_POTRF_SCR(K, MW_PB) + (size_t)K * MW_PB * MW_PB + (size_t)K * MW_TRI(MW_PB) + (size_t)K * MW_PB) * 8;
    c->sm_bp_panel = (size_t)K * MW_BP_PR * MW_PB * 8;
    c->sm_bp_inv = (size_t)K * MW_PB * MW_BP_IC * 8;
    MW_DISPATCH(c, { MW_TRY(mw_set_lds(k_mw_bp_diag<KK>, std::max(c->sm_bp_diag, c->sm_factor))); MW_TRY(mw_set_lds(k_mw_bp_panel<KK>, c->sm_bp_panel)); MW_TRY(mw_set_lds(k_mw_bp_inv<KK>, c->sm_bp_inv)); MW_TRY(mw_set_lds(k_mw_bp_inv_row<KK>, c->sm_bp_inv)); });
    q.rank = 0; q.world = 1; q.gathered = 0;
    MW_TRY(mw_dmalloc(c, &q.Qg, (i64)N * N * K)); MW_TRY(mw_dmalloc(c, &q.ug, (i64)N * K));
    {   // factor stage and solve products in fewer limbs, residuals and iterate in K (mw_kf_of): the LDS-resident paths with the full-limb refinement step
        bool all_lds = true;
        for (auto &cl : c->clu) all_lds = all_lds && cl.lds;
        (void)all_lds;
        const bool may = mw_kf_of(K) < K && c->refine == 1;      // (every factorisation and solve path carries the reduced form: LDS-resident, pipelined, blocked, row-parallel)
        c->kf_low = may ? mw_kf_of(K) : K;
        if (cfg_factor_limbs != 0 && cfg_factor_limbs != K && cfg_factor_limbs != c->kf_low) MW_BAIL(CLRS_ERR_INVALID, "factor_limbs: 0 (automatic), the context's limbs, or the reduced count of this limb count where the LDS-resident refined solve runs");
        if (cfg_factor_limbs == K) c->kf_low = K;
        c->kf_entry = cfg_factor_limbs == 0 ? K : cfg_factor_limbs;
        q.kf = c->kf_entry;
        q.km = cfg_km;
        unsigned long long *rs = nullptr;
        if (hipMalloc((void **)&rs, 4 * sizeof(unsigned long long)) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipMalloc failed");
        c->allocs.push_back(rs);
        if (hipMemset(rs, 0, 4 * sizeof(unsigned long long)) != hipSuccess) MW_BAIL(CLRS_ERR_HIP, "hipMemset failed");
        q.refstat = rs;
    }
    for (auto &e : c->ev) MWCHECK(hipEventCreate(&e));
    *out = c;
    return 0;
#undef MW_BAIL
#undef MW_TRY
}

extern "C" void clrs_mw_destroy(clrs_mw_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    mw_ipm_free(c);
    if (c->comm && g_nccl.CommDestroy) { (void)g_nccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm_side && g_nccl.CommDestroy) { (void)g_nccl.CommDestroy(c->comm_side); c->comm_side = nullptr; }
    if (c->lgroup) { std::lock_guard<std::mutex> lk(c->lgroup->m); c->lgroup->attached--; c->lgroup = nullptr; }
    for (void *p : c->allocs) (void)hipFree(p);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->h_info) (void)hipHostFree(c->h_info);
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int clrs_mw_limbs(const clrs_mw_ctx *c) { return c ? c->K : 0; }
extern "C" void *clrs_mw_stream(clrs_mw_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int clrs_mw_set_stream(clrs_mw_ctx *c, void *stream) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
    return 0;
}
extern "C" int clrs_mw_get_dims(const clrs_mw_ctx *c, clrs_dims *o) {
    if (!c || !o) return mw_fail(CLRS_ERR_INVALID, "null argument");
    o->xy_len = c->d.xylen; o->x_len = c->d.xlen; o->S_len = c->d.Slen; o->n_terms = c->d.T;
    o->n_free = c->d.N; o->n_clusters = c->d.J; o->n_blocks = c->d.NB; o->reserved = c->K;
    return 0;
}
extern "C" int clrs_mw_get_unique_count(const clrs_mw_ctx *c, int32_t block, int32_t *n_unique) {
    if (!c || !n_unique || block < 0 || block >= c->d.NB) return mw_fail(CLRS_ERR_INVALID, "bad argument");
    *n_unique = c->blk[block].U;
    return 0;
}
extern "C" int clrs_mw_set_timing(clrs_mw_ctx *c, int on) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    c->timing = on != 0;
    return 0;
}
extern "C" int clrs_mw_get_counters(const clrs_mw_ctx *c, double *assemble_muladds, double *factor_muladds, double *solve_muladds) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (assemble_muladds) *assemble_muladds = c->cnt_mul;
    if (factor_muladds) *factor_muladds = c->cnt_factor;
    if (solve_muladds) *solve_muladds = c->cnt_solve;
    return 0;
}

static int mw_reset_info(clrs_mw_ctx *c, int which) {
    if (c->ipm_arms_info) return 0;
    MWCHECK(hipMemsetAsync(c->d.info + which, 0x7f, sizeof(int), c->stream));      // MW_INFO_NONE is the byte 0x7f four times
    return 0;
}
static int mw_read_info(clrs_mw_ctx *c, int which, int *status) {
    MWCHECK(hipMemcpyAsync(c->h_info + which, c->d.info + which, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MWCHECK(hipStreamSynchronize(c->stream));
    *status = c->h_info[which] == MW_INFO_NONE ? 0 : c->h_info[which];
    return 0;
}

// ---- the path, device pointers (planar K x len arrays), enqueue only ------------------------------------------------
// d_Y2 (optional): a second block-diagonal matrix whose inverse Cholesky factors go to d_Yi, failures per block to d_yfail
static int mw_cholesky_blocks_dev2(clrs_mw_ctx *c, const double *d_X, double *d_Xchol, const double *d_Y2, double *d_Yi, int *d_yfail) {
    if (!c || !d_X || !d_Xchol) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = mw_reset_info(c, 1))) return rc;
    if (c->d.NB == 0) return 0;
    const int grid = d_Y2 ? 2 * c->d.NB : c->d.NB;
    bool in_mem = false;                                  // some block forms its inverse factor in memory, or in LDS beside a large block: four workgroups share its columns
    for (auto &k : c->blk) in_mem = in_mem || k.inv == 2 || mw_x_shares(k);
    if (c->pipe_X) {
        c->pipe_epoch = (c->pipe_epoch + 1) & 0x3ffffff;
        MW_DISPATCH(c, { if constexpr (KK <= 6) { hipLaunchKernelGGL(k_mw_potrf_x_pipe<KK>, dim3(mwp_blocks64(grid)), dim3(MWP_NT64), MWP_LDS_ALONE64, c->stream, c->d, d_X, d_Xchol, d_Y2, d_Yi, d_yfail,
                                                                     c->xpipe_L, c->xpipe_rd, c->xpipe_info, c->pipe_pcx, c->pipe_epoch); } });
    } else
    MW_DISPATCH(c, hipLaunchKernelGGL(k_mw_potrf_x<KK>, dim3(grid, in_mem && c->lds_x ? MW_INV_WG : 1), dim3(MW_PT), c->sm_x, c->stream, c->d, d_X, d_Xchol, c->lds_x ? 1 : 0, d_Y2, d_Yi, d_yfail));
    MWCHECK(hipGetLastError());
    c->xinv_valid = true;
    return 0;
}
extern "C" int clrs_mw_cholesky_blocks_dev(clrs_mw_ctx *c, const double *d_X, double *d_Xchol) { return mw_cholesky_blocks_dev2(c, d_X, d_Xchol, nullptr, nullptr, nullptr); }
extern "C" int clrs_mw_sync_status_cholesky(clrs_mw_ctx *c) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    int st = 0, rc;
    if ((rc = mw_read_info(c, 1, &st))) return rc;
    return st;
}

extern "C" int clrs_mw_schur_assemble_dev(clrs_mw_ctx *c, const double *d_Xchol, const double *d_Y) {
    if (!c || !d_Xchol || !d_Y) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    const MwDev &q = c->d;
    if (c->timing) MWCHECK(hipEventRecord(c->ev[0], c->stream));
    MW_DISPATCH(c, {
        const bool exact = c->mws_blocks > 0 && c->xinv_valid;      // the exact-product kernel needs chol(X)^-1 (k_mw_potrf_x of this context)
        bool dense_done = exact && q.ndn && !q.dn_big;          // 1 x 1 dense blocks ride on the launch of the exact pairing matrices ...
        const int pair_grid = q.nlr + (dense_done ? q.ndn : 0);
        if constexpr (DD <= 2) {                             // (data at the working precision: the exact slice products are off, clrs_mw_create_opts)
            if (exact && c->mws_turns == 1) hipLaunchKernelGGL((k_mws_pair<KK, DD, 1>), dim3(pair_grid), dim3(MWS_NT), c->sm_mws, c->stream, q, c->mws, d_Y);
            else if (exact) hipLaunchKernelGGL((k_mws_pair<KK, DD, 2>), dim3(pair_grid), dim3(MWS_NT), c->sm_mws, c->stream, q, c->mws, d_Y);
        }
        if (q.nlr && !(exact && c->mws_blocks == q.nlr)) {
            MwDev q2 = q;
            q2.mws_on = exact ? 1 : 0;
            const int gper = MW_NT / MW_GRAM_W;
            // few blocks, every one with its inverse factor: two columns per workgroup and eight lanes per entry
            const int zt_ct = (c->xinv_valid && c->all_inv && (i64)q.nlr * c->maxU <= 2048 && c->maxn <= g_cfg_mw_zt_small_maxn) ? 2 : MW_CT;
            hipLaunchKernelGGL((k_mw_zt<KK, DD>), dim3((c->maxU + zt_ct - 1) / zt_ct, q.nlr), dim3(MW_NT), c->sm_zt, c->stream, q2, d_Y, c->lds_zt_L ? 1 : 0, c->xinv_valid ? 1 : 0, zt_ct);
            const bool ride_gram = !dense_done && q.ndn && !q.dn_big;      // ... or on that of the expansion kernel
            dense_done = dense_done || ride_gram;
            q2.mwx_on = c->mwx_blocks > 0 ? 1 : 0;
            hipLaunchKernelGGL((k_mw_gram<KK, DD>), dim3((c->maxU * (c->maxU + 1) / 2 + gper - 1) / gper, q.nlr + (ride_gram ? q.ndn : 0)), dim3(MW_NT), 0, c->stream, q2, d_Y);
            if (q2.mwx_on) {                                 // the large pairing matrices: digits of Z and T, then 16 x 16 tiles on the matrix cores
                const int nt = c->mwx_maxU16 / 16;
                hipLaunchKernelGGL(k_mwx_slice<KK>, dim3(nt, 2, q.nlr), dim3(MWS_NT), 0, c->stream, q2, c->mwx);
                hipLaunchKernelGGL(k_mwx_gram<KK>, dim3((nt * (nt + 1) / 2 + MWS_NT / 64 - 1) / (MWS_NT / 64), 2, q.nlr), dim3(MWS_NT), 0, c->stream, q2, c->mwx);
            }
        }
        if (q.ndn && !dense_done) {
            const bool panels = c->xinv_valid && c->maxn_dense > 16 && c->maxn_dense <= MW_NT;      // dense blocks of side > 16 with inverse factors: column panels (k_mw_dense_tp)
            MwDev q3 = q;
            q3.mwd_on = (panels && c->mwd_blocks > 0) ? 1 : 0;     // ... or exact slice products for sides up to 32 (k_mwx_dense)
            hipLaunchKernelGGL((k_mw_dense_t<KK, DD>), dim3(q.ndn, q.dn_big ? c->maxcnt : 1), dim3(MW_NT), c->sm_dense, c->stream, q3, d_Y, c->xinv_valid ? 1 : 0, c->dense_two ? 1 : 0, panels ? 1 : 0);
            if (panels) {
                const int pcmin = std::max(1, MW_NT / c->maxn_dense);
                if (c->mwd_blocks < q.ndn)
                    hipLaunchKernelGGL((k_mw_dense_tp<KK, DD>), dim3(q.ndn, c->maxcnt, (c->maxn_dense + pcmin - 1) / pcmin), dim3(MW_NT), (size_t)2 * KK * MW_NT * 8, c->stream, q3, d_Y);
                c->mwd.stamps = c->mws.stamps;
                if constexpr (DD <= 2) { if (q3.mwd_on) hipLaunchKernelGGL((k_mwx_dense<KK, DD>), dim3(c->mwd_tasks), dim3(MWS_NT), c->sm_mwd, c->stream, q3, c->mwd, d_Y); }
            }
            const int pairs = c->maxcnt * (c->maxcnt + 1) / 2;
            const int ds_lanes = c->maxn_dense * c->maxn_dense <= 128 ? 8 : c->maxn_dense * c->maxn_dense <= 512 ? 16 : 64;      // per pair of the table
            if (q.dn_big) hipLaunchKernelGGL((k_mw_dense_s<KK, DD>), dim3(q.ndn, (pairs + MW_NT / ds_lanes - 1) / (MW_NT / ds_lanes)), dim3(MW_NT), 0, c->stream, q, ds_lanes);
        }
        // (the per-term pairings A_Y are written by k_mw_saccum, or by k_mw_saccum_one cluster by cluster when no cluster is left to k_mw_saccum)
        if (c->n_many_term) hipLaunchKernelGGL((k_mw_saccum<KK, DD>), dim3((c->maxP * (c->maxP + 1) / 2 * c->sa_lanes + MW_NT - 1) / MW_NT, q.J), dim3(MW_NT), 0, c->stream, q, c->sa_lanes);
        if (c->n_one_term) hipLaunchKernelGGL((k_mw_saccum_one<KK, DD>), dim3((c->maxP * (c->maxP + 1) / 2 + MW_NT - 1) / MW_NT, q.J), dim3(MW_NT), 0, c->stream, q, c->n_many_term ? 0 : 1);
    });
    MWCHECK(hipGetLastError());
    if (c->timing) MWCHECK(hipEventRecord(c->ev[1], c->stream));
    c->assembled = true;
    c->factored = false;
    c->local_factored = false;
    return 0;
}

// ---- cluster sharding: split-phase entry points and the RCCL exchange ---------------------------------------------------

extern "C" int clrs_comm_unique_id(void *id128) {
    if (!id128) return mw_fail(CLRS_ERR_INVALID, "null argument");
    int rc = mw_nccl_load();
    if (rc) return rc;
    NCCLCHECK(g_nccl.GetUniqueId((mw_nccl_id *)id128));
    return 0;
}

// This context holds the clusters of rank `rank` of `world` (the description it was created from lists only those clusters, with
// all N free variables).  world = 1 restores the unsharded behaviour.
extern "C" int clrs_mw_set_shard(clrs_mw_ctx *c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return mw_fail(CLRS_ERR_INVALID, "bad rank / world");
    MWCHECK(hipSetDevice(c->device));
    MWCHECK(hipStreamSynchronize(c->stream));
    MwDev &q = c->d;
    if (world != q.world) {
        int rc;
        if ((rc = mw_dmalloc(c, &q.Qg, (i64)world * q.N * q.N * c->K))) return rc;       // (the previous buffers stay in `allocs` until destroy)
        if ((rc = mw_dmalloc(c, &q.ug, (i64)world * q.N * c->K))) return rc;
    }
    q.rank = rank; q.world = world;
    q.gathered = (world > 1 || c->comm || c->lgroup) ? 1 : 0;
    return 0;
}
// The library performs the two exchanges itself with RCCL all-gathers on the context stream (one of limbs*N*N doubles per
// factorisation, one of limbs*N per solve): clrs_mw_schur_factor_dev / _solve_dev then are collective calls.
extern "C" int clrs_mw_comm_init(clrs_mw_ctx *c, const void *id128, int rank, int world) {
    if (!c || !id128) return mw_fail(CLRS_ERR_INVALID, "null argument");
    int rc = mw_nccl_load();
    if (rc) return rc;
    if ((rc = clrs_mw_set_shard(c, rank, world))) return rc;
    mw_nccl_id id;
    std::memcpy(&id, id128, sizeof(id));
    NCCLCHECK(g_nccl.CommInitRank(&c->comm, world, id, rank));
    c->d.gathered = 1;
    return 0;
}
extern "C" int clrs_mw_comm_destroy(clrs_mw_ctx *c) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        NCCLCHECK(g_nccl.CommDestroy(c->comm));
        c->comm = nullptr;
        if (c->comm_side) { NCCLCHECK(g_nccl.CommDestroy(c->comm_side)); c->comm_side = nullptr; }
    }
    if (c->lgroup) {                                     // detach from the in-process group (which counts its attached ranks)
        (void)hipStreamSynchronize(c->stream);
        std::lock_guard<std::mutex> lk(c->lgroup->m);
        c->lgroup->attached--;
        c->lgroup = nullptr;
    }
    c->d.gathered = c->d.world > 1 ? 1 : 0;
    return 0;
}
extern "C" double *clrs_mw_q_gather_dev(clrs_mw_ctx *c) { return c ? c->d.Qg : nullptr; }
extern "C" double *clrs_mw_u_gather_dev(clrs_mw_ctx *c) { return c ? c->d.ug : nullptr; }

// all-gather of `cnt` doubles per rank inside `base` ([world][cnt], this rank's slot written by earlier work on `stream`); channel 0 =
// the context's stream, 1 = the side stream of the interior-point iteration
static bool mw_has_comm(const clrs_mw_ctx *c, int chan) { return c->lgroup || (chan == 0 ? c->comm : c->comm_side); }
static int mw_allgather(clrs_mw_ctx *c, int chan, double *base, size_t cnt, hipStream_t stream) {
    const MwDev &q = c->d;
    if (c->lgroup) {
        clrs_mw_local_group *g = c->lgroup;
        auto &ch = g->ch[chan];
        const int r = q.rank;
        MWCHECK(hipEventRecord(ch.ready[r], stream));
        ch.base[r] = base;
        g->barrier(chan);
        for (int p = 0; p < g->world; p++) {
            if (p == r) continue;
            MWCHECK(hipStreamWaitEvent(stream, ch.ready[p], 0));
            MWCHECK(hipMemcpyAsync(base + (size_t)p * cnt, ch.base[p] + (size_t)p * cnt, cnt * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
        MWCHECK(hipEventRecord(ch.copied[r], stream));
        g->barrier(chan);
        for (int p = 0; p < g->world; p++)
            if (p != r) MWCHECK(hipStreamWaitEvent(stream, ch.copied[p], 0));
        return 0;
    }
    void *comm = chan == 0 ? c->comm : c->comm_side;
    if (!comm) return mw_fail(CLRS_ERR_STATE, chan == 0 ? "sharded context without a communicator: use the split-phase entry points or clrs_mw_comm_init"
                                                        : "the sharded interior-point iteration needs the side communicator: clrs_mw_comm_init_side");
    NCCLCHECK(g_nccl.AllGather(base + (size_t)q.rank * cnt, base, cnt, MW_NCCL_FLOAT64, comm, stream));
    return 0;
}
// Diagnostic for the multi-GPU bench: the cost of the exchanges of one sharded interior-point iteration on this context's communicator, measured with
// HIP events on the context's stream around `reps` all-gathers of each size, back to back: us[0] the partial Q (limbs N^2 doubles per rank), us[1] the
// partial u (limbs N), us[2] a scalar record (MWG_LEN).  Collective: every rank of the communicator must call it.  info[0..2] = rank, world, backend
// (0 none, 1 RCCL, 2 in-process group).
extern "C" int clrs_mw_comm_probe(clrs_mw_ctx *c, int reps, double us[3], int info[3]) {
    if (!c || !us || !info || reps < 1) return mw_fail(CLRS_ERR_INVALID, "bad argument");
    MWCHECK(hipSetDevice(c->device));
    const MwDev &q = c->d;
    info[0] = q.rank; info[1] = q.world; info[2] = c->comm ? 1 : c->lgroup ? 2 : 0;
    us[0] = us[1] = us[2] = 0.0;
    if (!mw_has_comm(c, 0) || q.N <= 0) return 0;
    const size_t cnt[3] = {(size_t)q.N * q.N * c->K, (size_t)q.N * c->K, (size_t)MWG_LEN(c->K, q.N)};      // (the record as the iteration exchanges it: clrs_mw_ipm.hip.h)
    double *scratch = nullptr;                                       // (a buffer of its own: the exchanges must not disturb Qg / ug)
    MWCHECK(hipMalloc((void **)&scratch, cnt[0] * q.world * sizeof(double)));
    MWCHECK(hipMemsetAsync(scratch, 0, cnt[0] * q.world * sizeof(double), c->stream));
    hipEvent_t e0, e1;
    MWCHECK(hipEventCreate(&e0)); MWCHECK(hipEventCreate(&e1));
    int rc = 0;
    for (int k = 0; k < 3 && !rc; k++) {
        for (int w = 0; w < 3 && !rc; w++) rc = mw_allgather(c, 0, scratch, cnt[k], c->stream);          // warm-up
        if (rc) break;
        (void)hipEventRecord(e0, c->stream);
        for (int i = 0; i < reps && !rc; i++) rc = mw_allgather(c, 0, scratch, cnt[k], c->stream);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess) rc = mw_fail(CLRS_ERR_HIP, "hipEventSynchronize failed");
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        us[k] = 1e3 * ms / reps;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(scratch);
    return rc;
}
// second RCCL communicator (its own unique id), for the exchanges the interior-point iteration issues on its side stream
extern "C" int clrs_mw_comm_init_side(clrs_mw_ctx *c, const void *id128) {
    if (!c || !id128) return mw_fail(CLRS_ERR_INVALID, "null argument");
    if (!c->comm) return mw_fail(CLRS_ERR_STATE, "clrs_mw_comm_init first");
    mw_nccl_id id;
    std::memcpy(&id, id128, sizeof(id));
    NCCLCHECK(g_nccl.CommInitRank(&c->comm_side, c->d.world, id, c->d.rank));
    return 0;
}
extern "C" int clrs_mw_local_group_create(int world, int device, clrs_mw_local_group **out) {
    if (!out || world < 1) return mw_fail(CLRS_ERR_INVALID, "bad argument");
    MWCHECK(hipSetDevice(device));
    clrs_mw_local_group *g = new clrs_mw_local_group();
    g->world = world;
    for (auto &ch : g->ch) {
        ch.base.assign(world, nullptr);
        ch.ready.assign(world, nullptr);
        ch.copied.assign(world, nullptr);
        for (int r = 0; r < world; r++) {
            MWCHECK(hipEventCreateWithFlags(&ch.ready[r], hipEventDisableTiming));
            MWCHECK(hipEventCreateWithFlags(&ch.copied[r], hipEventDisableTiming));
            // recorded once on the null stream, so that a wait issued before the first real record of a peer is well defined
            MWCHECK(hipEventRecord(ch.ready[r], nullptr));
            MWCHECK(hipEventRecord(ch.copied[r], nullptr));
        }
    }
    MWCHECK(hipDeviceSynchronize());
    *out = g;
    return 0;
}
// (refused -- the group is leaked and the last error set -- while contexts are attached: they would dereference freed events in their next exchange;
// clrs_mw_comm_destroy or clrs_mw_destroy on each rank first)
extern "C" void clrs_mw_local_group_destroy(clrs_mw_local_group *g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->attached > 0) { (void)mw_fail(CLRS_ERR_STATE, "clrs_mw_local_group_destroy: contexts are still attached (clrs_mw_comm_destroy them first)"); return; }
    }
    for (auto &ch : g->ch) {
        for (auto e : ch.ready) if (e) (void)hipEventDestroy(e);
        for (auto e : ch.copied) if (e) (void)hipEventDestroy(e);
    }
    delete g;
}
// this context is rank `rank` of the in-process group: the library's exchanges then go through it (one host thread per rank)
extern "C" int clrs_mw_comm_init_local(clrs_mw_ctx *c, clrs_mw_local_group *g, int rank) {
    if (!c || !g) return mw_fail(CLRS_ERR_INVALID, "null argument");
    int rc = clrs_mw_set_shard(c, rank, g->world);
    if (rc) return rc;
    if (c->lgroup != g) {
        if (c->lgroup) { std::lock_guard<std::mutex> lk(c->lgroup->m); c->lgroup->attached--; }
        std::lock_guard<std::mutex> lk(g->m);
        g->attached++;
    }
    c->lgroup = g;
    c->d.gathered = 1;
    return 0;
}

// blocked Cholesky and inverse factor of matrices in global memory over many workgroups (clrs_mw_kernels.hip.h, k_mw_bp_*): all matrices of the
// list side by side in every launch (their block columns are chains of five launches each: the clusters of Nsphere_packing with N = 3 wait
// for nothing but their own).  ride_factor: the first launch also carries the clusters that fit in LDS (k_mw_factor's work).
static int mw_potrf_blocked(clrs_mw_ctx *c, const std::vector<MwBp> &hm, const MwBp *d_ms, bool ride_factor) {
    const MwDev &q = c->d;
    const int MW_PB = MW_PB_OF(c->K), nm = (int)hm.size();
    int nmax = 0;
    for (auto &m : hm) nmax = std::max(nmax, m.n);
    const int nbk = (nmax + MW_PB - 1) / MW_PB;
    MW_DISPATCH(c, {
        for (int j0 = 0; j0 < nmax; j0 += MW_PB) {
            const int nb = std::min(MW_PB, nmax - j0), mm = nmax - j0 - nb;
            const bool ride = ride_factor && j0 == 0;
            if (c->pipe_bp) {
                // block row j0 / MW_PB - 1 of the inverse factors rides on this launch (its diagonal block was the previous launch's)
                const int inv_row = j0 / MW_PB - 1 >= 1 ? j0 / MW_PB - 1 : 0;
                c->pipe_epoch = (c->pipe_epoch + 1) & 0x3ffffff;
                hipLaunchKernelGGL(k_mw_bp_diag_pipe<KK>, dim3(mwp_blocks(nm) + (ride ? q.J * MW_INV_WG : 0) + inv_row * (MW_PB / MW_BP_IC) * nm), dim3(MWP_NT),
                                   ride ? std::max<size_t>(MWP_LDS_ALONE, c->sm_factor) : std::max<size_t>(MWP_LDS_ALONE, c->sm_bp_inv),
                                   c->stream, q, d_ms, nm, j0, c->pipe_epoch, ride ? q.J : 0, inv_row);
            } else
            hipLaunchKernelGGL(k_mw_bp_diag<KK>, dim3(MW_INV_WG, nm + (ride ? q.J : 0)), dim3(MW_PT), ride ? std::max(c->sm_bp_diag, c->sm_factor) : c->sm_bp_diag, c->stream, q, d_ms, nm, j0);
            if (mm > 0) {
                hipLaunchKernelGGL(k_mw_bp_panel<KK>, dim3((mm + MW_BP_PR - 1) / MW_BP_PR, nm), dim3(MW_PT), c->sm_bp_panel, c->stream, q, d_ms, j0);
                hipLaunchKernelGGL(k_mw_bp_syrk<KK>, dim3((unsigned)(((i64)mm * (mm + 1) / 2 * MW_BP_SW + MW_NT - 1) / MW_NT), nm), dim3(MW_NT), 0, c->stream, q, d_ms, j0);
            }
        }
        if (c->pipe_bp) {                // the rows 1 .. nbk - 2 rode on the launches of the diagonal blocks; the last row has none to ride on
            if (nbk >= 2) hipLaunchKernelGGL(k_mw_bp_inv_row<KK>, dim3(nbk - 1, MW_PB / MW_BP_IC, nm), dim3(MW_PT), c->sm_bp_inv, c->stream, q, d_ms, nbk - 1);
        } else
        for (int d = 1; d < nbk; d++)
            hipLaunchKernelGGL(k_mw_bp_inv<KK>, dim3(nbk - d, MW_PB / MW_BP_IC, nm), dim3(MW_PT), c->sm_bp_inv, c->stream, q, d_ms, d);
        hipLaunchKernelGGL(k_mw_bp_finish<KK>, dim3((unsigned)std::min<i64>(1024, ((i64)nmax * nmax + MW_NT - 1) / MW_NT), nm), dim3(MW_NT), 0, c->stream, q, d_ms);
    });
    MWCHECK(hipGetLastError());
    return 0;
}

// L_j, LinvB_j of this rank's clusters and its partial Q into slot `rank` of the gather buffer
extern "C" int clrs_mw_schur_factor_local_dev(clrs_mw_ctx *c) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (!c->assembled) return mw_fail(CLRS_ERR_STATE, "clrs_mw_schur_factor before clrs_mw_schur_assemble");
    MWCHECK(hipSetDevice(c->device));
    const MwDev &q = c->d;
    int rc;
    if ((rc = mw_reset_info(c, 0))) return rc;
    if (c->timing) MWCHECK(hipEventRecord(c->ev[2], c->stream));
    // clusters that do not fit in LDS: blocked over many workgroups, all of them side by side; the others ride on the first of those launches
    // when they take the same number of workgroups per matrix, else they have their launch (k_mw_factor)
    const bool ride = !c->bp_S.empty() && c->nw_factor == MW_INV_WG && c->any_lds_cluster;
    if (c->pipe_S64) {
        c->pipe_epoch = (c->pipe_epoch + 1) & 0x3ffffff;
        MwDev q64 = q;
        q64.pipe_pc = c->pipe_pc64;
        MW_DISPATCH(c, { if constexpr (KK <= 6) { hipLaunchKernelGGL(k_mw_factor_pipe64<KK>, dim3(mwp_blocks64(q.J)), dim3(MWP_NT64), MWP_LDS_ALONE64, c->stream, q64, c->pipe_epoch); } });
    } else if (c->pipe_S) {
        c->pipe_epoch = (c->pipe_epoch + 1) & 0x3ffffff;
        MW_DISPATCH(c, { hipLaunchKernelGGL(k_mw_factor_pipe<KK>, dim3(mwp_blocks(q.J)), dim3(MWP_NT), MWP_LDS_ALONE, c->stream, q, c->pipe_epoch); });
    } else if (!ride && c->any_lds_cluster) MW_DISPATCH(c, { hipLaunchKernelGGL(k_mw_factor<KK>, dim3(q.J, c->nw_factor), dim3(MW_PT), c->sm_factor, c->stream, q); });
    if (!c->bp_S.empty()) MW_DISPATCH(c, hipLaunchKernelGGL(k_mw_keep_S<KK>, dim3((unsigned)std::min<i64>(64, ((i64)c->maxP * c->maxP + MW_NT - 1) / MW_NT), q.J), dim3(MW_NT), 0, c->stream, q));
    if (!c->bp_S.empty() && (rc = mw_potrf_blocked(c, c->bp_S, c->d_bp, ride))) return rc;
    MW_DISPATCH(c, {
        if (q.N > 0)
            hipLaunchKernelGGL((k_mw_linvb<KK, DD>), dim3((c->maxP * q.N + MW_NT / MW_LBI_W - 1) / (MW_NT / MW_LBI_W), q.J), dim3(MW_NT), 0, c->stream, q);
        if (c->timing) (void)hipEventRecord(c->ev[3], c->stream);
        const int qw = (i64)q.N * (q.N + 1) / 2 * 32 <= 32768 && q.xlen >= 32 ? 32 : MW_Q_W;      // lanes per entry of Q
        if (q.N > 0) hipLaunchKernelGGL(k_mw_qgram<KK>, dim3((q.N * (q.N + 1) / 2 + MW_NT / qw - 1) / (MW_NT / qw)), dim3(MW_NT), 0, c->stream, q, qw);
        if (c->timing) (void)hipEventRecord(c->ev[4], c->stream);
    });
    MWCHECK(hipGetLastError());
    c->local_factored = true;
    c->factored = false;
    c->assembled = false;                       // S now holds L_j: a second factorisation needs a new assembly (as the fp64 entry point)
    return 0;
}
// after the gather buffer holds every rank's partial Q: Q = their sum, Cholesky of Q (redundantly on every rank)
extern "C" int clrs_mw_schur_factor_finish_dev(clrs_mw_ctx *c) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (!c->local_factored) return mw_fail(CLRS_ERR_STATE, "clrs_mw_schur_factor_finish before clrs_mw_schur_factor_local");
    MWCHECK(hipSetDevice(c->device));
    const MwDev &q = c->d;
    if (q.N > 0 && c->pipe_Q) {
        const bool ride = c->ride_fwd != nullptr && !c->wide_solve;
        c->pipe_epoch = (c->pipe_epoch + 1) & 0x3ffffff;
        MW_DISPATCH(c, { hipLaunchKernelGGL(k_mw_potrf_q_pipe<KK>, dim3(64 + (ride ? q.J * (q.ride2_rhs ? 2 : 1) : 0)), dim3(MWP_NT), std::max<size_t>(MWP_LDS_ALONE, ride ? c->sm_fwd : 0),
                                            c->stream, q, c->pipe_epoch, c->ride_fwd, ride ? c->ride_wait : (const int *)nullptr, c->ride_wait_value); });
        c->fwd_rode = ride;
    } else if (q.N > 0 && c->lds_q) {
        // the interior-point iteration hands over the right-hand side of its next solve: the solve's first product pair rides on this launch
        const bool ride = c->ride_fwd != nullptr && !c->wide_solve;
        MW_DISPATCH(c, { hipLaunchKernelGGL(k_mw_potrf_q<KK>, dim3(MW_INV_WG + (ride ? q.J * (q.ride2_rhs ? 2 : 1) : 0)), dim3(MW_PT), ride ? std::max(c->sm_q, c->sm_fwd) : c->sm_q, c->stream, q, MW_INV_WG, c->ride_fwd, ride ? c->ride_wait : (const int *)nullptr, c->ride_wait_value); });
        c->fwd_rode = ride;
    } else if (q.N > 0) {
        MW_DISPATCH(c, { hipLaunchKernelGGL(k_mw_qsum<KK>, dim3((unsigned)std::min<i64>(256, ((i64)q.N * q.N + MW_NT - 1) / MW_NT)), dim3(MW_NT), 0, c->stream, q); });
        int rc = mw_potrf_blocked(c, c->bp_Q, c->d_bp + c->bp_S.size(), false);
        if (rc) return rc;
    }
    MWCHECK(hipGetLastError());
    if (c->timing) MWCHECK(hipEventRecord(c->ev[5], c->stream));
    c->factored = true;
    return 0;
}
extern "C" int clrs_mw_schur_factor_dev(clrs_mw_ctx *c) {
    int rc = clrs_mw_schur_factor_local_dev(c);
    if (rc) return rc;
    const MwDev &q = c->d;
    if (q.gathered && q.N > 0) {
        int rc2 = mw_allgather(c, 0, q.Qg, (size_t)q.N * q.N * c->K, c->stream);
        if (rc2) return rc2;
    }
    return clrs_mw_schur_factor_finish_dev(c);
}
extern "C" int clrs_mw_sync_status(clrs_mw_ctx *c) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    int st = 0, rc;
    if ((rc = mw_read_info(c, 0, &st))) return rc;
    return st;
}

// t_j = L_j^-1 rhs_x[j] and this rank's partial u (slot `rank` of the u gather buffer when sharded)
extern "C" int clrs_mw_schur_solve_fwd_dev(clrs_mw_ctx *c, const double *d_rhs_x) {
    if (!c || !d_rhs_x) return mw_fail(CLRS_ERR_INVALID, "null argument");
    if (!c->local_factored) return mw_fail(CLRS_ERR_STATE, "clrs_mw_schur_solve before clrs_mw_schur_factor");
    MWCHECK(hipSetDevice(c->device));
    const MwDev &q = c->d;
    if (c->timing) MWCHECK(hipEventRecord(c->ev[6], c->stream));
    MW_DISPATCH(c, {
        hipLaunchKernelGGL(k_mw_solve_fwd<KK>, dim3(q.J), dim3(MW_NT), c->sm_fwd, c->stream, q, d_rhs_x);
        if (q.gathered && q.N > 0) hipLaunchKernelGGL(k_mw_usum<KK>, dim3(1), dim3(MW_NT), 0, c->stream, q);
    });
    MWCHECK(hipGetLastError());
    c->fwd_done = true;
    c->split_rhs_x = d_rhs_x;
    return 0;
}
// the backward half: dy = Q^-1 (rhs_y - sum u), dx_j = L_j^-T (t_j + LinvB_j dy).  mode 0: plain; 1: followed by the first half of the refinement step
// (residuals, t', u' into q.t, q.ub); 2: the correction's backward half, added to dx, dy (k_mw_solve_bwd).  full_kc: the correction in all K limbs.
#define MW_BWD(KCC, MODE) hipLaunchKernelGGL((k_mw_solve_bwd<KK, KCC, DD, MODE>), dim3(q.J), dim3(MW_NT), lds, c->stream, q, (const double *)dy_mid, d_dx, mid, d_dy, d_rhs_x)
static int mw_solve_bwd(clrs_mw_ctx *c, const MwDev &q, const double *d_rhs_x, const double *d_rhs_y, double *d_dx, double *d_dy, int mode, bool full_kc) {
    if (c->sm_mid + c->sm_bwd > MW_LDS_MAX) return mw_fail(CLRS_ERR_INVALID, "system too large for the one-workgroup-per-cluster solve kernels (a sharded solve with clusters + free variables beyond LDS)");
    const size_t lds = c->sm_mid + c->sm_bwd;
    MW_DISPATCH(c, {
        constexpr int KC = mw_kc(KK);
        // few clusters, one rank: dy is formed by every workgroup of the backward launch itself (one launch less on the chain of the iteration)
        const bool mid_in_bwd = q.N > 0 && !q.gathered && q.J <= 4;
        double *dy_mid = mode == 2 ? q.dy2 : d_dy;          // (the correction dy' has a buffer of its own; the backward launch adds it)
        if (q.N > 0 && !mid_in_bwd) {
            if (mode == 2 && !full_kc) hipLaunchKernelGGL((k_mw_solve_mid<KK, KC>), dim3(1), dim3(MW_NT), c->sm_mid, c->stream, q, d_rhs_y, dy_mid);
            else hipLaunchKernelGGL((k_mw_solve_mid<KK, KK>), dim3(1), dim3(MW_NT), c->sm_mid, c->stream, q, d_rhs_y, dy_mid);
        }
        const double *mid = mid_in_bwd ? d_rhs_y : (const double *)nullptr;
        if (mode == 0) MW_BWD(KK, 0);
        else if (mode == 1) { if (full_kc) MW_BWD(KK, 1); else MW_BWD(KC, 1); }
        else { if (full_kc) MW_BWD(KK, 2); else MW_BWD(KC, 2); }
    });
#undef MW_BWD
    MWCHECK(hipGetLastError());
    return 0;
}
extern "C" int clrs_mw_schur_solve_bwd_dev(clrs_mw_ctx *c, const double *d_rhs_y, double *d_dx, double *d_dy) {
    if (!c || !d_dx) return mw_fail(CLRS_ERR_INVALID, "null argument");
    if (!c->factored || !c->fwd_done) return mw_fail(CLRS_ERR_STATE, "clrs_mw_schur_solve_bwd before clrs_mw_schur_factor_finish / clrs_mw_schur_solve_fwd");
    const MwDev &q = c->d;
    if (q.N > 0 && (!d_rhs_y || !d_dy)) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    // split-phase callers with the refinement on: the launch also forms the residuals and the forward half of the correction, and leaves this rank's
    // partial u' in its gather slot -- after one more exchange of the u slots clrs_mw_schur_solve_refine_dev adds the correction (optional: without
    // that call (dx, dy) are the plain products' solution)
    const bool first_half = c->refine && c->split_rhs_x != nullptr;
    int rc = mw_solve_bwd(c, q, c->split_rhs_x, d_rhs_y, d_dx, d_dy, first_half ? 1 : 0, c->refine != 2);
    if (rc) return rc;
    if (first_half && q.gathered && q.N > 0) {
        MwDev q2 = q;
        q2.u = q.ub;
        MW_DISPATCH(c, hipLaunchKernelGGL(k_mw_usum<KK>, dim3(1), dim3(MW_NT), 0, c->stream, q2));
        MWCHECK(hipGetLastError());
    }
    c->refine_ready = first_half;
    c->split_rhs_x = nullptr;
    if (c->timing) MWCHECK(hipEventRecord(c->ev[7], c->stream));
    c->fwd_done = false;
    return 0;
}
// second half of the refinement step for split-phase callers: after clrs_mw_schur_solve_bwd_dev and one more exchange of the u gather slots
extern "C" int clrs_mw_schur_solve_refine_dev(clrs_mw_ctx *c, const double *d_rhs_y, double *d_dx, double *d_dy) {
    if (!c || !d_dx) return mw_fail(CLRS_ERR_INVALID, "null argument");
    if (!c->refine) return 0;                             // clrs_config_set("mw_refine", 0): nothing to add
    if (!c->refine_ready) return mw_fail(CLRS_ERR_STATE, "clrs_mw_schur_solve_refine before clrs_mw_schur_solve_fwd / clrs_mw_schur_solve_bwd");
    const MwDev &q = c->d;
    if (q.N > 0 && (!d_rhs_y || !d_dy)) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    MwDev q2 = q;
    q2.u = q.ub;
    c->refine_ready = false;
    return mw_solve_bwd(c, q2, nullptr, d_rhs_y, d_dx, d_dy, 2, c->refine != 2);
}

// one pass of the row-parallel solve (clusters or a Q beyond 64 rows, one rank): a launch per product
static int mw_solve_wide_once(clrs_mw_ctx *c, const double *d_rhs_x, const double *d_rhs_y, double *d_dx, double *d_dy) {
    const MwDev &q = c->d;
    constexpr int RPW = MW_NT / MW_SW_L;
    const dim3 gP((c->maxP + RPW - 1) / RPW, q.J), gN((q.N + RPW - 1) / RPW), gX((unsigned)((q.xlen + RPW - 1) / RPW));
    MW_DISPATCH(c, {
        hipLaunchKernelGGL(k_mw_solve_wide<KK>, gP, dim3(MW_NT), 0, c->stream, q, 1, d_rhs_x, d_rhs_y, d_dx, d_dy, c->vz);
        if (q.N > 0) {
            hipLaunchKernelGGL(k_mw_solve_wide<KK>, gN, dim3(MW_NT), 0, c->stream, q, 2, d_rhs_x, d_rhs_y, d_dx, d_dy, c->vz);
            hipLaunchKernelGGL(k_mw_solve_wide<KK>, gN, dim3(MW_NT), 0, c->stream, q, 3, d_rhs_x, d_rhs_y, d_dx, d_dy, c->vz);
            hipLaunchKernelGGL(k_mw_solve_wide<KK>, gN, dim3(MW_NT), 0, c->stream, q, 4, d_rhs_x, d_rhs_y, d_dx, d_dy, c->vz);
            hipLaunchKernelGGL(k_mw_solve_wide<KK>, gX, dim3(MW_NT), 0, c->stream, q, 5, d_rhs_x, d_rhs_y, d_dx, d_dy, c->vz);
        }
        hipLaunchKernelGGL(k_mw_solve_wide<KK>, gP, dim3(MW_NT), 0, c->stream, q, 6, d_rhs_x, d_rhs_y, d_dx, d_dy, c->vz);
    });
    MWCHECK(hipGetLastError());
    return 0;
}
// The solve stage (src/solver.jl:1527-1582): one pass of products with the inverse factors, then (c->refine, the default) one step of iterative
// refinement against the assembled S_j and B (k_mw_refine's header): the backward error of the reference's substitutions at the latency of products.
extern "C" int clrs_mw_schur_solve_dev(clrs_mw_ctx *c, const double *d_rhs_x, const double *d_rhs_y, double *d_dx, double *d_dy) {
    if (!c || !d_rhs_x || !d_dx) return mw_fail(CLRS_ERR_INVALID, "null argument");
    if (!c->factored) return mw_fail(CLRS_ERR_STATE, "clrs_mw_schur_solve before clrs_mw_schur_factor");
    MwDev &q = c->d;
    if (q.N > 0 && (!d_rhs_y || !d_dy)) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    int rc = 0;
    const int refine = c->refine_skip_next ? 0 : c->refine;
    c->refine_skip_next = false;
    if (c->wide_solve && !q.gathered) {                  // large clusters or a large Q: one launch per product, rows over many workgroups
        if (c->timing) MWCHECK(hipEventRecord(c->ev[6], c->stream));
        if ((rc = mw_solve_wide_once(c, d_rhs_x, d_rhs_y, d_dx, d_dy))) return rc;
        if (refine) {
            constexpr int RPW = MW_NT / MW_SW_L;
            const int rowsP = (c->maxP + RPW - 1) / RPW, rowsN = (q.N + RPW - 1) / RPW;
            MW_DISPATCH(c, hipLaunchKernelGGL((k_mw_refine<KK, DD>), dim3(std::max(rowsP, rowsN), q.J + (q.N > 0 ? 1 : 0)), dim3(MW_NT), 0, c->stream, q, 1, d_rhs_x, d_dx, d_dy));
            MWCHECK(hipGetLastError());
            q.uadd = q.N > 0 ? q.u2 : nullptr;
            rc = mw_solve_wide_once(c, q.rx2, d_rhs_y, q.dx2, q.dy2);
            q.uadd = nullptr;
            if (rc) return rc;
            MW_DISPATCH(c, hipLaunchKernelGGL((k_mw_refine<KK, DD>), dim3((unsigned)std::min<i64>(256, (q.xlen + q.N + MW_NT - 1) / MW_NT)), dim3(MW_NT), 0, c->stream, q, 3, d_rhs_x, d_dx, d_dy));
            MWCHECK(hipGetLastError());
        }
        if (c->timing) MWCHECK(hipEventRecord(c->ev[7], c->stream));
        return 0;
    }
    // one workgroup per cluster: forward half (unless it rode on an earlier launch), [exchange of the partial u], backward half; the refinement's
    // residuals and forward half ride on the backward launch, its backward half is one more launch [and one more exchange]
    if (c->fwd_rode) {                                   // t_j, u_j of this right-hand side are there already (k_mw_potrf_q's launch, or k_mwi_rows_fwd)
        c->fwd_rode = false;
        c->fwd_done = true;
        if (q.gathered && q.N > 0) {                     // sharded: this rank's partial u into its gather slot, as clrs_mw_schur_solve_fwd_dev does behind its launch
            MW_DISPATCH(c, hipLaunchKernelGGL(k_mw_usum<KK>, dim3(1), dim3(MW_NT), 0, c->stream, q));
            MWCHECK(hipGetLastError());
        }
    } else rc = clrs_mw_schur_solve_fwd_dev(c, d_rhs_x);
    if (rc) return rc;
    if (q.gathered && q.N > 0 && (rc = mw_allgather(c, 0, q.ug, (size_t)q.N * c->K, c->stream))) return rc;
    c->split_rhs_x = nullptr;
    if (!refine) { const int keep = c->refine; c->refine = 0; rc = clrs_mw_schur_solve_bwd_dev(c, d_rhs_y, d_dx, d_dy); c->refine = keep; return rc; }
    const bool full_kc = refine != 2;                  // 2: the correction in mw_kc(K) limbs (opt-in: valid while twice the lost bits fit in them)
    if ((rc = mw_solve_bwd(c, q, d_rhs_x, d_rhs_y, d_dx, d_dy, 1, full_kc))) return rc;
    MwDev q2 = q;
    q2.u = q.ub;                                         // (the first half of the step wrote its u' beside the u the other workgroups were still reading)
    if (q.gathered && q.N > 0) {
        MW_DISPATCH(c, hipLaunchKernelGGL(k_mw_usum<KK>, dim3(1), dim3(MW_NT), 0, c->stream, q2));
        MWCHECK(hipGetLastError());
        if ((rc = mw_allgather(c, 0, q.ug, (size_t)q.N * c->K, c->stream))) return rc;
    }
    if ((rc = mw_solve_bwd(c, q2, d_rhs_x, d_rhs_y, d_dx, d_dy, 2, full_kc))) return rc;
    if (c->timing) MWCHECK(hipEventRecord(c->ev[7], c->stream));
    c->fwd_done = false;
    return 0;
}

extern "C" double *clrs_mw_S_buffer_dev(clrs_mw_ctx *c) { return c ? c->d.S : nullptr; }
extern "C" double *clrs_mw_AY_buffer_dev(clrs_mw_ctx *c) { return c ? c->d.AY : nullptr; }

// S_j (S layout) and the per-term pairings A_Y of the last assembly, planar limbs, to host memory (either pointer may be NULL): for
// callers of the device-pointer entry points (clrs_mw_schur_assemble_dev) that want to look at what was assembled
extern "C" int clrs_mw_get_S(clrs_mw_ctx *c, double *S_out, double *AY_out) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (!c->assembled) return mw_fail(CLRS_ERR_STATE, "clrs_mw_get_S before clrs_mw_schur_assemble (or after the factorisation has overwritten S)");
    MWCHECK(hipSetDevice(c->device));
    MWCHECK(hipStreamSynchronize(c->stream));
    if (S_out) MWCHECK(hipMemcpy(S_out, c->d.S, (size_t)c->d.Slen * c->K * sizeof(double), hipMemcpyDeviceToHost));
    if (AY_out && c->d.T) MWCHECK(hipMemcpy(AY_out, c->d.AY, (size_t)c->d.T * c->K * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// diagnostic (-DCLRS_MW_STAMPS builds): step stamps of the pipelined factorisations, [16 roles][40]: rows 0-7 the workgroups of cluster 0 in
// k_mw_factor_pipe, rows 8-15 those of Q in k_mw_potrf_q_pipe; columns 0..n-1 the top of step k, 39 start, 38 end (100 MHz wall clock); behind them
// [8 roles][32 steps][4 who][4 points]: the stamps inside the steps of cluster 0 (clrs_mw_pipe.hip.h; scripts/pipe_substamps.py)
extern "C" int clrs_mw_debug_pipe_stamps(clrs_mw_ctx *c, unsigned long long *out) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
#ifndef CLRS_MW_STAMPS
    return mw_fail(CLRS_ERR_STATE, "this library was built without -DCLRS_MW_STAMPS");
#endif
    MWCHECK(hipSetDevice(c->device));
    MWCHECK(hipStreamSynchronize(c->stream));
    if (!c->d.pipe_stamps) {
        double *p = nullptr;
        int rc = mw_dmalloc(c, &p, 16 * 40 + 8 * 32 * 4 * 4);
        if (rc) return rc;
        c->d.pipe_stamps = (unsigned long long *)p;
    } else if (out) MWCHECK(hipMemcpy(out, c->d.pipe_stamps, (16 * 40 + 8 * 32 * 4 * 4) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
// diagnostic: phase stamps (100 MHz wall clock) of wave 0 of the first workgroup of the next k_mws_pair launches; out[16] = the last ones
extern "C" int clrs_mw_debug_exact_stamps(clrs_mw_ctx *c, unsigned long long *out) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
#ifndef CLRS_MW_STAMPS
    return mw_fail(CLRS_ERR_STATE, "this library was built without -DCLRS_MW_STAMPS (the product kernels carry no phase stamps): CLRS_MW_STAMPS=1 builds the diagnostic variant");
#endif
    MWCHECK(hipSetDevice(c->device));
    MWCHECK(hipStreamSynchronize(c->stream));
    if (!c->mws.stamps) {
        double *p = nullptr;
        int rc = mw_dmalloc(c, &p, 16);
        if (rc) return rc;
        c->mws.stamps = (unsigned long long *)p;
    } else if (out) MWCHECK(hipMemcpy(out, c->mws.stamps, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int clrs_mw_get_timings(clrs_mw_ctx *c, double t[6]) {
    if (!c || !t) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipStreamSynchronize(c->stream));
    if (c->timing) {
        float ms = 0;
        const int pairs[6][2] = {{0, 1}, {2, 3}, {3, 3}, {3, 4}, {4, 5}, {6, 7}};      // schur, cholS + LinvB (one kernel), -, Q, cholQ, solve
        for (int i = 0; i < 6; i++) {
            if (pairs[i][0] == pairs[i][1]) { c->times[i] = 0; continue; }
            if (hipEventElapsedTime(&ms, c->ev[pairs[i][0]], c->ev[pairs[i][1]]) == hipSuccess) c->times[i] = ms * 1e-3;
        }
    }
    for (int i = 0; i < 6; i++) t[i] = c->times[i];
    return 0;
}

// ---- host-pointer entry points (planar K x len host arrays; copy, run, synchronise) ----------------------------------
extern "C" int clrs_mw_cholesky_blocks(clrs_mw_ctx *c, const double *X, double *Xchol) {
    if (!c || !X || !Xchol) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->d.xylen * c->K * sizeof(double);
    MWCHECK(hipMemcpyAsync(c->d_Xin, X, bytes, hipMemcpyHostToDevice, c->stream));
    int rc = clrs_mw_cholesky_blocks_dev(c, c->d_Xin, c->d_Xc);
    if (rc) return rc;
    MWCHECK(hipMemcpyAsync(Xchol, c->d_Xc, bytes, hipMemcpyDeviceToHost, c->stream));
    return clrs_mw_sync_status_cholesky(c);
}
extern "C" int clrs_mw_schur_assemble(clrs_mw_ctx *c, const double *Xchol, const double *Y, double *S_out, double *AY_out) {
    if (!c || !Xchol || !Y) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->d.xylen * c->K * sizeof(double);
    MWCHECK(hipMemcpyAsync(c->d_Xc, Xchol, bytes, hipMemcpyHostToDevice, c->stream));
    MWCHECK(hipMemcpyAsync(c->d_Y, Y, bytes, hipMemcpyHostToDevice, c->stream));
    // the substitutions need the reciprocal diagonal of chol(X): recomputed from the factor the caller passes
    int rc;
    if ((rc = mw_launch_xrd(c, c->d_Xc))) return rc;
    if ((rc = clrs_mw_schur_assemble_dev(c, c->d_Xc, c->d_Y))) return rc;
    if (S_out) MWCHECK(hipMemcpyAsync(S_out, c->d.S, (size_t)c->d.Slen * c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (AY_out && c->d.T) MWCHECK(hipMemcpyAsync(AY_out, c->d.AY, (size_t)c->d.T * c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MWCHECK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int clrs_mw_schur_factor(clrs_mw_ctx *c) {
    int rc = clrs_mw_schur_factor_dev(c);
    if (rc) return rc;
    return clrs_mw_sync_status(c);
}
extern "C" int clrs_mw_get_factor(clrs_mw_ctx *c, double *L, double *LinvB, double *LQ) {
    if (!c) return mw_fail(CLRS_ERR_INVALID, "null context");
    if (!c->factored) return mw_fail(CLRS_ERR_STATE, "clrs_mw_get_factor before clrs_mw_schur_factor");
    MWCHECK(hipSetDevice(c->device));
    MWCHECK(hipStreamSynchronize(c->stream));
    const MwDev &q = c->d;
    const int K = c->K, N = q.N;
    if (L) MWCHECK(hipMemcpy(L, q.S, (size_t)q.Slen * K * sizeof(double), hipMemcpyDeviceToHost));
    if (LQ && N) MWCHECK(hipMemcpy(LQ, q.Q, (size_t)N * N * K * sizeof(double), hipMemcpyDeviceToHost));
    if (LinvB && N) {
        // device: stacked xlen x N; caller: per cluster P_j x N column-major, concatenated (as clrs_get_factor)
        std::vector<double> h((size_t)q.xlen * N * K);
        MWCHECK(hipMemcpy(h.data(), q.LB, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        const i64 plane = q.xlen * (i64)N;
        for (int l = 0; l < K; l++) {
            i64 off = 0;
            for (int j = 0; j < q.J; j++) {
                const int P = c->clu[j].P;
                for (int a = 0; a < N; a++)
                    for (int r = 0; r < P; r++) LinvB[l * plane + off + r + (i64)a * P] = h[l * plane + c->clu[j].coff + r + (i64)a * q.xlen];
                off += (i64)P * N;
            }
        }
    }
    return 0;
}
extern "C" int clrs_mw_schur_solve(clrs_mw_ctx *c, const double *rhs_x, const double *rhs_y, double *dx, double *dy) {
    if (!c || !rhs_x || !dx) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    const MwDev &q = c->d;
    const int K = c->K;
    MWCHECK(hipMemcpyAsync(c->d_rx, rhs_x, (size_t)q.xlen * K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (q.N) {
        if (!rhs_y || !dy) return mw_fail(CLRS_ERR_INVALID, "null argument");
        MWCHECK(hipMemcpyAsync(c->d_ry, rhs_y, (size_t)q.N * K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    int rc = clrs_mw_schur_solve_dev(c, c->d_rx, c->d_ry, c->d_dx, c->d_dy);
    if (rc) return rc;
    MWCHECK(hipMemcpyAsync(dx, c->d_dx, (size_t)q.xlen * K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (q.N) MWCHECK(hipMemcpyAsync(dy, c->d_dy, (size_t)q.N * K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MWCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

static int mw_launch_xrd(clrs_mw_ctx *c, const double *d_Xc) {
    if (c->d.NB == 0) return 0;
    MW_DISPATCH(c, hipLaunchKernelGGL(k_mw_xrd<KK>, dim3(c->d.NB), dim3(MW_NT), 0, c->stream, c->d, d_Xc));
    MWCHECK(hipGetLastError());
    c->xinv_valid = false;                  // the caller's factors: no inverse beside them, the assembly substitutes
    return 0;
}
// device-pointer form of the same (callers that bring their own Cholesky factors of X)
extern "C" int clrs_mw_set_xchol_dev(clrs_mw_ctx *c, const double *d_Xchol) {
    if (!c || !d_Xchol) return mw_fail(CLRS_ERR_INVALID, "null argument");
    MWCHECK(hipSetDevice(c->device));
    return mw_launch_xrd(c, d_Xchol);
}

#include "clrs_mw_ipm_host.inc"
