// clrs_mw_inst.h -- every kernel template of the multi-word path, once, as a list: `X` is `extern template` in the translation
// unit that launches them (clrs_mw.hip) and `template` in the unit that holds the code of one limb count (clrs_mw_inst.hip).
// The kernels of one K are some minutes of device code generation; one unit per K lets them compile side by side.
// A kernel missing from the list is still correct: it is then instantiated where it is launched.
#ifndef CLRS_MW_INST_H
#define CLRS_MW_INST_H

#define MW_KERNELS_K(X, K)                                                                                             \
    X __global__ void k_mw_potrf_x<K>(const MwDev, const double *, double *, int, const double *, double *, int *);    \
    X __global__ void k_mw_factor<K>(const MwDev);                                                                     \
    X __global__ void k_mw_qgram<K>(const MwDev, int);                                                                 \
    X __global__ void k_mw_potrf_q<K>(const MwDev, int, const double *, const int *, int);                             \
    X __global__ void k_mw_factor_pipe<K>(const MwDev, unsigned);                                                      \
    X __global__ void k_mw_factor_pipe64<K>(const MwDev, unsigned);                                                    \
    X __global__ void k_mw_potrf_x_pipe<K>(const MwDev, const double *, double *, const double *, double *, int *, double *, double *, int *, unsigned long long *, unsigned); \
    X __global__ void k_mw_potrf_q_pipe<K>(const MwDev, unsigned, const double *, const int *, int);                   \
    X __global__ void k_mw_bp_diag_pipe<K>(const MwDev, const MwBp *, int, int, unsigned, int, int);                   \
    X __global__ void k_mw_bp_inv_row<K>(const MwDev, const MwBp *, int);                                               \
    X __global__ void k_mw_qsum<K>(const MwDev);                                                                       \
    X __global__ void k_mw_bp_diag<K>(const MwDev, const MwBp *, int, int);                                             \
    X __global__ void k_mw_bp_panel<K>(const MwDev, const MwBp *, int);                                                \
    X __global__ void k_mw_bp_syrk<K>(const MwDev, const MwBp *, int);                                                 \
    X __global__ void k_mw_bp_inv<K>(const MwDev, const MwBp *, int);                                                  \
    X __global__ void k_mw_bp_finish<K>(const MwDev, const MwBp *);                                                    \
    X __global__ void k_mw_usum<K>(const MwDev);                                                                       \
    X __global__ void k_mw_keep_S<K>(const MwDev);                                                                     \
    X __global__ void k_mw_solve_fwd<K>(const MwDev, const double *);                                                  \
    X __global__ void k_mw_solve_mid<K, K>(const MwDev, const double *, double *);                                     \
    X __global__ void k_mw_solve_mid<K, mw_kc(K)>(const MwDev, const double *, double *);                              \
    X __global__ void k_mw_solve_wide<K>(const MwDev, int, const double *, const double *, double *, double *, double *);\
    X __global__ void k_mw_xrd<K>(const MwDev, const double *);                                                        \
    X __global__ void k_mwi_R<K>(const MwDev, const MwIpmDev, int);                                                    \
    X __global__ void k_mwi_Z<K>(const MwDev, const MwIpmDev, int, int);                                               \
    X __global__ void k_mwi_Zi<K>(const MwDev, const MwIpmDev, int);                                                   \
    X __global__ void k_mwi_bmm<K>(const MwDev, const MwIpmDev, int, int);                                             \
    X __global__ void k_mwx_slice<K>(const MwDev, const MwxDev);                                                       \
    X __global__ void k_mwx_gram<K>(const MwDev, const MwxDev);                                                        \
    X __global__ void k_mwi_update<K>(const MwDev, const MwIpmDev);                                                    \
    X __global__ void k_mwi_init<K>(const MwDev, const MwIpmDev, double, double);

#define MW_KERNELS_KD(X, K, DK)                                                                                        \
    X __global__ void k_mw_zt<K, DK>(const MwDev, const double *, int, int, int);                                      \
    X __global__ void k_mw_gram<K, DK>(const MwDev, const double *);                                                                   \
    X __global__ void k_mw_dense_t<K, DK>(const MwDev, const double *, int, int, int);                                 \
    X __global__ void k_mw_dense_tp<K, DK>(const MwDev, const double *);                                               \
    X __global__ void k_mw_dense_s<K, DK>(const MwDev, int);                                                           \
    X __global__ void k_mw_saccum<K, DK>(const MwDev, int);                                                                 \
    X __global__ void k_mw_saccum_one<K, DK>(const MwDev, int);                                                             \
    X __global__ void k_mw_linvb<K, DK>(const MwDev);                                                                  \
    X __global__ void k_mw_refine<K, DK>(const MwDev, int, const double *, double *, double *);                        \
    X __global__ void k_mw_solve_bwd<K, K, DK, 0>(const MwDev, const double *, double *, const double *, double *, const double *);        \
    X __global__ void k_mw_solve_bwd<K, K, DK, 1>(const MwDev, const double *, double *, const double *, double *, const double *);        \
    X __global__ void k_mw_solve_bwd<K, K, DK, 2>(const MwDev, const double *, double *, const double *, double *, const double *);        \
    X __global__ void k_mw_solve_bwd<K, mw_kc(K), DK, 1>(const MwDev, const double *, double *, const double *, double *, const double *); \
    X __global__ void k_mw_solve_bwd<K, mw_kc(K), DK, 2>(const MwDev, const double *, double *, const double *, double *, const double *); \
    X __global__ void k_mwi_scalar<K, DK>(const MwDev, const MwIpmDev, int, int);                                      \
    X __global__ void k_mwi_dots<K, DK>(const MwDev, const MwIpmDev, int, int);                                             \
    X __global__ void k_mwi_coef<K, DK>(const MwDev, const MwIpmDev, const double *);                                  \
    X __global__ void k_mwi_wA<K, DK>(const MwDev, const MwIpmDev, int, int, int);                                          \
    X __global__ void k_mwi_wB<K, DK>(const MwDev, const MwIpmDev, int);                                                    \
    X __global__ void k_mwi_MV<K, DK>(const MwDev, const double *);                                                    \
    X __global__ void k_mwi_rows_dn<K, DK>(const MwDev, const MwIpmDev, int);                                          \
    X __global__ void k_mwi_rows<K, DK>(const MwDev, const MwIpmDev, int, int);                                        \
    X __global__ void k_mwi_rows_fwd<K, DK>(const MwDev, const MwIpmDev);                                              \
    X __global__ void k_mwi_pv<K, DK>(const MwDev, const MwIpmDev, int);                                               \
    X __global__ void k_mwi_pvfin<K, DK>(const MwDev, const MwIpmDev);                                                 \
    X __global__ void k_mwi_gpack<K, DK>(const MwDev, const MwIpmDev, int);                                               \
    X __global__ void k_mwi_step<K, DK>(const MwDev, const MwIpmDev, int, int, int, int);

// the exact slice products on the matrix cores (clrs_mw_exact.hip.h): data limbs 1 and 2 only (their static digits are cut from two data limbs)
#define MW_KERNELS_KDX(X, K, DK)                                                                                       \
    X __global__ void k_mws_pair<K, DK, 1>(const MwDev, const MwsDev, const double *);                                 \
    X __global__ void k_mws_pair<K, DK, 2>(const MwDev, const MwsDev, const double *);                                 \
    X __global__ void k_mwx_dense<K, DK>(const MwDev, const MwdDev, const double *);

// (data limbs 1, 2 and K: fp64 data, double-double data, data at the working precision)
#define MW_KERNELS_ALL(X, K) MW_KERNELS_K(X, K) MW_KERNELS_KD(X, K, 1) MW_KERNELS_KDX(X, K, 1) MW_KERNELS_KD(X, K, 2) MW_KERNELS_KDX(X, K, 2) MW_KERNELS_KD(X, K, K)

#endif
