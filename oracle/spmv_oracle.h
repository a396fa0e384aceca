/*
 * oracle/spmv_oracle.h -- CPU restatement of the LIBSPMV hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in spmv_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the timed
 * CPU baseline.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic and evaluation order it restates.  Arithmetic is compiled
 * with -ffp-contract=off: the reference is built for baseline x86-64 (no FMA
 * target), so every multiply and add rounds separately.
 *
 * Parity status:
 *   SpMV (general, symmetric), gather, halo plan: pinned by the reference's
 *     own known-answer test (tests/test_spmv.cpp:56-80,159-160) and by the
 *     reference outputs recorded in SURVEY.md App. B / section 8c
 *     (tests/golden/kat.json).
 *   CG: PARITY UNPINNED -- no reference test calls cg(), and its ddot comes
 *     from an unpinned BLAS (cg.cpp:47,63,73).  The oracle fixes ddot to a
 *     left-to-right sum.
 */
#ifndef SPMV_ORACLE_H
#define SPMV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* spmv/csr_kernels.cpp:41-51 -- general CSR branch, rows ascending,
 * left-to-right sum, out[i] = alpha*sum + beta*out[i]. */
void oracle_csr_spmv(int32_t num_rows, const int32_t* rowptr,
                     const int32_t* colind, const double* values, double alpha,
                     const double* in, double beta, double* out);

/* spmv/csr_kernels.cpp:26-40 -- symmetric branch: strictly-lower CSR +
 * dense diagonal, sequential scatter into out[col]. rowptr may be NULL when
 * num_non_zeros == 0 (csr_matrix.cpp:34). */
void oracle_csr_spmv_sym(int32_t num_rows, int64_t num_non_zeros,
                         const int32_t* rowptr, const int32_t* colind,
                         const double* values, const double* diagonal,
                         double alpha, const double* in, double beta,
                         double* out);

/* fp32 instantiations (csr_kernels.cpp:63). */
void oracle_csr_spmv_f32(int32_t num_rows, const int32_t* rowptr,
                         const int32_t* colind, const float* values,
                         float alpha, const float* in, float beta, float* out);
void oracle_csr_spmv_sym_f32(int32_t num_rows, int64_t num_non_zeros,
                             const int32_t* rowptr, const int32_t* colind,
                             const float* values, const float* diagonal,
                             float alpha, const float* in, float beta,
                             float* out);

/* spmv/reference_executor.cpp:150-164 -- out[i] = in[indices[i]]. */
void oracle_gather_ghosts(int num_indices, const int32_t* indices,
                          const double* in, double* out);

/* spmv/openmp/csr_kernels.openmp.cpp:56-87 -- static nnz-balanced row split.
 * row_split has num_threads+1 entries. */
void oracle_omp_row_split(int32_t num_rows, int64_t num_non_zeros,
                          const int32_t* rowptr, int num_threads,
                          int32_t* row_split);

/* OpenMP path, general kernel (csr_kernels.openmp.cpp:226-242). Opaque plan
 * = the aux_data the reference builds in init (:26-170). */
typedef struct oracle_omp_plan oracle_omp_plan;
oracle_omp_plan* oracle_omp_init(int32_t num_rows, int64_t num_non_zeros,
                                 const int32_t* rowptr, const int32_t* colind,
                                 int symmetric, int num_threads);
void oracle_omp_free(oracle_omp_plan* plan);
void oracle_omp_spmv(const oracle_omp_plan* plan, int32_t num_rows,
                     int64_t num_non_zeros, const int32_t* rowptr,
                     const int32_t* colind, const double* values,
                     const double* diagonal, double alpha, const double* in,
                     double beta, double* out);

/* BLAS-1 with a pinned evaluation order (cg.cpp:47,63,69,70,73,84,85 call
 * cblas; order of ddot is implementation-defined there). */
double oracle_ddot(int64_t n, const double* x, const double* y);

/*
 * spmv/cg.cpp:21-98 for ONE rank and a matrix without ghosts (general CSR,
 * or symmetric CSR when diagonal != NULL).  x0 = 0, r = p = b.
 * Writes x[0..n), returns iteration count k.  If rnorm_hist != NULL it
 * receives rnorm0 followed by rnorm_new of every iteration (k+1 doubles).
 * num_threads <= 1 uses the sequential kernels (ReferenceExecutor
 * semantics); > 1 uses the OpenMP restatement (cg.openmp.cpp:23-100) with
 * BLAS-1 as OpenMP loops (per-thread partial dots summed in thread order).
 */
int oracle_cg(int32_t n, int64_t nnz, const int32_t* rowptr,
              const int32_t* colind, const double* values,
              const double* diagonal, const double* b, double* x, int kmax,
              double rtol, double* rnorm_hist, int num_threads);

/* Wall-clock helper for the cpu_baseline leg: runs `reps` applies of the
 * OpenMP (or sequential when num_threads<=1) general/symmetric SpMV and
 * returns seconds per apply (first-touch done by caller). */
double oracle_time_spmv(int32_t num_rows, int64_t nnz, const int32_t* rowptr,
                        const int32_t* colind, const double* values,
                        const double* diagonal, const double* in, double* out,
                        int reps, int num_threads);

void oracle_poisson3d(int32_t n, int32_t* rowptr, int32_t* colind,
                      double* values);
void oracle_poisson3d_lower(int32_t n, int32_t* rowptr, int32_t* colind,
                            double* values, double* diagonal);
double oracle_time_cg(int32_t n, int64_t nnz, const int32_t* rowptr,
                      const int32_t* colind, const double* values,
                      const double* b, double* x, int kmax, int num_threads,
                      int* iterations);
/* cpu_baseline leg: OpenMP SpMV + CG on the n^3 Poisson matrix with owner
 * first touch, spread threads, loops timed alone (see the .c file) */
typedef struct {
  double spmv_s_per_apply;
  double cg_loop_s;
  double setup_s;
  double rel_residual;
  double rel_residual_k10; /* after 10 iterations (0 when fewer were run) */
  int cg_iters;
  int threads;
} oracle_baseline_result;
int oracle_cpu_baseline(int32_t n, int threads, int spmv_reps, int cg_iters,
                        oracle_baseline_result* res);
int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
