/*
 * oracle/spmv_oracle.c -- CPU restatement of the LIBSPMV hot path.
 * TEST INFRASTRUCTURE ONLY (see spmv_oracle.h).  Build: oracle/Makefile.
 * All file:line citations are relative to /root/reference.
 */
#include "spmv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* spmv/csr_kernels.cpp:41-51 */
void oracle_csr_spmv(int32_t num_rows, const int32_t* rowptr,
                     const int32_t* colind, const double* values, double alpha,
                     const double* in, double beta, double* out)
{
  for (int32_t i = 0; i < num_rows; ++i) {
    double sum = 0.0;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
      sum += values[j] * in[colind[j]];
    out[i] = alpha * sum + beta * out[i];
  }
}

/* spmv/csr_kernels.cpp:26-40 */
void oracle_csr_spmv_sym(int32_t num_rows, int64_t num_non_zeros,
                         const int32_t* rowptr, const int32_t* colind,
                         const double* values, const double* diagonal,
                         double alpha, const double* in, double beta,
                         double* out)
{
  for (int32_t i = 0; i < num_rows; ++i) {
    double sum = diagonal[i] * in[i];
    if (num_non_zeros > 0) {
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        int32_t col = colind[j];
        double val = values[j];
        sum += val * in[col];
        out[col] += alpha * val * in[i];
      }
    }
    out[i] = alpha * sum + beta * out[i];
  }
}

/* fp32 instantiation, spmv/csr_kernels.cpp:63 */
void oracle_csr_spmv_f32(int32_t num_rows, const int32_t* rowptr,
                         const int32_t* colind, const float* values,
                         float alpha, const float* in, float beta, float* out)
{
  for (int32_t i = 0; i < num_rows; ++i) {
    float sum = 0.0f;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
      sum += values[j] * in[colind[j]];
    out[i] = alpha * sum + beta * out[i];
  }
}

void oracle_csr_spmv_sym_f32(int32_t num_rows, int64_t num_non_zeros,
                             const int32_t* rowptr, const int32_t* colind,
                             const float* values, const float* diagonal,
                             float alpha, const float* in, float beta,
                             float* out)
{
  for (int32_t i = 0; i < num_rows; ++i) {
    float sum = diagonal[i] * in[i];
    if (num_non_zeros > 0) {
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        int32_t col = colind[j];
        float val = values[j];
        sum += val * in[col];
        out[col] += alpha * val * in[i];
      }
    }
    out[i] = alpha * sum + beta * out[i];
  }
}

/* spmv/reference_executor.cpp:150-164 */
void oracle_gather_ghosts(int num_indices, const int32_t* indices,
                          const double* in, double* out)
{
  for (int i = 0; i < num_indices; ++i)
    out[i] = in[indices[i]];
}

/* ------------------------------------------------------------------------- */
/* spmv/openmp/csr_kernels.openmp.cpp:41-87 */
void oracle_omp_row_split(int32_t num_rows, int64_t num_non_zeros,
                          const int32_t* rowptr, int num_threads,
                          int32_t* row_split)
{
  if (num_threads == 1) { /* :41-44 */
    row_split[0] = 0;
    row_split[1] = num_rows;
    return;
  }
  int32_t nnz_per_split
      = (int32_t)((num_non_zeros + num_threads - 1) / num_threads); /* :57 */
  int32_t curr_nnz = 0, row_start = 0, split_cnt = 0;
  row_split[0] = row_start;
  for (int32_t i = 0; i < num_rows; i++) { /* :64-73 */
    curr_nnz += rowptr[i + 1] - rowptr[i];
    if (curr_nnz >= nnz_per_split) {
      row_start = i + 1;
      ++split_cnt;
      if (split_cnt <= num_threads)
        row_split[split_cnt] = row_start;
      curr_nnz = 0;
    }
  }
  if (curr_nnz < nnz_per_split && split_cnt <= num_threads) /* :76-78 */
    row_split[++split_cnt] = num_rows;
  if (split_cnt > num_threads) /* :81-83 */
    row_split[num_threads] = num_rows;
  for (int32_t i = split_cnt + 1; i <= num_threads; i++) /* :86-88 */
    row_split[i] = num_rows;
}

struct oracle_omp_plan {
  int num_threads;
  int symmetric;
  int32_t* row_split;
  /* symmetric only: "local vectors" scratch + conflict list (:94-167) */
  double* buffer;      /* num_threads * num_rows, zero-filled */
  int32_t* cnfl_pos;   /* target row of each conflict, ascending            */
  int16_t* cnfl_src;   /* thread whose private vector holds the addend      */
  int32_t* cnfl_start; /* per-thread [start,end) in the conflict list       */
  int32_t* cnfl_end;
  int32_t ncnfls;
};

static int cmp_i64(const void* a, const void* b)
{
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* spmv/openmp/csr_kernels.openmp.cpp:26-170 */
oracle_omp_plan* oracle_omp_init(int32_t num_rows, int64_t num_non_zeros,
                                 const int32_t* rowptr, const int32_t* colind,
                                 int symmetric, int num_threads)
{
  oracle_omp_plan* p = (oracle_omp_plan*)calloc(1, sizeof(*p));
  if (num_threads < 1)
    num_threads = 1;
  p->num_threads = num_threads;
  p->symmetric = symmetric;
  p->row_split = (int32_t*)malloc(sizeof(int32_t) * (num_threads + 1));
  if (num_non_zeros > 0) {
    oracle_omp_row_split(num_rows, num_non_zeros, rowptr, num_threads,
                         p->row_split);
  } else {
    /* The reference builds no aux data for an empty block (:34); a plain
     * even split keeps the diagonal-only loop (:222-225) well defined. */
    for (int t = 0; t <= num_threads; ++t)
      p->row_split[t] = (int32_t)(((int64_t)num_rows * t) / num_threads);
  }
  p->cnfl_start = (int32_t*)calloc(num_threads, sizeof(int32_t));
  p->cnfl_end = (int32_t*)calloc(num_threads, sizeof(int32_t));
  if (!symmetric || num_threads == 1 || num_non_zeros == 0)
    return p;

  /* :94 -- scratch, zero-filled here (the reference relies on fresh pages). */
  p->buffer = (double*)calloc((size_t)num_threads * num_rows, sizeof(double));

  /* :97-113 -- conflicts: (target_row, tid) pairs with target_row below the
   * thread's first row; unique per (row, tid). Encoded row*2^16+tid so one
   * sort gives the std::map order (ascending row); within a row we fix
   * ascending tid (the reference iterates an unordered_set there). */
  size_t cap = 1024, cnt = 0;
  int64_t* keys = (int64_t*)malloc(cap * sizeof(int64_t));
  for (int tid = 1; tid < num_threads; ++tid) {
    int32_t r0 = p->row_split[tid];
    for (int32_t i = r0; i < p->row_split[tid + 1]; ++i)
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
        if (colind[j] < r0) {
          if (cnt == cap) {
            cap *= 2;
            keys = (int64_t*)realloc(keys, cap * sizeof(int64_t));
          }
          keys[cnt++] = ((int64_t)colind[j] << 16) | tid;
        }
  }
  qsort(keys, cnt, sizeof(int64_t), cmp_i64);
  size_t u = 0;
  for (size_t k = 0; k < cnt; ++k)
    if (u == 0 || keys[k] != keys[u - 1])
      keys[u++] = keys[k];
  p->ncnfls = (int32_t)u;
  p->cnfl_pos = (int32_t*)malloc(sizeof(int32_t) * (u ? u : 1));
  p->cnfl_src = (int16_t*)malloc(sizeof(int16_t) * (u ? u : 1));
  for (size_t k = 0; k < u; ++k) { /* :118-131 */
    p->cnfl_pos[k] = (int32_t)(keys[k] >> 16);
    p->cnfl_src[k] = (int16_t)(keys[k] & 0xffff);
  }
  free(keys);

  /* :133-167 -- split the reduction so that all conflicts of one row go to
   * one thread.  Which thread performs an add does not change its value, so
   * the split rule is restated as: equal chunks, cut moved forward to the
   * next row boundary. */
  int32_t start = 0;
  for (int t = 0; t < num_threads; ++t) {
    int32_t end = (int32_t)(((int64_t)p->ncnfls * (t + 1)) / num_threads);
    if (end < start)
      end = start;
    while (end > 0 && end < p->ncnfls
           && p->cnfl_pos[end] == p->cnfl_pos[end - 1])
      ++end;
    if (t == num_threads - 1)
      end = p->ncnfls;
    p->cnfl_start[t] = start;
    p->cnfl_end[t] = end;
    start = end;
  }
  return p;
}

void oracle_omp_free(oracle_omp_plan* p)
{
  if (!p)
    return;
  free(p->row_split);
  free(p->buffer);
  free(p->cnfl_pos);
  free(p->cnfl_src);
  free(p->cnfl_start);
  free(p->cnfl_end);
  free(p);
}

/* spmv/openmp/csr_kernels.openmp.cpp:172-244 */
void oracle_omp_spmv(const oracle_omp_plan* plan, int32_t num_rows,
                     int64_t num_non_zeros, const int32_t* rowptr,
                     const int32_t* colind, const double* values,
                     const double* diagonal, double alpha, const double* in,
                     double beta, double* out)
{
  const int nt = plan->num_threads;
  const int32_t* row_split = plan->row_split;
  if (plan->symmetric && num_non_zeros > 0) { /* :179-221 */
    double* buffer = plan->buffer;
    const int16_t* cnfl_src = plan->cnfl_src;
    const int32_t* cnfl_pos = plan->cnfl_pos;
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      const int32_t row_offset = row_split[tid];
      /* local vectors phase, :195-212 */
      for (int32_t i = row_split[tid]; i < row_split[tid + 1]; ++i) {
        double sum = diagonal[i] * in[i];
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
          int32_t col = colind[j];
          double val = values[j];
          sum += val * in[col];
          if (col < row_offset)
            buffer[(size_t)tid * num_rows + col] += val * in[i];
          else
            out[col] += alpha * val * in[i];
        }
        out[i] = alpha * sum + beta * out[i];
      }
#pragma omp barrier
      /* reduction of conflicts phase, :215-220 */
      for (int32_t i = plan->cnfl_start[tid]; i < plan->cnfl_end[tid]; ++i) {
        int16_t vid = cnfl_src[i];
        int32_t pos = cnfl_pos[i];
        out[pos] += alpha * buffer[(size_t)vid * num_rows + pos];
        buffer[(size_t)vid * num_rows + pos] = 0.0;
      }
    }
  } else if (plan->symmetric) { /* :222-225 */
#pragma omp parallel for num_threads(nt)
    for (int32_t i = 0; i < num_rows; i++)
      out[i] = alpha * diagonal[i] * in[i] + beta * out[i];
  } else { /* :226-242 */
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      for (int32_t i = row_split[tid]; i < row_split[tid + 1]; ++i) {
        double sum = 0.0;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
          sum += values[j] * in[colind[j]];
        out[i] = alpha * sum + beta * out[i];
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* ddot with a pinned (left-to-right) order; see header. */
double oracle_ddot(int64_t n, const double* x, const double* y)
{
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i)
    s += x[i] * y[i];
  return s;
}

/* BLAS-1 as OpenMP loops: static contiguous chunks, partials summed in
 * thread order (deterministic for a fixed thread count). */
static double omp_ddot(int64_t n, const double* x, const double* y, int nt)
{
  if (nt <= 1)
    return oracle_ddot(n, x, y);
  double part[1024];
  for (int t = 0; t < nt; ++t)
    part[t] = 0.0;
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num();
#else
    int t = 0;
#endif
    int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
    double s = 0.0;
    for (int64_t i = lo; i < hi; ++i)
      s += x[i] * y[i];
    part[t] = s;
  }
  double s = 0.0;
  for (int t = 0; t < nt; ++t)
    s += part[t];
  return s;
}

static void omp_daxpy(int64_t n, double a, const double* x, double* y, int nt)
{
#pragma omp parallel for num_threads(nt > 1 ? nt : 1) schedule(static)
  for (int64_t i = 0; i < n; ++i)
    y[i] += a * x[i];
}

static void omp_dscal(int64_t n, double a, double* x, int nt)
{
#pragma omp parallel for num_threads(nt > 1 ? nt : 1) schedule(static)
  for (int64_t i = 0; i < n; ++i)
    x[i] *= a;
}

/* spmv/cg.cpp:21-98 (sequential) and spmv/openmp/cg.openmp.cpp:23-100 */
int oracle_cg(int32_t n, int64_t nnz, const int32_t* rowptr,
              const int32_t* colind, const double* values,
              const double* diagonal, const double* b, double* x, int kmax,
              double rtol, double* rnorm_hist, int num_threads)
{
  const int nt = num_threads < 1 ? 1 : num_threads;
  const int symmetric = diagonal != NULL;
  oracle_omp_plan* plan = NULL;
  if (nt > 1)
    plan = oracle_omp_init(n, nnz, rowptr, colind, symmetric, nt);

  double* r = (double*)malloc(sizeof(double) * n);         /* :39 */
  double* Ap = (double*)calloc(n, sizeof(double));         /* :40, zeroed F7b */
  double* x_padded = (double*)calloc(n, sizeof(double));   /* :41, x0=0 F7a */
  double* p = (double*)malloc(sizeof(double) * n);         /* :42 */
  memcpy(r, b, sizeof(double) * n);                        /* :44 */
  memcpy(p, b, sizeof(double) * n);                        /* :45 */

  double rnorm = omp_ddot(n, r, r, nt); /* :47 */
  double rnorm0 = sqrt(rnorm);          /* :48-50 (Allreduce over 1 rank) */
  if (rnorm_hist)
    rnorm_hist[0] = rnorm0;

  double rnorm_old = rnorm0;
  int k = 0;
  while (k < kmax) { /* :55 */
    ++k;
    /* :59-60 -- halo update is a no-op on one rank; Ap = A p (beta = 0,
     * Matrix.cpp:483-486 / :523-530). The symmetric kernel accumulates into
     * rows it has already finalised, so Ap needs no pre-zeroing in the
     * sequential order, but the OpenMP variant writes out[col] += before
     * other threads finalise -- zero it as the tests do (test_spmv.cpp:135). */
    if (symmetric && nt > 1)
      memset(Ap, 0, sizeof(double) * n);
    if (nt > 1)
      oracle_omp_spmv(plan, n, nnz, rowptr, colind, values, diagonal, 1.0, p,
                      0.0, Ap);
    else if (symmetric)
      oracle_csr_spmv_sym(n, nnz, rowptr, colind, values, diagonal, 1.0, p,
                          0.0, Ap);
    else
      oracle_csr_spmv(n, rowptr, colind, values, 1.0, p, 0.0, Ap);

    double pdotAp = omp_ddot(n, p, Ap, nt);              /* :63-65 */
    double alpha = (rnorm_old * rnorm_old) / pdotAp;     /* :66 */
    omp_daxpy(n, alpha, p, x_padded, nt);                /* :69 */
    omp_daxpy(n, -alpha, Ap, r, nt);                     /* :70 */
    rnorm = omp_ddot(n, r, r, nt);                       /* :73 */
    double rnorm_new = sqrt(rnorm);                      /* :74-76 */
    double beta = (rnorm_new * rnorm_new) / (rnorm_old * rnorm_old); /* :77 */
    rnorm_old = rnorm_new;                               /* :78 */
    if (rnorm_hist)
      rnorm_hist[k] = rnorm_new;
    if (rnorm_new / rnorm0 < rtol)                       /* :80-81 */
      break;
    omp_dscal(n, beta, p, nt);                           /* :84 */
    omp_daxpy(n, 1.0, r, p, nt);                         /* :85 */
  }
  memcpy(x, x_padded, sizeof(double) * n); /* :89 */
  free(r);
  free(Ap);
  free(x_padded);
  free(p);
  oracle_omp_free(plan);
  return k;
}

/* ------------------------------------------------------------------------- */
static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double oracle_time_spmv(int32_t num_rows, int64_t nnz, const int32_t* rowptr,
                        const int32_t* colind, const double* values,
                        const double* diagonal, const double* in, double* out,
                        int reps, int num_threads)
{
  const int nt = num_threads < 1 ? 1 : num_threads;
  const int symmetric = diagonal != NULL;
  oracle_omp_plan* plan
      = oracle_omp_init(num_rows, nnz, rowptr, colind, symmetric, nt);
  /* warm-up (demos/spmv.cpp:82-84) */
  if (symmetric)
    memset(out, 0, sizeof(double) * num_rows);
  oracle_omp_spmv(plan, num_rows, nnz, rowptr, colind, values, diagonal, 1.0,
                  in, 0.0, out);
  double t0 = now_s();
  for (int r = 0; r < reps; ++r) {
    if (symmetric)
      memset(out, 0, sizeof(double) * num_rows);
    oracle_omp_spmv(plan, num_rows, nnz, rowptr, colind, values, diagonal, 1.0,
                    in, 0.0, out);
  }
  double t1 = now_s();
  oracle_omp_free(plan);
  return (t1 - t0) / (reps > 0 ? reps : 1);
}

/* Synthetic input of the cpu_baseline leg: the same 3-D 7-point Poisson
 * matrix the product generates (not a reference
 * function).  Caller provides rowptr[n^3+1], colind/values[7n^3-6n^2]. */
void oracle_poisson3d(int32_t n, int32_t* rowptr, int32_t* colind,
                      double* values)
{
  const int64_t n2 = (int64_t)n * n, N = n2 * n;
  int64_t pos = 0;
  rowptr[0] = 0;
  for (int64_t i = 0; i < N; ++i) {
    const int64_t x = i % n, y = (i / n) % n, z = i / n2;
    if (z > 0) { colind[pos] = (int32_t)(i - n2); values[pos++] = -1.0; }
    if (y > 0) { colind[pos] = (int32_t)(i - n); values[pos++] = -1.0; }
    if (x > 0) { colind[pos] = (int32_t)(i - 1); values[pos++] = -1.0; }
    colind[pos] = (int32_t)i; values[pos++] = 6.0;
    if (x < n - 1) { colind[pos] = (int32_t)(i + 1); values[pos++] = -1.0; }
    if (y < n - 1) { colind[pos] = (int32_t)(i + n); values[pos++] = -1.0; }
    if (z < n - 1) { colind[pos] = (int32_t)(i + n2); values[pos++] = -1.0; }
    rowptr[i + 1] = (int32_t)pos;
  }
}

/* The same matrix in the symmetric storage of Matrix.cpp:337-349 on one rank:
 * strictly-lower CSR (rowptr[n^3+1], colind/values[3n^3-3n^2]) + diagonal. */
void oracle_poisson3d_lower(int32_t n, int32_t* rowptr, int32_t* colind,
                            double* values, double* diagonal)
{
  const int64_t n2 = (int64_t)n * n, N = n2 * n;
  int64_t pos = 0;
  rowptr[0] = 0;
  for (int64_t i = 0; i < N; ++i) {
    const int64_t x = i % n, y = (i / n) % n, z = i / n2;
    if (z > 0) { colind[pos] = (int32_t)(i - n2); values[pos++] = -1.0; }
    if (y > 0) { colind[pos] = (int32_t)(i - n); values[pos++] = -1.0; }
    if (x > 0) { colind[pos] = (int32_t)(i - 1); values[pos++] = -1.0; }
    diagonal[i] = 6.0;
    rowptr[i + 1] = (int32_t)pos;
  }
}

/* Wall-clock seconds of one oracle_cg call (cpu_baseline leg). */
double oracle_time_cg(int32_t n, int64_t nnz, const int32_t* rowptr,
                      const int32_t* colind, const double* values,
                      const double* b, double* x, int kmax, int num_threads,
                      int* iterations)
{
  double t0 = now_s();
  int k = oracle_cg(n, nnz, rowptr, colind, values, NULL, b, x, kmax, 0.0, NULL,
                    num_threads);
  double t1 = now_s();
  if (iterations)
    *iterations = k;
  return t1 - t0;
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline leg of bench.py: the OpenMP path above on the benchmark's own
 * matrix, set up the way SURVEY section 8d asks: every array is first touched
 * by the thread that later streams it (same static nnz-balanced row split as
 * the kernel, openmp/csr_kernels.openmp.cpp:56-87), threads spread over the
 * cores (OMP_PLACES=cores OMP_PROC_BIND=spread in the environment, proc_bind
 * here), and only the apply / iteration loops are timed -- no set-up, no
 * allocation, no copies. */
int oracle_cpu_baseline(int32_t n, int threads, int spmv_reps, int cg_iters,
                        oracle_baseline_result* res)
{
  const int nt = threads < 1 ? 1 : threads;
  const int64_t n2 = (int64_t)n * n, N = n2 * n;
  const int64_t nnz = 7 * N - 6 * n2;
  if (N > INT32_MAX || nnz > INT32_MAX)
    return 1;
  double t0 = now_s();
  int32_t* rowptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int32_t* colind = (int32_t*)malloc(sizeof(int32_t) * (size_t)nnz);
  double* values = (double*)malloc(sizeof(double) * (size_t)nnz);
  double *b = (double*)malloc(sizeof(double) * (size_t)N),
         *x = (double*)malloc(sizeof(double) * (size_t)N),
         *r = (double*)malloc(sizeof(double) * (size_t)N),
         *p = (double*)malloc(sizeof(double) * (size_t)N),
         *Ap = (double*)malloc(sizeof(double) * (size_t)N);
  if (!rowptr || !colind || !values || !b || !x || !r || !p || !Ap)
    return 2;
  /* row lengths (even split: this array is read by all threads anyway) */
  rowptr[0] = 0;
#pragma omp parallel for num_threads(nt) proc_bind(spread) schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    const int64_t xx = i % n, yy = (i / n) % n, zz = i / n2;
    rowptr[i + 1] = 1 + (xx > 0) + (xx < n - 1) + (yy > 0) + (yy < n - 1)
                    + (zz > 0) + (zz < n - 1);
  }
  for (int64_t i = 0; i < N; ++i)
    rowptr[i + 1] += rowptr[i];
  oracle_omp_plan* plan = oracle_omp_init((int32_t)N, nnz, rowptr, NULL, 0, nt);
  const int32_t* row_split = plan->row_split;
  /* matrix entries and vectors: first touch by the owner of the rows */
#pragma omp parallel num_threads(nt) proc_bind(spread)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    for (int64_t i = row_split[tid]; i < row_split[tid + 1]; ++i) {
      const int64_t xx = i % n, yy = (i / n) % n, zz = i / n2;
      int64_t pos = rowptr[i];
      if (zz > 0) { colind[pos] = (int32_t)(i - n2); values[pos++] = -1.0; }
      if (yy > 0) { colind[pos] = (int32_t)(i - n); values[pos++] = -1.0; }
      if (xx > 0) { colind[pos] = (int32_t)(i - 1); values[pos++] = -1.0; }
      colind[pos] = (int32_t)i; values[pos++] = 6.0;
      if (xx < n - 1) { colind[pos] = (int32_t)(i + 1); values[pos++] = -1.0; }
      if (yy < n - 1) { colind[pos] = (int32_t)(i + n); values[pos++] = -1.0; }
      if (zz < n - 1) { colind[pos] = (int32_t)(i + n2); values[pos++] = -1.0; }
      /* the right-hand side of the GPU line: the reference's Gaussian vector
         (demos/spmv.cpp:63-67), so that the residual after 10 iterations can
         be read beside the GPU's */
      const double z = (double)i / (double)N;
      const double u = 5 * (z - 0.5);
      b[i] = exp(-10 * (u * u));
      x[i] = 0.0;
      r[i] = b[i]; /* cg.cpp:44-45: r = p = b */
      p[i] = b[i];
      Ap[i] = 0.0;
    }
  }
  res->setup_s = now_s() - t0;
  res->threads = nt;
  res->rel_residual_k10 = 0.0;
  /* SpMV: 1 warm-up + reps timed (demos/spmv.cpp:73-96) */
  oracle_omp_spmv(plan, (int32_t)N, nnz, rowptr, colind, values, NULL, 1.0, p,
                  0.0, Ap);
  t0 = now_s();
  for (int k = 0; k < spmv_reps; ++k)
    oracle_omp_spmv(plan, (int32_t)N, nnz, rowptr, colind, values, NULL, 1.0, p,
                    0.0, Ap);
  res->spmv_s_per_apply = (now_s() - t0) / (spmv_reps > 0 ? spmv_reps : 1);
  /* CG iterations (openmp/cg.openmp.cpp:55-88), the loop alone */
  double rnorm0 = sqrt(omp_ddot(N, r, r, nt));
  double rnorm_old = rnorm0;
  int k = 0;
  t0 = now_s();
  while (k < cg_iters) {
    ++k;
    oracle_omp_spmv(plan, (int32_t)N, nnz, rowptr, colind, values, NULL, 1.0, p,
                    0.0, Ap);
    double pdotAp = omp_ddot(N, p, Ap, nt);
    double alpha = (rnorm_old * rnorm_old) / pdotAp;
    omp_daxpy(N, alpha, p, x, nt);
    omp_daxpy(N, -alpha, Ap, r, nt);
    double rnorm_new = sqrt(omp_ddot(N, r, r, nt));
    double beta = (rnorm_new * rnorm_new) / (rnorm_old * rnorm_old);
    rnorm_old = rnorm_new;
    if (k == 10)
      res->rel_residual_k10 = rnorm_new / rnorm0;
    omp_dscal(N, beta, p, nt);
    omp_daxpy(N, 1.0, r, p, nt);
  }
  res->cg_loop_s = now_s() - t0;
  res->cg_iters = k;
  res->rel_residual = rnorm_old / rnorm0;
  oracle_omp_free(plan);
  free(rowptr); free(colind); free(values);
  free(b); free(x); free(r); free(p); free(Ap);
  return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
