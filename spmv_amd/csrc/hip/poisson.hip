// 3-D 7-point Poisson matrix generator, directly into device memory.
//
// Not part of the reference (demos/CreateA.cpp builds a
// 1-D tridiagonal matrix).  The structure -- contiguous row ranges per rank,
// ghost columns renumbered after the owned ones in ascending global order --
// follows Matrix::create_matrix (spmv/Matrix.cpp:295-318) so the blocks this
// kernel writes are exactly what create_matrix would build from the host
// CSR; tests compare the two at small n.  Setup only, untimed.
//
// Grid n^3, natural ordering i = x + n (y + n z), diagonal 6, off-diagonal
// -1, neighbours outside the grid dropped.  Within a row, entries are stored
// in ascending local column order (owned columns, then ghosts).
//
// Context option "poisson_stencil" = 27: the 27-point operator on the same
// grid (all neighbours with |dx|, |dy|, |dz| <= 1; diagonal 26, off-diagonal
// -1 -- HPCG's matrix), for measurements of the kernels on wider stencils.
// Its row slabs must be whole planes (or the whole matrix).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace
{

constexpr int kMaxStencil = 27;

struct Geom {
  int32_t n;
  int32_t points; // 7 or 27
  int64_t n2, N;
  int64_t r0, r1;       // owned global rows (= owned global columns)
  int64_t gb_start, gb; // ghosts below: [gb_start, r0), gb of them
  int64_t ga;           // ghosts above: [r1, r1 + ga)
};

__host__ __device__ inline bool keep(int part, int64_t row, int64_t col,
                                     int64_t r0, int64_t r1)
{
  const bool owned = col >= r0 && col < r1;
  switch (part) {
  case SPMV_HIP_PART_ALL: return true;
  case SPMV_HIP_PART_LOCAL: return owned;
  case SPMV_HIP_PART_REMOTE: return !owned;
  default: return owned && row > col; // LOCAL_LOWER (Matrix.cpp:343-344)
  }
}

__device__ inline int32_t local_col(const Geom& g, int64_t col)
{
  if (col < g.r0)
    return (int32_t)((g.r1 - g.r0) + (col - g.gb_start));
  if (col >= g.r1)
    return (int32_t)((g.r1 - g.r0) + g.gb + (col - g.r1));
  return (int32_t)(col - g.r0);
}

// neighbours of global row i in ascending column order; returns count
__device__ inline int stencil(const Geom& g, int64_t i, int64_t* cols)
{
  const int64_t x = i % g.n, y = (i / g.n) % g.n, z = i / g.n2;
  int c = 0;
  if (g.points == 27) {
    // (dz, dy, dx) in lexicographic order = ascending index for n >= 3
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int64_t xx = x + dx, yy = y + dy, zz = z + dz;
          if (xx >= 0 && xx < g.n && yy >= 0 && yy < g.n && zz >= 0 && zz < g.n)
            cols[c++] = i + dz * g.n2 + dy * (int64_t)g.n + dx;
        }
    return c;
  }
  if (z > 0) cols[c++] = i - g.n2;
  if (y > 0) cols[c++] = i - g.n;
  if (x > 0) cols[c++] = i - 1;
  cols[c++] = i;
  if (x < g.n - 1) cols[c++] = i + 1;
  if (y < g.n - 1) cols[c++] = i + g.n;
  if (z < g.n - 1) cols[c++] = i + g.n2;
  return c;
}

__global__ __launch_bounds__(kBlock) void poisson_count_kernel(Geom g, int part,
                                                               int32_t* counts)
{
  const int64_t nrows = g.r1 - g.r0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= nrows;
       k += (int64_t)gridDim.x * blockDim.x) {
    int cnt = 0;
    if (k < nrows) {
      int64_t cols[kMaxStencil];
      const int64_t i = g.r0 + k;
      const int c = stencil(g, i, cols);
      for (int e = 0; e < c; ++e)
        cnt += keep(part, i, cols[e], g.r0, g.r1) ? 1 : 0;
    }
    counts[k] = cnt; // counts[nrows] = 0 closes the exclusive scan
  }
}

// skew != 0: a NON-symmetric variant for measurements (convection-diffusion
// like): lower neighbours -1 - skew, upper neighbours -1 + skew
__global__ __launch_bounds__(kBlock) void poisson_fill_kernel(
    Geom g, int part, const int32_t* __restrict__ rowptr,
    int32_t* __restrict__ colind, double* __restrict__ values,
    double* __restrict__ diagonal, double skew)
{
  const int64_t nrows = g.r1 - g.r0;
  const double diag = g.points == 27 ? 26.0 : 6.0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nrows;
       k += (int64_t)gridDim.x * blockDim.x) {
    int64_t cols[kMaxStencil];
    const int64_t i = g.r0 + k;
    const int c = stencil(g, i, cols);
    // Entries are stored in ascending LOCAL column order, as create_matrix
    // leaves them (Eigen setFromTriplets sorts each row): owned columns
    // first, then ghost columns (which are numbered after the owned ones).
    int64_t pos = rowptr[k];
    for (int pass = 0; pass < 2; ++pass)
      for (int e = 0; e < c; ++e) {
        const bool owned = cols[e] >= g.r0 && cols[e] < g.r1;
        if (owned != (pass == 0) || !keep(part, i, cols[e], g.r0, g.r1))
          continue;
        colind[pos] = local_col(g, cols[e]);
        values[pos] = (cols[e] == i) ? diag
                                     : (cols[e] < i ? -1.0 - skew : -1.0 + skew);
        ++pos;
      }
    if (diagonal)
      diagonal[k] = diag;
  }
}

// ---------------------------------------------------------------------------
// Seeded UNSTRUCTURED test matrix (measurements of the general kernels on a
// matrix without lattice or narrow-band structure; spmv_amd/poisson.py holds
// the numpy twin the tests compare with).  Square, `per_row` entries in every
// row; entry e = row * per_row + k draws h = mix64((e + 1) * GOLDEN + seed):
// with probability far_permille / 1000 a column anywhere, otherwise one within
// `band` columns of the diagonal; clipped into range, sorted ascending within
// the row (repeats stay: legal CSR, the reference loop adds them).  Values
// uniform in [-1, 1) from a second draw, assigned by position after the sort.
// ---------------------------------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t z) // splitmix64 finaliser
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

constexpr int kMaxPerRow = 32;

__global__ __launch_bounds__(kBlock) void unstructured_fill_kernel(
    int64_t nrows, int per_row, int64_t band, int far_permille, uint64_t seed,
    int32_t* __restrict__ rowptr, int32_t* __restrict__ colind,
    double* __restrict__ values)
{
  constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrows;
       i += (int64_t)gridDim.x * blockDim.x) {
    rowptr[i] = (int32_t)(i * per_row);
    if (i == nrows)
      break;
    int32_t c[kMaxPerRow];
    for (int k = 0; k < per_row; ++k) {
      const uint64_t e = (uint64_t)(i * per_row + k);
      const uint64_t h = mix64((e + 1) * kGolden + seed);
      const bool far = (int)(h % 1000u) < far_permille;
      const uint64_t r = h >> 10;
      int64_t col = far ? (int64_t)(r % (uint64_t)nrows)
                        : i - band + (int64_t)(r % (uint64_t)(2 * band + 1));
      col = col < 0 ? 0 : (col >= nrows ? nrows - 1 : col);
      // insertion into the sorted prefix
      int j = k;
      while (j > 0 && c[j - 1] > (int32_t)col) {
        c[j] = c[j - 1];
        --j;
      }
      c[j] = (int32_t)col;
    }
    for (int k = 0; k < per_row; ++k) {
      const uint64_t e = (uint64_t)(i * per_row + k);
      colind[e] = c[k];
      const uint64_t h2 = mix64((e + 1) * kGolden + seed + 1);
      values[e] = (double)(h2 >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
  }
}

int make_geom(int32_t n, int64_t r0, int64_t r1, Geom* g, int points = 7)
{
  if (n < 1 || n > 1290) // n^3 rows must fit the int32 local index space
    return SPMV_HIP_ERANGE;
  if (points != 7 && points != 27)
    return SPMV_HIP_EINVAL;
  const int64_t n2 = (int64_t)n * n, N = n2 * n;
  if (r0 < 0 || r1 < r0 || r1 > N)
    return SPMV_HIP_EINVAL;
  // Closed-form ghost numbering needs the ghost sets to be whole runs:
  // true when the block is the whole matrix or holds >= n^2 rows.
  const bool whole = (r0 == 0 && r1 == N);
  if (!whole && (r1 - r0) < n2)
    return SPMV_HIP_ENOTSUP;
  // 27 points: the index order of a row's neighbours needs n >= 3, and the
  // ghost sets are whole planes only for plane-aligned slabs
  if (points == 27 && (n < 3 || (!whole && (r0 % n2 != 0 || r1 % n2 != 0))))
    return SPMV_HIP_ENOTSUP;
  g->n = n;
  g->points = points;
  g->n2 = n2;
  g->N = N;
  g->r0 = r0;
  g->r1 = r1;
  g->gb = r0 < n2 ? r0 : n2;
  g->gb_start = r0 - g->gb;
  g->ga = (N - r1) < n2 ? (N - r1) : n2;
  if ((r1 - r0) + g->gb + g->ga > INT32_MAX)
    return SPMV_HIP_ERANGE;
  return SPMV_HIP_OK;
}

} // namespace

extern "C" {

int spmv_hip_poisson3d_ghosts(int32_t n, int64_t row_begin, int64_t row_end,
                              int64_t* ghosts_below, int64_t* ghosts_above)
{
  Geom g;
  int rc = make_geom(n, row_begin, row_end, &g);
  if (rc)
    return rc;
  if (ghosts_below)
    *ghosts_below = g.gb;
  if (ghosts_above)
    *ghosts_above = g.ga;
  return SPMV_HIP_OK;
}

int spmv_hip_poisson3d_count(spmv_hip_ctx* ctx, int32_t n, int64_t row_begin,
                             int64_t row_end, int part, int32_t* rowptr,
                             int64_t* host_nnz, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(rowptr && part >= SPMV_HIP_PART_ALL
               && part <= SPMV_HIP_PART_LOCAL_LOWER);
  Geom g;
  int rc = make_geom(n, row_begin, row_end, &g, ctx->poisson_stencil);
  if (rc)
    return rc;
  const int64_t nrows = g.r1 - g.r0;
  if ((int64_t)g.points * nrows > INT32_MAX)
    return SPMV_HIP_ERANGE; // int32 rowptr (csr_kernels.h:28)
  hipStream_t st = spmv_stream(ctx, stream);
  const int grid = spmv_grid_for(ctx, nrows + 1, kBlock);
  hipLaunchKernelGGL(poisson_count_kernel, dim3(grid), dim3(kBlock), 0, st, g,
                     part, rowptr);
  SPMV_CHECK_LAUNCH();
  // in-place exclusive scan of nrows+1 counts -> row pointer
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  SPMV_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(
      nullptr, tmp_bytes, rowptr, rowptr, (int)(nrows + 1), st));
  SPMV_CHECK_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, rowptr,
                                                  rowptr, (int)(nrows + 1), st);
  int32_t total = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&total, rowptr + nrows, sizeof(int32_t),
                       hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  if (e != hipSuccess)
    return static_cast<int>(e);
  if (host_nnz)
    *host_nnz = total;
  return SPMV_HIP_OK;
}

int spmv_hip_poisson3d_fill_f64(spmv_hip_ctx* ctx, int32_t n,
                                int64_t row_begin, int64_t row_end, int part,
                                const int32_t* rowptr, int32_t* colind,
                                double* values, double* diagonal, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(rowptr && part >= SPMV_HIP_PART_ALL
               && part <= SPMV_HIP_PART_LOCAL_LOWER);
  Geom g;
  int rc = make_geom(n, row_begin, row_end, &g, ctx->poisson_stencil);
  if (rc)
    return rc;
  const int64_t nrows = g.r1 - g.r0;
  if (nrows == 0)
    return SPMV_HIP_OK;
  // colind/values may be NULL only if the caller knows the part is empty
  // (e.g. REMOTE on one rank); the kernel would then write nothing, but a
  // NULL with entries to write must never reach the device.
  if (!colind || !values) {
    int32_t total = 0;
    SPMV_CHECK_HIP(hipMemcpy(&total, rowptr + nrows, sizeof(int32_t),
                             hipMemcpyDeviceToHost));
    SPMV_REQUIRE(total == 0);
    if (!diagonal)
      return SPMV_HIP_OK;
  }
  const int grid = spmv_grid_for(ctx, nrows, kBlock);
  hipLaunchKernelGGL(poisson_fill_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), g, part, rowptr, colind, values,
                     diagonal, 1e-6 * (double)ctx->poisson_skew_ppm);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_unstructured_fill_f64(spmv_hip_ctx* ctx, int64_t num_rows,
                                   int per_row, int64_t band, int far_permille,
                                   uint64_t seed, int32_t* rowptr,
                                   int32_t* colind, double* values, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_rows >= 1 && per_row >= 1 && per_row <= kMaxPerRow
               && band >= 0 && far_permille >= 0 && far_permille <= 1000
               && rowptr && colind && values);
  if (num_rows > INT32_MAX || num_rows * per_row > INT32_MAX)
    return SPMV_HIP_ERANGE; // int32 rowptr (csr_kernels.h:28)
  const int grid = spmv_grid_for(ctx, num_rows + 1, kBlock);
  hipLaunchKernelGGL(unstructured_fill_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), num_rows, per_row, band,
                     far_permille, seed, rowptr, colind, values);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // extern "C"
