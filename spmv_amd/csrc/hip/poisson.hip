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

// ---------------------------------------------------------------------------
// Seeded FEM-LIKE test matrix: ragged rows (and optionally a tail of very long
// ones) in a bandwidth-reducing order -- the shape of an unstructured 3-D mesh
// matrix after RCM: a row's columns sit in its own level set and the two
// adjacent ones.  Not in the reference; a synthetic input for the general
// kernels on what the PETSc reader delivers (read_petsc.cpp:40-228).  The
// numpy twin (spmv_amd/poisson.py:fem_like_csr) follows this text line by line.
//
//   h_row(i) = mix64((i + 1) G + seed);  u = bits 16..31 of h_row
//   tail row  (tail_permille > 0 and h_row mod 1000 < tail_permille):
//     n = tail_min + (h_row >> 32) mod (tail_max - tail_min + 1) entries in ONE
//     window of n * tail_stride columns centred on the row (moved inside the
//     matrix), one entry per stride-wide slot
//   short row: n = min_len + (span * ((u^3 + 2^16 u^2) / 2) >> 48),
//     span = max_len - min_len + 1 (skewed to the short side: mean = min_len +
//     0.29 span); 3n/10 entries around row - layer, 3n/10 around row + layer,
//     the rest around the row itself, each cluster in a window of 2 * jitter
//     columns, one entry per slot of (2 jitter / cluster entries) columns; a
//     side cluster is dropped when its window leaves the matrix or when the
//     centre window had to be moved inside the matrix
//   entry e (global position) in slot k of a window starting at lo, stride s:
//     column lo + k s + mix64((e + 1) G + seed + 1) mod s -- strictly ascending,
//     no repeats; the slot that holds the row's own column gets exactly it
//   values: diagonal = entries of the row + 1, others uniform in [-1, 1) from
//     mix64((e + 1) G + seed + 2)
// ---------------------------------------------------------------------------
struct FemRow {
  int32_t n[3];   // entries per cluster (lower, centre, upper); tail: centre only
  int64_t lo[3];  // first column of each cluster's window
  int32_t s[3];   // slot width
  int32_t kdiag;  // slot of the centre cluster that holds the diagonal
};

__host__ __device__ inline FemRow fem_row(const spmv_hip_fem_params& p, int64_t i)
{
  constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
  const uint64_t h = mix64((uint64_t)(i + 1) * kGolden + p.seed);
  FemRow r;
  r.n[0] = r.n[2] = 0;
  r.lo[0] = r.lo[2] = 0;
  r.s[0] = r.s[2] = 1;
  const int64_t N = p.num_rows;
  if (p.tail_permille > 0 && (int)(h % 1000u) < p.tail_permille) {
    const int32_t n
        = p.tail_min + (int32_t)((h >> 32) % (uint64_t)(p.tail_max - p.tail_min + 1));
    const int64_t W = (int64_t)n * p.tail_stride;
    int64_t lo = i - W / 2;
    lo = lo < 0 ? 0 : (lo > N - W ? N - W : lo);
    r.n[1] = n;
    r.lo[1] = lo;
    r.s[1] = p.tail_stride;
  } else {
    const uint64_t u = (h >> 16) & 0xFFFFu;
    const uint64_t f = (u * u * u + (u * u << 16)) >> 1; // < 2^48
    const int32_t n = p.min_len
                      + (int32_t)(((uint64_t)(p.max_len - p.min_len + 1) * f) >> 48);
    const int64_t W = 2 * (int64_t)p.jitter;
    int32_t n0 = (3 * n) / 10, n2 = (3 * n) / 10;
    const int32_t n1 = n - n0 - n2;
    // (a centre window moved inside the matrix would reach into the far one)
    if (i - p.layer - p.jitter < 0 || i + p.jitter > N)
      n0 = 0;
    if (i + p.layer + p.jitter > N || i - p.jitter < 0)
      n2 = 0;
    int64_t lo1 = i - p.jitter;
    lo1 = lo1 < 0 ? 0 : (lo1 > N - W ? N - W : lo1);
    r.n[0] = n0;
    r.n[1] = n1;
    r.n[2] = n2;
    r.lo[0] = i - p.layer - p.jitter;
    r.lo[1] = lo1;
    r.lo[2] = i + p.layer - p.jitter;
    r.s[0] = n0 ? (int32_t)(W / n0) : 1;
    r.s[1] = (int32_t)(W / n1);
    r.s[2] = n2 ? (int32_t)(W / n2) : 1;
  }
  const int64_t kd = (i - r.lo[1]) / r.s[1];
  r.kdiag = (int32_t)(kd < r.n[1] - 1 ? kd : r.n[1] - 1);
  return r;
}

// symmetric storage from a general block (Matrix.cpp:337-349): entries below
// the diagonal per row ...
__global__ __launch_bounds__(kBlock) void lower_count_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int32_t* __restrict__ lower_rowptr)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    int32_t n = 0;
    if (i < num_rows)
      for (int32_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
        n += colind[e] < (int32_t)i ? 1 : 0;
    lower_rowptr[i] = n;
  }
}

// ... kept in their order (one wave per row: a prefix count per 64 entries),
// the diagonal entries summed in their order
__global__ __launch_bounds__(kBlock) void lower_fill_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const double* __restrict__ values,
    const int32_t* __restrict__ lower_rowptr, int32_t* __restrict__ lower_colind,
    double* __restrict__ lower_values, double* __restrict__ diagonal)
{
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t i = wid; i < num_rows; i += nw) {
    const int32_t a = rowptr[i], b = rowptr[i + 1];
    int32_t dst = lower_rowptr[i];
    double d = 0.0;
    for (int32_t e0 = a; e0 < b; e0 += 64) {
      const int32_t e = e0 + lane;
      const int32_t c = e < b ? colind[e] : INT32_MAX;
      const double v = e < b ? values[e] : 0.0;
      const bool low = c < (int32_t)i;
      const uint64_t m = __ballot(low);
      if (low) {
        const int32_t at = dst + __popcll(m & ((1ull << lane) - 1ull));
        lower_colind[at] = c;
        lower_values[at] = v;
      }
      dst += __popcll(m);
      // the diagonal entries of these 64, added in entry order
      uint64_t dm = __ballot(c == (int32_t)i);
      while (dm) {
        const int j = __ffsll((long long)dm) - 1;
        d += __shfl(v, j, 64);
        dm &= dm - 1;
      }
    }
    if (lane == 0)
      diagonal[i] = d;
  }
}

__global__ __launch_bounds__(kBlock) void fem_count_kernel(spmv_hip_fem_params p,
                                                           int32_t* __restrict__ rowptr)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i <= p.num_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t n = 0;
    if (i < p.num_rows) {
      const FemRow r = fem_row(p, i);
      n = r.n[0] + r.n[1] + r.n[2];
    }
    rowptr[i] = n;
  }
}

// one lane per ENTRY (rows differ in length by three orders of magnitude): the
// row of entry e is found by bisection of the row pointer
__global__ __launch_bounds__(kBlock) void fem_fill_kernel(
    spmv_hip_fem_params p, int64_t nnz, const int32_t* __restrict__ rowptr,
    int32_t* __restrict__ colind, double* __restrict__ values)
{
  constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = p.num_rows; // last row with rowptr[row] <= e
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)rowptr[mid] <= e)
        lo = mid;
      else
        hi = mid;
    }
    const int64_t i = lo;
    const FemRow r = fem_row(p, i);
    int32_t k = (int32_t)(e - rowptr[i]);
    int c = 0;
    while (k >= r.n[c]) {
      k -= r.n[c];
      ++c;
    }
    const uint64_t h1 = mix64((uint64_t)(e + 1) * kGolden + p.seed + 1);
    const uint64_t h2 = mix64((uint64_t)(e + 1) * kGolden + p.seed + 2);
    int64_t col = r.lo[c] + (int64_t)k * r.s[c] + (int64_t)(h1 % (uint64_t)r.s[c]);
    double v = (double)(h2 >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    if (c == 1 && k == r.kdiag) {
      col = i;
      v = (double)(r.n[0] + r.n[1] + r.n[2]) + 1.0;
    }
    colind[e] = (int32_t)col;
    values[e] = v;
  }
}

struct ToI64 {
  __host__ __device__ int64_t operator()(int32_t v) const { return v; }
};

int fem_params_ok(const spmv_hip_fem_params& p)
{
  SPMV_REQUIRE(p.num_rows >= 1 && p.num_rows <= INT32_MAX);
  SPMV_REQUIRE(p.min_len >= 1 && p.max_len >= p.min_len && p.max_len <= 1024);
  SPMV_REQUIRE(p.jitter >= 1 && p.layer >= 2 * p.jitter
               && 2 * (int64_t)p.jitter <= p.num_rows);
  // every cluster gets at least one column per entry
  SPMV_REQUIRE(p.max_len <= 2 * p.jitter);
  SPMV_REQUIRE(p.tail_permille >= 0 && p.tail_permille <= 1000);
  if (p.tail_permille > 0)
    SPMV_REQUIRE(p.tail_min >= 1 && p.tail_max >= p.tail_min
                 && p.tail_stride >= 1
                 && (int64_t)p.tail_max * p.tail_stride <= p.num_rows);
  return SPMV_HIP_OK;
}

// ---------------------------------------------------------------------------
// The 7-point matrix on ONE BOX of a 3-D block partition (SURVEY 8f n4;
// Matrix::create_poisson3d_boxes): rows = the box's points, x fastest, in the
// rank-major global numbering (a rank's points are consecutive), so owned
// columns are `row + {-lx ly, -lx, -1, 0, 1, lx, lx ly}` and the ghost columns
// are the points one step outside the six faces.  Sorted by global id the
// ghosts come face by face -- the neighbour ranks ascend in the order -z, -y,
// -x, +x, +y, +z -- and inside a face in the neighbour's own local order,
// which is the order of the two in-face coordinates: closed-form local column
// numbers, no search.  Entry order within a row: owned columns ascending,
// then ghost columns ascending, as create_matrix leaves them.
// ---------------------------------------------------------------------------
struct BoxGeom {
  int32_t n;             // global grid
  int32_t f[3], l[3];    // first point and extents of the box
  int32_t has[6];        // a neighbour box behind face -z -y -x +x +y +z
  int64_t ghost_base[6]; // first ghost (0-based) of that face
  int64_t nloc, nghost;
};

__host__ __device__ inline int box_entries(const BoxGeom& g, int64_t k, int part,
                                           int32_t* col, double* val, double skew)
{
  const int64_t lx = g.l[0], ly = g.l[1];
  const int64_t x = k % lx, y = (k / lx) % ly, z = k / (lx * ly);
  // stencil order = ascending column within each group (owned / ghost)
  const int64_t delta[7] = {-lx * ly, -lx, -1, 0, 1, lx, lx * ly};
  const bool inside[7] = {z > 0, y > 0, x > 0, true, x < lx - 1, y < ly - 1,
                          z < g.l[2] - 1};
  const int face[7] = {0, 1, 2, -1, 3, 4, 5};
  // in-face index of the point behind each face
  const int64_t inface[6] = {x + lx * y, x + lx * z, y + ly * z,
                             y + ly * z, x + lx * z, x + lx * y};
  int c = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int e = 0; e < 7; ++e) {
      const bool owned = inside[e];
      if (owned != (pass == 0))
        continue;
      if (!owned && !g.has[face[e]])
        continue; // outside the global grid: dropped
      const bool lower = e < 3;
      bool keep_it;
      switch (part) {
      case SPMV_HIP_PART_ALL: keep_it = true; break;
      case SPMV_HIP_PART_LOCAL: keep_it = owned; break;
      case SPMV_HIP_PART_REMOTE: keep_it = !owned; break;
      default: keep_it = owned && lower; break; // LOCAL_LOWER
      }
      if (!keep_it)
        continue;
      if (col) {
        col[c] = owned ? (int32_t)(k + delta[e])
                       : (int32_t)(g.nloc + g.ghost_base[face[e]]
                                   + inface[face[e]]);
        val[c] = e == 3 ? 6.0 : (lower ? -1.0 - skew : -1.0 + skew);
      }
      ++c;
    }
  return c;
}

__global__ __launch_bounds__(kBlock) void box_count_kernel(BoxGeom g, int part,
                                                           int32_t* counts)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= g.nloc;
       k += (int64_t)gridDim.x * blockDim.x)
    counts[k] = k < g.nloc ? box_entries(g, k, part, nullptr, nullptr, 0.0) : 0;
}

__global__ __launch_bounds__(kBlock) void box_fill_kernel(
    BoxGeom g, int part, const int32_t* __restrict__ rowptr,
    int32_t* __restrict__ colind, double* __restrict__ values,
    double* __restrict__ diagonal, double skew)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < g.nloc;
       k += (int64_t)gridDim.x * blockDim.x) {
    int32_t col[7];
    double val[7];
    const int c = box_entries(g, k, part, col, val, skew);
    const int64_t pos = rowptr[k];
    for (int e = 0; e < c; ++e) {
      colind[pos + e] = col[e];
      values[pos + e] = val[e];
    }
    if (diagonal)
      diagonal[k] = 6.0;
  }
}

int make_box(int32_t n, const int32_t first[3], const int32_t len[3], BoxGeom* g)
{
  if (n < 1 || !first || !len)
    return SPMV_HIP_EINVAL;
  for (int a = 0; a < 3; ++a) {
    if (first[a] < 0 || len[a] < 1 || (int64_t)first[a] + len[a] > n)
      return SPMV_HIP_EINVAL;
    g->f[a] = first[a];
    g->l[a] = len[a];
  }
  g->n = n;
  g->nloc = (int64_t)len[0] * len[1] * len[2];
  const int64_t area[6] = {(int64_t)len[0] * len[1], (int64_t)len[0] * len[2],
                           (int64_t)len[1] * len[2], (int64_t)len[1] * len[2],
                           (int64_t)len[0] * len[2], (int64_t)len[0] * len[1]};
  const int axis[6] = {2, 1, 0, 0, 1, 2};
  int64_t base = 0;
  for (int s = 0; s < 6; ++s) {
    const int a = axis[s];
    g->has[s] = s < 3 ? first[a] > 0 : first[a] + len[a] < n;
    g->ghost_base[s] = base;
    if (g->has[s])
      base += area[s];
  }
  g->nghost = base;
  if (g->nloc + g->nghost > INT32_MAX || 7 * g->nloc > INT32_MAX)
    return SPMV_HIP_ERANGE; // int32 columns and row pointer (csr_kernels.h:28)
  return SPMV_HIP_OK;
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

int spmv_hip_fem_count(spmv_hip_ctx* ctx, const spmv_hip_fem_params* params,
                       int32_t* rowptr, int64_t* host_nnz, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(params && rowptr && host_nnz);
  const spmv_hip_fem_params p = *params;
  int rc = fem_params_ok(p);
  if (rc)
    return rc;
  hipStream_t st = spmv_stream(ctx, stream);
  const int grid = spmv_grid_for(ctx, p.num_rows + 1, kBlock);
  hipLaunchKernelGGL(fem_count_kernel, dim3(grid), dim3(kBlock), 0, st, p,
                     rowptr);
  SPMV_CHECK_LAUNCH();
  // the lengths are summed in 64 bits first: the row pointer of the format is
  // int32 (csr_kernels.h:28) and a tail of long rows can exceed it
  int64_t* d_total = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0, tb2 = 0;
  const int n1 = (int)(p.num_rows + 1);
  SPMV_CHECK_HIP(hipMalloc(&d_total, sizeof(int64_t)));
  hipError_t e = hipcub::DeviceReduce::Sum(
      nullptr, tmp_bytes,
      hipcub::TransformInputIterator<int64_t, ToI64, const int32_t*>(rowptr,
                                                                     ToI64()),
      d_total, n1, st);
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, rowptr, rowptr, n1, st);
  if (tb2 > tmp_bytes)
    tmp_bytes = tb2;
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(
        tmp, tmp_bytes,
        hipcub::TransformInputIterator<int64_t, ToI64, const int32_t*>(rowptr,
                                                                       ToI64()),
        d_total, n1, st);
  int64_t total = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  if (e == hipSuccess && total <= INT32_MAX) {
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, rowptr, rowptr, n1, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
  }
  (void)hipFree(tmp);
  (void)hipFree(d_total);
  if (e != hipSuccess)
    return static_cast<int>(e);
  *host_nnz = total;
  return total > INT32_MAX ? SPMV_HIP_ERANGE : SPMV_HIP_OK;
}

int spmv_hip_fem_fill_f64(spmv_hip_ctx* ctx, const spmv_hip_fem_params* params,
                          int64_t num_non_zeros, const int32_t* rowptr,
                          int32_t* colind, double* values, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(params && rowptr && colind && values && num_non_zeros >= 1
               && num_non_zeros <= INT32_MAX);
  const spmv_hip_fem_params p = *params;
  int rc = fem_params_ok(p);
  if (rc)
    return rc;
  const int grid = spmv_grid_for(ctx, num_non_zeros, kBlock);
  hipLaunchKernelGGL(fem_fill_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), p, num_non_zeros, rowptr, colind,
                     values);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_csr_lower_split_count(spmv_hip_ctx* ctx, int32_t num_rows,
                                   const int32_t* rowptr, const int32_t* colind,
                                   int32_t* lower_rowptr, int64_t* host_nnz,
                                   void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_rows >= 0 && rowptr && lower_rowptr && host_nnz);
  hipStream_t st = spmv_stream(ctx, stream);
  const int n1 = num_rows + 1;
  hipLaunchKernelGGL(lower_count_kernel, dim3(spmv_grid_for(ctx, n1, kBlock)),
                     dim3(kBlock), 0, st, num_rows, rowptr, colind, lower_rowptr);
  SPMV_CHECK_LAUNCH();
  void* tmp = nullptr;
  size_t tb = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, lower_rowptr,
                                                  lower_rowptr, n1, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, lower_rowptr, lower_rowptr, n1, st);
  int32_t total = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&total, lower_rowptr + num_rows, sizeof(int32_t),
                       hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  if (e != hipSuccess)
    return static_cast<int>(e);
  *host_nnz = total; // (a subset of an int32-indexed block)
  return SPMV_HIP_OK;
}

int spmv_hip_csr_lower_split_fill_f64(spmv_hip_ctx* ctx, int32_t num_rows,
                                      const int32_t* rowptr, const int32_t* colind,
                                      const double* values,
                                      const int32_t* lower_rowptr,
                                      int32_t* lower_colind, double* lower_values,
                                      double* diagonal, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_rows >= 0 && rowptr && lower_rowptr && diagonal);
  if (num_rows == 0)
    return SPMV_HIP_OK;
  hipLaunchKernelGGL(lower_fill_kernel, dim3(spmv_grid_for(ctx, num_rows, kBlock / 64)),
                     dim3(kBlock), 0, spmv_stream(ctx, stream), num_rows, rowptr,
                     colind, values, lower_rowptr, lower_colind, lower_values,
                     diagonal);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_poisson3d_box_count(spmv_hip_ctx* ctx, int32_t n,
                                 const int32_t first[3], const int32_t len[3],
                                 int part, int32_t* rowptr, int64_t* host_nnz,
                                 int64_t* num_ghosts, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(part >= SPMV_HIP_PART_ALL && part <= SPMV_HIP_PART_LOCAL_LOWER);
  if (ctx->poisson_stencil != 7) // the box generator knows the 7-point matrix only
    return SPMV_HIP_ENOTSUP;
  BoxGeom g;
  int rc = make_box(n, first, len, &g);
  if (rc)
    return rc;
  if (num_ghosts)
    *num_ghosts = g.nghost;
  if (!rowptr) // geometry query only
    return SPMV_HIP_OK;
  hipStream_t st = spmv_stream(ctx, stream);
  const int grid = spmv_grid_for(ctx, g.nloc + 1, kBlock);
  hipLaunchKernelGGL(box_count_kernel, dim3(grid), dim3(kBlock), 0, st, g, part,
                     rowptr);
  SPMV_CHECK_LAUNCH();
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  SPMV_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(
      nullptr, tmp_bytes, rowptr, rowptr, (int)(g.nloc + 1), st));
  SPMV_CHECK_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, rowptr, rowptr,
                                                  (int)(g.nloc + 1), st);
  int32_t total = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&total, rowptr + g.nloc, sizeof(int32_t),
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

int spmv_hip_poisson3d_box_fill_f64(spmv_hip_ctx* ctx, int32_t n,
                                    const int32_t first[3], const int32_t len[3],
                                    int part, const int32_t* rowptr,
                                    int32_t* colind, double* values,
                                    double* diagonal, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(rowptr && part >= SPMV_HIP_PART_ALL
               && part <= SPMV_HIP_PART_LOCAL_LOWER);
  if (ctx->poisson_stencil != 7)
    return SPMV_HIP_ENOTSUP;
  BoxGeom g;
  int rc = make_box(n, first, len, &g);
  if (rc)
    return rc;
  if (!colind || !values) { // only if the part is empty (REMOTE on one rank)
    int32_t total = 0;
    SPMV_CHECK_HIP(hipMemcpy(&total, rowptr + g.nloc, sizeof(int32_t),
                             hipMemcpyDeviceToHost));
    SPMV_REQUIRE(total == 0);
    if (!diagonal)
      return SPMV_HIP_OK;
  }
  const int grid = spmv_grid_for(ctx, g.nloc, kBlock);
  hipLaunchKernelGGL(box_fill_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), g, part, rowptr, colind, values,
                     diagonal, 1e-6 * (double)ctx->poisson_skew_ppm);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // extern "C"
