// Matrix<T>: see matrix.h.
#include "matrix.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>
#include <type_traits>

#include "spmv_hip.h"

namespace spmv
{

std::vector<int64_t> owner_ranges(int size, int64_t N)
{
  const int64_t q = N / size, r = N % size;
  std::vector<int64_t> ranges(size + 1);
  for (int k = 0; k <= size; ++k)
    ranges[k] = (k < r) ? k * (q + 1) : k * q + r;
  return ranges;
}

// ---------------------------------------------------------------------------
// constructors
// ---------------------------------------------------------------------------
template <typename T>
Matrix<T>::Matrix(const CsrHost<T>& mat, std::shared_ptr<L2GMap> col_map,
                  std::shared_ptr<L2GMap> row_map,
                  std::shared_ptr<DeviceExecutor> exec)
    : _exec(exec), _col_map(col_map), _row_map(row_map), _nnz(mat.non_zeros())
{
  if (col_map->overlapping()) // Matrix.cpp:28-29
    throw std::runtime_error("Ovelapping not supported in this format!");
  _mat_local.reset(new CSRMatrix<T>(exec, &mat));
}

template <typename T>
Matrix<T>::Matrix(const CsrHost<T>& mat_local, const CsrHost<T>& mat_remote,
                  std::shared_ptr<L2GMap> col_map,
                  std::shared_ptr<L2GMap> row_map,
                  std::shared_ptr<DeviceExecutor> exec)
    : _exec(exec), _col_map(col_map), _row_map(row_map),
      _nnz(mat_local.non_zeros() + mat_remote.non_zeros())
{
  if (!col_map->overlapping()) // Matrix.cpp:44-45
    throw std::runtime_error("Ovelapping not enabled in column mapping!");
  _mat_local.reset(new CSRMatrix<T>(exec, &mat_local));
  _mat_remote.reset(new CSRMatrix<T>(exec, &mat_remote));
}

template <typename T>
Matrix<T>::Matrix(const CsrHost<T>& mat_local, const CsrHost<T>& mat_remote,
                  const std::vector<T>& mat_diagonal,
                  std::shared_ptr<L2GMap> col_map,
                  std::shared_ptr<L2GMap> row_map, int64_t nnz_full,
                  std::shared_ptr<DeviceExecutor> exec)
    : _exec(exec), _col_map(col_map), _row_map(row_map), _nnz(nnz_full),
      _symmetric(true)
{
  if (exec->get_device_type() == DeviceType::undefined) // Matrix.cpp:68-70
    throw std::runtime_error("Device type not set!");
  _mat_local.reset(new CSRMatrix<T>(exec, &mat_local, &mat_diagonal, true));
  _mat_remote.reset(new CSRMatrix<T>(exec, &mat_remote));
}

// ---------------------------------------------------------------------------
// queries (Matrix.cpp:75-128)
// ---------------------------------------------------------------------------
template <typename T>
int Matrix<T>::rows() const
{
  if (_mat_local)
    return _mat_local->rows();
  return _mat_remote ? _mat_remote->rows() : 0;
}

template <typename T>
int Matrix<T>::cols() const
{
  if (_mat_local)
    return _mat_local->cols();
  return _mat_remote ? _mat_remote->cols() : 0;
}

template <typename T>
int64_t Matrix<T>::non_zeros() const
{
  if (_symmetric)
    return _nnz;
  if (_col_map->overlapping())
    return _mat_local->non_zeros() + _mat_remote->non_zeros();
  return _mat_local->non_zeros();
}

template <typename T>
size_t Matrix<T>::format_size() const
{
  size_t bytes = sizeof(int) * _mat_local->rows()
                 + (sizeof(int) + sizeof(T)) * _mat_local->non_zeros();
  if (_symmetric || _col_map->overlapping())
    bytes += sizeof(int) * _mat_remote->rows()
             + (sizeof(int) + sizeof(T)) * _mat_remote->non_zeros();
  if (_symmetric)
    bytes += sizeof(T) * _mat_local->rows();
  return bytes;
}

// ---------------------------------------------------------------------------
// SpMV (Matrix.cpp:131-141, 483-552)
// ---------------------------------------------------------------------------
template <typename T>
void Matrix<T>::mult(T* x, T* y) const
{
  if (_symmetric && _col_map->overlapping())
    spmv_sym_overlap(x, y);
  else if (_symmetric)
    spmv_sym(x, y);
  else if (_col_map->overlapping())
    spmv_overlap(x, y);
  else
    spmv(x, y);
}

template <typename T>
void Matrix<T>::spmv(T* x, T* y) const
{
  _mat_local->mult(1, x, 0, y);
}

template <typename T>
void Matrix<T>::spmv_overlap(T* x, T* y) const
{
  _mat_local->mult(1, x, 0, y);  // runs while the halo is in flight
  _col_map->update_finalise(x);  // compute stream waits for the halo event
  if (_mat_local->non_zeros() > 0)
    _mat_remote->mult(1, x, 1, y);
  else
    _mat_remote->mult(1, x, 0, y);
}

template <typename T>
void Matrix<T>::spmv_sym(T* x, T* y) const
{
  _mat_local->mult(1, x, 0, y);
  _mat_remote->mult(1, x, 1, y);
}

template <typename T>
void Matrix<T>::spmv_sym_overlap(T* x, T* y) const
{
  _mat_local->mult(1, x, 0, y);
  _col_map->update_finalise(x);
  _mat_remote->mult(1, x, 1, y);
}

template <typename T>
bool Matrix<T>::enable_mixed() const
{
  if (_symmetric || !std::is_same<T, double>::value)
    return false;
  // (a block that gave its CSR values back -- CSRMatrix::release_csr -- has
  // nothing to convert: no mixed mode for this matrix)
  if ((_mat_local && _mat_local->csr_released())
      || (_mat_remote && _mat_remote->csr_released()))
    return false;
  if (_mat_local)
    _mat_local->enable_mixed();
  if (_mat_remote)
    _mat_remote->enable_mixed();
  return true;
}

template <typename T>
void Matrix<T>::use_mixed(bool on) const
{
  if (_mat_local)
    _mat_local->use_mixed(on);
  if (_mat_remote)
    _mat_remote->use_mixed(on);
}

template <typename T>
bool Matrix<T>::mult_dot(T* x, T* y, double* dot_local, double* dot_remote,
                         void* ev_local_done) const
{
  double* const tl = dot_local;
  double* const tr = dot_remote;
  auto* hip = dynamic_cast<HipExecutor*>(_exec.get());
  auto mark = [&] {
    if (ev_local_done && hip)
      hip->record_event(ev_local_done, hip->get_stream());
  };
  if (!std::is_same<T, double>::value) {
    mult(x, y);
    mark();
    return false;
  }
  if (_symmetric) {
    // same sequence as spmv_sym / spmv_sym_overlap.  The symmetric kernel
    // carries its share of x.(A x) through the mirror identity (partials
    // only), the remote block adds its own share.
    const bool ok = _mat_local->mult_dot(1, x, 0, y, tl);
    if (!ok)
      _mat_local->mult(1, x, 0, y);
    mark();
    if (_col_map->overlapping())
      _col_map->update_finalise(x);
    if (_mat_remote) {
      if (!(ok && _mat_remote->mult_dot(1, x, 1, y, tr)))
        _mat_remote->mult(1, x, 1, y);
      // a remote block that could not fuse has no entries => zero share
    }
    return ok;
  }
  if (!_col_map->overlapping()) {
    const bool ok = _mat_local->mult_dot(1, x, 0, y, tl);
    if (!ok)
      mult(x, y);
    mark();
    return ok;
  }
  // overlapping: local share, halo wait, remote share (same order as
  // spmv_overlap).  An empty local block cannot fuse -> plain path.
  if (!_mat_local->mult_dot(1, x, 0, y, tl)) {
    mult(x, y);
    mark();
    return false;
  }
  mark();
  _col_map->update_finalise(x);
  // a remote block without entries has a zero share and nothing to add to y
  if (_mat_remote->non_zeros() > 0 && !_mat_remote->mult_dot(1, x, 1, y, tr))
    throw std::runtime_error("Matrix::mult_dot: the remote block could not "
                             "run on this executor");
  return true;
}

// ---------------------------------------------------------------------------
// create_matrix (Matrix.cpp:164-480, row_ghosts empty)
// ---------------------------------------------------------------------------
namespace
{

template <typename T>
struct Triplet {
  int32_t row, col;
  T val;
};

// Eigen::setFromTriplets: per row ascending columns, duplicates summed.
template <typename T>
CsrHost<T> assemble(int32_t nrows, int32_t ncols, std::vector<Triplet<T>>& tr)
{
  std::stable_sort(tr.begin(), tr.end(),
                   [](const Triplet<T>& a, const Triplet<T>& b) {
                     return a.row != b.row ? a.row < b.row : a.col < b.col;
                   });
  CsrHost<T> m;
  m.rows = nrows;
  m.cols = ncols;
  m.rowptr.assign(nrows + 1, 0);
  for (size_t k = 0; k < tr.size(); ++k) {
    if (k > 0 && tr[k].row == tr[k - 1].row && tr[k].col == tr[k - 1].col) {
      m.values.back() += tr[k].val;
      continue;
    }
    m.colind.push_back(tr[k].col);
    m.values.push_back(tr[k].val);
    ++m.rowptr[tr[k].row + 1];
  }
  std::partial_sum(m.rowptr.begin(), m.rowptr.end(), m.rowptr.begin());
  return m;
}

bool nonblocking(CommunicationModel cm)
{
  return cm == CommunicationModel::p2p_nonblocking
         || cm == CommunicationModel::collective_nonblocking;
}

} // namespace

template <typename T>
typename Matrix<T>::Split Matrix<T>::split_rows(
    const int32_t* rowptr, const int32_t* colind, const T* values,
    int64_t nrows_local, int64_t ncols_local, int64_t global_row_offset,
    int64_t global_col_offset, const std::vector<int64_t>& col_ghosts,
    bool symmetric, CommunicationModel cm, const std::vector<ExtraEntry>* extra)
{
  Split out;
  // ghost columns renumbered in ascending global order after the owned
  // columns (Matrix.cpp:295-318); received rows may bring new ones (:299-313)
  out.col_ghosts = col_ghosts;
  if (extra)
    for (const ExtraEntry& e : *extra)
      if (e.global_col < global_col_offset
          || e.global_col >= global_col_offset + ncols_local)
        out.col_ghosts.push_back(e.global_col);
  std::sort(out.col_ghosts.begin(), out.col_ghosts.end());
  out.col_ghosts.erase(
      std::unique(out.col_ghosts.begin(), out.col_ghosts.end()),
      out.col_ghosts.end());
  const std::vector<int64_t>& new_ghosts = out.col_ghosts;
  const int32_t ncols_all = static_cast<int32_t>(ncols_local + new_ghosts.size());

  std::vector<Triplet<T>> loc, rem;
  std::vector<char> has_diag;
  if (symmetric) {
    out.diagonal.assign(nrows_local, T(0)); // Matrix.cpp:429
    has_diag.assign(nrows_local, 0);
  }
  // one rule for local and received entries (Matrix.cpp:337-358, 388-407)
  auto place = [&](int32_t row, int64_t col, T val) {
    const Triplet<T> t{row, static_cast<int32_t>(col), val};
    if (symmetric) { // Matrix.cpp:337-349
      if (col < ncols_local) {
        const int64_t grow = row + global_row_offset;
        const int64_t gcol = col + global_col_offset;
        if (grow > gcol)
          loc.push_back(t);
        else if (grow == gcol) {
          out.diagonal[row] += val;
          has_diag[row] = 1;
        } // strictly-upper entries are implied by symmetry and dropped
      } else {
        rem.push_back(t);
      }
    } else if (nonblocking(cm)) { // Matrix.cpp:350-355
      (col < ncols_local ? loc : rem).push_back(t);
    } else { // Matrix.cpp:357
      loc.push_back(t);
    }
  };
  for (int64_t row = 0; row < nrows_local; ++row) {
    for (int32_t j = rowptr[row]; j < rowptr[row + 1]; ++j) {
      int64_t col = colind[j];
      if (col >= ncols_local) {
        const size_t g = static_cast<size_t>(col - ncols_local);
        if (g >= col_ghosts.size())
          throw std::runtime_error("create_matrix: ghost column out of range");
        col = ncols_local
              + (std::lower_bound(new_ghosts.begin(), new_ghosts.end(),
                                  col_ghosts[g])
                 - new_ghosts.begin());
      } else if (col < 0) {
        throw std::runtime_error("create_matrix: negative column index");
      }
      place(static_cast<int32_t>(row), col, values[j]);
    }
  }
  if (extra) // received ghost rows, after the local entries (Matrix.cpp:363-408)
    for (const ExtraEntry& e : *extra) {
      int64_t col;
      if (e.global_col >= global_col_offset
          && e.global_col < global_col_offset + ncols_local)
        col = e.global_col - global_col_offset;
      else
        col = ncols_local
              + (std::lower_bound(new_ghosts.begin(), new_ghosts.end(),
                                  e.global_col)
                 - new_ghosts.begin());
      if (e.row < 0 || e.row >= nrows_local)
        throw std::runtime_error("create_matrix: received row out of range");
      place(e.row, col, e.val);
    }
  const int32_t nrows = static_cast<int32_t>(nrows_local);
  if (symmetric) { // Matrix.cpp:415-446
    out.local = assemble(nrows, ncols_all, loc);
    out.remote = assemble(nrows, ncols_all, rem);
    out.two_blocks = true;
    out.nnz_full = 2 * out.local.non_zeros() + out.remote.non_zeros()
                   + std::count(has_diag.begin(), has_diag.end(), 1);
  } else if (nonblocking(cm)) { // Matrix.cpp:447-466
    out.local = assemble(nrows, static_cast<int32_t>(ncols_local), loc);
    out.remote = assemble(nrows, ncols_all, rem);
    out.two_blocks = true;
    out.nnz_full = out.local.non_zeros() + out.remote.non_zeros();
  } else { // Matrix.cpp:467-479
    out.local = assemble(nrows, ncols_all, loc);
    out.remote.rows = nrows;
    out.remote.cols = ncols_all;
    out.remote.rowptr.assign(nrows + 1, 0);
    out.nnz_full = out.local.non_zeros();
  }
  return out;
}

template <typename T>
typename Matrix<T>::Split Matrix<T>::split_rows_distributed(
    const Comm& comm, const int32_t* rowptr, const int32_t* colind,
    const T* values, int64_t nrows_local, int64_t ncols_local,
    const std::vector<int64_t>& row_ghosts,
    const std::vector<int64_t>& col_ghosts, bool symmetric,
    CommunicationModel cm)
{
  const int P = comm.size(), me = comm.rank();
  // global row / column ranges (Matrix.cpp:175-186)
  std::vector<int64_t> nr = comm.allgather_value<int64_t>(nrows_local);
  std::vector<int64_t> nc = comm.allgather_value<int64_t>(ncols_local);
  std::vector<int64_t> row_ranges(P + 1, 0), col_ranges(P + 1, 0);
  for (int r = 0; r < P; ++r) {
    row_ranges[r + 1] = row_ranges[r] + nr[r];
    col_ranges[r + 1] = col_ranges[r] + nc[r];
  }

  // does anybody hold ghost rows?  (collective decision)
  std::vector<int64_t> ng
      = comm.allgather_value<int64_t>(static_cast<int64_t>(row_ghosts.size()));
  bool any = false;
  for (int64_t c : ng)
    any = any || c > 0;
  std::vector<ExtraEntry> extra;
  if (any) {
    // Ship every ghost row to its owner with GLOBAL column ids
    // (Matrix.cpp:229-292).  Packet: {global row, nnz, cols...} in an int64
    // stream and the values at the same positions of a T stream; all ranks
    // see all packets (host all-gather) and keep the rows they own.
    std::vector<int64_t> idx;
    std::vector<T> val;
    for (size_t i = 0; i < row_ghosts.size(); ++i) {
      const int64_t grow = row_ghosts[i];
      if (grow < 0 || grow >= row_ranges[P]
          || (grow >= row_ranges[me] && grow < row_ranges[me + 1]))
        throw std::runtime_error("create_matrix: bad ghost row index");
      const int32_t a = rowptr[nrows_local + i], b = rowptr[nrows_local + i + 1];
      idx.push_back(grow);
      val.push_back(T(0));
      idx.push_back(b - a);
      val.push_back(T(0));
      for (int32_t j = a; j < b; ++j) {
        int64_t gcol;
        if (colind[j] < ncols_local) {
          gcol = colind[j] + col_ranges[me];
        } else {
          const size_t g = static_cast<size_t>(colind[j] - ncols_local);
          if (g >= col_ghosts.size())
            throw std::runtime_error("create_matrix: ghost column out of range");
          gcol = col_ghosts[g];
        }
        idx.push_back(gcol);
        val.push_back(values[j]);
      }
    }
    auto all_idx = comm.allgatherv_bytes(idx.data(), idx.size() * sizeof(int64_t));
    auto all_val = comm.allgatherv_bytes(val.data(), val.size() * sizeof(T));
    for (int src = 0; src < P; ++src) { // source-rank order, like alltoallv
      if (src == me)
        continue;
      const int64_t* pi = reinterpret_cast<const int64_t*>(all_idx[src].data());
      const T* pv = reinterpret_cast<const T*>(all_val[src].data());
      const size_t n = all_idx[src].size() / sizeof(int64_t);
      size_t pos = 0;
      while (pos < n) {
        const int64_t grow = pi[pos];
        const int64_t cnt = pi[pos + 1];
        pos += 2;
        const bool mine = grow >= row_ranges[me] && grow < row_ranges[me + 1];
        for (int64_t k = 0; k < cnt; ++k, ++pos)
          if (mine)
            extra.push_back({static_cast<int32_t>(grow - row_ranges[me]),
                             pi[pos], pv[pos]});
      }
    }
  }
  return split_rows(rowptr, colind, values, nrows_local, ncols_local,
                    row_ranges[me], col_ranges[me], col_ghosts, symmetric, cm,
                    any ? &extra : nullptr);
}

template <typename T>
Matrix<T>* Matrix<T>::create_matrix(
    std::shared_ptr<const Comm> comm, std::shared_ptr<DeviceExecutor> exec,
    const int32_t* rowptr, const int32_t* colind, const T* values,
    int64_t nrows_local, int64_t ncols_local, std::vector<int64_t> row_ghosts,
    std::vector<int64_t> col_ghosts, bool symmetric, CommunicationModel cm)
{
  Split s = split_rows_distributed(*comm, rowptr, colind, values, nrows_local,
                                   ncols_local, row_ghosts, col_ghosts,
                                   symmetric, cm);

  auto col_map = std::make_shared<L2GMap>(comm, ncols_local, s.col_ghosts, exec,
                                          cm);
  auto row_map = std::make_shared<L2GMap>(comm, nrows_local,
                                          std::vector<int64_t>(), exec);
  if (symmetric)
    return new Matrix<T>(s.local, s.remote, s.diagonal, col_map, row_map,
                         s.nnz_full, exec);
  if (nonblocking(cm))
    return new Matrix<T>(s.local, s.remote, col_map, row_map, exec);
  return new Matrix<T>(s.local, col_map, row_map, exec);
}

// ---------------------------------------------------------------------------
// create_poisson3d: blocks generated on the device
// ---------------------------------------------------------------------------
namespace
{

struct DeviceBlock {
  int32_t* rowptr = nullptr;
  int32_t* colind = nullptr;
  double* values = nullptr;
  double* diagonal = nullptr;
  int64_t nnz = 0;
};

// a generator that throws half-way gives its arrays back
void release(HipExecutor& hip, DeviceBlock& b)
{
  if (b.rowptr)
    hip.free(b.rowptr);
  if (b.colind)
    hip.free(b.colind);
  if (b.values)
    hip.free(b.values);
  if (b.diagonal)
    hip.free(b.diagonal);
  b = DeviceBlock();
}

DeviceBlock generate_block(HipExecutor& hip, int32_t n, int64_t r0, int64_t r1,
                           int part, bool with_diagonal)
{
  DeviceBlock b;
  const int64_t nrows = r1 - r0;
  try {
    b.rowptr = hip.alloc<int32_t>(nrows + 1);
    throw_on_error(spmv_hip_poisson3d_count(hip.context(), n, r0, r1, part,
                                            b.rowptr, &b.nnz, nullptr),
                   "spmv_hip_poisson3d_count");
    b.colind = hip.alloc<int32_t>(b.nnz);
    b.values = hip.alloc<double>(b.nnz);
    if (with_diagonal)
      b.diagonal = hip.alloc<double>(nrows);
    throw_on_error(spmv_hip_poisson3d_fill_f64(hip.context(), n, r0, r1, part,
                                               b.rowptr, b.colind, b.values,
                                               b.diagonal, nullptr),
                   "spmv_hip_poisson3d_fill_f64");
  } catch (...) {
    release(hip, b);
    throw;
  }
  return b;
}

} // namespace

template <typename T>
Matrix<T>* Matrix<T>::create_poisson3d(std::shared_ptr<const Comm> comm,
                                       std::shared_ptr<DeviceExecutor> exec,
                                       int32_t n, bool symmetric,
                                       CommunicationModel cm)
{
  if constexpr (!std::is_same<T, double>::value) {
    throw std::runtime_error("create_poisson3d is available for double only");
  } else {
    auto* hip = dynamic_cast<HipExecutor*>(exec.get());
    if (!hip)
      throw std::runtime_error("create_poisson3d needs a HipExecutor");
    const int P = comm->size(), me = comm->rank();
    const int64_t N = static_cast<int64_t>(n) * n * n;
    const std::vector<int64_t> ranges = owner_ranges(P, N);
    const int64_t r0 = ranges[me], r1 = ranges[me + 1];
    int64_t gb = 0, ga = 0;
    throw_on_error(spmv_hip_poisson3d_ghosts(n, r0, r1, &gb, &ga),
                   "spmv_hip_poisson3d_ghosts (each rank needs >= n^2 rows)");
    std::vector<int64_t> ghosts;
    ghosts.reserve(gb + ga);
    for (int64_t g = r0 - gb; g < r0; ++g)
      ghosts.push_back(g);
    for (int64_t g = r1; g < r1 + ga; ++g)
      ghosts.push_back(g);
    const int32_t nrows = static_cast<int32_t>(r1 - r0);
    const int32_t ncols_all = static_cast<int32_t>(nrows + gb + ga);

    auto col_map = std::make_shared<L2GMap>(comm, nrows, ghosts, exec, cm);
    auto row_map = std::make_shared<L2GMap>(comm, nrows, std::vector<int64_t>(),
                                            exec);
    std::unique_ptr<Matrix<T>> A(new Matrix<T>());
    A->_exec = exec;
    A->_col_map = col_map;
    A->_row_map = row_map;
    A->_symmetric = symmetric;
    using Adopt = typename CSRMatrix<T>::AdoptDevice;
    auto adopt = [&](const DeviceBlock& b, int32_t ncols, bool sym) {
      return new CSRMatrix<T>(Adopt{}, exec, nrows, ncols, b.nnz, b.rowptr,
                              b.colind, b.values, b.diagonal, sym);
    };
    if (symmetric) {
      DeviceBlock L = generate_block(*hip, n, r0, r1,
                                     SPMV_HIP_PART_LOCAL_LOWER, true);
      DeviceBlock R = generate_block(*hip, n, r0, r1, SPMV_HIP_PART_REMOTE,
                                     false);
      A->_mat_local.reset(adopt(L, ncols_all, true));
      A->_mat_remote.reset(adopt(R, ncols_all, false));
      A->_nnz = 2 * L.nnz + R.nnz + nrows; // Matrix.cpp:443-444
    } else if (nonblocking(cm)) {
      DeviceBlock L = generate_block(*hip, n, r0, r1, SPMV_HIP_PART_LOCAL,
                                     false);
      DeviceBlock R = generate_block(*hip, n, r0, r1, SPMV_HIP_PART_REMOTE,
                                     false);
      A->_mat_local.reset(adopt(L, nrows, false));
      A->_mat_remote.reset(adopt(R, ncols_all, false));
      A->_nnz = L.nnz + R.nnz;
    } else {
      DeviceBlock B = generate_block(*hip, n, r0, r1, SPMV_HIP_PART_ALL, false);
      A->_mat_local.reset(adopt(B, ncols_all, false));
      A->_nnz = B.nnz;
    }
    return A.release();
  }
}

template <typename T>
Matrix<T>* Matrix<T>::create_unstructured(std::shared_ptr<const Comm> comm,
                                          std::shared_ptr<DeviceExecutor> exec,
                                          int64_t nrows, int per_row,
                                          int64_t band, int far_permille,
                                          uint64_t seed)
{
  if constexpr (!std::is_same<T, double>::value) {
    throw std::runtime_error("create_unstructured is available for double only");
  } else {
    auto* hip = dynamic_cast<HipExecutor*>(exec.get());
    if (!hip)
      throw std::runtime_error("create_unstructured needs a HipExecutor");
    if (comm->size() != 1)
      throw std::runtime_error("create_unstructured: one rank only");
    if (nrows < 1 || per_row < 1 || nrows * per_row > INT32_MAX)
      throw std::runtime_error("create_unstructured: size out of range");
    DeviceBlock b;
    b.nnz = nrows * per_row;
    try {
      b.rowptr = hip->alloc<int32_t>(nrows + 1);
      b.colind = hip->alloc<int32_t>(b.nnz);
      b.values = hip->alloc<double>(b.nnz);
      throw_on_error(spmv_hip_unstructured_fill_f64(hip->context(), nrows, per_row,
                                                    band, far_permille, seed,
                                                    b.rowptr, b.colind, b.values,
                                                    nullptr),
                     "spmv_hip_unstructured_fill_f64");
    } catch (...) {
      release(*hip, b);
      throw;
    }
    const int32_t n32 = static_cast<int32_t>(nrows);
    auto col_map = std::make_shared<L2GMap>(comm, n32, std::vector<int64_t>(),
                                            exec);
    auto row_map = std::make_shared<L2GMap>(comm, n32, std::vector<int64_t>(),
                                            exec);
    std::unique_ptr<Matrix<T>> A(new Matrix<T>());
    A->_exec = exec;
    A->_col_map = col_map;
    A->_row_map = row_map;
    A->_symmetric = false;
    using Adopt = typename CSRMatrix<T>::AdoptDevice;
    A->_mat_local.reset(new CSRMatrix<T>(Adopt{}, exec, n32, n32, b.nnz, b.rowptr,
                                         b.colind, b.values, nullptr, false));
    A->_nnz = b.nnz;
    return A.release();
  }
}

template <typename T>
Matrix<T>* Matrix<T>::create_fem_like(std::shared_ptr<const Comm> comm,
                                      std::shared_ptr<DeviceExecutor> exec,
                                      const spmv_hip_fem_params& params,
                                      bool symmetric)
{
  if constexpr (!std::is_same<T, double>::value) {
    throw std::runtime_error("create_fem_like is available for double only");
  } else {
    auto* hip = dynamic_cast<HipExecutor*>(exec.get());
    if (!hip)
      throw std::runtime_error("create_fem_like needs a HipExecutor");
    if (comm->size() != 1)
      throw std::runtime_error("create_fem_like: one rank only");
    if (params.num_rows < 1 || params.num_rows > INT32_MAX)
      throw std::runtime_error("create_fem_like: size out of range");
    const int64_t nrows = params.num_rows;
    DeviceBlock b;
    try {
      b.rowptr = hip->alloc<int32_t>(nrows + 1);
      throw_on_error(spmv_hip_fem_count(hip->context(), &params, b.rowptr, &b.nnz,
                                        nullptr),
                     "spmv_hip_fem_count");
      b.colind = hip->alloc<int32_t>(b.nnz);
      b.values = hip->alloc<double>(b.nnz);
      throw_on_error(spmv_hip_fem_fill_f64(hip->context(), &params, b.nnz,
                                           b.rowptr, b.colind, b.values, nullptr),
                     "spmv_hip_fem_fill_f64");
    } catch (...) {
      release(*hip, b);
      throw;
    }
    const int32_t n32 = static_cast<int32_t>(nrows);
    if (symmetric) { // strictly lower block + diagonal of the generated matrix
      DeviceBlock l;
      try {
        l.rowptr = hip->alloc<int32_t>(nrows + 1);
        throw_on_error(spmv_hip_csr_lower_split_count(hip->context(), n32, b.rowptr,
                                                      b.colind, l.rowptr, &l.nnz,
                                                      nullptr),
                       "spmv_hip_csr_lower_split_count");
        l.colind = hip->alloc<int32_t>(l.nnz > 0 ? l.nnz : 1);
        l.values = hip->alloc<double>(l.nnz > 0 ? l.nnz : 1);
        l.diagonal = hip->alloc<double>(nrows);
        throw_on_error(spmv_hip_csr_lower_split_fill_f64(
                           hip->context(), n32, b.rowptr, b.colind, b.values, l.rowptr,
                           l.colind, l.values, l.diagonal, nullptr),
                       "spmv_hip_csr_lower_split_fill_f64");
        hip->synchronize();
      } catch (...) {
        release(*hip, l);
        release(*hip, b);
        throw;
      }
      release(*hip, b);
      b = l;
    }
    auto col_map = std::make_shared<L2GMap>(comm, n32, std::vector<int64_t>(),
                                            exec);
    auto row_map = std::make_shared<L2GMap>(comm, n32, std::vector<int64_t>(),
                                            exec);
    std::unique_ptr<Matrix<T>> A(new Matrix<T>());
    A->_exec = exec;
    A->_col_map = col_map;
    A->_row_map = row_map;
    A->_symmetric = symmetric;
    using Adopt = typename CSRMatrix<T>::AdoptDevice;
    A->_mat_local.reset(new CSRMatrix<T>(Adopt{}, exec, n32, n32, b.nnz, b.rowptr,
                                         b.colind, b.values, b.diagonal, symmetric));
    A->_nnz = b.nnz;
    if (symmetric) { // symmetric storage always has its (here: empty) remote block
      A->_mat_remote.reset(new CSRMatrix<T>(Adopt{}, exec, n32, n32, 0, nullptr,
                                            nullptr, nullptr, nullptr, false));
      A->_nnz = 2 * b.nnz + n32; // Matrix.cpp:443-444
    }
    return A.release();
  }
}

// ---------------------------------------------------------------------------
// 3-D block partition of the Poisson matrix
// ---------------------------------------------------------------------------
namespace
{
// first index and extent of part i of an axis of n points cut into p parts
// (read_petsc.cpp:20-37 per axis: the first n mod p parts get one more)
inline void axis_part(int64_t n, int p, int i, int64_t* first, int64_t* len)
{
  const int64_t q = n / p, rem = n % p;
  *len = q + (i < rem ? 1 : 0);
  *first = i * q + (i < rem ? i : rem);
}
inline int axis_owner(int64_t n, int p, int64_t c)
{
  const int64_t q = n / p, rem = n % p;
  const int64_t big = rem * (q + 1); // points in the parts of size q + 1
  return c < big ? (int)(c / (q + 1)) : (int)(rem + (c - big) / q);
}
} // namespace

template <typename T>
typename Matrix<T>::BoxRows Matrix<T>::poisson3d_box_rows(int32_t n, int px,
                                                          int py, int pz,
                                                          int rank)
{
  if (n < 1 || px < 1 || py < 1 || pz < 1 || px > n || py > n || pz > n
      || rank < 0 || rank >= px * py * pz)
    throw std::runtime_error("poisson3d_box_rows: bad partition");
  const int P[3] = {px, py, pz};
  // box of every rank: first point and extents; global offset = points of the
  // lower ranks
  auto box_of = [&](int r, int64_t first[3], int64_t len[3]) {
    const int i[3] = {r % px, (r / px) % py, r / (px * py)};
    for (int a = 0; a < 3; ++a)
      axis_part(n, P[a], i[a], &first[a], &len[a]);
  };
  const int nranks = px * py * pz;
  std::vector<int64_t> offset(nranks + 1, 0);
  for (int r = 0; r < nranks; ++r) {
    int64_t f[3], l[3];
    box_of(r, f, l);
    offset[r + 1] = offset[r] + l[0] * l[1] * l[2];
  }
  // global id of grid point (x, y, z)
  auto global_id = [&](int64_t x, int64_t y, int64_t z) {
    const int r = axis_owner(n, px, x)
                  + px * (axis_owner(n, py, y) + py * axis_owner(n, pz, z));
    int64_t f[3], l[3];
    box_of(r, f, l);
    return offset[r] + (x - f[0]) + l[0] * ((y - f[1]) + l[1] * (z - f[2]));
  };
  int64_t f[3], l[3];
  box_of(rank, f, l);
  BoxRows out;
  out.global_row_offset = offset[rank];
  out.box[0] = l[0];
  out.box[1] = l[1];
  out.box[2] = l[2];
  const int64_t nloc = l[0] * l[1] * l[2];
  const int64_t g0 = offset[rank], g1 = offset[rank + 1];
  // pass 1: the ghost columns (points one step outside the six faces)
  std::vector<int64_t>& ghosts = out.col_ghosts;
  for (int64_t z = 0; z < l[2]; ++z)
    for (int64_t y = 0; y < l[1]; ++y)
      for (int64_t x = 0; x < l[0]; ++x) {
        const bool face = x == 0 || y == 0 || z == 0 || x == l[0] - 1
                          || y == l[1] - 1 || z == l[2] - 1;
        if (!face)
          continue;
        const int64_t X = f[0] + x, Y = f[1] + y, Z = f[2] + z;
        const int64_t nb[6][3] = {{X - 1, Y, Z}, {X + 1, Y, Z}, {X, Y - 1, Z},
                                  {X, Y + 1, Z}, {X, Y, Z - 1}, {X, Y, Z + 1}};
        for (const auto& q : nb) {
          if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[0] >= n || q[1] >= n
              || q[2] >= n)
            continue;
          const int64_t g = global_id(q[0], q[1], q[2]);
          if (g < g0 || g >= g1)
            ghosts.push_back(g);
        }
      }
  std::sort(ghosts.begin(), ghosts.end());
  ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
  // pass 2: the rows, entries ascending by global column
  CsrHost<T>& A = out.rows;
  A.rows = static_cast<int32_t>(nloc);
  A.cols = static_cast<int32_t>(nloc + (int64_t)ghosts.size());
  A.rowptr.assign(1, 0);
  A.rowptr.reserve(nloc + 1);
  A.colind.reserve(7 * nloc);
  A.values.reserve(7 * nloc);
  for (int64_t z = 0; z < l[2]; ++z)
    for (int64_t y = 0; y < l[1]; ++y)
      for (int64_t x = 0; x < l[0]; ++x) {
        const int64_t X = f[0] + x, Y = f[1] + y, Z = f[2] + z;
        const int64_t nb[7][3] = {{X, Y, Z - 1}, {X, Y - 1, Z}, {X - 1, Y, Z},
                                  {X, Y, Z},     {X + 1, Y, Z}, {X, Y + 1, Z},
                                  {X, Y, Z + 1}};
        std::pair<int64_t, T> ent[7];
        int ne = 0;
        for (int k = 0; k < 7; ++k) {
          const auto& q = nb[k];
          if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[0] >= n || q[1] >= n
              || q[2] >= n)
            continue;
          ent[ne++] = {global_id(q[0], q[1], q[2]), k == 3 ? T(6) : T(-1)};
        }
        std::sort(ent, ent + ne,
                  [](const std::pair<int64_t, T>& a,
                     const std::pair<int64_t, T>& b) { return a.first < b.first; });
        for (int k = 0; k < ne; ++k) {
          const int64_t g = ent[k].first;
          int32_t c;
          if (g >= g0 && g < g1)
            c = static_cast<int32_t>(g - g0);
          else
            c = static_cast<int32_t>(
                nloc
                + (std::lower_bound(ghosts.begin(), ghosts.end(), g)
                   - ghosts.begin()));
          A.colind.push_back(c);
          A.values.push_back(ent[k].second);
        }
        A.rowptr.push_back(static_cast<int32_t>(A.colind.size()));
      }
  return out;
}

namespace
{
DeviceBlock generate_box_block(HipExecutor& hip, int32_t n, const int32_t f[3],
                               const int32_t l[3], int part, bool with_diagonal)
{
  DeviceBlock b;
  const int64_t nrows = (int64_t)l[0] * l[1] * l[2];
  try {
    b.rowptr = hip.alloc<int32_t>(nrows + 1);
    throw_on_error(spmv_hip_poisson3d_box_count(hip.context(), n, f, l, part,
                                                b.rowptr, &b.nnz, nullptr,
                                                nullptr),
                   "spmv_hip_poisson3d_box_count");
    b.colind = hip.alloc<int32_t>(b.nnz);
    b.values = hip.alloc<double>(b.nnz);
    if (with_diagonal)
      b.diagonal = hip.alloc<double>(nrows);
    throw_on_error(spmv_hip_poisson3d_box_fill_f64(hip.context(), n, f, l, part,
                                                   b.rowptr, b.colind, b.values,
                                                   b.diagonal, nullptr),
                   "spmv_hip_poisson3d_box_fill_f64");
  } catch (...) {
    release(hip, b);
    throw;
  }
  return b;
}
} // namespace

template <typename T>
Matrix<T>* Matrix<T>::create_poisson3d_boxes(std::shared_ptr<const Comm> comm,
                                             std::shared_ptr<DeviceExecutor> exec,
                                             int32_t n, int px, int py, int pz,
                                             bool symmetric,
                                             CommunicationModel cm)
{
  if (px * py * pz != comm->size())
    throw std::runtime_error(
        "create_poisson3d_boxes: px * py * pz must equal the number of ranks");
  auto* hip = dynamic_cast<HipExecutor*>(exec.get());
  if constexpr (!std::is_same<T, double>::value) {
    hip = nullptr;
  }
  if (!hip) { // rows on the host, through create_matrix
    BoxRows b = poisson3d_box_rows(n, px, py, pz, comm->rank());
    return create_matrix(comm, exec, b.rows.rowptr.data(), b.rows.colind.data(),
                         b.rows.values.data(), b.rows.rows, b.rows.rows, {},
                         b.col_ghosts, symmetric, cm);
  }
  if constexpr (std::is_same<T, double>::value) {
    // The blocks are generated on the device (spmv_hip_poisson3d_box_*); the
    // host only lists the ghost columns -- the points behind the six faces, in
    // ascending global id: face by face (the neighbour ranks ascend in the
    // order -z -y -x +x +y +z), inside a face in the neighbour's local order.
    if (n < 1 || px < 1 || py < 1 || pz < 1 || px > n || py > n || pz > n)
      throw std::runtime_error("create_poisson3d_boxes: bad partition");
    const int P[3] = {px, py, pz};
    auto box_of = [&](int r, int64_t first[3], int64_t len[3]) {
      const int i[3] = {r % px, (r / px) % py, r / (px * py)};
      for (int a = 0; a < 3; ++a)
        axis_part(n, P[a], i[a], &first[a], &len[a]);
    };
    const int nranks = px * py * pz, me = comm->rank();
    std::vector<int64_t> offset(nranks + 1, 0);
    for (int r = 0; r < nranks; ++r) {
      int64_t f[3], l[3];
      box_of(r, f, l);
      offset[r + 1] = offset[r] + l[0] * l[1] * l[2];
    }
    int64_t f[3], l[3];
    box_of(me, f, l);
    const int dr[6] = {-px * py, -px, -1, 1, px, px * py};
    const int axis[6] = {2, 1, 0, 0, 1, 2};
    std::vector<int64_t> ghosts;
    for (int s = 0; s < 6; ++s) {
      const int a = axis[s];
      const bool has = s < 3 ? f[a] > 0 : f[a] + l[a] < n;
      if (!has)
        continue;
      const int rn = me + dr[s];
      int64_t fn[3], ln[3];
      box_of(rn, fn, ln);
      // the neighbour's layer that touches this box, its other two
      // coordinates running over the shared face (u fastest)
      const int64_t fixed = s < 3 ? ln[a] - 1 : 0;
      const int u = a == 0 ? 1 : 0, v = a == 2 ? 1 : 2; // in-face axes, u < v
      for (int64_t cv = 0; cv < l[v]; ++cv)
        for (int64_t cu = 0; cu < l[u]; ++cu) {
          int64_t c[3];
          c[a] = fixed;
          c[u] = cu;
          c[v] = cv;
          ghosts.push_back(offset[rn] + c[0] + ln[0] * (c[1] + ln[1] * c[2]));
        }
    }
    const int64_t nloc = l[0] * l[1] * l[2];
    if (nloc + (int64_t)ghosts.size() > INT32_MAX)
      throw std::runtime_error("create_poisson3d_boxes: box too large");
    const int32_t nrows = static_cast<int32_t>(nloc);
    const int32_t ncols_all = static_cast<int32_t>(nloc + (int64_t)ghosts.size());
    const int32_t f32[3] = {(int32_t)f[0], (int32_t)f[1], (int32_t)f[2]};
    const int32_t l32[3] = {(int32_t)l[0], (int32_t)l[1], (int32_t)l[2]};

    auto col_map = std::make_shared<L2GMap>(comm, nrows, ghosts, exec, cm);
    auto row_map = std::make_shared<L2GMap>(comm, nrows, std::vector<int64_t>(),
                                            exec);
    std::unique_ptr<Matrix<T>> A(new Matrix<T>());
    A->_exec = exec;
    A->_col_map = col_map;
    A->_row_map = row_map;
    A->_symmetric = symmetric;
    using Adopt = typename CSRMatrix<T>::AdoptDevice;
    auto adopt = [&](const DeviceBlock& b, int32_t ncols, bool sym) {
      return new CSRMatrix<T>(Adopt{}, exec, nrows, ncols, b.nnz, b.rowptr,
                              b.colind, b.values, b.diagonal, sym);
    };
    auto gen = [&](int part, bool diag) {
      return generate_box_block(*hip, n, f32, l32, part, diag);
    };
    if (symmetric) {
      DeviceBlock L = gen(SPMV_HIP_PART_LOCAL_LOWER, true);
      DeviceBlock R = gen(SPMV_HIP_PART_REMOTE, false);
      A->_mat_local.reset(adopt(L, ncols_all, true));
      A->_mat_remote.reset(adopt(R, ncols_all, false));
      A->_nnz = 2 * L.nnz + R.nnz + nrows; // Matrix.cpp:443-444
    } else if (nonblocking(cm)) {
      DeviceBlock L = gen(SPMV_HIP_PART_LOCAL, false);
      DeviceBlock R = gen(SPMV_HIP_PART_REMOTE, false);
      A->_mat_local.reset(adopt(L, nrows, false));
      A->_mat_remote.reset(adopt(R, ncols_all, false));
      A->_nnz = L.nnz + R.nnz;
    } else {
      DeviceBlock B = gen(SPMV_HIP_PART_ALL, false);
      A->_mat_local.reset(adopt(B, ncols_all, false));
      A->_nnz = B.nnz;
    }
    return A.release();
  }
  return nullptr; // not reached
}

template class Matrix<float>;
template class Matrix<double>;

} // namespace spmv
