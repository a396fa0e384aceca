// Distributed sparse matrix, mirroring spmv/Matrix.{h,cpp}:
//   three constructors (Matrix.cpp:21-72), mult dispatcher (:131-141), the
//   four SpMV variants (:483-552), create_matrix raw-pointer overload
//   (:164-480) and the size queries (:75-128).
// Differences, all forced by the environment or the scope:
//   * Eigen::SparseMatrix arguments become CsrHost<T> (no Eigen here).
//   * MPI_Comm becomes std::shared_ptr<const Comm>.
//   * ghost rows are shipped to their owners over the communicator's host
//     all-gather instead of MPI neighbourhood collectives (Matrix.cpp:188-292).
//   * create_poisson3d builds the blocks directly on the device (the
//     reference has no Poisson generator).
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

struct spmv_hip_fem_params; // include/spmv_hip.h

#include "comm.h"
#include "csr.h"
#include "l2gmap.h"

namespace spmv
{

template <typename T>
class Matrix
{
public:
  // single block, non-overlapping column map (Matrix.cpp:21-32)
  Matrix(const CsrHost<T>& mat, std::shared_ptr<L2GMap> col_map,
         std::shared_ptr<L2GMap> row_map, std::shared_ptr<DeviceExecutor> exec);
  // local + remote blocks, overlapping column map (Matrix.cpp:35-49)
  Matrix(const CsrHost<T>& mat_local, const CsrHost<T>& mat_remote,
         std::shared_ptr<L2GMap> col_map, std::shared_ptr<L2GMap> row_map,
         std::shared_ptr<DeviceExecutor> exec);
  // symmetric: strictly-lower local + remote + diagonal (Matrix.cpp:52-72)
  Matrix(const CsrHost<T>& mat_local, const CsrHost<T>& mat_remote,
         const std::vector<T>& mat_diagonal, std::shared_ptr<L2GMap> col_map,
         std::shared_ptr<L2GMap> row_map, int64_t nnz_full,
         std::shared_ptr<DeviceExecutor> exec);
  ~Matrix() = default;

  int rows() const;
  int cols() const;
  int64_t non_zeros() const;
  bool symmetric() const { return _symmetric; }
  size_t format_size() const;

  // y = A x ; x must hold local_size + num_ghosts entries and col_map()->
  // update(x) must have been called (Matrix.cpp:131-141).  Device pointers.
  void mult(T* x, T* y) const;
  // mult + the dot product x.y fused into the SpMV kernels where possible.
  // Emits partial sums into dot_local / dot_remote (each
  // spmv_hip_dot_partials_len() device doubles; their total is x[0:rows].y).
  // Returns false if it fell back to plain mult() (symmetric matrices): the
  // caller then computes the dot product itself.
  // `ev_local_done` (optional, a HipExecutor event) is recorded right after
  // the local block's kernel was enqueued -- benchmarks time that kernel.
  bool mult_dot(T* x, T* y, double* dot_local, double* dot_remote,
                void* ev_local_done = nullptr) const;

  // Mixed precision: enable_mixed() builds fp32 copies of the
  // blocks' values (general storage only, returns false otherwise);
  // use_mixed(true) then makes mult / mult_dot stream those, x, y and the
  // arithmetic staying fp64.  cg() switches it per SpMV.
  bool enable_mixed() const;
  void use_mixed(bool on) const;

  std::shared_ptr<L2GMap> row_map() const { return _row_map; }
  std::shared_ptr<const L2GMap> col_map() const { return _col_map; }

  // Matrix.cpp:164-480, raw-pointer overload.  rowptr/colind/values are HOST
  // arrays of this rank's rows; colind uses local numbering with ghost
  // columns >= ncols_local indexing into col_ghosts.  Caller owns the result.
  static Matrix<T>* create_matrix(
      std::shared_ptr<const Comm> comm, std::shared_ptr<DeviceExecutor> exec,
      const int32_t* rowptr, const int32_t* colind, const T* values,
      int64_t nrows_local, int64_t ncols_local,
      std::vector<int64_t> row_ghosts, std::vector<int64_t> col_ghosts,
      bool symmetric = false,
      CommunicationModel cm = CommunicationModel::collective_blocking);

  // The host half of create_matrix (Matrix.cpp:295-408): ghost-column
  // renumbering and the local / remote / diagonal split, without touching a
  // device.  create_matrix = split_rows + block upload + L2GMap.
  struct Split {
    CsrHost<T> local, remote;        // remote is empty for blocking models
    std::vector<T> diagonal;         // symmetric only
    std::vector<int64_t> col_ghosts; // ascending, unique
    int64_t nnz_full = 0;            // what Matrix::non_zeros() reports
    bool two_blocks = false;
  };
  // Contributions to this rank's rows received from other ranks (ghost-row
  // elimination): local row, GLOBAL column, value.
  struct ExtraEntry {
    int32_t row;
    int64_t global_col;
    T val;
  };
  static Split split_rows(const int32_t* rowptr, const int32_t* colind,
                          const T* values, int64_t nrows_local,
                          int64_t ncols_local, int64_t global_row_offset,
                          int64_t global_col_offset,
                          const std::vector<int64_t>& col_ghosts,
                          bool symmetric, CommunicationModel cm,
                          const std::vector<ExtraEntry>* extra = nullptr);
  // Host half of create_matrix including the ghost-row exchange
  // (Matrix.cpp:175-292): collective over `comm`, touches no device.
  static Split split_rows_distributed(
      const Comm& comm, const int32_t* rowptr, const int32_t* colind,
      const T* values, int64_t nrows_local, int64_t ncols_local,
      const std::vector<int64_t>& row_ghosts,
      const std::vector<int64_t>& col_ghosts, bool symmetric,
      CommunicationModel cm);

  // 3-D 7-point Poisson matrix on an n^3 grid, rows split over the ranks of
  // `comm` by the reference's even rule (read_petsc.cpp:20-37), generated on
  // the device.  Blocks obey the same split rules as create_matrix.
  static Matrix<T>* create_poisson3d(
      std::shared_ptr<const Comm> comm, std::shared_ptr<DeviceExecutor> exec,
      int32_t n, bool symmetric = false,
      CommunicationModel cm = CommunicationModel::collective_blocking);

  // Seeded unstructured test matrix generated on the device
  // (spmv_hip_unstructured_fill_f64): one rank only, general storage, ONE
  // block.  A synthetic input for measurements, like create_poisson3d.
  static Matrix<T>* create_unstructured(
      std::shared_ptr<const Comm> comm, std::shared_ptr<DeviceExecutor> exec,
      int64_t nrows, int per_row, int64_t band, int far_permille, uint64_t seed);

  // Seeded FEM-like test matrix generated on the device (spmv_hip_fem_count /
  // _fill_f64: ragged rows, optionally a tail of very long ones, in a
  // bandwidth-reducing order): one rank only, ONE block.  symmetric: the
  // strictly lower part and the diagonal of that matrix in symmetric storage
  // (spmv_hip_csr_lower_split_*: the rule of Matrix.cpp:337-349 on the device).
  static Matrix<T>* create_fem_like(std::shared_ptr<const Comm> comm,
                                    std::shared_ptr<DeviceExecutor> exec,
                                    const spmv_hip_fem_params& params,
                                    bool symmetric = false);

  // The same matrix on a 3-D BLOCK partition (the reference has
  // row slabs only): the n^3 grid is cut into px * py * pz boxes (sizes by the
  // even rule per axis), rank = ix + px (iy + py iz) owns one box, and the
  // global numbering is rank-major -- a box's points are consecutive, in
  // x-fastest order -- so that every rank still owns one contiguous row range,
  // as L2GMap requires.  The halo is the six faces of the box (surface, not
  // two full planes); it goes through the general pack / exchange path.  The
  // blocks are generated on the device (spmv_hip_poisson3d_box_*; fp32 and
  // host executors: rows on the host, handed to create_matrix).
  static Matrix<T>* create_poisson3d_boxes(
      std::shared_ptr<const Comm> comm, std::shared_ptr<DeviceExecutor> exec,
      int32_t n, int px, int py, int pz, bool symmetric = false,
      CommunicationModel cm = CommunicationModel::collective_blocking);
  // its host half: this rank's rows in create_matrix's input form (columns
  // local: owned first, ghosts as col_ghosts.size()-relative indices behind
  // them, col_ghosts ascending global ids; entries of a row ascending by
  // global column)
  struct BoxRows {
    CsrHost<T> rows;
    std::vector<int64_t> col_ghosts;
    int64_t global_row_offset = 0;
    int64_t box[3] = {0, 0, 0}; // box extents
  };
  static BoxRows poisson3d_box_rows(int32_t n, int px, int py, int pz, int rank);

  const SubMatrix<T>* local_block() const { return _mat_local.get(); }
  const SubMatrix<T>* remote_block() const { return _mat_remote.get(); }

private:
  Matrix() = default;
  std::shared_ptr<DeviceExecutor> _exec;
  std::unique_ptr<CSRMatrix<T>> _mat_local;
  std::unique_ptr<CSRMatrix<T>> _mat_remote;
  std::shared_ptr<L2GMap> _col_map;
  std::shared_ptr<L2GMap> _row_map;
  int64_t _nnz = 0; // the reference keeps an int (Matrix.h:122); 512^3 fits
  bool _symmetric = false;

  void spmv(T* x, T* y) const;
  void spmv_overlap(T* x, T* y) const;
  void spmv_sym(T* x, T* y) const;
  void spmv_sym_overlap(T* x, T* y) const;
};

// even contiguous split, first N % size ranks get one extra row
// (read_petsc.cpp:20-37, tests/test_spmv.cpp:25-41)
std::vector<int64_t> owner_ranges(int size, int64_t N);

} // namespace spmv
