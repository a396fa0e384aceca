// PETSc binary matrix / vector ingest, mirroring spmv/read_petsc.{h,cpp}
// ("next" row n1): lets the reference's own demo inputs
// (`-ksp_view_mat binary` files) drive the MI355X backend.
//
// File format (read_petsc.cpp:56-110, 259-303), all big-endian:
//   matrix: int32 {1211216, nrows, ncols, nnz}, nrows x int32 row lengths,
//           nnz x int32 column ids, nnz x fp64 values
//   vector: int32 {1211214, n}, n x fp64
// Rows (and columns) are split over the ranks by owner_ranges
// (read_petsc.cpp:20-37); every rank reads only its own slice of the file.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "comm.h"
#include "executor.h"
#include "matrix.h"

namespace spmv
{

// This rank's rows of the file in the numbering create_matrix expects: owned
// columns shifted to [0, ncols_local), ghost columns >= ncols_local indexing
// into `col_ghosts` (ascending global ids, read_petsc.cpp:128-151).
struct PetscRows {
  int64_t nrows_global = 0, ncols_global = 0, nnz_global = 0;
  int64_t row_begin = 0, row_end = 0; // owned global rows
  int64_t col_begin = 0, col_end = 0; // owned global columns
  std::vector<int32_t> rowptr = {0};
  std::vector<int32_t> colind;
  std::vector<double> values;
  std::vector<int64_t> col_ghosts;
};

// Host-only part: parse the file for rank `rank` of `size`.
PetscRows read_petsc_binary_rows(const std::string& filename, int rank,
                                 int size);

// read_petsc.cpp:40-228.  Same split rules as create_matrix; the symmetric
// matrix reports the file's total nnz (read_petsc.cpp:219-221).
std::unique_ptr<Matrix<double>> read_petsc_binary_matrix(
    const std::string& filename, std::shared_ptr<const Comm> comm,
    std::shared_ptr<DeviceExecutor> exec, bool symmetric = false,
    CommunicationModel cm = CommunicationModel::collective_blocking);

// read_petsc.cpp:230-303: this rank's slice of the vector, copied to the
// device; the caller frees it with exec->free().  `nrows_local` (optional)
// receives the slice length.
double* read_petsc_binary_vector(const Comm& comm, const DeviceExecutor* exec,
                                 const std::string& filename,
                                 int64_t* nrows_local = nullptr);

} // namespace spmv
