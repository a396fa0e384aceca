// PETSc binary ingest: see read_petsc.h.
#include "read_petsc.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace spmv
{

namespace
{

constexpr int32_t kMatrixId = 1211216; // read_petsc.cpp:75
constexpr int32_t kVectorId = 1211214; // read_petsc.cpp:259

int32_t be32(const unsigned char* p)
{
  return static_cast<int32_t>((uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16)
                              | (uint32_t(p[2]) << 8) | uint32_t(p[3]));
}

double be64(const unsigned char* p)
{
  uint64_t u = 0;
  for (int i = 0; i < 8; ++i)
    u = (u << 8) | p[i];
  double d;
  std::memcpy(&d, &u, sizeof(d));
  return d;
}

std::vector<unsigned char> read_bytes(std::ifstream& f, std::streamoff pos,
                                      size_t count)
{
  std::vector<unsigned char> buf(count);
  f.seekg(pos, std::ios::beg);
  if (count)
    f.read(reinterpret_cast<char*>(buf.data()),
           static_cast<std::streamsize>(count));
  if (!f)
    throw std::runtime_error("PETSc file is truncated");
  return buf;
}

} // namespace

PetscRows read_petsc_binary_rows(const std::string& filename, int rank,
                                 int size)
{
  std::ifstream file(filename.c_str(), std::ios::in | std::ios::binary);
  if (!file.is_open())
    throw std::runtime_error("Could not open file"); // read_petsc.cpp:59
  const std::vector<unsigned char> head = read_bytes(file, 0, 16);
  if (be32(head.data()) != kMatrixId)
    throw std::runtime_error("Bad signature in PETSc Matrix file"); // :76
  PetscRows out;
  out.nrows_global = be32(head.data() + 4);
  out.ncols_global = be32(head.data() + 8);
  out.nnz_global = be32(head.data() + 12);
  if (out.nrows_global < 0 || out.ncols_global < 0 || out.nnz_global < 0)
    throw std::runtime_error("Negative size in PETSc Matrix file");
  const std::vector<int64_t> rr = owner_ranges(size, out.nrows_global);
  const std::vector<int64_t> cr = owner_ranges(size, out.ncols_global);
  out.row_begin = rr[rank];
  out.row_end = rr[rank + 1];
  out.col_begin = cr[rank];
  out.col_end = cr[rank + 1];

  // row lengths of ALL rows (needed for this rank's offset, :92-113)
  const std::vector<unsigned char> lens
      = read_bytes(file, 16, static_cast<size_t>(out.nrows_global) * 4);
  int64_t nnz_offset = 0, nnz_size = 0, nnz_sum = 0;
  const int64_t nloc = out.row_end - out.row_begin;
  out.rowptr.assign(nloc + 1, 0);
  for (int64_t i = 0; i < out.nrows_global; ++i) {
    const int32_t len = be32(lens.data() + 4 * i);
    if (len < 0)
      throw std::runtime_error("Negative row length in PETSc Matrix file");
    nnz_sum += len;
    if (i < out.row_begin)
      nnz_offset += len;
    else if (i < out.row_end) {
      nnz_size += len;
      out.rowptr[i - out.row_begin + 1] = static_cast<int32_t>(nnz_size);
    }
  }
  if (nnz_sum != out.nnz_global)
    throw std::runtime_error("Row lengths do not add up to nnz"); // :104
  if (nnz_size > INT32_MAX)
    throw std::runtime_error("Local nnz exceeds the int32 row pointer");

  const std::streamoff col_pos = 16 + out.nrows_global * 4;
  const std::streamoff val_pos = col_pos + out.nnz_global * 4;
  const std::vector<unsigned char> cols
      = read_bytes(file, col_pos + nnz_offset * 4,
                   static_cast<size_t>(nnz_size) * 4);
  const std::vector<unsigned char> vals
      = read_bytes(file, val_pos + nnz_offset * 8,
                   static_cast<size_t>(nnz_size) * 8);

  // ghost columns in ascending global order after the owned ones (:128-151)
  std::vector<int64_t> gcol(nnz_size);
  for (int64_t j = 0; j < nnz_size; ++j) {
    gcol[j] = be32(cols.data() + 4 * j);
    if (gcol[j] < 0 || gcol[j] >= out.ncols_global)
      throw std::runtime_error("Column index out of range in PETSc file");
    if (gcol[j] < out.col_begin || gcol[j] >= out.col_end)
      out.col_ghosts.push_back(gcol[j]);
  }
  std::sort(out.col_ghosts.begin(), out.col_ghosts.end());
  out.col_ghosts.erase(
      std::unique(out.col_ghosts.begin(), out.col_ghosts.end()),
      out.col_ghosts.end());
  const int64_t ncols_local = out.col_end - out.col_begin;
  out.colind.resize(nnz_size);
  out.values.resize(nnz_size);
  for (int64_t j = 0; j < nnz_size; ++j) {
    const int64_t g = gcol[j];
    if (g >= out.col_begin && g < out.col_end)
      out.colind[j] = static_cast<int32_t>(g - out.col_begin);
    else
      out.colind[j] = static_cast<int32_t>(
          ncols_local
          + (std::lower_bound(out.col_ghosts.begin(), out.col_ghosts.end(), g)
             - out.col_ghosts.begin()));
    out.values[j] = be64(vals.data() + 8 * j);
  }
  return out;
}

std::unique_ptr<Matrix<double>>
read_petsc_binary_matrix(const std::string& filename,
                         std::shared_ptr<const Comm> comm,
                         std::shared_ptr<DeviceExecutor> exec, bool symmetric,
                         CommunicationModel cm)
{
  const PetscRows rows
      = read_petsc_binary_rows(filename, comm->rank(), comm->size());
  const int64_t nrows_local = rows.row_end - rows.row_begin;
  const int64_t ncols_local = rows.col_end - rows.col_begin;
  Matrix<double>::Split s = Matrix<double>::split_rows(
      rows.rowptr.data(), rows.colind.data(), rows.values.data(), nrows_local,
      ncols_local, rows.row_begin, rows.col_begin, rows.col_ghosts, symmetric,
      cm);
  auto col_map = std::make_shared<L2GMap>(comm, ncols_local, s.col_ghosts, exec,
                                          cm); // :209-210
  auto row_map = std::make_shared<L2GMap>(comm, nrows_local,
                                          std::vector<int64_t>(), exec, cm);
  if (symmetric) // the file's nnz, read_petsc.cpp:219-221
    return std::make_unique<Matrix<double>>(s.local, s.remote, s.diagonal,
                                            col_map, row_map, rows.nnz_global,
                                            exec);
  if (col_map->overlapping())
    return std::make_unique<Matrix<double>>(s.local, s.remote, col_map, row_map,
                                            exec);
  return std::make_unique<Matrix<double>>(s.local, col_map, row_map, exec);
}

double* read_petsc_binary_vector(const Comm& comm, const DeviceExecutor* exec,
                                 const std::string& filename,
                                 int64_t* nrows_local_out)
{
  std::ifstream file(filename.c_str(), std::ios::in | std::ios::binary);
  if (!file.is_open())
    throw std::runtime_error("Could not open file"); // :299
  const std::vector<unsigned char> head = read_bytes(file, 0, 8);
  if (be32(head.data()) != kVectorId)
    throw std::runtime_error("Bad signature in PETSc Vector file"); // :260
  const int64_t nrows = be32(head.data() + 4);
  const std::vector<int64_t> ranges = owner_ranges(comm.size(), nrows);
  const int64_t r0 = ranges[comm.rank()], r1 = ranges[comm.rank() + 1];
  const int64_t nloc = r1 - r0;
  const std::vector<unsigned char> raw
      = read_bytes(file, 8 + r0 * 8, static_cast<size_t>(nloc) * 8);
  std::vector<double> host(nloc);
  for (int64_t i = 0; i < nloc; ++i)
    host[i] = be64(raw.data() + 8 * i);
  double* dev = exec->alloc<double>(nloc); // :292-294
  exec->copy_from<double>(dev, exec->get_host(), host.data(), nloc);
  if (nrows_local_out)
    *nrows_local_out = nloc;
  return dev;
}

} // namespace spmv
