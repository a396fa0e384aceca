// Conjugate Gradient on the MI355X backend, mirroring the reference's
// per-executor overload set of spmv::cg (spmv/cg.h, spmv/cuda/cg_cuda.h:30-32).
#pragma once

#include <cstdint>
#include <vector>

#include "comm.h"
#include "executor.h"
#include "matrix.h"

struct spmv_hip_cg_ws;

namespace spmv
{

// Work vectors + device scalars of one solve (cg.cpp:39-42 allocates and
// frees them on every call).  Passing the same CgWorkspace to repeated cg()
// calls keeps the allocations; it regrows itself when a call needs more.
class CgWorkspace
{
public:
  explicit CgWorkspace(HipExecutor& exec) : _exec(exec) {}
  ~CgWorkspace();
  CgWorkspace(const CgWorkspace&) = delete;
  CgWorkspace& operator=(const CgWorkspace&) = delete;

  // ---- internal to cg() ----
  void ensure(int64_t M, int64_t N_padded, int kmax, int partials_len);
  // events for CgOptions::time_spmv of a solve of up to `iterations` steps,
  // created ahead of it (a benchmark keeps them out of its timed region)
  void reserve_timing(int iterations);
  void release();

  HipExecutor& _exec;
  spmv_hip_cg_ws* ws = nullptr;
  int kmax_cap = -1;
  int64_t m_cap = -1, n_cap = -1;
  double *r = nullptr, *Ap = nullptr, *x = nullptr, *p = nullptr;
  double* dot2 = nullptr;   // partials of the remote block's p.Ap share
  int32_t* flags = nullptr; // pinned {done, kstop}
  void* stream = nullptr;   // compute stream of the solve
  void* poll_event = nullptr;
  std::vector<void*> timing_ev; // CgOptions::time_spmv: 2 events per iteration
};

struct CgOptions {
  int poll_every = 16;    // host looks at the device's `done` flag this often
  bool time_spmv = false; // bracket every local-block SpMV with HIP events
  // One rank only (no all-reduce between producer and consumer): the update
  // kernels add the dot-product partials themselves, in the reducers' order,
  // so an iteration is 3 launches and the scalars keep their bits.  With more
  // than one rank (or when switched off) every dot product is finished by a
  // single-workgroup reducer kernel (5 launches per iteration).
  bool consumer_reductions = true;
  // Mixed precision: the SpMV of every iteration streams an
  // fp32 copy of the matrix values (half the matrix bytes; x, p, r and all
  // arithmetic stay fp64).  Every `replace_every` iterations the recurrence
  // residual is replaced by the true one, r = b - A x, computed with the fp64
  // values (residual replacement keeps the single Krylov sequence); when the
  // loop ends the true residual is checked once more and, if it misses rtol,
  // the correction equation A d = r is solved with the fp64 values and added
  // (CgStats reports both).  General storage only; ignored for symmetric.
  bool mixed = false;
  int replace_every = 50;
};

struct CgStats {
  int spmv_launches = 0;    // local-block SpMV kernels timed
  double spmv_ms_total = 0; // sum of their durations (HIP events, same stream)
  // CgOptions::mixed
  int replacements = 0;             // residual replacements inside the loop
  double true_rel_residual = -1.0;  // ||b - A x|| / ||r_0|| with fp64 values,
                                    // at the end of the mixed loop
  int continuation_iterations = 0;  // fp64 iterations of the correction solve
  double final_true_rel_residual = -1.0; // ... after it (= the above if none)
};

// Unpreconditioned CG from x0 = 0 (spmv/cg.cpp:21-98).  `b` and `x` are
// DEVICE pointers of A.row_map()->local_size() doubles (cuda/cg.cuda.cu:70).
// Stops when k == kmax or ||r_k|| / ||r_0|| < rtol; returns k.
//
// All scalars (alpha, beta, the residual history) stay on the device; the
// host enqueues iterations without waiting and only looks at a pinned flag
// every `poll_every` iterations to stop enqueuing once the device has
// declared convergence.  Kernels issued after convergence are no-ops, so x
// is exactly the iterate of the returned k.
//
// `x` IS the iterate (cg.cpp keeps a padded work vector and copies it out at
// the end, :89): it is zeroed at the start of the solve and updated in place
// from iteration 1 on, so
//   * `x` must not overlap `b` (std::runtime_error; the reference would
//     tolerate x == b because it writes x only once, after the loop), and
//   * if the solve throws, `x` holds whatever iterate had been reached --
//     unlike the reference it is not left untouched.
// (An `x` that is not 16-byte aligned, and every mixed-precision solve, go
// through the workspace's own vector and one copy at the end instead.)
//
// If rnorm_history != nullptr it receives ||r_0||, ..., ||r_k||.
int cg(const Comm& comm, HipExecutor& exec, const Matrix<double>& A,
       const double* b, double* x, int kmax, double rtol,
       std::vector<double>* rnorm_history = nullptr,
       const CgOptions* options = nullptr, CgStats* stats = nullptr,
       CgWorkspace* workspace = nullptr);

} // namespace spmv
