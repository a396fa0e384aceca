// Conjugate Gradient on the MI355X backend, mirroring the reference's
// per-executor overload set of spmv::cg (spmv/cg.h, spmv/cuda/cg_cuda.h:30-32).
#pragma once

#include <vector>

#include "comm.h"
#include "executor.h"
#include "matrix.h"

namespace spmv
{

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
// If rnorm_history != nullptr it receives ||r_0||, ..., ||r_k||.
int cg(const Comm& comm, HipExecutor& exec, const Matrix<double>& A,
       const double* b, double* x, int kmax, double rtol,
       std::vector<double>* rnorm_history = nullptr, int poll_every = 16);

} // namespace spmv
