// spmv::cg for HipExecutor: see cg.h.  Algebra and update order follow
// spmv/cg.cpp:21-98; the launch structure is MI355X-specific:
//
//   per iteration (compute stream)            reference line
//     halo start on the map's side stream      cg.cpp:59
//     SpMV local block  (+ fused p.Ap share)   cg.cpp:60,63
//     [wait halo event] SpMV remote block      Matrix.cpp:498-511
//     reduce partials -> pAp[k]; all-reduce    cg.cpp:64-65
//     x += a p; r -= a Ap; partials of r.r     cg.cpp:66-73
//     reduce partials -> rr[k];  all-reduce    cg.cpp:74-76
//     stop test; p = beta p + r                cg.cpp:77-85
//
// 5 kernel launches + (multi-rank) 2 RCCL all-reduces of one double; the
// reference's CUDA path needs 7 cuBLAS calls, 5 scalar kernels and 3 host
// synchronisations for the same step (cuda/cg.cuda.cu:101-151).
#include "cg.h"

#include <cmath>
#include <stdexcept>

#include "spmv_hip.h"

namespace spmv
{

namespace
{
struct Workspace {
  HipExecutor& exec;
  spmv_hip_cg_ws* ws = nullptr;
  double *r = nullptr, *Ap = nullptr, *x = nullptr, *p = nullptr;
  double* dot2 = nullptr;   // partials of the remote block's p.Ap share
  int32_t* flags = nullptr; // pinned {done, kstop}
  void* stream = nullptr;
  void* prev_stream = nullptr;
  void* poll_event = nullptr;
  explicit Workspace(HipExecutor& e) : exec(e) {}
  ~Workspace()
  {
    try {
      exec.set_stream(prev_stream);
      if (stream)
        exec.synchronize_stream(stream);
      exec.destroy_event(poll_event);
      if (stream)
        exec.destroy_stream(stream);
      spmv_hip_cg_ws_destroy(ws);
      exec.free(r);
      exec.free(Ap);
      exec.free(x);
      exec.free(p);
      exec.free(dot2);
      spmv_hip_host_free(exec.context(), flags);
    } catch (...) {
    }
  }
};
} // namespace

int cg(const Comm& comm, HipExecutor& exec, const Matrix<double>& A,
       const double* b, double* x, int kmax, double rtol,
       std::vector<double>* rnorm_history, int poll_every)
{
  std::shared_ptr<const L2GMap> col_l2g = A.col_map();
  std::shared_ptr<const L2GMap> row_l2g = A.row_map();
  if (row_l2g->num_ghosts() > 0) // cg.cpp:32-33
    throw std::runtime_error("spmv::cg - Error: A.row_map() has ghost entries");
  if (kmax < 0)
    throw std::runtime_error("spmv::cg - Error: kmax < 0");
  if (poll_every < 1)
    poll_every = 1;

  const int64_t M = row_l2g->local_size();
  const int64_t N_padded = col_l2g->local_size() + col_l2g->num_ghosts();
  spmv_hip_ctx* ctx = exec.context();

  Workspace w(exec);
  w.prev_stream = exec.get_stream();
  w.stream = exec.create_stream();
  { // order after whatever the caller enqueued (b may still be in flight)
    void* ev = exec.create_event();
    exec.record_event(ev, w.prev_stream);
    exec.stream_wait_event(w.stream, ev);
    exec.destroy_event(ev);
  }
  exec.set_stream(w.stream); // every launch below goes to this stream

  throw_on_error(spmv_hip_cg_ws_create(ctx, kmax, &w.ws),
                 "spmv_hip_cg_ws_create");
  throw_on_error(spmv_hip_cg_ws_reset(w.ws, rtol, nullptr),
                 "spmv_hip_cg_ws_reset");
  int len = 0;
  throw_on_error(spmv_hip_dot_partials_len(ctx, &len),
                 "spmv_hip_dot_partials_len");
  double* partials = nullptr;
  throw_on_error(spmv_hip_cg_ws_partials(w.ws, &partials),
                 "spmv_hip_cg_ws_partials");

  // work vectors (cg.cpp:39-45); x0 = 0 and the ghost tail of p are defined
  // here instead of relying on fresh pages
  w.r = exec.alloc<double>(M);
  w.Ap = exec.alloc<double>(M);
  w.x = exec.alloc<double>(N_padded);
  w.p = exec.alloc<double>(N_padded);
  w.dot2 = exec.alloc<double>(len);
  exec.memset<double>(w.x, 0, N_padded);
  exec.memset<double>(w.p, 0, N_padded);
  exec.memset<double>(w.dot2, 0, len);
  exec.copy<double>(w.r, b, M);
  exec.copy<double>(w.p, b, M);
  void* flags_mem = nullptr;
  throw_on_error(spmv_hip_host_alloc(ctx, 2 * sizeof(int32_t), &flags_mem),
                 "spmv_hip_host_alloc");
  w.flags = static_cast<int32_t*>(flags_mem);
  w.flags[0] = 0;
  w.flags[1] = -1;
  w.poll_event = exec.create_event();

  auto slot = [&](bool rr, int k) {
    double* s = nullptr;
    throw_on_error(rr ? spmv_hip_cg_ws_rr(w.ws, k, &s)
                      : spmv_hip_cg_ws_pAp(w.ws, k, &s),
                   "spmv_hip_cg_ws slot");
    return s;
  };

  // rnorm0 (cg.cpp:47-50)
  throw_on_error(spmv_hip_cg_dot_rr_f64(ctx, w.ws, M, w.r, nullptr),
                 "spmv_hip_cg_dot_rr_f64");
  throw_on_error(spmv_hip_cg_reduce_rr(ctx, w.ws, 0, nullptr),
                 "spmv_hip_cg_reduce_rr");
  comm.allreduce_sum(slot(true, 0), 1, w.stream);

  int k = 0;
  bool stopped = false;
  bool poll_pending = false;
  while (k < kmax && !stopped) { // cg.cpp:55
    ++k;
    col_l2g->update(w.p); // cg.cpp:59 (starts on the side stream)
    // cg.cpp:60,63: Ap = A p with the p.Ap partials fused into the kernels
    const bool fused = A.mult_dot(w.p, w.Ap, partials, w.dot2);
    if (fused) {
      // local + remote shares -> pAp[k]
      throw_on_error(spmv_hip_cg_reduce_pAp2(ctx, w.ws, k, w.dot2, nullptr),
                     "spmv_hip_cg_reduce_pAp2");
    } else {
      throw_on_error(spmv_hip_dot_partial_f64(ctx, M, w.p, w.Ap, partials,
                                              nullptr),
                     "spmv_hip_dot_partial_f64");
      throw_on_error(spmv_hip_cg_reduce_pAp(ctx, w.ws, k, nullptr),
                     "spmv_hip_cg_reduce_pAp");
    }
    comm.allreduce_sum(slot(false, k), 1, w.stream); // cg.cpp:65
    throw_on_error(spmv_hip_cg_update_xr_f64(ctx, w.ws, k, M, w.p, w.Ap, w.x,
                                             w.r, nullptr),
                   "spmv_hip_cg_update_xr_f64"); // cg.cpp:66-73
    throw_on_error(spmv_hip_cg_reduce_rr(ctx, w.ws, k, nullptr),
                   "spmv_hip_cg_reduce_rr");
    comm.allreduce_sum(slot(true, k), 1, w.stream); // cg.cpp:75
    throw_on_error(spmv_hip_cg_update_p_f64(ctx, w.ws, k, M, w.r, w.p, nullptr),
                   "spmv_hip_cg_update_p_f64"); // cg.cpp:77-85

    if (k % poll_every == 0 && k < kmax) {
      // Lagging look at the flag: wait for the copy issued `poll_every`
      // iterations ago (bounds the host's run-ahead, never drains the queue),
      // then issue the next one.
      if (poll_pending) {
        exec.synchronize_event(w.poll_event);
        stopped = w.flags[0] != 0;
      }
      if (!stopped) {
        throw_on_error(spmv_hip_cg_ws_read_async(w.ws, w.flags, nullptr,
                                                 nullptr),
                       "spmv_hip_cg_ws_read_async");
        exec.record_event(w.poll_event, w.stream);
        poll_pending = true;
      }
    }
  }

  // final state: {done, kstop} and the squared-residual history
  std::vector<double> rr(kmax + 1, 0.0);
  throw_on_error(spmv_hip_cg_ws_read_async(w.ws, w.flags, rr.data(), nullptr),
                 "spmv_hip_cg_ws_read_async");
  exec.copy<double>(x, w.x, M); // cg.cpp:89
  exec.synchronize_stream(w.stream);

  int k_final = k;
  if (w.flags[0] != 0) {
    k_final = w.flags[1];
  } else {
    // `done` is raised by the p.Ap reducer of the NEXT iteration; when the
    // loop ends first, apply the same test (cg.cpp:80) to the last entry on
    // the host.  Either way the value returned is the reference's k.
    const double rnorm0 = std::sqrt(rr[0]);
    for (int j = 1; j <= k; ++j)
      if (std::sqrt(rr[j]) / rnorm0 < rtol) {
        k_final = j;
        break;
      }
  }
  if (rnorm_history) {
    rnorm_history->resize(k_final + 1);
    for (int j = 0; j <= k_final; ++j)
      (*rnorm_history)[j] = std::sqrt(rr[j]);
  }
  return k_final;
}

} // namespace spmv
