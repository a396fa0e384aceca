// spmv::cg for HipExecutor: see cg.h.  Algebra and update order follow
// spmv/cg.cpp:21-98; the launch structure is MI355X-specific:
//
//   per iteration (compute stream)            reference line
//     halo start on the map's side stream      cg.cpp:59
//     SpMV local block  (+ fused p.Ap share)   cg.cpp:60,63
//     [wait halo event] SpMV remote block      Matrix.cpp:498-511
//     reduce partials -> pAp[k]; all-reduce    cg.cpp:64-65
//     r -= a Ap; partials of r.r               cg.cpp:66,70,73
//     reduce partials -> rr[k];  all-reduce    cg.cpp:74-76
//     x += a p; stop test; p = beta p + r      cg.cpp:69,77-85
//
// The partial sums of a dot product are added in a fixed order either by the
// consuming update kernel itself (one rank) or by a single-workgroup reducer
// kernel.  3 (or 5) kernel launches per iteration (+ the small remote-block
// kernel and two one-double RCCL all-reduces with more than one rank); the
// reference's CUDA path needs 7 cuBLAS calls, 5 scalar kernels and 3 host
// synchronisations for the same step (cuda/cg.cuda.cu:101-151).
#include "cg.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "spmv_hip.h"

namespace spmv
{

// ---------------------------------------------------------------------------
CgWorkspace::~CgWorkspace() { release(); }

void CgWorkspace::release()
{
  try {
    if (stream)
      _exec.synchronize_stream(stream);
    _exec.destroy_event(poll_event);
    for (void* e : timing_ev)
      _exec.destroy_event(e);
    if (stream)
      _exec.destroy_stream(stream);
    spmv_hip_cg_ws_destroy(ws);
    _exec.free(r);
    _exec.free(Ap);
    _exec.free(x);
    _exec.free(p);
    _exec.free(dot2);
    spmv_hip_host_free(_exec.context(), flags);
  } catch (...) {
  }
  timing_ev.clear();
  ws = nullptr;
  r = Ap = x = p = dot2 = nullptr;
  flags = nullptr;
  stream = poll_event = nullptr;
  kmax_cap = -1;
  m_cap = n_cap = -1;
}

void CgWorkspace::ensure(int64_t M, int64_t N_padded, int kmax, int len)
{
  spmv_hip_ctx* ctx = _exec.context();
  if (!stream) {
    stream = _exec.create_stream();
    poll_event = _exec.create_event();
    void* mem = nullptr;
    throw_on_error(spmv_hip_host_alloc(ctx, 2 * sizeof(int32_t), &mem),
                   "spmv_hip_host_alloc");
    flags = static_cast<int32_t*>(mem);
    dot2 = _exec.alloc<double>(len);
  }
  if (kmax > kmax_cap) {
    spmv_hip_cg_ws_destroy(ws);
    ws = nullptr;
    throw_on_error(spmv_hip_cg_ws_create(ctx, kmax, &ws),
                   "spmv_hip_cg_ws_create");
    kmax_cap = kmax;
  }
  if (M > m_cap) {
    _exec.free(r);
    _exec.free(Ap);
    r = _exec.alloc<double>(M); // cg.cpp:39-40
    Ap = _exec.alloc<double>(M);
    m_cap = M;
  }
  if (N_padded > n_cap) {
    _exec.free(x);
    _exec.free(p);
    x = _exec.alloc<double>(N_padded); // cg.cpp:41-42
    p = _exec.alloc<double>(N_padded);
    n_cap = N_padded;
  }
}

void CgWorkspace::reserve_timing(int iterations)
{
  while (timing_ev.size() < 2 * (size_t)(iterations < 0 ? 0 : iterations))
    timing_ev.push_back(_exec.create_event(true));
}

namespace
{
// restores the executor's stream when cg() leaves, also on exceptions
struct StreamGuard {
  HipExecutor& exec;
  void* prev;
  ~StreamGuard()
  {
    try {
      exec.set_stream(prev);
    } catch (...) {
    }
  }
};

} // namespace

int cg(const Comm& comm, HipExecutor& exec, const Matrix<double>& A,
       const double* b, double* x, int kmax, double rtol,
       std::vector<double>* rnorm_history, const CgOptions* options,
       CgStats* stats, CgWorkspace* workspace)
{
  std::shared_ptr<const L2GMap> col_l2g = A.col_map();
  std::shared_ptr<const L2GMap> row_l2g = A.row_map();
  if (row_l2g->num_ghosts() > 0) // cg.cpp:32-33
    throw std::runtime_error("spmv::cg - Error: A.row_map() has ghost entries");
  if (kmax < 0)
    throw std::runtime_error("spmv::cg - Error: kmax < 0");
  const CgOptions opt = options ? *options : CgOptions();
  const int poll_every = opt.poll_every < 1 ? 1 : opt.poll_every;

  const int64_t M = row_l2g->local_size();
  const int64_t N_padded = col_l2g->local_size() + col_l2g->num_ghosts();
  spmv_hip_ctx* ctx = exec.context();
  int len = 0;
  throw_on_error(spmv_hip_dot_partials_len(ctx, &len),
                 "spmv_hip_dot_partials_len");

  { // x is the iterate from the first kernel on (cg.h): it cannot share b
    const uintptr_t xb = reinterpret_cast<uintptr_t>(x),
                    bb = reinterpret_cast<uintptr_t>(b);
    const uintptr_t bytes = (uintptr_t)M * sizeof(double);
    if (M > 0 && xb < bb + bytes && bb < xb + bytes)
      throw std::runtime_error("cg: x overlaps b (x is updated in place)");
  }
  CgWorkspace own(exec);
  CgWorkspace& w = workspace ? *workspace : own;
  w.ensure(M, N_padded, kmax, len);

  StreamGuard guard{exec, exec.get_stream()};
  { // order after whatever the caller enqueued (b may still be in flight)
    void* ev = exec.create_event();
    exec.record_event(ev, guard.prev);
    exec.stream_wait_event(w.stream, ev);
    exec.destroy_event(ev);
  }
  exec.set_stream(w.stream); // every launch below goes to this stream

  throw_on_error(spmv_hip_cg_ws_reset(w.ws, rtol, nullptr),
                 "spmv_hip_cg_ws_reset");
  double* partials = nullptr;
  throw_on_error(spmv_hip_cg_ws_partials(w.ws, &partials),
                 "spmv_hip_cg_ws_partials");

  // The iterate lives in the caller's x (no copy at the end, cg.cpp:89) unless
  // the mixed mode needs its halo (then in the padded work vector).
  const bool mixed = opt.mixed && A.enable_mixed();
  const bool x_aligned = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  double* const xi = (mixed || !x_aligned) ? w.x : x;
  // r = p = b, x0 = 0, partials of r.r: one pass (cg.cpp:41-47; x0 and the
  // ghost tails are defined here instead of relying on fresh pages)
  if (N_padded > M) {
    exec.memset<double>(w.p + M, 0, N_padded - M);
    if (mixed)
      exec.memset<double>(w.x + M, 0, N_padded - M);
  }
  exec.memset<double>(w.dot2, 0, len);
  throw_on_error(spmv_hip_cg_init_f64(ctx, w.ws, M, b, w.r, w.p, xi, nullptr),
                 "spmv_hip_cg_init_f64");
  w.flags[0] = 0;
  w.flags[1] = -1;

  auto slot = [&](bool rr, int k) {
    double* s = nullptr;
    throw_on_error(rr ? spmv_hip_cg_ws_rr(w.ws, k, &s)
                      : spmv_hip_cg_ws_pAp(w.ws, k, &s),
                   "spmv_hip_cg_ws slot");
    return s;
  };

  // rnorm0 (cg.cpp:47-50)
  throw_on_error(spmv_hip_cg_reduce_rr(ctx, w.ws, 0, nullptr),
                 "spmv_hip_cg_reduce_rr");
  comm.reduce_sum(slot(true, 0), 1, w.stream);

  const bool consume = opt.consumer_reductions && comm.size() == 1;
  // whatever happens below, leave the matrix in fp64 mode
  struct MixedGuard {
    const Matrix<double>& A;
    ~MixedGuard() { A.use_mixed(false); }
  } mixed_guard{A};
  int replacements = 0;
  // Timing events live in the workspace: a solve that reuses one (the
  // benchmark, after its warm-up) creates nothing inside its timed region.
  std::vector<void*>& timing_ev = w.timing_ev;
  if (opt.time_spmv)
    w.reserve_timing(kmax);
  int k = 0;
  bool stopped = false;
  bool poll_pending = false;
  while (k < kmax && !stopped) { // cg.cpp:55
    ++k;
    col_l2g->update(w.p); // cg.cpp:59 (starts on the side stream)
    void* ev1 = nullptr;
    if (opt.time_spmv) {
      ev1 = timing_ev[2 * (size_t)(k - 1) + 1];
      exec.record_event(timing_ev[2 * (size_t)(k - 1)], w.stream);
    }
    // cg.cpp:60,63: Ap = A p with the p.Ap partials produced by the SpMV
    // kernels themselves (local block's share + remote block's share)
    const bool replace
        = mixed && opt.replace_every > 0 && k % opt.replace_every == 0;
    A.use_mixed(mixed);
    if (replace) {
      // residual replacement: the usual iteration in the reference's grouping
      // (x first), then r := b - A x with the fp64 values instead of the
      // recurrence, rr[k] from it, p from both
      const bool fused = A.mult_dot(w.p, w.Ap, partials, w.dot2, ev1);
      if (fused) {
        throw_on_error(spmv_hip_cg_reduce_pAp2(ctx, w.ws, k, w.dot2, nullptr),
                       "spmv_hip_cg_reduce_pAp2");
      } else {
        throw_on_error(spmv_hip_dot_partial_f64(ctx, M, w.p, w.Ap, partials,
                                                nullptr),
                       "spmv_hip_dot_partial_f64");
        throw_on_error(spmv_hip_cg_reduce_pAp(ctx, w.ws, k, nullptr),
                       "spmv_hip_cg_reduce_pAp");
      }
      comm.reduce_sum(slot(false, k), 1, w.stream);
      throw_on_error(spmv_hip_cg_update_xr_f64(ctx, w.ws, k, M, w.p, w.Ap, xi,
                                               w.r, nullptr),
                     "spmv_hip_cg_update_xr_f64");
      A.use_mixed(false);
      col_l2g->update(xi);
      A.mult(xi, w.Ap);
      throw_on_error(spmv_hip_cg_residual_f64(ctx, w.ws, 1, M, b, w.Ap, w.r,
                                              nullptr),
                     "spmv_hip_cg_residual_f64");
      throw_on_error(spmv_hip_cg_reduce_rr(ctx, w.ws, k, nullptr),
                     "spmv_hip_cg_reduce_rr");
      comm.reduce_sum(slot(true, k), 1, w.stream);
      throw_on_error(spmv_hip_cg_update_p_f64(ctx, w.ws, k, M, w.r, w.p,
                                              nullptr),
                     "spmv_hip_cg_update_p_f64");
      ++replacements;
    } else if (consume) {
      // one rank: the update kernels add the partials themselves
      const bool fused = A.mult_dot(w.p, w.Ap, partials, w.dot2, ev1);
      if (!fused)
        throw_on_error(spmv_hip_dot_partial_f64(ctx, M, w.p, w.Ap, partials,
                                                nullptr),
                       "spmv_hip_dot_partial_f64");
      throw_on_error(spmv_hip_cg_update_r_cs_f64(ctx, w.ws, k, M, w.Ap, w.r,
                                                 fused ? w.dot2 : nullptr,
                                                 nullptr),
                     "spmv_hip_cg_update_r_cs_f64");
      throw_on_error(spmv_hip_cg_update_xp_cs_f64(ctx, w.ws, k, M, w.r, xi,
                                                  w.p, nullptr),
                     "spmv_hip_cg_update_xp_cs_f64");
    } else {
      const bool fused = A.mult_dot(w.p, w.Ap, partials, w.dot2, ev1);
      if (fused) {
        throw_on_error(spmv_hip_cg_reduce_pAp2(ctx, w.ws, k, w.dot2, nullptr),
                       "spmv_hip_cg_reduce_pAp2");
      } else {
        throw_on_error(spmv_hip_dot_partial_f64(ctx, M, w.p, w.Ap, partials,
                                                nullptr),
                       "spmv_hip_dot_partial_f64");
        throw_on_error(spmv_hip_cg_reduce_pAp(ctx, w.ws, k, nullptr),
                       "spmv_hip_cg_reduce_pAp");
      }
      comm.reduce_sum(slot(false, k), 1, w.stream); // cg.cpp:65
      // r -= alpha Ap with the r.r partials (cg.cpp:66,70,73); the x update
      // of :69 rides with the p update below so p is read once per iteration
      throw_on_error(spmv_hip_cg_update_r_f64(ctx, w.ws, k, M, w.Ap, w.r,
                                              nullptr),
                     "spmv_hip_cg_update_r_f64");
      throw_on_error(spmv_hip_cg_reduce_rr(ctx, w.ws, k, nullptr),
                     "spmv_hip_cg_reduce_rr");
    }
    if (!consume && !replace) {
      comm.reduce_sum(slot(true, k), 1, w.stream); // cg.cpp:75
      // x += alpha p ; stop test ; p = beta p + r   (cg.cpp:69,77-85)
      throw_on_error(spmv_hip_cg_update_xp_f64(ctx, w.ws, k, M, w.r, xi, w.p,
                                               nullptr),
                     "spmv_hip_cg_update_xp_f64");
    }

    if (k % poll_every == 0 && k < kmax) {
      // Lagging look at the flag: wait for the copy issued `poll_every`
      // iterations ago (bounds the host's run-ahead, never drains the queue),
      // then issue the next one.
      if (poll_pending) {
        exec.synchronize_event(w.poll_event);
        stopped = w.flags[0] != 0;
      }
      if (!stopped) {
        throw_on_error(spmv_hip_cg_ws_read_async(w.ws, w.flags, nullptr, 0,
                                                 nullptr),
                       "spmv_hip_cg_ws_read_async");
        exec.record_event(w.poll_event, w.stream);
        poll_pending = true;
      }
    }
  }

  // final state: {done, kstop} and the squared-residual history
  // (the device history has the WORKSPACE's capacity, which an earlier solve
  // with a larger kmax may have set: the copy is that long, and the C ABI
  // refuses a shorter destination)
  int cap = 0;
  throw_on_error(spmv_hip_cg_ws_capacity(w.ws, &cap), "spmv_hip_cg_ws_capacity");
  std::vector<double> rr((size_t)std::max(kmax, cap) + 1, 0.0);
  throw_on_error(spmv_hip_cg_ws_read_async(w.ws, w.flags, rr.data(), rr.size(),
                                           nullptr),
                 "spmv_hip_cg_ws_read_async");
  double true_rr = -1.0;
  if (mixed) {
    // the true residual of what the mixed loop produced, with the fp64 values
    A.use_mixed(false);
    col_l2g->update(xi);
    A.mult(xi, w.Ap);
    throw_on_error(spmv_hip_cg_residual_f64(ctx, w.ws, 0, M, b, w.Ap, w.r,
                                            nullptr),
                   "spmv_hip_cg_residual_f64");
    throw_on_error(spmv_hip_reduce_partials_f64(ctx, partials, w.dot2, nullptr),
                   "spmv_hip_reduce_partials_f64");
    comm.reduce_sum(w.dot2, 1, w.stream);
    throw_on_error(spmv_hip_copy_d2h_async(ctx, &true_rr, w.dot2,
                                           sizeof(double), nullptr),
                   "spmv_hip_copy_d2h_async");
  }
  if (xi != x)
    exec.copy<double>(x, xi, M); // cg.cpp:89
  exec.synchronize_stream(w.stream);

  if (stats) {
    stats->spmv_launches = 0;
    stats->spmv_ms_total = 0.0;
    for (size_t i = 0; opt.time_spmv && i + 1 < 2 * (size_t)k; i += 2) {
      float ms = 0.f;
      throw_on_error(spmv_hip_event_elapsed_ms(ctx, timing_ev[i],
                                               timing_ev[i + 1], &ms),
                     "spmv_hip_event_elapsed_ms");
      stats->spmv_ms_total += ms;
      ++stats->spmv_launches;
    }
  }

  int k_final = k;
  if (w.flags[0] != 0) {
    k_final = w.flags[1];
  } else {
    // `done` is raised by the p.Ap reducer of the NEXT iteration; when the
    // loop ends first, apply the same test (cg.cpp:80) to the history on the
    // host.  Either way the value returned is the reference's k.
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
  if (mixed) {
    const double rnorm0 = std::sqrt(rr[0]);
    const double true_rel = rnorm0 > 0 ? std::sqrt(true_rr) / rnorm0 : 0.0;
    int extra = 0;
    double final_rel = true_rel;
    if (true_rel >= rtol && rtol > 0 && k_final < kmax) {
      // The fp32 values took the iteration as far as they could: solve the
      // correction equation A d = b - A x with the fp64 values (w.r still
      // holds that residual) and add it.  A fresh workspace: this one's
      // vectors are the operands.
      A.use_mixed(false);
      double* d = exec.alloc<double>(M);
      double* rhs = exec.alloc<double>(M);
      exec.copy<double>(rhs, w.r, M);
      exec.synchronize_stream(w.stream);
      exec.set_stream(guard.prev);
      std::vector<double> h2;
      CgOptions o2 = opt;
      o2.mixed = false;
      o2.time_spmv = false;
      try {
        extra = cg(comm, exec, A, rhs, d, kmax - k_final, rtol / true_rel, &h2,
                   &o2, nullptr, nullptr);
        throw_on_error(spmv_hip_axpy_f64(ctx, M, 1.0, d, x, nullptr),
                       "spmv_hip_axpy_f64");
        exec.synchronize();
      } catch (...) {
        exec.free(d);
        exec.free(rhs);
        throw;
      }
      exec.free(d);
      exec.free(rhs);
      if (!h2.empty() && rnorm0 > 0)
        final_rel = h2.back() / rnorm0;
      if (rnorm_history)
        for (size_t j = 1; j < h2.size(); ++j)
          rnorm_history->push_back(h2[j]);
    }
    if (stats) {
      // iterations enqueued past the converged one were no-ops on the device
      stats->replacements
          = opt.replace_every > 0 ? k_final / opt.replace_every : 0;
      (void)replacements;
      stats->true_rel_residual = true_rel;
      stats->continuation_iterations = extra;
      stats->final_true_rel_residual = final_rel;
    }
    return k_final + extra;
  }
  return k_final;
}

} // namespace spmv
