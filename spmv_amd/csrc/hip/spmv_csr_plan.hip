// The plan API of the CSR SpMV (gfx950): spmv_hip_csr_plan_create / _destroy /
// _bake_values_* / _values_changed / _owns_matrix / _release_matrix / _set /
// _get -- CSRSpMV<T>::init / finalize (spmv/csr_kernels.h:26-78) and the knobs
// of the plan.  Split from spmv_csr.hip in round 6: the builders it calls are in
// spmv_csr_forms.hip and the files of the individual forms, the kernels and
// the launch entry points in spmv_csr.hip.
#include "csr_plan.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

thread_local int64_t spmv_plan_mem_ns = 0; // (plan_malloc.h)

namespace
{
// files what the memory calls of one plan-API call took under the plan
struct MemClock {
  spmv_hip_csr_plan* pl;
  int64_t t0;
  explicit MemClock(spmv_hip_csr_plan* p = nullptr) : pl(p), t0(spmv_plan_mem_ns) {}
  ~MemClock()
  {
    if (pl)
      pl->plan_mem_us += (int)((spmv_plan_mem_ns - t0) / 1000);
  }
};
} // namespace

// Symmetric storage of a matrix without lattice structure (FEM matrices, what
// read_petsc_binary_matrix delivers with symmetric = true): the reference's
// loop (csr_kernels.cpp:26-40) seen from the row, in the sliced jagged form of
// the MERGED matrix -- per row its stored lower entries, then the entries of
// its column in the reference's order (the transposed map has them).
// ENOTSUP: the form does not apply (the transposed-map kernel stays).
template <typename T>
static int sym_sj_bake(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan, const T* values,
                       const T* diagonal, hipStream_t st)
{
  auto bake = [&](spmv_hip_csr_plan* p, const T* v, const int32_t* map) {
    if constexpr (sizeof(T) == 8)
      return spmv_sjds_bake_f64(p, v, map, st);
    else
      return spmv_sjds_bake_f32(p, v, map, st);
  };
  if (values == nullptr) { // drop the copy
    plan->sym_sj = 0;
    plan->sj = 0;
    plan->sj_diag0 = nullptr;
    return plan->sjt && plan->sjt->sj_lenperm ? bake(plan->sjt, nullptr, nullptr)
                                              : SPMV_HIP_ENOTSUP;
  }
  if (!plan->symmetric || !plan->sym_det || !plan->t_ptr || plan->slat
      || plan->nnz < ctx->sj_min_nnz || plan->num_rows < 64 || !diagonal)
    return SPMV_HIP_ENOTSUP;
  const auto t0 = std::chrono::steady_clock::now();
  if (!plan->sjt) {
    // the structure: the long rows' list (parent), the merged matrix sliced
    // jagged (child) -- spmv_sjds_plan.hip
    const int rs = spmv_sjds_sym_build(ctx, plan, st);
    if (rs != SPMV_HIP_OK)
      return rs;
  }
  plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                       std::chrono::steady_clock::now() - t0)
                       .count();
  const int rc = bake(plan->sjt, values, plan->sjv_map);
  plan->plan_us += plan->sjt->plan_us; // (the bake counted itself there)
  plan->sjt->plan_us = 0;
  if (rc != SPMV_HIP_OK) {
    plan->sym_sj = 0;
    return rc;
  }
  plan->sj_diag0 = diagonal;
  plan->sym_sj = 1;
  plan->sj = 1;
  return SPMV_HIP_OK;
}


// the arrays a launch with the baked pointers does not read (plan_owns_matrix)
int spmv_plan_owned_mask(const spmv_hip_csr_plan* pl)
{
  if (pl->nnz == 0)
    return 0;
  // symmetric storage in the merged sliced jagged form, no long rows (those
  // are streamed from the caller's arrays): the kernel reads the merged copy,
  // the caller's row pointer and diagonal
  if (pl->symmetric)
    return pl->sym_det && pl->sym_sj && pl->sj && pl->sjt && pl->sjt->sj_val
                   && pl->sjt->sj_values0 && pl->sj_nlong == 0 && pl->num_cols >= 2
               ? 3
               : 0;
  // an fp32 twin for the mixed SpMV: its launches may fall back to CSR order
  if (pl->sdia32_val || pl->wdia32_val || pl->sj_val32)
    return 0;
  if (pl->sdia && pl->sdia_val && pl->sdia_general && pl->sdia_values0)
    return 3;
  if (pl->wdia && pl->wdia_val && pl->wdia_values0)
    return 3;
  if (pl->sj && pl->sj_val && pl->sj_values0 && pl->sj_nlong == 0 && pl->num_cols >= 2)
    return 3;
  return 0;
}

extern "C" {

int spmv_hip_csr_plan_create(spmv_hip_ctx* ctx, int32_t num_rows,
                             int32_t num_cols, int64_t num_non_zeros,
                             const int32_t* rowptr, const int32_t* colind,
                             int symmetric, int algo, spmv_hip_csr_plan** plan)
{
  SPMV_REQUIRE(ctx && plan && num_rows >= 0 && num_cols >= 0
               && num_non_zeros >= 0);
  SPMV_REQUIRE(num_non_zeros == 0 || (rowptr && colind));
  // (the plan's clock counts the plan's work, not kernels of the caller still
  // running on the stream -- a device-side generator's fill, say)
  if (num_non_zeros > 0) {
    SPMV_SET_DEVICE(ctx);
    SPMV_CHECK_HIP(hipStreamSynchronize(spmv_stream(ctx, nullptr)));
  }
  const auto t_begin = std::chrono::steady_clock::now();
  // rowptr is int32 in the reference format (csr_kernels.h:28)
  if (num_non_zeros > INT32_MAX)
    return SPMV_HIP_ERANGE;
  spmv_hip_csr_plan* pl = new (std::nothrow) spmv_hip_csr_plan;
  if (!pl)
    return SPMV_HIP_ENOMEM;
  pl->ctx = ctx;
  MemClock mem_clock(pl); // (a failed creation destroys pl before this is read:
                          // every such path below hands the clock a null plan)
  pl->num_rows = num_rows;
  pl->num_cols = num_cols;
  pl->nnz = num_non_zeros;
  pl->symmetric = symmetric != 0;
  pl->rowptr0 = rowptr;
  pl->colind0 = colind;
  const double avg = num_rows > 0 ? (double)num_non_zeros / num_rows : 0.0;
  if (algo == SPMV_HIP_ALGO_AUTO) {
    // fewer entries than a quarter of the rows: most rows are empty, walk
    // only the non-empty ones (the remote block of a partitioned matrix)
    if (!symmetric && num_non_zeros > 0 && num_non_zeros * 4 < num_rows)
      algo = SPMV_HIP_ALGO_ROWLIST;
    else // long rows: the sliced jagged form (below) where it is built,
         // else a sub-wavefront per row
      algo = (avg <= 64.0 || num_non_zeros >= ctx->sj_min_nnz)
                 ? SPMV_HIP_ALGO_ROWBLOCK
                 : SPMV_HIP_ALGO_VECTOR;
  }
  if (algo < SPMV_HIP_ALGO_ROWBLOCK || algo > SPMV_HIP_ALGO_ROWLIST
      || (algo == SPMV_HIP_ALGO_ROWLIST && (symmetric || num_non_zeros == 0))) {
    mem_clock.pl = nullptr;
    delete pl;
    return SPMV_HIP_EINVAL;
  }
  pl->algo = algo;
  if (algo == SPMV_HIP_ALGO_ROWLIST) {
    int rc = spmv_build_row_list(pl, rowptr);
    if (rc != SPMV_HIP_OK) {
      mem_clock.pl = nullptr;
      delete pl;
      return rc;
    }
  }
  int lpr = 4;
  while (lpr < 64 && lpr < avg / 2)
    lpr *= 2;
  pl->lanes_per_row = lpr;
  // Non-temporal matrix loads keep the read-once stream out of the caches so
  // that x stays resident; measured +5 % when x fits the 256 MiB Infinity
  // Cache with room to spare (216^3) and -3 % when it does not (512^3).
  pl->nontemporal = ((int64_t)num_cols * 8 <= (int64_t)128 << 20) ? 1 : 0;
  if (!symmetric && algo == SPMV_HIP_ALGO_ROWBLOCK) {
    // Lattice form first (spmv_lat.hip): when every row block's columns are
    // row + one of <= 8 constant offsets the kernel needs no index stream at
    // all.  Otherwise the LX form: from ctx->lx_min_nnz entries on (set-up
    // time, +2 B per entry of memory) and rows short enough for the plan
    // kernel's sort; measured faster than the gather kernel at every size
    // from 128^3 to 512^3.
    int rc = SPMV_HIP_OK;
    if (num_non_zeros >= ctx->lat_min_nnz && avg <= 8.0)
      rc = spmv_lat_build(pl, rowptr, colind);
    // (ctx option "csr_in_place": the plan makes no copy of the index or value
    // stream -- neither the LX form's offsets nor the sliced jagged arrays)
    const bool copies = !ctx->csr_in_place;
    if (rc == SPMV_HIP_OK && !pl->lat && copies && num_non_zeros >= ctx->lx_min_nnz
        && avg <= 16.0 && (int64_t)num_cols * 8 <= ctx->lx_max_x_bytes)
      rc = spmv_build_lx(pl, rowptr, colind);
    // Neither: the sliced jagged form (spmv_sjds.hip) -- ragged rows, more
    // than 16 entries per row, column windows too wide for the LX form -- is
    // built by plan_bake_values, structure and values together, once the
    // diagonal forms have refused the matrix (a 27-point stencil has 27
    // entries per row too, and its analysis would be 50 ms for nothing).
    pl->sj_wanted
        = !pl->lat && !pl->lx && copies && num_non_zeros >= ctx->sj_min_nnz;
    // Neither of them and no sliced jagged form to come: the caller's CSR
    // arrays as they are.  From ctx->xw_min_nnz entries on the XW kernel
    // (spmv_lxw.hip): values and the 32-bit column indices by LDS-DMA, the x
    // windows of every row block staged -- the gather kernel fetched x across
    // the fabric 3.5 times at 512^3 -- in the plane-walk order when the
    // matrix sits on a 3-D grid.  With default options a matrix XW can stage
    // is one the LX form can stage too (the same window analysis, 16 windows
    // against 8), so XW is what "csr_in_place" plans get, what is left when
    // the LX form's 2 B per entry could not be allocated, and -- below, in
    // plan_bake_values -- what a plan whose sliced jagged form was declined
    // runs instead of the gather kernel.
    if (rc == SPMV_HIP_OK && !pl->sj_wanted && spmv_xw_applies(pl))
      rc = spmv_build_xw_and_walk(pl, rowptr, colind);
    if (rc != SPMV_HIP_OK) {
      mem_clock.pl = nullptr;
      spmv_hip_csr_plan_destroy(pl);
      return rc;
    }
  }
  if (symmetric && num_non_zeros > 0) {
    // atomic-free, bit-exact forms (the default when the block is strictly
    // lower triangular): the symmetric lattice form when the matrix has it,
    // else the transposed map
    int rc = SPMV_HIP_OK;
    const bool trace = getenv("SPMV_PLAN_TRACE") != nullptr;
    auto mark = [&](const char* what) {
      if (trace)
        fprintf(stderr, "plan_create %-10s %8.3f ms\n", what,
                std::chrono::duration<double, std::milli>(
                    std::chrono::steady_clock::now() - t_begin)
                    .count());
    };
    mark("begin");
    if (num_non_zeros >= ctx->lat_min_nnz)
      rc = spmv_slat_build(pl, rowptr, colind);
    mark("slat");
    if (rc == SPMV_HIP_OK && !pl->slat)
      rc = spmv_symt_build(pl, rowptr, colind);
    mark("symt");
    if (rc != SPMV_HIP_OK) {
      mem_clock.pl = nullptr;
      spmv_hip_csr_plan_destroy(pl);
      return rc;
    }
  }
  // what the analysis cost (every builder has synchronised its stream)
  pl->plan_us = (int)std::chrono::duration_cast<std::chrono::microseconds>(
                    std::chrono::steady_clock::now() - t_begin)
                    .count();
  *plan = pl;
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_destroy(spmv_hip_csr_plan* plan)
{
  if (plan && plan->sjt) {
    (void)hipSetDevice(plan->ctx->device);
    spmv_sjds_free(plan->sjt);
    delete plan->sjt;
    plan->sjt = nullptr;
    (void)hipFree(plan->sjv_ptr);
    (void)hipFree(plan->sjv_col);
    (void)hipFree(plan->sjv_map);
    plan->sjv_ptr = plan->sjv_col = plan->sjv_map = nullptr;
  }
  if (plan
      && (plan->row_list || plan->lx_lidx || plan->lat_tab || plan->t_ptr
          || plan->slat_mask || plan->zw_table || plan->wdia_val
          || plan->sdia_val || plan->sj_lenperm || plan->xw_rec)) {
    (void)hipSetDevice(plan->ctx->device);
    (void)hipFree(plan->row_list);
    spmv_free_lx(plan);
    spmv_free_xw(plan);
    spmv_lat_free(plan);
    spmv_symt_free(plan);
    spmv_sdia_free(plan);
    spmv_wdia_free(plan);
    spmv_sjds_free(plan);
    spmv_slat_free(plan);
    spmv_zwalk_free(plan);
  }
  delete plan;
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_bake_values_f64(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                      const double* values,
                                      const double* diagonal, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  MemClock mem_clock(plan);
  SPMV_REQUIRE(!plan->released); // (its source arrays were given up)
  hipStream_t st = spmv_stream(ctx, stream);
  if (values) // (as in plan_create: the caller's kernels are not plan time)
    SPMV_CHECK_HIP(hipStreamSynchronize(st));
  int rc = spmv_sdia_bake_f64(plan, values, diagonal, st);
  // a general matrix the diagonal form refuses (more than three lower
  // offsets, no lattice form): the wide diagonal form, up to 32 diagonals
  if (!plan->symmetric && (rc == SPMV_HIP_ENOTSUP || values == nullptr)) {
    const int rw = spmv_wdia_bake_f64(plan, values, st);
    rc = values == nullptr ? (rw != SPMV_HIP_OK ? rw : rc) : rw;
  } else if (!plan->symmetric && rc == SPMV_HIP_OK) {
    (void)spmv_wdia_bake_f64(plan, nullptr, st); // superseded
  }
  // a plan in the sliced jagged form keeps its own copy of the values in that
  // order (the diagonal forms never coexist with it)
  // (not from plan_values_changed: that call builds no new form and allocates
  // nothing -- a matrix the diagonal forms no longer hold goes back to the
  // CSR-order kernels, as its contract says)
  if (!plan->symmetric && values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted
      && !plan->sj_lenperm && !plan->no_new_forms) {
    // the structure of the sliced jagged form, now that it is known to be used
    const auto t0 = std::chrono::steady_clock::now();
    const int rb = spmv_sjds_build(plan, plan->rowptr0, plan->colind0,
                                   ctx->sj_wpb, ctx->sj_unit, 0);
    if (rb != SPMV_HIP_OK)
      return rb;
    plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                         std::chrono::steady_clock::now() - t0)
                         .count();
  }
  if (!plan->symmetric && plan->sj_lenperm
      && (values == nullptr || rc == SPMV_HIP_ENOTSUP)) {
    const int rj = spmv_sjds_bake_f64(plan, values, nullptr, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  // the sliced jagged form was wanted and could not be had (rows too long for
  // its length field, no memory for the copy): stage the x windows over the
  // caller's arrays rather than gather, where that applies
  if (values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted && !plan->sj_lenperm
      && !plan->no_new_forms) {
    plan->sj_wanted = false; // (declined: later bakes do not analyse it again)
    if (!plan->xw_rec && spmv_xw_applies(plan)) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rx = spmv_build_xw_and_walk(plan, plan->rowptr0, plan->colind0);
      if (rx != SPMV_HIP_OK)
        return rx;
      plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                           std::chrono::steady_clock::now() - t0)
                           .count();
    }
  }
  // symmetric storage without lattice structure: both blocks sliced jagged
  if (plan->symmetric && (values == nullptr ? plan->sjt != nullptr
                                            : rc == SPMV_HIP_ENOTSUP)) {
    const int rj = sym_sj_bake<double>(ctx, plan, values, diagonal, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  return rc;
}

int spmv_hip_csr_plan_bake_values_f32(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                      const float* values, const float* diagonal,
                                      void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  MemClock mem_clock(plan);
  SPMV_REQUIRE(!plan->released); // (its source arrays were given up)
  hipStream_t st = spmv_stream(ctx, stream);
  if (values)
    SPMV_CHECK_HIP(hipStreamSynchronize(st));
  int rc = spmv_sdia_bake_f32(plan, values, diagonal, st);
  if (!plan->symmetric && (rc == SPMV_HIP_ENOTSUP || values == nullptr)) {
    const int rw = spmv_wdia_bake_f32(plan, values, st);
    rc = values == nullptr ? (rw != SPMV_HIP_OK ? rw : rc) : rw;
  } else if (!plan->symmetric && rc == SPMV_HIP_OK) {
    (void)spmv_wdia_bake_f32(plan, nullptr, st);
  }
  // (not from plan_values_changed: that call builds no new form and allocates
  // nothing -- a matrix the diagonal forms no longer hold goes back to the
  // CSR-order kernels, as its contract says)
  if (!plan->symmetric && values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted
      && !plan->sj_lenperm && !plan->no_new_forms) {
    // the structure of the sliced jagged form, now that it is known to be used
    const auto t0 = std::chrono::steady_clock::now();
    const int rb = spmv_sjds_build(plan, plan->rowptr0, plan->colind0,
                                   ctx->sj_wpb, ctx->sj_unit, 0);
    if (rb != SPMV_HIP_OK)
      return rb;
    plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                         std::chrono::steady_clock::now() - t0)
                         .count();
  }
  if (!plan->symmetric && plan->sj_lenperm
      && (values == nullptr || rc == SPMV_HIP_ENOTSUP)) {
    const int rj = spmv_sjds_bake_f32(plan, values, nullptr, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  // the sliced jagged form was wanted and could not be had (rows too long for
  // its length field, no memory for the copy): stage the x windows over the
  // caller's arrays rather than gather, where that applies
  if (values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted && !plan->sj_lenperm
      && !plan->no_new_forms) {
    plan->sj_wanted = false; // (declined: later bakes do not analyse it again)
    if (!plan->xw_rec && spmv_xw_applies(plan)) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rx = spmv_build_xw_and_walk(plan, plan->rowptr0, plan->colind0);
      if (rx != SPMV_HIP_OK)
        return rx;
      plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                           std::chrono::steady_clock::now() - t0)
                           .count();
    }
  }
  if (plan->symmetric && (values == nullptr ? plan->sjt != nullptr
                                            : rc == SPMV_HIP_ENOTSUP)) {
    const int rj = sym_sj_bake<float>(ctx, plan, values, diagonal, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  return rc;
}

int spmv_hip_csr_plan_bake_values_f32f64(spmv_hip_ctx* ctx,
                                         spmv_hip_csr_plan* plan,
                                         const float* values32, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  MemClock mem_clock(plan);
  SPMV_REQUIRE(!plan->released); // (its source arrays were given up)
  hipStream_t st = spmv_stream(ctx, stream);
  // whichever form holds the fp64 values (by offset, in jagged order) gets
  // its fp32 twin
  if (plan->sj_val && !plan->symmetric && !plan->sdia_val && !plan->wdia_val)
    return spmv_sjds_bake_f32f64(plan, values32, st);
  if (plan->wdia_val && !plan->sdia_val)
    return spmv_wdia_bake_f32f64(plan, values32, st);
  if (values32 == nullptr) {
    (void)spmv_wdia_bake_f32f64(plan, nullptr, st);
    (void)spmv_sjds_bake_f32f64(plan, nullptr, st);
  }
  return spmv_sdia_bake_f32f64(plan, values32, st);
}

int spmv_hip_csr_plan_owns_matrix(const spmv_hip_csr_plan* plan, int* mask)
{
  SPMV_REQUIRE(plan && mask);
  *mask = spmv_plan_owned_mask(plan);
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_release_matrix(spmv_hip_csr_plan* plan, int mask)
{
  SPMV_REQUIRE(plan && mask >= 0 && (mask & ~spmv_plan_owned_mask(plan)) == 0);
  plan->released |= mask;
  if (plan->symmetric && plan->released == 3 && plan->sjt) {
    // what only the refused paths would read goes too: the transposed map (the
    // fallback kernel) and the positions of the merged values (values_changed)
    // -- 16 B per stored entry: symmetric storage then holds the merged copy,
    // the row pointer and the diagonal, 1.75 times its own CSR bytes
    SPMV_CHECK_HIP(hipSetDevice(plan->ctx->device));
    SPMV_CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(plan->t_pos);
    (void)hipFree(plan->t_row);
    plan->t_pos = plan->t_row = nullptr;
    (void)hipFree(plan->sjv_map);
    plan->sjv_map = nullptr;
  }
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_values_changed(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                     void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  MemClock mem_clock(plan);
  // (the arrays the copies would be refreshed from are gone)
  SPMV_REQUIRE(!plan->released);
  hipStream_t st = spmv_stream(ctx, stream);
  const auto t_begin = std::chrono::steady_clock::now();
  const int plan_us0 = plan->plan_us;
  int rc = SPMV_HIP_OK;
  // the sliced jagged copy: rewritten in place
  if (plan->sj_val && plan->sj_values0) {
    rc = plan->sj_elem == 8
             ? spmv_sjds_bake_f64(plan, static_cast<const double*>(plan->sj_values0),
                                  nullptr, st)
             : spmv_sjds_bake_f32(plan, static_cast<const float*>(plan->sj_values0),
                                  nullptr, st);
    // (the fp32 twin of the mixed SpMV)
    if (rc == SPMV_HIP_OK && plan->sj_val32 && plan->sj32_values0)
      rc = spmv_sjds_bake_f32f64(plan, static_cast<const float*>(plan->sj32_values0),
                                 st);
  }
  // (symmetric storage: the merged matrix's copy)
  if (rc == SPMV_HIP_OK && plan->sjt && plan->sjt->sj_val && plan->sjt->sj_values0) {
    const void* v0 = plan->sjt->sj_values0;
    rc = plan->sjt->sj_elem == 8
             ? spmv_sjds_bake_f64(plan->sjt, static_cast<const double*>(v0),
                                  plan->sjv_map, st)
             : spmv_sjds_bake_f32(plan->sjt, static_cast<const float*>(v0),
                                  plan->sjv_map, st);
  }
  // the diagonal forms: the device checks decide the form again (a matrix
  // they no longer hold: ENOTSUP = back to the CSR-order kernels, which is a
  // correct outcome of this call)
  if (rc == SPMV_HIP_OK && (plan->sdia_val || plan->wdia_val)) {
    struct NoNewForms { // (reset on every path out of this block)
      spmv_hip_csr_plan* p;
      ~NoNewForms() { p->no_new_forms = 0; }
    } guard{plan};
    plan->no_new_forms = 1;
    const void* v32 = plan->sdia32_values0 ? plan->sdia32_values0
                                           : plan->wdia32_values0;
    if (plan->sdia_val ? plan->sdia_elem == 8 : plan->wdia_elem == 8) {
      const double* v = static_cast<const double*>(
          plan->sdia_val ? plan->sdia_values0 : plan->wdia_values0);
      const double* d = static_cast<const double*>(plan->sdia_val ? plan->sdia_diag0
                                                                  : nullptr);
      rc = spmv_hip_csr_plan_bake_values_f64(ctx, plan, v, d, stream);
      if (rc == SPMV_HIP_OK && v32) {
        rc = spmv_hip_csr_plan_bake_values_f32f64(
            ctx, plan, static_cast<const float*>(v32), stream);
        if (rc == SPMV_HIP_ENOTSUP)
          rc = SPMV_HIP_OK;
      }
    } else {
      const float* v = static_cast<const float*>(
          plan->sdia_val ? plan->sdia_values0 : plan->wdia_values0);
      const float* d = static_cast<const float*>(plan->sdia_val ? plan->sdia_diag0
                                                                 : nullptr);
      rc = spmv_hip_csr_plan_bake_values_f32(ctx, plan, v, d, stream);
    }
    if (rc == SPMV_HIP_ENOTSUP)
      rc = SPMV_HIP_OK;
  }
  SPMV_CHECK_HIP(hipStreamSynchronize(st));
  plan->plan_us = plan_us0; // (the bakes added themselves: not plan creation)
  plan->values_changed_us
      = (int)std::chrono::duration_cast<std::chrono::microseconds>(
            std::chrono::steady_clock::now() - t_begin)
            .count();
  return rc;
}

int spmv_hip_csr_plan_algo(const spmv_hip_csr_plan* plan, int* algo)
{
  SPMV_REQUIRE(plan && algo);
  *algo = plan->algo;
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_set(spmv_hip_csr_plan* plan, const char* key, int value)
{
  SPMV_REQUIRE(plan && key);
  // arrays given up (plan_release_matrix): no key may select a kernel that
  // would read them
  if (plan->released)
    for (const char* k : {"algo", "sdia", "wdia", "sjds", "lat", "lx", "lxw", "xw",
                          "sym_det", "slat"})
      SPMV_REQUIRE(strcmp(key, k) != 0);
  if (!strcmp(key, "algo")) {
    // ROWLIST needs the list built at plan creation
    SPMV_REQUIRE(value >= SPMV_HIP_ALGO_ROWBLOCK
                 && (value <= SPMV_HIP_ALGO_SCALAR
                     || (value == SPMV_HIP_ALGO_ROWLIST && plan->row_list)));
    plan->algo = value;
  } else if (!strcmp(key, "lanes_per_row")) {
    SPMV_REQUIRE(value == 4 || value == 8 || value == 16 || value == 32
                 || value == 64);
    plan->lanes_per_row = value;
  } else if (!strcmp(key, "chunks")) {
    SPMV_REQUIRE(value == 1 || value == 2 || value == 4);
    plan->chunks = value;
  } else if (!strcmp(key, "nontemporal")) {
    plan->nontemporal = value != 0;
  } else if (!strcmp(key, "xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->xcd_group = value;
  } else if (!strcmp(key, "sym_window")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096 && value % 256 == 0);
    plan->sym_window = value;
  } else if (!strcmp(key, "sym_rows")) {
    SPMV_REQUIRE(value == 512 || value == 1024 || value == 2048);
    plan->sym_rows = value;
  } else if (!strcmp(key, "blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->blocks_per_cu = value;
    if (plan->zw_table && !plan->lat_tab && !plan->symmetric
        && !plan->sdia_val) // (LX or plain row blocks) tied to the grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "lx")) {
    // 1 needs the LX form built at plan creation (or by "lx_build")
    SPMV_REQUIRE(value == 0 || plan->lx_lidx);
    plan->lx = value != 0;
  } else if (!strcmp(key, "lxw")) {
    // the LDS-DMA kernel of the LX form (needs its records: ctx "lx_dma")
    SPMV_REQUIRE(value == 0 || plan->lxw_rec);
    plan->lxw = value != 0;
    if (plan->zw_table && plan->lx_lidx && !plan->lat_tab) // another grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "xw")) {
    // the LDS-DMA kernel on the caller's CSR arrays (needs its records)
    SPMV_REQUIRE(value == 0 || plan->xw_rec);
    plan->xw = value != 0;
    if (plan->xw_probe && value) { // asked for by name: no probe decides
      plan->xw_probe->decided = 1;
      plan->xw_probe->use_xw = 1;
    }
    if (plan->zw_table && plan->xw_rec && !plan->lat_tab && !plan->lx_lidx)
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "xw_probe")) {
    // 1: (re)start the choice between XW and the gather kernel by the next
    // four launches; 0: XW from here on
    SPMV_REQUIRE((value == 0 || value == 1) && plan->xw_rec);
    if (!plan->xw_probe)
      plan->xw_probe = new (std::nothrow) XwProbe;
    SPMV_REQUIRE(plan->xw_probe);
    xw_probe_drop_events(plan->xw_probe);
    *plan->xw_probe = XwProbe();
    plan->xw_probe->decided = value ? 0 : 1;
  } else if (!strcmp(key, "lxw_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 0 && value <= kBlocksPerCU);
    plan->lxw_blocks_per_cu = value;
    if (plan->zw_table && !plan->lat_tab
        && ((plan->lxw_rec && plan->lxw) || (plan->xw_rec && plan->xw)))
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "lx_chunks")) {
    SPMV_REQUIRE(value == 1 || value == 2);
    plan->lx_chunks = value;
  } else if (!strcmp(key, "nt_store")) {
    plan->nt_store = value != 0;
  } else if (!strcmp(key, "slat")) {
    // 1 needs the symmetric lattice form built at plan creation
    SPMV_REQUIRE(value == 0 || plan->slat_mask);
    plan->slat = value != 0;
  } else if (!strcmp(key, "sdia")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    SPMV_REQUIRE(value == 0 || plan->sdia_val);
    plan->sdia = value; // the CSR-order kernel has a grid of its own
    if (plan->zw_table && plan->sdia_val)
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "sjds")) {
    SPMV_REQUIRE(value == 0 || plan->sj_val || (plan->sjt && plan->sjt->sj_val));
    plan->sj = value != 0;
  } else if (!strcmp(key, "sj_phases")) { // ablation for measurements only
    SPMV_REQUIRE(value >= 1 && value <= 3);
    plan->sj_phases = value;
  } else if (!strcmp(key, "sj_long_panels")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->sj_long_panels = value;
  } else if (!strcmp(key, "sj_long_table")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->sj_long_table = value;
  } else if (!strcmp(key, "sj_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 0 && value <= kBlocksPerCU);
    plan->sj_blocks_per_cu = value;
    if (plan->sjt) // (symmetric storage: the merged matrix's plan is launched)
      plan->sjt->sj_blocks_per_cu = value;
  } else if (!strcmp(key, "sj_xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->sj_xcd_group = value;
  } else if (!strcmp(key, "wdia")) {
    SPMV_REQUIRE(value == 0 || plan->wdia_val);
    plan->wdia = value != 0;
  } else if (!strcmp(key, "wdia_box")) {
    // lines per lane of the constant 27-point box kernel (0 = general kernel)
    SPMV_REQUIRE((value == 0 || value == 2 || value == 4) && plan->wdia_val
                 && plan->wdia_const);
    return spmv_wdia_box_build(plan, value, 0, false);
  } else if (!strcmp(key, "wdia_hbox")) {
    // the marched kernel for the half form of a 27-point box (0 = the general
    // wide diagonal kernel)
    SPMV_REQUIRE((value == 0 || value == 1) && plan->wdia_val);
    const int rh = spmv_wdia_hbox_build(plan, value);
    if (rh != SPMV_HIP_OK)
      return rh;
    if (!plan->wdia_hbox && plan->wdia_d2 > 0 && !plan->wdia_zw_table)
      return spmv_wdia_walk_build(plan, 0, false); // the general kernel's order
    return SPMV_HIP_OK;
  } else if (!strcmp(key, "wdia_hbox_segs")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->wdia_hbox_segs = value;
  } else if (!strcmp(key, "wdia_box_segments")) {
    SPMV_REQUIRE(value >= 0 && plan->wdia_box > 1);
    return spmv_wdia_box_build(plan, plan->wdia_box, value, true);
  } else if (!strcmp(key, "wdia_box_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->wdia_box_blocks_per_cu = value;
    if (plan->wdia_box > 1)
      return spmv_wdia_box_build(plan, plan->wdia_box, 0,
                                 plan->wdia_box_table != nullptr);
  } else if (!strcmp(key, "wdia_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->wdia_blocks_per_cu = value;
    if (plan->wdia_val && plan->wdia_zw_table) // the table is tied to the grid
      return spmv_wdia_walk_build(plan, 0, true);
  } else if (!strcmp(key, "wdia_zwalk")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->wdia_zwalk = value;
  } else if (!strcmp(key, "wdia_zwalk_segments")) {
    // (re)build the wide diagonal form's plane-walk table, whatever the size
    SPMV_REQUIRE(value >= 0 && plan->wdia_val && plan->wdia_d2 > 0);
    return spmv_wdia_walk_build(plan, value, true);
  } else if (!strcmp(key, "wdia_xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->wdia_xcd_group = value;
  } else if (!strcmp(key, "slat_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->slat_blocks_per_cu = value;
    if (plan->zw_table && (plan->slat_mask || plan->sdia_val)) // tied to the grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "sdia_tile")) {
    // lines per lane of the constant-diagonal kernel (1, 2, 4)
    SPMV_REQUIRE((value == 1 || value == 2 || value == 4) && plan->sdia_val
                 && plan->sdia_const);
    return spmv_sdia_tile_build(plan, value, 0, false);
  } else if (!strcmp(key, "sdia_tile_segments")) {
    SPMV_REQUIRE(value >= 0 && plan->sdia_tile > 1);
    return spmv_sdia_tile_build(plan, plan->sdia_tile, value, true);
  } else if (!strcmp(key, "sdia_tile_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->sdia_tile_blocks_per_cu = value;
    if (plan->sdia_tile > 1)
      return spmv_sdia_tile_build(plan, plan->sdia_tile, 0,
                                  plan->sdia_tile_table != nullptr);
  } else if (!strcmp(key, "sdia_nt")) {
    SPMV_REQUIRE(value >= 0 && value < 32);
    plan->sdia_nt = value;
  } else if (!strcmp(key, "sdia_chain")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->sdia_chain = value; // LDS footprint, hence the grid, may change
    if (plan->zw_table && plan->sdia_val)
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "zwalk")) {
    SPMV_REQUIRE(value == 0 || plan->zw_table);
    plan->zwalk = value != 0;
  } else if (!strcmp(key, "zwalk_segments")) {
    // (re)build the plane-walk table of the plan's lattice kernel with `value`
    // runs along the plane axis (0 = choose), whatever the size
    SPMV_REQUIRE(value >= 0 && plan->zw_d2 > 0);
    return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), value,
                                  true);
  } else if (!strcmp(key, "sym_det")) {
    // 1 needs the transposed map built at plan creation
    SPMV_REQUIRE(value == 0 || plan->t_ptr);
    plan->sym_det = value != 0;
  } else if (!strcmp(key, "lat")) {
    // 1 needs the lattice form built at plan creation
    SPMV_REQUIRE(value == 0 || plan->lat_tab);
    plan->lat = value != 0;
  } else if (!strcmp(key, "lat_chain")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->lat_chain = value;
  } else if (!strcmp(key, "lat_xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->lat_xcd_group = value;
  } else if (!strcmp(key, "lat_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->lat_blocks_per_cu = value;
    if (plan->zw_table && plan->lat_tab) // the table is tied to the grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else {
    return SPMV_HIP_EINVAL;
  }
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_get(const spmv_hip_csr_plan* plan, const char* key,
                          int* value)
{
  SPMV_REQUIRE(plan && key && value);
  if (!strcmp(key, "algo"))
    *value = plan->algo;
  else if (!strcmp(key, "sym_det"))
    *value = plan->sym_det;
  else if (!strcmp(key, "slat"))
    *value = plan->slat;
  else if (!strcmp(key, "sdia"))
    *value = plan->sdia && plan->sdia_val ? 1 : 0;
  else if (!strcmp(key, "sjds"))
    *value = plan->sj && (plan->sj_val || (plan->sjt && plan->sjt->sj_val)) ? 1 : 0;
  else if (!strcmp(key, "sym_sj")) // symmetric storage, both blocks sliced jagged
    *value = plan->symmetric && plan->sym_sj && plan->sj && plan->sjt
                     && plan->sjt->sj_val
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_mixed")) // the fp32 twin of the jagged copy is baked
    *value = plan->sj && plan->sj_val32 ? 1 : 0;
  else if (!strcmp(key, "sj_built"))
    *value = plan->sj_lenperm || (plan->sjt && plan->sjt->sj_lenperm) ? 1 : 0;
  else if (!strcmp(key, "sj_wpb")) // (symmetric storage: the merged matrix's)
    *value = plan->sj_lenperm ? plan->sj_wpb
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_wpb : 0;
  else if (!strcmp(key, "sj_sigma"))
    *value = plan->sj_lenperm ? plan->sj_sigma
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_sigma : 0;
  else if (!strcmp(key, "sj_unit"))
    *value = plan->sj_lenperm ? plan->sj_unit
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_unit : 0;
  else if (!strcmp(key, "sj_max_chunks"))
    *value = plan->sj_lenperm ? plan->sj_maxk
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_maxk : 0;
  else if (!strcmp(key, "sj_far_permille")) {
    // (symmetric storage: of the merged matrix)
    const spmv_hip_csr_plan* c
        = plan->sj_lenperm ? plan : (plan->sjt && plan->sjt->sj_lenperm ? plan->sjt : nullptr);
    *value = c && c->nnz > 0 ? (int)((c->sj_far * 1000 + c->nnz - 1) / c->nnz) : 0;
  }
  else if (!strcmp(key, "sj_staged_bytes_per_entry_x100")) {
    const spmv_hip_csr_plan* c
        = plan->sj_lenperm ? plan : (plan->sjt && plan->sjt->sj_lenperm ? plan->sjt : nullptr);
    *value = c && c->nnz > 0 ? (int)(c->sj_sumk * 12800 / c->nnz) : 0;
  }
  else if (!strcmp(key, "sj_pad_permille")) // entries of padding per 1000 stored
    *value = plan->sj_lenperm && plan->nnz > 0
                 ? (int)((plan->sj_units * plan->sj_unit * 1000) / plan->nnz)
                 : 0;
  // (symmetric storage: the long rows of the stored block are the PARENT's)
  else if (!strcmp(key, "sj_long_panels"))
    *value = (plan->sj_lenperm || plan->sjt) && plan->sj_long_sorted
                     && plan->sj_long_panels
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_long_sorted"))
    *value = (plan->sj_lenperm || plan->sjt) && plan->sj_nlong > 0
                     && plan->sj_long_sorted
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_long_table"))
    *value = (plan->sj_lenperm || plan->sjt) && plan->sj_long_sorted
                     && plan->sj_long_panels && plan->sj_lt_tab && plan->sj_long_table
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_long_table_kib"))
    *value = (int)((plan->sj_lt_entries * 4 + (int64_t)plan->sj_lt_nsg * 16) / 1024);
  else if (!strcmp(key, "sj_long_rows"))
    *value = (plan->sj_lenperm || plan->sjt) ? plan->sj_nlong : 0;
  else if (!strcmp(key, "sj_wide"))
    *value = plan->sj_lenperm ? plan->sj_wide_alloc
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_wide_alloc : 0;
  else if (!strcmp(key, "sj_blocks_per_cu"))
    *value = plan->sj_blocks_per_cu;
  else if (!strcmp(key, "wdia"))
    *value = plan->wdia && plan->wdia_val ? 1 : 0;
  else if (!strcmp(key, "wdia_hbox"))
    *value = plan->wdia && plan->wdia_val ? plan->wdia_hbox : 0;
  else if (!strcmp(key, "wdia_offsets"))
    *value = plan->wdia_val ? plan->wdia_K : 0;
  else if (!strcmp(key, "xw"))
    *value = plan->xw && plan->xw_rec ? 1 : 0;
  else if (!strcmp(key, "xw_staged"))
    *value = plan->xw_staged;
  else if (!strcmp(key, "xw_pick")) // -1: the probe is still running
    *value = !(plan->xw && plan->xw_rec) ? 0
             : !plan->xw_probe           ? 1
             : plan->xw_probe->decided   ? plan->xw_probe->use_xw
                                         : -1;
  else if (!strcmp(key, "xw_probe_xw_us"))
    *value = plan->xw_probe ? (int)plan->xw_probe->us_xw : 0;
  else if (!strcmp(key, "xw_probe_gather_us"))
    *value = plan->xw_probe ? (int)plan->xw_probe->us_gather : 0;
  else if (!strcmp(key, "plan_us"))
    *value = plan->plan_us;
  else if (!strcmp(key, "plan_mem_us")) // of plan_us: inside hipMalloc / hipFree
    *value = plan->plan_mem_us;
  else if (!strcmp(key, "values_changed_us"))
    *value = plan->values_changed_us;
  else if (!strcmp(key, "plan_kib")) {
    // device memory the plan owns beyond the caller's CSR arrays
    const int64_t n = plan->num_rows, nnz = plan->nnz;
    const int64_t nrb = (n + kRows - 1) / kRows;
    int64_t b = 0;
    if (plan->row_list)
      b += 4 * (int64_t)plan->num_listed;
    if (plan->lx_lidx)
      b += 2 * (nnz + 8) + 4 * nrb * kLxRec;
    if (plan->lxw_rec)
      b += 4 * nrb * kLxwRec;
    if (plan->xw_rec)
      b += 4 * nrb * kXwRec;
    if (plan->lat_tab)
      b += 48 * nrb + n;
    if (plan->slat_mask)
      b += n;
    const int64_t narr
        = plan->sdia_general == 2 ? 2 * plan->sdia_nd + 1 : plan->sdia_nd + 1;
    if (plan->sdia_val)
      b += narr * plan->sdia_len * plan->sdia_elem + n;
    if (plan->sdia32_val && !plan->sdia_const)
      b += narr * plan->sdia_len * 4 + n;
    if (plan->wdia_val)
      b += (int64_t)plan->wdia_narr * plan->wdia_len * plan->wdia_elem + 4 * n;
    if (plan->wdia32_val && !plan->wdia_const)
      b += (int64_t)plan->wdia_narr * plan->wdia_len * 4;
    if (plan->sj_lenperm)
      b += 4 * ((n + 63) / 64 * 64) + 8 * (int64_t)plan->sj_nblk
           + 4 * (int64_t)plan->sj_nblk * plan->sj_stride
           + (plan->sj_wide_alloc ? 4 : 2) * plan->sj_units * plan->sj_unit
           + 4 * (int64_t)plan->sj_nlong + 4 * ((n + 63) / 64 + 1)
           + 4 * plan->sj_lt_entries + 16 * (int64_t)plan->sj_lt_nsg
           + 2 * plan->sj_lt_codes_n + 8 * (int64_t)plan->sj_nlong;
    if (plan->sj_val)
      b += (int64_t)plan->sj_elem * plan->sj_units * plan->sj_unit;
    if (plan->sj_val32)
      b += 4 * plan->sj_units * plan->sj_unit;
    if (plan->sjt && plan->sjt->sj_lenperm) { // symmetric storage: the merged matrix
      const spmv_hip_csr_plan* c = plan->sjt;
      // its row pointer, the positions of its values (until released)
      b += 4 * (n + 1) + (plan->sjv_map ? 4 * c->nnz : 0);
      if (plan->sj_long_rows) // the stored block's long rows: list, table, codes
        b += 4 * (int64_t)plan->sj_nlong + 4 * plan->sj_lt_entries
             + 16 * (int64_t)plan->sj_lt_nsg + 2 * plan->sj_lt_codes_n
             + 8 * (int64_t)plan->sj_nlong;
      b += 4 * ((n + 63) / 64 * 64) + 8 * (int64_t)c->sj_nblk
           + 4 * (int64_t)c->sj_nblk * c->sj_stride
           + (c->sj_wide_alloc ? 4 : 2) * c->sj_units * c->sj_unit
           + 4 * ((n + 63) / 64 + 1);
      if (c->sj_val)
        b += (int64_t)c->sj_elem * c->sj_units * c->sj_unit;
    }
    if (plan->t_ptr)
      b += 4 * (n + 1) + (plan->t_row ? 8 * nnz : 0);
    if (plan->zw_table)
      b += 4 * (int64_t)plan->zw_slots;
    *value = (int)((b + 1023) / 1024);
  }
  else if (!strcmp(key, "sdia_offsets"))
    *value = plan->sdia_val ? plan->sdia_nd : 0;
  else if (!strcmp(key, "sdia_general"))
    *value = plan->sdia_val ? plan->sdia_general : 0;
  else if (!strcmp(key, "sdia_mixed"))
    *value = plan->sdia32_val ? 1 : 0;
  else if (!strcmp(key, "sdia_tile"))
    *value = plan->sdia_val ? plan->sdia_tile : 0;
  else if (!strcmp(key, "sdia_tile_walk"))
    *value = plan->sdia_val && plan->sdia_tile_table ? plan->sdia_tile_segments : 0;
  else if (!strcmp(key, "sdia_const"))
    *value = plan->sdia_val ? plan->sdia_const : 0;
  else if (!strcmp(key, "wdia_zwalk"))
    *value = plan->wdia_val && plan->wdia_zwalk && plan->wdia_zw_table ? 1 : 0;
  else if (!strcmp(key, "wdia_zwalk_segments"))
    *value = plan->wdia_zw_table ? plan->wdia_zw_segments : 0;
  else if (!strcmp(key, "wdia_d2"))
    *value = plan->wdia_val ? plan->wdia_d2 : 0;
  else if (!strcmp(key, "wdia_box"))
    *value = plan->wdia_val ? plan->wdia_box : 0;
  else if (!strcmp(key, "wdia_const"))
    *value = plan->wdia_val ? plan->wdia_const : 0;
  else if (!strcmp(key, "wdia_half"))
    *value = plan->wdia_val && !plan->wdia_const
                     && plan->wdia_narr < plan->wdia_K
                 ? 1
                 : 0;
  else if (!strcmp(key, "wdia_mixed"))
    *value = plan->wdia32_val ? 1 : 0;
  else if (!strcmp(key, "sdia_chain"))
    *value = plan->sdia_chain;
  else if (!strcmp(key, "sdia_nt"))
    *value = plan->sdia_nt;
  else if (!strcmp(key, "zwalk"))
    *value = plan->zwalk && plan->zw_table ? 1 : 0;
  else if (!strcmp(key, "zwalk_segments"))
    *value = plan->zw_table ? plan->zw_segments : 0;
  else if (!strcmp(key, "zwalk_grid"))
    *value = plan->zw_table ? plan->zw_grid : 0;
  else if (!strcmp(key, "lattice_d1"))
    *value = plan->lattice_d1;
  else if (!strcmp(key, "lattice_d2"))
    *value = plan->lattice_d2;
  else if (!strcmp(key, "lat"))
    *value = plan->lat;
  else if (!strcmp(key, "lat_chain"))
    *value = plan->lat_chain;
  else if (!strcmp(key, "lat_blocks"))
    *value = plan->lat_blocks;
  else if (!strcmp(key, "lx"))
    *value = plan->lx;
  else if (!strcmp(key, "lxw"))
    *value = plan->lxw && plan->lxw_rec ? 1 : 0;
  else if (!strcmp(key, "lx_staged"))
    *value = plan->lx_staged;
  else if (!strcmp(key, "lx_blocks"))
    *value = plan->lx_blocks;
  else if (!strcmp(key, "blocks_per_cu"))
    *value = plan->blocks_per_cu;
  else if (!strcmp(key, "nontemporal"))
    *value = plan->nontemporal;
  else
    return SPMV_HIP_EINVAL;
  return SPMV_HIP_OK;
}

} // extern "C"
