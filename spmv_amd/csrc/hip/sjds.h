// Sliced jagged form: what its three translation units share --
//   spmv_sjds_plan.hip  plan time: chunk selection, sigma sort, fill, bake, the
//                       long rows' list and table, the merged matrix of
//                       symmetric storage
//   spmv_sjds_long.hip  the long rows' kernels (eight lanes per row; the
//                       table-driven one) and their launches
//   spmv_sjds.hip       the slices' kernel (general and symmetric storage), the
//                       launches and the run entry points
// (the form itself is described at the top of spmv_sjds.hip)
#pragma once

#include "csr_plan.h"

#include <type_traits>

constexpr int kSjChunk = 16;       // columns per staged chunk
constexpr int kSjSpanWords = 2048; // bitmap words of the plan analysis: 65,536
                                   // chunks = 2^20 columns around the block
constexpr int kSjGroup = 8;        // entries per lane and load group (two groups
                                   // in flight): 8 / E steps
constexpr int kSjTailMin = 48;     // entries past the second-longest row from
                                   // which the wave takes lane 0's row over
constexpr int kSjLongMin = 96;     // LONG rows: more than 4 x the average and
                                   // more than this many entries
constexpr uint32_t kSjLongFlag = 0x80000000u; // ... marked in their lenperm word
constexpr int kSjSlack = 128;      // units of slack behind the jagged arrays: a
                                   // step past a slice's end reads, never uses

// LONG rows (the long-row kernel takes them, the slices leave them out): more
// than `thr` entries -- and not within kSjLongPad entries of the arrays' end,
// so that the kernel's loads may run past a row's end without a clamp
constexpr int kSjLongPad = 160;
__host__ __device__ __forceinline__ bool sj_is_long(int32_t a, int32_t b, int thr,
                                                    int64_t nnz)
{
  return b - a > thr && (int64_t)b + kSjLongPad <= nnz;
}

// E consecutive entries of a row: one aligned load
template <typename X, int E>
struct __attribute__((aligned(sizeof(X) * E))) SjUnit {
  X e[E];
};

// The long rows keep their row order (neighbours in x share the staged panels)
// and are sorted by length inside runs of 2^6 = one workgroup's 64 rows.
// Measured on the 1 % tail (same box): runs of 16 / 32 / 64 / 128 / 256 / 1024
// rows 0.45 / 0.44 / 0.44 / 0.48 / 0.58 / 1.24 ms -- what the longer runs gain in
// waves that end together they lose several times over in panels (the rows of
// a workgroup are no longer neighbours); 4-wave workgroups 0.50-0.65.
// The table-driven kernel (csr_sjds_longt_kernel) takes supergroups of 64 RS
// rows, an 8-lane group RS of them (RS = 1: see the measurements there): the
// runs are its supergroups.
#ifndef SJ_LT_RS
#define SJ_LT_RS 1
#endif
#ifndef SJ_LT_G
#define SJ_LT_G 8
#endif
#define SJ_LT_G_ SJ_LT_G
constexpr int kSjLtRS = SJ_LT_RS;        // rows per group and supergroup
constexpr int kSjLtRun = 512 / SJ_LT_G_ * kSjLtRS; // rows per supergroup
static_assert(kSjLtRun == 64 || kSjLtRun == 128 || kSjLtRun == 256 || kSjLtRun == 512,
              "runs of 64 ... 512 rows");
#ifndef SJ_LONG_RUN_SHIFT
#define SJ_LONG_RUN_SHIFT (kSjLtRun == 64 ? 6 : kSjLtRun == 128 ? 7 : kSjLtRun == 256 ? 8 : 9)
#endif
constexpr int kSjLongRunShift = SJ_LONG_RUN_SHIFT;

// ---------------------------------------------------------------------------
// the SpMV kernel
// ---------------------------------------------------------------------------
// TV: the type of the stored values (fp32 under fp64 vectors and arithmetic:
// the mixed-precision SpMV)
template <typename T, typename TV = T>
struct SjArgs {
  int32_t num_rows, num_cols;
  int32_t nblk;
  int32_t maxk;      // staged chunks the LDS buffer holds
  int32_t stride;    // chunk-list entries per block
  int32_t wide_alloc;
  const int32_t* rowptr;  // the caller's (long rows)
  const uint32_t* ubase;  // first unit of every slice
  const int32_t* lenperm;
  const int32_t* blk;     // per block: chunks, wide
  const int32_t* chunks;
  const unsigned char* codes;
  const TV* val;          // jagged order
  // long rows (phase 0): straight from the caller's CSR arrays
  int32_t phases; // measurement only (plan_set "sj_phases"): 1 = long rows, 2 = slices
  int32_t nlong;
  int32_t long_sorted; // every long row's columns ascend: x by panels
  int32_t long_panel;  // ... of this many columns
  const int32_t* long_rows;
  const int32_t* colind;
  const TV* values;
  // the table-driven long-row kernel (csr_sjds_longt_kernel)
  const int32_t* lt_cmin;
  const int32_t* lt_np;
  const int64_t* lt_off;
  const int32_t* lt_tab;
  const uint16_t* lt_codes; // per entry of the listed rows: column - its panel's first
  const int64_t* lt_coff;   // per listed row: its first code
  // symmetric storage: the rows of the stored lower block with more than
  // sym_long_thr entries (sj_is_long, against the block's sym_nnz entries) give
  // their LOWER part to the long-row kernels, which run FIRST on the caller's
  // arrays, start the row's sum at d_i x_i (sym_diag; null: at 0) and leave
  // y_i = fl(alpha sum) + fl(beta y0_i) -- the row's value where its column's
  // entries begin; the slices' kernel starts such a row from that y_i
  const T* sym_diag = nullptr;
  int32_t sym_long_thr = INT32_MAX;
  int64_t sym_nnz = 0;
};

template <typename T>
__device__ __forceinline__ T sj_readlane(T v, int j);
template <>
__device__ __forceinline__ double sj_readlane<double>(double v, int j)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float sj_readlane<float>(float v, int j)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

constexpr int kSjLpr = 8;
#ifndef SJ_PANEL_U
#define SJ_PANEL_U 4
#endif
constexpr int kSjPanelU = SJ_PANEL_U; // steps per trip of the panel walk
#ifndef SJ_LONG_SETS
#define SJ_LONG_SETS 1
#endif
// sets of eight rows per wave that share the staged panels.  Measured on the
// 1 % tail (same box, alternating builds): 1 / 2 / 4 sets 0.44 / 0.48 / 0.57 ms --
// half the x staged per entry does not pay for the longer walk per panel (121
// / 146 registers): the kernel is bound by the latency of its trips, not by
// the panels' bytes
constexpr int kSjLongSets = SJ_LONG_SETS;
constexpr int kSjLU = 4; // steps per load group of the long-row phase

constexpr int kSjLtMaxPanels = 64;       // wider supergroups: rows one by one
#ifndef SJ_LT_PANEL_COLS
#define SJ_LT_PANEL_COLS 8192
#endif
// columns of x per panel: 64 KiB of fp64, two workgroups per CU; a multiple of
// the 1024 columns one round of the workgroup's 16-byte loads stages
constexpr int kSjLtPanel = SJ_LT_PANEL_COLS;
#ifndef SJ_LT_G
#define SJ_LT_G 8
#endif
#ifndef SJ_LT_EPL
#define SJ_LT_EPL 4
#endif
constexpr int kSjLtG = SJ_LT_G;           // lanes per row (a power of two <= 16)
constexpr int kSjLtEpl = SJ_LT_EPL;       // entries per lane and trip (4 or 8)
#ifndef SJ_LT_DEPTH
#define SJ_LT_DEPTH 1
#endif
constexpr int kSjLtDepth = SJ_LT_DEPTH;   // trips of loads in flight ahead
constexpr int kSjLtTrip = kSjLtG * kSjLtEpl; // ... per group and trip
#ifndef SJ_LT_CODES
#define SJ_LT_CODES 1
#endif
// the long rows' columns as 16-bit positions inside their panel (the plan's own
// array, 2 B per entry) instead of the caller's 4-byte colind: 10 instead of 12
// bytes per entry streamed
constexpr bool kSjLtCodes = SJ_LT_CODES != 0;
static_assert(kSjLtPanel <= 65536, "16-bit panel positions");
static_assert(kSjLtTrip * (kSjLtDepth + 1) + 8 <= kSjLongPad, "loads past a row's end");
static_assert(kSjLtPanel % 1024 == 0, "whole staging rounds");

template <typename X, int N>
struct __attribute__((packed, aligned(sizeof(X)))) SjPack {
  X e[N];
};

// DPP move within rows of 16 lanes: lanes without a source keep `old`
template <int CTRL>
__device__ __forceinline__ double sj_dpp(double old, double src)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src),
                                             CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src),
                                             CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float sj_dpp(float old, float src)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(old), __float_as_int(src), CTRL, 0xF, 0xF, false));
}
constexpr int kDppRowShr1 = 0x111; // lane l <- lane l - 1
constexpr int kDppRowShlBack = 0x100 + kSjLtG - 1; // lane l <- lane l + G - 1
static_assert(kSjLtG == 2 || kSjLtG == 4 || kSjLtG == 8 || kSjLtG == 16, "DPP rows");

// SIGMA layout: blocks of this many rows sorted by length across the block; the
// row inside its block takes this many bits of its (length, row) word
constexpr int kSjSigRows = 1024;
constexpr int kSjSigBits = 10;

// workgroups of a CU that fit `lds` bytes each, and the waves they bring (the
// kernel's registers allow 16 waves per CU)
inline int sj_wgs_per_cu(int wpb, int64_t lds)
{
  int wgs = (int)((160 * 1024 - 2048) / (lds > 1 ? lds : 1));
  const int by_waves = 16 / wpb;
  wgs = wgs < by_waves ? wgs : by_waves;
  return wgs < 1 ? 1 : wgs;
}

// ---- cross-file entry points ------------------------------------------------
// spmv_sjds_long.hip: the long rows of A.long_rows (listed by the plan `pl`)
// after -- general storage -- or before -- symmetric storage -- the slices'
// kernel; dot partials (dot.partials != nullptr) go behind the slices', from
// slot `dot_slot0` on, of which `dot_room` are free
int spmv_sj_long_launch_f64(const spmv_hip_csr_plan* pl, SjArgs<double, double> A,
                            hipStream_t st, double alpha, const double* in, double beta,
                            double* out, DotOut dot, int dot_slot0, int dot_room);
int spmv_sj_long_launch_f32(const spmv_hip_csr_plan* pl, SjArgs<float, float> A,
                            hipStream_t st, float alpha, const float* in, float beta,
                            float* out, DotOut dot, int dot_slot0, int dot_room);
int spmv_sj_long_launch_f32f64(const spmv_hip_csr_plan* pl, SjArgs<double, float> A,
                               hipStream_t st, double alpha, const double* in,
                               double beta, double* out, DotOut dot, int dot_slot0,
                               int dot_room);
// raise the dynamic-LDS limit of the table-driven kernel on the current device
int spmv_sj_lt_raise_lds();
// spmv_sjds_plan.hip
void spmv_sj_lt_free(spmv_hip_csr_plan* pl);
int spmv_sj_build_long_list(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                            const int32_t* colind, int thr, hipStream_t st);
int spmv_sj_build_long_table(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                             const int32_t* colind, hipStream_t st);
