// Shared internals of the CSR SpMV translation units (gfx950 only):
//   spmv_csr.hip        general kernels (ROWBLOCK, LX form, VECTOR, SCALAR, ROWLIST),
//                       their launches, the C entry points of the product
//   spmv_csr_forms.hip  plan-time builders (LX / XW windows, row list, plane walk)
//   spmv_csr_plan.hip   the plan API (create / bake / set / get)
//   spmv_sym.hip   symmetric-storage kernels (csr_kernels.cpp:26-40)
//   spmv_lat.hip   lattice form: constant column offsets per row block
//   spmv_symlat.hip  the same idea for the symmetric storage
//   spmv_symdia.hip  ... with the values re-laid out by offset (baked copy)
//   spmv_sjds.hip    sliced jagged form: ragged / long rows, x staged in LDS
#pragma once

#include "common.h"

#include <chrono>
#include <cstdio>

constexpr int kRows = kBlock; // rows per workgroup in ROWBLOCK kernels

// clang ext-vector types: 16-byte loads/stores, accepted by the
// non-temporal builtins (HIP's double2/int4 wrapper structs are not).
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct VecOf;
template <>
struct VecOf<double> {
  static constexpr int V = 2; // 16 B of values per lane per load
  using val_t = f64x2;
  using col_t = i32x2;
};
template <>
struct VecOf<float> {
  static constexpr int V = 4;
  using val_t = f32x4;
  using col_t = i32x4;
};

template <bool NT, typename P>
__device__ __forceinline__ P stream_load(const P* p)
{
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}

// ---------------------------------------------------------------------------
// Row-block traversal order.  A persistent grid walks "slots" it = blockIdx.x,
// blockIdx.x + gridDim.x, ...; this maps a slot to the row block it computes.
// Placement only changes speed, never the result.  Slots it with equal it % 8
// run on the same XCD, because workgroups are dealt round-robin over the XCDs.
//   table      plan-time order table (plane walk, spmv_zwalk_order_build);
//              -1 marks an empty slot
//   xcd_group  no table: inside each run of 8G row blocks XCD k owns G
//              consecutive ones:  (it / 8G) * 8G + (it % 8) * G + (it / 8) % G
//   otherwise  identity
// ---------------------------------------------------------------------------
struct RowBlockOrder {
  const int32_t* table;
  int num_slots; // table length
  int xcd_group;
  int num_row_blocks;
  int nt_store; // write y non-temporally (plain row-block kernel)
};

__device__ __forceinline__ int order_slots(const RowBlockOrder& o)
{
  if (o.table)
    return o.num_slots;
  if (o.xcd_group > 0) {
    const int super = 8 * o.xcd_group;
    return ((o.num_row_blocks + super - 1) / super) * super;
  }
  return o.num_row_blocks;
}

// row block of slot `it`, or -1 for an empty slot (uniform per workgroup)
__device__ __forceinline__ int order_row_block(const RowBlockOrder& o, int it)
{
  int rb = it;
  if (o.table) {
    rb = o.table[it];
  } else if (o.xcd_group > 0) {
    const int super = 8 * o.xcd_group;
    const int q = it % super;
    rb = (it - q) + (q & 7) * o.xcd_group + (q >> 3);
  }
  return rb < o.num_row_blocks ? rb : -1;
}

#ifdef __HIPCC__
// Slot -> row block for the kernels that walk EVERY slot (an empty slot, -1, is
// a step without work) instead of searching for the next non-empty one.  With
// an order table the entry is a global load: order_slot_raw issues it ahead
// into a (uniform) vector register and order_slot_decode reads it after the
// step's own vmcnt(0), so that it is never waited for by itself -- a wait
// there would drain the LDS-DMA prefetch with it.
__device__ __forceinline__ int order_slot_raw(const RowBlockOrder& ord, int it,
                                              int num_slots)
{
  if (it >= num_slots)
    return -1;
  if (ord.table)
    return ord.table[it];
  RowBlockOrder o = ord;
  o.num_row_blocks = INT32_MAX; // bounds are checked by decode
  return order_row_block(o, it);
}
// The same with the SOURCE of the entry fixed at compile time (TAB: the order
// table).  The two sources must not share a register in one kernel: a table
// entry is a loaded value, and before the computed alternative may overwrite
// that register the compiler waits for every load in flight -- including the
// LDS-DMA pieces a step has just issued, which serialises the step (found in
// round 3: it cost the lattice kernels their prefetch whenever no table was
// in use).
template <bool TAB>
__device__ __forceinline__ int order_slot_raw_t(const RowBlockOrder& ord, int it,
                                                int num_slots)
{
  if (it >= num_slots)
    return -1;
  if constexpr (TAB) {
    return ord.table[it];
  } else {
    RowBlockOrder o = ord;
    o.table = nullptr;
    o.num_row_blocks = INT32_MAX; // bounds are checked by decode
    return order_row_block(o, it);
  }
}
__device__ __forceinline__ int order_slot_decode(const RowBlockOrder& ord,
                                                 int raw)
{
  const int rb = __builtin_amdgcn_readfirstlane(raw);
  return rb < ord.num_row_blocks ? rb : -1;
}
#endif

// LX form (spmv_csr.hip: register-staged kernel; spmv_lxw.hip: LDS-DMA kernel)
constexpr int kLxMaxWin = 16;
constexpr int kLxRec = 36;   // ints per row-block record (1 + 16 + 17, padded)
constexpr int kLxCap = 1344; // staged x elements per row block (10.5 KiB fp64)
constexpr int kLxGap = 16;   // columns closer than this share a window
// ... the DMA kernel's own record (one per row block):
//   [0] number of windows, or -1 = direct (global gather)
//   [1] first entry of the block's span in `values`   [2] its entry count
//   [3] number of staged pieces (low byte) | (staged position of column
//       rb * kRows + 1) << 8 when the block's own columns [rb * kRows, + rows)
//       lie inside one staged window, 0 there otherwise: the fused dot's x_i
//   [kLxwPieces0 + p]  first column of piece p (kLxwPiece elements each)
constexpr int kLxwPiece = 128;     // staged elements per piece
constexpr int kLxwMaxPieces = 16;  // 2048 staged elements (16 KiB fp64)
constexpr int kLxwPieces0 = 4;
constexpr int kLxwNpMask = 0xff;   // word [3]: pieces
constexpr int kLxwOwnShift = 8;    // ... own position + 1
constexpr int kLxwRec = 20;        // ints per record (4 + 16)
constexpr int kLxwAlign = 4;       // window starts: multiples of 4 columns
// XW: the same kernel on the CALLER's column indices (no 16-bit copy): the
// record carries, after the LXW words, the windows themselves -- an entry's
// staged position is its column + the delta of the last window that starts at
// or below it.
//   [kXwFirst0 + k]  first column of window k + 1 (k = 0..6; INT32_MAX: none)
//   [kXwDelta0 + k]  staged offset - first column of window k (k = 0..7)
// (the 7-point matrix of a 512^3 grid has five windows per row block: the two
// planes, the two lines -- 256 columns away from the block's own -- and its own)
constexpr int kXwMaxWin = 8;
constexpr int kXwFirst0 = kLxwPieces0 + kLxwMaxPieces; // 20
constexpr int kXwDelta0 = kXwFirst0 + kXwMaxWin;       // 28
constexpr int kXwRec = 36;

// Lattice form (spmv_lat.hip)
constexpr int kLatMaxOff = 8; // offsets per row block = mask bits
constexpr int kLatRec = 12;   // ints per row-block record: count, 3 pad, offsets
                              // (the offsets 16-byte aligned: one scalar load)
// Wide diagonal form (spmv_wdia.hip): up to this many distinct col - row
constexpr int kWdiaMaxOff = 32;

// SPMV_PLAN_TRACE=1: where a plan's set-up time goes (stderr, milliseconds since
// the tracer was made; every mark synchronises the device first so that the
// kernels of a phase are counted in it).  Measurement only.
struct PlanTrace {
  bool on;
  std::chrono::steady_clock::time_point t0;
  const char* who;
  explicit PlanTrace(const char* w)
      : on(getenv("SPMV_PLAN_TRACE") != nullptr), t0(std::chrono::steady_clock::now()),
        who(w)
  {
  }
  void mark(const char* what) const
  {
    if (!on)
      return;
    (void)hipDeviceSynchronize();
    fprintf(stderr, "%s %-14s %9.3f ms\n", who, what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now()
                                                      - t0)
                .count());
  }
};

template <typename T>
static inline bool aligned16(const T* p)
{
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// out *= beta (zero-fill for beta == 0 without reading out: SURVEY F7b)
template <typename T>
__global__ __launch_bounds__(kBlock) void scale_kernel(int64_t n, T beta,
                                                       T* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (beta == T(0)) ? T(0) : beta * out[i];
}

// ---------------------------------------------------------------------------
// Plan = CSRSpMV::_aux_data
// ---------------------------------------------------------------------------
struct spmv_hip_csr_plan {
  spmv_hip_ctx* ctx = nullptr;
  int32_t num_rows = 0, num_cols = 0;
  int64_t nnz = 0;
  bool symmetric = false;
  int algo = SPMV_HIP_ALGO_ROWBLOCK;
  int lanes_per_row = 8;  // VECTOR
  int chunks = 1;         // ROWBLOCK: 16-B loads per lane per tile (1, 2, 4)
  int nontemporal = 0;    // ROWBLOCK: nt loads on the matrix stream
  int xcd_group = 16;     // ROWBLOCK: consecutive row blocks per XCD (0 = off)
  // All 8 workgroups per CU (= 32 waves, the occupancy limit).  Leaving one
  // slot per CU free for the RCCL halo kernel cost the LX kernel 6 % and, at
  // 19.6 KB of LDS per workgroup, would not leave a communication kernel the
  // LDS it needs anyway; the halo is enqueued first, on a high-priority
  // stream, and takes its slots before the persistent grid fills the chip.
  int blocks_per_cu = kBlocksPerCU;
  int sym_window = 256;   // symmetric: LDS window below the block (0 = plain
                          // per-entry global atomics)
  int sym_rows = 1024;    // symmetric: rows per workgroup (512, 1024, 2048)
  // symmetric, deterministic: the transposed map (spmv_sym.hip); sym_det = 0
  // runs the atomic kernels (plan_set "sym_det")
  int32_t* t_ptr = nullptr; // rows + 1
  int32_t* t_pos = nullptr; // per stored entry: position in `values` ...
  int32_t* t_row = nullptr; // ... and its row, sorted by column
  int sym_det = 0;
  // symmetric lattice form (spmv_symlat.hip): <= 3 constant lower offsets
  uint8_t* slat_mask = nullptr; // per row: bit k = lower offset k present
  int slat = 0;                 // use it (plan_set "slat")
  int slat_nd = 0, slat_ext = 0, slat_nw = 0, slat_sub_bytes = 0;
  int slat_D[3] = {0, 0, 0}, slat_win_of_k[3] = {0, 0, 0};
  int slat_u_of_w[4] = {0, 0, 0, 0};
  int slat_blocks_per_cu = kBlocksPerCU;
  // symmetric diagonal form (spmv_symdia.hip): the plan's own copy of the
  // values, one array per lower offset + the diagonal, made by
  // spmv_hip_csr_plan_bake_values_*; used when a launch passes the baked
  // pointers
  void* sdia_val = nullptr;       // (slat_nd + 1) arrays of sdia_len entries
  uint8_t* sdia_cmask = nullptr;  // per row: own bits 0..2, column bits 4..6
  int64_t sdia_len = 0;
  int sdia_elem = 0;              // sizeof the baked value type
  const void* sdia_values0 = nullptr;
  const void* sdia_diag0 = nullptr;
  int sdia = 0;                   // use it (plan_set "sdia")
  int sdia_nd = 0;                // lower offsets ...
  int sdia_U[3] = {0, 0, 0};      // ... their row distances, descending
  int sdia_general = 0;           // baked from a GENERAL matrix (the kernel sums
                                  // in the general order): 1 = found symmetric,
                                  // lower half stored; 2 = not symmetric, FULL
                                  // form with arrays for the upper entries too
  // CONSTANT diagonals: no copy of the values -- the mask and one number per
  // diagonal (lower k | diagonal | upper k; sdia_val is a 64-byte marker)
  int sdia_const = 0;
  double sdia_cval[7] = {};
  double sdia32_cval[7] = {}; // ... of the fp32 values of the mixed SpMV
  // ... of a 3-D lattice: R lines per lane (csr_const_dia_tile_kernel), with a
  // plane-walk table of its own over the kernel's index space
  int sdia_tile = 0;          // R in use (0 / 1: the one-line kernel)
  int sdia_tile_blocks_per_cu = 8; // (R = 4: four resident, the rest queue)
  int32_t* sdia_tile_table = nullptr;
  int sdia_tile_slots = 0, sdia_tile_grid = 0, sdia_tile_segments = 0;
  // ... and the fp32 copy for the mixed-precision SpMV (general, fp64 plans)
  void* sdia32_val = nullptr;
  uint8_t* sdia32_cmask = nullptr;
  const void* sdia32_values0 = nullptr;
  int sdia_chain = 1;             // plane chain where the geometry allows
  // non-temporal streams (bit mask, see sdia_geom).  512^3: ring planes and
  // diagonal -0.5 %, y stores +-0, the far windows two workgroups share +10 %
  int sdia_nt = 3;
  // Wide diagonal form (spmv_wdia.hip): a general matrix on <= 32 diagonals
  // that the diagonal form above refused (more than 3 lower offsets): the
  // plan's copy of the values by offset + a 32-bit presence mask per row
  void* wdia_val = nullptr;       // wdia_K arrays of wdia_len entries
  uint32_t* wdia_mask = nullptr;
  int64_t wdia_len = 0;
  int wdia_elem = 0;              // sizeof the baked value type
  int wdia_K = 0;
  int wdia_narr = 0;              // arrays kept: K (full) or the offsets <= 0
                                  // (HALF form: the matrix is symmetric)
  int32_t wdia_D[kWdiaMaxOff] = {};
  int32_t wdia_A[kWdiaMaxOff] = {}; // array and row shift of offset k
  int32_t wdia_S[kWdiaMaxOff] = {};
  int wdia_const = 0;               // constant diagonals: no arrays, one number
  double wdia_cval[kWdiaMaxOff] = {};   // per offset (wdia_val is a marker)
  double wdia32_cval[kWdiaMaxOff] = {}; // ... of the mixed SpMV's fp32 values
  // ... a constant 27-point box stencil: R lattice lines per lane
  // (csr_box27_const_kernel), with its own plane-walk table
  int wdia_box = 0; // R in use (0: the general kernel)
  int wdia_hbox = 0;      // half form of a box with varying values: the marched
                          // kernel (csr_box27_half_kernel) takes it
  int wdia_hbox_segs = 0; // runs of planes (0: about one unit per CU)
  int wdia_box_P = 0, wdia_box_L = 0;
  int wdia_box_blocks_per_cu = 8;
  int32_t* wdia_box_table = nullptr;
  int wdia_box_slots = 0, wdia_box_grid = 0, wdia_box_segments = 0;
  // its own plane-walk table (the offsets' widest cluster = the plane distance;
  // the half form reads the plane ahead and finds it in the L2 one step later)
  int32_t* wdia_zw_table = nullptr;
  int wdia_zw_slots = 0, wdia_zw_grid = 0, wdia_zw_segments = 0;
  int64_t wdia_d2 = 0;
  int wdia_zwalk = 1; // use it (plan_set "wdia_zwalk")
  int wdia_blocks_per_cu = kBlocksPerCU;
  const void* wdia_values0 = nullptr;
  void* wdia32_val = nullptr;     // ... and the fp32 copy of the mixed SpMV
  const void* wdia32_values0 = nullptr;
  int wdia = 0;                   // use it (plan_set "wdia")
  int wdia_xcd_group = 4;         // consecutive row blocks per XCD (0 = off;
                                  // 27-point 256^3: 0.820 -> 0.803 ms)
  // Sliced jagged form (spmv_sjds.hip): general matrices without lattice
  // structure -- slices of 64 rows stored as jagged diagonals (lane = row), x
  // staged in LDS per block of sj_wpb slices, 16-bit column codes
  int32_t* sj_lenperm = nullptr; // per jagged lane: (row length << 6) | row in slice
  int32_t* sj_blk = nullptr;     // per block: staged chunks, 32-bit codes?
  int32_t* sj_chunks = nullptr;  // per block: sj_stride chunk numbers
  unsigned char* sj_codes = nullptr; // column codes in jagged order
  void* sj_val = nullptr;        // the values in jagged order (plan_bake_values)
  const void* sj_values0 = nullptr;
  void* sj_val32 = nullptr;      // ... and their fp32 twin (plan_bake_values_f32f64:
  const void* sj32_values0 = nullptr; // the mixed-precision SpMV)
  int sj_elem = 0;               // sizeof the baked value type
  int sj = 0;                    // use it (plan_set "sjds")
  bool sj_wanted = false;        // plan_bake_values builds it when the diagonal
                                 // forms refuse the matrix
  uint32_t* sj_ubase = nullptr;  // first unit of every slice (+ the total)
  int sj_unit = 1;               // entries per lane and step (1, 2, 4)
  int64_t sj_units = 0;          // units in the jagged arrays
  int sj_wpb = 0;                // slices (waves) per block: 4, 8, 16
  int sj_sigma = 0;              // 16-slice blocks sorted by length across the block,
                                 // a wave takes two slices (spmv_sjds.hip)
  int sj_nblk = 0, sj_maxk = 0, sj_stride = 0, sj_wide_alloc = 0;
  int64_t sj_far = 0, sj_sumk = 0; // entries gathered from memory; staged chunks
  int32_t* sj_long_rows = nullptr; // rows the slices leave out (one wave each)
  int sj_nlong = 0, sj_long_thr = 0;
  int sj_long_sorted = 0; // their columns ascend: x by LDS panels
  int sj_long_panels = 1; // use that (plan_set "sj_long_panels")
  // ... by the table-driven kernel (csr_sjds_longt_kernel): per supergroup of
  // long rows its first column and number of panels, per row and panel
  // boundary the row's first entry at or behind it
  int32_t* sj_lt_cmin = nullptr;
  int32_t* sj_lt_np = nullptr;
  int64_t* sj_lt_off = nullptr;
  int32_t* sj_lt_tab = nullptr;
  uint16_t* sj_lt_codes = nullptr; // the listed rows' columns: positions in their panels
  int64_t* sj_lt_coff = nullptr;   // ... per listed row its first code
  int64_t sj_lt_codes_n = 0;
  int64_t sj_lt_entries = 0;
  int sj_lt_nsg = 0;
  int sj_long_table = 1;  // use that (plan_set "sj_long_table")
  // symmetric storage of a matrix without lattice structure: the MERGED matrix
  // (a row's stored lower entries followed by its column's entries in the
  // reference's order; the transposed map delivers those) in the sliced jagged
  // form, in a plan of its own; its CSR arrays and the positions of its values
  // in the caller's array are this plan's
  spmv_hip_csr_plan* sjt = nullptr;
  int32_t* sjv_ptr = nullptr;
  int32_t* sjv_col = nullptr;
  int32_t* sjv_map = nullptr;
  const void* sj_diag0 = nullptr;
  int sym_sj = 0;         // its copy of the values is baked
  int sj_phases = 3;             // measurement only: 1 = long rows, 2 = slices
  int sj_blocks_per_cu = 0;      // 0 = what the LDS footprint allows
  int sj_xcd_group = 8;          // consecutive blocks per XCD (0 = off)
  int32_t* row_list = nullptr; // ROWLIST: device list of non-empty rows
  int32_t num_listed = 0;
  int nt_store = 0; // non-temporal y stores
  int plan_us = 0;  // wall time of plan creation (analysis kernels included)
  int plan_mem_us = 0; // ... of it (and of later bakes): inside hipMalloc / hipFree
  int values_changed_us = 0; // ... of the last spmv_hip_csr_plan_values_changed
  // The arrays the plan analysed.  Every form beyond the plain gather kernels
  // bakes their CONTENT in (offsets, masks, row lists, transposed map), so a
  // launch with other arrays of the same shape would silently use the wrong
  // structure: it is refused (SPMV_HIP_EINVAL).
  const int32_t* rowptr0 = nullptr;
  const int32_t* colind0 = nullptr;
  // arrays the caller has freed (spmv_hip_csr_plan_release_matrix): bit 0 =
  // colind, bit 1 = values -- compared, never read, from then on
  int released = 0;
  // set while plan_values_changed re-runs the value-dependent checks: no form
  // that was not there before is built (no allocation inside that call)
  int no_new_forms = 0;
  bool structure_baked() const
  {
    return row_list || lx_lidx || lat_tab || slat_mask || t_ptr || lxw_rec
           || xw_rec || wdia_val || sj_lenperm;
  }
  // ROWBLOCK "LX" form: LDS-staged x windows + 16-bit local column indices
  // (csr_rowblock_lx_kernel); built by plan_create when most row blocks qualify
  uint16_t* lx_lidx = nullptr;
  int32_t* lx_tab = nullptr; // kLxRec ints per row block
  int lx = 0;            // use it (plan_set "lx")
  int lx_chunks = 2;     // 16-byte value loads per lane per tile (1 or 2):
                         // 2 = half the barriers, measured +7-8 % at every size
  int lx_staged = 0;     // row blocks that take the staged path
  int lx_blocks = 0;     // row blocks analysed
  // ... and its LDS-DMA kernel (spmv_lxw.hip): values, 16-bit offsets and x
  // windows arrive by LDS-DMA one row block ahead; built when the context
  // option "lx_dma" is on (the staged layout differs: windows padded to whole
  // DMA pieces)
  int32_t* lxw_rec = nullptr; // kLxwRec ints per row block
  int lxw = 0;                // use it (plan_set "lxw")
  int lxw_max_cnt = 0;        // most entries in a staged row block
  int lxw_max_pieces = 0;     // most staged pieces of a row block
  int lxw_blocks_per_cu = 0;  // 0 = what the LDS footprint allows
  // XW: the LDS-DMA kernel on the caller's CSR arrays as they are (values,
  // 32-bit column indices), x windows staged: what a plan without lattice / LX
  // / sliced jagged form runs instead of the gather kernel (spmv_lxw.hip)
  int32_t* xw_rec = nullptr; // kXwRec ints per row block
  int xw = 0;                // use it (plan_set "xw")
  int xw_staged = 0;         // row blocks whose windows are staged
  int xw_max_cnt = 0, xw_max_pieces = 0;
  // ... or the gather kernel, whichever the FIRST LAUNCHES show to be faster
  // on this device: the two stream the same arrays at rates a few per cent
  // apart and which one leads differs from box to box (XwProbe, spmv_csr.hip)
  struct XwProbe* xw_probe = nullptr;
  // Lattice form (spmv_lat.hip): every row block's columns are row + one of
  // <= 8 constant offsets => no index stream, values arrive by LDS-DMA
  int32_t* lat_tab = nullptr;  // kLatRec ints per row block: count, offsets
  uint8_t* lat_mask = nullptr; // per row: bit k = offset k of its block present
  int lat = 0;                 // use it (plan_set "lat")
  int lat_blocks = 0;          // row blocks in lattice form (all, or lat == 0)
  int lat_blocks_per_cu = 4;   // 2 x 17 KiB of LDS per workgroup
  int lat_xcd_group = 0;       // consecutive row blocks per XCD (0 = off)
  int lat_chain = 1;           // hand x from plane to plane (lattice_d2)

  int lattice_d1 = 0, lattice_d2 = 0; // line and plane distance (rows) of a
                                      // 3-D lattice, 0 = none found

  // Plane-walk order (spmv_zwalk_order_build): every workgroup of a grid of
  // zw_grid walks one 256-row column of the lattice from plane to plane, so the
  // windows one plane ahead (x, and the symmetric forms' column values) are
  // the ones it -- or a neighbour on the same XCD -- loads as its own one step
  // later.  The table is tied to the grid size.
  int32_t* zw_table = nullptr;
  int zw_slots = 0;
  int zw_grid = 0;
  int zw_segments = 0; // runs the plane axis is cut into
  int64_t zw_d2 = 0;   // plane distance the last build was asked for
  int zwalk = 0;       // use the table (plan_set "zwalk")

  RowBlockOrder row_block_order(int nrb) const
  {
    RowBlockOrder o;
    o.table = nullptr;
    o.num_slots = 0;
    o.xcd_group = xcd_group;
    o.num_row_blocks = nrb;
    o.nt_store = nt_store;
    return o;
  }
};


// XW or the gather kernel: the state of a plan's first-launches probe
// (xw_probe_pick, spmv_csr.hip)
struct XwProbe {
  int launches = 0;
  int decided = 0;  // the choice is fixed
  int use_xw = 1;   // ... to this
  hipEvent_t ev[4][2] = {};
  float us_xw = 0.f, us_gather = 0.f;
};

// --- cross-file entry points (one definition each) -------------------------
// spmv_csr.hip
int spmv_rowblock_grid(const spmv_hip_csr_plan* pl); // grid of the row-block kernels
// spmv_csr_forms.hip: plan-time builders of the general plans
int spmv_build_row_list(spmv_hip_csr_plan* pl, const int32_t* rowptr);
void spmv_free_lx(spmv_hip_csr_plan* pl);
void spmv_free_xw(spmv_hip_csr_plan* pl);
int spmv_build_lx(spmv_hip_csr_plan* pl, const int32_t* rowptr, const int32_t* colind);
// the XW records, + the plane-walk order when the matrix sits on a 3-D grid
int spmv_build_xw_and_walk(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                           const int32_t* colind);
bool spmv_xw_applies(const spmv_hip_csr_plan* pl); // may this plan stage x windows?
void xw_probe_free(spmv_hip_csr_plan* pl);
void xw_probe_drop_events(XwProbe* pb);
// spmv_csr_plan.hip: the arrays a launch with the baked pointers does not read
int spmv_plan_owned_mask(const spmv_hip_csr_plan* pl);
// spmv_sym.hip
int spmv_run_symmetric_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                           const int32_t* rowptr, const int32_t* colind,
                           const double* values, const double* diagonal,
                           double alpha, const double* in, double beta,
                           double* out, DotOut dot);
int spmv_run_symmetric_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                           const int32_t* rowptr, const int32_t* colind,
                           const float* values, const float* diagonal,
                           float alpha, const float* in, float beta,
                           float* out);
int spmv_symt_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                    const int32_t* colind);
void spmv_symt_free(spmv_hip_csr_plan* pl);
// spmv_csr_forms.hip: (re)build the plane-walk table for planes `d2` rows apart, a
// grid of `grid` workgroups and `segments` runs along the plane axis (0 =
// choose); leaves zw_table null when the lattice is too small for it to pay
// (unless `force`)
int spmv_zwalk_order_build(spmv_hip_csr_plan* pl, int64_t d2, int grid,
                           int segments, bool force);
void spmv_zwalk_free(spmv_hip_csr_plan* pl);
// the same table for a kernel that keeps its own (device array, hipFree; null
// when the lattice is too small for one to pay)
int spmv_zwalk_table_device(const spmv_hip_csr_plan* pl, int64_t rows,
                            int64_t d2, int grid, int segments, bool force,
                            int32_t** table, int* slots, int* segs);
int spmv_walk_grid(const spmv_hip_csr_plan* pl); // grid of the plan's lattice kernel
// spmv_symlat.hip
int spmv_slat_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                    const int32_t* colind);
void spmv_slat_free(spmv_hip_csr_plan* pl);
int spmv_slat_grid(const spmv_hip_csr_plan* pl); // launch grid
int spmv_slat_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                      const int32_t* rowptr, const double* values,
                      const double* diagonal, double alpha, const double* in,
                      double beta, double* out, DotOut dot);
int spmv_slat_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                      const int32_t* rowptr, const float* values,
                      const float* diagonal, float alpha, const float* in,
                      float beta, float* out);
// spmv_symdia.hip
void spmv_sdia_free(spmv_hip_csr_plan* pl);
int spmv_sdia_tile_build(spmv_hip_csr_plan* pl, int R, int segments, bool force);
int spmv_sdia_bake_f64(spmv_hip_csr_plan* pl, const double* values,
                       const double* diagonal, hipStream_t st);
int spmv_sdia_bake_f32(spmv_hip_csr_plan* pl, const float* values,
                       const float* diagonal, hipStream_t st);
int spmv_sdia_grid(const spmv_hip_csr_plan* pl); // launch grid without a table
int spmv_sdia_bake_f32f64(spmv_hip_csr_plan* pl, const float* values32,
                          hipStream_t st);
int spmv_sdia_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                         double alpha, const double* in, double beta,
                         double* out, DotOut dot);
int spmv_sdia_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                      const double* in, double beta, double* out, DotOut dot);
int spmv_sdia_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st, float alpha,
                      const float* in, float beta, float* out);
// spmv_lxw.hip
int spmv_lxw_grid(const spmv_hip_csr_plan* pl, int elem_bytes);
int spmv_lxw_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const int32_t* colind,
                     const double* values, double alpha, const double* in,
                     double beta, double* out, DotOut dot);
int spmv_xw_grid(const spmv_hip_csr_plan* pl, int elem_bytes);
int spmv_xw_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                    const int32_t* rowptr, const int32_t* colind,
                    const double* values, double alpha, const double* in,
                    double beta, double* out, DotOut dot);
int spmv_xw_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                    const int32_t* rowptr, const int32_t* colind,
                    const float* values, float alpha, const float* in,
                    float beta, float* out);
int spmv_lxw_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const int32_t* colind,
                     const float* values, float alpha, const float* in,
                     float beta, float* out);
// spmv_wdia.hip
void spmv_wdia_free(spmv_hip_csr_plan* pl);
int spmv_wdia_walk_build(spmv_hip_csr_plan* pl, int segments, bool force);
int spmv_wdia_box_build(spmv_hip_csr_plan* pl, int R, int segments, bool force);
int spmv_wdia_hbox_build(spmv_hip_csr_plan* pl, int on);
int spmv_wdia_bake_f64(spmv_hip_csr_plan* pl, const double* values,
                       hipStream_t st); // values == nullptr: drop the copy
int spmv_wdia_bake_f32(spmv_hip_csr_plan* pl, const float* values,
                       hipStream_t st);
int spmv_wdia_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                      const double* in, double beta, double* out, DotOut dot);
int spmv_wdia_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st, float alpha,
                      const float* in, float beta, float* out);
int spmv_wdia_bake_f32f64(spmv_hip_csr_plan* pl, const float* values32,
                          hipStream_t st); // fp32 copy for the mixed SpMV
int spmv_wdia_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                         double alpha, const double* in, double beta,
                         double* out, DotOut dot);
// spmv_sjds.hip
int spmv_sjds_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                    const int32_t* colind, int wpb_force, int unit_force,
                    int no_long);
void spmv_sjds_free(spmv_hip_csr_plan* pl);
int spmv_sjds_long_entries(spmv_hip_ctx* ctx, int32_t num_rows, int64_t nnz,
                           const int32_t* rowptr, int64_t* entries, hipStream_t st);
// values == nullptr: drop the copy; map: entry e is values[map[e]]
int spmv_sjds_bake_f64(spmv_hip_csr_plan* pl, const double* values,
                       const int32_t* map, hipStream_t st);
int spmv_sjds_bake_f32(spmv_hip_csr_plan* pl, const float* values, const int32_t* map,
                       hipStream_t st);
int spmv_sjds_bake_f32f64(spmv_hip_csr_plan* pl, const float* values32, hipStream_t st);
int spmv_sjds_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                         const double* in, double beta, double* out, DotOut dot);
int spmv_sjds_sym_merge(const spmv_hip_csr_plan* pl, int long_thr, int32_t** vptr,
                        int32_t** vcol, int32_t** vmap, int64_t* total, hipStream_t st);
int spmv_sjds_sym_build(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan, hipStream_t st);
int spmv_sjds_run_sym_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                          const double* diagonal, double alpha, const double* in,
                          double beta, double* out, DotOut dot);
int spmv_sjds_run_sym_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                          const float* diagonal, float alpha, const float* in,
                          float beta, float* out);
int spmv_sjds_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                      const double* in, double beta, double* out, DotOut dot);
int spmv_sjds_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st, float alpha,
                      const float* in, float beta, float* out);
// spmv_lat.hip
int spmv_lat_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                   const int32_t* colind);
void spmv_lat_free(spmv_hip_csr_plan* pl);
int spmv_lat_grid(const spmv_hip_csr_plan* pl); // launch grid
int spmv_lat_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const double* values, double alpha,
                     const double* in, double beta, double* out, DotOut dot);
int spmv_lat_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const float* values, float alpha,
                     const float* in, float beta, float* out);
int spmv_lat_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                        const int32_t* rowptr, const float* values,
                        double alpha, const double* in, double beta,
                        double* out, DotOut dot);
