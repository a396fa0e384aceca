"""PROBE (round 6): the plan of the marched exchange, built on the HOST with numpy
(see march_probe.hip for what the kernel does with it).  Input: symmetric
storage -- strictly lower CSR (rowptr, colind, values) + diagonal.

Tiles: level l = row // S, position j = (row % S) // B: tile (l, j) holds the
rows [l S + j B, min(l S + (j + 1) B, (l + 1) S, N)).  Chain j walks the levels;
a UNIT is a run of `lseg` levels of one chain: one workgroup, one step per tile
plus a drain step.  The column entry (r, c) of row c (r > c its source row) is
  captured near   r in the tile of c                        -> stream Bc (2 B)
  streamed early  r before the first row of the next tile   -> stream Bs (10 B)
  captured far    r in the next tile of c's chain AND unit  -> stream Cc (2 B)
  streamed late   the rest                                  -> stream Cs (10 B)
in this order, which is ascending (r, position) because the four classes are
consecutive ranges of r.  A captured entry's code is the position of the SOURCE
entry in its tile's lower stream A (= where the producer's lane leaves its
product in the stash); a streamed entry's code is the staged position of x_r.
"""
import numpy as np

B = 1024
SLICES = 16


def far_offset(rowptr, colind, n_sample=200_000):
    """the dominant far offset row - col of the stored lower block (0: none)"""
    N = len(rowptr) - 1
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
    step = max(1, len(rows) // n_sample)
    d = (rows - colind)[::step]
    d = d[d > 4 * B]
    if len(d) < 100:
        return 0
    return int(np.median(d))


def build(rowptr, colind, values, diag, S, lseg, stash_cap, x_cap, sort_rows=True):
    N = len(rowptr) - 1
    nnz = len(colind)
    rowptr = rowptr.astype(np.int64)
    colind = colind.astype(np.int64)
    nlev = (N + S - 1) // S
    nchain = (S + B - 1) // B
    nseg = (nlev + lseg - 1) // lseg

    def tile_of(row):
        lev = row // S
        j = (row % S) // B
        return lev, j
    # ---- tiles in unit order: (chain, segment, level) -------------------------
    lev_all, j_all = np.meshgrid(np.arange(nlev), np.arange(nchain), indexing="ij")
    lev_all, j_all = lev_all.ravel(), j_all.ravel()
    row0 = lev_all * S + j_all * B
    row1 = np.minimum(np.minimum(row0 + B, (lev_all + 1) * S), N)
    keep = row1 > row0
    lev_all, j_all, row0, row1 = lev_all[keep], j_all[keep], row0[keep], row1[keep]
    seg_all = lev_all // lseg
    order = np.lexsort((lev_all, seg_all, j_all))
    lev_t, j_t, seg_t = lev_all[order], j_all[order], seg_all[order]
    row0_t, row1_t = row0[order], row1[order]
    ntiles = len(row0_t)
    # tile id of (lev, j)
    tid_of = -np.ones((nlev, nchain), dtype=np.int64)
    tid_of[lev_t, j_t] = np.arange(ntiles)
    unit_key = j_t * nseg + seg_t
    ukeys, ufirst, ucount = np.unique(unit_key, return_index=True, return_counts=True)
    nunits = len(ukeys)
    unit_of_tile = np.searchsorted(ukeys, unit_key)
    # steps: per unit its tiles + a drain step
    unit_step0 = np.zeros(nunits + 1, dtype=np.int64)
    np.cumsum(ucount + 1, out=unit_step0[1:])
    nsteps = int(unit_step0[-1])
    step_tile = -np.ones(nsteps, dtype=np.int64)
    step_of_tile = unit_step0[unit_of_tile] + (np.arange(ntiles) - ufirst[unit_of_tile])
    step_tile[step_of_tile] = np.arange(ntiles)
    # the tile finished at a step (phase C): the previous step's tile
    # ---- rows ------------------------------------------------------------------
    r = np.arange(N, dtype=np.int64)
    rl, rj = tile_of(r)
    row_tile = tid_of[rl, rj]
    row_loc = r - row0_t[row_tile]
    lens_low = np.diff(rowptr)
    # ---- entries: the lower part (stream A) -----------------------------------
    e_row = np.repeat(r, lens_low)
    e_k = np.arange(nnz, dtype=np.int64) - rowptr[e_row]
    e_col = colind
    # ---- the column part: entries sorted by (col, row, position) --------------
    corder = np.lexsort((np.arange(nnz), e_row, e_col))
    c_tgt = e_col[corder]         # the row that receives (c)
    c_src = e_row[corder]         # the source row (r)
    c_ent = corder                # the source entry (position in the lower arrays)
    tgt_tile = row_tile[c_tgt]
    src_tile = row_tile[c_src]
    nxt_row0 = (lev_t[tgt_tile] + 1) * S + j_t[tgt_tile] * B  # first row of the next tile
    near_in = src_tile == tgt_tile
    nxt_tid = np.where(lev_t[tgt_tile] + 1 < nlev,
                       tid_of[np.minimum(lev_t[tgt_tile] + 1, nlev - 1), j_t[tgt_tile]], -1)
    far_in = (src_tile == nxt_tid) & (nxt_tid >= 0) \
        & (unit_of_tile[np.maximum(nxt_tid, 0)] == unit_of_tile[tgt_tile])
    early = ~near_in & ~far_in & (c_src < nxt_row0)
    late = ~near_in & ~far_in & ~early
    cls = np.where(near_in, 1, np.where(early, 2, np.where(far_in, 3, 4)))
    # within a target row the classes must ascend (they are ranges of r)
    same = c_tgt[1:] == c_tgt[:-1]
    assert np.all(cls[1:][same] >= cls[:-1][same]), "classes out of order"
    # rank of a column entry inside its (target row, class)
    key = c_tgt * 8 + cls
    first = np.r_[True, key[1:] != key[:-1]]
    grp_start = np.maximum.accumulate(np.where(first, np.arange(nnz), 0))
    c_k = np.arange(nnz) - grp_start
    # per row lengths of the five streams
    lenA = lens_low
    lens = [lenA]
    for c in (1, 2, 3, 4):
        lens.append(np.bincount(c_tgt[cls == c], minlength=N))
    lens = np.stack(lens, axis=1)  # [N, 5]
    assert lens.max() < 256, "a stream of a row is longer than a byte"
    # ---- lanes: rows of a tile sorted by lower length (descending, stable) ----
    if sort_rows:
        o = np.lexsort((row_loc, -lenA, row_tile))
    else:
        o = np.lexsort((row_loc, row_tile))
    # position of a row inside its tile's lane order
    tile_first = np.searchsorted(row_tile[o], np.arange(ntiles))
    lane_of_row = np.empty(N, dtype=np.int64)
    lane_of_row[o] = np.arange(N) - tile_first[row_tile[o]]
    vrow = row_tile * B + lane_of_row
    meta = np.zeros(ntiles * B, dtype=np.uint64)
    m = np.uint64(1) << np.uint64(63)
    for c in range(5):
        m = m | (lens[:, c].astype(np.uint64) << np.uint64(8 * c))
    m = m | (row_loc.astype(np.uint64) << np.uint64(40))
    meta[vrow] = m

    # ---- jagged positions: entries ordered by (tile, slice, k, lane) ----------
    def jagged(ent_row, ent_k, tile_for):
        """positions of the entries of one stream (each belongs to row ent_row,
        is its ent_k-th there, and is read at the step of tile tile_for[row])"""
        t = tile_for
        lane = lane_of_row[ent_row]
        sl = lane // 64
        order = np.lexsort((lane % 64, ent_k, sl, t))
        pos = np.empty(len(ent_row), dtype=np.int64)
        pos[order] = np.arange(len(ent_row))
        # first entry of every (tile, slice)
        ts = (t * SLICES + sl)[order]
        sb = np.searchsorted(ts, np.arange(ntiles * SLICES + 1)).astype(np.int64)
        return pos, sb
    posA, sbA = jagged(e_row, e_k, row_tile[e_row])
    streams = {}
    stash_idx = posA - sbA[row_tile[e_row] * SLICES]  # position in the tile's stream A
    assert stash_idx.max() < stash_cap, ("stash", int(stash_idx.max()))
    pos_c, sb_c = {}, {}
    for c in (1, 2, 3, 4):
        sel = cls == c
        pos_c[c], sb_c[c] = jagged(c_tgt[sel], c_k[sel], row_tile[c_tgt[sel]])
    # ---- staged x per step -----------------------------------------------------
    # (step, column) pairs every step needs: A's columns and the tile's own rows
    # at the tile's step; Bs sources at the tile's step; Cs sources one step later
    step_A = step_of_tile[row_tile[e_row]]
    need_step = [step_A, step_of_tile[row_tile]]
    need_col = [e_col, r]
    sel2, sel4 = cls == 2, cls == 4
    need_step += [step_of_tile[tgt_tile[sel2]], step_of_tile[tgt_tile[sel4]] + 1]
    need_col += [c_src[sel2], c_src[sel4]]
    ns_ = np.concatenate(need_step)
    nc_ = np.concatenate(need_col) // 16
    # own chunks first: key = (step, not own, chunk); a chunk is "own" for a step
    # when the step's tile has rows in it
    own_lo = np.full(nsteps, 1 << 40, dtype=np.int64)
    own_hi = np.full(nsteps, -1, dtype=np.int64)
    tws = step_tile[step_tile >= 0]
    own_lo[step_of_tile[tws]] = row0_t[tws] // 16
    own_hi[step_of_tile[tws]] = (row1_t[tws] - 1) // 16

    def full_key(step, chunk):
        isown = (chunk >= own_lo[step]) & (chunk <= own_hi[step])
        return step * (1 << 29) + (~isown).astype(np.int64) * (1 << 28) + chunk
    full = np.unique(full_key(ns_, nc_))
    ch_step = full >> 29
    ch_chunk = full & ((1 << 28) - 1)
    step_chunk0 = np.searchsorted(ch_step, np.arange(nsteps + 1)).astype(np.int64)
    assert np.diff(step_chunk0).max() * 16 <= x_cap, ("x", int(np.diff(step_chunk0).max()))

    def staged(step, col):
        fk = full_key(step, col // 16)
        idx = np.searchsorted(full, fk)
        assert np.all(full[idx] == fk)
        return (idx - step_chunk0[step]) * 16 + col % 16
    tiles_with_step = step_tile[step_tile >= 0]
    step_own0 = np.zeros(nsteps, dtype=np.int64)
    st_ = step_of_tile[tiles_with_step]
    step_own0[st_] = staged(st_, row0_t[tiles_with_step])
    # own rows contiguous?  (their chunks are consecutive and first)
    chk = staged(step_of_tile[row_tile], r)
    assert np.all(chk == step_own0[step_of_tile[row_tile]] + row_loc), "own rows"
    # ---- the streams' arrays ---------------------------------------------------
    SL = 4096  # slack: loads of idle lanes run past a stream's end

    def arr(n, dt):
        return np.zeros(n + SL, dtype=dt)
    a_val, a_code = arr(nnz, np.float64), arr(nnz, np.uint16)
    a_val[posA] = values
    a_code[posA] = staged(step_A, e_col)
    out = {"a_val": a_val, "a_code": a_code}
    for c, name, valued in ((1, "bc", False), (2, "bs", True), (3, "cc", False),
                            (4, "cs", True)):
        sel = cls == c
        n = int(sel.sum())
        code = arr(n, np.uint16)
        if valued:
            val = arr(n, np.float64)
            val[pos_c[c]] = values[c_ent[sel]]
            st = step_of_tile[tgt_tile[sel]] + (1 if c == 4 else 0)
            code[pos_c[c]] = staged(st, c_src[sel])
            out[name + "_val"] = val
        else:
            code[pos_c[c]] = stash_idx[c_ent[sel]]
        out[name + "_code"] = code
    out.update({
        "nunits": nunits, "unit_step0": unit_step0.astype(np.int32),
        "step_tile": step_tile.astype(np.int32),
        "step_chunk0": step_chunk0.astype(np.int32),
        "step_own0": step_own0.astype(np.int32),
        "chunks": ch_chunk.astype(np.int32),
        "tile_row0": row0_t.astype(np.int32), "meta": meta,
        "sb0": sbA.astype(np.uint32), "sb1": sb_c[1].astype(np.uint32),
        "sb2": sb_c[2].astype(np.uint32), "sb3": sb_c[3].astype(np.uint32),
        "sb4": sb_c[4].astype(np.uint32), "diag": diag,
        "stats": {"rows": N, "stored": nnz, "tiles": ntiles, "units": nunits,
                  "steps": nsteps,
                  "captured_near": int(near_in.sum()), "captured_far": int(far_in.sum()),
                  "streamed_early": int(early.sum()), "streamed_late": int(late.sum()),
                  "max_chunks_per_step": int(np.diff(step_chunk0).max()),
                  "max_stash": int(stash_idx.max()) + 1,
                  "staged_x_elements": int(len(ch_chunk)) * 16}})
    return out


def pack_v2(p):
    """the arrays of march_probe2.hip from build()'s: step descriptors, stream
    bases packed per (tile, slice), the diagonal per lane, stream A's code with
    the entry's stash position in its upper half"""
    nsteps = len(p["step_tile"])
    ntiles = len(p["tile_row0"])
    steps = np.zeros((nsteps + 2, 8), dtype=np.int32)
    steps[:, 0] = -1
    steps[:nsteps, 0] = p["step_tile"]
    steps[:nsteps, 1] = p["step_chunk0"][:-1]
    steps[:nsteps, 2] = np.diff(p["step_chunk0"])
    steps[:nsteps, 3] = p["step_own0"]
    t = np.maximum(p["step_tile"], 0)
    steps[:nsteps, 4] = p["tile_row0"][t]
    steps[nsteps:, 1] = p["step_chunk0"][-1]
    sbp = np.zeros((ntiles * SLICES, 8), dtype=np.uint32)
    for c in range(5):
        sbp[:, c] = p["sb%d" % c][:-1]
    meta = p["meta"]
    valid = (meta >> np.uint64(63)) != 0
    loc = ((meta >> np.uint64(40)) & np.uint64(0xffff)).astype(np.int64)
    tile = np.arange(len(meta)) // B
    row = np.where(valid, p["tile_row0"][tile] + loc, 0)
    dval = np.where(valid, p["diag"][row], 0.0)
    nA = len(p["a_code"])
    # (position of an entry in its tile's stream: its index minus the tile's base)
    tile_of_pos = np.searchsorted(p["sb0"][::SLICES].astype(np.int64),
                                  np.arange(nA), side="right") - 1
    tile_of_pos = np.clip(tile_of_pos, 0, ntiles - 1)
    idx = np.arange(nA) - p["sb0"][::SLICES].astype(np.int64)[tile_of_pos]
    idx = np.clip(idx, 0, 65535)
    a_code = p["a_code"].astype(np.uint32) | (idx.astype(np.uint32) << np.uint32(16))
    chunks = np.concatenate([p["chunks"], np.zeros(1024, dtype=np.int32)])
    return {"steps": steps, "sbp": sbp, "dval": dval, "a_code32": a_code,
            "chunks_padded": chunks}


def bytes_moved(p):
    s = p["stats"]
    cap = s["captured_near"] + s["captured_far"]
    strm = s["streamed_early"] + s["streamed_late"]
    return (s["stored"] * 10 + cap * 2 + strm * 10 + s["rows"] * (8 + 8 + 8)
            + s["staged_x_elements"] * 8)
