"""tests/tile_emulator.py — numpy walk over a row-block stream image, unit for unit what spmv_rowblock_kernel does.

Test infrastructure: lets the CPU-only suite check the load-time re-tiling (stream_tiles.cpp) and the
kernel's element semantics against the oracle without a GPU, for both stream formats.  Never imported by the product.
"""
import numpy as np

WAVE, CONSUMERS, SUB_TILE = 64, 14, 8192
SWEEP_WAVES = 8              # SWEEP: wavefronts per workgroup (stream_tiles.h: kSweepWaves)
CHUNK_BYTES = 512            # PAIRS: 64 x {u32 value, u32 row << 16 | col}
RECORD_BYTES = 768           # DELTA: two slots per lane: 64 x {u32 value A, u32 value B}, then 64 x {u16 gap A, u16 gap B}
BRIDGE = 0xFFFF


def _q_mul(a, b):
    wide = a.astype(np.uint64) * b.astype(np.uint64)
    r = (wide >> np.uint64(24)) + ((wide >> np.uint64(23)) & np.uint64(1))
    return np.minimum(r, np.uint64(0xFFFFFFFF))


def _accumulate(ys, is_float, row, val, xv):
    if is_float:
        np.add.at(ys, row, (val.view(np.float32) * xv.view(np.float32)).astype(np.float32).astype(np.float64))
    else:
        np.add.at(ys, row, _q_mul(val, xv))


def _chunk(image, at, aux24):
    """(value words, position words) of one 64-slot chunk: 512 bytes of {value, position} pairs, or -- 24-bit position words --
    448 bytes = 64 value dwords followed by 64 x 3 bytes."""
    if not aux24:
        c = image[at: at + CHUNK_BYTES].view(np.uint32).reshape(WAVE, 2)
        return c[:, 0], c[:, 1]
    val = image[at: at + 256].view(np.uint32)
    a = image[at + 256: at + 448].reshape(WAVE, 3).astype(np.uint32)
    return val, a[:, 0] | (a[:, 1] << 8) | (a[:, 2] << 16)


def _block_pairs(image, blk, units, x_words, ys, is_float, aux24=False):
    nrows = int(blk["nrows"])
    cb, shift, mask = (448, 13, 8191) if aux24 else (CHUNK_BYTES, 16, 0xFFFF)
    assert not aux24 or nrows <= 2046
    step = [0] * CONSUMERS
    for u in range(int(blk["unit_begin"]), int(blk["unit_end"])):
        unit = units[u]
        col0, ncols = int(unit["col0"]), int(unit["ncols"])
        xt = x_words[col0: col0 + ncols]
        for w in range(CONSUMERS):
            end = int(unit["end_step"][w])
            base = int(blk["wave_offset"][w])
            for s in range(step[w], end):
                val, cr = _chunk(image, base + s * cb * CONSUMERS, aux24)
                col, row = (cr & mask).astype(np.int64), (cr >> shift).astype(np.int64)
                assert (col < ncols).all() and (row <= nrows).all()
                assert (val[row == nrows] == 0).all()                 # padding aims a zero at the spare accumulator
                _accumulate(ys, is_float, row, val, xt[col])
            step[w] = end


def _block_delta(image, blk, units, x_words, ys, is_float):
    nrows = int(blk["nrows"])
    step = [0] * CONSUMERS
    for u in range(int(blk["unit_begin"]), int(blk["unit_end"])):
        unit = units[u]
        col0, ncols = int(unit["col0"]), int(unit["ncols"])
        xt = np.zeros(SUB_TILE, dtype=np.uint32)                        # the LDS buffer: stale words past ncols
        xt[:ncols] = x_words[col0: col0 + ncols]
        for w in range(CONSUMERS):
            end = int(unit["end_step"][w])
            base = int(blk["wave_offset"][w])
            assert end >= step[w]                                      # a (unit, wavefront) has no records, or head + >= 1 slots
            pos = None
            for s in range(step[w], end):
                rec = image[base + s * RECORD_BYTES: base + (s + 1) * RECORD_BYTES]
                vals = rec[:WAVE * 8].view(np.uint32).reshape(WAVE, 2)
                gaps = rec[WAVE * 8:].view(np.uint16).reshape(WAVE, 2).astype(np.int64)
                for half in (0, 1):
                    val, gap = vals[:, half], gaps[:, half]
                    if pos is None:                                     # head slot (slot A of the run's first record): absolute start position per lane
                        assert half == 0
                        pos = val.astype(np.int64)
                        assert (pos <= nrows * SUB_TILE).all()
                        continue
                    pos = (pos + gap) & 0xFFFFFFFF                      # u32 arithmetic like the kernel
                    row, col = pos >> 13, pos & (SUB_TILE - 1)
                    live = gap != BRIDGE
                    if is_float:
                        # bridge slots add a literal zero at min(row, nrows); live slots must be real positions
                        assert (row[live] < nrows).all() and (col[live] < ncols).all()
                        _accumulate(ys, True, row[live], val[live], xt[col[live]])
                    else:
                        # no live test in the fixed-point kernel: every slot multiplies; dead slots carry value 0 and must
                        # still aim at an accumulator of the block (or the spare one)
                        assert (row <= nrows).all() and (val[~live] == 0).all()
                        real = live & (val != 0)
                        assert (row[real] < nrows).all() and (col[real] < ncols).all()
                        _accumulate(ys, False, row, val, xt[col])
            step[w] = end


OWNER_RECORD_BYTES, OWNER_RECORD_STEPS, OWNER_SPARE = 1792, 4, 2047


def _owner24_step(image, base, S):
    """(value words, 24-bit position words) of step S of a wavefront's OWNER24 stream: slot S % 4 of record S // 4 -- 64 lanes x 4
    value words, then 64 lanes x 4 x 3 bytes."""
    rec = image[base + (S // OWNER_RECORD_STEPS) * OWNER_RECORD_BYTES: base + (S // OWNER_RECORD_STEPS + 1) * OWNER_RECORD_BYTES]
    j = S % OWNER_RECORD_STEPS
    val = rec[:1024].view(np.uint32).reshape(WAVE, OWNER_RECORD_STEPS)[:, j]
    a = rec[1024:].reshape(WAVE, OWNER_RECORD_STEPS, 3)[:, j, :].astype(np.uint32)
    return val, a[:, 0] | (a[:, 1] << 8) | (a[:, 2] << 16)


def _block_owner(image, blk, units, x_words, ys, is_float, aux24=False):
    """OWNER (float only): per-wavefront contiguous 512-byte chunks of {value, row << 13 | col}; lane l holds a consecutive run of
    the wavefront's share, rows never decrease from lane to lane and step to step, no row is shared between wavefronts WITHIN A UNIT
    (the unit barrier orders the accumulator writes, so the shares are cut per unit), padding aims a zero at the wavefront's own spare
    accumulator nrows + w; the shares of a unit differ by at most one step unless a row is longer than a share.
    aux24 = OWNER24: the same steps in records of four (_owner24_step), 24-bit position words whose row is relative to the first row
    of the (unit, wavefront) share -- carried in the high half of Unit.end_step[w] -- with 2047 = the spare accumulator."""
    assert is_float or aux24       # fixed point: OWNER24 only (saturating 32-bit adds in any order == the exact sum, clamped once, below)
    nrows = int(blk["nrows"])
    step = [0] * CONSUMERS
    for u in range(int(blk["unit_begin"]), int(blk["unit_end"])):
        unit = units[u]
        col0, ncols = int(unit["col0"]), int(unit["ncols"])
        xt = x_words[col0: col0 + ncols]
        owner_of = {}
        for w in range(CONSUMERS):
            end, share0 = int(unit["end_step"][w]), 0
            if aux24:
                end, share0 = end & 0xFFFF, end >> 16
                if u + 1 == int(blk["unit_end"]):
                    assert int(blk["total_steps"][w]) == end
                if u == int(blk["unit_begin"]):
                    assert int(blk["first_end"][w]) == int(unit["end_step"][w])
            base = int(blk["wave_offset"][w])
            assert end >= step[w]
            if end == step[w]:
                continue
            if aux24:
                pairs = [_owner24_step(image, base, st) for st in range(step[w], end)]
            else:
                pairs = [_chunk(image, base + st * CHUNK_BYTES, False) for st in range(step[w], end)]
            val, where = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
            row, col = (where >> 13).astype(np.int64), (where & 8191).astype(np.int64)
            if aux24:
                assert ((row <= 2046) | (row == OWNER_SPARE)).all()
                assert (row != OWNER_SPARE).any() and row[row != OWNER_SPARE].min() == 0      # row_base IS the share's first row
                row = np.where(row == OWNER_SPARE, nrows + w, row + share0)
            order = row.T.reshape(-1)                                   # lane-major: lane 0's run, then lane 1's, ...
            assert (np.diff(order) >= 0).all()                          # sorted by row across (lane, step)
            pad = row == nrows + w
            assert ((row < nrows) | pad).all() and (val[pad] == 0).all() and (col[~pad] < ncols).all()
            for r in np.unique(row[~pad]):
                assert owner_of.setdefault(int(r), w) == w              # one owner per row while the unit lasts
            _accumulate(ys, is_float, row[~pad], val[~pad], xt[col[~pad]])
            step[w] = end


WAVESEG_DTYPE = np.dtype([("row_begin", "<u4"), ("row_end", "<u4"), ("g_begin", "<u4"), ("g_end", "<u4"), ("value", "<u8"), ("mask", "<u8"),
                          ("pad", "<u4", (8,))])
BITMAP_WAVES, GROUP = 16, 64


def _block_bitmap(image, blk, units, x_words, ys, is_float):
    """spmv_bitmap_kernel: every wavefront walks its run of (row, 64-column group) steps; mask bit l = column 64 g + l is set,
    its value is the next compacted one."""
    nrows, col0, gs = int(blk["nrows"]), int(blk["first_col0"]), int(blk["first_ncols"])
    assert int(blk["unit_end"]) - int(blk["unit_begin"]) == BITMAP_WAVES * 5      # run header = WaveSeg + a copy of its first 32 masks
    headers = units[int(blk["unit_begin"]): int(blk["unit_end"])].view(np.uint8).reshape(BITMAP_WAVES, 5 * 64)
    segs = np.ascontiguousarray(headers[:, :64]).view(WAVESEG_DTYPE).reshape(-1)
    head_masks = np.ascontiguousarray(headers[:, 64:]).view(np.uint64)
    masks = image[: image.size // 8 * 8].view(np.uint64)
    values = image[: image.size // 4 * 4].view(np.uint32)
    covered = np.zeros((nrows, gs), dtype=np.int32)
    for w in range(BITMAP_WAVES):
        sg = segs[w]
        r0, r1, g0, g1 = int(sg["row_begin"]), int(sg["row_end"]), int(sg["g_begin"]), int(sg["g_end"])
        assert r0 <= r1 <= nrows and g0 <= g1 <= gs
        assert r1 - r0 <= 1 or (g0 == 0 and g1 == gs)                 # several rows: whole rows only
        mp, vp = int(sg["mask"]), int(sg["value"])
        n_first = min(32, g1 - g0) if r1 > r0 else 0                    # the inline copy: the run's first masks, zero beyond
        assert np.array_equal(head_masks[w, :n_first], masks[mp: mp + n_first]) and not head_masks[w, n_first:].any()
        for r in range(r0, r1):
            covered[r, g0:g1] += 1
            m = masks[mp: mp + (g1 - g0)]
            stride = (g1 - g0 + 7) // 8 * 8 + 16                         # zero masks up to a multiple of 8, plus two batches
            assert not masks[mp + (g1 - g0): mp + stride].any()
            mp += stride
            bits = np.unpackbits(m.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little").astype(bool)    # [group, lane]
            n = int(bits.sum())
            g, lane = np.nonzero(bits)
            col = col0 + (g0 + g) * GROUP + lane
            assert (col < x_words.size).all()
            _accumulate(ys, is_float, np.full(n, r, dtype=np.int64), values[vp: vp + n], x_words[col])
            vp += n
    # every (row, group) of the block belongs to exactly one wavefront -- or to none at all: a block made of the matrix's empty padding rows
    # (float_stall rounds the rows up to a multiple of 1024) has sixteen idle runs and only writes its rows' zeros (bitmap_tiles.cpp, round 4)
    assert (covered == 1).all() or (covered == 0).all()


def _block_sweep(image, blk, x_words, ys, is_float):
    """SWEEP (stream_tiles.h): chunk k of the block = step k // 8 of wavefront k % 8, its base column in the table [wavefront][step];
    the chunks, taken in order, hold the block's elements in non-decreasing column order inside the block's column slice."""
    nrows, steps = int(blk["nrows"]), int(blk["total_steps"][0])
    stream, table_at = int(blk["wave_offset"][0]), int(blk["wave_offset"][1])
    col_lo, col_hi = int(blk["first_col0"]), int(blk["first_col0"]) + int(blk["first_ncols"])
    table = image[table_at: table_at + steps * SWEEP_WAVES * 4].view(np.uint32).reshape(SWEEP_WAVES, steps) if steps else None
    previous = -1
    for k in range(steps * SWEEP_WAVES):
        s, w = divmod(k, SWEEP_WAVES)
        val, cr = _chunk(image, stream + k * CHUNK_BYTES, False)
        row, col = (cr >> 16).astype(np.int64), int(table[w, s]) + (cr & 0xFFFF).astype(np.int64)
        real = row < nrows
        assert (row <= nrows).all() and (val[~real] == 0).all() and ((cr[~real] & 0xFFFF) == 0).all()      # padding: value 0 at the spare accumulator
        assert (col < len(x_words)).all()                                                              # padding gathers x too
        if real.any():
            assert not real[np.argmin(real):].any() if not real.all() else True                        # padding only behind the elements
            c = col[real]
            assert (np.diff(c) >= 0).all() and c[0] >= previous and col_lo <= c[0] and c[-1] < col_hi
            previous = int(c[-1])
        _accumulate(ys, is_float, row, val, x_words[col])


def run(tiles, impl, x_words, num_rows, row_part_filter=-1, y_init=None, rows_per_part=None):
    """Returns packed y words.  tiles: dict from hisparse_amd.device.build_tiles.
    row_part_filter >= 0 (hs_run_partition; needs rows_per_part = 128 x ob_bank): every workgroup starts at the first block of its chain
    that reaches into the partition (part_heads) and goes on while the next block begins in the partition or before (Block::next_part);
    only the partition's own rows receive results -- blocks may reach over partition borders since round 5."""
    is_float = impl != 0
    image, blocks, units = tiles["image"], tiles["blocks"], tiles["units"]
    delta = tiles["format"] == "delta"
    bitmap = tiles["format"] == "bitmap"
    owner = tiles["format"] in ("owner", "owner24")
    aux24 = tiles["format"] in ("pairs24", "owner24")
    sweep = tiles["format"] == "sweep"
    y = np.zeros(num_rows, dtype=np.uint32) if y_init is None else y_init.copy()
    slices = int(tiles.get("col_slices", 1))
    filtered = row_part_filter >= 0
    if filtered:
        assert rows_per_part, "a partition run needs rows_per_part"
        part_lo, part_hi = row_part_filter * rows_per_part, min(num_rows, (row_part_filter + 1) * rows_per_part)
    # one slice: the kernel writes y itself (a partition run: a side buffer, the partition's rows copied over); slices: per-slice partials
    out = (y if not filtered else np.zeros(num_rows, dtype=np.uint32)) if slices == 1 else np.zeros(slices * num_rows, dtype=np.uint32)
    touched = np.zeros(num_rows, dtype=bool)
    done = np.zeros(len(blocks), dtype=bool)
    if not bitmap and not sweep:
        assert 2 <= tiles["ring_buffers"] <= 4
        assert (units["ncols"] % 8 == 0).all() and (units["ncols"] > 0).all() and (units["ncols"] <= SUB_TILE).all()
    for g in range(tiles["num_workgroups"]):
        chain = [int(b) for b in tiles["block_order"][tiles["wg_first"][g]: tiles["wg_first"][g + 1]]]
        for k, b in enumerate(chain):      # the chain the kernel follows through Block::next
            assert not done[b]
            done[b] = True
            assert int(blocks[b]["next"]) == (chain[k + 1] if k + 1 < len(chain) else 0)
            assert int(blocks[b]["next_part"]) == (int(blocks[chain[k + 1]]["row_part"]) if k + 1 < len(chain) else 0xFFFFFFFF)
        if filtered:
            p = row_part_filter
            start = next((k for k, b in enumerate(chain) if int(blocks[b]["nrows"]) and int(blocks[b]["row_part"]) <= p <= int(blocks[b]["last_part"])), None)
            if start is None:
                continue
            stop = start
            while int(blocks[chain[stop]]["next_part"]) <= p:
                stop += 1
            todo = chain[start: stop + 1]
            # nothing of the partition lies outside the stretch the kernel walks
            for b in chain[:start] + chain[stop + 1:]:
                assert not (int(blocks[b]["nrows"]) and int(blocks[b]["row_part"]) <= p <= int(blocks[b]["last_part"]))
        else:
            todo = chain
        for b in todo:
            blk = blocks[b]
            nrows, row0, out0 = int(blk["nrows"]), int(blk["row0"]), int(blk["out_offset"])
            if sweep:      # the LDS holds the accumulators and nothing else: doubles / 32-bit sums + a carry bit per row
                assert ((nrows + 1) * 8 if is_float else (nrows + 1) * 4 + (nrows + 32) // 32 * 4) <= 160 * 1024
            elif owner:
                assert (nrows + CONSUMERS) * 4 + tiles["ring_buffers"] * 32768 <= 160 * 1024     # float accumulators + the x ring
            else:
                assert nrows <= (8191 if bitmap else (32 if slices == 1 else 96) * 1024 // 8 - 1)   # LDS plan of the kernels: 8-byte accumulators
            assert out0 % num_rows == row0 and out0 // num_rows < slices
            touched[row0: row0 + nrows] = True
            ys = np.zeros(nrows + 1, dtype=np.float64 if is_float else np.uint64)     # double sums of float products
            if sweep:
                _block_sweep(image, blk, x_words, ys, is_float)
            elif owner or (not bitmap and not delta):
                (_block_owner if owner else _block_pairs)(image, blk, units, x_words, ys, is_float, aux24)
            else:
                (_block_bitmap if bitmap else _block_delta)(image, blk, units, x_words, ys, is_float)
            if is_float:
                out[out0: out0 + nrows] = ys[:nrows].astype(np.float32).view(np.uint32)
            else:
                out[out0: out0 + nrows] = np.minimum(ys[:nrows], np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert done.all()
    if filtered:
        touched[:part_lo] = False
        touched[part_hi:] = False
    if slices > 1:   # combine_slices_kernel (a partition run: over the partition's rows only)
        parts = out.reshape(slices, num_rows)[:, touched]
        if is_float:
            acc = np.zeros(parts.shape[1], dtype=np.float32)
            for k in range(slices):
                acc = (acc + parts[k].view(np.float32)).astype(np.float32)
            y[touched] = acc.view(np.uint32)
        else:
            y[touched] = np.minimum(parts.astype(np.uint64).sum(axis=0), np.uint64(0xFFFFFFFF)).astype(np.uint32)
    elif filtered:
        y[part_lo:part_hi] = out[part_lo:part_hi]
    return y
