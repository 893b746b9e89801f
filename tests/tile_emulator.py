"""tests/tile_emulator.py — numpy walk over a row-block stream image, unit for unit what spmv_rowblock_kernel does.

Test infrastructure: lets the CPU-only suite check the load-time re-tiling (stream_tiles.cpp) and the
kernel's element semantics against the oracle without a GPU.  Never imported by the product.
"""
import numpy as np

WAVE, CONSUMERS, CHUNK_BYTES, SUB_TILE = 64, 14, 512, 8192


def _q_mul(a, b):
    wide = a.astype(np.uint64) * b.astype(np.uint64)
    r = (wide >> np.uint64(24)) + ((wide >> np.uint64(23)) & np.uint64(1))
    return np.minimum(r, np.uint64(0xFFFFFFFF))


def run(tiles, impl, x_words, num_rows, row_part_filter=-1, y_init=None):
    """Returns packed y words.  tiles: dict from hisparse_amd.device.build_tiles."""
    is_float = impl != 0
    image, blocks, units = tiles["image"], tiles["blocks"], tiles["units"]
    y = np.zeros(num_rows, dtype=np.uint32) if y_init is None else y_init.copy()
    slices = int(tiles.get("col_slices", 1))
    out = y if slices == 1 else np.zeros(slices * num_rows, dtype=np.uint32)   # per-slice partial results
    touched = np.zeros(num_rows, dtype=bool)
    done = np.zeros(len(blocks), dtype=bool)
    for g in range(tiles["num_workgroups"]):
        for q in range(tiles["wg_first"][g], tiles["wg_first"][g + 1]):
            b = int(tiles["block_order"][q])
            assert not done[b]
            done[b] = True
            blk = blocks[b]
            if row_part_filter >= 0 and int(blk["row_part"]) != row_part_filter:
                continue
            nrows, row0, out0 = int(blk["nrows"]), int(blk["row0"]), int(blk["out_offset"])
            gather = tiles["ring_buffers"] == 0            # gather mode: no x ring, only row accumulators in LDS
            assert nrows <= (16383 if gather else 4095 if slices == 1 else 12287)
            assert out0 % num_rows == row0 and out0 // num_rows < slices
            assert gather or 2 <= tiles["ring_buffers"] <= 4
            touched[row0: row0 + nrows] = True
            ys = np.zeros(nrows + 1, dtype=np.float32 if is_float else np.uint64)
            pos = [0] * CONSUMERS
            for u in range(int(blk["unit_begin"]), int(blk["unit_end"])):
                unit = units[u]
                col0, ncols = int(unit["col0"]), int(unit["ncols"])
                assert ncols % 8 == 0 and 0 < ncols <= SUB_TILE
                xt = x_words[col0: col0 + ncols]
                for w in range(CONSUMERS):
                    end = int(unit["end_step"][w])
                    base = int(blk["wave_offset"][w])
                    for s in range(pos[w], end):
                        chunk = image[base + s * CHUNK_BYTES * CONSUMERS: base + s * CHUNK_BYTES * CONSUMERS + CHUNK_BYTES].view(np.uint32).reshape(WAVE, 2)
                        val, cr = chunk[:, 0], chunk[:, 1]
                        col, row = (cr & 0xFFFF).astype(np.int64), (cr >> 16).astype(np.int64)
                        assert (col < ncols).all() and (row <= nrows).all()
                        xv = xt[col]
                        if is_float:
                            np.add.at(ys, row, (val.view(np.float32) * xv.view(np.float32)).astype(np.float32))
                        else:
                            np.add.at(ys, row, _q_mul(val, xv))
                    pos[w] = end
            if is_float:
                out[out0: out0 + nrows] = ys[:nrows].view(np.uint32)
            else:
                out[out0: out0 + nrows] = np.minimum(ys[:nrows], np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert done.all()
    if slices > 1:   # combine_slices_kernel
        parts = out.reshape(slices, num_rows)[:, touched]
        if is_float:
            acc = np.zeros(parts.shape[1], dtype=np.float32)
            for k in range(slices):
                acc = (acc + parts[k].view(np.float32)).astype(np.float32)
            y[touched] = acc.view(np.uint32)
        else:
            y[touched] = np.minimum(parts.astype(np.uint64).sum(axis=0), np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return y
