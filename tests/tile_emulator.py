"""tests/tile_emulator.py — numpy walk over a stream-tiles image, step for step what spmv_stream_kernel does.

Test infrastructure: lets the CPU-only suite check the load-time re-tiling (stream_tiles.cpp) and the
kernel's element semantics against the oracle without a GPU.  Never imported by the product.
"""
import numpy as np

WAVE, WAVES, GROUP_BYTES, HEADER = 64, 16, 64 * 4 * 6, 256
SPECIAL = 0xFFFF


def chunk_bytes(steps):
    return HEADER + steps * 8 + (steps // 4) * GROUP_BYTES


def _q_mul(a, b):
    wide = a.astype(np.uint64) * b.astype(np.uint64)
    r = (wide >> np.uint64(24)) + ((wide >> np.uint64(23)) & np.uint64(1))
    return np.minimum(r, np.uint64(0xFFFFFFFF))


def run(tiles, impl, x_words, num_rows, num_cols, tile_cols, row_part_filter=-1, y_prev=None):
    """Returns packed y words.  tiles: dict from hisparse_amd.device.build_tiles."""
    is_float = impl != 0
    image, pieces, stride = tiles["image"], tiles["pieces"], tiles["row_stride"]
    slack = 2048
    acc = np.zeros(num_rows + slack, dtype=np.float32 if is_float else np.uint64)
    lanes = np.arange(WAVE, dtype=np.uint64)
    for p in pieces:
        if row_part_filter >= 0 and int(p["row_part"]) != row_part_filter:
            continue
        steps, t = int(p["steps"]), int(p["col_tile"])
        assert steps % 8 == 0
        cols_here = min(tile_cols, num_cols - t * tile_cols)
        xt = x_words[t * tile_cols: t * tile_cols + cols_here]
        for w in range(WAVES):
            base = int(p["offset"]) + w * chunk_bytes(steps)
            ch = image[base: base + chunk_bytes(steps)]
            row = ch[:HEADER].view(np.uint32).astype(np.int64).copy()
            flags = ch[HEADER: HEADER + steps * 8].view(np.uint64)
            groups = ch[HEADER + steps * 8:]
            s = np.zeros(WAVE, dtype=np.float32 if is_float else np.uint64)
            for g in range(steps // 4):
                gb = groups[g * GROUP_BYTES: (g + 1) * GROUP_BYTES]
                cols = gb[:512].view(np.uint16).reshape(WAVE, 4)
                vals = gb[512:].view(np.uint32).reshape(WAVE, 4)
                for i in range(4):
                    c = cols[:, i].astype(np.int64)
                    v = vals[:, i]
                    special = c == SPECIAL
                    xv = xt[np.minimum(c, cols_here - 1)]
                    if is_float:
                        prod = (v.view(np.float32) * xv.view(np.float32)).astype(np.float32)
                        s = (s + np.where(special, np.float32(0), prod)).astype(np.float32)
                    else:
                        s = s + _q_mul(np.where(special, 0, v).astype(np.uint32), xv)
                    flagged = ((flags[g * 4 + i] >> lanes) & np.uint64(1)).astype(bool)
                    fl = np.nonzero(flagged & (s != 0))[0]
                    np.add.at(acc, row[fl], s[fl])
                    s = np.where(flagged, 0, s).astype(s.dtype)
                    row = np.where(flagged, np.where(special, v.astype(np.int64), row + stride), row)
            nz = np.nonzero(s != 0)[0]
            np.add.at(acc, row[nz], s[nz])
    if is_float:
        y = acc[:num_rows].view(np.uint32).copy()
    else:
        y = np.minimum(acc[:num_rows], np.uint64(0xFFFFFFFF)).astype(np.uint32)
    if row_part_filter >= 0 and y_prev is not None:
        return y, acc
    return y
