"""hisparse_amd.sharding — one matrix across the GPUs of a node, by row blocks.

The reference is a single-device design (no communication backend of any kind); what it
does have is row partitioning with no cross-partition state (sw/data_formatter.h:494,500-511,
sw/benchmark.cpp:318-338).  That is the axis used here: every GPU gets a contiguous slab of rows whose
boundaries are multiples of the row padding granule (128 * interleave), formats its slab with the
ordinary host pipeline (so a slab is a complete CPSR matrix of its own) and keeps a full copy of x.
The only exchange is the all-gather of the y slabs (RCCL over xGMI via torch.distributed's "nccl"
backend; "gloo" on CPU in the tests).  No collective sits inside the SpMV itself.
"""
import numpy as np


def split_rows_by_nnz(indptr, parts, granule):
    """Row boundaries [b_0=0, ..., b_parts=rows] balancing non-zeros, interior ones multiples of `granule`.

    Every slab gets at least one granule of rows when the matrix has that many (a slab without rows cannot be loaded:
    hs_load_matrix refuses an empty matrix), so a few very heavy rows cannot starve the ranks behind them; with fewer
    than parts * granule rows the trailing slabs are empty (lo == hi) and the caller skips them (`nonempty`)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = indptr.size - 1
    nnz = int(indptr[-1])
    granules = (rows + granule - 1) // granule          # the last granule may be short
    bounds = [0]
    for p in range(1, parts):
        target = nnz * p / parts
        r = int(np.searchsorted(indptr, target, side="left"))
        g = (2 * r + granule) // (2 * granule)             # nearest granule boundary, halves up (same rule as row_sharding.h)
        lo_g = bounds[-1] // granule + 1                 # at least one granule for slab p-1 ...
        hi_g = granules - (parts - p)                    # ... and one for every slab still to come
        g = min(max(g, lo_g), hi_g) if hi_g >= lo_g else min(bounds[-1] // granule + 1, granules)
        bounds.append(min(g * granule, rows))
    bounds.append(rows)
    return bounds


def nonempty(bounds):
    """[(slab index, lo, hi)] of the slabs that hold rows."""
    return [(i, bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]


def slab_arrays(indptr, indices, data, lo, hi):
    """CSR arrays of rows [lo, hi)."""
    indptr = np.asarray(indptr)
    a, b = int(indptr[lo]), int(indptr[hi])
    return (indptr[lo:hi + 1].astype(np.int64) - a).astype(np.uint32), np.asarray(indices[a:b]), np.asarray(data[a:b])


def padded_rows(rows, granule):
    return (rows + granule - 1) // granule * granule


def gather_layout(slab_rows, granule):
    """(padded slab length used as the all-gather chunk, list of (offset_in_gathered, true_rows)) for unequal slabs."""
    chunk = max(padded_rows(r, granule) for r in slab_rows) if slab_rows else 0
    return chunk, [(i * chunk, r) for i, r in enumerate(slab_rows)]


def assemble(gathered, layout):
    """Concatenate the true rows of every rank's chunk from an all-gathered flat array."""
    chunk, spans = layout
    return np.concatenate([np.asarray(gathered[o:o + r]) for o, r in spans]) if spans else np.zeros(0, dtype=np.uint32)
