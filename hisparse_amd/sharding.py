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
    """Row boundaries [b_0=0, ..., b_parts=rows] balancing non-zeros, interior ones multiples of `granule`."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = indptr.size - 1
    nnz = int(indptr[-1])
    bounds = [0]
    for p in range(1, parts):
        target = nnz * p / parts
        r = int(np.searchsorted(indptr, target, side="left"))
        r = int(round(r / granule)) * granule
        r = min(max(r, bounds[-1]), rows)
        bounds.append(r)
    bounds.append(rows)
    return bounds


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
