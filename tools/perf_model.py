"""Where the kernel time goes, per configuration: an additive model of spmv_rowblock_kernel re-targeted from the idea of the
reference's performance_model.cpp:431-441 (format efficiency beta, vector-tile and write-back terms) to MI355X, printed
next to the measured kernel time.

  python tools/perf_model.py [config ...]          (needs a GPU for the "measured" column; the model itself does not)
  python tools/perf_model.py --sweep <config>      tile-size sweep: the model over (column slices x rows per block), the counterpart
                                                   of the reference's v/o sweep (performance_model/design_space_exp.cpp:515-540);
                                                   with a GPU every point is also measured

Constants are measurements of this repository (tools/*.hip micro-benchmarks and HISPARSE_ABLATE builds on ogbl-ppa,
ogbn-products, mouse_gene), not fits per matrix.
"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

CUS = 256
STREAM_B_PER_US = {"pairs": 6.60e6, "delta": 6.44e6, "owner": 5.86e6, "bitmap": 5.5e6}   # per-wavefront streams (record_stream_bench.hip;
                                                                                         # owner / bitmap: ablation builds and the bitmap timeline)
BITMAP_FRONT_US, BITMAP_TAIL_US = 4.0, 3.3   # spmv_bitmap_kernel (tools/bitmap_timeline.py): dispatch ramp + descriptor -> masks -> first values;
                                             # wavefronts finishing apart + row sums + barrier + store
STARTUP_US = 3.0            # launch -> block header -> first records landed
PROLOGUE_US = 0.5           # per block: zero the accumulators, first x sub-tile, barrier (mostly behind the primed stream ring)
STORE_B_PER_US = 2.4e6      # result store burst when all workgroups finish together (9.2 MB in ~3.8 us) ...
STORE_LATENCY_US = 1.2      # ... plus the write latency at the end of a block
REFILL_LAND_US = 0.8        # one x sub-tile refill (LDS-DMA) lands
REFILL_B_PER_US_CU = 120e3  # L2 -> LDS through one CU
CU_STREAM_B_PER_US = 25e3   # one CU's share of the stream
BARRIER_US = 0.05           # per (block, sub-tile) unit
QUANTISATION_EXPOSED = 0.3  # share of the barrier-synchronisation slack that is not absorbed by the other wavefronts' bandwidth
REFILL_VOLUME_EXPOSED = 0.5 # share of the x refill volume that does not hide behind the stream
LDS_PS_PER_ELEMENT = {0: 11.5, 1: 18.0, 2: 18.0}   # exposed LDS gather + atomic cost per element and CU (u64 / f64)


def model(name, cp=None, impl=None):
    if cp is None:
        cfg, csr = datasets.load(name)
        impl = host.impl_id(cfg.impl)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    t = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, CUS)
    blocks, units = t["blocks"], t["units"]
    fmt, slices, ring = t["format"], t["col_slices"], t["ring_buffers"]
    fmt = fmt[:-2] if fmt.endswith("24") else fmt       # 7-byte forms of PAIRS / OWNER: same machinery, fewer stream bytes
    groups = t["num_workgroups"]
    if fmt == "bitmap":     # no units, no x ring: a front, the stream, a tail (per block of the busiest workgroup)
        per_wg = max(t["wg_first"][g + 1] - t["wg_first"][g] for g in range(groups))
        parts = {"stream": len(t["image"]) / STREAM_B_PER_US[fmt], "front (ramp, descriptor -> masks -> values)": BITMAP_FRONT_US,
                 "tail (finish spread, row sums, store)": BITMAP_TAIL_US * per_wg, "launch": 3.0}
        return cp, impl, t, parts
    # critical workgroup: steps serialised by the per-unit barrier (sum over units of the slowest wavefront)
    es = units["end_step"].astype(np.int64)
    step_bytes = 384 if fmt == "delta" else 448 if t["format"].endswith("24") else 512
    sync = np.zeros(groups)
    ideal = np.zeros(groups)
    nunits = np.zeros(groups)
    nblocks = np.zeros(groups)
    rows = np.zeros(groups)
    for g in range(groups):
        for b in t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]:
            blk = blocks[b]
            nblocks[g] += 1
            rows[g] += int(blk["nrows"])
            if blk["unit_end"] == blk["unit_begin"]:
                continue
            e = es[blk["unit_begin"]:blk["unit_end"]]
            d = np.diff(np.vstack([np.zeros((1, 14), dtype=np.int64), e]), axis=0)
            sync[g] += d.max(axis=1).sum()
            ideal[g] += d.sum() / 14.0
            nunits[g] += len(e)
    stream_us = len(t["image"]) / STREAM_B_PER_US[fmt]
    crit = int(np.argmax(sync))
    quant_us = QUANTISATION_EXPOSED * stream_us * (sync.max() / max(ideal.mean(), 1e-9) - 1.0)   # barrier-synchronised wavefronts + imbalance
    unit_stream_us = len(t["image"]) / max(len(units), 1) / CU_STREAM_B_PER_US
    refill_us = nunits[crit] * max(0.0, REFILL_LAND_US / max(ring - 1, 1) - unit_stream_us) + REFILL_VOLUME_EXPOSED * nunits[crit] * 32768 / REFILL_B_PER_US_CU
    store_us = nblocks[crit] * STORE_LATENCY_US + rows.sum() * 4 / STORE_B_PER_US / max(1.0, nblocks.mean())
    parts = {
        "stream": stream_us, "startup": STARTUP_US, "prologues": PROLOGUE_US * nblocks[crit], "result store": store_us,
        "unit barriers": BARRIER_US * nunits[crit], "step quantisation": quant_us, "x refill": refill_us,
        "LDS work": cp.nnz / CUS * (12.0 if fmt == "owner" else LDS_PS_PER_ELEMENT[impl]) * 1e-6,   # owner: gather + read-modify-write, 3.6 lanes/clk
    }
    return cp, impl, t, parts


def measure(cp, impl):
    eng = device.SpmvEngine(impl)
    eng.load_matrix(cp)
    eng.load_vector(host.pack_vector(impl, np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)))
    best = min(eng.time_runs(5, 30)[1] / 30 for _ in range(3)) * 1e3
    eng.close()
    return best


def sweep(name, measure_points=True, out=sys.stdout):
    """Model (and, with a GPU, measurement) over column slices x rows per block; returns {(slices, rows): (model_us, measured_us)}."""
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    grid = {}
    saved = {k: os.environ.get(k) for k in ("HISPARSE_COL_SLICES", "HISPARSE_MAX_ROWS")}
    try:
        for cs in (1, 2, 3, 4, 5, 6, 8):
            for rows in (128, 512, 2047, 4095, 8191, 12287, 16369, 24561):
                os.environ["HISPARSE_COL_SLICES"], os.environ["HISPARSE_MAX_ROWS"] = str(cs), str(rows)
                try:
                    _, _, t, parts = model(name, cp, impl)
                except device.DeviceError:
                    continue                      # e.g. more slices than sub-tiles
                if t["col_slices"] != cs or (cs, int(t["max_block_rows"])) in grid:
                    continue                      # the cap did not bind: same tiling as a point already listed
                measured = float("nan")
                if measure_points:
                    try:
                        measured = measure(cp, impl)
                    except device.DeviceError:
                        measure_points = False
                grid[(cs, int(t["max_block_rows"]))] = (sum(parts.values()), measured)
                print(f"  slices {cs}  rows/block {int(t['max_block_rows']):6d}  ring {t['ring_buffers']}  {t['format']:6s} units {len(t['units']):6d}  "
                      f"model {sum(parts.values()):7.1f} us  measured {measured:7.1f} us", file=out)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if grid:
        best = min(grid, key=lambda k: grid[k][0])
        print(f"  model optimum: {best[0]} slice(s), {best[1]} rows per block ({grid[best][0]:.1f} us)", file=out)
    return grid


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--sweep":
        print(f"{sys.argv[2]}: tile-size sweep")
        sweep(sys.argv[2])
        sys.exit(0)
    names = sys.argv[1:] or ["ogbl_ppa", "mouse_gene", "transformer_50", "ogbn_products"]
    for name in names:
        cp, impl, t, parts = model(name)
        total = sum(parts.values())
        try:
            measured = measure(cp, impl)
        except device.DeviceError:
            measured = float("nan")
        beta = 8.0 * cp.nnz / len(t["image"])
        print(f"{name}: {t['format']}, {t['col_slices']} slice(s), ring {t['ring_buffers']}, {len(t['blocks'])} blocks, {len(t['units'])} units, "
              f"beta = 8*nnz / streamed bytes = {beta:.2f}")
        print(f"  model {total:7.1f} us | measured {measured:7.1f} us | 8*nnz at 8 TB/s = {8.0 * cp.nnz / 8e6:6.1f} us")
        print("  " + ", ".join(f"{k} {v:.1f}" for k, v in parts.items()))
