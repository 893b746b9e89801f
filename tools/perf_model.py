"""Where the kernel time goes, per configuration: an additive model of spmv_rowblock_kernel re-targeted from the idea of the
reference's performance_model.cpp:431-441 (format efficiency beta, vector-tile and write-back terms) to MI355X, printed
next to the measured kernel time.

  python tools/perf_model.py [config ...]          (needs a GPU for the "measured" column; the model itself does not)
  python tools/perf_model.py --sweep <config>      tile-size sweep: the model over (column slices x rows per block), the counterpart
                                                   of the reference's v/o sweep (performance_model/design_space_exp.cpp:515-540);
                                                   with a GPU every point is also measured

Constants are measurements of this repository (round 3: the timeline builds HISPARSE_ABLATE=512 / 64, the OWNER profile, ablation
builds and tools/*.hip micro-benchmarks), not fits per matrix.  tests/test_perf_model.py keeps the model within
15 % of the measured kernel on the five configurations (-m gpu).
"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

CUS = 256
# ---- constants: measurements of round 3 (timeline builds HISPARSE_ABLATE=512 / 64, ablation builds, tools/*.hip) ---------------------
LAUNCH_US = 3.0             # HIP-event start -> first wavefront's first instruction, + last store -> event end (events vs timeline stamps)
RAMP_US = 4.3               # a 1024-thread workgroup's wavefronts are started 2-4 us apart; zero + first x sub-tile + barrier ride on it
NEXT_BLOCK_US = 3.5         # the same for a workgroup's further blocks (descriptor, zeroing, first sub-tile; no launch ramp)
STORE_US = 1.0              # result store of a block (+ the write latency at its end)
SPREAD = 0.05               # workgroups finish up to 5 % of the main loop apart (dynamic: memory-system fairness, not step counts)
SWEEP_WAVES = 8              # stream_tiles.h: kSweepWaves
SWEEP_NS_PER_ELEMENT, SWEEP_NS_PER_LINE = 0.33, 1.36      # SWEEP (tools/gather_bench.hip): 8 bytes at ~24 GB/s per CU; a gathered line of x every ~3.3 clocks
CU_STREAM_B_PER_US = {"pairs": 25.8e3, "delta": 25.8e3, "owner": 26.5e3}   # one CU's 14 consumer rings while nothing else binds: 6.6-6.8 TB/s / 256
# a (row block, x sub-tile) UNIT costs at least this much, however little it holds: end-of-unit flush + barrier + the loaders' refill issue
# (8 LDS-DMA instructions, ~1000 clocks, tools/owner_profile.py) + its landing.  OWNER / OWNER24: pokec (<= 3 steps per wavefront and unit)
# 1.38 us, ogbn-products (<= 5) 1.5 us; the atomic formats' units have no flush and a ring of 3-4: 0.45 us
UNIT_FLOOR_US = {"owner": lambda steps: 1.2 + 0.06 * steps, "pairs": lambda steps: 0.45, "delta": lambda steps: 0.45}
LDS_EXPOSED_PS = {"pairs": {0: 6.0, 1: 9.0, 2: 9.0}, "delta": {0: 9.0, 1: 14.0, 2: 14.0}, "owner": {0: 0.0, 1: 0.0, 2: 0.0}}   # per element and CU, beside a saturated stream
BITMAP_FRONT_US, BITMAP_TAIL_US = 4.0, 2.0   # spmv_bitmap_kernel (tools/bitmap_timeline.py): ramp + descriptor -> masks -> first values; row sums + barrier + store
BITMAP_BATCH_US = 0.62      # per batch of 8 steps and wavefront, averaged over the workgroup's 16 (their shares are weighted by issue priority so that they finish together: the MEAN counts; transformer-50 and -80 take the same time)
BITMAP_LAUNCH_US = 2.5        # event pair around one launch (the measured column); back-to-back steps overlap most of it


def model(name, cp=None, impl=None):
    """(ChannelPackets, impl, tiles, {term: microseconds}) -- the terms add up to the modelled duration of the SpMV kernel as a HIP-event
    pair around the launch sees it."""
    if cp is None:
        cfg, csr = datasets.load(name)
        impl = host.impl_id(cfg.impl)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    t = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, CUS)
    blocks, units = t["blocks"], t["units"]
    family = {"pairs24": "pairs", "owner24": "owner"}.get(t["format"], t["format"])
    groups = t["num_workgroups"]
    if family == "bitmap":     # no units, no x ring: a front, the runs, a tail (per block of the busiest workgroup)
        per_wg = max(t["wg_first"][g + 1] - t["wg_first"][g] for g in range(groups))
        segs = units.view(np.uint8).reshape(len(blocks), 16, 5 * 64)[:, :, :16].copy().view(np.uint32).reshape(len(blocks), 16, 4)
        steps = (segs[:, :, 1] - segs[:, :, 0]).astype(np.int64) * (segs[:, :, 3] - segs[:, :, 2]).astype(np.int64)     # rows x groups of a wavefront's run
        batches = steps.sum(axis=1) / 16.0 / 8.0                              # mean over the 16 wavefronts of every block
        run_us = float(batches.max()) * BITMAP_BATCH_US * per_wg
        stream_us = len(t["image"]) / 6.6e6                                   # what the bytes alone would need
        parts = {"launch": BITMAP_LAUNCH_US, "front (ramp, descriptor -> masks -> values)": BITMAP_FRONT_US,
                 "stream": max(run_us, stream_us), "tail (finish spread, row sums, store)": BITMAP_TAIL_US * per_wg}
        return cp, impl, t, parts
    if family == "sweep":      # no units: every element at one CU's stream rate + every 128-byte line of the block's slice of x gathered once
        per = np.zeros(groups); nblocks = np.zeros(groups)
        for g in range(groups):
            for b in t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]:
                blk = blocks[b]
                nblocks[g] += 1
                per[g] += int(blk["total_steps"][0]) * SWEEP_WAVES * 64 * SWEEP_NS_PER_ELEMENT * 1e-3 + int(blk["first_ncols"]) / 32.0 * SWEEP_NS_PER_LINE * 1e-3
        crit = int(np.argmax(per + nblocks * (NEXT_BLOCK_US + STORE_US)))
        parts = {"launch": LAUNCH_US, "ramp + prologue": RAMP_US, "further blocks' prologues": NEXT_BLOCK_US * max(0.0, nblocks[crit] - 1),
                 "stream": per[crit], "result stores": STORE_US * nblocks[crit], "finish spread": SPREAD * per[crit]}
        return cp, impl, t, parts
    # per workgroup: its blocks one after the other; a block's main loop is the slower of (its stream at one CU's rate) and (its units at
    # the per-unit floor, each unit as long as its slowest wavefront's steps)
    packed = t["format"] == "owner24"
    es = (units["end_step"] & 0xFFFF if packed else units["end_step"]).astype(np.int64)
    step_bytes = {"pairs": 512, "pairs24": 448, "delta": 768, "owner": 512, "owner24": 448}[t["format"]]      # per wavefront step (DELTA: a record of two slots per lane)
    rate, floor = CU_STREAM_B_PER_US[family], UNIT_FLOOR_US[family]
    main = np.zeros(groups); stream = np.zeros(groups); floors = np.zeros(groups); nblocks = np.zeros(groups); elems = np.zeros(groups)
    for g in range(groups):
        for b in t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]:
            blk = blocks[b]
            nblocks[g] += 1
            if blk["unit_end"] == blk["unit_begin"]:
                continue
            e = es[blk["unit_begin"]:blk["unit_end"]]
            d = np.diff(np.vstack([np.zeros((1, 14), dtype=np.int64), e]), axis=0)
            s_us = d.sum() * step_bytes / rate
            f_us = sum(floor(int(m)) for m in d.max(axis=1))
            stream[g] += s_us
            floors[g] += f_us
            main[g] += max(s_us, f_us)
            elems[g] += d.sum() * 64
    crit = int(np.argmax(main + nblocks * (NEXT_BLOCK_US + STORE_US)))
    lds_us = elems[crit] * LDS_EXPOSED_PS[family][impl] * 1e-6
    parts = {
        "launch": LAUNCH_US, "ramp + prologue": RAMP_US, "further blocks' prologues": NEXT_BLOCK_US * max(0.0, nblocks[crit] - 1),
        "stream": min(stream[crit], main[crit]), "unit floor beyond the stream (flush, barrier, refill)": max(0.0, main[crit] - stream[crit]),
        "LDS work beside the stream": lds_us, "result stores": STORE_US * nblocks[crit], "finish spread": SPREAD * main[crit],
    }
    return cp, impl, t, parts


def measure(cp, impl):
    eng = device.SpmvEngine(impl)
    eng.load_matrix(cp)
    eng.load_vector(host.pack_vector(impl, np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)))
    for _ in range(100):      # clocks up
        eng.run()
    best = min(eng.time_runs(5, 30)[1] / 30 for _ in range(3)) * 1e3
    eng.close()
    return best


def sweep(name, measure_points=True, out=sys.stdout, rows_options=(128, 512, 2047, 4095, 8191, 12287, 16369, 24561)):
    """Model (and, with a GPU, measurement) over column slices x rows per block; returns {(slices, rows): (model_us, measured_us)}."""
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    grid = {}
    saved = {k: os.environ.get(k) for k in ("HISPARSE_COL_SLICES", "HISPARSE_MAX_ROWS")}
    try:
        for cs in (1, 2, 3, 4, 5, 6, 8):
            for rows in rows_options:
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
