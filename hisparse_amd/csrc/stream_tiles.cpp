// stream_tiles.cpp — CPSR image -> row-block element streams (see stream_tiles.h for the format and the why).
//
// The decode follows the reference's loader in MEANING (header layout, per-lane lengths, marker = row
// advance, interleaved virtual channels):
//   spmv/libfpga/spmv_cluster.h:41-98        fixed point, INTERLEAVE_FACTOR 1
//   spmv-fp/libfpga/spmv_cluster.h:46-117    float, INTERLEAVE_FACTOR 1 or 8
// with the row <-> (channel, lane, round) mapping of sw/data_formatter.h:410,432 and the result drain
// order of spmv/spmv_result_drain.cpp:36,104-113 (net effect: natural row order in y).
#include "stream_tiles.h"
#include "tiles_common.h"
#include "gpu_tiles.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <thread>

namespace hisparse {
namespace dev {

namespace {

using namespace detail;

thread_local bool g_no_owner = false;      // second attempt of build_stream_tiles after a fixed-point OWNER24 image turned out not to fit
const char* const kOwnerDoesNotFit = "owner24: a share or a step count exceeds the record format";

bool build_stream_tiles_once(const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                             const Geometry& geom, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                             uint32_t num_col_partitions, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                             void* gpu_stream, bool use_gpu, uint64_t image_slack, const CsrView* csr);

}  // namespace

bool build_stream_tiles(const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                        const Geometry& geom, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                        uint32_t num_col_partitions, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                        void* gpu_stream, bool use_gpu, uint64_t image_slack, const CsrView* csr) {
    g_no_owner = false;
    bool ok = build_stream_tiles_once(channel, n_packets, geom, num_rows, num_cols, num_row_partitions, num_col_partitions, max_workgroups, out, error, gpu_stream,
                                      use_gpu, image_slack, csr);
    if (!ok && error == kOwnerDoesNotFit) {
        g_no_owner = true;
        error.clear();
        ok = build_stream_tiles_once(channel, n_packets, geom, num_rows, num_cols, num_row_partitions, num_col_partitions, max_workgroups, out, error, gpu_stream,
                                     use_gpu, image_slack, csr);
        g_no_owner = false;
    }
    return ok;
}

namespace {

// ---- tile census (round 6) -------------------------------------------------------------------------------------------------------------
// Until round 5 the plan was made from the rows' non-zero counts alone, i.e. as if every row range met every x sub-tile with the same number
// of elements.  True enough for the scrambled power-law graphs and the Bernoulli layers the constants were measured on -- and wrong by
// 2-6 x on anything with STRUCTURE (tools/planner_check.py, profiles/r06_planner_check_before.txt): a banded or block-diagonal matrix keeps a
// row range's elements in two or three sub-tiles, so a plan of 51 row ranges x 5 column slices has 102 blocks that hold anything, on 102 of
// 256 workgroups (banded 400 K: 56.9 us in the planner's 5 slices, 21.3 us in one).  The census is what the model lacked: non-zeros per
// (fine row range, sub-tile) for `fine` ranges of equal non-zero count -- one more counting pass (the pass-1 kernel / walk with another row
// map) -- from which every candidate plan's REAL units (the non-empty ones), block loads (sub-tiles dealt to slices the way the builder
// deals them) and workgroup loads (heaviest block first, the way assign_workgroups balances) follow.
struct TileCensus {
    uint32_t fine = 0, tiles = 0;
    std::vector<uint32_t> cnt;          // [fine][tiles]
    double populated = 1.0;             // fraction of the (fine range, sub-tile) cells that hold anything
    struct Eval { double nonempty_units, max_wg_load; };
    // a plan of `plan_ranges` row ranges (equal non-zero count, in row order) x `cs` column slices on G workgroups
    Eval evaluate(uint64_t plan_ranges, uint32_t cs, uint32_t G, uint64_t nnz) const {
        plan_ranges = std::max<uint64_t>(1, plan_ranges);
        if (cnt.empty() || !fine) {      // no census (matrix without non-zeros): the uniform picture
            const double blocks = double(plan_ranges) * cs, per_wg = std::ceil(blocks / G);
            return {double(plan_ranges) * tiles, double(nnz) / blocks * per_wg};
        }
        // more plan ranges than census rows: every census row stands for `split` plan ranges of 1 / split of its non-zeros
        const uint64_t split = plan_ranges > fine ? (plan_ranges + fine - 1) / fine : 1;
        const uint64_t groups = plan_ranges > fine ? fine : plan_ranges;
        std::vector<double> t(tiles), block_load;
        std::vector<uint32_t> order(tiles);
        block_load.reserve(size_t(groups * split) * cs);
        double nonempty = 0.0;
        std::vector<double> slice_load(cs);
        std::vector<uint32_t> slice_tiles(cs);
        for (uint64_t j = 0; j < groups; ++j) {
            const uint64_t lo = j * fine / groups, hi = (j + 1) * fine / groups;
            std::fill(t.begin(), t.end(), 0.0);
            for (uint64_t f = lo; f < hi; ++f)
                for (uint32_t k = 0; k < tiles; ++k) t[k] += cnt[f * tiles + k];
            uint32_t live = 0;
            for (uint32_t k = 0; k < tiles; ++k) live += t[k] > 0.0;
            nonempty += double(live) * double(split);
            std::fill(slice_load.begin(), slice_load.end(), 0.0);
            if (cs == 1) {
                for (uint32_t k = 0; k < tiles; ++k) slice_load[0] += t[k];
            } else {      // the builder's dealing: heaviest sub-tile first, each to the lightest slice so far (ties: the one with fewer sub-tiles)
                std::iota(order.begin(), order.end(), 0u);
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return t[a] > t[b]; });
                std::fill(slice_tiles.begin(), slice_tiles.end(), 0u);
                for (uint32_t k : order) {
                    uint32_t best = 0;
                    for (uint32_t c = 1; c < cs; ++c)
                        if (slice_load[c] < slice_load[best] || (slice_load[c] == slice_load[best] && slice_tiles[c] < slice_tiles[best])) best = c;
                    slice_load[best] += t[k];
                    slice_tiles[best] += 1;
                }
            }
            for (uint64_t rep = 0; rep < split; ++rep)
                for (uint32_t c = 0; c < cs; ++c) block_load.push_back(slice_load[c] / double(split));
        }
        // heaviest block first, each to the workgroup with the least work so far (assign_workgroups)
        std::sort(block_load.begin(), block_load.end(), std::greater<double>());
        std::vector<double> wg(std::max<uint32_t>(1, G), 0.0);
        std::make_heap(wg.begin(), wg.end(), std::greater<double>());
        double worst = 0.0;
        for (double b : block_load) {
            std::pop_heap(wg.begin(), wg.end(), std::greater<double>());
            wg.back() += b;
            worst = std::max(worst, wg.back());
            std::push_heap(wg.begin(), wg.end(), std::greater<double>());
        }
        return {std::max(1.0, nonempty), std::max(worst, double(nnz) / G)};
    }
};

bool build_stream_tiles_once(const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                        const Geometry& geom, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                        uint32_t num_col_partitions, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                             void* gpu_stream, bool use_gpu, uint64_t image_slack, const CsrView* csr) {
    Layout L;
    L.g = &geom;
    L.num_rows = num_rows;
    L.num_cols = num_cols;
    L.row_parts = num_row_partitions;
    L.col_parts = num_col_partitions;
    L.F = geom.interleave;
    L.sub_width = uint32_t(std::min<uint64_t>(kSubTileCols, geom.logical_vb));
    L.subs_per_cp = uint32_t((geom.logical_vb + L.sub_width - 1) / L.sub_width);
    if (const char* cross = env_switch("HISPARSE_CROSS_PARTITIONS")) L.cross_parts = std::atoi(cross) != 0;
    const uint32_t F = L.F, CP = num_col_partitions, RP = num_row_partitions, S = L.subs_per_cp;
    const bool is_float = geom.impl != IMPL_FIXED;
    const uint64_t header_pkts = uint64_t(RP) * CP * (1 + F);
    if (csr && !use_gpu) { error = "the CSR source needs the GPU re-tile"; return false; }
    for (uint32_t c = 0; !csr && c < NUM_HBM_CHANNELS; ++c) {
        if (!channel[c] && n_packets[c]) { error = "null channel buffer"; return false; }
        if (n_packets[c] < header_pkts) { error = "channel " + std::to_string(c) + " is shorter than its partition headers"; return false; }
    }
    out = StreamTiles();
    auto chan = [&](uint32_t pc) { return static_cast<const MatPkt*>(channel[pc]); };

    PhaseTimer timer;
    // ---- pass 0: non-zeros per row (rows of different physical channels are disjoint) ------------
    std::vector<uint32_t> row_nnz(num_rows, 0);
    std::unique_ptr<GpuTiler> gpu;       // the per-non-zero passes on the device (gpu_tiles.h) instead of the host walks below
    if (use_gpu) {
        if (csr) gpu.reset(new GpuTiler(L, *csr, static_cast<hipStream_t>(gpu_stream)));
        else gpu.reset(new GpuTiler(L, channel, n_packets, static_cast<hipStream_t>(gpu_stream)));
        if (!gpu->count_rows(row_nnz, out.nnz)) { error = gpu->error(); return false; }
    } else {
    std::vector<WalkResult> res0(size_t(RP) * NUM_HBM_CHANNELS);
    parallel_for(res0.size(), [&](size_t w) {
        const uint32_t rp = uint32_t(w / NUM_HBM_CHANNELS), pc = uint32_t(w % NUM_HBM_CHANNELS);
        for (uint32_t cp = 0; cp < CP && res0[w].ok; ++cp) {
            WalkResult r = walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp,
                                                  [&](uint32_t row, uint32_t, uint32_t) { row_nnz[row]++; });
            if (!r.ok) res0[w] = r; else res0[w].nnz += r.nnz;
        }
    });
    for (const auto& r : res0) {
        if (!r.ok) { error = r.error; return false; }
        out.nnz += r.nnz;
    }
    }

    timer.lap("pass 0 (row counts)");
    // ---- tile census (TileCensus above): non-zeros per (fine row range of equal non-zero count, x sub-tile) -----------------------------
    TileCensus census;
    census.tiles = CP * S;
    {
        const char* off = env_switch("HISPARSE_PLAN_CENSUS");      // 0: plan as rounds 1-5 did, from the row counts alone (A/B, tools/planner_check.py)
        if (out.nnz && !(off && std::atoi(off) == 0)) {
            census.fine = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>({512, (uint64_t(4) << 20) / std::max<uint32_t>(1, census.tiles), num_rows})));
            std::vector<uint32_t> fine_of_row(num_rows);
            uint64_t seen = 0;
            for (uint32_t r = 0; r < num_rows; ++r) {
                fine_of_row[r] = uint32_t(std::min<uint64_t>(census.fine - 1, seen * census.fine / out.nnz));
                seen += row_nnz[r];
            }
            if (gpu) {
                if (!gpu->count_tiles(fine_of_row, census.fine, census.cnt)) { error = gpu->error(); return false; }
            } else {
                const size_t cells = size_t(census.fine) * census.tiles;
                std::unique_ptr<std::atomic<uint32_t>[]> cell(new std::atomic<uint32_t>[cells]);
                for (size_t i = 0; i < cells; ++i) cell[i].store(0, std::memory_order_relaxed);
                parallel_for(size_t(RP) * NUM_HBM_CHANNELS, [&](size_t w) {
                    const uint32_t rp = uint32_t(w / NUM_HBM_CHANNELS), pc = uint32_t(w % NUM_HBM_CHANNELS);
                    for (uint32_t cp = 0; cp < CP; ++cp)
                        walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp, [&](uint32_t row, uint32_t col, uint32_t) {
                            cell[size_t(fine_of_row[row]) * census.tiles + size_t(cp) * S + col / L.sub_width].fetch_add(1, std::memory_order_relaxed);
                        });
                });
                census.cnt.resize(cells);
                for (size_t i = 0; i < cells; ++i) census.cnt[i] = cell[i].load(std::memory_order_relaxed);
            }
            // (over the sub-tiles that exist: the last column partition's table ends where the matrix does)
            size_t live = 0, existing = 0;
            for (uint32_t cp = 0; cp < CP; ++cp)
                for (uint32_t sub = 0; sub < S; ++sub) existing += uint64_t(sub) * L.sub_width < L.cols_in_part(cp);
            existing *= census.fine;
            for (uint32_t c : census.cnt) live += c != 0;
            census.populated = existing ? std::min(1.0, std::max(1.0 / double(existing), double(live) / double(existing))) : 1.0;
            if (env_switch("HISPARSE_PLAN_DEBUG"))
                std::fprintf(stderr, "census: %u fine row ranges x %u sub-tiles, %.1f %% of the cells hold anything\n", census.fine, census.tiles, census.populated * 100.0);
        }
    }
    timer.lap("tile census");
    // ---- hub rows (round 6) ---------------------------------------------------------------------------------------------------------------
    // A row that holds a large part of its row block puts most lanes of a step on ONE LDS accumulator: same-address ds_add serialises, and a
    // matrix whose 50 hub rows hold 64 % of the non-zeros (tools/planner_check.py: hubs_500k) ran 89 us as the PAIRS image the mean gap
    // picked, 94 us as a DELTA image -- and 44.9 us as a DELTA image whose lanes sum their runs in registers (kBlockDenseRows: one LDS add per
    // lane and row change), which until now only blocks of uniformly long rows got.  So: (a) a DELTA block whose heaviest row holds an
    // eighth of it is flagged for per-lane sums whatever its mean gap (below, "enumerate blocks"); (b) where rows heavy enough to fill a
    // quarter of a workgroup's share hold >= 30 % of the matrix, the DELTA image is kept even if PAIRS would be smaller.  Forcing per-lane
    // sums on every block costs an ordinary graph 20 % (rmat19: 32.0 -> 39.1 us), hence per block.
    double hub_share = 0.0;
    {
        const uint64_t hub_min = std::max<uint64_t>(8192, out.nnz / (4ull * std::max<uint32_t>(1, max_workgroups)));
        uint64_t in_hubs = 0;
        for (uint32_t r = 0; r < num_rows; ++r) in_hubs += row_nnz[r] >= hub_min ? row_nnz[r] : 0u;
        hub_share = out.nnz ? double(in_hubs) / double(out.nnz) : 0.0;
    }
    bool keep_delta_for_hubs = false;
    bool prefer_sliced_delta = false, sliced_delta_possible = false;      // (decided in the BITMAP / LIGHT blocks below)
    double sliced_delta_us = 0.0;
    // ---- dense-row matrices (pruned-NN layers): BITMAP rows, their own builder and kernel (stream_tiles.h) --------------
    {
        // density of the rows that hold anything (padding rows and empty rows cost a mask per group and nothing else -- as long as
        // all masks together stay below a quarter of the 8 bytes per non-zero they replace)
        uint64_t live_rows = 0;
        for (uint32_t r = 0; r < num_rows; ++r) live_rows += row_nnz[r] != 0;
        const double density = live_rows ? double(out.nnz) / (double(live_rows) * double(num_cols)) : 0.0;
        const double mask_bytes = double(num_rows) * double((num_cols + kBitmapGroupCols - 1) / kBitmapGroupCols) * 8.0;
        bool bitmap = density >= kBitmapMinDensity && num_cols >= kBitmapMinCols && mask_bytes <= 2.0 * double(out.nnz);
        // Round 5: SMALL dense-row layers in FIXED point as a sliced DELTA plan.  With the combine pass carried into the next step's kernel
        // (hs_api.cpp) a plan of one column slice per x sub-tile is ONE launch without x refills and unit barriers, and its lanes sum their
        // rows in registers (kBlockDenseRows): measured on the 512 x 33 288 pruned-NN layers (profiles/r05_sliced_delta_vs_bitmap.txt, fixed
        // point, whole step; with the lane-major dealing of the runs, "after the dealing" there): 10 % dense 7.5 us against 8.6 (LIGHT), 20 % 9.0
        // against 11.9 (BITMAP), 30 % 11.0 against 12.0, 40 % 12.5 against 12.4, 5 % 6.9 against 5.9 (LIGHT) -- ~5.8 us + 1.0 us per million
        // non-zeros, where the BITMAP kernel pays for every 64-column group
        // whatever it holds (~5 us + 7.5 ns per step and CU) and the LIGHT kernel 3.1 us + 3.2 us per million.  The float modes keep their
        // plans: their BITMAP kernel is 2 us faster and their DELTA path 1 us slower, which leaves 0.4-0.5 us at 10 % and 20 % density and a
        // loss everywhere else.
        {
            const uint32_t live_tiles = (num_cols + kSubTileCols - 1) / kSubTileCols;
            const double scale = 256.0 / std::max<uint32_t>(1, max_workgroups);
            sliced_delta_us = 5.8 + double(out.nnz) * 1.0e-6 * scale;
            sliced_delta_possible = !is_float && live_tiles >= 1 && live_tiles <= kMaxColSlices && density >= 0.04 && num_cols >= kBitmapMinCols &&
                                    double(out.nnz) * 7.0 < double(kSlicedDeltaMaxImageBytes) && RP == 1 && out.nnz >= (1u << 20);      // (measured between 0.85 and 8.5 M non-zeros)
            const double bitmap_us = 5.0 + double(num_rows) * double((num_cols + kBitmapGroupCols - 1) / kBitmapGroupCols) / std::max<uint32_t>(1, max_workgroups) * 7.5e-3;
            if (bitmap && sliced_delta_possible && sliced_delta_us < 0.97 * bitmap_us && !env_switch("HISPARSE_STREAM_FORMAT")) {
                bitmap = false;
                prefer_sliced_delta = true;
            }
            if (env_switch("HISPARSE_PLAN_DEBUG"))
                std::fprintf(stderr, "format: dense rows (density %.3f): bitmap %.1f us, sliced delta %.1f us (%s) -> %s\n", density, bitmap_us, sliced_delta_us,
                             sliced_delta_possible ? "possible" : "not possible", prefer_sliced_delta ? "sliced delta" : bitmap ? "bitmap" : "element streams");
        }
        if (const char* force = env_switch("HISPARSE_STREAM_FORMAT")) {
            const std::string f(force);
            if (f == "bitmap") bitmap = true;
            else if (f == "pairs" || f == "delta") bitmap = false;
            else if (f == "owner" || f == "owner24" || f == "sweep") bitmap = false;
            else if (!f.empty()) { error = "HISPARSE_STREAM_FORMAT must be pairs, delta, owner, owner24, sweep or bitmap"; return false; }
        }
        if (bitmap) {
            // the per-non-zero passes of the BITMAP builder are kernels too (HISPARSE_BITMAP_BUILD=host: the host loops of round 2,
            // from the row counts the device returned -- the checker of tests/test_gpu_retile.py)
            const char* where = env_switch("HISPARSE_BITMAP_BUILD");
            const bool on_host = !gpu || (where && std::string(where) == "host");
            if (on_host && !csr) gpu.reset();      // (a CSR source has no host fallback for the element formats: keep the tiler)
            if (build_bitmap_tiles(L, channel, n_packets, row_nnz, max_workgroups, out, error, csr, on_host ? nullptr : gpu.get(), image_slack)) return true;
            if (error.rfind("bitmap:", 0) != 0) return false;     // a real decode error
            error.clear();                                       // not representable as a bitmap (duplicate entries): element streams
            // The bitmap builder may have filled `out` before it found the duplicate (the device builder sees it only in its mask
            // pass, after blocks / units / max_block_rows / col_slices were laid out): the element-format path below push_backs onto
            // these tables and derives its sort-key widths from their sizes, so give it `out` as pass 0 left it.
            {
                const uint64_t nnz_keep = out.nnz;      // (no device image to give back: the tiler keeps its buffers until a build succeeds)
                out = StreamTiles();
                out.nnz = nnz_keep;
            }
        }
    }
    // ---- SWEEP (stream_tiles.h): hyper-sparse matrices whose x is gathered from L2 instead of staged in LDS -- its own builder (host threads)
    //      and kernel.  HISPARSE_SWEEP=0|1 and HISPARSE_STREAM_FORMAT=sweep force.
    {
        // Unforced: wherever OWNER24 would be taken (mean position gap > kOwnerMinMeanGap) and SWEEP's plan is modelled faster than OWNER24's.
        // OWNER24 pays ~1.2 us + 0.06 us per wavefront step for every (row range x sub-tile) unit whatever it holds (tools/perf_model.py:
        // UNIT_FLOOR_US, fitted to the rocprofv3 kernels), and its planner cuts at least max_workgroups / 8 row ranges to fill the CUs; the
        // estimate below lands 10 % under the measured steps on six matrices (pokec 87 / 95.5 us, ogbn-products 190 / 205, an 8-way slab of it
        // 60.5 / 59.2, power-law squares 42.5 / 48, 86.6 / 99, 173 / 185), SWEEP's model within 3 %: hence the factor.  What the comparison
        // reproduces (stream_tiles.h, "SWEEP format", has the tables): pokec -> SWEEP, ogbn-products -> OWNER24, ogbn-products cut into 8
        // row slabs (same gap, a quarter of the row ranges: 59.2 -> 46.3 us) -> SWEEP.
        // (the mean position gap INSIDE the (row range x sub-tile) cells that hold anything: a banded or block-diagonal matrix of 12 non-zeros per
        //  row over a million columns is not hyper-sparse where its elements are -- banded 1 M x 1 M, float_stall: 63 us as the SWEEP image the
        //  plain gap asked for, 27 us as a DELTA image)
        const double gap = out.nnz ? double(num_rows) * double(num_cols) / double(out.nnz) * census.populated : 0.0;
        bool sweep = false;
        // (from kSweepMinNnz on; smaller matrices too where x is wider than the LIGHT plan's sixteen sub-tiles -- a quarter slab of a 100 K x 4 M
        //  bipartite graph, 1.5 M non-zeros over 489 sub-tiles: 55.9 us as an OWNER24 image of 15 648 units, 11.2 us as a SWEEP image)
        if (gap > kOwnerMinMeanGap && (out.nnz >= kSweepMinNnz || (out.nnz >= kSweepMinNnzWide && uint64_t(CP) * S > kLightMaxUnits)) && uint64_t(num_cols) * 4 < (1ull << 32)) {
            uint32_t cs = 1, rows_cap = 0;
            uint64_t want = 1;
            const double sweep_us = sweep_plan(L, out.nnz, max_workgroups, cs, want, rows_cap);
            const uint32_t cap = owner_max_block_rows(2), G = std::max<uint32_t>(1, max_workgroups);
            uint64_t by_cap = 0;
            for (uint32_t rp = 0; rp < RP; ++rp) by_cap += (uint64_t(L.rows_in_part(rp)) + cap - 1) / cap;
            const double ranges = double(std::max<uint64_t>(by_cap, G / kMaxColSlices));
            const double units = std::max(1.0, ranges * double(CP) * S * census.populated), per_wg = units / G, unit_steps = double(out.nnz) / units / (kConsumerWaves * kWaveLanes);
            const double owner_slices = std::min<double>(kMaxColSlices, std::max(1.0, std::ceil(G / ranges)));
            const double owner_combine = owner_slices > 1.0 ? 2.0 + double(num_rows) * 4.0 * (owner_slices + 1.0) / 8e6 : 0.0;
            const double owner_us = 1.1 * (std::max(double(out.nnz) * 7.06 / 6.2e6, per_wg * (1.2 + 0.06 * unit_steps)) + 8.0 + owner_combine);
            sweep = sweep_us < owner_us;
            if (env_switch("HISPARSE_PLAN_DEBUG")) std::fprintf(stderr, "format: sweep %.1f us (%u slices) against owner24 %.1f us (%.0f units per workgroup) -> %s\n", sweep_us, cs, owner_us, per_wg, sweep ? "sweep" : "owner24");
        }
        // Short, wide, moderately sparse slabs whose image stays in the Infinity Cache (round 5, the round's last measurement,
        // profiles/r05_hollywood_slab_sweep.txt): one rank's slab of hollywood split 8 ways -- 133 K rows x 1.07 M columns, gap 10 K, 113 MB -- runs
        // 25.4-25.8 us as a SWEEP image (9 slices; ring depth 4, streamed without `nt`) against 31.0 us under the row-block planner's choice (PAIRS, 8
        // slices x 16 units per block of 3.4 K elements: a barrier and an x refill per unit).  Fixed point only, >= 6 columns per row and a gap
        // above kSweepSlabMinMeanGap: what was measured, no further; the float modes and the 4-way slabs keep their plans until they are.
        if (!sweep && !is_float && gap > kSweepSlabMinMeanGap && gap <= kOwnerMinMeanGap && out.nnz >= kSweepMinNnz && uint64_t(num_cols) >= 6ull * num_rows &&
            double(out.nnz) * 8.1 <= double(kResidentMaxImageBytes) && uint64_t(num_cols) * 4 < (1ull << 32)) {
            sweep = true;
            if (env_switch("HISPARSE_PLAN_DEBUG")) std::fprintf(stderr, "format: sweep for a short, wide slab (gap %.0f, %u x %u)\n", gap, num_rows, num_cols);
        }
        // "spmm_vectors" = 4: the caller wants the four-vector SpMM kernel, which runs SWEEP images only (spmm_sweep.hip)
        const char* spmm = env_switch("HISPARSE_SPMM_VECTORS");
        const bool for_spmm = spmm && std::atoi(spmm) == 4 && out.nnz > 0 && uint64_t(num_cols) * 16 < (1ull << 32);
        if (for_spmm) sweep = true;
        if (const char* force = env_switch("HISPARSE_SWEEP")) sweep = std::atoi(force) != 0;      // (1: whatever the matrix)
        if (const char* force = env_switch("HISPARSE_STREAM_FORMAT")) sweep = std::string(force) == "sweep";
        if (sweep) {
            const uint64_t nnz_keep = out.nnz;
            out = StreamTiles();
            out.nnz = nnz_keep;
            if (!build_sweep_tiles(L, channel, n_packets, row_nnz, max_workgroups, out, error, csr, gpu.get(), image_slack)) return false;
            out.spmm_vectors = for_spmm ? 4u : 1u;
            return true;
        }
    }
    // ---- stream format (stream_tiles.h): DELTA for matrices that are sparse but not hyper-sparse; hyper-sparse float matrices: OWNER --
    {
        const double mean_gap = out.nnz ? double(num_rows) * double(num_cols) / double(out.nnz) * census.populated : 1e30;      // (inside the populated cells, see SWEEP above)
        out.format = (mean_gap >= kDeltaMinMeanGap && mean_gap <= kDeltaMaxMeanGap) ? kFormatDelta : kFormatPairs;
        // hyper-sparse matrices: OWNER, in its 7-byte record form (OWNER24) unless that turns out larger (decided after the sort).  Fixed
        // point too since round 3: saturating 32-bit accumulators (spmv_kernels.hip: OwnerOps) -- pokec in PAIRS, with 8-byte atomic
        // accumulators, 12287-row blocks and 26 600 units of 1 150 elements, ran at 24 % of the roofline
        if (mean_gap > kOwnerMinMeanGap && out.nnz >= 4096 && !g_no_owner) out.format = kFormatOwner24;
        if (prefer_sliced_delta) out.format = kFormatDelta;
        if (out.format == kFormatDelta && hub_share >= 0.3 && !prefer_sliced_delta) keep_delta_for_hubs = true;      // (hub rows, above)
        if (env_switch("HISPARSE_PLAN_DEBUG") && hub_share > 0.0)
            std::fprintf(stderr, "format: %.1f %% of the non-zeros in hub rows%s\n", hub_share * 100.0, keep_delta_for_hubs ? " -> DELTA kept for its per-lane row sums" : "");
        if (const char* force = env_switch("HISPARSE_STREAM_FORMAT")) {
            const std::string f(force);
            if (f == "pairs") out.format = kFormatPairs;
            else if (f == "delta") out.format = kFormatDelta;
            else if (f == "owner") out.format = is_float ? kFormatOwner : kFormatPairs;   // float accumulators only
            else if (f == "owner24") out.format = g_no_owner ? kFormatPairs : kFormatOwner24;
            else if (f == "bitmap") {}   // was tried above and is not representable (duplicate entries): automatic choice
            else if (!f.empty()) { error = "HISPARSE_STREAM_FORMAT must be pairs, delta, owner, owner24, sweep or bitmap"; return false; }
        }
    }
    // ---- LIGHT plan (stream_tiles.h): a small matrix is launch-bound in the row-block kernel -- one slice, up to 4 x CUs small blocks of a
    //      PAIRS image, linear dealing, spmv_light_kernel.  Automatic when no format is forced; HISPARSE_LIGHT=0|1 forces (1: with any
    //      matrix of at most kLightMaxUnits sub-tiles whose format is not forced to something other than pairs).
    bool light = false;
    {
        const char* forced = env_switch("HISPARSE_STREAM_FORMAT");
        const bool fits = out.nnz > 0 && uint64_t(CP) * S <= kLightMaxUnits && num_rows < (1u << 31);
        light = fits && !forced && out.nnz <= kLightMaxNnz && !prefer_sliced_delta;
        if (light && sliced_delta_possible && sliced_delta_us < 0.97 * (3.1 + double(out.nnz) * 3.2e-6 * 256.0 / std::max<uint32_t>(1, max_workgroups))) {
            light = false;                       // (10 %-dense layers: see the BITMAP block above)
            prefer_sliced_delta = true;
            out.format = kFormatDelta;
        }
        if (const char* force = env_switch("HISPARSE_LIGHT")) light = std::atoi(force) != 0 && fits && (!forced || std::string(forced) == "pairs");
        if (const char* force_slices = env_switch("HISPARSE_COL_SLICES")) light = light && std::atoi(force_slices) <= 1;      // a forced sliced plan is the row-block kernel's
        if (light) out.format = kFormatPairs;
        out.light = light;
    }
    // Round 6: the format family and the tile plan are decided TOGETHER where they depend on each other -- a float-mode matrix whose row-block plan
    // comes out as ONE column slice of a PAIRS image (ds_add_f64 row sums, 4 095-row blocks) runs 10-30 % faster as an OWNER24 image (owned rows,
    // plain read-modify-write on 4-byte sums, 24 561-row blocks) once it is large enough to amortise OWNER's longer prologue: banded 400 K 30.4 ->
    // 23.5 us, block-diagonal 200 K 18.6 -> 15.0, 600 K 32.1 -> 22.3, tall 2 M x 50 K 31.1 -> 28.0, 3 M x 8 K 31.2 -> 25.8, in float_pob and
    // float_stall alike; sliced plans are a wash (gplus, rmat19, er_300k: +-3 %) and small ones lose (a 4 M-non-zero slab 8.1 -> 9.4 us)
    // (profiles/r06_float_pairs_vs_owner24.txt).  So: plan as before; if that gives float / PAIRS-family / one slice / >= kFloatOneSliceOwnerMinNnz
    // non-zeros, plan again as OWNER24 and take it.
    bool delta = false, owner = false, owner24 = false, format_forced = false;
    uint32_t acc_bytes = kAccumulatorBytes, spare_rows = 1u, light_wgs = kLightWorkgroupsPerCu, G = 1, slices = 1, max_rows = 1;
    double best = 1e30, best_units = 1.0;      // the chosen tile plan's modelled cost BESIDE its stream (us); its non-empty (row range x sub-tile) units
    for (int attempt = 0; attempt < 2; ++attempt) {
        delta = out.format == kFormatDelta;
        owner = out.format == kFormatOwner || out.format == kFormatOwner24;
        owner24 = out.format == kFormatOwner24;      // may still fall back to the 8-byte form (below)
        format_forced = env_switch("HISPARSE_STREAM_FORMAT") != nullptr || prefer_sliced_delta || (keep_delta_for_hubs && out.format == kFormatDelta);      // (the sliced DELTA plan of a dense-row layer is DELTA for its per-lane row sums, not for its bytes)
        acc_bytes = owner ? kOwnerAccumulatorBytes : kAccumulatorBytes;
        spare_rows = owner ? kConsumerWaves : 1u;     // accumulators behind the block's rows that padding elements aim at

        // ---- tile plan: column slices x (rows per block, x ring depth) ------------------------------------------------
        // More column slices = longer row ranges = less x pulled through every CU, at the price of the combine pass; fewer
        // rows per block = deeper x ring = refill latency hidden even when a (row range, sub-tile) unit holds only a few
        // thousand non-zeros (hyper-sparse matrices).  Cost model in microseconds, constants measured on MI355X:
        //   x volume through one CU at ~120 GB/s; a refill takes ~0.8 us to land, ring-1 of them overlap, a unit's stream
        //   time (~25 GB/s per CU) hides the rest; ~8 us of prologue + epilogue per block; the combine kernel.
        light_wgs = kLightWorkgroupsPerCu;
        if (const char* force = env_switch("HISPARSE_LIGHT_WGS")) light_wgs = std::min<uint32_t>(6u, std::max(1, std::atoi(force)));
        G = std::max<uint32_t>(1, max_workgroups) * (light ? light_wgs : 1u);
        slices = 1;
        max_rows = light ? kLightMaxBlockRows : max_block_rows(false);
        if (light) {
            if (const char* force_rows = env_switch("HISPARSE_MAX_ROWS")) max_rows = std::min<uint32_t>(max_rows, std::max(1, std::atoi(force_rows)));   // tests: chains of blocks
        } else {
            const char* force_slices = env_switch("HISPARSE_COL_SLICES");
            const char* force_rows = env_switch("HISPARSE_MAX_ROWS");   // experiments
            struct Shape { uint32_t cap, ring; };
            // OWNER: 4-byte accumulators -> 24561 rows with a ring of 2, 16369 with a ring of 3, sliced or not
            const Shape sliced[2] = {{owner ? owner_max_block_rows(2) : max_block_rows(true), 2},
                                     {owner ? owner_max_block_rows(3) : (kMaxLdsBytes - 3 * kSubTileCols * 4) / kAccumulatorBytes - 1, 3}};   // 12287 / 8191 rows
            const Shape whole[1] = {{max_block_rows(false), kMaxXBuffers}};                                       // 4095 rows, ring 4
            const double sub_tiles = double(CP) * S;
            std::map<uint64_t, TileCensus::Eval> census_memo;
            best = 1e30;
            for (uint32_t cs = 1; cs <= (force_slices ? kMaxForcedColSlices : kMaxColSlices); ++cs) {
                // unforced: every count the cost model likes.  (Through round 4 only 1, 2, 4, 8 for matrices of more than sixteen sub-tiles -- everything
                // in between for OWNER, where the x volume decides: ogbn-products runs 241 us in 5 slices (102 ranges of 24 K rows, 2 blocks per
                // workgroup) against 280 in 2 (127 ranges) and 275 in 4 -- because five slices had measured as a wash on ogbl-ppa and 3 us slower on
                // its R-MAT stand-in, a PAIRS image then.  Measured again in round 5 (profiles/r05_any_slice_count.txt, whole step, alternating):
                // ogbl-ppa 55.2-56.0 us in 4 slices, 54.0-54.5 in 5 (51 row ranges x 5 = 255 blocks: fewer, longer units); the R-MAT stand-in
                // 58.4-59.0 -> 55.2-55.4; hollywood keeps 2, its slabs and ogbl-ppa's keep 8 (a 2-way slab takes 5 or 7: +-1 %).
                // HISPARSE_POW2_SLICES=1 brings the old rule back for the A/B.)
                // (a matrix of at most sixteen sub-tiles: a slice per sub-tile (or two) is the plan without x refills and
                // unit barriers (gplus, 14 sub-tiles: 23.7 us in 7 slices, 26.1 in 8), and a power of two above the sub-tile count would leave
                // whole slices, i.e. workgroups, empty)
                const uint32_t live_tiles = (num_cols + kSubTileCols - 1) / kSubTileCols;
                if (force_slices ? uint32_t(std::atoi(force_slices)) != cs : (!owner && (cs & (cs - 1)) != 0 && live_tiles > 2 * kMaxColSlices && env_switch("HISPARSE_POW2_SLICES"))) continue;
                if (cs > 1 && uint64_t(CP) * S < cs) continue;                                    // fewer sub-tiles than slices
                if (!force_slices && !owner && live_tiles <= kMaxColSlices && cs > live_tiles) continue;
                for (const Shape& shape : (cs > 1 || owner) ? std::vector<Shape>(sliced, sliced + 2) : std::vector<Shape>(whole, whole + 1)) {
                    uint32_t cap = shape.cap, ring = shape.ring;
                    if (force_rows) {
                        cap = std::min<uint32_t>(cap, std::max(1, std::atoi(force_rows)));
                        ring = std::max(kMinXBuffers, std::min(kMaxXBuffers, (kMaxLdsBytes - (cap + spare_rows) * acc_bytes) / (kSubTileCols * 4u)));
                    }
                    const uint64_t per_round = std::max<uint32_t>(1, G / cs);
                    const uint64_t need = (uint64_t(num_rows) + cap - 1) / cap;
                    const double ranges = double(per_round * std::max<uint64_t>(1, (need + per_round - 1) / per_round));
                    const double blocks_per_wg = ranges * cs / G;
                    // the plan's real units and loads (TileCensus): the non-empty (row range x sub-tile) cells, the heaviest workgroup's share
                    const uint64_t memo_key = (uint64_t(ranges) << 8) | cs;
                    auto found = census_memo.find(memo_key);
                    if (found == census_memo.end()) found = census_memo.emplace(memo_key, census.evaluate(uint64_t(ranges), cs, G, out.nnz)).first;
                    const TileCensus::Eval& real = found->second;
                    const double units_per_wg = std::max(1.0, real.nonempty_units / G);
                    const double unit_stream_us = double(out.nnz) * 8.0 / (units_per_wg * G) / 25e3;
                    // x pulled through a CU: 120 GB/s next to a DELTA / PAIRS stream (ogbl-ppa: 0.1 us per row range); OWNER's units are
                    // short and every one ends in a flush and a barrier, which also scale with the ranges: 1.34 us per range on
                    // ogbn-products = 29 GB/s (tools/slices_probe.sh)
                    const double volume_us = real.nonempty_units * double(L.sub_width) * 4.0 / G / (owner ? 29e3 : 120e3);      // (uniform matrix: ranges x num_cols x 4 bytes)
                    double latency_us = units_per_wg * std::max(0.0, 0.8 / (ring - 1) - unit_stream_us);
                    // Blocks of a few long rows (<= kDenseBlockRows) take the dense-row path: a wavefront sums a row in registers and pays a
                    // wavefront-wide reduction at every row change.  That is right for rows that fill many chunks of a sub-tile (pruned-NN
                    // layers: 16 K non-zeros per row) and slow when a (row, sub-tile) holds only a chunk or two -- one rank's slab of mouse_gene
                    // split 8 ways (5632 rows x 45 K columns, 22-row blocks, 117 non-zeros per row and sub-tile) ran 2.5 us per unit, 22.7 us
                    // for 29 MB; in 3 column slices (blocks of 66 rows, ordinary path) 11.5 us + the combine pass.  Price it.
                    // an unsliced block walks ALL sub-tiles: every unit boundary costs it a head record per wavefront, a barrier and a refill
                    // issue, ~0.3 us that the stream does not hide (gplus, 14 units per block: 28.6 us in one slice, 24.0 in seven, same
                    // format; mouse_gene's 2-way slabs 21.9 -> 20.8) -- sliced plans have a fraction of the units and pay the combine pass instead
                    if (!owner && cs == 1) latency_us += units_per_wg * 0.3;
                    // few sub-tiles dealt to slices that do not divide them: the blocks of the slices with one sub-tile more set the time (gplus,
                    // 14 sub-tiles: 23.9 / 27.4 / 24.7 / 26.1 us in 5 / 6 / 7 / 8 slices)
                    if (!owner && cs > 1 && live_tiles <= 2 * kMaxColSlices)
                        latency_us += 0.75 * (double(out.nnz) * 8.0 / G / 25e3) * (double((live_tiles + cs - 1) / cs) * cs / live_tiles - 1.0);
                    const double rows_per_block = double(num_rows) / ranges, per_row_and_tile = double(out.nnz) / std::max(1.0, double(num_rows) * sub_tiles * census.populated);
                    if (!owner && rows_per_block <= kDenseBlockRows && per_row_and_tile < 4.0 * kWaveLanes) latency_us += units_per_wg * 1.75;
                    // PAIRS deals a unit's elements, sorted by (row, column), to the lanes in consecutive runs: the 64 lanes of a step sit
                    // 1/896 of the unit apart, and when the block has fewer than 896 rows several of them are in the SAME row -- their
                    // ds_add_u64 on one accumulator are serialised.  One rank's slab of mouse_gene split 4 ways (44-row blocks, ~20 lanes
                    // per row): 15-26 us in one slice against 12.7-13.7 us in six (268-row blocks, one sub-tile each, combine pass included).
                    // ~2 clocks per extra lane and wavefront step, all wavefronts of a workgroup through the one LDS.  (DELTA blocks of
                    // long rows keep per-lane sums instead -- no atomics to collide.)
                    const double lanes_per_row = std::min(64.0, 896.0 / std::max(1.0, rows_per_block));
                    // (DELTA is still tentative here: below ~1.6 bytes saved per non-zero x nnz < the threshold it falls back to PAIRS, see "DELTA or PAIRS")
                    const bool pairs_likely = !delta || (!format_forced && double(out.nnz) * 1.6 < double(is_float ? kDeltaMinSavedBytesFloat : kDeltaMinSavedBytes));
                    const double conflict_us = (!owner && pairs_likely && lanes_per_row > 1.0 && per_row_and_tile >= 16.0)
                                                   ? double(out.nnz) / G / kWaveLanes * (lanes_per_row - 1.0) * 2.0 / 2400.0 : 0.0;
                    // the combine pass: a launch of its own (3.5 us) + its traffic -- or ~1 us of the NEXT step's kernel where the image is small
                    // enough for the carried combine (hs_api.cpp; stream_tiles.h: kCarryMaxImageBytes)
                    const bool carried = double(out.nnz) * 8.1 < double(kCarryMaxImageBytes);
                    const double combine_us = cs > 1 ? (carried ? 1.0 : 3.5) + double(num_rows) * 4.0 * (cs + 1) / 4e6 : 0.0;
                    // workgroup slots that get no block (7 slices x 36 row ranges = 252 blocks on 256 workgroups): the stream they would have taken
                    // is the others' -- what tells 7 slices from 8 on mid-size wide matrices (profiles/r05_any_slice_count.txt)
                    // -- round 6: the heaviest workgroup's real share (TileCensus): the same term for a uniform matrix, and what makes column slices of a
                    // banded matrix as expensive as they are (most of its (row range x slice) blocks are empty)
                    // (charged beyond the uniform picture only where the real imbalance exceeds it by more than 15 %: the slice counts of the scrambled
                    //  graphs were settled by measurement to within a microsecond -- ogbl-ppa 5 slices, gplus 7 -- and the census rows are coarser than that)
                    const double uniform_load = double(out.nnz) / std::max(1.0, ranges * cs) * std::ceil(blocks_per_wg);
                    const double idle_us = double(out.nnz) * 8.0 / 6.2e6 * ((std::ceil(blocks_per_wg) / std::max(1e-9, blocks_per_wg) - 1.0) +
                                                                         std::max(0.0, real.max_wg_load / std::max(1.0, uniform_load) - 1.15) * uniform_load / std::max(1.0, double(out.nnz) / G));
                    const double cost = volume_us + latency_us + conflict_us + 8.0 * blocks_per_wg + combine_us + idle_us;
                    if (detail::env_switch("HISPARSE_PLAN_DEBUG"))
                        std::fprintf(stderr, "plan cs %u cap %u ring %u: ranges %.0f volume %.1f latency %.1f conflicts %.1f blocks/wg %.2f idle %.2f combine %.1f => %.2f us\n", cs, cap, ring, ranges,
                                     volume_us, latency_us, conflict_us, blocks_per_wg, idle_us, combine_us, cost);
                    if (cost < best) { best = cost; best_units = real.nonempty_units; slices = cs; max_rows = cap; }
                }
            }
        }
        // Round 6: below the hyper-sparse border a SWEEP plan can still replace the row-block plan -- where the units are tiny AND the models agree.  A
        // (row range x sub-tile) unit costs the row-block kernel a barrier, a refill and a head record whatever it holds; a matrix of few long rows
        // over millions of columns -- 2048 x 8 M, 2000 per row, mean gap 4000: below every gap rule -- has 131 elements per unit: 67.5 us as the PAIRS
        // image the gap rule gives it, 22.3 us as a SWEEP image (tools/planner_check.py --second).  Both conditions: fewer than 1024 elements per
        // non-empty unit of the chosen plan (the mechanism), and SWEEP's whole modelled step under 60 % of the row-block plan's stream + plan cost
        // (the two models were fitted apart and the row-block one reads 30-80 % high in absolute terms: on their own they would send one rank's slab
        // of mouse_gene -- 14 K elements per unit, 8.3 us as PAIRS, 12.5 as SWEEP -- the wrong way).
        if (attempt == 0 && !owner && !light && !format_forced && !env_switch("HISPARSE_SWEEP") && !env_switch("HISPARSE_COL_SLICES") && !env_switch("HISPARSE_MAX_ROWS") &&
            out.nnz >= kSweepMinNnzWide && uint64_t(num_cols) * 4 < (1ull << 32) && best < 1e29) {
            uint32_t cs = 1, rows_cap = 0;
            uint64_t want = 1;
            const double sweep_us = sweep_plan(L, out.nnz, max_workgroups, cs, want, rows_cap);
            const double rowblock_us = double(out.nnz) * 8.0 / 6.2e6 + best;
            if (env_switch("HISPARSE_PLAN_DEBUG")) std::fprintf(stderr, "format: row-block plan %.1f us (stream + %.1f; %.0f elements per unit) against sweep %.1f us (%u slices)\n", rowblock_us, best, double(out.nnz) / std::max(1.0, best_units), sweep_us, cs);
            if (sweep_us < 0.6 * rowblock_us && double(out.nnz) / std::max(1.0, best_units) < 1024.0) {
                const uint64_t nnz_keep = out.nnz;
                out = StreamTiles();
                out.nnz = nnz_keep;
                return build_sweep_tiles(L, channel, n_packets, row_nnz, max_workgroups, out, error, csr, gpu.get(), image_slack);
            }
        }
        const bool pairs_family = out.format == kFormatPairs || (out.format == kFormatDelta && double(out.nnz) * 1.6 < double(kDeltaMinSavedBytesFloat));
        if (attempt == 0 && is_float && !owner && !light && !format_forced && !g_no_owner && pairs_family && slices == 1 && out.nnz >= kFloatOneSliceOwnerMinNnz &&
            !env_switch("HISPARSE_COL_SLICES") && !env_switch("HISPARSE_MAX_ROWS")) {
            if (env_switch("HISPARSE_PLAN_DEBUG")) std::fprintf(stderr, "format: float mode, one-slice PAIRS-family plan of %llu non-zeros -> planned again as OWNER24\n", (unsigned long long)out.nnz);
            out.format = kFormatOwner24;
            continue;
        }
        break;
    }
    if (uint64_t(slices) * num_rows > 0xffffffffull) {   // Block::out_offset = slice * num_rows + row0 is a 32-bit word offset
        while (slices > 1 && uint64_t(slices) * num_rows > 0xffffffffull) slices /= 2;
        max_rows = owner ? owner_max_block_rows(2) : slices > 1 ? max_block_rows(true) : max_block_rows(false);
    }
    out.col_slices = slices;

    // ---- row ranges: equal non-zero count, <= max_rows rows, never across a row partition ---------------------------
    using Range = RowRange;
    std::vector<Range> ranges;
    std::vector<uint64_t> range_nnz;
    // as many ranges as workgroup slots (G / slices), or the next multiple of that when the LDS row cap forces more, so that
    // every workgroup ends up with the same number of blocks
    const uint64_t per_round = std::max<uint32_t>(1, G / slices);
    const uint64_t rounds = std::max<uint64_t>(1, ((uint64_t(num_rows) + max_rows - 1) / max_rows + per_round - 1) / per_round);
    const uint64_t want_ranges = std::max<uint64_t>(1, std::min<uint64_t>(per_round * rounds, out.nnz / (light ? kLightMinBlockNnz : 4096u)));
    build_row_ranges_at_most(L, row_nnz, out.nnz, want_ranges, max_rows, ranges, range_nnz, out.nnz / 4096 >= per_round * rounds ? per_round : 0);
    const uint32_t NR = uint32_t(ranges.size());
    std::vector<uint32_t> block_of_row(num_rows);   // row -> row range
    for (uint32_t b = 0; b < NR; ++b) {
        std::fill(block_of_row.begin() + ranges[b].row0, block_of_row.begin() + ranges[b].row0 + ranges[b].nrows, b);
        out.max_block_rows = std::max(out.max_block_rows, ranges[b].nrows);
    }
    const uint32_t ring_fit = (kMaxLdsBytes - (((out.max_block_rows + spare_rows) * acc_bytes + 15u) & ~15u)) / (kSubTileCols * 4u);
    out.ring_buffers = std::max(kMinXBuffers, std::min(kMaxXBuffers, ring_fit));

    timer.lap("plan + row ranges");
    // ---- pass 1: elements per (row range, column partition, sub-tile, source channel) -------------------
    const size_t slots_per_range = size_t(CP) * S * NUM_HBM_CHANNELS;
    // (row range, sub-tile, source channel) counters: rows x columns / (rows per block x 8192) x 16 words -- a matrix of 10^8 x 10^8
    // would ask for gigabytes here and for as many Unit descriptors: refuse instead of swapping
    if (uint64_t(NR) * slots_per_range > (uint64_t(1) << 28)) {
        error = "matrix too large for this build: " + std::to_string(NR) + " row ranges x " + std::to_string(uint64_t(CP) * S) + " x sub-tiles exceed the unit table";
        return false;
    }
    std::vector<uint32_t> cnt(size_t(NR) * slots_per_range, 0);
    auto slot = [&](uint32_t b, uint32_t cp, uint32_t s, uint32_t pc) { return size_t(b) * slots_per_range + (size_t(cp) * S + s) * NUM_HBM_CHANNELS + pc; };
    std::vector<WalkResult> res1(size_t(RP) * CP * NUM_HBM_CHANNELS);
    const size_t walk_tasks = L.cross_parts ? size_t(CP) * NUM_HBM_CHANNELS : res1.size();      // host passes 1 and 2 (see pass 1)
    if (gpu) {      // totals per (range, sub-tile); the per-source-channel split only serves the host's scatter
        std::vector<uint32_t> totals;
        if (!gpu->count_tiles(block_of_row, NR, totals)) { error = gpu->error(); return false; }
        for (uint32_t b = 0; b < NR; ++b)
            for (uint32_t cp = 0; cp < CP; ++cp)
                for (uint32_t sub = 0; sub < S; ++sub) cnt[slot(b, cp, sub, 0)] = totals[(size_t(b) * CP + cp) * S + sub];
    } else {
    // (row ranges that cross partition borders: the counters of a (range, sub-tile, channel) are fed from two row partitions, so one task
    //  takes ALL row partitions of its (column partition, channel), one after the other)
    parallel_for(walk_tasks, [&](size_t w) {
        const uint32_t pc = uint32_t(w % NUM_HBM_CHANNELS), cp = uint32_t((w / NUM_HBM_CHANNELS) % CP);
        for (uint32_t rp = L.cross_parts ? 0u : uint32_t(w / NUM_HBM_CHANNELS / CP), rp_end = L.cross_parts ? RP : rp + 1; rp < rp_end; ++rp)
            res1[(size_t(rp) * CP + cp) * NUM_HBM_CHANNELS + pc] = walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp, [&](uint32_t row, uint32_t col, uint32_t) {
                cnt[slot(block_of_row[row], cp, col / L.sub_width, pc)]++;
            });
    });
    for (const auto& r : res1)
        if (!r.ok) { error = r.error; return false; }
    }

    timer.lap("pass 1 (unit counts)");
    // ---- enumerate blocks (row range x column slice) and their units; counts -> offsets into a scratch element list ----
    std::vector<UnitPlan> plans;
    std::vector<uint32_t> unit_of(size_t(NR) * CP * S, 0xffffffffu);  // (row range, cp, s) -> unit index
    const uint32_t sub_tiles = CP * S;
    uint64_t scratch_elems = 0;
    std::vector<uint64_t> tile_nnz(sub_tiles);
    std::vector<uint32_t> slice_of(sub_tiles), by_weight(sub_tiles);
    for (uint32_t b = 0; b < NR; ++b) {
        // Column slices of this row range: its sub-tiles are dealt to the slices heaviest first, each to the lightest slice
        // so far.  (Round-robin by index left the slices of a power-law graph 7 % apart -- the columns of sub-tile 0 are
        // the popular ones in EVERY row range -- and with one block per workgroup the slowest block is the kernel time.)
        for (uint32_t k = 0; k < sub_tiles; ++k) {
            tile_nnz[k] = 0;
            for (uint32_t pc = 0; pc < NUM_HBM_CHANNELS; ++pc) tile_nnz[k] += cnt[slot(b, k / S, k % S, pc)];
        }
        std::iota(by_weight.begin(), by_weight.end(), 0u);
        std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t x, uint32_t y) { return tile_nnz[x] > tile_nnz[y]; });
        uint64_t slice_load[kMaxForcedColSlices] = {0};
        uint32_t slice_tiles[kMaxForcedColSlices] = {0};
        for (uint32_t k : by_weight) {
            uint32_t best = 0;
            for (uint32_t c = 1; c < slices; ++c)
                if (slice_load[c] < slice_load[best] || (slice_load[c] == slice_load[best] && slice_tiles[c] < slice_tiles[best])) best = c;
            slice_of[k] = best;
            slice_load[best] += tile_nnz[k];
            slice_tiles[best] += 1;     // empty sub-tiles still spread evenly (they cost nothing, but keep the rule simple)
        }
        for (uint32_t slice = 0; slice < slices; ++slice) {   // device block index (before finish_blocks) = b * slices + slice
            Block blk{};
            blk.row0 = ranges[b].row0;
            blk.nrows = ranges[b].nrows;
            blk.row_part = ranges[b].row_part;
            blk.last_part = ranges[b].last_part;
            if (delta) {   // long rows: position gaps well inside a row (HISPARSE_ROW_RUNS=0|1 forces, for the tests)
                // (over the rows that HAVE non-zeros: the padding rows at the end of a float_stall matrix would make the last block look sparse)
                uint32_t live_rows = 0, heaviest = 0;
                for (uint32_t r = 0; r < ranges[b].nrows; ++r) {
                    live_rows += row_nnz[ranges[b].row0 + r] != 0;
                    heaviest = std::max(heaviest, row_nnz[ranges[b].row0 + r]);
                }
                const double gap = range_nnz[b] ? double(live_rows) * double(num_cols) / double(range_nnz[b]) : 1e30;
                // (a hub row: an eighth of the block in one row -- eight lanes of every step, more in the hub's own sub-tiles, would add to ONE accumulator; see "hub rows" above)
                const bool hub_block = heaviest >= 4096 && uint64_t(heaviest) * 8 >= range_nnz[b];
                const char* force = env_switch("HISPARSE_ROW_RUNS");
                blk.flags = (force ? std::atoi(force) != 0 : (gap < kDenseMeanGap || hub_block)) ? kBlockDenseRows : 0u;
            } else if (owner) {
                blk.flags = 0;
            } else if (light) {
                blk.flags = 0;                    // strided dealing for every block: a lane of the light kernel walks consecutive sorted elements
            } else {
                blk.flags = (ranges[b].nrows <= kDenseBlockRows && range_nnz[b] >= 64ull * ranges[b].nrows) ? kBlockDenseRows : 0u;
            }
            blk.out_offset = slices > 1 ? slice * num_rows + ranges[b].row0 : ranges[b].row0;
            blk.unit_begin = uint32_t(out.units.size());
            for (uint32_t k = 0; k < sub_tiles; ++k) {
                if (slice_of[k] != slice) continue;
                const uint32_t cp = k / S, s = k % S;
                uint64_t n = 0;
                for (uint32_t pc = 0; pc < NUM_HBM_CHANNELS; ++pc) {   // counts -> exclusive offsets inside the unit
                    const uint32_t c = cnt[slot(b, cp, s, pc)];
                    cnt[slot(b, cp, s, pc)] = uint32_t(n);
                    n += c;
                }
                if (n == 0) continue;
                if (n > 0x7fffffffull) { error = "unit too large"; return false; }
                UnitPlan up;
                up.n = uint32_t(n);
                up.scratch = scratch_elems;
                scratch_elems += n;
                Unit u{};
                u.col0 = uint32_t(uint64_t(cp) * geom.logical_vb + uint64_t(s) * L.sub_width);
                u.ncols = std::min<uint32_t>(L.sub_width, L.cols_in_part(cp) - s * L.sub_width);
                unit_of[(size_t(b) * CP + cp) * S + s] = uint32_t(out.units.size());
                out.units.push_back(u);
                plans.push_back(up);
            }
            blk.unit_end = uint32_t(out.units.size());
            out.blocks.push_back(blk);
        }
    }
    const uint32_t NB = uint32_t(out.blocks.size());
    const uint32_t NU = uint32_t(out.units.size());
    std::vector<uint32_t> range_of_block(NB);
    for (uint32_t bi = 0; bi < NB; ++bi) range_of_block[bi] = bi / slices;     // blocks were pushed range by range, slice by slice

    timer.lap("enumerate blocks + units");
    // ---- pass 2: collect every unit's elements as (position, value), position = local_row * 8192 + local_col -------
    std::vector<uint64_t> scratch;   // high word position, low word value: sorts by position (host path)
    std::vector<uint32_t> block_of_unit(NU);
    for (uint32_t bi = 0; bi < NB; ++bi)
        for (uint32_t u = out.blocks[bi].unit_begin; u < out.blocks[bi].unit_end; ++u) block_of_unit[u] = bi;
    if (gpu) {
        std::vector<uint32_t> range_row0(NR);
        for (uint32_t b = 0; b < NR; ++b) range_row0[b] = ranges[b].row0;
        bool duplicates = false;
        if (!gpu->sort_elements(block_of_row, range_row0, unit_of, plans, duplicates)) { error = gpu->error(); return false; }
        if (duplicates) { error = "gpu re-tile: duplicate entries"; return false; }      // the caller rebuilds on the host
        std::vector<uint32_t>().swap(cnt);
        timer.lap("gpu: keys + radix sort");
        if (delta) {
            if (!gpu->delta_slots(plans)) { error = gpu->error(); return false; }
        } else {
            for (UnitPlan& up : plans) up.slots = up.n;
        }
    } else {
    scratch.resize(scratch_elems);
    parallel_for(walk_tasks, [&](size_t w) {
        const uint32_t pc = uint32_t(w % NUM_HBM_CHANNELS), cp = uint32_t((w / NUM_HBM_CHANNELS) % CP);
        for (uint32_t rp = L.cross_parts ? 0u : uint32_t(w / NUM_HBM_CHANNELS / CP), rp_end = L.cross_parts ? RP : rp + 1; rp < rp_end; ++rp)
            walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp, [&](uint32_t row, uint32_t col, uint32_t val) {
                const uint32_t b = block_of_row[row], s = col / L.sub_width;
                const UnitPlan& up = plans[unit_of[(size_t(b) * CP + cp) * S + s]];
                const uint32_t pos = (row - ranges[b].row0) * kSubTileCols + (col - s * L.sub_width);
                scratch[up.scratch + cnt[slot(b, cp, s, pc)]++] = (uint64_t(pos) << 32) | val;
            });
    });
    std::vector<uint32_t>().swap(cnt);

    timer.lap("pass 2 (scatter)");
    // ---- per unit: sort by position; DELTA: count slots (elements + bridges for gaps that do not fit 16 bits) -------
    parallel_for(NU, [&](size_t u) {
        UnitPlan& up = plans[u];
        uint64_t* e = scratch.data() + up.scratch;
        std::sort(e, e + up.n);
        uint64_t slots = up.n;
        for (uint32_t i = 1; delta && i < up.n; ++i) {
            const uint64_t d = (e[i] >> 32) - (e[i - 1] >> 32);
            if (d > kMaxGap) slots += (d - kMaxGap + kBridgeAdvance - 1) / kBridgeAdvance;
        }
        up.slots = slots;
    });
    }

    // DELTA or PAIRS, now that every unit's slots are known (automatic choice only):
    //  * DELTA pays for every position gap beyond 16 bits with a bridge slot.  A graph whose gaps are heavy-tailed (R-MAT: a quarter
    //    of the rows empty, hubs of 10^5 non-zeros) needs one for every 25th element although its MEAN gap looks fine: 4 % more slots
    //    and still 11 % fewer bytes than PAIRS (58.5-58.9 us against 59.4-59.8; what made it 82-85 us through round 4 was the dealing of
    //    the runs, see first_slot above, not the bridges).  More than 5 % bridge slots -> PAIRS.
    //  * DELTA's 6-byte slots only pay when the stream is what bounds the kernel.  Measured over 20 shapes (tools/probe_synth.py,
    //    40000^2 and 400000 x 100000 power-law matrices at mean gaps 16 ... 4096, ogbl-ppa, mouse_gene):
    //    t(DELTA) - t(PAIRS) = (bytes saved) / 6.5 TB/s - c with c = 3.5 us fixed point, 6 us float (more instructions per element,
    //    a head record per unit and wavefront).  So: DELTA only when it saves more than kDeltaMinSavedBytes of stream.
    if (delta && !format_forced) {
        uint64_t slots = 0, pairs_bytes = 0, delta_bytes = 0;
        for (const UnitPlan& up : plans) {
            slots += up.slots;
            const uint64_t pairs_chunks = (uint64_t(up.n) + kWaveLanes - 1) / kWaveLanes, records = (up.slots + kWaveLanes - 1) / kWaveLanes;
            pairs_bytes += pairs_chunks * kChunkBytes;
            // slots + one head per wavefront with work, in records of two slots (a run's last record is half empty every other time)
            delta_bytes += (records + std::min<uint64_t>(records, kConsumerWaves) * 3 / 2) * (kRecordBytes / 2);
        }
        // (the fixed cost c is mostly the head record per unit and wavefront: 14 units per block -> 3.5 us, but a sliced plan with one or two
        // units per block pays ~1.2 us -- gplus in 7 slices: 24 MB saved, 24.7 us in PAIRS, 21.9 in DELTA.  Fixed point only: measured there.)
        const double units_per_block = double(plans.size()) / std::max<uint32_t>(1, NB);
        const uint64_t min_saved = is_float ? kDeltaMinSavedBytesFloat
                                            : std::min<uint64_t>(kDeltaMinSavedBytes, uint64_t((1.0 + 0.18 * units_per_block) * 6.5e6));
        if (double(slots) > 1.05 * double(out.nnz) || pairs_bytes < delta_bytes + min_saved) {
            delta = false;
            out.format = kFormatPairs;
            for (UnitPlan& up : plans) up.slots = up.n;
            for (uint32_t bi = 0; bi < NB; ++bi) {
                const uint32_t b = range_of_block[bi];
                out.blocks[bi].flags = (ranges[b].nrows <= kDenseBlockRows && range_nnz[b] >= 64ull * ranges[b].nrows) ? kBlockDenseRows : 0u;
            }
        }
    }
    timer.lap("sort units");
    // ---- OWNER: every unit's elements, sorted by (row, column), are cut into the 14 wavefronts' shares (balanced_owner_shares) -----
    auto owner_shares = [&](uint32_t max_span) -> bool {
        if (gpu) {
            if (!gpu->owner_shares(plans, max_span)) { error = gpu->error(); return false; }
            return true;
        }
        parallel_for(NU, [&](size_t u) {
            UnitPlan& up = plans[u];
            const uint64_t* e = scratch.data() + up.scratch;
            auto row_of = [&](uint32_t i) { return uint32_t(e[i] >> (32 + kOwnerColBits)); };
            balanced_owner_shares(up.n, row_of, up.own_begin, max_span);
            for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                const bool any = up.own_begin[w + 1] > up.own_begin[w];
                up.own_row[w] = any ? row_of(up.own_begin[w]) : 0u;
                up.own_last[w] = any ? row_of(up.own_begin[w + 1] - 1) : 0u;
            }
        });
        return true;
    };
    if (owner) {
        if (!owner_shares(owner24 ? kOwnerShareRows : 0xffffffffu)) return false;
        if (owner24) {
            // OWNER24 holds a share's rows relative to its first row in 11 bits and a wavefront's step count in 16: otherwise, or when
            // the row cap has cut so many shares short that the 7-byte records are no smaller than 8-byte chunks, keep the 8-byte form
            bool fits = true;
            uint64_t bytes24 = 0, bytes32 = 0;
            for (uint32_t bi = 0; bi < NB && fits; ++bi) {
                uint64_t steps[kConsumerWaves] = {0};
                for (uint32_t u = out.blocks[bi].unit_begin; u < out.blocks[bi].unit_end; ++u) {
                    const UnitPlan& up = plans[u];
                    bytes32 += (uint64_t(up.n) + kWaveLanes - 1) / kWaveLanes * kChunkBytes;
                    for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                        steps[w] += (up.own_begin[w + 1] - up.own_begin[w] + kWaveLanes - 1) / kWaveLanes;
                        if (up.own_last[w] - up.own_row[w] >= kOwnerShareRows || up.own_row[w] > 0xffffu) fits = false;
                    }
                }
                for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                    if (steps[w] > kOwnerStepMask) fits = false;
                    bytes24 += (steps[w] + kOwnerRecordSteps - 1) / kOwnerRecordSteps * kOwnerRecordBytes;
                }
            }
            if (!is_float && !fits) {      // fixed point has no 8-byte OWNER form to fall back to: plan again without OWNER
                error = kOwnerDoesNotFit;
                return false;
            }
            if (is_float && (!fits || (!format_forced && double(bytes24) > 0.97 * double(bytes32)))) {
                owner24 = false;
                out.format = kFormatOwner;
                if (!owner_shares(0xffffffffu)) return false;
            }
        }
        timer.lap("owner shares");
    }
    // ---- PAIRS with 24-bit position words (stream_tiles.h: PAIRS24): 7 bytes per element where 11 bits of row are enough ----------
    bool aux24 = false;
    {
        // Opt-in (HISPARSE_AUX_BITS=24): measured SLOWER than the 8-byte form although it streams 12 % fewer bytes (mouse_gene 40.8 vs
        // 39.7 us): a step becomes two loads (one of them unaligned) instead of one dwordx2, and the kernels are bound by the number of
        // memory requests a CU keeps in flight, not by the bytes.
        const char* bits = env_switch("HISPARSE_AUX_BITS");
        const bool allowed = bits && std::atoi(bits) == 24;
        if (!owner && !delta && !light) aux24 = allowed && out.max_block_rows <= kAux24MaxRows;
        if (aux24) out.format = kFormatPairs24;
    }
    const uint32_t chunk_bytes = aux24 ? kChunkBytes24 : kChunkBytes, wave_stride = chunk_bytes * kConsumerWaves;
    // one element slot of a chunk: value word + position word (32-bit: interleaved pairs; 24-bit: 64 values, then 64 x 3 bytes)
    auto put = [&](uint8_t* chunk, uint32_t lane, uint32_t value, uint32_t where) {
        if (aux24) {
            reinterpret_cast<uint32_t*>(chunk)[lane] = value;
            uint8_t* a = chunk + kWaveLanes * 4 + lane * 3;
            a[0] = uint8_t(where); a[1] = uint8_t(where >> 8); a[2] = uint8_t(where >> 16);
        } else {
            reinterpret_cast<uint32_t*>(chunk)[2 * lane] = value;
            reinterpret_cast<uint32_t*>(chunk)[2 * lane + 1] = where;
        }
    };

    // ---- per block: deal every unit's 64-slot chunks to the consumer wavefronts round-robin; lay out the streams -------
    // DELTA runs are dealt lane-major ("spread"; HISPARSE_DELTA_DEAL=wave: rounds 1-4's dealing, kept for the A/B): see first_slot below
    bool delta_spread = true;
    if (const char* deal = env_switch("HISPARSE_DELTA_DEAL")) delta_spread = std::string(deal) != "wave";
    std::vector<uint64_t> block_nnz(NB, 0);   // weight of a block for the workgroup assignment
    uint64_t image_bytes = 0;
    for (uint32_t bi = 0; bi < NB; ++bi) {
        Block& blk = out.blocks[bi];
        uint32_t pos[kConsumerWaves] = {0};   // chunk / record position of every wavefront in its stream
        uint32_t chunk_counter = 0;
        for (uint32_t u = blk.unit_begin; u < blk.unit_end; ++u) {
            UnitPlan& up = plans[u];
            if (up.slots > 0x7fffffffull) { error = "unit too large"; return false; }
            if (owner) {
                // the unit's elements are sorted by (row, column): wavefront w's share is the contiguous stretch of its rows;
                // steps = its 64-slot chunks; lane l takes elements [l * steps, (l + 1) * steps) of the share
                for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                    const uint32_t steps = (up.own_begin[w + 1] - up.own_begin[w] + kWaveLanes - 1) / kWaveLanes;
                    up.run_len[w] = steps;
                    up.start_step[w] = up.start_record[w] = pos[w];
                    pos[w] += steps;
                    // OWNER24: the share's first row rides in the high half (stream_tiles.h)
                    out.units[u].end_step[w] = owner24 ? (pos[w] | up.own_row[w] << 16) : pos[w];
                    out.elements += uint64_t(steps) * kWaveLanes;
                }
                continue;
            }
            const uint32_t chunks = uint32_t((up.slots + kWaveLanes - 1) / kWaveLanes);
            uint32_t run[kConsumerWaves] = {0};
            for (uint32_t c = 0; c < chunks; ++c) run[(chunk_counter + c) % kConsumerWaves]++;
            up.chunks = chunks;
            up.base = chunk_counter;
            chunk_counter += chunks;
            uint64_t first = 0;
            for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                up.run_len[w] = run[w];
                // DELTA: which run of the position-sorted unit lane l of wavefront w walks.  "spread": runs are dealt lane-major (run
                // l * 14 + w), so the 64 lanes of a wavefront sit a 64th of the unit apart, like PAIRS' chunks.  "wave" (rounds 1-4): the
                // wavefront owns a contiguous 1/14 of the unit, its lanes consecutive runs of it -- then a hub row of a heavy-tailed
                // graph (R-MAT: 850 of a unit's 10 K elements in ONE row) holds ALL 64 lanes of a wavefront on one accumulator, step after
                // step: 64 ds_add_u64 on one address.  Measured (profiles/r05_delta_dealing.txt, whole step): R-MAT ogbl-ppa 85.1 -> 58.9 us
                // (PAIRS: 59.4-59.8), transformer-80 10.9 -> 9.9-10.2, mouse_gene 34.2 -> 33.9, gplus 20.1 -> 19.9, ogbl-ppa / hollywood equal.
                up.first_slot[w] = delta_spread ? first / kWaveLanes : first;
                up.lane_stride[w] = delta_spread ? chunks : run[w];
                first += uint64_t(run[w]) * kWaveLanes;
                up.start_step[w] = up.start_record[w] = pos[w];
                // DELTA: one head record (absolute positions) in front of the wavefront's records of this unit
                pos[w] += delta ? (run[w] ? (run[w] + 2) / 2 : 0) : run[w];      // DELTA: records of two slots, the head is the first slot
                out.units[u].end_step[w] = pos[w];
            }
            out.elements += uint64_t(chunks) * kWaveLanes;
        }
        for (uint32_t w = 0; w < kConsumerWaves; ++w) block_nnz[bi] += pos[w];   // the block's weight: wavefront steps, heads included
        // the kernel addresses a wavefront's stream with a 32-bit byte offset from Block::wave_offset
        for (uint32_t w = 0; w < kConsumerWaves; ++w)
            if (uint64_t(pos[w]) * (delta ? kRecordBytes : owner ? chunk_bytes : wave_stride) >= (1ull << 32)) { error = "row block stream exceeds 4 GiB"; return false; }
        if (delta || owner) {      // every wavefront's steps are contiguous
            for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                blk.wave_offset[w] = image_bytes;
                // OWNER24: whole records of four steps (the steps behind the last one are never consumed)
                image_bytes += owner24 ? uint64_t((pos[w] + kOwnerRecordSteps - 1) / kOwnerRecordSteps) * kOwnerRecordBytes
                                       : uint64_t(pos[w]) * (delta ? kRecordBytes : chunk_bytes);
            }
        } else {
            // chunks are stored in dealing order (global chunk g of the block at g * 512 bytes; wavefront w consumes
            // g = w, w + 14, w + 28, ...): the 14 wavefronts of a workgroup sweep ONE contiguous region together
            for (uint32_t w = 0; w < kConsumerWaves; ++w) blk.wave_offset[w] = image_bytes + uint64_t(w) * chunk_bytes;
            image_bytes += uint64_t(chunk_counter) * chunk_bytes;
        }
    }

    // ---- workgroups: longest-processing-time assignment of blocks (tiles_common.h) ------------------------------
    std::vector<std::vector<uint32_t>> mine;
    {
        // Blocks to XCDs by column slice (tiles_common.h: assign_workgroups_by_slice) -- OPT-IN (HISPARSE_XCD_AFFINITY=1).  Built in round 3
        // for matrices whose x outgrows an XCD's L2 (ogbn-products: 9.8 MB of x, 5 slices; ~20 % of the x refills miss L2) and measured:
        // same kernel time (203.6 vs 205.5 us, same box) and, by the counters, the SAME traffic (1084.7 vs 1087.5 MB of reads per launch:
        // 208 MB of it x either way) -- the refills are evicted by the matrix stream passing through the same L2, not by the other
        // slices' x.  Kept for experiments; the default stays the spread assignment.
        bool by_slice = false;
        if (const char* force = env_switch("HISPARSE_XCD_AFFINITY")) by_slice = std::atoi(force) != 0 && slices > 1 && G % 8 == 0 && NB >= G;
        if (by_slice) {
            std::vector<uint32_t> slice_of_block(NB);
            for (uint32_t bi = 0; bi < NB; ++bi) slice_of_block[bi] = bi % slices;      // blocks were pushed range by range, slice by slice
            assign_workgroups_by_slice(out, block_nnz, G, RP, slice_of_block, mine);
        } else {
            assign_workgroups(out, block_nnz, G, RP, mine);
        }
    }
    // Final block order (chain_blocks) after the copies of unit data the kernel wants inside the Block have been filled in.
    auto finish_blocks = [&]() {
        for (Block& blk : out.blocks) {
            if (blk.unit_end > blk.unit_begin) {
                for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                    blk.total_steps[w] = out.units[blk.unit_end - 1].end_step[w] & (owner24 ? kOwnerStepMask : 0xffffffffu);
                    blk.first_end[w] = out.units[blk.unit_begin].end_step[w];      // OWNER24: with the first share's row_base
                }
                blk.first_col0 = out.units[blk.unit_begin].col0;
                blk.first_ncols = out.units[blk.unit_begin].ncols;
            }
        }
        chain_blocks(out, mine, RP);
    };

    timer.lap("stream layout + workgroups");
    if (gpu) {
        if (!gpu->emit(out.format, image_bytes, image_slack, plans, block_of_unit, out.blocks, is_float)) { error = gpu->error(); return false; }
        out.d_image = gpu->release_image();
        out.image_bytes = image_bytes;
        finish_blocks();
        timer.lap("gpu: emit");
        return true;
    }
    resize_zeroed(out.image, image_bytes);
    out.image_bytes = image_bytes;
    uint8_t* image = out.image.data();

    if (owner) {
        // ---- OWNER: per (unit, wavefront) share: slot (step s, lane l) holds element l * steps + s of the share; the position word
        //      IS the element's (local_row << 13 | local_col); padding aims a zero at the wavefront's own spare accumulator.
        //      OWNER24: step S of the wavefront's stream = slot S % 4 of record S / 4, rows relative to the share's first row -------
        parallel_for(NU, [&](size_t u) {
            const UnitPlan& up = plans[u];
            const Block& blk = out.blocks[block_of_unit[u]];
            const uint64_t* e = scratch.data() + up.scratch;
            for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                const uint32_t steps = up.run_len[w], n = up.own_begin[w + 1] - up.own_begin[w];
                const uint64_t* mine = e + up.own_begin[w];
                uint8_t* stream = image + blk.wave_offset[w];
                for (uint32_t st = 0; st < steps; ++st) {
                    const uint32_t S = up.start_step[w] + st;
                    for (uint32_t l = 0; l < kWaveLanes; ++l) {
                        const uint64_t i = uint64_t(l) * steps + st;
                        const uint32_t value = i < n ? uint32_t(mine[i]) : 0u, pos = i < n ? uint32_t(mine[i] >> 32) : 0u;
                        if (owner24) {
                            uint8_t* rec = stream + uint64_t(S / kOwnerRecordSteps) * kOwnerRecordBytes;
                            const uint32_t j = S % kOwnerRecordSteps;
                            const uint32_t where = i < n ? pos - (up.own_row[w] << kOwnerColBits) : kOwnerSpareField << kOwnerColBits;
                            reinterpret_cast<uint32_t*>(rec)[l * kOwnerRecordSteps + j] = value;
                            uint8_t* a = rec + kOwnerRecordValueBytes + (l * kOwnerRecordSteps + j) * 3;
                            a[0] = uint8_t(where); a[1] = uint8_t(where >> 8); a[2] = uint8_t(where >> 16);
                        } else {
                            put(stream + uint64_t(S) * chunk_bytes, l, value, i < n ? pos : (blk.nrows + w) << kOwnerColBits);
                        }
                    }
                }
            }
        });
        finish_blocks();
        timer.lap("emit OWNER");
        return true;
    }
    if (!delta) {
        // ---- PAIRS: normal units: slot (chunk c, lane l) holds sorted element l * chunks + c (neighbouring lanes far
        //      apart in the unit); dense-row units: element i sits in chunk i / 64, lane i % 64 ------------------------------
        parallel_for(NU, [&](size_t u) {
            const UnitPlan& up = plans[u];
            const Block& blk = out.blocks[block_of_unit[u]];
            const bool dense = blk.flags & kBlockDenseRows;
            const uint64_t* e = scratch.data() + up.scratch;
            const uint64_t total = uint64_t(up.chunks) * kWaveLanes;
            for (uint64_t i = 0; i < total; ++i) {
                const uint32_t lane = dense ? uint32_t(i % kWaveLanes) : uint32_t(i / up.chunks);
                const uint32_t c = dense ? uint32_t(i / kWaveLanes) : uint32_t(i % up.chunks);
                const uint32_t g = up.base + c, w = g % kConsumerWaves;
                const uint32_t first = up.base + (w + kConsumerWaves - up.base % kConsumerWaves) % kConsumerWaves;  // first chunk of wave w in this unit
                const uint32_t step = up.start_step[w] + (g - first) / kConsumerWaves;
                uint8_t* chunk = image + blk.wave_offset[w] + uint64_t(step) * wave_stride;
                const uint32_t row_shift = aux24 ? kOwnerColBits : 16u;
                if (i < up.n) {
                    const uint32_t pos = uint32_t(e[i] >> 32);
                    put(chunk, lane, uint32_t(e[i]), ((pos / kSubTileCols) << row_shift) | (pos % kSubTileCols));
                } else {            // padding: zero value aimed at the block's scratch row
                    put(chunk, lane, 0u, blk.nrows << row_shift);
                }
            }
        });
        finish_blocks();
        timer.lap("emit PAIRS");
        return true;
    }

    // ---- DELTA: lane l of wavefront w owns run_len[w] consecutive slots of the sorted unit --------------------------------
    parallel_for(NU, [&](size_t u) {
        const UnitPlan& up = plans[u];
        const Block& blk = out.blocks[block_of_unit[u]];
        const uint64_t* e = scratch.data() + up.scratch;
        // expand to the slot sequence: gap (16 bit, 0xffff = bridge: advance 65535, no element), value, position after the slot
        std::vector<uint16_t> gap(up.slots);
        std::vector<uint32_t> val(up.slots), after(up.slots);
        uint64_t k = 0;
        uint32_t at = uint32_t(e[0] >> 32);
        for (uint32_t i = 0; i < up.n; ++i) {
            uint64_t d = (e[i] >> 32) - at;
            while (d > kMaxGap) { at += kBridgeAdvance; d -= kBridgeAdvance; gap[k] = kBridgeGap; val[k] = 0; after[k] = at; ++k; }
            at += uint32_t(d);
            gap[k] = uint16_t(d); val[k] = uint32_t(e[i]); after[k] = at; ++k;
        }
        const uint16_t pad_gap = is_float ? kBridgeGap : 0;
        const uint32_t scratch_pos = blk.nrows * kSubTileCols;   // local row nrows = the scratch accumulator
        for (uint32_t w = 0; w < kConsumerWaves; ++w) {
            if (!up.run_len[w]) continue;
            uint8_t* rec = image + blk.wave_offset[w] + uint64_t(up.start_record[w]) * kRecordBytes;
            // slot q of the run (q = 0: the head) sits in record q / 2, half q % 2: value word at (2 lane + half) * 4, gap at 512 + lane * 4 + half * 2
            auto value_at = [&](uint32_t q, uint32_t l) -> uint32_t& { return reinterpret_cast<uint32_t*>(rec + uint64_t(q / 2) * kRecordBytes)[2 * l + q % 2]; };
            auto gap_at = [&](uint32_t q, uint32_t l) -> uint16_t& { return reinterpret_cast<uint16_t*>(rec + uint64_t(q / 2) * kRecordBytes + kWaveLanes * 8)[2 * l + q % 2]; };
            const uint32_t slots_in_records = (up.run_len[w] + 2) / 2 * 2;      // head + run, rounded up to whole records
            for (uint32_t l = 0; l < kWaveLanes; ++l) {
                const uint64_t s0 = up.first_slot[w] + uint64_t(l) * up.lane_stride[w];
                // position BEFORE the run's first slot; slot 0 carries gap 0 from the first element's own position
                value_at(0, l) = s0 >= up.slots ? scratch_pos : (s0 == 0 ? uint32_t(e[0] >> 32) : after[s0 - 1]);
                for (uint32_t j = 0; j < up.run_len[w]; ++j) {
                    const uint64_t si = s0 + j;
                    value_at(j + 1, l) = si < up.slots ? val[si] : 0u;
                    gap_at(j + 1, l) = si < up.slots ? gap[si] : pad_gap;   // padding: see consume_block
                }
                for (uint32_t q = up.run_len[w] + 1; q < slots_in_records; ++q) gap_at(q, l) = pad_gap;      // the dead slot of an odd run
            }
        }
    });
    finish_blocks();
    timer.lap("emit DELTA");
    return true;
}

}  // namespace

}  // namespace dev
}  // namespace hisparse
