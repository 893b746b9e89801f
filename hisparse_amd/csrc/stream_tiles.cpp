// stream_tiles.cpp — CPSR image -> row-block element streams (see stream_tiles.h for the format and the why).
//
// The decode follows the reference's loader in MEANING (header layout, per-lane lengths, marker = row
// advance, interleaved virtual channels):
//   spmv/libfpga/spmv_cluster.h:41-98        fixed point, INTERLEAVE_FACTOR 1
//   spmv-fp/libfpga/spmv_cluster.h:46-117    float, INTERLEAVE_FACTOR 1 or 8
// with the row <-> (channel, lane, round) mapping of sw/data_formatter.h:410,432 and the result drain
// order of spmv/spmv_result_drain.cpp:36,104-113 (net effect: natural row order in y).
#include "stream_tiles.h"

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>

namespace hisparse {
namespace dev {

namespace {

template <typename Fn>
void parallel_for(size_t n, Fn fn) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned threads = unsigned(std::min<size_t>(hw ? hw : 1u, n));
    if (threads <= 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<size_t> next(0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        });
    for (auto& th : pool) th.join();
}

struct Layout {
    const Geometry* g;
    uint32_t num_rows, num_cols, row_parts, col_parts, F;
    uint32_t sub_width;     // columns per x sub-tile
    uint32_t subs_per_cp;   // sub-tiles per column partition
    uint32_t rows_in_part(uint32_t rp) const {
        uint64_t lo = uint64_t(rp) * g->logical_ob;
        return uint32_t(std::min<uint64_t>(g->logical_ob, num_rows - lo));
    }
    uint32_t cols_in_part(uint32_t cp) const {
        uint64_t lo = uint64_t(cp) * g->logical_vb;
        return uint32_t(std::min<uint64_t>(g->logical_vb, num_cols - lo));
    }
};

struct WalkResult {
    bool ok = true;
    std::string error;
    uint64_t nnz = 0;
};

// Visit every non-zero of physical channel `pc` in partition (rp, cp): visit(absolute_row, partition_local_col, value_word).
template <typename Visit>
WalkResult walk_channel_partition(const Layout& L, const MatPkt* buf, uint64_t n_pkts, uint32_t pc, uint32_t rp, uint32_t cp,
                                  Visit visit) {
    WalkResult res;
    const uint32_t F = L.F;
    const uint32_t parts = L.row_parts * L.col_parts;
    const uint64_t pid = uint64_t(rp) * L.col_parts + cp;     // j outer, i inner (sw/benchmark.cpp:142-143)
    const uint64_t header = pid * (1 + F);
    const uint64_t payload_base = uint64_t(parts) * (1 + F);  // spmv_cluster.h:41 / fp :46
    auto fail = [&](const std::string& what) {
        res.ok = false;
        res.error = "channel " + std::to_string(pc) + ", row partition " + std::to_string(rp) + ", column partition " +
                    std::to_string(cp) + ": " + what;
        return res;
    };
    if (header + 1 + F > n_pkts) return fail("partition header lies outside the channel buffer");
    const uint64_t start = buf[header].indices.data[0];        // already multiplied by F (benchmark.cpp:178-179)
    const uint64_t stride = uint64_t(PACK_SIZE) * NUM_HBM_CHANNELS * F;  // rows between two rows of one lane stream
    const uint64_t row_base = uint64_t(rp) * L.g->logical_ob;
    const uint64_t row_limit = row_base + L.rows_in_part(rp);
    const uint32_t col_limit = L.cols_in_part(cp);
    const bool fixed = L.g->impl == IMPL_FIXED;

    for (uint32_t f = 0; f < F; ++f) {
        const PackedWord& lens = buf[header + 1 + f].indices;
        uint32_t longest = 0;
        for (uint32_t k = 0; k < PACK_SIZE; ++k) longest = std::max(longest, lens.data[k]);
        if (longest && payload_base + start + uint64_t(longest - 1) * F + f >= n_pkts)
            return fail("payload runs past the end of the channel buffer");
        const uint32_t vc = pc + f * NUM_HBM_CHANNELS;         // benchmark.cpp:146
        uint64_t row[PACK_SIZE];
        for (uint32_t k = 0; k < PACK_SIZE; ++k) row[k] = row_base + uint64_t(vc) * PACK_SIZE + k;  // round 0, data_formatter.h:410
        const MatPkt* pkt = buf + payload_base + start + f;
        for (uint32_t p = 0; p < longest; ++p, pkt += F) {
            for (uint32_t k = 0; k < PACK_SIZE; ++k) {
                if (p >= lens.data[k]) continue;               // lane exhausted: zero padding
                const uint32_t col = pkt->indices.data[k], val = pkt->vals.data[k];
                if (col == IDX_MARKER) {
                    // fixed: integer part of the Q8.24 word (spmv_cluster.h:82); float: raw bits (fp :104)
                    row[k] += uint64_t(fixed ? (val >> 24) : val) * stride;
                } else {
                    if (col >= col_limit) return fail("column index " + std::to_string(col) + " outside the column partition");
                    if (row[k] >= row_limit) return fail("decoded row outside the row partition (marker count wrapped?)");
                    visit(uint32_t(row[k]), col, val);
                    ++res.nnz;
                }
            }
        }
    }
    return res;
}

struct UnitPlan {           // host-side companion of a device Unit
    uint64_t n = 0;         // real elements
    uint32_t chunks = 0;    // ceil(n / 64)
    uint32_t base = 0;      // chunk counter of the block at the unit's first chunk
    bool dense = false;     // row-sorted, linear dealing (Block::flags & kBlockDenseRows)
    size_t row_cursor = 0;  // dense: index of this unit's per-row cursors in `dense_cnt`
    uint32_t start_step[kConsumerWaves];
};

}  // namespace

bool build_stream_tiles(const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                        const Geometry& geom, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                        uint32_t num_col_partitions, uint32_t max_workgroups, StreamTiles& out, std::string& error) {
    Layout L;
    L.g = &geom;
    L.num_rows = num_rows;
    L.num_cols = num_cols;
    L.row_parts = num_row_partitions;
    L.col_parts = num_col_partitions;
    L.F = geom.interleave;
    L.sub_width = uint32_t(std::min<uint64_t>(kSubTileCols, geom.logical_vb));
    L.subs_per_cp = uint32_t((geom.logical_vb + L.sub_width - 1) / L.sub_width);
    const uint32_t F = L.F, CP = num_col_partitions, RP = num_row_partitions, S = L.subs_per_cp;
    const uint64_t header_pkts = uint64_t(RP) * CP * (1 + F);
    for (uint32_t c = 0; c < NUM_HBM_CHANNELS; ++c) {
        if (!channel[c] && n_packets[c]) { error = "null channel buffer"; return false; }
        if (n_packets[c] < header_pkts) { error = "channel " + std::to_string(c) + " is shorter than its partition headers"; return false; }
    }
    out = StreamTiles();
    auto chan = [&](uint32_t pc) { return static_cast<const MatPkt*>(channel[pc]); };

    // ---- pass 0: non-zeros per row (rows of different physical channels are disjoint) ------------
    std::vector<uint32_t> row_nnz(num_rows, 0);
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

    // ---- column slices: trade x broadcast (every workgroup pulls its slice of x through its CU at ~120 GB/s) against
    //      the combine pass (one more small kernel reading `slices` partial vectors) --------------------------------
    const uint32_t G = std::max<uint32_t>(1, max_workgroups);
    uint32_t slices = 1;
    {
        const char* force = std::getenv("HISPARSE_COL_SLICES");
        double best = 1e30;
        for (uint32_t cs = 1; cs <= kMaxColSlices; cs *= 2) {
            if (force && uint32_t(std::atoi(force)) != cs) continue;
            if (cs > 1 && uint64_t(CP) * S < cs) continue;                                    // fewer sub-tiles than slices
            const uint32_t cap = cs > 1 ? kMaxSlicedBlockRows : kMaxBlockRows;
            const uint64_t ranges_est = std::max<uint64_t>((G + cs - 1) / cs, (uint64_t(num_rows) + cap - 1) / cap);
            // every row range pulls all of x (split over its slices) through the CUs that own it
            const double fill_us = double(ranges_est) * double(num_cols) * 4.0 / double(G) / 120e3;   // bytes / (120 GB/s) in us
            const double combine_us = cs > 1 ? 3.5 + double(num_rows) * 4.0 * (cs + 1) / 4e6 : 0.0;
            if (fill_us + combine_us < best) { best = fill_us + combine_us; slices = cs; }
        }
    }
    // gather mode: with the best LDS plan, how many non-zeros would one (row range, sub-tile) unit hold?
    {
        const uint32_t cap = slices > 1 ? kMaxSlicedBlockRows : kMaxBlockRows;
        const uint64_t ranges_est = std::max<uint64_t>((G + slices - 1) / slices, (uint64_t(num_rows) + cap - 1) / cap);
        const double per_unit = double(out.nnz) / (double(ranges_est) * double(CP) * double(S));
        const char* mode = std::getenv("HISPARSE_XMODE");
        // Measured on MI355X: uncoalesced 4-byte gathers run at ~0.8 lanes/clk/CU (ogbl-ppa 218 us vs 62 us with LDS
        // staging; ogbn-products 907 us vs 780-880 us), so gather mode never wins today.  It stays selectable
        // (HISPARSE_XMODE=gather) as the tested fallback for matrices whose units are tiny; `per_unit` is what a
        // future heuristic would look at.
        (void)per_unit;
        out.gather_x = mode && std::string(mode) == "gather";
        if (out.gather_x) slices = 1;
    }
    out.col_slices = slices;
    const uint32_t max_rows = out.gather_x ? kMaxGatherBlockRows : (slices > 1 ? kMaxSlicedBlockRows : kMaxBlockRows);

    // ---- row ranges: equal non-zero count, <= max_rows rows, never across a row partition ---------------------------
    struct Range { uint32_t row0, nrows, row_part; };
    std::vector<Range> ranges;
    std::vector<uint64_t> range_nnz;
    // as many ranges as workgroup slots (G / slices), or the next multiple of that when the LDS row cap forces more, so that
    // every workgroup ends up with the same number of blocks
    const uint64_t per_round = std::max<uint32_t>(1, G / slices);
    const uint64_t rounds = std::max<uint64_t>(1, ((uint64_t(num_rows) + max_rows - 1) / max_rows + per_round - 1) / per_round);
    const uint64_t want_ranges = std::max<uint64_t>(1, std::min<uint64_t>(per_round * rounds, out.nnz / 4096));
    const uint64_t target = std::max<uint64_t>(1, (out.nnz + want_ranges - 1) / want_ranges);
    for (uint32_t rp = 0; rp < RP; ++rp) {
        const uint32_t lo = uint32_t(uint64_t(rp) * geom.logical_ob), hi = lo + L.rows_in_part(rp);
        uint32_t r0 = lo;
        uint64_t acc = 0;
        for (uint32_t r = lo; r < hi; ++r) {
            // close the range BEFORE a row that would overshoot the target by more than the range undershoots now
            const uint64_t with = acc + row_nnz[r];
            if (r > r0 && (r - r0 == max_rows || (with > target && with - target > target - std::min(acc, target)))) {
                ranges.push_back(Range{r0, r - r0, rp});
                range_nnz.push_back(acc);
                r0 = r;
                acc = 0;
            }
            acc += row_nnz[r];
        }
        if (hi > r0) {
            ranges.push_back(Range{r0, hi - r0, rp});
            range_nnz.push_back(acc);
        }
    }
    const uint32_t NR = uint32_t(ranges.size());
    std::vector<uint32_t> block_of_row(num_rows);   // row -> row range
    for (uint32_t b = 0; b < NR; ++b) {
        std::fill(block_of_row.begin() + ranges[b].row0, block_of_row.begin() + ranges[b].row0 + ranges[b].nrows, b);
        out.max_block_rows = std::max(out.max_block_rows, ranges[b].nrows);
    }
    const uint32_t ring_fit = (kMaxLdsBytes - (out.max_block_rows + 1) * 8u) / (kSubTileCols * 4u);
    out.ring_buffers = out.gather_x ? 0u : std::max(kMinXBuffers, std::min(kMaxXBuffers, ring_fit));

    // dense-row blocks additionally count per (sub-tile, row): their units are stored sorted by row
    std::vector<size_t> dense_base(NR, SIZE_MAX);
    size_t dense_total = 0;
    for (uint32_t b = 0; b < NR; ++b)
        if (ranges[b].nrows <= kDenseBlockRows && range_nnz[b] >= 64ull * ranges[b].nrows) {
            dense_base[b] = dense_total;
            dense_total += size_t(CP) * S * ranges[b].nrows;
        }
    std::vector<uint32_t> dense_cnt(dense_total, 0);
    auto dense_slot = [&](uint32_t b, uint32_t cp, uint32_t s, uint32_t local_row) {
        return dense_base[b] + (size_t(cp) * S + s) * ranges[b].nrows + local_row;
    };

    // ---- pass 1: elements per (block, column partition, sub-tile, source channel) -------------------
    const size_t slots_per_block = size_t(CP) * S * NUM_HBM_CHANNELS;
    std::vector<uint32_t> cnt(size_t(NR) * slots_per_block, 0);
    auto slot = [&](uint32_t b, uint32_t cp, uint32_t s, uint32_t pc) { return size_t(b) * slots_per_block + (size_t(cp) * S + s) * NUM_HBM_CHANNELS + pc; };
    std::vector<WalkResult> res1(size_t(RP) * CP * NUM_HBM_CHANNELS);
    parallel_for(res1.size(), [&](size_t w) {
        const uint32_t pc = uint32_t(w % NUM_HBM_CHANNELS), cp = uint32_t((w / NUM_HBM_CHANNELS) % CP), rp = uint32_t(w / NUM_HBM_CHANNELS / CP);
        res1[w] = walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp, [&](uint32_t row, uint32_t col, uint32_t) {
            const uint32_t b = block_of_row[row], s = col / L.sub_width;
            cnt[slot(b, cp, s, pc)]++;
            if (dense_base[b] != SIZE_MAX) dense_cnt[dense_slot(b, cp, s, row - ranges[b].row0)]++;   // a row belongs to one channel: no race
        });
    });
    for (const auto& r : res1)
        if (!r.ok) { error = r.error; return false; }

    // ---- plan units and wavefront streams ---------------------------------------------------------------
    const bool rotate = [] { const char* e = std::getenv("HISPARSE_ROTATE"); return e ? std::atoi(e) != 0 : false; }();
    std::vector<UnitPlan> plans;
    std::vector<uint32_t> unit_of(size_t(NR) * CP * S, 0xffffffffu);  // (row range, cp, s) -> unit index
    std::vector<uint64_t> block_nnz;
    uint64_t image_bytes = 0;
    const uint32_t sub_tiles = CP * S;
    for (uint32_t b = 0; b < NR; ++b) {
        for (uint32_t slice = 0; slice < slices; ++slice) {   // device block index = b * slices + slice
            Block blk{};
            blk.row0 = ranges[b].row0;
            blk.nrows = ranges[b].nrows;
            blk.row_part = ranges[b].row_part;
            blk.flags = dense_base[b] != SIZE_MAX ? kBlockDenseRows : 0u;
            blk.out_offset = slices > 1 ? slice * num_rows + ranges[b].row0 : ranges[b].row0;
            blk.unit_begin = uint32_t(out.units.size());
            uint32_t pos[kConsumerWaves] = {0};   // stream position of every wavefront, in steps
            uint32_t chunk_counter = 0;
            uint64_t elems = 0;
            // HISPARSE_ROTATE=1 (experiment, off by default: measured neutral) starts every block at a different sub-tile
            const uint32_t rot = rotate ? uint32_t((uint64_t(b) * 0x9e3779b1u) % sub_tiles) : 0u;
            for (uint32_t k0 = 0; k0 < sub_tiles; ++k0) {
                const uint32_t k = (k0 + rot) % sub_tiles;
                if (k % slices != slice) continue;                    // sub-tiles are dealt round-robin to the column slices
                const uint32_t cp = k / S, s = k % S;
                uint64_t n = 0;
                for (uint32_t pc = 0; pc < NUM_HBM_CHANNELS; ++pc) {   // counts -> exclusive offsets inside the unit
                    const uint32_t c = cnt[slot(b, cp, s, pc)];
                    cnt[slot(b, cp, s, pc)] = uint32_t(n);
                    n += c;
                }
                if (n == 0) continue;
                if (n > 0xffffffffull) { error = "unit too large"; return false; }
                UnitPlan up;
                up.n = n;
                up.chunks = uint32_t((n + kWaveLanes - 1) / kWaveLanes);
                up.base = chunk_counter;
                if (dense_base[b] != SIZE_MAX) {   // per-row counts -> exclusive offsets (the fill pass bumps them)
                    up.dense = true;
                    up.row_cursor = dense_slot(b, cp, s, 0);
                    uint32_t acc = 0;
                    for (uint32_t r = 0; r < blk.nrows; ++r) {
                        const uint32_t c = dense_cnt[up.row_cursor + r];
                        dense_cnt[up.row_cursor + r] = acc;
                        acc += c;
                    }
                }
                Unit u{};
                u.col0 = uint32_t(uint64_t(cp) * geom.logical_vb + uint64_t(s) * L.sub_width);
                u.ncols = std::min<uint32_t>(L.sub_width, L.cols_in_part(cp) - s * L.sub_width);
                for (uint32_t w = 0; w < kConsumerWaves; ++w) up.start_step[w] = pos[w];
                for (uint32_t c = 0; c < up.chunks; ++c) pos[(chunk_counter + c) % kConsumerWaves]++;
                chunk_counter += up.chunks;
                for (uint32_t w = 0; w < kConsumerWaves; ++w) u.end_step[w] = pos[w];
                unit_of[(size_t(b) * CP + cp) * S + s] = uint32_t(out.units.size());
                out.units.push_back(u);
                plans.push_back(up);
                out.elements += uint64_t(up.chunks) * kWaveLanes;
                elems += n;
            }
            blk.unit_end = uint32_t(out.units.size());
            // chunks are stored in dealing order (global chunk g of the block at g * 512 bytes; wavefront w consumes
            // g = w, w + 14, w + 28, ...): the 14 wavefronts of a workgroup sweep ONE contiguous region together instead of
            // 14 separate ones (3584 independent sequential streams chip-wide thrash the DRAM row buffers: measured 5.5 vs
            // 7.0 TB/s for a plain streaming read)
            for (uint32_t w = 0; w < kConsumerWaves; ++w) blk.wave_offset[w] = image_bytes + uint64_t(w) * kChunkBytes;
            image_bytes += uint64_t(chunk_counter) * kChunkBytes;
            out.blocks.push_back(blk);
            block_nnz.push_back(elems);
        }
    }
    const uint32_t NB = uint32_t(out.blocks.size());

    // ---- workgroups: longest-processing-time assignment of blocks ------------------------------------------
    {
        const uint32_t groups = std::min<uint32_t>(G, std::max<uint32_t>(1, NB));
        out.num_workgroups = groups;
        std::vector<uint32_t> order(NB);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return block_nnz[a] > block_nnz[b]; });
        std::vector<std::vector<uint32_t>> mine(groups);
        std::vector<uint64_t> load(groups, 0);
        for (uint32_t b : order) {
            uint32_t best = uint32_t(std::min_element(load.begin(), load.end()) - load.begin());
            mine[best].push_back(b);
            load[best] += block_nnz[b] + 1024;   // every block also costs a fixed prologue/epilogue
        }
        out.wg_first.assign(groups + 1, 0);
        for (uint32_t g = 0; g < groups; ++g) {
            out.wg_first[g] = uint32_t(out.block_order.size());
            out.block_order.insert(out.block_order.end(), mine[g].begin(), mine[g].end());
        }
        out.wg_first[groups] = uint32_t(out.block_order.size());
    }

    // ---- pass 2: scatter the elements into the wavefront streams -------------------------------------------
    out.image.assign(image_bytes, 0);
    uint8_t* image = out.image.data();
    // normal units: slot (chunk c, lane l) holds element l * chunks + c (neighbouring lanes far apart in the unit);
    // dense-row units: element i sits in chunk i / 64, lane i % 64 of the row-sorted unit.
    auto element_address = [&](const Block& blk, const UnitPlan& up, uint64_t i) -> uint8_t* {
        const uint32_t lane = up.dense ? uint32_t(i % kWaveLanes) : uint32_t(i / up.chunks);
        const uint32_t c = up.dense ? uint32_t(i / kWaveLanes) : uint32_t(i % up.chunks);
        const uint32_t g = up.base + c, w = g % kConsumerWaves;
        const uint32_t first = up.base + (w + kConsumerWaves - up.base % kConsumerWaves) % kConsumerWaves;  // first chunk of wave w in this unit
        const uint32_t step = up.start_step[w] + (g - first) / kConsumerWaves;
        return image + blk.wave_offset[w] + uint64_t(step) * kWaveStrideBytes + lane * 8;
    };
    parallel_for(res1.size(), [&](size_t w) {
        const uint32_t pc = uint32_t(w % NUM_HBM_CHANNELS), cp = uint32_t((w / NUM_HBM_CHANNELS) % CP), rp = uint32_t(w / NUM_HBM_CHANNELS / CP);
        walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp, [&](uint32_t row, uint32_t col, uint32_t val) {
            const uint32_t b = block_of_row[row], s = col / L.sub_width;
            const Block& blk = out.blocks[size_t(b) * slices + (cp * S + s) % slices];
            const UnitPlan& up = plans[unit_of[(size_t(b) * CP + cp) * S + s]];
            const uint64_t i = up.dense ? dense_cnt[up.row_cursor + (row - blk.row0)]++ : cnt[slot(b, cp, s, pc)]++;
            uint32_t* e = reinterpret_cast<uint32_t*>(element_address(blk, up, i));
            e[0] = val;
            e[1] = ((row - blk.row0) << 16) | (col - s * L.sub_width);
        });
    });
    // padding slots: zero value aimed at the block's scratch row
    parallel_for(NB, [&](size_t b) {
        const Block& blk = out.blocks[b];
        for (uint32_t u = blk.unit_begin; u < blk.unit_end; ++u) {
            const UnitPlan& up = plans[u];
            for (uint64_t i = up.n; i < uint64_t(up.chunks) * kWaveLanes; ++i) {
                uint32_t* e = reinterpret_cast<uint32_t*>(element_address(blk, up, i));
                e[0] = 0;
                e[1] = blk.nrows << 16;
            }
        }
    });
    return true;
}

}  // namespace dev
}  // namespace hisparse
