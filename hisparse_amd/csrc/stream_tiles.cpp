// stream_tiles.cpp — CPSR image -> stream tiles (see stream_tiles.h for the format and the why).
//
// The decode follows the reference's loader line by line in MEANING (header layout, per-lane
// lengths, marker = row advance, interleaved virtual channels):
//   spmv/libfpga/spmv_cluster.h:41-98        fixed point, INTERLEAVE_FACTOR 1
//   spmv-fp/libfpga/spmv_cluster.h:46-117    float, INTERLEAVE_FACTOR 1 or 8
// and the row <-> (channel, lane, round) mapping of sw/data_formatter.h:410,432 plus the result
// drain order of spmv/spmv_result_drain.cpp:36,104-113 (net effect: natural row order).
#include "stream_tiles.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace hisparse {
namespace dev {

namespace {

unsigned host_threads() {
    unsigned hw = std::thread::hardware_concurrency();
    return hw ? hw : 1u;
}

template <typename Fn>
void parallel_for(size_t n, Fn fn) {
    unsigned threads = unsigned(std::min<size_t>(host_threads(), n));
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

struct TilePiece {   // a piece as seen from its tile: element range [a, b) of the tile's virtual stream
    uint64_t a, b;
    uint32_t steps;
    uint64_t offset;
};

struct Layout {
    const Geometry* g;
    uint32_t num_rows, num_cols, row_parts, col_parts;
    uint32_t F, streams_per_tile;
    uint32_t tile_id(uint32_t rp, uint32_t cp) const { return cp * row_parts + rp; }  // column tile outermost
    uint32_t rows_in_part(uint32_t rp) const {
        uint64_t lo = uint64_t(rp) * g->logical_ob;
        return uint32_t(std::min<uint64_t>(g->logical_ob, num_rows - lo));
    }
    uint32_t cols_in_tile(uint32_t cp) const {
        uint64_t lo = uint64_t(cp) * g->logical_vb;
        return uint32_t(std::min<uint64_t>(g->logical_vb, num_cols - lo));
    }
};

// Counting sink (pass 1) / writing sink (pass 2) share the per-lane encoder below.
struct CountSink {
    void put(uint64_t, uint32_t, uint32_t, bool, uint32_t) {}
    void flag(uint64_t) {}
};

struct WriteSink {
    uint8_t* image;
    const TilePiece* pieces;  // of the current tile, contiguous ranges in ascending order
    size_t num_pieces;
    size_t cursor = 0;

    const TilePiece& locate(uint64_t e) {
        while (e >= pieces[cursor].b) ++cursor;
        while (e < pieces[cursor].a) --cursor;
        return pieces[cursor];
    }
    static uint8_t* chunk_of(uint8_t* image, const TilePiece& p, uint64_t rel, uint32_t& lane, uint32_t& step) {
        const uint64_t run = rel / p.steps;
        step = uint32_t(rel % p.steps);
        lane = uint32_t(run % kWaveLanes);
        return image + p.offset + (run / kWaveLanes) * chunk_bytes(p.steps);
    }
    void put(uint64_t e, uint32_t col, uint32_t val, bool flagged, uint32_t row_before) {
        const TilePiece& p = locate(e);
        uint32_t lane, step;
        uint8_t* chunk = chunk_of(image, p, e - p.a, lane, step);
        if (step == 0) reinterpret_cast<uint32_t*>(chunk)[lane] = row_before;
        uint8_t* group = chunk + kChunkHeaderBytes + uint64_t(p.steps) * 8 + uint64_t(step / kStepsPerGroup) * kGroupBytes;
        const uint32_t slot = lane * kStepsPerGroup + (step % kStepsPerGroup);
        reinterpret_cast<uint16_t*>(group)[slot] = uint16_t(col);
        reinterpret_cast<uint32_t*>(group + kGroupColsBytes)[slot] = val;
        if (flagged) set_flag(chunk, step, lane);
    }
    void flag(uint64_t e) {
        const TilePiece& p = locate(e);
        uint32_t lane, step;
        uint8_t* chunk = chunk_of(image, p, e - p.a, lane, step);
        set_flag(chunk, step, lane);
    }
    static void set_flag(uint8_t* chunk, uint32_t step, uint32_t lane) {
        // several host threads fill different lanes of one wavefront chunk concurrently
        uint64_t* word = reinterpret_cast<uint64_t*>(chunk + kChunkHeaderBytes) + step;
        __atomic_fetch_or(word, uint64_t(1) << lane, __ATOMIC_RELAXED);
    }
};

// State of one lane stream while it is re-encoded.
struct LaneEncoder {
    uint64_t cur_row = 0;      // absolute row the next non-zero belongs to
    uint32_t kernel_row = 0;   // row the kernel's run-local counter holds at this point
    bool kernel_row_valid = false;
    bool row_open = false;     // the current row already emitted a non-zero
    uint64_t last_nnz = 0;     // element index of that row's latest non-zero
    uint64_t e = 0;            // next element index in the tile's virtual stream

    template <typename Sink>
    void nonzero(Sink& sink, uint32_t col, uint32_t val) {
        if (!kernel_row_valid || kernel_row != cur_row) {
            sink.put(e++, kSpecialCol, uint32_t(cur_row), true, kernel_row_valid ? kernel_row : 0u);  // ROWSET
            kernel_row = uint32_t(cur_row);
            kernel_row_valid = true;
        }
        sink.put(e, col, val, false, kernel_row);
        last_nnz = e++;
        row_open = true;
    }
    template <typename Sink>
    void marker(Sink& sink, uint64_t advance_rows, uint32_t stride) {
        if (advance_rows == 0) return;  // a zero-count marker leaves the row index unchanged (spmv_cluster.h:82)
        if (row_open) {
            sink.flag(last_nnz);        // end of row: the kernel flushes and steps its counter by one stride
            kernel_row += stride;
            row_open = false;
        }
        cur_row += advance_rows * stride;
    }
};

struct GroupResult {
    bool ok = true;
    std::string error;
    uint64_t nnz = 0;
};

// Walk the 8 lanes x F virtual channels of physical channel `pc` in partition (rp, cp).
// `first_e[f*8+k]` = element index at which lane stream (pc, f, k) starts inside the tile's virtual stream.
template <typename Sink>
GroupResult walk_channel_partition(const Layout& L, const MatPkt* buf, uint64_t n_pkts, uint32_t pc, uint32_t rp, uint32_t cp,
                                   const uint64_t* first_e, uint32_t* enc_len, Sink& sink) {
    GroupResult res;
    const uint32_t F = L.F;
    const uint32_t parts = L.row_parts * L.col_parts;
    const uint64_t pid = uint64_t(rp) * L.col_parts + cp;       // j outer, i inner (sw/benchmark.cpp:142-143)
    const uint64_t header = pid * (1 + F);
    const uint64_t payload_base = uint64_t(parts) * (1 + F);    // spmv_cluster.h:41 / fp :46
    auto fail = [&](const std::string& what) {
        res.ok = false;
        res.error = "channel " + std::to_string(pc) + ", row partition " + std::to_string(rp) + ", column partition " +
                    std::to_string(cp) + ": " + what;
        return res;
    };
    if (header + 1 + F > n_pkts) return fail("partition header lies outside the channel buffer");
    const uint64_t start = buf[header].indices.data[0];          // already multiplied by F (benchmark.cpp:178-179)
    const uint32_t stride = PACK_SIZE * NUM_HBM_CHANNELS * F;    // rows between two rows of one lane stream
    const uint64_t row_base = uint64_t(rp) * L.g->logical_ob;
    const uint64_t row_limit = row_base + L.rows_in_part(rp);
    const uint32_t col_limit = L.cols_in_tile(cp);
    const bool fixed = L.g->impl == IMPL_FIXED;

    for (uint32_t f = 0; f < F; ++f) {
        const PackedWord& lens = buf[header + 1 + f].indices;
        uint32_t longest = 0;
        for (uint32_t k = 0; k < PACK_SIZE; ++k) longest = std::max(longest, lens.data[k]);
        if (longest && payload_base + start + uint64_t(longest - 1) * F + f >= n_pkts)
            return fail("payload runs past the end of the channel buffer");
        LaneEncoder enc[PACK_SIZE];
        const uint32_t vc = pc + f * NUM_HBM_CHANNELS;           // benchmark.cpp:146
        for (uint32_t k = 0; k < PACK_SIZE; ++k) {
            enc[k].cur_row = row_base + uint64_t(vc) * PACK_SIZE + k;  // round 0 of data_formatter.h:410
            enc[k].e = first_e ? first_e[f * PACK_SIZE + k] : 0;
        }
        const MatPkt* pkt = buf + payload_base + start + f;
        for (uint32_t p = 0; p < longest; ++p, pkt += F) {
            for (uint32_t k = 0; k < PACK_SIZE; ++k) {
                if (p >= lens.data[k]) continue;                 // lane exhausted: zero padding
                const uint32_t col = pkt->indices.data[k], val = pkt->vals.data[k];
                if (col == IDX_MARKER) {
                    // fixed: integer part of the Q8.24 word (spmv_cluster.h:82); float: raw bits (fp :104)
                    enc[k].marker(sink, fixed ? (val >> 24) : val, stride);
                } else {
                    if (col >= col_limit) return fail("column index " + std::to_string(col) + " outside the column partition");
                    if (enc[k].cur_row >= row_limit) return fail("decoded row outside the row partition (marker count wrapped?)");
                    enc[k].nonzero(sink, col, val);
                    ++res.nnz;
                }
            }
        }
        for (uint32_t k = 0; k < PACK_SIZE; ++k) {
            const uint64_t begin = first_e ? first_e[f * PACK_SIZE + k] : 0;
            if (enc_len) enc_len[f * PACK_SIZE + k] = uint32_t(enc[k].e - begin);
            if (enc[k].e - begin > 0xffffffffull) return fail("lane stream too long");
        }
    }
    return res;
}

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
    L.streams_per_tile = NUM_HBM_CHANNELS * geom.interleave * PACK_SIZE;
    const uint32_t F = L.F;
    const uint32_t tiles = num_row_partitions * num_col_partitions;
    const uint32_t lanes_per_group = F * PACK_SIZE;  // lane streams of one physical channel in one partition
    const uint64_t header_pkts = uint64_t(tiles) * (1 + F);
    for (uint32_t c = 0; c < NUM_HBM_CHANNELS; ++c) {
        if (!channel[c] && n_packets[c]) { error = "null channel buffer"; return false; }
        if (n_packets[c] < header_pkts) { error = "channel " + std::to_string(c) + " is shorter than its partition headers"; return false; }
    }
    out = StreamTiles();
    out.row_stride = PACK_SIZE * NUM_HBM_CHANNELS * F;
    if (tiles == 0) { out.wg_first.assign(2, 0); out.num_workgroups = 1; return true; }

    // ---- pass 1: encoded length of every lane stream ------------------------------------------
    // enc_len[tile][pc][f*8+k]
    std::vector<uint32_t> enc_len(size_t(tiles) * L.streams_per_tile, 0);
    std::vector<GroupResult> results(size_t(tiles) * NUM_HBM_CHANNELS);
    parallel_for(results.size(), [&](size_t w) {
        const uint32_t tile = uint32_t(w / NUM_HBM_CHANNELS), pc = uint32_t(w % NUM_HBM_CHANNELS);
        const uint32_t cp = tile / num_row_partitions, rp = tile % num_row_partitions;
        CountSink sink;
        results[w] = walk_channel_partition(L, static_cast<const MatPkt*>(channel[pc]), n_packets[pc], pc, rp, cp, nullptr,
                                            &enc_len[size_t(tile) * L.streams_per_tile + size_t(pc) * lanes_per_group], sink);
    });
    for (const auto& r : results) {
        if (!r.ok) { error = r.error; return false; }
        out.nnz += r.nnz;
    }
    // start of every lane stream inside its tile's virtual stream, and the tile totals
    std::vector<uint64_t> first_e(enc_len.size());
    std::vector<uint64_t> tile_elems(tiles), tile_begin(tiles + 1, 0);
    for (uint32_t t = 0; t < tiles; ++t) {
        uint64_t acc = 0;
        for (uint32_t s = 0; s < L.streams_per_tile; ++s) {
            first_e[size_t(t) * L.streams_per_tile + s] = acc;
            acc += enc_len[size_t(t) * L.streams_per_tile + s];
        }
        tile_elems[t] = acc;
        tile_begin[t + 1] = tile_begin[t] + acc;
    }
    const uint64_t total = tile_begin[tiles];
    out.elements = total;

    // ---- plan: split the concatenation of all tiles evenly over the workgroups ------------------
    uint32_t G = std::max<uint32_t>(1, max_workgroups);
    G = uint32_t(std::min<uint64_t>(G, std::max<uint64_t>(1, total / 4096)));  // tiny matrices: fewer, fuller workgroups
    out.num_workgroups = G;
    std::vector<uint64_t> cut(G + 1);
    for (uint32_t g = 0; g <= G; ++g) cut[g] = uint64_t((__uint128_t(total) * g) / G);
    // snap a cut onto a nearby tile boundary: avoids pieces that are all x-tile load and no work
    const uint64_t snap = total / G / 8;
    for (uint32_t g = 1; g < G; ++g) {
        auto it = std::lower_bound(tile_begin.begin(), tile_begin.end(), cut[g]);
        uint64_t best = cut[g], dist = snap + 1;
        if (it != tile_begin.end() && *it - cut[g] < dist) { best = *it; dist = *it - cut[g]; }
        if (it != tile_begin.begin() && cut[g] - *(it - 1) < dist) { best = *(it - 1); }
        cut[g] = std::max(best, cut[g - 1]);
    }
    std::vector<std::vector<TilePiece>> tile_pieces(tiles);
    out.wg_first.assign(G + 1, 0);
    uint64_t image_bytes = 0;
    {
        uint32_t t = 0;
        for (uint32_t g = 0; g < G; ++g) {
            out.wg_first[g] = uint32_t(out.pieces.size());
            uint64_t lo = cut[g], hi = cut[g + 1];
            while (lo < hi) {
                while (tile_begin[t + 1] <= lo) ++t;
                const uint64_t end = std::min(hi, tile_begin[t + 1]);
                const uint64_t n = end - lo;
                uint32_t steps = uint32_t((n + kRunsPerWorkgroup - 1) / kRunsPerWorkgroup);
                steps = (steps + kStepQuantum - 1) / kStepQuantum * kStepQuantum;
                Piece p;
                p.col_tile = t / num_row_partitions;
                p.row_part = t % num_row_partitions;
                p.steps = steps;
                p.reserved = 0;
                p.offset = image_bytes;
                out.pieces.push_back(p);
                tile_pieces[t].push_back(TilePiece{lo - tile_begin[t], end - tile_begin[t], steps, image_bytes});
                image_bytes += (chunk_bytes(steps) * kWavesPerWorkgroup + 255) / 256 * 256;
                lo = end;
            }
        }
        out.wg_first[G] = uint32_t(out.pieces.size());
    }

    // ---- pass 2: fill the image ---------------------------------------------------------------------
    out.image.assign(image_bytes, 0);
    parallel_for(results.size(), [&](size_t w) {
        const uint32_t tile = uint32_t(w / NUM_HBM_CHANNELS), pc = uint32_t(w % NUM_HBM_CHANNELS);
        if (tile_pieces[tile].empty()) return;
        const uint32_t cp = tile / num_row_partitions, rp = tile % num_row_partitions;
        WriteSink sink{out.image.data(), tile_pieces[tile].data(), tile_pieces[tile].size()};
        results[w] = walk_channel_partition(L, static_cast<const MatPkt*>(channel[pc]), n_packets[pc], pc, rp, cp,
                                            &first_e[size_t(tile) * L.streams_per_tile + size_t(pc) * lanes_per_group], nullptr, sink);
    });
    // padding slots at the tail of every piece: special column, no flag, zero value
    parallel_for(tiles, [&](size_t t) {
        for (const TilePiece& p : tile_pieces[t]) {
            const uint64_t n = p.b - p.a, slots = uint64_t(p.steps) * kRunsPerWorkgroup;
            for (uint64_t rel = n; rel < slots; ++rel) {
                uint32_t lane, step;
                uint8_t* chunk = WriteSink::chunk_of(out.image.data(), p, rel, lane, step);
                uint8_t* group = chunk + kChunkHeaderBytes + uint64_t(p.steps) * 8 + uint64_t(step / kStepsPerGroup) * kGroupBytes;
                reinterpret_cast<uint16_t*>(group)[lane * kStepsPerGroup + (step % kStepsPerGroup)] = uint16_t(kSpecialCol);
            }
        }
    });
    return true;
}

}  // namespace dev
}  // namespace hisparse
