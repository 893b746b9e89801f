// tiles_common.h — pieces shared by the load-time builders (stream_tiles.cpp: PAIRS / DELTA element streams,
// bitmap_tiles.cpp: BITMAP rows): the CPSR walk, the thread pool, workgroup assignment and block chaining.
//
// The decode follows the reference's loader in MEANING (header layout, per-lane lengths, marker = row
// advance, interleaved virtual channels):
//   spmv/libfpga/spmv_cluster.h:41-98        fixed point, INTERLEAVE_FACTOR 1
//   spmv-fp/libfpga/spmv_cluster.h:46-117    float, INTERLEAVE_FACTOR 1 or 8
// with the row <-> (channel, lane, round) mapping of sw/data_formatter.h:410,432.
#ifndef HISPARSE_TILES_COMMON_H_
#define HISPARSE_TILES_COMMON_H_

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "hisparse/worker_pool.h"
#include "stream_tiles.h"

namespace hisparse {
namespace dev {
namespace detail {

// Tuning switches.  The configuration surface of the library is hs_set_option(ctx, key, value) (hisparse_hip.h): a context's options are
// in force while ITS matrix is planned and built.  The scope is per THREAD (OptionScope, taken by hs_load_matrix* on the calling thread)
// and handed to the worker threads task by task (parallel_for below sets it around every task it runs): loads of different contexts -- on
// different devices, from different host threads -- neither wait for each other nor see each other's options, and a caller outside any
// scope (tiles_capi builders, tools) sees the environment only.  (Until round 4 this was one process-wide pointer under a process-wide
// mutex held for the whole load.)  The environment variable HISPARSE_<KEY> remains as the fallback for tools and tests; set-but-empty
// counts as not set (HISPARSE_MAX_ROWS= used to mean "one row per block").  A context's map must not change while its load runs (calls
// on one context never overlap: hisparse_hip.h).
using OptionMap = std::map<std::string, std::string>;
inline const OptionMap*& active_options() {
    static thread_local const OptionMap* active = nullptr;
    return active;
}
struct OptionScope {
    const OptionMap* const saved;
    explicit OptionScope(const OptionMap* options) : saved(active_options()) { active_options() = options; }
    ~OptionScope() { active_options() = saved; }
    OptionScope(const OptionScope&) = delete;
    OptionScope& operator=(const OptionScope&) = delete;
};
// name: the environment spelling, "HISPARSE_<KEY>"
inline const char* option_lookup(const OptionMap* options, const char* name) {
    if (options) {
        const auto it = options->find(name);
        if (it != options->end()) return it->second.empty() ? nullptr : it->second.c_str();
    }
    const char* v = std::getenv(name);
    return v && *v ? v : nullptr;
}
inline const char* env_switch(const char* name) { return option_lookup(active_options(), name); }

struct PhaseTimer {   // HISPARSE_PLAN_DEBUG=1: wall time of the load-time passes
    const bool on = env_switch("HISPARSE_PLAN_DEBUG") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (on) std::fprintf(stderr, "re-tile %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

template <typename Fn>
void parallel_for(size_t n, Fn fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const OptionMap* const options = active_options();      // the caller's scope travels with every task
    auto scoped = [&fn, options](size_t i) {
        const OptionScope scope(options);
        fn(i);
    };
    hisparse::pooled_for(n, hw ? hw : 1u, scoped);
}

// v = n zero bytes, filled (and first touched) by many threads
inline void resize_zeroed(ImageBytes& v, size_t n) {
    v.clear();
    v.resize(n);
    constexpr size_t kPiece = size_t(2) << 20;
    uint8_t* p = v.data();
    parallel_for((n + kPiece - 1) / kPiece, [&](size_t i) { std::memset(p + i * kPiece, 0, std::min(kPiece, n - i * kPiece)); });
}

struct Layout {
    const Geometry* g;
    uint32_t num_rows, num_cols, row_parts, col_parts, F;
    // Row ranges may CROSS row-partition borders (round 5).  The reference's partitions are the size of its on-chip output banks
    // (131 072 rows in float_pob: 19 of them on ogbn-products) -- a bank the GPU does not have; cutting every range at their borders cost
    // 16-50 % more (row range x sub-tile) units and uneven rounds of blocks.  hs_run_partition still runs one partition at a time: the
    // blocks that INTERSECT it (Block::row_part <= p <= Block::last_part), results kept for the partition's rows only (hs_api.cpp).
    // HISPARSE_CROSS_PARTITIONS=0: ranges end at the borders again.
    bool cross_parts = true;
    uint32_t part_of_row(uint32_t row) const { return uint32_t(row / g->logical_ob); }
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
                                  Visit visit, int only_lane = -1) {
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
                if (only_lane >= 0 && k != uint32_t(only_lane)) continue;   // one task per lane stream (rows r % 8 == k are disjoint)
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
    uint32_t n = 0;         // real elements
    uint64_t scratch = 0;   // index of the unit's first element in the position-sorted scratch list
    // PAIRS
    uint32_t chunks = 0;    // ceil(n / 64)
    uint32_t base = 0;      // chunk counter of the block at the unit's first chunk
    uint32_t start_step[kConsumerWaves];   // chunk index of the unit's first chunk in wavefront w's stream
    // DELTA
    uint64_t slots = 0;     // elements + bridge slots
    uint32_t run_len[kConsumerWaves];      // slots per lane of wavefront w in this unit
    uint64_t first_slot[kConsumerWaves];   // first slot of lane 0 of wavefront w
    uint32_t lane_stride[kConsumerWaves];  // DELTA: slots from the run of lane l to the run of lane l + 1 of wavefront w
    uint32_t start_record[kConsumerWaves]; // record index of the unit's head record in wavefront w's stream
    // OWNER
    uint32_t own_begin[kConsumerWaves + 1]; // wavefront w owns sorted elements [own_begin[w], own_begin[w + 1]) of the unit
    uint32_t own_row[kConsumerWaves] = {};  // local row of the share's first element (0 for an empty share): OWNER24's row_base ...
    uint32_t own_last[kConsumerWaves] = {}; // ... and of its last one
};

struct RowRange { uint32_t row0, nrows, row_part, last_part; };      // row_part / last_part: row partition of the first / last row

// OWNER: cut one unit's n sorted elements into the 14 wavefronts' shares (stream_tiles.h).  Rows may only change hands BETWEEN units
// (the unit barrier orders the accumulator writes), so the cut is made per unit, for equal work: the unit's ceil(n / 64) chunks are
// dealt as evenly as they go (every wavefront ceil or floor of chunks / 14 steps), a share ends on a row boundary at or below its
// 64 x steps elements (a row longer than that stays whole), what it leaves goes to the next wavefront.  Against fixed row ownership
// (a wavefront's rows for the whole block, cut at equal non-zero count over the block) the slowest wavefront of a hyper-sparse unit
// does 5 steps instead of 5.4 on average on ogbn-products and the image carries 64 chunks per unit instead of 70.
// max_span (OWNER24: kOwnerShareRows): no share spans more than max_span rows -- its position words carry the row relative to the
// share's first row in 11 bits.  A share that would is cut at that row; and a share ends no earlier than the row from which the
// remaining wavefronts can still cover the rest of the unit, max_span rows each (possible whenever the unit's rows span at most
// 14 x max_span, which the LDS row cap guarantees; the builder re-checks every share).
// row_of(i) = local row of sorted element i.  Same code on the host and in gpu_tiles.hip.
template <typename RowOf>
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void balanced_owner_shares(uint32_t n, RowOf row_of, uint32_t own_begin[kConsumerWaves + 1], uint32_t max_span = 0xffffffffu) {
    uint32_t begin = 0;
    const uint32_t last_row = n ? row_of(n - 1) : 0u;
    auto first_at_or_above = [&](uint32_t lo, uint32_t want) {      // first index in [lo, n) whose row is >= want (rows never decrease)
        uint32_t hi = n;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (row_of(mid) < want) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    for (uint32_t w = 0; w < kConsumerWaves; ++w) {
        own_begin[w] = begin;
        const uint32_t left = n - begin, waves_left = kConsumerWaves - w;
        const uint32_t chunks_left = (left + kWaveLanes - 1) / kWaveLanes;
        const uint32_t steps = (chunks_left + waves_left - 1) / waves_left;
        uint32_t end = begin + (left < steps * kWaveLanes ? left : steps * kWaveLanes);
        if (waves_left == 1) end = n;
        while (end > begin && end < n && row_of(end) == row_of(end - 1)) --end;        // back to a row boundary
        if (end == begin && left) {                                                    // one row longer than the share: take it whole
            end = begin + (left < steps * kWaveLanes ? left : steps * kWaveLanes);
            while (end < n && row_of(end) == row_of(end - 1)) ++end;
        }
        if (max_span != 0xffffffffu && begin < n) {
            const uint32_t row0 = row_of(begin);
            if (end > begin && row_of(end - 1) - row0 >= max_span) end = first_at_or_above(begin, row0 + max_span);
            if (end < n) {
                const uint64_t room = uint64_t(waves_left - 1) * max_span;          // rows the remaining wavefronts can cover
                if (uint64_t(last_row) - row_of(end) + 1 > room) end = first_at_or_above(end, uint32_t(uint64_t(last_row) + 1 - room));
            }
        }
        begin = end;
    }
    own_begin[kConsumerWaves] = n;
}

// Workgroups: longest-processing-time assignment of blocks, row partition by row partition (a launch of hs_run_partition
// runs ONE of them and wants it spread over all workgroups), heaviest block first, each to the workgroup with the least
// work in this partition -- ties to the one with the least work overall, so that the partitions' leftovers do not pile up
// on the same workgroups.  Group i (i-th heaviest first block) becomes workgroup (i % 8) * groups/8 + i / 8: the kernels run
// logical workgroups [x * groups/8, (x+1) * groups/8) on XCD x, so every XCD -- its L2 and its share of the fabric to HBM --
// gets the same mix of heavy and light blocks.  Sets out.num_workgroups; mine[g] = blocks of workgroup g in execution order.
inline void assign_workgroups(StreamTiles& out, const std::vector<uint64_t>& block_weight, uint32_t max_workgroups, uint32_t row_parts,
                              std::vector<std::vector<uint32_t>>& mine) {
    const uint32_t NB = uint32_t(out.blocks.size());
    const uint32_t groups = std::min<uint32_t>(std::max<uint32_t>(1, max_workgroups), std::max<uint32_t>(1, NB));
    out.num_workgroups = groups;
    mine.assign(groups, {});
    std::vector<uint64_t> load(groups, 0), part_load(groups, 0);
    std::vector<std::vector<uint32_t>> by_rank(groups);
    // blocks that cross partition borders: ONE balance over the whole SpMV (what hs_run launches), every workgroup's blocks then in
    // row order, which is partition order; hs_run_partition finds the blocks of a partition spread as they fall
    bool crossing = false;
    for (uint32_t b = 0; b < NB; ++b) crossing = crossing || out.blocks[b].last_part != out.blocks[b].row_part;
    if (crossing) row_parts = 1;
    for (uint32_t rp = 0; rp < row_parts; ++rp) {
        std::vector<uint32_t> order;
        for (uint32_t b = 0; b < NB; ++b)
            if (crossing || out.blocks[b].row_part == rp) order.push_back(b);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return block_weight[a] > block_weight[b]; });
        std::fill(part_load.begin(), part_load.end(), 0);
        for (uint32_t b : order) {
            uint32_t best = 0;
            for (uint32_t g = 1; g < groups; ++g)
                if (part_load[g] < part_load[best] || (part_load[g] == part_load[best] && load[g] < load[best])) best = g;
            by_rank[best].push_back(b);
            part_load[best] += block_weight[b] + 16;   // every block also costs a fixed prologue/epilogue (in steps)
            load[best] += block_weight[b] + 16;
        }
    }
    for (uint32_t i = 0; i < groups; ++i) {
        const uint32_t g = groups % 8 == 0 ? (i % 8) * (groups / 8) + i / 8 : i;
        if (crossing) std::stable_sort(by_rank[i].begin(), by_rank[i].end(), [&](uint32_t a, uint32_t b) { return out.blocks[a].row0 < out.blocks[b].row0; });
        mine[g].swap(by_rank[i]);
    }
}

// Column-sliced matrices whose x does not fit an XCD's L2 (4 MiB; ogbn-products: 9.8 MB of x, 5 slices): the same rule INSIDE every
// XCD, but which XCD gets which blocks is decided by column slice.  The blocks, listed slice by slice, are cut into 8 stretches of
// equal count: an XCD then works on one slice, at most two, and its 32 workgroups pull the same 1/slices of x through the one L2 at
// about the same time -- every sub-tile is fetched from HBM once per XCD instead of missing in every workgroup that asks for it
// (with blocks of all slices on every XCD 16 % of the x refills missed L2, and a refill that waits for HBM holds its whole unit up).
// `slice_of_block[b]`: column slice of block b.  Needs a multiple of 8 workgroups (the kernels' workgroup -> XCD rule); the caller
// falls back to assign_workgroups otherwise.
inline void assign_workgroups_by_slice(StreamTiles& out, const std::vector<uint64_t>& block_weight, uint32_t max_workgroups, uint32_t row_parts,
                                       const std::vector<uint32_t>& slice_of_block, std::vector<std::vector<uint32_t>>& mine) {
    const uint32_t NB = uint32_t(out.blocks.size());
    const uint32_t groups = max_workgroups, per_xcd = groups / 8;
    out.num_workgroups = groups;
    mine.assign(groups, {});
    std::vector<uint32_t> order(NB);
    for (uint32_t b = 0; b < NB; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return slice_of_block[a] < slice_of_block[b]; });
    // inside an XCD: heaviest block first, each to the workgroup with the least work so far (ties: the fewest blocks).  A stretch of the
    // slice-by-slice list holds the row partitions in very unequal numbers, so balancing partition by partition (assign_workgroups)
    // would leave some workgroups with three blocks and others with one; the whole SpMV is what is balanced here, and every
    // workgroup's blocks are then put in row-partition order (chain_blocks and hs_run_partition rely on that order only).
    std::vector<uint64_t> load(groups, 0);
    for (uint32_t x = 0; x < 8; ++x) {
        const uint32_t lo = uint32_t(uint64_t(NB) * x / 8), hi = uint32_t(uint64_t(NB) * (x + 1) / 8);
        std::vector<uint32_t> todo(order.begin() + lo, order.begin() + hi);
        std::stable_sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) { return block_weight[a] > block_weight[b]; });
        for (uint32_t b : todo) {
            uint32_t best = x * per_xcd;
            for (uint32_t g = best + 1; g < (x + 1) * per_xcd; ++g)
                if (load[g] < load[best] || (load[g] == load[best] && mine[g].size() < mine[best].size())) best = g;
            mine[best].push_back(b);
            load[best] += block_weight[b] + 16;
        }
        for (uint32_t g = x * per_xcd; g < (x + 1) * per_xcd; ++g)
            std::stable_sort(mine[g].begin(), mine[g].end(), [&](uint32_t a, uint32_t b) { return out.blocks[a].row0 < out.blocks[b].row0; });      // row order = partition order
    }
    (void)row_parts;
}

// Final block order: the first block of workgroup g sits at blocks[g] (ONE dependent load before the kernel's first
// stream load), further blocks of a workgroup are chained through Block::next.  Fills wg_first / block_order / part_heads.
inline void chain_blocks(StreamTiles& out, const std::vector<std::vector<uint32_t>>& mine, uint32_t row_parts) {
    const uint32_t groups = out.num_workgroups, NB = uint32_t(out.blocks.size());
    std::vector<uint32_t> new_index(NB, 0);
    uint32_t tail = 0;
    for (uint32_t g = 0; g < groups; ++g) tail += mine[g].empty() ? 0u : 1u;   // == groups unless NB == 0
    uint32_t heads = 0;
    for (uint32_t g = 0; g < groups; ++g)
        for (size_t k = 0; k < mine[g].size(); ++k) new_index[mine[g][k]] = k == 0 ? heads++ : tail++;
    std::vector<Block> moved(NB);
    out.wg_first.assign(groups + 1, 0);
    out.block_order.clear();
    out.part_heads.assign(size_t(row_parts) * groups, kNoBlock);
    for (uint32_t g = 0; g < groups; ++g) {
        out.wg_first[g] = uint32_t(out.block_order.size());
        for (size_t k = 0; k < mine[g].size(); ++k) {
            Block blk = out.blocks[mine[g][k]];
            // the chain is in partition order (blocks inside partitions: sorted by row_part; crossing blocks: sorted by row0).  A run of
            // partition p starts at the first block that reaches p (part_heads) and goes on while the next block begins in p or before
            // (the kernels test next_part <= p: several column slices of one crossing row range may follow each other)
            const bool last_of_part = k + 1 == mine[g].size() || out.blocks[mine[g][k + 1]].row_part > blk.last_part;
            blk.next = k + 1 < mine[g].size() ? new_index[mine[g][k + 1]] : 0u;
            blk.next_part = k + 1 < mine[g].size() ? out.blocks[mine[g][k + 1]].row_part : 0xffffffffu;
            if (last_of_part) blk.flags |= kBlockLastOfPartition;
            for (uint32_t p = blk.row_part; p <= blk.last_part && p < row_parts; ++p)
                if (blk.nrows && out.part_heads[size_t(p) * groups + g] == kNoBlock) out.part_heads[size_t(p) * groups + g] = new_index[mine[g][k]];
            moved[new_index[mine[g][k]]] = blk;
            out.block_order.push_back(new_index[mine[g][k]]);
        }
    }
    out.wg_first[groups] = uint32_t(out.block_order.size());
    out.blocks.swap(moved);
}

// Row ranges of roughly `target` non-zeros each, at most max_rows rows, never across a row partition.
inline void build_row_ranges(const Layout& L, const std::vector<uint32_t>& row_nnz, uint64_t target, uint32_t max_rows,
                             std::vector<RowRange>& ranges, std::vector<uint64_t>& range_nnz) {
    const uint32_t stretches = L.cross_parts ? 1u : L.row_parts;      // crossing: the rows are ONE stretch
    for (uint32_t rp = 0; rp < stretches; ++rp) {
        const uint32_t lo = L.cross_parts ? 0u : uint32_t(uint64_t(rp) * L.g->logical_ob), hi = L.cross_parts ? L.num_rows : lo + L.rows_in_part(rp);
        uint32_t r0 = lo;
        uint64_t acc = 0;
        for (uint32_t r = lo; r < hi; ++r) {
            // close the range BEFORE a row that would overshoot the target by more than the range undershoots now
            const uint64_t with = acc + row_nnz[r];
            if (r > r0 && (r - r0 == max_rows || (with > target && with - target > target - std::min(acc, target)))) {
                ranges.push_back(RowRange{r0, r - r0, L.part_of_row(r0), L.part_of_row(r - 1)});
                range_nnz.push_back(acc);
                r0 = r;
                acc = 0;
            }
            acc += row_nnz[r];
        }
        if (hi > r0) {
            ranges.push_back(RowRange{r0, hi - r0, L.part_of_row(r0), L.part_of_row(hi - 1)});
            range_nnz.push_back(acc);
        }
    }
}

// At most `want` ranges if the row cap allows it: the greedy cut above can leave one small extra range per row partition,
// and one range too many means one workgroup with two blocks -- twice the kernel time.  The target is raised in small steps
// until the count fits (or the cap on rows per range makes that impossible).
// round_to (0: off): when the row cap forces MORE ranges than wanted -- float_pob's 131 072-row partitions under a 24 561-row block cap:
// 19 x 6 = 114 ranges on ogbn-products where 2 x 51 were wanted -- the count is raised to the next multiple of round_to (the ranges one
// round of workgroups takes), so that every workgroup gets the same number of (smaller) blocks instead of a third of them one block more:
// 570 blocks over 256 workgroups ran 335.7 us, the round-3 figure of the same matrix in float_stall's 3 partitions 206.8 (round 4).
// ranges the row cap alone forces
inline uint64_t ranges_by_cap(const Layout& L, uint32_t max_rows) {
    if (L.cross_parts) return (uint64_t(L.num_rows) + max_rows - 1) / max_rows;
    uint64_t by_cap = 0;
    for (uint32_t rp = 0; rp < L.row_parts; ++rp) by_cap += (uint64_t(L.rows_in_part(rp)) + max_rows - 1) / max_rows;
    return by_cap;
}

inline void build_row_ranges_at_most(const Layout& L, const std::vector<uint32_t>& row_nnz, uint64_t nnz, uint64_t want, uint32_t max_rows,
                                     std::vector<RowRange>& ranges, std::vector<uint64_t>& range_nnz, uint64_t round_to = 0) {
    if (round_to) {
        const uint64_t by_cap = ranges_by_cap(L, max_rows);
        if (by_cap > want) want = (by_cap + round_to - 1) / round_to * round_to;
    }
    uint64_t target = std::max<uint64_t>(1, (nnz + want - 1) / want);
    for (int attempt = 0; attempt < 64; ++attempt) {
        ranges.clear();
        range_nnz.clear();
        build_row_ranges(L, row_nnz, target, max_rows, ranges, range_nnz);
        if (ranges.size() <= want) return;
        if (ranges_by_cap(L, max_rows) > want) return;      // the row cap alone forces more
        target += std::max<uint64_t>(1, target / 128);
    }
}

}  // namespace detail

// bitmap_tiles.cpp: the BITMAP builder (called by build_stream_tiles once the format is chosen; `row_nnz` from its pass 0)
bool build_bitmap_tiles(const detail::Layout& L, const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                        const std::vector<uint32_t>& row_nnz, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                        const CsrView* csr = nullptr, class GpuTiler* gpu = nullptr, uint64_t image_slack = 0);

// sweep_tiles.cpp: the SWEEP plan (column slices, row ranges wanted, rows per block) and its modelled cost in microseconds
double sweep_plan(const detail::Layout& L, uint64_t nnz, uint32_t max_workgroups, uint32_t& slices, uint64_t& want_ranges, uint32_t& max_rows);
// sweep_tiles.cpp: the SWEEP builder (called by build_stream_tiles once the format is chosen; `row_nnz` from its pass 0; gpu: the per-non-zero
// passes on the device)
bool build_sweep_tiles(const detail::Layout& L, const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                       const std::vector<uint32_t>& row_nnz, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                       const CsrView* csr = nullptr, class GpuTiler* gpu = nullptr, uint64_t image_slack = 0);

}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_TILES_COMMON_H_
