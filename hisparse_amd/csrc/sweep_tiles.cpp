// sweep_tiles.cpp — CPSR image (or CSR rows) -> SWEEP blocks (stream_tiles.h "SWEEP format"; kernel: spmv_sweep.hip).
//
// Hyper-sparse matrices (pokec: 19 non-zeros per row over 1.6 M columns) pay in the row-block kernel for every (row range x 8192-column
// sub-tile) unit -- a flush, a barrier and a 32 KiB refill of x for ~2 000 elements.  SWEEP drops the units: the
// vector stays in L2 and the elements of a block come in COLUMN order, so that the 64 lanes of one gather touch a handful of 128-byte
// lines and the workgroup moves over its slice of x once, left to right (measured at block level first: tools/gather_bench.hip,
// profiles/r04_gather_bench.txt).  What replaces the FPGA's column partitions (spmv/libfpga/vecbuf_access_unit.h:66-72: the vector buffer
// holds one partition of x at a time) is the column slice; what replaces its output buffer (pe.h:121-135) is the block's LDS accumulators.
// The CPSR image is decoded once (tiles_common.h: the same walk as the other formats), the rows are put back into column order, every
// block's elements are sorted by (column, row) and cut into chunks of 64.  With a GpuTiler (gpu_tiles.h) the three things that touch every
// non-zero -- the per-line counts behind the slice boundaries, the sort, the emit -- are kernels and the image never exists on the host; the
// planning and the layout below are the same code for both, so the two leave the same bytes (tests/test_gpu_retile.py).  Two cases go to the
// host loops even then: duplicate (row, column) entries (the host orders them by value word) and blocks whose columns lie so far apart that a
// chunk must be cut short (more than 65535 columns inside 64 consecutive elements).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>

#include "gpu_tiles.h"
#include "hisparse/q8_24.h"
#include "tiles_common.h"

namespace hisparse {
namespace dev {

using namespace detail;

namespace {

// cost of a block in ns on one CU (tools/gather_bench.hip, then fitted to pokec and ogbn-products: 73 / 210 us of kernel): 8 bytes per
// element at ~24 GB/s per CU, a gathered 128-byte line of x every ~3.3 clocks, ~10 us per block for launch ramp, prologue, epilogue and the
// spread between blocks; the combine pass as measured on 5- and 7-slice plans (9.4 / 8.5 us)
// Round 4, profiling builds on the real matrices (profiles/r04_sweep_ablations.txt): the gathers cost pokec 23 us with 20 K-row blocks and 11 us
// with 40 K-row blocks (per line), ogbn-products 38 us either way -- a floor of ~12 clocks per gather instruction, 0.085 ns per element; and
// every slice is another rows x 4 bytes of partial sums written by the kernel and read back by the combine pass (~4 TB/s + ~8 TB/s).
// Round 6: fitted again on the kernel of round 5 (ring depth 4, the epilogue, `sc1` streams for images that stay in the Infinity Cache), whole steps
// of nine matrices the round-4 constants had never seen (tools/planner_check.py, profiles/r06_planner_check_before.txt; pokec, one rank's slabs of
// hollywood / ogbn-products): the round-4 model (0.33 ns per element + max(1.36 ns per line, 0.085 ns per element) + 10 us per block) read 13 % high
// on pokec and 40-75 % high on everything smaller -- an R-MAT graph of 24 M non-zeros modelled at 60 us and measured at 42.8 went to OWNER24 (58.7 us)
// because of it.  Now, per block: 0.27 ns per element where the image stays in the cache (<= kResidentMaxImageBytes), 0.38 where it is streamed
// from HBM every time, + 0.66 ns per 128-byte line of x in the block's slice, + 4.2 us; the combine pass as before.  Residuals on the nine: -7 ... +4 %
// in fixed point, float modes up to 15 % slower than modelled.
constexpr double kSweepNsPerElementResident = 0.27, kSweepNsPerElementStreamed = 0.38, kSweepNsPerLine = 0.66, kSweepBlockUs = 4.2;
// the weights the slice borders are cut by (equal modelled cost per slice): the round-4 pair, kept -- the borders of the measured images do not move
constexpr double kSweepCutNsPerElement = 0.33, kSweepCutNsPerLine = 1.36;

struct Placed { uint64_t key; uint32_t value; };      // key = column << 16 | local row

}  // namespace

// The plan: column slices x row ranges at the lowest modelled cost (microseconds; the model's terms are in the comment above).  Also what
// build_stream_tiles compares with its estimate for OWNER24 when it chooses between the two formats.
double sweep_plan(const Layout& L, uint64_t nnz, uint32_t max_workgroups, uint32_t& slices, uint64_t& want_ranges, uint32_t& max_rows) {
    const uint32_t num_rows = L.num_rows, num_cols = L.num_cols;
    const uint32_t G = std::max<uint32_t>(1, max_workgroups);
    max_rows = L.g->impl == IMPL_FIXED ? kSweepMaxBlockRowsFixed : kSweepMaxBlockRowsFloat;
    // an image planned for the four-vector SpMM kernel (spmm_sweep.hip): four sets of accumulators per block, so a quarter of the rows
    if (const char* v = env_switch("HISPARSE_SPMM_VECTORS")) if (std::atoi(v) == 4) max_rows = std::max(1u, max_rows / 4u - 1u);
    if (const char* force = env_switch("HISPARSE_MAX_ROWS")) max_rows = std::min<uint32_t>(max_rows, std::max(1, std::atoi(force)));
    const uint64_t by_cap = detail::ranges_by_cap(L, max_rows);
    const uint32_t lines = (num_cols + kSweepColAlign - 1) / kSweepColAlign;
    const char* force_slices = env_switch("HISPARSE_COL_SLICES");
    double best = 1e30;
    slices = 1;
    want_ranges = 1;
    for (uint32_t cs = 1; cs <= std::min<uint32_t>(kMaxSweepSlices, std::max<uint32_t>(1, lines)); ++cs) {
        if (force_slices && uint32_t(std::atoi(force_slices)) != cs) continue;
        if (uint64_t(cs) * num_rows > 0xffffffffull) continue;      // Block::out_offset is a 32-bit word offset
        const uint64_t per_round = std::max<uint32_t>(1, G / cs);
        const uint64_t rounds = std::max<uint64_t>(1, (by_cap + per_round - 1) / per_round);
        const uint64_t ranges = std::min<uint64_t>(per_round * rounds, std::max<uint64_t>(by_cap, std::max<uint64_t>(1, nnz / 4096)));
        const double blocks = double(ranges) * cs, blocks_per_wg = std::ceil(blocks / G);
        // more than eight slices (round 5): measured on one rank's slab of an 8-way split (profiles/r05_sweep_16_slices.txt) -- ogbn-products
        // (60 K elements per block) 46.2 -> 43.6 us in 16 slices, pokec (15 K per block) 22.2 -> 23.3: with so little in a block the x lines
        // it saves are not what the block waits for, and every slice is another set of partial rows.  So only where a block holds >= 32 K.
        if (cs > kMaxColSlices && !force_slices && double(nnz) / blocks < 32768.0) continue;
        const double per_element = double(nnz) * 8.1 <= double(kResidentMaxImageBytes) ? kSweepNsPerElementResident : kSweepNsPerElementStreamed;
        const double block_ns = double(nnz) / blocks * per_element + double(lines) / cs * kSweepNsPerLine;
        const double combine_us = double(num_rows) * 4.0 * cs / 4e6 + (cs > 1 ? 2.0 + double(num_rows) * 4.0 * (cs + 1) / 8e6 : 0.0);
        const double cost = blocks_per_wg * (block_ns * 1e-3 + kSweepBlockUs) + combine_us;
        if (env_switch("HISPARSE_PLAN_DEBUG")) std::fprintf(stderr, "sweep plan cs %u: ranges %llu block %.1f us combine %.1f => %.1f us\n", cs, (unsigned long long)ranges, block_ns * 1e-3, combine_us, cost);
        if (cost < best) { best = cost; slices = cs; want_ranges = ranges; }
    }
    return best;
}

bool build_sweep_tiles(const Layout& L, const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                       const std::vector<uint32_t>& row_nnz, uint32_t max_workgroups, StreamTiles& out, std::string& error, const CsrView* csr,
                       GpuTiler* gpu, uint64_t image_slack) {
    const uint32_t num_rows = L.num_rows, num_cols = L.num_cols, RP = L.row_parts, CP = L.col_parts;
    const uint32_t G = std::max<uint32_t>(1, max_workgroups);
    auto chan = [&](uint32_t pc) { return static_cast<const MatPkt*>(channel[pc]); };
    if (uint64_t(num_cols) * 4 >= (1ull << 32)) { error = "sweep: x does not fit a 32-bit byte offset"; return false; }
    PhaseTimer timer;

    // ---- rows back in CSR form: (absolute column, value word) per row, in column order (host builder only) ------------------------
    std::vector<uint64_t> row_ptr(size_t(num_rows) + 1, 0);
    for (uint32_t r = 0; r < num_rows; ++r) row_ptr[r + 1] = row_ptr[r] + row_nnz[r];
    const uint64_t nnz = row_ptr[num_rows];
    std::unique_ptr<uint64_t[]> elems_buf;
    uint64_t* elems = nullptr;                             // column << 32 | value word
    auto rows_on_the_host = [&]() -> bool {
    if (elems) return true;
    elems_buf.reset(new uint64_t[std::max<uint64_t>(nnz, 1)]);
    elems = elems_buf.get();
    if (csr) {              // value words as csr_matrix_convert_from_float gives them (sw/data_loader.h:76-84)
        const bool fixed = L.g->impl == IMPL_FIXED;
        std::atomic<bool> bad_column(false);
        parallel_for((csr->num_rows + 1023) / 1024, [&](size_t piece) {
            for (uint32_t r = uint32_t(piece) * 1024; r < std::min<uint64_t>(csr->num_rows, (piece + 1) * 1024); ++r)
                for (uint64_t e = csr->indptr[r], o = row_ptr[r]; e < csr->indptr[r + 1]; ++e, ++o) {
                    if (csr->indices[e] >= csr->num_cols) bad_column = true;
                    uint32_t word;
                    if (fixed) word = q8_24_raw_from_double(double(csr->values[e]));
                    else std::memcpy(&word, &csr->values[e], 4);
                    elems[o] = (uint64_t(csr->indices[e]) << 32) | word;
                }
        });
        if (bad_column) { error = "CSR column index outside the matrix"; return false; }
    } else {
        std::vector<uint32_t> cursor(num_rows, 0);
        const uint32_t split = std::thread::hardware_concurrency() > 2 * RP * NUM_HBM_CHANNELS ? PACK_SIZE : 1;   // a task per packet lane where threads are plenty
        std::vector<WalkResult> res(size_t(RP) * NUM_HBM_CHANNELS * split);
        parallel_for(res.size(), [&](size_t w) {
            const uint32_t lane = uint32_t(w % split), pc = uint32_t(w / split % NUM_HBM_CHANNELS), rp = uint32_t(w / split / NUM_HBM_CHANNELS);
            for (uint32_t cp = 0; cp < CP && res[w].ok; ++cp) {
                const uint64_t col_base = uint64_t(cp) * L.g->logical_vb;
                WalkResult r = walk_channel_partition(L, chan(pc), n_packets[pc], pc, rp, cp, [&](uint32_t row, uint32_t col, uint32_t val) {
                    elems[row_ptr[row] + cursor[row]++] = ((col_base + col) << 32) | val;
                }, split > 1 ? int(lane) : -1);
                if (!r.ok) res[w] = r;
            }
        });
        for (const auto& r : res)
            if (!r.ok) { error = r.error; return false; }
    }
    parallel_for((num_rows + 1023) / 1024, [&](size_t piece) {      // (the reference does not require sorted CSR input; duplicates are legal here)
        for (uint32_t r = uint32_t(piece) * 1024; r < std::min<uint64_t>(num_rows, (piece + 1) * 1024); ++r) {
            uint64_t* e = elems + row_ptr[r];
            if (!std::is_sorted(e, e + row_nnz[r])) std::sort(e, e + row_nnz[r]);
        }
    });
    timer.lap("sweep: rows in column order");
    return true;
    };
    if (!gpu && !rows_on_the_host()) return false;

    // ---- plan: row ranges x contiguous column slices ----------------------------------------------------------------------------
    // Every block gathers each 128-byte line of its slice of x about once, so the lines through the chip are (row ranges x |x| / 128)
    // whatever the slice count; slices exist to fill the CUs when the LDS row cap allows fewer row ranges than there are workgroups,
    // at the price of the combine pass (same model as stream_tiles.cpp).
    uint32_t max_rows = 0, slices = 1;
    uint64_t want_ranges = 1;
    const uint32_t lines = (num_cols + kSweepColAlign - 1) / kSweepColAlign;
    if (sweep_plan(L, nnz, max_workgroups, slices, want_ranges, max_rows) >= 1e30) { error = "sweep: no plan (HISPARSE_COL_SLICES out of range?)"; return false; }
    std::vector<RowRange> ranges;
    std::vector<uint64_t> range_nnz;
    build_row_ranges_at_most(L, row_nnz, nnz, want_ranges, max_rows, ranges, range_nnz, std::max<uint32_t>(1, G / slices));
    const uint32_t NR = uint32_t(ranges.size()), NB = NR * slices;

    // slice boundaries (the same for every row range), on 128-byte lines of x, at equal block cost: elements + lines
    std::vector<uint32_t> slice_col(slices + 1, 0);
    {
        std::vector<uint64_t> line_nnz(lines, 0);
        if (gpu) {
            if (!gpu->sweep_line_counts(lines, line_nnz)) { error = gpu->error(); return false; }
        } else {
        std::mutex merge;
        const size_t pieces = std::max<size_t>(1, std::min<size_t>(64, num_rows / 1024));
        parallel_for(pieces, [&](size_t piece) {
            std::vector<uint32_t> mine(lines, 0);
            const uint64_t lo = row_ptr[uint64_t(num_rows) * piece / pieces], hi = row_ptr[uint64_t(num_rows) * (piece + 1) / pieces];
            for (uint64_t e = lo; e < hi; ++e) mine[uint32_t(elems[e] >> 32) / kSweepColAlign]++;
            std::lock_guard<std::mutex> lock(merge);
            for (uint32_t l = 0; l < lines; ++l) line_nnz[l] += mine[l];
        });
        }
        std::vector<double> upto(lines + 1, 0.0);
        for (uint32_t l = 0; l < lines; ++l) upto[l + 1] = upto[l] + double(line_nnz[l]) / std::max<uint32_t>(1, NR) * kSweepCutNsPerElement + kSweepCutNsPerLine;
        uint32_t line_before = 0;
        for (uint32_t k = 1; k < slices; ++k) {
            const uint32_t l = uint32_t(std::lower_bound(upto.begin(), upto.end(), upto[lines] * k / slices) - upto.begin());
            line_before = std::max(line_before, std::min(l, lines));
            slice_col[k] = uint32_t(std::min<uint64_t>(uint64_t(line_before) * kSweepColAlign, num_cols));
        }
        slice_col[slices] = num_cols;
    }
    timer.lap("sweep: plan");

    // ---- blocks: elements sorted by (column, row), cut into chunks ---------------------------------------------------------------
    out.nnz = nnz;
    out.format = kFormatSweep;
    out.col_slices = slices;
    out.ring_buffers = 0;
    out.light = false;
    out.blocks.assign(NB, Block{});
    out.units.assign(1, Unit{});             // (no units; one descriptor so that the table is never empty)
    out.max_block_rows = 0;
    for (const RowRange& rg : ranges) out.max_block_rows = std::max(out.max_block_rows, rg.nrows);
    std::vector<std::vector<Placed>> sorted(NB);
    std::vector<std::vector<uint32_t>> chunk_first(NB);      // index of every chunk's first element (+ the end)
    std::vector<uint64_t> block_start;                       // device builder: block bi = sorted elements [block_start[bi], block_start[bi + 1])
    if (gpu && NB <= 65536) {
        std::vector<uint32_t> range_of_row(num_rows), range_row0(NR);
        for (uint32_t b = 0; b < NR; ++b) {
            range_row0[b] = ranges[b].row0;
            std::fill(range_of_row.begin() + ranges[b].row0, range_of_row.begin() + ranges[b].row0 + ranges[b].nrows, b);
        }
        bool unsupported = false;
        if (!gpu->sweep_sort(range_of_row, range_row0, slice_col, NB, block_start, unsupported)) { error = gpu->error(); return false; }
        if (unsupported) gpu = nullptr;
        timer.lap("sweep: gpu sort");
    } else {
        gpu = nullptr;
    }
    if (!gpu) {
    if (!rows_on_the_host()) return false;
    parallel_for(NB, [&](size_t bi) {
        const RowRange& rg = ranges[bi / slices];
        const uint64_t c0 = slice_col[bi % slices], c1 = slice_col[bi % slices + 1];
        std::vector<Placed>& mine = sorted[bi];
        for (uint32_t r = 0; r < rg.nrows; ++r) {
            const uint64_t* e = elems + row_ptr[rg.row0 + r];
            const uint32_t n = row_nnz[rg.row0 + r];
            const uint64_t* lo = slices == 1 ? e : std::lower_bound(e, e + n, c0 << 32);
            const uint64_t* hi = slices == 1 ? e + n : std::lower_bound(lo, e + n, c1 << 32);
            for (const uint64_t* p = lo; p < hi; ++p) mine.push_back(Placed{(*p >> 32) << 16 | r, uint32_t(*p)});
        }
        // (column, row) order: the elements were collected row by row, each row in (column, value) order, so a STABLE sort by column is
        // enough -- least-significant-digit radix passes of 11 bits over column - c0 (two passes up to 4 M columns per slice; std::sort took
        // 195 of the 323 ms pokec's load spent here)
        {
            uint32_t bits = 1;
            while (bits < 32 && ((c1 - c0 - 1) >> bits) != 0) ++bits;
            std::vector<Placed> other(mine.size());
            std::vector<uint32_t> count(1u << 11);
            for (uint32_t shift = 0; shift < bits && c1 > c0; shift += 11) {
                std::fill(count.begin(), count.end(), 0u);
                for (const Placed& e : mine) count[(((e.key >> 16) - c0) >> shift) & 2047u]++;
                uint32_t at = 0;
                for (uint32_t& c : count) { const uint32_t n = c; c = at; at += n; }
                for (const Placed& e : mine) other[count[(((e.key >> 16) - c0) >> shift) & 2047u]++] = e;
                mine.swap(other);
            }
        }
        // a chunk: up to 64 elements whose columns lie within 65535 of the first one's
        std::vector<uint32_t>& first = chunk_first[bi];
        for (uint32_t i = 0; i < mine.size();) {
            first.push_back(i);
            const uint64_t base = mine[i].key >> 16;
            uint32_t j = i + 1;
            while (j < mine.size() && j - i < kWaveLanes && (mine[j].key >> 16) - base <= 0xffffu) ++j;
            i = j;
        }
        first.push_back(uint32_t(mine.size()));
    });
    timer.lap("sweep: sort blocks");
    }

    std::vector<uint64_t> block_weight(NB, 0), stream_at(NB, 0), table_at(NB, 0);
    uint64_t stream_bytes = 0, table_bytes = 0;
    for (uint32_t bi = 0; bi < NB; ++bi) {
        const uint64_t chunks = gpu ? (block_start[bi + 1] - block_start[bi] + kWaveLanes - 1) / kWaveLanes : chunk_first[bi].size() - 1;
        const uint32_t steps = uint32_t(std::min<uint64_t>((chunks + kSweepWaves - 1) / kSweepWaves, 0xffffffffu));
        if (uint64_t(steps) * kSweepWaves * kChunkBytes >= (1ull << 32)) { error = "sweep: a block's stream exceeds 4 GiB"; return false; }
        const RowRange& rg = ranges[bi / slices];
        Block& blk = out.blocks[bi];
        blk.row0 = rg.row0;
        blk.nrows = rg.nrows;
        blk.row_part = rg.row_part;
        blk.last_part = rg.last_part;
        blk.flags = 0;
        blk.out_offset = slices > 1 ? (bi % slices) * num_rows + rg.row0 : rg.row0;
        blk.total_steps[0] = steps;
        blk.first_col0 = slice_col[bi % slices];
        blk.first_ncols = slice_col[bi % slices + 1] - slice_col[bi % slices];
        stream_at[bi] = stream_bytes;
        stream_bytes += uint64_t(steps) * kSweepWaves * kChunkBytes;
        table_at[bi] = table_bytes;
        table_bytes += uint64_t(steps) * kSweepWaves * 4;
        block_weight[bi] = steps;
        out.elements += uint64_t(steps) * kSweepWaves * kWaveLanes;
    }
    for (uint32_t bi = 0; bi < NB; ++bi) {
        out.blocks[bi].wave_offset[0] = stream_at[bi];
        out.blocks[bi].wave_offset[1] = stream_bytes + table_at[bi];
    }
    out.image_bytes = stream_bytes + table_bytes;
    out.sweep_table_bytes = table_bytes;
    if (gpu) {
        std::vector<GpuTiler::SweepBlock> layout(NB);
        uint64_t chunk0 = 0;
        for (uint32_t bi = 0; bi < NB; ++bi) {
            GpuTiler::SweepBlock& b = layout[bi];
            b.first = block_start[bi];
            b.count = uint32_t(block_start[bi + 1] - block_start[bi]);
            b.steps = out.blocks[bi].total_steps[0];
            b.nrows = out.blocks[bi].nrows;
            b.pad_col = out.blocks[bi].first_ncols ? out.blocks[bi].first_col0 : 0u;
            b.stream_at = stream_at[bi];
            b.table_at = stream_bytes + table_at[bi];
            b.chunk0 = chunk0;
            chunk0 += uint64_t(b.steps) * kSweepWaves;
        }
        if (!gpu->sweep_emit(layout, out.image_bytes, image_slack)) { error = gpu->error(); return false; }
        out.d_image = gpu->release_image();
        timer.lap("sweep: gpu emit");
    } else {
    resize_zeroed(out.image, out.image_bytes);
    parallel_for(NB, [&](size_t bi) {
        const std::vector<Placed>& mine = sorted[bi];
        const std::vector<uint32_t>& first = chunk_first[bi];
        const uint32_t chunks = uint32_t(first.size()) - 1, steps = out.blocks[bi].total_steps[0], nrows = out.blocks[bi].nrows;
        uint32_t* stream = reinterpret_cast<uint32_t*>(out.image.data() + stream_at[bi]);
        uint32_t* table = reinterpret_cast<uint32_t*>(out.image.data() + stream_bytes + table_at[bi]);
        const uint32_t pad_col = out.blocks[bi].first_ncols ? out.blocks[bi].first_col0 : 0u;     // (a column that exists: padding gathers x too)
        for (uint32_t k = 0; k < steps * kSweepWaves; ++k) {
            const uint32_t s = k / kSweepWaves, w = k % kSweepWaves;
            const uint32_t lo = k < chunks ? first[k] : 0u, hi = k < chunks ? first[k + 1] : 0u;
            const uint32_t base = hi > lo ? uint32_t(mine[lo].key >> 16) : (chunks ? uint32_t(mine.back().key >> 16) : pad_col);
            table[size_t(w) * steps + s] = base;
            uint32_t* chunk = stream + size_t(k) * (kChunkBytes / 4);
            for (uint32_t l = 0; l < kWaveLanes; ++l) {
                if (lo + l < hi) {
                    const Placed& p = mine[lo + l];
                    chunk[2 * l] = p.value;
                    chunk[2 * l + 1] = uint32_t(p.key & 0xffffu) << 16 | uint32_t((p.key >> 16) - base);
                } else {
                    chunk[2 * l] = 0;
                    chunk[2 * l + 1] = nrows << 16;
                }
            }
        }
    });
    sorted.clear();
    timer.lap("sweep: write image");
    }

    std::vector<std::vector<uint32_t>> mine;
    bool by_slice = slices > 1 && G % 8 == 0 && RP == 1;
    if (const char* force = env_switch("HISPARSE_XCD_AFFINITY")) by_slice = by_slice && std::atoi(force) != 0;
    if (by_slice) {
        // An XCD works on one slice of x, two at most, so that its L2 (4 MiB) holds what its workgroups gather: pokec's x is 6.5 MB, and with
        // the blocks of all slices on every XCD the gathers that miss L2 show up as 80 MB of fabric traffic per SpMV.  The rule needs a block
        // for every workgroup (the kernels map workgroup -> XCD by index): a plan of 255 blocks gets an idle one (no rows, no steps).
        std::vector<uint32_t> slice_of_block(NB);
        for (uint32_t bi = 0; bi < NB; ++bi) slice_of_block[bi] = bi % slices;
        while (out.blocks.size() < G) {
            Block idle{};
            idle.row_part = 0;
            idle.wave_offset[1] = stream_bytes;
            slice_of_block.push_back(uint32_t(out.blocks.size()) % slices);      // (the index the idle block is about to get)
            out.blocks.push_back(idle);
            block_weight.push_back(0);
        }
        assign_workgroups_by_slice(out, block_weight, G, RP, slice_of_block, mine);
    } else {
        assign_workgroups(out, block_weight, G, RP, mine);
    }
    chain_blocks(out, mine, RP);
    timer.lap("sweep: workgroups");
    return true;
}

}  // namespace dev
}  // namespace hisparse
