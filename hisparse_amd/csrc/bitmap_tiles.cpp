// bitmap_tiles.cpp — CPSR image -> BITMAP rows (stream_tiles.h "BITMAP format"; kernel: spmv_bitmap.hip).
//
// For dense-row matrices the 8 bytes per non-zero the FPGA streams (value + column index, spmv/libfpga/common.h:44-50) are
// mostly index: at 50 % density a bit per column position says the same in 1/16 of the space.  The CPSR image is decoded once
// (tiles_common.h: the same walk as the other formats), the rows are put back into column order, and every (row, 64-column
// group) becomes a 64-bit occupancy mask plus its compacted values.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>

#include "gpu_tiles.h"
#include "hisparse/q8_24.h"
#include "tiles_common.h"

namespace hisparse {
namespace dev {

using namespace detail;

bool build_bitmap_tiles(const Layout& L, const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                        const std::vector<uint32_t>& row_nnz, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                        const CsrView* csr, GpuTiler* gpu, uint64_t image_slack) {
    const uint32_t num_rows = L.num_rows, num_cols = L.num_cols, RP = L.row_parts, CP = L.col_parts;
    const uint32_t G = std::max<uint32_t>(1, max_workgroups);
    // a mask per 64 columns of every row: only sensible for dense rows; when the format is FORCED onto a big sparse matrix
    // (tests) the element streams take over beyond 1 GiB of masks
    if (double(num_rows) * double((num_cols + kBitmapGroupCols - 1) / kBitmapGroupCols) * 8.0 > double(1ull << 30)) {
        error = "bitmap: more than 1 GiB of masks";
        return false;
    }
    auto chan = [&](uint32_t pc) { return static_cast<const MatPkt*>(channel[pc]); };
    PhaseTimer timer;

    // ---- rows back in CSR form: (absolute column, value word) per row ------------------------------------------------------
    // (host builder only: with a GpuTiler the elements stay on the device and every pass over them is a kernel of gpu_tiles.hip;
    // the plan, the block layout and the wavefront runs below are the same code for both, so the two leave the same bytes)
    std::vector<uint64_t> row_ptr(size_t(num_rows) + 1, 0);
    for (uint32_t r = 0; r < num_rows; ++r) row_ptr[r + 1] = row_ptr[r] + row_nnz[r];
    const uint64_t nnz = row_ptr[num_rows];
    const std::unique_ptr<uint64_t[]> elems_buf(new uint64_t[gpu ? 1 : std::max<uint64_t>(nnz, 1)]);   // (not zeroed: every entry is written below)
    uint64_t* const elems = elems_buf.get();               // column << 32 | value word: sorts by column
    if (gpu) {
    } else if (csr) {       // the rows are there already; value words as csr_matrix_convert_from_float gives them (sw/data_loader.h:76-84)
        const bool fixed = L.g->impl == IMPL_FIXED;
        std::atomic<bool> bad_column(false);
        parallel_for((csr->num_rows + 1023) / 1024, [&](size_t chunk) {
            for (uint32_t r = uint32_t(chunk) * 1024; r < std::min<uint64_t>(csr->num_rows, (chunk + 1) * 1024); ++r)
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
        // (a task per packet lane re-reads the packets eight times: only where there are more hardware threads than channel tasks)
        // (HISPARSE_WALK_LANES=0|1 forces, for the tests)
        const char* force_split = env_switch("HISPARSE_WALK_LANES");
        const uint32_t split = (force_split ? std::atoi(force_split) != 0 : std::thread::hardware_concurrency() > 2 * RP * NUM_HBM_CHANNELS) ? PACK_SIZE : 1;
        std::vector<WalkResult> res(size_t(RP) * NUM_HBM_CHANNELS * split);
        // rows of different physical channels and packet lanes are disjoint, and one task takes the column partitions of its rows in
        // ascending order: a row's elements arrive in the order the CSR input had them (sw/data_formatter.h:256-313 keeps it)
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
    // column order inside a row (the reference does not require sorted CSR input); a column that occurs twice in a row cannot be
    // a bit in a mask -> the caller falls back to the element-stream formats
    std::atomic<bool> duplicates(false);
    if (!gpu) parallel_for((num_rows + 1023) / 1024, [&](size_t chunk) {
        for (uint32_t r = uint32_t(chunk) * 1024; r < std::min<uint64_t>(num_rows, (chunk + 1) * 1024); ++r) {
            uint64_t* e = elems + row_ptr[r];
            const uint32_t n = row_nnz[r];
            bool sorted = true;
            for (uint32_t i = 1; i < n && sorted; ++i) sorted = (e[i] >> 32) > (e[i - 1] >> 32);
            if (sorted) continue;
            std::stable_sort(e, e + n, [](uint64_t a, uint64_t b) { return (a >> 32) < (b >> 32); });
            for (uint32_t i = 1; i < n; ++i)
                if ((e[i] >> 32) == (e[i - 1] >> 32)) duplicates = true;
        }
    });
    if (duplicates) { error = "bitmap: duplicate column in a row"; out.format = kFormatPairs; return false; }
    timer.lap("bitmap: rows in column order");

    // ---- plan: column slices only when there are fewer rows than workgroups; row ranges of equal non-zero count --------------
    const uint32_t GR = (num_cols + kBitmapGroupCols - 1) / kBitmapGroupCols;     // groups per row
    uint32_t slices = 1;
    if (const char* force = env_switch("HISPARSE_COL_SLICES")) slices = std::max(1, std::atoi(force));
    else while (slices * 2 <= kMaxColSlices && uint64_t(num_rows) * slices * 2 <= G) slices *= 2;
    slices = std::min<uint32_t>(std::min<uint32_t>(slices, kMaxColSlices), GR);
    while (slices > 1 && uint64_t(slices) * num_rows > 0xffffffffull) slices /= 2;
    uint32_t max_rows = kBitmapMaxBlockRows;
    if (const char* force = env_switch("HISPARSE_MAX_ROWS")) max_rows = std::min<uint32_t>(max_rows, std::max(1, std::atoi(force)));
    const uint64_t per_round = std::max<uint32_t>(1, G / slices);
    const uint64_t rounds = std::max<uint64_t>(1, ((uint64_t(num_rows) + max_rows - 1) / max_rows + per_round - 1) / per_round);
    const uint64_t want_ranges = std::max<uint64_t>(1, std::min<uint64_t>(per_round * rounds, std::max<uint64_t>(1, nnz / 1024)));
    std::vector<RowRange> ranges;
    std::vector<uint64_t> range_nnz;
    build_row_ranges_at_most(L, row_nnz, nnz, want_ranges, max_rows, ranges, range_nnz);
    // Rows without a non-zero at the END of a range (the padding of util_round_csr_matrix_dim: float_stall rounds 512 rows up to 1024) get
    // a range of their own whose wavefront runs are all idle -- the kernel just writes their zeros.  Left in the last real range they made it
    // a 514-row block of whole-row runs, 521 mask steps per EMPTY row: transformer-80 / float_stall ran 1164 us instead of 11 (round 4).
    {
        std::vector<RowRange> cut;
        std::vector<uint64_t> cut_nnz;
        for (size_t i = 0; i < ranges.size(); ++i) {
            const RowRange rg = ranges[i];
            uint32_t live = rg.nrows;
            while (live > 0 && row_nnz[rg.row0 + live - 1] == 0) --live;
            if (live == 0 || live == rg.nrows) { cut.push_back(rg); cut_nnz.push_back(range_nnz[i]); continue; }
            cut.push_back(RowRange{rg.row0, live, rg.row_part, L.part_of_row(rg.row0 + live - 1)});
            cut_nnz.push_back(range_nnz[i]);
            cut.push_back(RowRange{rg.row0 + live, rg.nrows - live, L.part_of_row(rg.row0 + live), rg.last_part});
            cut_nnz.push_back(0);
        }
        ranges.swap(cut);
        range_nnz.swap(cut_nnz);
    }
    const uint32_t NR = uint32_t(ranges.size());
    const uint32_t NB = NR * slices;

    // ---- blocks: layout of masks + values, wavefront runs -------------------------------------------------------------------
    out.nnz = nnz;
    out.elements = nnz;
    out.format = kFormatBitmap;
    out.col_slices = slices;
    out.ring_buffers = 0;
    out.blocks.assign(NB, Block{});
    out.units.assign(size_t(NB) * kBitmapWaves * kBitmapRunSlots, Unit{});
    std::vector<uint64_t> block_nnz(NB, 0), block_base(NB + 1, 0), block_weight(NB, 0);
    // non-zeros of every (row, slice of groups) and of every block (row range x slice)
    std::vector<uint32_t> slice_nnz;          // [row * slices + k]
    if (slices == 1) {
        slice_nnz = row_nnz;
    } else if (gpu) {
        if (!gpu->bitmap_slice_counts(slices, GR, slice_nnz)) { error = gpu->error(); return false; }
    } else {
        slice_nnz.assign(size_t(num_rows) * slices, 0);
        parallel_for((num_rows + 1023) / 1024, [&](size_t chunk) {
            for (uint32_t r = uint32_t(chunk) * 1024; r < std::min<uint64_t>(num_rows, (chunk + 1) * 1024); ++r) {
                const uint64_t* e = elems + row_ptr[r];
                for (uint32_t k = 0; k < slices; ++k) {
                    const uint64_t c0 = uint64_t(k) * GR / slices * kBitmapGroupCols, c1 = uint64_t(k + 1) * GR / slices * kBitmapGroupCols;
                    slice_nnz[size_t(r) * slices + k] = uint32_t(std::lower_bound(e, e + row_nnz[r], c1 << 32) - std::lower_bound(e, e + row_nnz[r], c0 << 32));
                }
            }
        });
    }
    parallel_for(NB, [&](size_t bi) {
        const RowRange& rg = ranges[bi / slices];
        uint64_t n = 0;
        for (uint32_t r = rg.row0; r < rg.row0 + rg.nrows; ++r) n += slice_nnz[size_t(r) * slices + bi % slices];
        block_nnz[bi] = n;
    });
    // A row is cut into `pieces` runs (weighted group counts, below) -- one per wavefront when the block has few rows, one run per
    // row otherwise -- and every run's masks are followed by zero masks up to a multiple of 8 plus two whole batches: the kernel
    // fetches masks eight at a time and issues up to two batches past the end of a run, which then find "no column set".
    auto pieces_of = [](uint32_t nrows) { return nrows * 2 <= kBitmapWaves ? kBitmapWaves / nrows : 1u; };
    auto padded = [](uint32_t steps) { return (steps + 7u) / 8u * 8u + 16u; };
    // The 16 wavefronts of a workgroup do NOT run at the same speed: a SIMD's instruction arbiter serves its oldest wavefront first, so
    // wavefronts 0-3 (the first on each SIMD) finished an equal share after 6.9 us, 4-7 after 8.7, 8-11 after 10.5 and 12-15 after
    // 12.1 us (transformer-50, tools/bitmap_timeline.py, round 3) -- and the first twelve then sat at the barrier.  So the shares are
    // weighted by the wavefront's place on its SIMD (kBitmapSkew; HISPARSE_BITMAP_SKEW=a/b/c/d overrides, 100/100/100/100 = equal
    // shares: 14.3 us against 13.0 on transformer-50, same box, profiles/r03_bitmap_weighted_shares.txt), and the pieces of a row go to
    // wavefronts piece * nrows + row, so that every row gets old and young wavefronts alike.
    uint32_t skew[4] = {kBitmapSkew[0], kBitmapSkew[1], kBitmapSkew[2], kBitmapSkew[3]};
    if (const char* e = env_switch("HISPARSE_BITMAP_SKEW")) {
        unsigned a, b, c, d;
        if (std::sscanf(e, "%u%*[,/ ]%u%*[,/ ]%u%*[,/ ]%u", &a, &b, &c, &d) == 4 && a && b && c && d && a < 10000 && b < 10000 && c < 10000 && d < 10000) { skew[0] = a; skew[1] = b; skew[2] = c; skew[3] = d; }
    }
    auto wave_weight = [&](uint32_t w) { return skew[std::min<uint32_t>(w / 4, 3)]; };
    // first group of piece j of a row (of any row of the block: the weight of a piece is that of the wavefront row 0's piece goes to)
    auto cut = [&](uint32_t GS, uint32_t pieces, uint32_t nrows, uint32_t j) {
        uint64_t before = 0, all = 0;
        for (uint32_t i = 0; i < pieces; ++i) {
            if (i < j) before += wave_weight(i * nrows);
            all += wave_weight(i * nrows);
        }
        // (cutting at whole batches of 8 steps instead was measured too: no difference)
        return j >= pieces ? GS : uint32_t(before * GS / all);
    };
    auto row_stride = [&](uint32_t GS, uint32_t pieces, uint32_t nrows) {
        uint32_t n = 0;
        for (uint32_t j = 0; j < pieces; ++j) n += padded(cut(GS, pieces, nrows, j + 1) - cut(GS, pieces, nrows, j));
        return n;
    };
    for (uint32_t bi = 0; bi < NB; ++bi) {
        const RowRange& rg = ranges[bi / slices];
        const uint32_t k = bi % slices;
        const uint32_t gs0 = uint32_t(uint64_t(k) * GR / slices), gs1 = uint32_t(uint64_t(k + 1) * GR / slices);
        const uint64_t masks = uint64_t(rg.nrows) * row_stride(gs1 - gs0, pieces_of(rg.nrows), rg.nrows);
        // [masks: 8 bytes each][values: 4 bytes each, padded to 8]
        block_base[bi + 1] = block_base[bi] + masks * 8 + ((block_nnz[bi] + 1) & ~uint64_t(1)) * 4;
        block_weight[bi] = block_nnz[bi] ? uint64_t(rg.nrows) * (gs1 - gs0) / kBitmapWaves + 1 : 1;          // wavefront steps (a block of empty rows: none)
        out.max_block_rows = std::max(out.max_block_rows, rg.nrows);
    }
    if (block_base[NB] / 4 >= (1ull << 40)) { error = "matrix too large for the bitmap image"; return false; }
    if (!gpu) resize_zeroed(out.image, block_base[NB]);
    out.image_bytes = block_base[NB];
    timer.lap("bitmap: plan");
    // device builder: what its kernels need of the layout (filled per block below), and every wavefront run's start
    std::vector<GpuTiler::BitmapBlock> dev_blocks(gpu ? NB : 0);
    std::vector<GpuTiler::BitmapRun> dev_runs(gpu ? size_t(NB) * kBitmapWaves : 0);
    std::vector<uint64_t> row_value_base(gpu ? size_t(num_rows) * slices : 0), run_value(gpu ? size_t(NB) * kBitmapWaves : 0);
    if (gpu) {
        uint64_t prefix0 = 0;
        for (uint32_t bi = 0; bi < NB; ++bi) {
            const RowRange& rg = ranges[bi / slices];
            const uint32_t k = bi % slices, gs0 = uint32_t(uint64_t(k) * GR / slices), gs1 = uint32_t(uint64_t(k + 1) * GR / slices);
            const uint32_t pieces = pieces_of(rg.nrows), stride = row_stride(gs1 - gs0, pieces, rg.nrows);
            dev_blocks[bi] = GpuTiler::BitmapBlock{rg.row0, rg.nrows, gs0, gs1 - gs0, pieces, stride, block_base[bi] / 8, prefix0, {}};
            for (uint32_t j = 0; j <= pieces; ++j) dev_blocks[bi].piece_cut[j] = cut(gs1 - gs0, pieces, rg.nrows, j);
            prefix0 += uint64_t(rg.nrows) * stride;
        }
    }

    parallel_for(NB, [&](size_t bi) {
        const RowRange& rg = ranges[bi / slices];
        const uint32_t k = uint32_t(bi % slices);
        const uint32_t gs0 = uint32_t(uint64_t(k) * GR / slices), gs1 = uint32_t(uint64_t(k + 1) * GR / slices), GS = gs1 - gs0;
        const uint64_t c0 = uint64_t(gs0) * kBitmapGroupCols;
        Block& blk = out.blocks[bi];
        blk.row0 = rg.row0;
        blk.nrows = rg.nrows;
        blk.row_part = rg.row_part;
        blk.last_part = rg.last_part;
        blk.flags = 0;
        blk.out_offset = slices > 1 ? k * num_rows + rg.row0 : rg.row0;
        blk.unit_begin = uint32_t(bi * kBitmapWaves * kBitmapRunSlots);
        blk.unit_end = blk.unit_begin + kBitmapWaves * kBitmapRunSlots;
        blk.first_col0 = uint32_t(c0);
        blk.first_ncols = GS;
        const uint32_t pieces = pieces_of(rg.nrows), stride = row_stride(GS, pieces, rg.nrows);
        std::vector<uint32_t> piece_of(GS), piece_at(pieces + 1, 0), piece_cut(pieces + 1);   // group -> piece; piece -> offset inside the row's masks
        for (uint32_t j = 0; j <= pieces; ++j) piece_cut[j] = cut(GS, pieces, rg.nrows, j);
        for (uint32_t j = 0; j < pieces; ++j) {
            for (uint32_t g = piece_cut[j]; g < piece_cut[j + 1]; ++g) piece_of[g] = j;
            piece_at[j + 1] = piece_at[j] + padded(piece_cut[j + 1] - piece_cut[j]);
        }
        auto mask_index = [&](uint32_t lr, uint32_t g) { return uint64_t(lr) * stride + piece_at[piece_of[g]] + (g - piece_cut[piece_of[g]]); };
        uint64_t* mask = gpu ? nullptr : reinterpret_cast<uint64_t*>(out.image.data() + block_base[bi]);
        uint32_t* value = gpu ? nullptr : reinterpret_cast<uint32_t*>(out.image.data() + block_base[bi] + uint64_t(rg.nrows) * stride * 8);
        const uint64_t mask_word0 = block_base[bi] / 8, value_word0 = (block_base[bi] + uint64_t(rg.nrows) * stride * 8) / 4;
        // masks + compacted values in (row, group) order; value_at[r] = values of the block before local row r
        std::vector<uint64_t> value_at(size_t(rg.nrows) + 1, 0);
        for (uint32_t lr = 0; lr < rg.nrows; ++lr) {
            value_at[lr + 1] = value_at[lr] + slice_nnz[size_t(rg.row0 + lr) * slices + k];
            if (gpu) row_value_base[size_t(rg.row0 + lr) * slices + k] = value_word0 + value_at[lr];
        }
        uint64_t at = 0;
        for (uint32_t lr = 0; !gpu && lr < rg.nrows; ++lr) {
            const uint64_t* e = elems + row_ptr[rg.row0 + lr];
            const uint32_t n = row_nnz[rg.row0 + lr];
            const uint64_t* p = std::lower_bound(e, e + n, c0 << 32);
            const uint64_t c1 = uint64_t(gs1) * kBitmapGroupCols;
            for (; p < e + n && (*p >> 32) < c1; ++p) {
                const uint64_t rel = (*p >> 32) - c0;
                mask[mask_index(lr, uint32_t(rel / kBitmapGroupCols))] |= 1ull << (rel % kBitmapGroupCols);
                value[at++] = uint32_t(*p);
            }
        }
        // wavefront runs.  Few rows: every row is cut into floor(16 / nrows) runs of weighted group count.  Many rows: contiguous whole
        // rows per wavefront, balanced by steps-plus-non-zeros.
        // piece: the piece of the row the run is (partial-row runs), or -1.  A piece may hold NO group (a slice of 1 group cut into 2 pieces):
        // its run has no steps, and its masks are the 16 zero ones of its own place in the row, not the next piece's (round 4)
        auto set_seg = [&](uint32_t w, uint32_t r0, uint32_t r1, uint32_t g0, uint32_t g1, int piece = -1) {
            WaveSeg& s = *reinterpret_cast<WaveSeg*>(out.units.data() + blk.unit_begin + size_t(w) * kBitmapRunSlots);
            s.row_begin = r0; s.row_end = r1; s.g_begin = g0; s.g_end = g1;
            uint64_t v = value_word0 + value_at[std::min(r0, rg.nrows)];
            const bool partial = r1 == r0 + 1 && g0 > 0;     // partial row: + the values of the groups in front of g0
            const uint64_t mw = mask_word0 + (r0 >= rg.nrows ? 0 : (piece >= 0 && g1 == g0) ? uint64_t(r0) * stride + piece_at[piece]
                                                                                            : mask_index(r0, r1 == r0 + 1 && g0 < GS ? g0 : 0));
            s.mask_lo = uint32_t(mw); s.mask_hi = uint32_t(mw >> 32);
            const uint32_t steps = r1 > r0 ? g1 - g0 : 0;
            if (gpu) {    // the device looks both up in what it built (run_value + prefix, heads: patched in below)
                dev_runs[bi * kBitmapWaves + w] = GpuTiler::BitmapRun{mw, partial ? dev_blocks[bi].prefix0 + mask_index(r0, g0) : 0, steps, partial ? 1u : 0u};
                run_value[bi * kBitmapWaves + w] = v;
                return;
            }
            if (partial)
                for (uint32_t g = 0; g < g0; ++g) v += uint64_t(__builtin_popcountll(mask[mask_index(r0, g)]));
            s.value_lo = uint32_t(v); s.value_hi = uint32_t(v >> 32);
            // the run's first 32 masks once more, right behind the descriptor (zero beyond the first row-run's end)
            uint64_t* head = reinterpret_cast<uint64_t*>(&s + 1);
            for (uint32_t j = 0; j < kBitmapMaskBatch; ++j)
                head[j] = j < steps ? reinterpret_cast<const uint64_t*>(out.image.data())[mw + j] : 0;
        };
        if (block_nnz[bi] == 0) {
            for (uint32_t w = 0; w < kBitmapWaves; ++w) set_seg(w, rg.nrows, rg.nrows, 0, 0);      // nothing to stream: the block only writes its rows' zeros
        } else if (pieces > 1) {
            for (uint32_t lr = 0; lr < rg.nrows; ++lr)
                for (uint32_t j = 0; j < pieces; ++j) set_seg(j * rg.nrows + lr, lr, lr + 1, piece_cut[j], piece_cut[j + 1], int(j));
            for (uint32_t w = rg.nrows * pieces; w < kBitmapWaves; ++w) set_seg(w, rg.nrows, rg.nrows, 0, 0);      // idle wavefronts
        } else {
            // cost of a row = its steps + its non-zeros / 16 (issue slots vs. bytes); cut the prefix sum into 16 parts by the wavefronts' weights
            std::vector<uint64_t> cost(size_t(rg.nrows) + 1, 0);
            for (uint32_t lr = 0; lr < rg.nrows; ++lr) cost[lr + 1] = cost[lr] + GS + (value_at[lr + 1] - value_at[lr]) / 16;
            uint64_t weight_all = 0, weight_so_far = 0;
            for (uint32_t w = 0; w < kBitmapWaves; ++w) weight_all += wave_weight(w);
            uint32_t r0 = 0;
            for (uint32_t w = 0; w < kBitmapWaves; ++w) {
                weight_so_far += wave_weight(w);
                const uint64_t goal = cost[rg.nrows] * weight_so_far / weight_all;
                uint32_t r1 = uint32_t(std::lower_bound(cost.begin() + r0, cost.end(), goal) - cost.begin());
                r1 = w + 1 == kBitmapWaves ? rg.nrows : std::min(std::max(r1, r0), rg.nrows);
                // a whole-row run must not be mistaken for a partial one: one row = [0, GS) of that row, which is the same thing
                set_seg(w, r0, r1, 0, GS);
                r0 = r1;
            }
        }
    });
    timer.lap("bitmap: emit");

    // ---- float modes: the second image, for the SpMM on the matrix engine (stream_tiles.h: MfmaImage; kernel: spmm_mfma.hip) -------------
    if (L.g->impl != IMPL_FIXED && !env_switch("HISPARSE_NO_MFMA_IMAGE")) {
        MfmaImage& mi = out.mfma;
        mi.tiles = (num_rows + kMfmaTileRows - 1) / kMfmaTileRows;
        mi.groups = GR;
        // wavefront units of `chunk` groups: enough of them for every SIMD of the chip (1024) to get two or three; four units (one
        // workgroup) never straddle two row tiles
        // groups per wavefront unit: the workgroups (4 units each) run in ceil(workgroups / CUs) rounds of `chunk` groups; a unit costs ~1.5
        // groups' worth of set-up and its share of the finish pass; and a CU with only one or two workgroups is short of wavefronts to hide
        // its loads behind.  transformer-50 (32 tiles x 521 groups, k = 16, same box): chunk 4 / 5 / 6 / 7 / 8 / 9 / 10 / 11 / 13 / 17 ->
        // 28.6 / 27.0 / 25.0 / 27.3 / 29.0 / 25.6 / 26.7 / 27.2 / 31.0 / 29.9 us (profiles/r03_spmm_mfma_chunk.txt)
        double best_cost = 1e30;
        for (uint32_t chunk = 4; chunk <= 32; ++chunk) {
            const uint64_t wgs = uint64_t(mi.tiles) * (((GR + chunk - 1) / chunk + 3) / 4);
            const double rounds = double((wgs + G - 1) / G);
            const double cost = rounds * (chunk + 1.5) * (1.0 + 0.7 / rounds);
            if (cost < best_cost) { best_cost = cost; mi.chunk = chunk; }
        }
        if (const char* force = env_switch("HISPARSE_MFMA_CHUNK")) mi.chunk = std::max(1, std::atoi(force));      // experiments
        mi.chunks = ((GR + mi.chunk - 1) / mi.chunk + 3) / 4 * 4;
        const uint64_t mask_words = uint64_t(mi.tiles) * GR * kMfmaTileRows * 2;
        mi.offsets_word = mask_words;
        mi.values_word = mi.offsets_word + uint64_t(mi.tiles) * mi.chunks;
        mi.words_bytes = (mi.values_word + nnz + 64) * 4;
        if ((mi.values_word + nnz + 64) >= (uint64_t(1) << 32)) {
            mi = MfmaImage();          // value offsets are 32-bit words: beyond that the SpMM keeps its other path
        } else if (!gpu) {
            resize_zeroed(mi.words, mi.words_bytes);
            uint32_t* words = reinterpret_cast<uint32_t*>(mi.words.data());
            uint64_t* masks = reinterpret_cast<uint64_t*>(words);
            uint32_t* unit_base = words + mi.offsets_word;
            uint32_t* values = words + mi.values_word;
            // masks: the 16 rows of a tile side by side per group
            parallel_for(num_rows, [&](size_t r) {
                const uint32_t t = uint32_t(r / kMfmaTileRows), i = uint32_t(r % kMfmaTileRows);
                const uint64_t* e = elems + row_ptr[r];
                for (uint32_t k = 0; k < row_nnz[r]; ++k) {
                    const uint64_t col = e[k] >> 32;
                    masks[(uint64_t(t) * GR + col / kBitmapGroupCols) * kMfmaTileRows + i] |= 1ull << (col % kBitmapGroupCols);
                }
            });
            // values in the order the kernel's lanes take them: tile, group, MFMA step s (columns 4 s .. 4 s + 3 of the group), lane
            // l = 16 k + i (row i of the tile, column 4 s + k): the values of one step are one contiguous, coalesced load
            std::vector<uint64_t> tile_base(size_t(mi.tiles) + 1, 0);
            for (uint32_t t = 0; t < mi.tiles; ++t) {
                uint64_t n = 0;
                for (uint32_t r = t * kMfmaTileRows; r < std::min(num_rows, (t + 1) * kMfmaTileRows); ++r) n += row_nnz[r];
                tile_base[t + 1] = tile_base[t] + n;
            }
            parallel_for(mi.tiles, [&](size_t t) {
                uint32_t cursor[kMfmaTileRows] = {0};            // next element of every row of the tile (rows are in column order)
                uint64_t at = tile_base[t];
                for (uint32_t g = 0; g < GR; ++g) {
                    if (g % mi.chunk == 0) unit_base[t * mi.chunks + g / mi.chunk] = uint32_t(at);
                    const uint64_t* m = masks + (uint64_t(t) * GR + g) * kMfmaTileRows;
                    uint32_t first[kMfmaTileRows];               // the row's first element of this group
                    for (uint32_t i = 0; i < kMfmaTileRows; ++i) first[i] = cursor[i];
                    for (uint32_t s4 = 0; s4 < 16; ++s4)
                        for (uint32_t k = 0; k < 4; ++k)
                            for (uint32_t i = 0; i < kMfmaTileRows; ++i) {
                                const uint32_t p = 4 * s4 + k;
                                if (!((m[i] >> p) & 1ull)) continue;
                                const uint32_t r = uint32_t(t) * kMfmaTileRows + i;
                                const uint32_t rank = uint32_t(__builtin_popcountll(m[i] & ((1ull << p) - 1ull)));
                                values[at++] = uint32_t(elems[row_ptr[r] + first[i] + rank]);
                            }
                    for (uint32_t i = 0; i < kMfmaTileRows; ++i) cursor[i] += uint32_t(__builtin_popcountll(m[i]));
                }
                for (uint32_t c = (GR + mi.chunk - 1) / mi.chunk; c < mi.chunks; ++c) unit_base[t * mi.chunks + c] = uint32_t(at);   // units past the last group: empty
            });
        }
        timer.lap("bitmap: second image (matrix engine SpMM)");
    }
    if (gpu) {      // masks, values, run heads and the second image: kernels (gpu_tiles.hip); the descriptors above are completed from what they return
        std::vector<uint32_t> range_of_row(num_rows), run_prefix;
        for (uint32_t b = 0; b < NR; ++b) std::fill(range_of_row.begin() + ranges[b].row0, range_of_row.begin() + ranges[b].row0 + ranges[b].nrows, b);
        std::vector<uint64_t> run_heads;
        bool dup = false;
        if (!gpu->bitmap_emit(slices, GR, range_of_row, dev_blocks, row_value_base, out.image_bytes, image_slack, dev_runs, run_prefix, run_heads,
                              out.mfma.words_bytes ? &out.mfma : nullptr, dup)) {
            error = gpu->error();
            return false;
        }
        if (dup) { error = "bitmap: duplicate column in a row"; out.format = kFormatPairs; out.mfma = MfmaImage(); return false; }
        for (size_t r = 0; r < dev_runs.size(); ++r) {
            WaveSeg& s = *reinterpret_cast<WaveSeg*>(out.units.data() + r * kBitmapRunSlots);
            const uint64_t v = run_value[r] + run_prefix[r];
            s.value_lo = uint32_t(v); s.value_hi = uint32_t(v >> 32);
            std::copy(run_heads.begin() + r * kBitmapMaskBatch, run_heads.begin() + (r + 1) * kBitmapMaskBatch, reinterpret_cast<uint64_t*>(&s + 1));
        }
        out.d_image = gpu->release_image();
        out.mfma.d_words = gpu->release_mfma();
        timer.lap("bitmap: device emit");
    }

    // x in LDS (spmv_bitmap.hip, kXLds): when a block's stretch of x and its accumulators fit the LDS together and the stretch is used by
    // more than one row (HISPARSE_BITMAP_X_LDS=0|1 forces)
    {
        uint32_t widest = 0;
        for (uint32_t k = 0; k < slices; ++k) widest = std::max(widest, uint32_t(uint64_t(k + 1) * GR / slices) - uint32_t(uint64_t(k) * GR / slices));
        const uint64_t acc_bytes = ((uint64_t(out.max_block_rows) + 1) * kAccumulatorBytes + 15u) & ~uint64_t(15);
        const bool fits = widest <= kBitmapMaxXLdsGroups && acc_bytes + uint64_t(widest) * kBitmapGroupCols * 4 <= kMaxLdsBytes;
        bool want = fits && uint64_t(num_rows) >= 2ull * NR;
        if (const char* force = env_switch("HISPARSE_BITMAP_X_LDS")) want = fits && std::atoi(force) != 0;
        out.bitmap_x_groups = want ? widest : 0u;
    }
    std::vector<std::vector<uint32_t>> mine;
    assign_workgroups(out, block_weight, G, RP, mine);
    chain_blocks(out, mine, RP);
    // the kernel finds the runs of block i at units[16 i ..] without reading the block first: store them in final block order
    constexpr uint32_t kPerBlock = kBitmapWaves * kBitmapRunSlots;
    std::vector<Unit> moved(out.units.size());
    for (uint32_t i = 0; i < NB; ++i) {
        std::copy(out.units.begin() + out.blocks[i].unit_begin, out.units.begin() + out.blocks[i].unit_end, moved.begin() + size_t(i) * kPerBlock);
        out.blocks[i].unit_begin = i * kPerBlock;
        out.blocks[i].unit_end = (i + 1) * kPerBlock;
    }
    out.units.swap(moved);
    return true;
}

}  // namespace dev
}  // namespace hisparse
