// gpu_tiles.hip — GpuTiler: the per-non-zero passes of the load-time re-tiling on gfx950 (see gpu_tiles.h).
#include "gpu_tiles.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>

namespace hisparse {
namespace dev {

using detail::Layout;
using detail::UnitPlan;

namespace {

constexpr uint32_t kPosBits = 28;                       // position = local_row * 8192 + local_col < 2^28 (rows per block <= 24561)
constexpr uint64_t kPosMask = (1ull << kPosBits) - 1;
constexpr uint32_t kErrColumn = 1, kErrRow = 2;

// The 8 lane streams of one virtual channel pc + 16 f in partition (rp, cp) -- what one cluster's 8 PE loader FIFOs see.
struct StreamGroup {
    uint64_t byte_off;      // first packet of the group inside the uploaded channel block
    uint32_t stride;        // bytes between the group's packets (INTERLEAVE_FACTOR * 64)
    uint32_t len[PACK_SIZE];// elements including markers, per lane
    uint32_t row;           // absolute row of lane 0 in round 0 (sw/data_formatter.h:410); lane k: + k
    uint32_t row_stride;    // rows between two rows of one lane stream: PACK_SIZE * channels * F
    uint32_t row_limit;     // first row past the row partition
    uint32_t col_limit;     // columns in the column partition
    uint32_t cp;            // column partition
    uint32_t fixed;         // marker count = value word >> 24 (fixed point, spmv_cluster.h:82) or the raw word (float, fp :104)
    uint32_t first;         // first segment slot of the group
    uint32_t nseg;          // segments per lane: ceil(longest lane / kSegment)
};

// A lane stream is decoded in SEGMENTS of kSegment consecutive entries: the running row index is a prefix sum of the in-band marker
// counts, so a first kernel sums the markers of every segment, one device scan turns the sums into every segment's starting row, and
// all later passes start anywhere.  Thread t of a pass = (group, segment j, lane k), k fastest: the 8 lanes of a packet are 8 adjacent
// threads (32 contiguous bytes of indices, 32 of values).  Scan slot of that thread = first + k * nseg + j (lane-major, so that the
// prefix inside one lane stream is a contiguous piece of the scan).
constexpr uint32_t kSegment = 64;

struct Segment {
    const uint8_t* p;       // first entry's index word
    uint32_t n;             // entries
    uint32_t stride;
    uint32_t slot;          // scan slot
    uint32_t lane_slot;     // scan slot of the lane stream's first segment
    uint32_t group;
    uint32_t lane;
};

__device__ __forceinline__ bool find_segment(const uint8_t* __restrict__ channels, const StreamGroup* __restrict__ groups, uint32_t num_groups,
                                             uint32_t total_slots, uint32_t t, Segment& s) {
    if (t >= total_slots) return false;
    uint32_t lo = 0, hi = num_groups;                 // last group with first <= t
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (groups[mid].first <= t) lo = mid; else hi = mid;
    }
    const StreamGroup& g = groups[lo];
    const uint32_t local = t - g.first, j = local / PACK_SIZE, k = local % PACK_SIZE;
    const uint32_t len = g.len[k], begin = j * kSegment;
    s.group = lo;
    s.lane = k;
    s.lane_slot = g.first + k * g.nseg;
    s.slot = s.lane_slot + j;
    s.stride = g.stride;
    s.n = begin < len ? min(kSegment, len - begin) : 0u;
    s.p = channels + g.byte_off + uint64_t(begin) * g.stride + k * 4u;
    return true;
}

// markers and non-zeros of every segment
__global__ __launch_bounds__(256) void segment_sums_kernel(const uint8_t* __restrict__ channels, const StreamGroup* __restrict__ groups, uint32_t num_groups,
                                                          uint32_t total_slots, uint64_t* __restrict__ advance, uint64_t* __restrict__ count) {
    Segment s;
    if (!find_segment(channels, groups, num_groups, total_slots, blockIdx.x * blockDim.x + threadIdx.x, s)) return;
    const bool fixed = groups[s.group].fixed;
    uint64_t adv = 0;
    uint32_t cnt = 0;
    const uint8_t* p = s.p;
    for (uint32_t i = 0; i < s.n; ++i, p += s.stride) {
        const uint32_t col = *reinterpret_cast<const uint32_t*>(p);
        if (col == IDX_MARKER) {
            const uint32_t val = *reinterpret_cast<const uint32_t*>(p + 32);
            adv += fixed ? (val >> 24) : val;
        } else {
            ++cnt;
        }
    }
    advance[s.slot] = adv;
    count[s.slot] = cnt;
}

// Walk one segment: visit(absolute row, partition-local column, value word) for every non-zero.  Returns 0 or an error code.
template <typename Visit>
__device__ __forceinline__ uint32_t walk_segment(const StreamGroup& g, const Segment& s, const uint64_t* __restrict__ advance, Visit visit) {
    // rows advanced before this segment (exclusive scan, relative to the lane stream's first segment); anything beyond 2^32 rounds is
    // outside every partition and must not wrap the product
    const uint64_t rounds = min(advance[s.slot] - advance[s.lane_slot], uint64_t(1) << 32);
    uint64_t row = uint64_t(g.row) + s.lane + rounds * g.row_stride;
    const uint8_t* p = s.p;
    for (uint32_t i = 0; i < s.n; ++i, p += s.stride) {
        const uint32_t col = *reinterpret_cast<const uint32_t*>(p);
        const uint32_t val = *reinterpret_cast<const uint32_t*>(p + 32);
        if (col == IDX_MARKER) {
            row += uint64_t(g.fixed ? (val >> 24) : val) * g.row_stride;
        } else {
            if (col >= g.col_limit) return kErrColumn;
            if (row >= g.row_limit) return kErrRow;
            visit(uint32_t(row), col, val);
        }
    }
    return 0;
}

__device__ __forceinline__ void report(uint32_t* scalar, uint32_t code, uint32_t group, uint32_t lane) {
    if (code && atomicCAS(scalar, 0u, code) == 0u) { scalar[1] = group; scalar[2] = lane; }
}

__global__ __launch_bounds__(256) void count_rows_kernel(const uint8_t* __restrict__ channels, const StreamGroup* __restrict__ groups, uint32_t num_groups,
                                                        uint32_t total_slots, const uint64_t* __restrict__ advance, uint32_t* __restrict__ row_nnz,
                                                        uint32_t* scalar) {
    Segment s;
    if (!find_segment(channels, groups, num_groups, total_slots, blockIdx.x * blockDim.x + threadIdx.x, s)) return;
    // rows only go up inside a stream: one atomic per run of equal rows
    uint32_t cur = 0xffffffffu, run = 0;
    const uint32_t err = walk_segment(groups[s.group], s, advance, [&](uint32_t row, uint32_t, uint32_t) {
        if (row != cur) {
            if (run) atomicAdd(row_nnz + cur, run);
            cur = row;
            run = 0;
        }
        ++run;
    });
    if (run) atomicAdd(row_nnz + cur, run);
    report(scalar, err, s.group, s.lane);
}

__global__ __launch_bounds__(256) void count_tiles_kernel(const uint8_t* __restrict__ channels, const StreamGroup* __restrict__ groups, uint32_t num_groups,
                                                         uint32_t total_slots, const uint64_t* __restrict__ advance,
                                                         const uint32_t* __restrict__ block_of_row, uint32_t tiles, uint32_t S, uint32_t sub_width,
                                                         uint32_t* __restrict__ cnt) {
    Segment s;
    if (!find_segment(channels, groups, num_groups, total_slots, blockIdx.x * blockDim.x + threadIdx.x, s)) return;
    const StreamGroup& g = groups[s.group];
    const uint32_t cp = g.cp;
    // count per (current row range, sub-tile) in registers, one atomic per range change
    uint32_t cur = 0xffffffffu, local[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto flush = [&]() {
        if (cur == 0xffffffffu) return;
        for (uint32_t k = 0; k < 8; ++k)
            if (local[k]) { atomicAdd(cnt + size_t(cur) * tiles + cp * S + k, local[k]); local[k] = 0; }
    };
    walk_segment(g, s, advance, [&](uint32_t row, uint32_t col, uint32_t) {
        const uint32_t b = block_of_row[row], k = col / sub_width;
        if (S > 8) { atomicAdd(cnt + size_t(b) * tiles + cp * S + k, 1u); return; }
        if (b != cur) { flush(); cur = b; }
        local[k]++;
    });
    flush();
}

__global__ __launch_bounds__(256) void keys_kernel(const uint8_t* __restrict__ channels, const StreamGroup* __restrict__ groups, uint32_t num_groups,
                                                  uint32_t total_slots, const uint64_t* __restrict__ advance, const uint64_t* __restrict__ base,
                                                  const uint32_t* __restrict__ block_of_row, const uint32_t* __restrict__ range_row0,
                                                  const uint32_t* __restrict__ unit_of, uint32_t tiles, uint32_t S, uint32_t sub_width,
                                                  uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    Segment s;
    if (!find_segment(channels, groups, num_groups, total_slots, blockIdx.x * blockDim.x + threadIdx.x, s)) return;
    const StreamGroup& g = groups[s.group];
    const uint32_t cp = g.cp;
    uint64_t at = base[s.slot];
    walk_segment(g, s, advance, [&](uint32_t row, uint32_t col, uint32_t val) {
        const uint32_t b = block_of_row[row], k = col / sub_width;
        const uint64_t unit = unit_of[size_t(b) * tiles + cp * S + k];
        const uint64_t pos = uint64_t(row - range_row0[b]) * kSubTileCols + (col - k * sub_width);
        keys[at] = (unit << kPosBits) | pos;
        vals[at] = val;
        ++at;
    });
}

// ---- CSR source ---------------------------------------------------------------------------------------------------------------
// One thread per kCsrSegment consecutive non-zeros: the row of the first one by binary search in indptr, then forward.
constexpr uint32_t kCsrSegment = 16;

struct CsrCursor {
    uint32_t row;
    uint64_t e, end;
};
__device__ __forceinline__ bool csr_segment(const uint32_t* __restrict__ indptr, uint32_t num_rows, uint64_t nnz, uint64_t t, CsrCursor& c) {
    c.e = t * kCsrSegment;
    if (c.e >= nnz) return false;
    c.end = min(c.e + kCsrSegment, nnz);
    uint32_t lo = 0, hi = num_rows;              // last row with indptr[row] <= e
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (indptr[mid] <= c.e) lo = mid; else hi = mid;
    }
    c.row = lo;
    return true;
}

// value word of a CSR float: csr_matrix_convert_from_float (sw/data_loader.h:76-84) = the float's bits, or the Q8.24 conversion of
// include/hisparse/q8_24.h (negatives, zeros and NaN -> 0; round half up; saturate) -- all exact in double, so bit-identical to the host
__device__ __forceinline__ uint32_t value_word(float v, bool fixed) {
    if (!fixed) return __float_as_uint(v);
    const double d = double(v);
    if (!(d > 0.0)) return 0u;
    const double scaled = floor(d * 16777216.0 + 0.5);
    return scaled >= 4294967296.0 ? 0xffffffffu : uint32_t(scaled);
}

__global__ __launch_bounds__(256) void csr_count_tiles_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ indices, uint32_t num_rows,
                                                             uint64_t nnz, uint32_t num_cols, uint32_t logical_vb, const uint32_t* __restrict__ block_of_row,
                                                             uint32_t tiles, uint32_t S, uint32_t sub_width, uint32_t* __restrict__ cnt, uint32_t* scalar) {
    CsrCursor c;
    if (!csr_segment(indptr, num_rows, nnz, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, c)) return;
    size_t cur = ~size_t(0);
    uint32_t run = 0;
    for (; c.e < c.end; ++c.e) {
        while (c.e >= indptr[c.row + 1]) ++c.row;
        const uint32_t col = indices[c.e];
        if (col >= num_cols) { report(scalar, kErrColumn, c.row / PACK_SIZE, c.row % PACK_SIZE); return; }
        const uint32_t cp = col / logical_vb, k = (col - cp * logical_vb) / sub_width;
        const size_t at = size_t(block_of_row[c.row]) * tiles + cp * S + k;
        if (at != cur) {
            if (run) atomicAdd(cnt + cur, run);
            cur = at;
            run = 0;
        }
        ++run;
    }
    if (run) atomicAdd(cnt + cur, run);
}

__global__ __launch_bounds__(256) void csr_keys_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ indices, const float* __restrict__ values,
                                                      uint32_t num_rows, uint64_t nnz, uint32_t logical_vb, uint32_t fixed,
                                                      const uint32_t* __restrict__ block_of_row, const uint32_t* __restrict__ range_row0,
                                                      const uint32_t* __restrict__ unit_of, uint32_t tiles, uint32_t S, uint32_t sub_width,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    CsrCursor c;
    if (!csr_segment(indptr, num_rows, nnz, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, c)) return;
    for (; c.e < c.end; ++c.e) {
        while (c.e >= indptr[c.row + 1]) ++c.row;
        const uint32_t col = indices[c.e], cp = col / logical_vb, local = col - cp * logical_vb, k = local / sub_width;
        const uint32_t b = block_of_row[c.row];
        const uint64_t unit = unit_of[size_t(b) * tiles + cp * S + k];
        const uint64_t pos = uint64_t(c.row - range_row0[b]) * kSubTileCols + (local - k * sub_width);
        keys[c.e] = (unit << kPosBits) | pos;
        vals[c.e] = value_word(values[c.e], fixed != 0);
    }
}

__global__ __launch_bounds__(256) void duplicates_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* flag) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i + 1 < n && keys[i] == keys[i + 1]) *flag = 1;
}

// DELTA: bridge slots in front of element i (same rule as the host: gaps beyond 65534 advance in steps of 65535)
__global__ __launch_bounds__(256) void bridges_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint64_t* __restrict__ bridges) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b = 0;
    if (i > 0 && (keys[i] >> kPosBits) == (keys[i - 1] >> kPosBits)) {
        const uint64_t d = (keys[i] & kPosMask) - (keys[i - 1] & kPosMask);
        if (d > kMaxGap) b = (d - kMaxGap + kBridgeAdvance - 1) / kBridgeAdvance;
    }
    bridges[i] = b;
}

struct DevicePlan {        // what the emit kernels need of a UnitPlan + its block
    uint64_t start;        // first element in the sorted arrays
    uint64_t slots;        // DELTA
    uint64_t first_slot[kConsumerWaves];
    uint64_t wave_offset[kConsumerWaves];
    uint32_t n, chunks, base, nrows, flags;
    uint32_t start_step[kConsumerWaves], run_len[kConsumerWaves], lane_stride[kConsumerWaves];
    uint32_t own_begin[kConsumerWaves + 1];
    uint32_t row_base[kConsumerWaves];     // OWNER24: first local row of the wavefront's share of this unit
};

__global__ __launch_bounds__(256) void unit_slots_kernel(const DevicePlan* __restrict__ plans, uint32_t nu, const uint64_t* __restrict__ bridges,
                                                        uint64_t* __restrict__ slots) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nu) return;
    const DevicePlan& p = plans[u];
    uint64_t b = 0;
    if (p.n) b = bridges[p.start + p.n - 1] - (p.start ? bridges[p.start - 1] : 0);
    slots[u] = p.n + b;
}

// per unit: own[u][0..14] = share boundaries, [15..28] = first local row of every share, [29..42] = its last local row
constexpr uint32_t kShareWords = (kConsumerWaves + 1) + 2 * kConsumerWaves;
__global__ __launch_bounds__(256) void owner_balanced_shares_kernel(const uint64_t* __restrict__ plan_start, const uint32_t* __restrict__ plan_n, uint32_t nu,
                                                                   const uint64_t* __restrict__ keys, uint32_t max_span, uint32_t* __restrict__ own) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nu) return;
    const uint64_t* e = keys + plan_start[u];
    uint32_t shares[kConsumerWaves + 1];
    auto row_of = [&](uint32_t i) { return uint32_t((e[i] & kPosMask) >> kOwnerColBits); };
    detail::balanced_owner_shares(plan_n[u], row_of, shares, max_span);
    uint32_t* out = own + size_t(u) * kShareWords;
    for (uint32_t w = 0; w <= kConsumerWaves; ++w) out[w] = shares[w];
    for (uint32_t w = 0; w < kConsumerWaves; ++w) {
        const bool any = shares[w + 1] > shares[w];
        out[kConsumerWaves + 1 + w] = any ? row_of(shares[w]) : 0u;
        out[2 * kConsumerWaves + 1 + w] = any ? row_of(shares[w + 1] - 1) : 0u;
    }
}

template <bool k24>
__device__ __forceinline__ void put(uint8_t* chunk, uint32_t lane, uint32_t value, uint32_t where) {
    if (k24) {
        reinterpret_cast<uint32_t*>(chunk)[lane] = value;
        uint8_t* a = chunk + kWaveLanes * 4 + lane * 3;
        a[0] = uint8_t(where); a[1] = uint8_t(where >> 8); a[2] = uint8_t(where >> 16);
    } else {
        reinterpret_cast<uint2*>(chunk)[lane] = make_uint2(value, where);
    }
}

// One workgroup per unit.  PAIRS: slot (chunk c, lane l) holds sorted element l * chunks + c (dense-row blocks: element i in chunk i / 64,
// lane i % 64), chunks dealt round-robin to the wavefronts.
template <bool k24>
__global__ __launch_bounds__(256) void emit_pairs_kernel(const DevicePlan* __restrict__ plans, const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals, uint8_t* __restrict__ image) {
    const DevicePlan& p = plans[blockIdx.x];
    constexpr uint32_t kStride = (k24 ? kChunkBytes24 : kChunkBytes) * kConsumerWaves, kShift = k24 ? kOwnerColBits : 16u;
    const bool dense = p.flags & kBlockDenseRows;
    const uint64_t total = uint64_t(p.chunks) * kWaveLanes;
    for (uint64_t i = threadIdx.x; i < total; i += blockDim.x) {
        const uint32_t lane = dense ? uint32_t(i % kWaveLanes) : uint32_t(i / p.chunks);
        const uint32_t c = dense ? uint32_t(i / kWaveLanes) : uint32_t(i % p.chunks);
        const uint32_t g = p.base + c, w = g % kConsumerWaves;
        const uint32_t first = p.base + (w + kConsumerWaves - p.base % kConsumerWaves) % kConsumerWaves;
        const uint32_t step = p.start_step[w] + (g - first) / kConsumerWaves;
        uint8_t* chunk = image + p.wave_offset[w] + uint64_t(step) * kStride;
        if (i < p.n) {
            const uint32_t pos = uint32_t(keys[p.start + i] & kPosMask);
            put<k24>(chunk, lane, vals[p.start + i], ((pos / kSubTileCols) << kShift) | (pos % kSubTileCols));
        } else {
            put<k24>(chunk, lane, 0u, p.nrows << kShift);
        }
    }
}

// OWNER: per wavefront share, slot (step s, lane l) holds element l * steps + s of the share; the position word IS the key's low bits.
// k24 (OWNER24, stream_tiles.h): step S of the wavefront's stream is slot S % 4 of its record S / 4 -- value word at lane * 16 + 4 j,
// 24-bit position word (row relative to the share's first row) at 1024 + lane * 12 + 3 j.
template <bool k24>
__global__ __launch_bounds__(256) void emit_owner_kernel(const DevicePlan* __restrict__ plans, const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals, uint8_t* __restrict__ image) {
    const DevicePlan& p = plans[blockIdx.x];
    for (uint32_t w = 0; w < kConsumerWaves; ++w) {
        const uint32_t steps = p.run_len[w], n = p.own_begin[w + 1] - p.own_begin[w];
        const uint64_t mine = p.start + p.own_begin[w];
        uint8_t* stream = image + p.wave_offset[w];
        const uint32_t row_base = k24 ? p.row_base[w] : 0u;
        for (uint32_t idx = threadIdx.x; idx < steps * kWaveLanes; idx += blockDim.x) {
            const uint32_t st = idx / kWaveLanes, l = idx % kWaveLanes;
            const uint64_t i = uint64_t(l) * steps + st;
            const uint32_t S = p.start_step[w] + st;
            const uint32_t value = i < n ? vals[mine + i] : 0u;
            if (k24) {
                uint8_t* rec = stream + uint64_t(S / kOwnerRecordSteps) * kOwnerRecordBytes;
                const uint32_t j = S % kOwnerRecordSteps;
                const uint32_t where = i < n ? uint32_t(keys[mine + i] & kPosMask) - (row_base << kOwnerColBits) : kOwnerSpareField << kOwnerColBits;
                reinterpret_cast<uint32_t*>(rec)[l * kOwnerRecordSteps + j] = value;
                uint8_t* a = rec + kOwnerRecordValueBytes + (l * kOwnerRecordSteps + j) * 3;
                a[0] = uint8_t(where); a[1] = uint8_t(where >> 8); a[2] = uint8_t(where >> 16);
            } else {
                reinterpret_cast<uint2*>(stream + uint64_t(S) * kChunkBytes)[l] =
                    make_uint2(value, i < n ? uint32_t(keys[mine + i] & kPosMask) : (p.nrows + w) << kOwnerColBits);
            }
        }
    }
}

// DELTA: slot -> (gap, value, position after the slot).  S(i) = slot of element i inside its unit = i + bridges up to and including i's.
struct Slot { uint32_t gap, val, after; };
__device__ __forceinline__ Slot delta_slot(const DevicePlan& p, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                           const uint64_t* __restrict__ bridges, uint64_t si) {
    const uint64_t bbase = p.start ? bridges[p.start - 1] : 0;
    uint32_t lo = 0, hi = p.n;                       // first element i with S(i) >= si
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (mid + (bridges[p.start + mid] - bbase) < si) lo = mid + 1; else hi = mid;
    }
    const uint32_t i = lo;
    const uint64_t s_i = i + (bridges[p.start + i] - bbase);
    const uint32_t pos = uint32_t(keys[p.start + i] & kPosMask);
    if (si == s_i) {
        uint32_t d = 0;
        if (i > 0) {
            const uint32_t prev = uint32_t(keys[p.start + i - 1] & kPosMask);
            const uint64_t b = bridges[p.start + i] - bridges[p.start + i - 1];
            d = uint32_t(pos - prev - b * kBridgeAdvance);
        }
        return Slot{d, vals[p.start + i], pos};
    }
    const uint32_t prev = uint32_t(keys[p.start + i - 1] & kPosMask);       // i > 0: element 0 has no bridges
    const uint64_t s_prev = (i - 1) + (bridges[p.start + i - 1] - bbase);
    const uint64_t k = si - (s_prev + 1);                                    // 0-based bridge in front of element i
    return Slot{kBridgeGap, 0u, uint32_t(prev + (k + 1) * kBridgeAdvance)};
}

__global__ __launch_bounds__(256) void emit_delta_kernel(const DevicePlan* __restrict__ plans, const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals, const uint64_t* __restrict__ bridges,
                                                        uint8_t* __restrict__ image, uint32_t pad_gap) {
    const DevicePlan& p = plans[blockIdx.x];
    if (!p.n) return;
    const uint32_t scratch_pos = p.nrows * kSubTileCols;      // local row nrows = the spare accumulator
    const uint32_t first_pos = uint32_t(keys[p.start] & kPosMask);
    for (uint32_t w = 0; w < kConsumerWaves; ++w) {
        const uint32_t run = p.run_len[w];
        if (!run) continue;
        uint8_t* rec = image + p.wave_offset[w] + uint64_t(p.start_step[w]) * kRecordBytes;      // start_record == start_step
        const uint32_t slots_in_records = (run + 2) / 2 * 2;                                     // head + run, in whole two-slot records
        for (uint32_t idx = threadIdx.x; idx < slots_in_records * kWaveLanes; idx += blockDim.x) {
            const uint32_t j = idx / kWaveLanes, l = idx % kWaveLanes;                           // j = 0: the head slot
            uint8_t* r = rec + uint64_t(j / 2) * kRecordBytes;
            uint32_t* value_word = reinterpret_cast<uint32_t*>(r) + 2 * l + j % 2;
            uint16_t* gap_word = reinterpret_cast<uint16_t*>(r + kWaveLanes * 8) + 2 * l + j % 2;
            const uint64_t s0 = p.first_slot[w] + uint64_t(l) * p.lane_stride[w];
            if (j == 0) {
                uint32_t head = scratch_pos;
                if (s0 < p.slots) head = s0 == 0 ? first_pos : delta_slot(p, keys, vals, bridges, s0 - 1).after;
                *value_word = head;
                continue;
            }
            if (j > run) { *gap_word = uint16_t(pad_gap); continue; }                            // the dead slot of an odd run
            const uint64_t si = s0 + (j - 1);
            uint32_t value = 0, gap = pad_gap;
            if (si < p.slots) { const Slot s = delta_slot(p, keys, vals, bridges, si); value = s.val; gap = s.gap; }
            *value_word = value;
            *gap_word = uint16_t(gap);
        }
    }
}

// ---- BITMAP (bitmap_tiles.cpp; kernel: spmv_bitmap.hip) -------------------------------------------------------------------------
// The planning (row ranges, slices, block layout, wavefront runs) stays on the host; these kernels do what touches every non-zero:
// count per (row, slice), set the bit, prefix-count the masks per row, place the value, and the same for the matrix-engine image.
struct ElementSource {          // either the uploaded CPSR image or the CSR arrays
    const uint8_t* channels;
    const StreamGroup* groups;
    const uint64_t* advance;
    const uint32_t* indptr;
    const uint32_t* indices;
    const float* values;
    uint64_t nnz;
    uint32_t num_groups, total_slots, num_rows, num_cols, logical_vb, fixed, csr;
};

// visit(row, absolute column, value word) for the elements thread t is responsible for; false + *err on a column outside the matrix
template <typename Visit>
__device__ __forceinline__ void visit_elements(const ElementSource& src, uint64_t t, uint32_t* err, Visit visit) {
    if (src.csr) {
        CsrCursor c;
        if (!csr_segment(src.indptr, src.num_rows, src.nnz, t, c)) return;
        for (; c.e < c.end; ++c.e) {
            while (c.e >= src.indptr[c.row + 1]) ++c.row;
            const uint32_t col = src.indices[c.e];
            if (col >= src.num_cols) { report(err, kErrColumn, c.row / PACK_SIZE, c.row % PACK_SIZE); return; }
            visit(c.row, col, value_word(src.values[c.e], src.fixed != 0));
        }
        return;
    }
    Segment s;
    if (t >= src.total_slots || !find_segment(src.channels, src.groups, src.num_groups, src.total_slots, uint32_t(t), s)) return;
    const StreamGroup& g = src.groups[s.group];
    const uint32_t col_base = g.cp * src.logical_vb;
    walk_segment(g, s, src.advance, [&](uint32_t row, uint32_t col, uint32_t val) { visit(row, col_base + col, val); });
}

// part k of n items cut into `parts`: [k n / parts, (k + 1) n / parts) -- the cut bitmap_tiles.cpp uses for column slices and row pieces
__device__ __forceinline__ uint32_t cut_at(uint32_t n, uint32_t parts, uint32_t k) { return uint32_t(uint64_t(k) * n / parts); }
__device__ __forceinline__ uint32_t part_of(uint32_t n, uint32_t parts, uint32_t x) {
    uint32_t k = min(parts - 1, uint32_t((uint64_t(x) + 1) * parts / n));
    while (cut_at(n, parts, k) > x) --k;
    while (k + 1 < parts && cut_at(n, parts, k + 1) <= x) ++k;
    return k;
}
__device__ __forceinline__ uint32_t padded_masks(uint32_t steps) { return (steps + 7u) / 8u * 8u + 16u; }
// offset of group g (block-local) inside a row's masks: the pieces one after the other, each padded (bitmap_tiles.cpp: mask_index)
__device__ __forceinline__ uint32_t mask_offset(const GpuTiler::BitmapBlock& b, uint32_t g) {
    if (b.pieces <= 1) return g;
    uint32_t j = 0, at = 0;
    while (j + 1 < b.pieces && b.piece_cut[j + 1] <= g) {
        at += padded_masks(b.piece_cut[j + 1] - b.piece_cut[j]);
        ++j;
    }
    return at + (g - b.piece_cut[j]);
}

__global__ __launch_bounds__(256) void bitmap_slice_counts_kernel(ElementSource src, uint32_t slices, uint32_t GR, uint32_t* __restrict__ cnt, uint32_t* err) {
    visit_elements(src, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, err, [&](uint32_t row, uint32_t col, uint32_t) {
        atomicAdd(cnt + size_t(row) * slices + part_of(GR, slices, col / kBitmapGroupCols), 1u);
    });
}

// every element sets its bit; flags[0] = a bit was set already.  masks2: the matrix-engine image's masks (or null)
__global__ __launch_bounds__(256) void bitmap_masks_kernel(ElementSource src, uint32_t slices, uint32_t GR, const uint32_t* __restrict__ range_of_row,
                                                          const GpuTiler::BitmapBlock* __restrict__ blocks, unsigned long long* __restrict__ image64,
                                                          unsigned long long* __restrict__ masks2, uint32_t* flags, uint32_t* err) {
    visit_elements(src, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, err, [&](uint32_t row, uint32_t col, uint32_t) {
        const uint32_t G = col / kBitmapGroupCols, k = slices > 1 ? part_of(GR, slices, G) : 0u;
        const GpuTiler::BitmapBlock& b = blocks[size_t(range_of_row[row]) * slices + k];
        const unsigned long long bit = 1ull << (col % kBitmapGroupCols);
        const uint64_t mw = b.mask_word0 + uint64_t(row - b.row0) * b.stride + mask_offset(b, G - b.gs0);
        if (atomicOr(image64 + mw, bit) & bit) flags[0] = 1u;
        if (masks2) atomicOr(masks2 + (uint64_t(row / kMfmaTileRows) * GR + G) * kMfmaTileRows + row % kMfmaTileRows, bit);
    });
}

// one wavefront per (row, slice): prefix[i] = set bits in the row's masks in front of mask i (exclusive), over the row's `stride` masks
__global__ __launch_bounds__(256) void bitmap_prefix_kernel(uint32_t num_rows, uint32_t slices, const uint32_t* __restrict__ range_of_row,
                                                           const GpuTiler::BitmapBlock* __restrict__ blocks, const unsigned long long* __restrict__ image64,
                                                           uint32_t* __restrict__ prefix) {
    const uint64_t w = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kWaveLanes;
    const uint32_t lane = threadIdx.x % kWaveLanes;
    if (w >= uint64_t(num_rows) * slices) return;
    const uint32_t row = uint32_t(w / slices), k = uint32_t(w % slices);
    const GpuTiler::BitmapBlock& b = blocks[size_t(range_of_row[row]) * slices + k];
    const unsigned long long* m = image64 + b.mask_word0 + uint64_t(row - b.row0) * b.stride;
    uint32_t* out = prefix + b.prefix0 + uint64_t(row - b.row0) * b.stride;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < b.stride; i0 += kWaveLanes) {
        const uint32_t i = i0 + lane;
        const uint32_t c = i < b.stride ? uint32_t(__popcll(m[i])) : 0u;
        uint32_t incl = c;
        for (uint32_t d = 1; d < kWaveLanes; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, kWaveLanes);
            if (lane >= d) incl += up;
        }
        if (i < b.stride) out[i] = carry + incl - c;
        carry += __shfl(incl, kWaveLanes - 1, kWaveLanes);
    }
}

// every element's value word goes to (values before its row) + (values of its row in front of its group) + (set bits below its own)
__global__ __launch_bounds__(256) void bitmap_values_kernel(ElementSource src, uint32_t slices, uint32_t GR, const uint32_t* __restrict__ range_of_row,
                                                           const GpuTiler::BitmapBlock* __restrict__ blocks, const uint64_t* __restrict__ row_value_base,
                                                           const uint32_t* __restrict__ prefix, uint8_t* __restrict__ image, uint32_t* err) {
    const unsigned long long* image64 = reinterpret_cast<const unsigned long long*>(image);
    uint32_t* image32 = reinterpret_cast<uint32_t*>(image);
    visit_elements(src, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, err, [&](uint32_t row, uint32_t col, uint32_t val) {
        const uint32_t G = col / kBitmapGroupCols, k = slices > 1 ? part_of(GR, slices, G) : 0u;
        const GpuTiler::BitmapBlock& b = blocks[size_t(range_of_row[row]) * slices + k];
        const uint64_t in_block = uint64_t(row - b.row0) * b.stride + mask_offset(b, G - b.gs0);
        const unsigned long long m = image64[b.mask_word0 + in_block];
        const uint32_t below = uint32_t(__popcll(m & ((1ull << (col % kBitmapGroupCols)) - 1ull)));
        image32[row_value_base[size_t(row) * slices + k] + prefix[b.prefix0 + in_block] + below] = val;
    });
}

// a run's first 32 masks (zero beyond its steps) and the values of its row in front of its first group
__global__ __launch_bounds__(256) void bitmap_run_heads_kernel(const GpuTiler::BitmapRun* __restrict__ runs, uint32_t num_runs, const unsigned long long* __restrict__ image64,
                                                              const uint32_t* __restrict__ prefix, uint32_t* __restrict__ run_prefix, unsigned long long* __restrict__ heads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, r = t / kBitmapMaskBatch, j = t % kBitmapMaskBatch;
    if (r >= num_runs) return;
    const GpuTiler::BitmapRun run = runs[r];
    heads[t] = j < run.steps ? image64[run.mask_word + j] : 0ull;
    if (j == 0) run_prefix[r] = run.pad ? prefix[run.prefix_at] : 0u;      // pad = the run starts inside a row (also when it is empty)
}

// matrix-engine image: non-zeros of every (row tile, group)
__global__ __launch_bounds__(256) void mfma_group_counts_kernel(const unsigned long long* __restrict__ masks2, uint64_t tile_groups, uint32_t* __restrict__ cnt) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t > tile_groups) return;
    uint32_t c = 0;
    if (t < tile_groups)
        for (uint32_t i = 0; i < kMfmaTileRows; ++i) c += uint32_t(__popcll(masks2[t * kMfmaTileRows + i]));
    cnt[t] = c;            // [tile_groups] = 0: the exclusive scan ends with the total
}
__global__ __launch_bounds__(256) void mfma_unit_base_kernel(const uint32_t* __restrict__ group_pos, uint32_t tiles, uint32_t GR, uint32_t chunk, uint32_t chunks,
                                                            uint32_t* __restrict__ unit_base) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles * chunks) return;
    const uint32_t tile = t / chunks, c = t % chunks;
    unit_base[t] = uint64_t(c) * chunk < GR ? group_pos[size_t(tile) * GR + c * chunk] : group_pos[size_t(tile + 1) * GR];
}
// values in the order the kernel's lanes take them: tile, group, column of the group, row of the tile (bitmap_tiles.cpp)
__global__ __launch_bounds__(256) void mfma_values_kernel(ElementSource src, uint32_t GR, const unsigned long long* __restrict__ masks2,
                                                         const uint32_t* __restrict__ group_pos, uint32_t* __restrict__ values, uint32_t* err) {
    visit_elements(src, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, err, [&](uint32_t row, uint32_t col, uint32_t val) {
        const uint32_t G = col / kBitmapGroupCols, p = col % kBitmapGroupCols, i = row % kMfmaTileRows;
        const uint64_t tg = uint64_t(row / kMfmaTileRows) * GR + G;
        const unsigned long long* m = masks2 + tg * kMfmaTileRows;
        uint32_t rank = 0;
        for (uint32_t r = 0; r < kMfmaTileRows; ++r) {
            rank += uint32_t(__popcll(m[r] & ((1ull << p) - 1ull)));
            if (r < i) rank += uint32_t((m[r] >> p) & 1ull);
        }
        values[group_pos[tg] + rank] = val;
    });
}

template <typename T>
hipError_t upload(T** dst, const std::vector<T>& src, hipStream_t stream) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(src.size() * sizeof(T), 16));
    if (e != hipSuccess || src.empty()) return e;
    return hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, stream);
}

}  // namespace

// hs_create: load this file's code object now instead of inside the first hs_load_matrix
hipError_t warm_gpu_tiler() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&segment_sums_kernel));
}

GpuTiler::GpuTiler(const Layout& layout, const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS], hipStream_t stream)
    : L_(layout), geom_(*layout.g), channel_(channel), n_packets_(n_packets), stream_(stream) {
    L_.g = &geom_;
}

GpuTiler::GpuTiler(const Layout& layout, const CsrView& csr, hipStream_t stream)
    : L_(layout), geom_(*layout.g), channel_(nullptr), n_packets_(nullptr), stream_(stream), csr_(&csr) {
    L_.g = &geom_;
}

GpuTiler::~GpuTiler() {
    for (void* p : {static_cast<void*>(d_indptr_), static_cast<void*>(d_indices_), static_cast<void*>(d_values_)})
        if (p) (void)hipFree(p);
    for (void* p : {static_cast<void*>(d_channels_), d_groups_, static_cast<void*>(d_advance_), static_cast<void*>(d_base_), static_cast<void*>(d_scalar_),
                    static_cast<void*>(d_block_of_row_), static_cast<void*>(d_keys_), static_cast<void*>(d_vals_), static_cast<void*>(d_bridges_),
                    static_cast<void*>(d_image_), static_cast<void*>(d_mfma_)})
        if (p) (void)hipFree(p);
}

bool GpuTiler::fail(const std::string& what) {
    error_ = what;
    return false;
}
bool GpuTiler::check(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    return fail(std::string("gpu re-tile: ") + what + ": " + hipGetErrorString(e));
}

// The 16 buffers back to back on the device + the lane-stream table (header parsing and bounds checks as in
// detail::walk_channel_partition, on the host: a few thousand headers).
bool GpuTiler::upload_channels() {
    const uint32_t F = L_.F, RP = L_.row_parts, CP = L_.col_parts;
    const uint64_t parts = uint64_t(RP) * CP, payload_base = parts * (1 + F);
    uint64_t offset[NUM_HBM_CHANNELS], total = 0;
    for (uint32_t pc = 0; pc < NUM_HBM_CHANNELS; ++pc) {
        offset[pc] = total;
        total += n_packets_[pc] * sizeof(MatPkt);
    }
    std::vector<StreamGroup> groups;
    groups.reserve(size_t(parts) * NUM_HBM_CHANNELS * F);
    const uint64_t stride_rows = uint64_t(PACK_SIZE) * NUM_HBM_CHANNELS * F;
    uint64_t slots = 0;
    for (uint32_t rp = 0; rp < RP; ++rp)
        for (uint32_t cp = 0; cp < CP; ++cp)
            for (uint32_t pc = 0; pc < NUM_HBM_CHANNELS; ++pc) {
                const MatPkt* buf = static_cast<const MatPkt*>(channel_[pc]);
                const uint64_t pid = uint64_t(rp) * CP + cp, header = pid * (1 + F);
                auto where = [&]() { return "channel " + std::to_string(pc) + ", row partition " + std::to_string(rp) + ", column partition " + std::to_string(cp) + ": "; };
                if (header + 1 + F > n_packets_[pc]) return fail(where() + "partition header lies outside the channel buffer");
                const uint64_t start = buf[header].indices.data[0];
                const uint64_t row_base = uint64_t(rp) * geom_.logical_ob;
                for (uint32_t f = 0; f < F; ++f) {
                    const PackedWord& lens = buf[header + 1 + f].indices;
                    uint32_t longest = 0;
                    for (uint32_t k = 0; k < PACK_SIZE; ++k) longest = std::max(longest, lens.data[k]);
                    if (longest && payload_base + start + uint64_t(longest - 1) * F + f >= n_packets_[pc])
                        return fail(where() + "payload runs past the end of the channel buffer");
                    if (!longest) continue;
                    const uint32_t vc = pc + f * NUM_HBM_CHANNELS;
                    StreamGroup g{};
                    g.byte_off = offset[pc] + (payload_base + start + f) * sizeof(MatPkt);
                    g.stride = F * uint32_t(sizeof(MatPkt));
                    for (uint32_t k = 0; k < PACK_SIZE; ++k) g.len[k] = lens.data[k];
                    g.row = uint32_t(row_base + uint64_t(vc) * PACK_SIZE);
                    g.row_stride = uint32_t(stride_rows);
                    g.row_limit = uint32_t(row_base + L_.rows_in_part(rp));
                    g.col_limit = L_.cols_in_part(cp);
                    g.cp = cp;
                    g.fixed = geom_.impl == IMPL_FIXED;
                    g.first = uint32_t(slots);
                    g.nseg = (longest + kSegment - 1) / kSegment;
                    slots += uint64_t(g.nseg) * PACK_SIZE;
                    groups.push_back(g);
                }
            }
    if (slots >= (uint64_t(1) << 32)) return fail("gpu re-tile: image too large for 32-bit segment indices");
    num_groups_ = uint32_t(groups.size());
    total_slots_ = uint32_t(slots);
    if (!check(hipMalloc(reinterpret_cast<void**>(&d_channels_), std::max<uint64_t>(total, 64)), "hipMalloc(channels)")) return false;
    for (uint32_t pc = 0; pc < NUM_HBM_CHANNELS; ++pc)
        if (n_packets_[pc] &&
            !check(hipMemcpyAsync(d_channels_ + offset[pc], channel_[pc], n_packets_[pc] * sizeof(MatPkt), hipMemcpyHostToDevice, stream_), "upload channel"))
            return false;
    StreamGroup* d = nullptr;
    if (!check(upload(&d, groups, stream_), "upload stream groups")) return false;
    d_groups_ = d;
    if (!check(hipMalloc(reinterpret_cast<void**>(&d_scalar_), 64), "hipMalloc")) return false;
    if (!check(hipMemsetAsync(d_scalar_, 0, 64, stream_), "hipMemset")) return false;
    return check(hipStreamSynchronize(stream_), "upload");       // the group table is a temporary
}

bool GpuTiler::decode_error(const char* pass) {
    uint32_t words[3] = {0, 0, 0};
    if (!check(hipMemcpyAsync(words, d_scalar_, 12, hipMemcpyDeviceToHost, stream_), pass)) return false;
    if (!check(hipStreamSynchronize(stream_), pass)) return false;
    if (!words[0]) return true;
    // which stream: for the message only (the host walk names channel / partitions too)
    return fail(std::string("lane stream ") + std::to_string(uint64_t(words[1]) * PACK_SIZE + words[2]) + ": " +
                (words[0] == kErrColumn ? "column index outside the column partition" : "decoded row outside the row partition (marker count wrapped?)"));
}

// CSR source: validate indptr (host, a pass over num_rows words), upload the three arrays; indptr is extended over the padding rows
bool GpuTiler::upload_csr(std::vector<uint32_t>& row_nnz) {
    const CsrView& m = *csr_;
    if (m.num_rows > L_.num_rows || m.num_cols > L_.num_cols) return fail("CSR matrix larger than the padded dimensions");
    if (!m.indptr || m.indptr[0] != 0) return fail("CSR indptr must start at 0");
    for (uint32_t r = 0; r < m.num_rows; ++r)
        if (m.indptr[r + 1] < m.indptr[r]) return fail("CSR indptr decreases at row " + std::to_string(r));
    total_ = m.indptr[m.num_rows];
    if (total_ && (!m.indices || !m.values)) return fail("CSR arrays missing");
    row_nnz.assign(L_.num_rows, 0);
    std::vector<uint32_t> indptr(size_t(L_.num_rows) + 1, uint32_t(total_));
    for (uint32_t r = 0; r < m.num_rows; ++r) { indptr[r] = m.indptr[r]; row_nnz[r] = m.indptr[r + 1] - m.indptr[r]; }
    const size_t n = std::max<uint64_t>(total_, 1);
    bool ok = check(upload(&d_indptr_, indptr, stream_), "upload indptr") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_indices_), n * 4), "hipMalloc(indices)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_values_), n * 4), "hipMalloc(values)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_scalar_), 64), "hipMalloc") && check(hipMemsetAsync(d_scalar_, 0, 64, stream_), "hipMemset");
    if (ok && total_)
        ok = check(hipMemcpyAsync(d_indices_, m.indices, size_t(total_) * 4, hipMemcpyHostToDevice, stream_), "upload indices") &&
             check(hipMemcpyAsync(d_values_, m.values, size_t(total_) * 4, hipMemcpyHostToDevice, stream_), "upload values");
    return ok && check(hipStreamSynchronize(stream_), "upload");      // `indptr` is a temporary
}

bool GpuTiler::count_rows(std::vector<uint32_t>& row_nnz, uint64_t& nnz) {
    detail::PhaseTimer timer;
    if (csr_) {
        const bool ok = upload_csr(row_nnz);
        nnz = total_;
        timer.lap("gpu: upload CSR");
        return ok;
    }
    if (!upload_channels()) return false;
    timer.lap("gpu: upload CPSR image");
    const StreamGroup* groups = static_cast<const StreamGroup*>(d_groups_);
    const size_t slots = size_t(total_slots_) + 1;                   // + one zero: the exclusive scans end with the totals
    const dim3 grid((total_slots_ + 255) / 256), block(256);
    uint32_t* d_rows = nullptr;
    uint64_t *d_adv_in = nullptr, *d_cnt_in = nullptr;
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_rows), std::max<size_t>(size_t(L_.num_rows) * 4, 16)), "hipMalloc") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_adv_in), slots * 8), "hipMalloc") && check(hipMalloc(reinterpret_cast<void**>(&d_cnt_in), slots * 8), "hipMalloc") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_advance_), slots * 8), "hipMalloc") && check(hipMalloc(reinterpret_cast<void**>(&d_base_), slots * 8), "hipMalloc") &&
              check(hipMemsetAsync(d_rows, 0, size_t(L_.num_rows) * 4, stream_), "hipMemset") &&
              check(hipMemsetAsync(d_adv_in + total_slots_, 0, 8, stream_), "hipMemset") && check(hipMemsetAsync(d_cnt_in + total_slots_, 0, 8, stream_), "hipMemset") &&
              check(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_adv_in, d_advance_, slots, stream_), "scan (size)") &&
              check(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), "hipMalloc(scan)");
    auto lap = [&](const char* what) { if (timer.on) { (void)hipStreamSynchronize(stream_); timer.lap(what); } };
    lap("gpu:   allocations");
    if (ok && total_slots_) {
        hipLaunchKernelGGL(segment_sums_kernel, grid, block, 0, stream_, d_channels_, groups, num_groups_, total_slots_, d_adv_in, d_cnt_in);
        ok = check(hipGetLastError(), "segment_sums_kernel");
    }
    lap("gpu:   segment sums");
    ok = ok && check(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_adv_in, d_advance_, slots, stream_), "scan") &&
         check(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_cnt_in, d_base_, slots, stream_), "scan");
    lap("gpu:   scans");
    if (ok && total_slots_) {
        hipLaunchKernelGGL(count_rows_kernel, grid, block, 0, stream_, d_channels_, groups, num_groups_, total_slots_, d_advance_, d_rows, d_scalar_);
        ok = check(hipGetLastError(), "count_rows_kernel") && decode_error("count rows");
    }
    lap("gpu:   row counts");
    row_nnz.assign(L_.num_rows, 0);
    uint64_t all = 0;
    ok = ok && (L_.num_rows == 0 || check(hipMemcpyAsync(row_nnz.data(), d_rows, size_t(L_.num_rows) * 4, hipMemcpyDeviceToHost, stream_), "read row counts")) &&
         check(hipMemcpyAsync(&all, d_base_ + total_slots_, 8, hipMemcpyDeviceToHost, stream_), "read total") &&
         check(hipStreamSynchronize(stream_), "count rows");
    lap("gpu:   read back");
    for (void* p : {static_cast<void*>(d_rows), static_cast<void*>(d_adv_in), static_cast<void*>(d_cnt_in), d_temp})
        if (p) (void)hipFree(p);
    lap("gpu:   frees");
    total_ = nnz = all;
    timer.lap("gpu: segment scan + row counts");
    return ok;
}

bool GpuTiler::count_tiles(const std::vector<uint32_t>& block_of_row, uint32_t num_ranges, std::vector<uint32_t>& cnt) {
    const uint32_t S = L_.subs_per_cp, tiles = L_.col_parts * S;
    cnt.assign(size_t(num_ranges) * tiles, 0);
    uint32_t* d_cnt = nullptr;
    if (!check(upload(&d_block_of_row_, block_of_row, stream_), "upload block_of_row")) return false;
    if (!check(hipMalloc(reinterpret_cast<void**>(&d_cnt), std::max<size_t>(cnt.size() * 4, 16)), "hipMalloc")) return false;
    bool ok = check(hipMemsetAsync(d_cnt, 0, cnt.size() * 4, stream_), "hipMemset");
    if (ok && csr_ && total_) {
        const uint64_t threads = (total_ + kCsrSegment - 1) / kCsrSegment;
        hipLaunchKernelGGL(csr_count_tiles_kernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream_, d_indptr_, d_indices_, L_.num_rows, total_,
                           csr_->num_cols, uint32_t(geom_.logical_vb), d_block_of_row_, tiles, S, L_.sub_width, d_cnt, d_scalar_);
        ok = check(hipGetLastError(), "csr_count_tiles_kernel");
        if (ok) {
            uint32_t words[3] = {0, 0, 0};
            ok = check(hipMemcpyAsync(words, d_scalar_, 12, hipMemcpyDeviceToHost, stream_), "count tiles") && check(hipStreamSynchronize(stream_), "count tiles");
            if (ok && words[0]) ok = fail("CSR row " + std::to_string(uint64_t(words[1]) * PACK_SIZE + words[2]) + ": column index outside the matrix");
        }
    } else if (ok && total_slots_) {
        hipLaunchKernelGGL(count_tiles_kernel, dim3((total_slots_ + 255) / 256), dim3(256), 0, stream_, d_channels_, static_cast<const StreamGroup*>(d_groups_),
                           num_groups_, total_slots_, d_advance_, d_block_of_row_, tiles, S, L_.sub_width, d_cnt);
        ok = check(hipGetLastError(), "count_tiles_kernel");
    }
    ok = ok && (cnt.empty() || check(hipMemcpyAsync(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost, stream_), "read tile counts")) &&
         check(hipStreamSynchronize(stream_), "count tiles");
    (void)hipFree(d_cnt);
    return ok;
}

bool GpuTiler::sort_elements(const std::vector<uint32_t>& block_of_row, const std::vector<uint32_t>& range_row0, const std::vector<uint32_t>& unit_of,
                             const std::vector<UnitPlan>& plans, bool& duplicates) {
    (void)block_of_row;      // already on the device (count_tiles)
    detail::PhaseTimer timer;
    duplicates = false;
    const uint32_t S = L_.subs_per_cp, tiles = L_.col_parts * S;
    uint32_t *d_row0 = nullptr, *d_unit_of = nullptr, *d_vals_in = nullptr;
    uint64_t* d_keys_in = nullptr;
    void* d_temp = nullptr;
    const size_t n = std::max<uint64_t>(total_, 1);
    bool ok = check(upload(&d_row0, range_row0, stream_), "upload range rows") && check(upload(&d_unit_of, unit_of, stream_), "upload unit table") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_keys_in), n * 8), "hipMalloc(keys)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_vals_in), n * 4), "hipMalloc(values)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_keys_), n * 8), "hipMalloc(keys)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_vals_), n * 4), "hipMalloc(values)");
    if (ok && csr_ && total_) {
        const uint64_t threads = (total_ + kCsrSegment - 1) / kCsrSegment;
        hipLaunchKernelGGL(csr_keys_kernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream_, d_indptr_, d_indices_, d_values_, L_.num_rows, total_,
                           uint32_t(geom_.logical_vb), uint32_t(geom_.impl == IMPL_FIXED), d_block_of_row_, d_row0, d_unit_of, tiles, S, L_.sub_width, d_keys_in,
                           d_vals_in);
        ok = check(hipGetLastError(), "csr_keys_kernel");
    } else if (ok && total_slots_) {
        hipLaunchKernelGGL(keys_kernel, dim3((total_slots_ + 255) / 256), dim3(256), 0, stream_, d_channels_, static_cast<const StreamGroup*>(d_groups_),
                           num_groups_, total_slots_, d_advance_, d_base_, d_block_of_row_, d_row0, d_unit_of, tiles, S, L_.sub_width, d_keys_in, d_vals_in);
        ok = check(hipGetLastError(), "keys_kernel");
    }
    if (timer.on) { (void)hipStreamSynchronize(stream_); timer.lap("gpu: keys"); }
    if (ok && total_) {
        uint32_t unit_bits = 1;
        while ((uint64_t(1) << unit_bits) < plans.size() + 1) ++unit_bits;
        const int end_bit = int(kPosBits + unit_bits);
        size_t temp_bytes = 0;
        ok = check(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys_in, d_keys_, d_vals_in, d_vals_, total_, 0, end_bit, stream_), "radix sort (size)") &&
             check(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), "hipMalloc(sort)") &&
             check(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys_in, d_keys_, d_vals_in, d_vals_, total_, 0, end_bit, stream_), "radix sort");
        if (ok) {
            (void)hipMemsetAsync(d_scalar_ + 4, 0, 4, stream_);
            hipLaunchKernelGGL(duplicates_kernel, dim3(uint32_t((total_ + 255) / 256)), dim3(256), 0, stream_, d_keys_, total_, d_scalar_ + 4);
            uint32_t flag = 0;
            ok = check(hipMemcpyAsync(&flag, d_scalar_ + 4, 4, hipMemcpyDeviceToHost, stream_), "duplicates") && check(hipStreamSynchronize(stream_), "sort");
            duplicates = flag != 0;
        }
    }
    ok = ok && check(hipStreamSynchronize(stream_), "sort");
    timer.lap("gpu: radix sort");
    for (void* p : {static_cast<void*>(d_row0), static_cast<void*>(d_unit_of), static_cast<void*>(d_keys_in), static_cast<void*>(d_vals_in), d_temp})
        if (p) (void)hipFree(p);
    // the source is not needed any more
    if (d_channels_) { (void)hipFree(d_channels_); d_channels_ = nullptr; }
    for (void** p : {reinterpret_cast<void**>(&d_indptr_), reinterpret_cast<void**>(&d_indices_), reinterpret_cast<void**>(&d_values_)})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    return ok;
}

namespace {
bool make_device_plans(const std::vector<UnitPlan>& plans, const std::vector<uint32_t>& block_of_unit, const std::vector<Block>& blocks,
                       std::vector<DevicePlan>& out) {
    out.resize(plans.size());
    for (size_t u = 0; u < plans.size(); ++u) {
        const UnitPlan& up = plans[u];
        DevicePlan& d = out[u];
        std::memset(&d, 0, sizeof(d));
        d.start = up.scratch; d.slots = up.slots; d.n = up.n; d.chunks = up.chunks; d.base = up.base;
        if (!blocks.empty()) {
            const Block& blk = blocks[block_of_unit[u]];
            d.nrows = blk.nrows; d.flags = blk.flags;
            for (uint32_t w = 0; w < kConsumerWaves; ++w) { d.wave_offset[w] = blk.wave_offset[w]; d.row_base[w] = up.own_row[w]; }
        }
        for (uint32_t w = 0; w < kConsumerWaves; ++w) {
            d.first_slot[w] = up.first_slot[w]; d.start_step[w] = up.start_step[w]; d.run_len[w] = up.run_len[w]; d.lane_stride[w] = up.lane_stride[w]; d.own_begin[w] = up.own_begin[w];
        }
        d.own_begin[kConsumerWaves] = up.own_begin[kConsumerWaves];
    }
    return true;
}
}  // namespace

bool GpuTiler::delta_slots(std::vector<UnitPlan>& plans) {
    const size_t n = std::max<uint64_t>(total_, 1);
    uint64_t *d_in = nullptr, *d_slots = nullptr;
    DevicePlan* d_plans = nullptr;
    void* d_temp = nullptr;
    std::vector<DevicePlan> dp;
    make_device_plans(plans, {}, {}, dp);
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_in), n * 8), "hipMalloc") && check(hipMalloc(reinterpret_cast<void**>(&d_bridges_), n * 8), "hipMalloc") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_slots), std::max<size_t>(plans.size() * 8, 16)), "hipMalloc") && check(upload(&d_plans, dp, stream_), "upload plans");
    if (ok && total_) {
        hipLaunchKernelGGL(bridges_kernel, dim3(uint32_t((total_ + 255) / 256)), dim3(256), 0, stream_, d_keys_, total_, d_in);
        size_t temp_bytes = 0;
        ok = check(hipGetLastError(), "bridges_kernel") &&
             check(hipcub::DeviceScan::InclusiveSum(nullptr, temp_bytes, d_in, d_bridges_, total_, stream_), "scan (size)") &&
             check(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), "hipMalloc(scan)") &&
             check(hipcub::DeviceScan::InclusiveSum(d_temp, temp_bytes, d_in, d_bridges_, total_, stream_), "scan");
    }
    std::vector<uint64_t> slots(plans.size(), 0);
    if (ok && !plans.empty()) {
        hipLaunchKernelGGL(unit_slots_kernel, dim3(uint32_t((plans.size() + 255) / 256)), dim3(256), 0, stream_, d_plans, uint32_t(plans.size()), d_bridges_, d_slots);
        ok = check(hipGetLastError(), "unit_slots_kernel") &&
             check(hipMemcpyAsync(slots.data(), d_slots, slots.size() * 8, hipMemcpyDeviceToHost, stream_), "read slots");
    }
    ok = ok && check(hipStreamSynchronize(stream_), "delta slots");
    for (void* p : {static_cast<void*>(d_in), static_cast<void*>(d_slots), static_cast<void*>(d_plans), d_temp})
        if (p) (void)hipFree(p);
    if (ok)
        for (size_t u = 0; u < plans.size(); ++u) plans[u].slots = slots[u];
    return ok;
}

bool GpuTiler::owner_shares(std::vector<UnitPlan>& plans, uint32_t max_span) {
    const uint32_t nu = uint32_t(plans.size());
    if (!nu) return true;
    std::vector<uint64_t> start(nu);
    std::vector<uint32_t> count(nu);
    for (uint32_t u = 0; u < nu; ++u) { start[u] = plans[u].scratch; count[u] = plans[u].n; }
    uint64_t* d_start = nullptr;
    uint32_t *d_n = nullptr, *d_own = nullptr;
    std::vector<uint32_t> own(size_t(nu) * kShareWords);
    bool ok = check(upload(&d_start, start, stream_), "upload") && check(upload(&d_n, count, stream_), "upload") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_own), own.size() * 4), "hipMalloc");
    if (ok) {
        hipLaunchKernelGGL(owner_balanced_shares_kernel, dim3((nu + 255) / 256), dim3(256), 0, stream_, d_start, d_n, nu, d_keys_, max_span, d_own);
        ok = check(hipGetLastError(), "owner_balanced_shares_kernel") && check(hipMemcpyAsync(own.data(), d_own, own.size() * 4, hipMemcpyDeviceToHost, stream_), "read shares") &&
             check(hipStreamSynchronize(stream_), "owner shares");
    }
    for (void* p : {static_cast<void*>(d_start), static_cast<void*>(d_n), static_cast<void*>(d_own)})
        if (p) (void)hipFree(p);
    if (ok)
        for (uint32_t u = 0; u < nu; ++u) {
            const uint32_t* o = own.data() + size_t(u) * kShareWords;
            for (uint32_t w = 0; w <= kConsumerWaves; ++w) plans[u].own_begin[w] = o[w];
            for (uint32_t w = 0; w < kConsumerWaves; ++w) {
                plans[u].own_row[w] = o[kConsumerWaves + 1 + w];
                plans[u].own_last[w] = o[2 * kConsumerWaves + 1 + w];
            }
        }
    return ok;
}

bool GpuTiler::emit(StreamFormat format, uint64_t image_bytes, uint64_t slack_bytes, const std::vector<UnitPlan>& plans,
                    const std::vector<uint32_t>& block_of_unit, const std::vector<Block>& blocks, bool is_float) {
    std::vector<DevicePlan> dp;
    make_device_plans(plans, block_of_unit, blocks, dp);
    DevicePlan* d_plans = nullptr;
    const size_t bytes = std::max<uint64_t>(image_bytes + slack_bytes, 256);
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_image_), bytes), "hipMalloc(image)") && check(hipMemsetAsync(d_image_, 0, bytes, stream_), "hipMemset(image)") &&
              check(upload(&d_plans, dp, stream_), "upload plans");
    const dim3 grid(uint32_t(plans.size())), block(256);
    if (ok && !plans.empty()) {
        switch (format) {
            case kFormatPairs: hipLaunchKernelGGL(emit_pairs_kernel<false>, grid, block, 0, stream_, d_plans, d_keys_, d_vals_, d_image_); break;
            case kFormatPairs24: hipLaunchKernelGGL(emit_pairs_kernel<true>, grid, block, 0, stream_, d_plans, d_keys_, d_vals_, d_image_); break;
            case kFormatOwner: hipLaunchKernelGGL(emit_owner_kernel<false>, grid, block, 0, stream_, d_plans, d_keys_, d_vals_, d_image_); break;
            case kFormatOwner24: hipLaunchKernelGGL(emit_owner_kernel<true>, grid, block, 0, stream_, d_plans, d_keys_, d_vals_, d_image_); break;
            case kFormatDelta:
                hipLaunchKernelGGL(emit_delta_kernel, grid, block, 0, stream_, d_plans, d_keys_, d_vals_, d_bridges_, d_image_, is_float ? kBridgeGap : 0u);
                break;
            default: ok = fail("gpu re-tile: format not supported");
        }
        ok = ok && check(hipGetLastError(), "emit kernel");
    }
    ok = ok && check(hipStreamSynchronize(stream_), "emit");
    if (d_plans) (void)hipFree(d_plans);
    for (void** p : {reinterpret_cast<void**>(&d_keys_), reinterpret_cast<void**>(&d_vals_), reinterpret_cast<void**>(&d_bridges_)})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    return ok;
}

// ---- BITMAP ------------------------------------------------------------------------------------------------------------------------
bool GpuTiler::bitmap_slice_counts(uint32_t slices, uint32_t GR, std::vector<uint32_t>& cnt) {
    cnt.assign(size_t(L_.num_rows) * slices, 0);
    ElementSource src{d_channels_, static_cast<const StreamGroup*>(d_groups_), d_advance_, d_indptr_, d_indices_, d_values_, total_, num_groups_, total_slots_,
                      L_.num_rows, csr_ ? csr_->num_cols : L_.num_cols, uint32_t(geom_.logical_vb), uint32_t(geom_.impl == IMPL_FIXED), csr_ ? 1u : 0u};
    const uint64_t threads = csr_ ? (total_ + kCsrSegment - 1) / kCsrSegment : total_slots_;
    uint32_t* d_cnt = nullptr;
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_cnt), std::max<size_t>(cnt.size() * 4, 16)), "hipMalloc") &&
              check(hipMemsetAsync(d_cnt, 0, cnt.size() * 4, stream_), "hipMemset");
    if (ok && threads) {
        hipLaunchKernelGGL(bitmap_slice_counts_kernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream_, src, slices, GR, d_cnt, d_scalar_);
        ok = check(hipGetLastError(), "bitmap_slice_counts_kernel");
    }
    ok = ok && (cnt.empty() || check(hipMemcpyAsync(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost, stream_), "read slice counts"));
    if (ok) {
        uint32_t words[3] = {0, 0, 0};
        ok = check(hipMemcpyAsync(words, d_scalar_, 12, hipMemcpyDeviceToHost, stream_), "slice counts") && check(hipStreamSynchronize(stream_), "slice counts");
        if (ok && words[0]) ok = fail("CSR row " + std::to_string(uint64_t(words[1]) * PACK_SIZE + words[2]) + ": column index outside the matrix");
    }
    if (d_cnt) (void)hipFree(d_cnt);
    return ok;
}

bool GpuTiler::bitmap_emit(uint32_t slices, uint32_t GR, const std::vector<uint32_t>& range_of_row, const std::vector<BitmapBlock>& blocks,
                           const std::vector<uint64_t>& row_value_base, uint64_t image_bytes, uint64_t slack_bytes, const std::vector<BitmapRun>& runs,
                           std::vector<uint32_t>& run_prefix, std::vector<uint64_t>& run_heads, const MfmaImage* mfma, bool& duplicates) {
    detail::PhaseTimer timer;
    duplicates = false;
    ElementSource src{d_channels_, static_cast<const StreamGroup*>(d_groups_), d_advance_, d_indptr_, d_indices_, d_values_, total_, num_groups_, total_slots_,
                      L_.num_rows, csr_ ? csr_->num_cols : L_.num_cols, uint32_t(geom_.logical_vb), uint32_t(geom_.impl == IMPL_FIXED), csr_ ? 1u : 0u};
    const uint64_t threads = csr_ ? (total_ + kCsrSegment - 1) / kCsrSegment : total_slots_;
    const dim3 egrid(uint32_t((threads + 255) / 256)), block(256);
    uint64_t prefix_words = 0;
    for (const BitmapBlock& b : blocks) prefix_words = std::max<uint64_t>(prefix_words, b.prefix0 + uint64_t(b.nrows) * b.stride);
    const size_t bytes = std::max<uint64_t>(image_bytes + slack_bytes, 256);
    const uint32_t num_runs = uint32_t(runs.size());
    run_prefix.assign(num_runs, 0);
    run_heads.assign(size_t(num_runs) * kBitmapMaskBatch, 0);
    uint32_t *d_range = nullptr, *d_prefix = nullptr, *d_flags = nullptr, *d_run_prefix = nullptr, *d_cnt = nullptr, *d_pos = nullptr;
    BitmapBlock* d_blocks = nullptr;
    BitmapRun* d_runs = nullptr;
    uint64_t* d_row_base = nullptr;
    unsigned long long* d_heads = nullptr;
    void* d_temp = nullptr;
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_image_), bytes), "hipMalloc(image)") && check(hipMemsetAsync(d_image_, 0, bytes, stream_), "hipMemset(image)") &&
              check(upload(&d_range, range_of_row, stream_), "upload row ranges") && check(upload(&d_blocks, blocks, stream_), "upload blocks") &&
              check(upload(&d_row_base, row_value_base, stream_), "upload row bases") && check(upload(&d_runs, runs, stream_), "upload runs") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_prefix), std::max<size_t>(prefix_words * 4, 16)), "hipMalloc(prefix)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_flags), 16), "hipMalloc") && check(hipMemsetAsync(d_flags, 0, 16, stream_), "hipMemset") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_run_prefix), std::max<size_t>(size_t(num_runs) * 4, 16)), "hipMalloc") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_heads), std::max<size_t>(run_heads.size() * 8, 16)), "hipMalloc");
    unsigned long long* image64 = reinterpret_cast<unsigned long long*>(d_image_);
    unsigned long long* masks2 = nullptr;
    if (ok && mfma) {
        ok = check(hipMalloc(reinterpret_cast<void**>(&d_mfma_), std::max<size_t>(mfma->words_bytes, 16)), "hipMalloc(second image)") &&
             check(hipMemsetAsync(d_mfma_, 0, mfma->words_bytes, stream_), "hipMemset(second image)");
        masks2 = reinterpret_cast<unsigned long long*>(d_mfma_);
    }
    if (ok && threads) {
        hipLaunchKernelGGL(bitmap_masks_kernel, egrid, block, 0, stream_, src, slices, GR, d_range, d_blocks, image64, masks2, d_flags, d_scalar_);
        ok = check(hipGetLastError(), "bitmap_masks_kernel");
    }
    uint32_t flags[4] = {0, 0, 0, 0}, errw[3] = {0, 0, 0};
    ok = ok && check(hipMemcpyAsync(flags, d_flags, 16, hipMemcpyDeviceToHost, stream_), "read flags") &&
         check(hipMemcpyAsync(errw, d_scalar_, 12, hipMemcpyDeviceToHost, stream_), "read flags") && check(hipStreamSynchronize(stream_), "bitmap masks");
    if (ok && errw[0]) ok = fail("CSR row " + std::to_string(uint64_t(errw[1]) * PACK_SIZE + errw[2]) + ": column index outside the matrix");
    timer.lap("gpu: bitmap masks");
    duplicates = ok && flags[0] != 0;
    if (ok && !duplicates) {
        const uint64_t waves = uint64_t(L_.num_rows) * slices;
        if (waves) {
            hipLaunchKernelGGL(bitmap_prefix_kernel, dim3(uint32_t((waves * kWaveLanes + 255) / 256)), block, 0, stream_, L_.num_rows, slices, d_range, d_blocks, image64, d_prefix);
            ok = check(hipGetLastError(), "bitmap_prefix_kernel");
        }
        if (ok && threads) {
            hipLaunchKernelGGL(bitmap_values_kernel, egrid, block, 0, stream_, src, slices, GR, d_range, d_blocks, d_row_base, d_prefix, d_image_, d_scalar_);
            ok = check(hipGetLastError(), "bitmap_values_kernel");
        }
        if (ok && num_runs) {
            hipLaunchKernelGGL(bitmap_run_heads_kernel, dim3((num_runs * kBitmapMaskBatch + 255) / 256), block, 0, stream_, d_runs, num_runs, image64, d_prefix, d_run_prefix, d_heads);
            ok = check(hipGetLastError(), "bitmap_run_heads_kernel") &&
                 check(hipMemcpyAsync(run_prefix.data(), d_run_prefix, size_t(num_runs) * 4, hipMemcpyDeviceToHost, stream_), "read run offsets") &&
                 check(hipMemcpyAsync(run_heads.data(), d_heads, run_heads.size() * 8, hipMemcpyDeviceToHost, stream_), "read run heads");
        }
        if (ok && mfma) {       // [masks: tiles x groups x 16 x 8 bytes][first value of every unit][values]
            const uint64_t tile_groups = uint64_t(mfma->tiles) * mfma->groups;
            uint32_t* words = reinterpret_cast<uint32_t*>(d_mfma_);
            size_t temp_bytes = 0;
            ok = check(hipMalloc(reinterpret_cast<void**>(&d_cnt), (tile_groups + 1) * 4), "hipMalloc") && check(hipMalloc(reinterpret_cast<void**>(&d_pos), (tile_groups + 1) * 4), "hipMalloc") &&
                 check(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_cnt, d_pos, tile_groups + 1, stream_), "scan (size)") &&
                 check(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), "hipMalloc(scan)");
            if (ok) {
                hipLaunchKernelGGL(mfma_group_counts_kernel, dim3(uint32_t((tile_groups + 1 + 255) / 256)), block, 0, stream_, masks2, tile_groups, d_cnt);
                ok = check(hipGetLastError(), "mfma_group_counts_kernel") &&
                     check(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_cnt, d_pos, tile_groups + 1, stream_), "scan");
            }
            if (ok) {
                hipLaunchKernelGGL(mfma_unit_base_kernel, dim3((mfma->tiles * mfma->chunks + 255) / 256), block, 0, stream_, d_pos, mfma->tiles, mfma->groups, mfma->chunk,
                                   mfma->chunks, words + mfma->offsets_word);
                ok = check(hipGetLastError(), "mfma_unit_base_kernel");
            }
            if (ok && threads) {
                hipLaunchKernelGGL(mfma_values_kernel, egrid, block, 0, stream_, src, mfma->groups, masks2, d_pos, words + mfma->values_word, d_scalar_);
                ok = check(hipGetLastError(), "mfma_values_kernel");
            }
        }
    }
    ok = ok && check(hipStreamSynchronize(stream_), "bitmap emit");
    timer.lap("gpu: bitmap prefix + values + run heads + second image");
    for (void* p : {static_cast<void*>(d_range), static_cast<void*>(d_prefix), static_cast<void*>(d_flags), static_cast<void*>(d_run_prefix), static_cast<void*>(d_cnt),
                    static_cast<void*>(d_pos), static_cast<void*>(d_blocks), static_cast<void*>(d_runs), static_cast<void*>(d_row_base), static_cast<void*>(d_heads), d_temp})
        if (p) (void)hipFree(p);
    if (!ok || duplicates) {
        if (d_image_) { (void)hipFree(d_image_); d_image_ = nullptr; }
        if (d_mfma_) { (void)hipFree(d_mfma_); d_mfma_ = nullptr; }
    }
    return ok;
}

// ---- SWEEP (sweep_tiles.cpp; kernel: spmv_sweep.hip) ---------------------------------------------------------------------------
namespace {

struct SweepSlices { uint32_t n; uint32_t col[kMaxSweepSlices + 1]; };

__global__ __launch_bounds__(256) void sweep_lines_kernel(ElementSource src, uint32_t* __restrict__ line_nnz, uint32_t* err) {
    visit_elements(src, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, err, [&](uint32_t, uint32_t col, uint32_t) { atomicAdd(line_nnz + col / kSweepColAlign, 1u); });
}

// (the order the elements arrive in does not matter: the sort key is the element's complete identity -- two equal keys are a duplicate entry,
// which the host builder takes)
__global__ __launch_bounds__(256) void sweep_keys_kernel(ElementSource src, const uint32_t* __restrict__ range_of_row, const uint32_t* __restrict__ range_row0,
                                                        SweepSlices slices, unsigned long long* cursor, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint32_t* err) {
    visit_elements(src, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, err, [&](uint32_t row, uint32_t col, uint32_t val) {
        const uint32_t range = range_of_row[row];
        uint32_t k = 0;
        while (k + 1 < slices.n && col >= slices.col[k + 1]) ++k;
        const uint64_t at = atomicAdd(cursor, 1ull);
        keys[at] = (uint64_t(range) * slices.n + k) << 48 | uint64_t(col) << 16 | (row - range_row0[range]);
        vals[at] = val;
    });
}

__global__ __launch_bounds__(256) void sweep_starts_kernel(const uint64_t* __restrict__ keys, uint64_t n, unsigned long long* __restrict__ block_start, uint32_t* flags) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = keys[i] >> 48;
    if (i == 0 || (keys[i - 1] >> 48) != b) block_start[b] = i;
    if (i + 1 < n && keys[i] == keys[i + 1]) flags[0] = 1u;
}

__global__ __launch_bounds__(256) void sweep_spans_kernel(const uint64_t* __restrict__ keys, uint64_t n, const unsigned long long* __restrict__ block_start, uint32_t* flags) {
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t first = block_start[keys[i] >> 48], chunk_first = i - (i - first) % kWaveLanes;
    if (((keys[i] >> 16) & 0xffffffffull) - ((keys[chunk_first] >> 16) & 0xffffffffull) > 0xffffull) flags[1] = 1u;
}

// one thread per element slot of the image: chunk c of all blocks' chunks, lane l
__global__ __launch_bounds__(256) void sweep_emit_kernel(const GpuTiler::SweepBlock* __restrict__ blocks, uint32_t num_blocks, uint64_t total_chunks,
                                                        const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint8_t* __restrict__ image) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, c = t / kWaveLanes;
    const uint32_t lane = uint32_t(t % kWaveLanes);
    if (c >= total_chunks) return;
    uint32_t lo = 0, hi = num_blocks;              // the last block with chunk0 <= c (blocks without chunks share their successor's chunk0: skipped by the search)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (blocks[mid].chunk0 <= c) lo = mid; else hi = mid;
    }
    const GpuTiler::SweepBlock& b = blocks[lo];
    const uint32_t k = uint32_t(c - b.chunk0), s = k / kSweepWaves, w = k % kSweepWaves;
    const uint64_t first = uint64_t(k) * kWaveLanes;
    const uint32_t base = first < b.count ? uint32_t(keys[b.first + first] >> 16) : b.count ? uint32_t(keys[b.first + b.count - 1] >> 16) : b.pad_col;
    uint32_t* slot = reinterpret_cast<uint32_t*>(image + b.stream_at + uint64_t(k) * kChunkBytes) + 2 * lane;
    if (first + lane < b.count) {
        const uint64_t key = keys[b.first + first + lane];
        slot[0] = vals[b.first + first + lane];
        slot[1] = uint32_t(key & 0xffffu) << 16 | (uint32_t(key >> 16) - base);
    } else {
        slot[0] = 0u;
        slot[1] = b.nrows << 16;
    }
    if (lane == 0) reinterpret_cast<uint32_t*>(image + b.table_at)[size_t(w) * b.steps + s] = base;
}

}  // namespace

bool GpuTiler::sweep_line_counts(uint32_t lines, std::vector<uint64_t>& line_nnz) {
    ElementSource src{d_channels_, static_cast<const StreamGroup*>(d_groups_), d_advance_, d_indptr_, d_indices_, d_values_, total_, num_groups_, total_slots_,
                      L_.num_rows, csr_ ? csr_->num_cols : L_.num_cols, uint32_t(geom_.logical_vb), uint32_t(geom_.impl == IMPL_FIXED), csr_ ? 1u : 0u};
    const uint64_t threads = csr_ ? (total_ + kCsrSegment - 1) / kCsrSegment : total_slots_;
    uint32_t* d_cnt = nullptr;
    std::vector<uint32_t> cnt(lines, 0);
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_cnt), std::max<size_t>(size_t(lines) * 4, 16)), "hipMalloc(line counts)") &&
              check(hipMemsetAsync(d_cnt, 0, std::max<size_t>(size_t(lines) * 4, 16), stream_), "hipMemset");
    if (ok && threads) {
        hipLaunchKernelGGL(sweep_lines_kernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream_, src, d_cnt, d_scalar_);
        ok = check(hipGetLastError(), "sweep_lines_kernel");
    }
    uint32_t errw[3] = {0, 0, 0};
    ok = ok && check(hipMemcpyAsync(cnt.data(), d_cnt, size_t(lines) * 4, hipMemcpyDeviceToHost, stream_), "read line counts") &&
         check(hipMemcpyAsync(errw, d_scalar_, 12, hipMemcpyDeviceToHost, stream_), "read flags") && check(hipStreamSynchronize(stream_), "sweep line counts");
    if (ok && errw[0]) ok = fail("CSR row " + std::to_string(uint64_t(errw[1]) * PACK_SIZE + errw[2]) + ": column index outside the matrix");
    if (d_cnt) (void)hipFree(d_cnt);
    line_nnz.assign(cnt.begin(), cnt.end());
    return ok;
}

bool GpuTiler::sweep_sort(const std::vector<uint32_t>& range_of_row, const std::vector<uint32_t>& range_row0, const std::vector<uint32_t>& slice_col,
                          uint32_t num_blocks, std::vector<uint64_t>& block_start, bool& unsupported) {
    detail::PhaseTimer timer;
    unsupported = false;
    ElementSource src{d_channels_, static_cast<const StreamGroup*>(d_groups_), d_advance_, d_indptr_, d_indices_, d_values_, total_, num_groups_, total_slots_,
                      L_.num_rows, csr_ ? csr_->num_cols : L_.num_cols, uint32_t(geom_.logical_vb), uint32_t(geom_.impl == IMPL_FIXED), csr_ ? 1u : 0u};
    const uint64_t threads = csr_ ? (total_ + kCsrSegment - 1) / kCsrSegment : total_slots_;
    SweepSlices sl{};
    sl.n = uint32_t(slice_col.size()) - 1;
    for (uint32_t k = 0; k <= sl.n; ++k) sl.col[k] = slice_col[k];
    const size_t n = std::max<uint64_t>(total_, 1);
    uint32_t *d_range = nullptr, *d_row0 = nullptr, *d_vals_in = nullptr, *d_flags = nullptr;
    uint64_t* d_keys_in = nullptr;
    unsigned long long *d_cursor = nullptr, *d_start = nullptr;
    void* d_temp = nullptr;
    bool ok = check(upload(&d_range, range_of_row, stream_), "upload row ranges") && check(upload(&d_row0, range_row0, stream_), "upload range rows") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_keys_in), n * 8), "hipMalloc(keys)") && check(hipMalloc(reinterpret_cast<void**>(&d_vals_in), n * 4), "hipMalloc(values)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_keys_), n * 8), "hipMalloc(keys)") && check(hipMalloc(reinterpret_cast<void**>(&d_vals_), n * 4), "hipMalloc(values)") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_cursor), 8), "hipMalloc") && check(hipMemsetAsync(d_cursor, 0, 8, stream_), "hipMemset") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_flags), 16), "hipMalloc") && check(hipMemsetAsync(d_flags, 0, 16, stream_), "hipMemset") &&
              check(hipMalloc(reinterpret_cast<void**>(&d_start), (size_t(num_blocks) + 1) * 8), "hipMalloc(block starts)") &&
              check(hipMemsetAsync(d_start, 0xff, (size_t(num_blocks) + 1) * 8, stream_), "hipMemset");
    if (ok && threads) {
        hipLaunchKernelGGL(sweep_keys_kernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream_, src, d_range, d_row0, sl, d_cursor, d_keys_in, d_vals_in, d_scalar_);
        ok = check(hipGetLastError(), "sweep_keys_kernel");
    }
    if (timer.on) { (void)hipStreamSynchronize(stream_); timer.lap("gpu: sweep keys"); }
    if (ok && total_) {
        uint32_t block_bits = 1;
        while ((uint64_t(1) << block_bits) < num_blocks) ++block_bits;
        size_t temp_bytes = 0;
        const int end_bit = int(48 + block_bits);
        ok = end_bit <= 64 && check(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys_in, d_keys_, d_vals_in, d_vals_, total_, 0, end_bit, stream_), "radix sort (size)") &&
             check(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), "hipMalloc(sort)") &&
             check(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys_in, d_keys_, d_vals_in, d_vals_, total_, 0, end_bit, stream_), "radix sort");
        if (ok) {
            const dim3 grid(uint32_t((total_ + 255) / 256));
            hipLaunchKernelGGL(sweep_starts_kernel, grid, dim3(256), 0, stream_, d_keys_, total_, d_start, d_flags);
            hipLaunchKernelGGL(sweep_spans_kernel, grid, dim3(256), 0, stream_, d_keys_, total_, d_start, d_flags);
            ok = check(hipGetLastError(), "sweep_starts_kernel");
        }
    }
    std::vector<unsigned long long> starts(size_t(num_blocks) + 1, ~0ull);
    uint32_t flags[4] = {0, 0, 0, 0};
    unsigned long long placed = 0;
    ok = ok && check(hipMemcpyAsync(starts.data(), d_start, starts.size() * 8, hipMemcpyDeviceToHost, stream_), "read block starts") &&
         check(hipMemcpyAsync(flags, d_flags, 16, hipMemcpyDeviceToHost, stream_), "read flags") &&
         check(hipMemcpyAsync(&placed, d_cursor, 8, hipMemcpyDeviceToHost, stream_), "read cursor") && check(hipStreamSynchronize(stream_), "sweep sort");
    if (ok && placed != total_) ok = fail("gpu sweep: element count changed between passes");
    timer.lap("gpu: sweep radix sort");
    unsupported = ok && (flags[0] != 0 || flags[1] != 0);
    block_start.assign(size_t(num_blocks) + 1, total_);
    for (uint32_t b = num_blocks; b-- > 0;) block_start[b] = starts[b] == ~0ull ? block_start[b + 1] : starts[b];      // blocks without elements start where the next one does
    for (void* p : {static_cast<void*>(d_range), static_cast<void*>(d_row0), static_cast<void*>(d_keys_in), static_cast<void*>(d_vals_in), static_cast<void*>(d_flags),
                    static_cast<void*>(d_cursor), static_cast<void*>(d_start), d_temp})
        if (p) (void)hipFree(p);
    if (d_channels_) { (void)hipFree(d_channels_); d_channels_ = nullptr; }      // the source is not needed any more
    for (void** p : {reinterpret_cast<void**>(&d_indptr_), reinterpret_cast<void**>(&d_indices_), reinterpret_cast<void**>(&d_values_)})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    return ok;
}

bool GpuTiler::sweep_emit(const std::vector<SweepBlock>& blocks, uint64_t image_bytes, uint64_t slack_bytes) {
    detail::PhaseTimer timer;
    const size_t bytes = std::max<uint64_t>(image_bytes + slack_bytes, 256);
    const uint64_t total_chunks = blocks.empty() ? 0 : blocks.back().chunk0 + uint64_t(blocks.back().steps) * kSweepWaves;
    SweepBlock* d_blocks = nullptr;
    bool ok = check(hipMalloc(reinterpret_cast<void**>(&d_image_), bytes), "hipMalloc(image)") && check(upload(&d_blocks, blocks, stream_), "upload blocks");
    if (ok && slack_bytes) ok = check(hipMemsetAsync(d_image_ + image_bytes, 0, slack_bytes, stream_), "hipMemset(slack)");
    if (ok && total_chunks) {
        hipLaunchKernelGGL(sweep_emit_kernel, dim3(uint32_t((total_chunks * kWaveLanes + 255) / 256)), dim3(256), 0, stream_, d_blocks, uint32_t(blocks.size()), total_chunks,
                           d_keys_, d_vals_, d_image_);
        ok = check(hipGetLastError(), "sweep_emit_kernel");
    }
    ok = ok && check(hipStreamSynchronize(stream_), "sweep emit");
    timer.lap("gpu: sweep emit");
    if (d_blocks) (void)hipFree(d_blocks);
    if (!ok && d_image_) { (void)hipFree(d_image_); d_image_ = nullptr; }
    return ok;
}

}  // namespace dev
}  // namespace hisparse
