// host_capi.cpp — libhisparse_host.so: C-ABI over the C++ host library (include/hisparse/*.h).
// Declared in include/hisparse_host.h.  Plain C++17, no GPU code.
#include "hisparse_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "hisparse/channel_packets.h"
#include "hisparse/common.h"
#include "hisparse/data_formatter.h"
#include "hisparse/data_loader.h"
#include "hisparse/row_sharding.h"

using spmv::io::CSRMatrix;

struct hsf_csr {
    CSRMatrix<float> m;
};
struct hsf_matrix {
    hisparse::ChannelPackets p;
};

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

template <typename Fn>
int guarded(Fn fn) {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return fail(HSF_NO_MEMORY, "out of memory");
    } catch (const std::invalid_argument& e) {
        return fail(HSF_BAD_ARG, e.what());
    } catch (const std::exception& e) {
        return fail(HSF_FORMAT_ERROR, e.what());
    }
}

// ---- small deterministic RNG (splitmix64), one stream per (seed, row) -------------------------
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};
uint64_t mix(uint64_t a, uint64_t b) {
    Rng r(a ^ (b * 0xd6e8feb86659fd93ull));
    return r.next();
}

// multiplier coprime to n, for a cheap bijection i -> (i * A) % n
uint64_t coprime_multiplier(uint64_t n, uint64_t hint) {
    if (n <= 2) return 1;
    uint64_t a = hint % n;
    if (a < 2) a = 2;
    auto gcd = [](uint64_t x, uint64_t y) { while (y) { uint64_t t = x % y; x = y; y = t; } return x; };
    while (gcd(a, n) != 1) ++a;
    return a;
}

template <typename RowFn>
void build_rows_parallel(CSRMatrix<float>& m, RowFn row_fn) {
    // row_fn(row, cols_out, vals_out) appends one row; rows are processed in contiguous blocks per thread
    const uint32_t rows = m.num_rows;
    unsigned threads = std::max(1u, std::thread::hardware_concurrency());
    threads = std::min<unsigned>(threads, std::max<uint32_t>(1u, rows / 64u));
    std::vector<std::vector<uint32_t>> cols(threads);
    std::vector<std::vector<float>> vals(threads);
    std::vector<std::vector<uint32_t>> lens(threads);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            uint32_t lo = uint32_t(uint64_t(rows) * t / threads), hi = uint32_t(uint64_t(rows) * (t + 1) / threads);
            lens[t].reserve(hi - lo);
            for (uint32_t r = lo; r < hi; ++r) {
                size_t before = cols[t].size();
                row_fn(r, cols[t], vals[t]);
                lens[t].push_back(uint32_t(cols[t].size() - before));
            }
        });
    for (auto& th : pool) th.join();
    m.adj_indptr.assign(size_t(rows) + 1, 0);
    size_t r = 0;
    for (unsigned t = 0; t < threads; ++t)
        for (uint32_t l : lens[t]) { m.adj_indptr[r + 1] = m.adj_indptr[r] + l; ++r; }
    m.adj_indices.resize(m.adj_indptr[rows]);
    m.adj_data.resize(m.adj_indptr[rows]);
    size_t at = 0;
    for (unsigned t = 0; t < threads; ++t) {
        std::copy(cols[t].begin(), cols[t].end(), m.adj_indices.begin() + at);
        std::copy(vals[t].begin(), vals[t].end(), m.adj_data.begin() + at);
        at += cols[t].size();
    }
}

}  // namespace

extern "C" {

const char* hsf_last_error(void) { return g_error.c_str(); }

int hsf_csr_load_npz(const char* path, hsf_csr** out) {
    if (!path || !out) return fail(HSF_BAD_ARG, "null argument");
    try {
        std::unique_ptr<hsf_csr> h(new hsf_csr);
        h->m = spmv::io::load_csr_matrix_from_float_npz(path);
        *out = h.release();
        return HSF_OK;
    } catch (const std::bad_alloc&) {
        return fail(HSF_NO_MEMORY, "out of memory");
    } catch (const std::exception& e) {
        return fail(HSF_IO_ERROR, e.what());
    }
}

int hsf_csr_from_arrays(uint32_t num_rows, uint32_t num_cols, uint64_t nnz, const uint32_t* indptr, const uint32_t* indices,
                        const float* data, hsf_csr** out) {
    if (!indptr || !out || (nnz && (!indices || !data))) return fail(HSF_BAD_ARG, "null argument");
    return guarded([&]() {
        if (indptr[0] != 0 || indptr[num_rows] != nnz) return fail(HSF_BAD_ARG, "indptr must start at 0 and end at nnz");
        for (uint32_t r = 0; r < num_rows; ++r)
            if (indptr[r + 1] < indptr[r]) return fail(HSF_BAD_ARG, "indptr must be non-decreasing");
        for (uint64_t e = 0; e < nnz; ++e)
            if (indices[e] >= num_cols) return fail(HSF_BAD_ARG, "column index out of range");
        hsf_csr* h = new hsf_csr;
        h->m.num_rows = num_rows;
        h->m.num_cols = num_cols;
        h->m.adj_indptr.assign(indptr, indptr + num_rows + 1);
        h->m.adj_indices.assign(indices, indices + nnz);
        h->m.adj_data.assign(data, data + nnz);
        *out = h;
        return int(HSF_OK);
    });
}

int hsf_csr_dims(const hsf_csr* m, uint32_t* num_rows, uint32_t* num_cols, uint64_t* nnz) {
    if (!m) return fail(HSF_BAD_ARG, "null handle");
    if (num_rows) *num_rows = m->m.num_rows;
    if (num_cols) *num_cols = m->m.num_cols;
    if (nnz) *nnz = m->m.adj_data.size();
    return HSF_OK;
}

int hsf_csr_copy(const hsf_csr* m, uint32_t* indptr, uint32_t* indices, float* data) {
    if (!m) return fail(HSF_BAD_ARG, "null handle");
    if (indptr) std::copy(m->m.adj_indptr.begin(), m->m.adj_indptr.end(), indptr);
    if (indices) std::copy(m->m.adj_indices.begin(), m->m.adj_indices.end(), indices);
    if (data) std::copy(m->m.adj_data.begin(), m->m.adj_data.end(), data);
    return HSF_OK;
}

int hsf_csr_fill(hsf_csr* m, float value) {
    if (!m) return fail(HSF_BAD_ARG, "null handle");
    std::fill(m->m.adj_data.begin(), m->m.adj_data.end(), value);
    return HSF_OK;
}

int hsf_csr_normalize_by_outdegree(hsf_csr* m) {
    if (!m) return fail(HSF_BAD_ARG, "null handle");
    spmv::io::util_normalize_csr_matrix_by_outdegree(m->m);
    return HSF_OK;
}

void hsf_csr_free(hsf_csr* m) { delete m; }

int hsf_csr_generate(const char* kind, uint32_t num_rows, uint32_t num_cols, double a, double b, double c, uint64_t seed,
                     hsf_csr** out) {
    if (!kind || !out || num_rows == 0 || num_cols == 0) return fail(HSF_BAD_ARG, "bad generator arguments");
    const std::string k(kind);
    return guarded([&]() {
        std::unique_ptr<hsf_csr> h(new hsf_csr);   // released only on success: a throwing generator must not leak it
        CSRMatrix<float>& m = h->m;
        m.num_rows = num_rows;
        m.num_cols = num_cols;
        if (k == "dense") {
            build_rows_parallel(m, [&](uint32_t, std::vector<uint32_t>& cols, std::vector<float>& vals) {
                for (uint32_t j = 0; j < num_cols; ++j) { cols.push_back(j); vals.push_back(1.0f); }
            });
        } else if (k == "uniform") {
            // column of the j-th non-zero of row i = (floor(cols / nnz_per_row) * j + i) % cols, value 1
            const uint32_t per_row = uint32_t(a);
            if (per_row == 0 || per_row > num_cols) { return fail(HSF_BAD_ARG, "uniform: bad nnz_per_row"); }
            const uint32_t step = num_cols / per_row;
            build_rows_parallel(m, [&](uint32_t i, std::vector<uint32_t>& cols, std::vector<float>& vals) {
                for (uint32_t j = 0; j < per_row; ++j) { cols.push_back(uint32_t((uint64_t(step) * j + i) % num_cols)); vals.push_back(1.0f); }
            });
        } else if (k == "bernoulli") {
            if (!(b > 0.0 && b <= 1.0)) { return fail(HSF_BAD_ARG, "bernoulli: density must be in (0,1]"); }
            build_rows_parallel(m, [&](uint32_t i, std::vector<uint32_t>& cols, std::vector<float>& vals) {
                Rng rng(mix(seed, i));
                for (uint32_t j = 0; j < num_cols; ++j)
                    if (rng.uniform() < b) { cols.push_back(j); vals.push_back(float(rng.normal() * c)); }
            });
        } else if (k == "powerlaw") {
            const double target = a, beta = b;
            if (!(target > 0) || !(beta >= 0.0 && beta < 1.0)) { return fail(HSF_BAD_ARG, "powerlaw: need nnz > 0 and 0 <= beta < 1"); }
            // node weight w(rank) = (rank+1)^-beta; rank = bijective scramble of the id so hubs are spread out
            const uint64_t row_mul = coprime_multiplier(num_rows, 0x9e3779b1ull + seed * 7919u);
            const uint64_t col_mul = coprime_multiplier(num_cols, 0x85ebca6bull + seed * 104729u);
            double wsum = 0.0;
            for (uint32_t r = 0; r < num_rows; ++r) wsum += std::pow(double(r) + 1.0, -beta);
            const double inv_exp = 1.0 / (1.0 - beta);
            build_rows_parallel(m, [&](uint32_t i, std::vector<uint32_t>& cols, std::vector<float>& vals) {
                Rng rng(mix(seed, i));
                const uint64_t rank = (uint64_t(i) * row_mul) % num_rows;
                double want = target * std::pow(double(rank) + 1.0, -beta) / wsum;
                uint32_t deg = uint32_t(want);
                if (rng.uniform() < want - deg) ++deg;  // stochastic rounding keeps the total on target
                deg = std::min<uint32_t>(deg, num_cols);
                std::vector<uint32_t> pick;
                pick.reserve(deg + deg / 8 + 4);
                for (int attempt = 0; attempt < 6 && pick.size() < deg; ++attempt) {
                    size_t need = deg - pick.size();
                    for (size_t d = 0; d < need; ++d) {
                        // inverse CDF of a density ~ k^-beta on [0, cols)
                        uint64_t crank = uint64_t(double(num_cols) * std::pow(rng.uniform(), inv_exp));
                        if (crank >= num_cols) crank = num_cols - 1;
                        pick.push_back(uint32_t((crank * col_mul) % num_cols));
                    }
                    std::sort(pick.begin(), pick.end());
                    pick.erase(std::unique(pick.begin(), pick.end()), pick.end());
                }
                for (uint32_t col : pick) { cols.push_back(col); vals.push_back(float(rng.uniform() * c)); }
            });
        } else if (k == "rmat") {
            // Symmetric R-MAT graph (a = .57, b = .19, c = .19, d = .05: the Graph500 / SURVEY.md section 8d parameters): `b` = 1 scrambles the vertex ids with a bijection (0 keeps the recursive order: hubs at the
            // low ids), values uniform(0, c); `a` counts the entries DRAWN, repeats are dropped afterwards.  Edges are drawn in 1024 fixed chunks with their own RNG streams, so the matrix is the
            // same for every thread count; (i, j) and (j, i) are both stored, duplicates and nothing else are dropped.
            if (!(a > 0) || num_rows != num_cols) return fail(HSF_BAD_ARG, "rmat: need nnz > 0 and a square matrix");
            const uint32_t n = num_rows;
            uint32_t scale = 0;
            while ((uint64_t(1) << scale) < n) ++scale;
            const uint64_t mul = b != 0.0 ? coprime_multiplier(n, 0x9e3779b1ull + seed * 7919u) : 1;
            const uint64_t pairs = uint64_t(a / 2.0) + 1;
            constexpr uint32_t kChunks = 1024;
            std::vector<std::vector<uint64_t>> drawn(kChunks);
            {
                std::atomic<uint32_t> next(0);
                std::vector<std::thread> pool;
                const unsigned threads = std::max(1u, std::thread::hardware_concurrency());
                for (unsigned t = 0; t < threads; ++t)
                    pool.emplace_back([&]() {
                        for (uint32_t ch = next.fetch_add(1); ch < kChunks; ch = next.fetch_add(1)) {
                            Rng rng(mix(seed, 0x726d6174ull + ch));
                            const uint64_t lo = pairs * ch / kChunks, hi = pairs * (ch + 1) / kChunks;
                            auto& out = drawn[ch];
                            out.reserve(size_t(hi - lo) * 2);
                            for (uint64_t e = lo; e < hi; ++e) {
                                uint32_t i = 0, j = 0;
                                for (uint32_t bit = 0; bit < scale; ++bit) {
                                    const double u = rng.uniform();
                                    const uint32_t q = u < 0.57 ? 0u : u < 0.76 ? 1u : u < 0.95 ? 2u : 3u;      // a | b | c | d
                                    i = (i << 1) | (q >> 1);
                                    j = (j << 1) | (q & 1u);
                                }
                                if (i >= n || j >= n || i == j) continue;
                                const uint64_t si = uint64_t(i) * mul % n, sj = uint64_t(j) * mul % n;
                                out.push_back((si << 32) | sj);
                                out.push_back((sj << 32) | si);
                            }
                        }
                    });
                for (auto& th : pool) th.join();
            }
            // bucket by row, sort + unique per row
            std::vector<uint64_t> count(size_t(n) + 1, 0);
            for (const auto& v : drawn)
                for (uint64_t e : v) count[(e >> 32) + 1]++;
            for (uint32_t r = 0; r < n; ++r) count[r + 1] += count[r];
            std::vector<uint32_t> cols(count[n]);
            {
                std::vector<uint64_t> cursor(count.begin(), count.end() - 1);
                for (auto& v : drawn) {
                    for (uint64_t e : v) cols[cursor[e >> 32]++] = uint32_t(e);
                    std::vector<uint64_t>().swap(v);
                }
            }
            build_rows_parallel(m, [&](uint32_t i, std::vector<uint32_t>& out_cols, std::vector<float>& out_vals) {
                std::vector<uint32_t> mine(cols.begin() + count[i], cols.begin() + count[i + 1]);
                std::sort(mine.begin(), mine.end());
                mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
                Rng rng(mix(seed, i));
                for (uint32_t col : mine) { out_cols.push_back(col); out_vals.push_back(float(rng.uniform() * c)); }
            });
        } else {
            return fail(HSF_BAD_ARG, "unknown generator kind: " + k);
        }
        *out = h.release();
        return int(HSF_OK);
    });
}

int hsf_format(hsf_csr* m, int impl, uint32_t ob_bank, uint32_t vb_bank, int skip_empty_rows, hsf_matrix** out) {
    if (!m || !out) return fail(HSF_BAD_ARG, "null argument");
    if (!hisparse::impl_valid(impl)) return fail(HSF_BAD_ARG, "impl must be 0 (fixed), 1 (float_pob) or 2 (float_stall)");
    if (ob_bank == 0 || vb_bank == 0) return fail(HSF_BAD_ARG, "bank sizes must be positive");
    return guarded([&]() {
        hisparse::Geometry g = hisparse::make_geometry(impl, ob_bank, vb_bank);
        hsf_matrix* h = new hsf_matrix;
        try {
            h->p = hisparse::format_matrix(m->m, g, skip_empty_rows != 0);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
        return int(HSF_OK);
    });
}

int hsf_matrix_get_info(const hsf_matrix* m, hsf_matrix_info* info) {
    if (!m || !info) return fail(HSF_BAD_ARG, "null argument");
    const auto& p = m->p;
    info->impl = p.geom.impl;
    info->interleave = p.geom.interleave;
    info->ob_bank = p.geom.ob_bank;
    info->vb_bank = p.geom.vb_bank;
    info->num_rows = p.num_rows;
    info->num_cols = p.num_cols;
    info->num_row_partitions = p.num_row_partitions;
    info->num_col_partitions = p.num_col_partitions;
    info->nnz = p.nnz;
    info->streamed_bytes = p.streamed_bytes();
    info->skip_empty_rows = p.skip_empty_rows ? 1 : 0;
    return HSF_OK;
}

int hsf_matrix_channel(const hsf_matrix* m, uint32_t c, const void** packets, uint64_t* num_packets) {
    if (!m || c >= hisparse::NUM_HBM_CHANNELS) return fail(HSF_BAD_ARG, "bad channel");
    if (packets) *packets = m->p.channel[c].data();
    if (num_packets) *num_packets = m->p.channel[c].size();
    return HSF_OK;
}

int hsf_matrix_part_len(const hsf_matrix* m, uint32_t row_partition, uint32_t* part_len) {
    if (!m || !part_len || row_partition >= m->p.num_row_partitions) return fail(HSF_BAD_ARG, "bad row partition");
    *part_len = m->p.part_len(row_partition);
    return HSF_OK;
}

void hsf_matrix_free(hsf_matrix* m) { delete m; }

int hsf_pack_vector(int impl, const float* x, uint64_t n, uint32_t* words) {
    if (!hisparse::impl_valid(impl) || (n && (!x || !words))) return fail(HSF_BAD_ARG, "bad argument");
    hisparse::pack_vector(impl, x, n, words);
    return HSF_OK;
}

int hsf_unpack_result(int impl, const uint32_t* words, uint64_t n, float* y) {
    if (!hisparse::impl_valid(impl) || (n && (!y || !words))) return fail(HSF_BAD_ARG, "bad argument");
    hisparse::unpack_result(impl, words, n, y);
    return HSF_OK;
}

int hsf_csr_to_csc(const hsf_csr* m, int impl, uint32_t* indptr, uint32_t* row_indices, uint32_t* value_words) {
    if (!m || !indptr || !hisparse::impl_valid(impl)) return fail(HSF_BAD_ARG, "bad argument");
    return guarded([&]() {
        // sw/data_loader.h:109-144 (csr2csc) + :147-157 (csc_matrix_convert_from_float): rows stay ascending inside a column
        const spmv::io::CSCMatrix<float> csc = spmv::io::csr2csc(m->m);
        std::copy(csc.adj_indptr.begin(), csc.adj_indptr.end(), indptr);
        if (row_indices) std::copy(csc.adj_indices.begin(), csc.adj_indices.end(), row_indices);
        if (value_words) hisparse::pack_vector(impl, csc.adj_data.data(), csc.adj_data.size(), value_words);
        return int(HSF_OK);
    });
}

int hsf_split_rows_by_nnz(const uint32_t* indptr, uint32_t num_rows, uint32_t parts, uint32_t granule, uint32_t* bounds) {
    if (!indptr || !bounds || parts == 0 || granule == 0) return fail(HSF_BAD_ARG, "bad argument");
    return guarded([&]() {
        const std::vector<uint32_t> ip(indptr, indptr + size_t(num_rows) + 1);
        const std::vector<uint32_t> b = hisparse::split_rows_by_nnz(ip, parts, granule);
        std::copy(b.begin(), b.end(), bounds);
        return int(HSF_OK);
    });
}

}  // extern "C"
