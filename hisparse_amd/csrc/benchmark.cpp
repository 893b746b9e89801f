// benchmark.cpp — the benchmark entry of the reference (sw/benchmark.cpp), on the HIP C-ABI.
//
// Same shape as the reference's driver: positional arguments, pre-processing timed separately, NUM_RUNS = 50
// back-to-back SpMVs timed with a host clock, and the one-line result
//     {Preprocessing: <s> s | SpMV: <ms> ms | <GBPS> GBPS | <GOPS> GOPS }
// (sw/benchmark.cpp:80-87,311-346; formulas restated in 64-bit).
//
//   reference:  ./benchmark <hw-xclbin> <dataset> <v> <o>           (:355-365)
//   here:       ./benchmark <impl>      <dataset> <v> <o> [device]
// The bitstream argument selected the numeric mode (one xclbin per IMPL, sw/Makefile:2-12); here the mode is
// named directly: fixed | float_pob | float_stall.  <dataset> is a scipy .npz, or synth:<kind>:<rows>:<cols>:<a>:<b>:<c>:<seed>
// for the generators of libhisparse_host (hsf_csr_generate).  <v>/<o> are the bank sizes in K words.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "hisparse/channel_packets.h"
#include "hisparse/data_formatter.h"
#include "hisparse/data_loader.h"
#include "hisparse_hip.h"
#include "hisparse_host.h"

namespace {

const unsigned NUM_RUNS = 50;  // sw/benchmark.cpp:29

struct benchmark_result {
    double preprocess_time_s;
    double spmv_time_ms;
    double throughput_GBPS;
    double throughput_GOPS;
};

std::ostream& operator<<(std::ostream& os, const benchmark_result& p) {
    os << '{' << "Preprocessing: " << p.preprocess_time_s << " s | "
       << "SpMV: " << p.spmv_time_ms << " ms | " << p.throughput_GBPS << " GBPS | " << p.throughput_GOPS << " GOPS }";
    return os;
}

// the reference's OCL_CHECK / CHECK_ERR convention: print and exit (xcl2.hpp:40-46, benchmark.cpp:56-61)
void check(int rc, hs_context* ctx, const char* what) {
    if (rc == HS_OK) return;
    std::printf("HS Error at %s: %s (%s)\n", what, hs_strerror(rc), hs_last_error(ctx));
    std::exit(EXIT_FAILURE);
}

int parse_impl(const std::string& s) {
    if (s == "fixed") return hisparse::IMPL_FIXED;
    if (s == "float_pob") return hisparse::IMPL_FLOAT_POB;
    if (s == "float_stall") return hisparse::IMPL_FLOAT_STALL;
    return -1;
}

benchmark_result spmv_benchmark(int impl, unsigned vb_bank_size, unsigned ob_bank_size, int device,
                                spmv::io::CSRMatrix<float>& ext_matrix, bool skip_empty_rows) {
    using clock = std::chrono::steady_clock;
    benchmark_result res{};
    std::cout << "INFO : Test started" << std::endl;
    const auto t0 = clock::now();
    hisparse::Geometry g = hisparse::make_geometry(impl, ob_bank_size, vb_bank_size);
    hisparse::ChannelPackets packets = hisparse::format_matrix(ext_matrix, g, skip_empty_rows);
    const auto t1 = clock::now();
    res.preprocess_time_s = std::chrono::duration<double>(t1 - t0).count();
    std::cout << "INFO : Matrix loading/preprocessing complete!" << std::endl;
    std::cout << "  row_partitions: " << packets.num_row_partitions << std::endl;
    std::cout << "  col_partitions: " << packets.num_col_partitions << std::endl;

    // x = rand() % 2 like the reference (benchmark.cpp:205-212), packed by the value type's converting constructor
    std::vector<float> vector_f(packets.num_cols);
    for (auto& v : vector_f) v = float(std::rand() % 2);
    std::vector<uint32_t> vector(packets.num_cols), result(packets.num_rows, 0);
    hisparse::pack_vector(impl, vector_f.data(), vector_f.size(), vector.data());
    std::cout << "INFO : Input/result initialization complete!" << std::endl;

    hs_context* ctx = nullptr;
    check(hs_create(&ctx, device, impl, ob_bank_size, vb_bank_size), nullptr, "hs_create");
    const void* ch[HS_NUM_CHANNELS];
    uint64_t n[HS_NUM_CHANNELS];
    for (unsigned c = 0; c < HS_NUM_CHANNELS; ++c) {
        ch[c] = packets.channel[c].data();
        n[c] = packets.channel[c].size();
    }
    check(hs_load_matrix(ctx, ch, n, packets.num_rows, packets.num_cols, packets.num_row_partitions, packets.num_col_partitions), ctx, "hs_load_matrix");
    check(hs_load_vector(ctx, vector.data(), packets.num_cols), ctx, "hs_load_vector");
    std::cout << "INFO : Host -> Device data transfer complete!" << std::endl;
    hs_stats st;
    hs_get_stats(ctx, &st);
    std::cout << "  device load (decode + re-tile + H2D): " << st.load_seconds << " s, " << st.num_blocks << " row blocks on "
              << st.num_workgroups << " workgroups, stream " << st.stream_bytes / 1e6 << " MB" << std::endl;

    std::cout << "INFO : Invoking kernel:" << std::endl;
    for (int i = 0; i < 5; ++i) check(hs_run(ctx), ctx, "hs_run");  // untimed warm-ups (the reference has none)
    check(hs_sync(ctx), ctx, "hs_sync");
    double total_ms = 0;
    for (unsigned i = 0; i < NUM_RUNS; ++i) {
        const auto a = clock::now();
        check(hs_run(ctx), ctx, "hs_run");    // every row partition (benchmark.cpp:318-339) in one launch
        check(hs_sync(ctx), ctx, "hs_sync");  // queue.finish()
        total_ms += std::chrono::duration<double, std::milli>(clock::now() - a).count();
    }
    const double nnz = double(packets.nnz);
    res.spmv_time_ms = total_ms / NUM_RUNS;
    res.throughput_GBPS = nnz * 8.0 / 1024.0 / 1024.0 / 1024.0 / (res.spmv_time_ms / 1000.0);  // GiB/s, as the reference prints
    res.throughput_GOPS = 2.0 * nnz / 1e6 / res.spmv_time_ms;
    float ev_ms = 0, k_ms = 0;
    check(hs_time_runs(ctx, 0, int(NUM_RUNS), &ev_ms, &k_ms), ctx, "hs_time_runs");
    std::cout << "  device-side: " << ev_ms / NUM_RUNS << " ms per SpMV back-to-back, kernel alone " << k_ms / NUM_RUNS << " ms = "
              << nnz * 8.0 / (k_ms / NUM_RUNS * 1e-3) / 1e9 << " GB/s = " << nnz * 8.0 / (k_ms / NUM_RUNS * 1e-3) / 8e12 * 100
              << " % of the 8 TB/s HBM roofline" << std::endl;
    check(hs_read_result(ctx, result.data(), packets.num_rows), ctx, "hs_read_result");
    hs_destroy(ctx);
    return res;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 5 && argc != 6) {
        std::cout << "Usage: " << argv[0] << " <fixed|float_pob|float_stall> <dataset.npz | synth:kind:rows:cols:a:b:c:seed> <v> <o> [device]" << std::endl;
        return 0;
    }
    const int impl = parse_impl(argv[1]);
    if (impl < 0) {
        std::cout << "ERROR : unknown implementation " << argv[1] << std::endl;
        return 1;
    }
    const std::string dataset = argv[2];
    const unsigned vb_bank_size = unsigned(std::atoi(argv[3])) * 1024;  // benchmark.cpp:364-365
    const unsigned ob_bank_size = unsigned(std::atoi(argv[4])) * 1024;
    const int device = argc == 6 ? std::atoi(argv[5]) : 0;

    std::cout << "------ Running benchmark on " << dataset << std::endl;
    spmv::io::CSRMatrix<float> mat_f;
    try {
        if (dataset.rfind("synth:", 0) == 0) {
            // synth:kind:rows:cols:a:b:c:seed -> the seeded generators of libhisparse_host (hsf_csr_generate)
            std::vector<std::string> f;
            std::stringstream ss(dataset);
            for (std::string tok; std::getline(ss, tok, ':');) f.push_back(tok);
            if (f.size() != 8) { std::cout << "ERROR : expected synth:kind:rows:cols:a:b:c:seed" << std::endl; return 1; }
            hsf_csr* h = nullptr;
            if (hsf_csr_generate(f[1].c_str(), uint32_t(std::stoul(f[2])), uint32_t(std::stoul(f[3])), std::stod(f[4]), std::stod(f[5]),
                                 std::stod(f[6]), std::stoull(f[7]), &h) != HSF_OK) {
                std::cout << "ERROR : " << hsf_last_error() << std::endl;
                return 1;
            }
            uint64_t nnz = 0;
            hsf_csr_dims(h, &mat_f.num_rows, &mat_f.num_cols, &nnz);
            mat_f.adj_indptr.resize(size_t(mat_f.num_rows) + 1);
            mat_f.adj_indices.resize(nnz);
            mat_f.adj_data.resize(nnz);
            hsf_csr_copy(h, mat_f.adj_indptr.data(), mat_f.adj_indices.data(), mat_f.adj_data.data());
            hsf_csr_free(h);
            std::cout << spmv_benchmark(impl, vb_bank_size, ob_bank_size, device, mat_f, true) << std::endl;
            std::cout << "===== Benchmark Finished =====" << std::endl;
            return 0;
        }
        mat_f = spmv::io::load_csr_matrix_from_float_npz(dataset);
    } catch (const std::exception& e) {
        std::cout << "ERROR : " << e.what() << std::endl;
        return 1;
    }
    // The reference sets every value to `1 / num_cols`, which is integer division and yields 0.0 (benchmark.cpp:411);
    // the intent, a small non-degenerate constant, is used here so that the result is worth reading back.
    for (auto& v : mat_f.adj_data) v = 1.0f / float(mat_f.num_cols);
    std::cout << spmv_benchmark(impl, vb_bank_size, ob_bank_size, device, mat_f, true) << std::endl;
    std::cout << "===== Benchmark Finished =====" << std::endl;
    return 0;
}
