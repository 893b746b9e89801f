// benchmark.cpp — the benchmark entry of the reference (sw/benchmark.cpp), on the HIP C-ABI.
//
// Same shape as the reference's driver: positional arguments, pre-processing timed separately, NUM_RUNS = 50
// back-to-back SpMVs timed with a host clock, and the one-line result
//     {Preprocessing: <s> s | SpMV: <ms> ms | <GBPS> GBPS | <GOPS> GOPS }
// (sw/benchmark.cpp:80-87,311-346; formulas restated in 64-bit).
//
//   reference:  ./benchmark <hw-xclbin> <dataset> <v> <o>           (:355-365)
//   here:       ./benchmark <impl>      <dataset> <v> <o> [device] [options]
// The bitstream argument selected the numeric mode (one xclbin per IMPL, sw/Makefile:2-12); here the mode is
// named directly: fixed | float_pob | float_stall.  <dataset> is a scipy .npz, or synth:<kind>:<rows>:<cols>:<a>:<b>:<c>:<seed>
// for the generators of libhisparse_host (hsf_csr_generate).  <v>/<o> are the bank sizes in K words.
//
// Options (none of them exists in the reference; defaults reproduce round 1's behaviour):
//   --values literal|intent|keep   literal: every value := `1 / num_cols` in INTEGER arithmetic = 0.0, exactly what
//                                  sw/benchmark.cpp:411 does (timing parity; y is all zeros); intent (default for .npz):
//                                  1.0f / num_cols; keep (default for synth:): the data set's own values
//   --from-csr                     load with hs_load_matrix_csr (pad + convert + re-tile on the device) instead of csr2cpsr + hs_load_matrix
//   --partition-loop               time the reference's literal launch loop: one hs_run_partition + hs_sync (= finish())
//                                  per row partition (:318-338) instead of one launch for the whole SpMV
//   --runs K                       NUM_RUNS (default 50, :29)
//   --dump-x FILE / --dump-y FILE  raw little-endian u32 value words of the packed x / y (for the parity test)
//   --verify [--verify-eps E]      the check of the reference's hardware harness (sw/host.cpp:33-74,370; spmv_csim/csim.cpp:143-184): y is read
//                                  back, converted to float, and compared with a float32 CSR loop over the matrix as loaded (before the
//                                  conversion to the device's value type) and the float x, |difference| < 1e-4 ABSOLUTE (E replaces it);
//                                  prints the verdict, the first failing row and the largest difference; exit code 3 on failure
//   --share-gpu                    with --gpus N: all N contexts on ONE device (the [device] argument) -- the dry run of the row-slab path on a
//                                  single-GPU machine: slabs, streams, result binding and the gather by peer stores (hs_push_result) all run;
//                                  only RCCL is left out (it cannot place two ranks on one device).  Never a measurement.
//   --gpus N [--no-gather] [--peer-gather]   (--peer-gather: also time the gather as peer stores over xGMI, hs_push_result, no collective)
//   --gpus N [--no-gather]         shard the matrix by row slabs (hisparse/row_sharding.h) over devices 0..N-1 of this node:
//                                  one hs_context per device, this one host thread issuing to the N streams, and one
//                                  ncclAllGather (RCCL over xGMI) of the y slabs per SpMV; both the compute-only and the
//                                  compute + gather times are reported
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "hisparse/channel_packets.h"
#include "hisparse/data_formatter.h"
#include "hisparse/data_loader.h"
#include "hisparse/row_sharding.h"
#include "hisparse_hip.h"
#include "hisparse_host.h"

namespace {

struct Options {
    int impl = -1;
    std::string dataset;
    unsigned vb_bank_size = 0, ob_bank_size = 0;
    int device = 0;
    unsigned runs = 50;  // NUM_RUNS, sw/benchmark.cpp:29
    std::string values;  // literal | intent | keep ("" = default for the dataset kind)
    bool partition_loop = false;
    bool from_csr = false;     // --from-csr: hs_load_matrix_csr instead of csr2cpsr + hs_load_matrix (single GPU)
    std::string dump_x, dump_y;
    int gpus = 1;
    bool gather = true;
    bool peer_gather = false;   // --peer-gather: ALSO time the gather as peer stores over xGMI (hs_push_result) instead of ncclAllGather
    bool sharded = false;   // take the multi-GPU path even with one GPU (exercises RCCL on a single-GPU box)
    bool share_gpu = false; // --gpus N on ONE device: N contexts, N streams, gather by peer stores; no RCCL (dry run of the N-GPU path)
    bool verify = false;    // read y back and check it against a float CSR loop (sw/host.cpp:33-74)
    double verify_eps = 1e-4;
};

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
void hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return;
    std::printf("HIP Error at %s: %s\n", what, hipGetErrorString(e));
    std::exit(EXIT_FAILURE);
}
void nccl_check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return;
    std::printf("RCCL Error at %s: %s\n", what, ncclGetErrorString(r));
    std::exit(EXIT_FAILURE);
}

int parse_impl(const std::string& s) {
    if (s == "fixed") return hisparse::IMPL_FIXED;
    if (s == "float_pob") return hisparse::IMPL_FLOAT_POB;
    if (s == "float_stall") return hisparse::IMPL_FLOAT_STALL;
    return -1;
}

void dump_words(const std::string& path, const std::vector<uint32_t>& words) {
    if (path.empty()) return;
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(words.data()), std::streamsize(words.size() * 4));
    if (!f) {
        std::cout << "ERROR : cannot write " << path << std::endl;
        std::exit(EXIT_FAILURE);
    }
}

void fill_result(benchmark_result& res, double nnz, double ms) {
    res.spmv_time_ms = ms;
    res.throughput_GBPS = nnz * 8.0 / 1024.0 / 1024.0 / 1024.0 / (ms / 1000.0);  // GiB/s, as the reference prints
    res.throughput_GOPS = 2.0 * nnz / 1e6 / ms;
}

// x = rand() % 2 like the reference (benchmark.cpp:205-212; rand() is never seeded: glibc's default sequence), packed
// by the value type's converting constructor
std::vector<uint32_t> make_vector(int impl, uint32_t num_cols) {
    std::vector<float> vector_f(num_cols);
    for (auto& v : vector_f) v = float(std::rand() % 2);
    std::vector<uint32_t> vector(num_cols);
    hisparse::pack_vector(impl, vector_f.data(), vector_f.size(), vector.data());
    return vector;
}

// ---- --verify: the reference harness's own check (sw/host.cpp:33-74,370; spmv_csim/csim.cpp:143-184), restated ---------------------------
// compute_ref: ref[row] += mat.adj_data[i] * vector[idx] in float32 over the float matrix and the float vector; verify: every
// |float(kernel result) - ref| < epsilon (absolute, 1e-4).  float(VAL_T): Q8.24 raw / 2^24, or the fp32 bits themselves.
// The vector is given as packed words (what the device multiplied by) and converted back: for rand() % 2 that is exact in both modes.
float word_to_float(int impl, uint32_t w) {
    if (impl == hisparse::IMPL_FIXED) return float(double(w) / 16777216.0);
    float f;
    std::memcpy(&f, &w, 4);
    return f;
}
bool verify_result(const Options& o, const spmv::io::CSRMatrix<float>& mat, const std::vector<uint32_t>& x_words, const std::vector<uint32_t>& y_words) {
    std::vector<float> x(x_words.size());
    for (size_t i = 0; i < x.size(); ++i) x[i] = word_to_float(o.impl, x_words[i]);
    uint64_t bad = 0, first_bad = 0;
    double worst = 0.0;
    uint32_t worst_row = 0;
    float first_ref = 0, first_got = 0;
    for (uint32_t row = 0; row < mat.num_rows; ++row) {
        float ref = 0.0f;
        for (uint32_t i = mat.adj_indptr[row]; i < mat.adj_indptr[row + 1]; ++i) ref += mat.adj_data[i] * x[mat.adj_indices[i]];
        const float got = row < y_words.size() ? word_to_float(o.impl, y_words[row]) : 0.0f;
        const double d = std::fabs(double(got) - double(ref));
        if (d > worst || d != d) { worst = d; worst_row = row; }
        if (!(d < o.verify_eps)) {
            if (!bad) { first_bad = row; first_ref = ref; first_got = got; }
            ++bad;
        }
    }
    if (bad) {
        std::cout << "Error: Result mismatch" << std::endl;
        std::cout << "  i = " << first_bad << "  Reference result = " << first_ref << "  Kernel result = " << first_got << std::endl;
        std::cout << "INFO : verify FAILED: " << bad << " of " << mat.num_rows << " rows differ by " << o.verify_eps << " or more (largest difference " << worst
                  << " at row " << worst_row << ")" << std::endl;
        return false;
    }
    std::cout << "INFO : verify PASSED: " << mat.num_rows << " rows within " << o.verify_eps << " absolute of the float32 CSR loop (largest difference " << worst
              << " at row " << worst_row << ")" << std::endl;
    return true;
}

struct DeviceSlab {   // one device's share of the matrix
    hs_context* ctx = nullptr;
    hisparse::ChannelPackets packets;
};

void load_slab(DeviceSlab& s, const Options& o, int device, const std::vector<uint32_t>& vector, const spmv::io::CSRMatrix<float>* csr = nullptr) {
    check(hs_create(&s.ctx, device, o.impl, o.ob_bank_size, o.vb_bank_size), nullptr, "hs_create");
    if (csr) {     // straight from CSR: the device pads, converts and re-tiles (hisparse_hip.h)
        check(hs_load_matrix_csr(s.ctx, csr->num_rows, csr->num_cols, csr->adj_indptr.data(), csr->adj_indices.data(), csr->adj_data.data(), nullptr, nullptr),
              s.ctx, "hs_load_matrix_csr");
        check(hs_load_vector(s.ctx, vector.data(), s.packets.num_cols), s.ctx, "hs_load_vector");
        return;
    }
    const void* ch[HS_NUM_CHANNELS];
    uint64_t n[HS_NUM_CHANNELS];
    for (unsigned c = 0; c < HS_NUM_CHANNELS; ++c) {
        ch[c] = s.packets.channel[c].data();
        n[c] = s.packets.channel[c].size();
    }
    check(hs_load_matrix(s.ctx, ch, n, s.packets.num_rows, s.packets.num_cols, s.packets.num_row_partitions, s.packets.num_col_partitions), s.ctx,
          "hs_load_matrix");
    check(hs_load_vector(s.ctx, vector.data(), s.packets.num_cols), s.ctx, "hs_load_vector");
}

void run_once(DeviceSlab& s, bool partition_loop) {
    if (!partition_loop) {
        check(hs_run(s.ctx), s.ctx, "hs_run");    // every row partition (benchmark.cpp:318-339) in one launch
        return;
    }
    for (uint32_t j = 0; j < s.packets.num_row_partitions; ++j) {   // the reference's loop, literally (:318-338)
        check(hs_run_partition(s.ctx, j, s.packets.part_len(j)), s.ctx, "hs_run_partition");
        check(hs_sync(s.ctx), s.ctx, "hs_sync");  // queue.finish()
    }
}

benchmark_result spmv_benchmark(const Options& o, spmv::io::CSRMatrix<float>& ext_matrix, bool skip_empty_rows) {
    using clock = std::chrono::steady_clock;
    benchmark_result res{};
    std::cout << "INFO : Test started" << std::endl;
    const auto t0 = clock::now();
    hisparse::Geometry g = hisparse::make_geometry(o.impl, o.ob_bank_size, o.vb_bank_size);
    DeviceSlab s;
    if (o.from_csr) {      // no CPSR at all: only the padded dimensions and the partition counts the launch loop needs
        s.packets.geom = g;
        s.packets.num_rows = uint32_t((uint64_t(ext_matrix.num_rows) + g.row_divisor - 1) / g.row_divisor * g.row_divisor);
        s.packets.num_cols = (ext_matrix.num_cols + hisparse::PACK_SIZE - 1) / hisparse::PACK_SIZE * hisparse::PACK_SIZE;
        s.packets.num_row_partitions = uint32_t((s.packets.num_rows + g.logical_ob - 1) / g.logical_ob);
        s.packets.num_col_partitions = uint32_t((s.packets.num_cols + g.logical_vb - 1) / g.logical_vb);
        s.packets.nnz = ext_matrix.adj_data.size();
    } else {
        s.packets = hisparse::format_matrix(ext_matrix, g, skip_empty_rows);
    }
    const hisparse::ChannelPackets& packets = s.packets;
    const auto t1 = clock::now();
    res.preprocess_time_s = std::chrono::duration<double>(t1 - t0).count();
    std::cout << "INFO : Matrix loading/preprocessing complete!" << std::endl;
    std::cout << "  row_partitions: " << packets.num_row_partitions << std::endl;
    std::cout << "  col_partitions: " << packets.num_col_partitions << std::endl;

    std::vector<uint32_t> vector = make_vector(o.impl, packets.num_cols), result(packets.num_rows, 0);
    std::cout << "INFO : Input/result initialization complete!" << std::endl;
    load_slab(s, o, o.device, vector, o.from_csr ? &ext_matrix : nullptr);
    hs_context* ctx = s.ctx;
    std::cout << "INFO : Host -> Device data transfer complete!" << std::endl;
    hs_stats st;
    hs_get_stats(ctx, &st);
    std::cout << "  device load (decode + re-tile + H2D): " << st.load_seconds << " s, " << st.num_blocks << " row blocks on "
              << st.num_workgroups << " workgroups, stream " << st.stream_bytes / 1e6 << " MB" << std::endl;

    std::cout << "INFO : Invoking kernel:" << std::endl;
    for (int i = 0; i < 5; ++i) run_once(s, o.partition_loop);  // untimed warm-ups (the reference has none)
    check(hs_sync(ctx), ctx, "hs_sync");
    double total_ms = 0;
    for (unsigned i = 0; i < o.runs; ++i) {
        const auto a = clock::now();
        run_once(s, o.partition_loop);
        check(hs_sync(ctx), ctx, "hs_sync");  // queue.finish()
        total_ms += std::chrono::duration<double, std::milli>(clock::now() - a).count();
    }
    const double nnz = double(packets.nnz);
    fill_result(res, nnz, total_ms / o.runs);
    float ev_ms = 0, k_ms = 0;
    check(hs_time_runs(ctx, 0, int(o.runs), &ev_ms, &k_ms), ctx, "hs_time_runs");
    std::cout << "  device-side: " << ev_ms / o.runs << " ms per SpMV back-to-back, kernel alone " << k_ms / o.runs << " ms = "
              << nnz * 8.0 / (k_ms / o.runs * 1e-3) / 1e9 << " GB/s = " << nnz * 8.0 / (k_ms / o.runs * 1e-3) / 8e12 * 100
              << " % of the 8 TB/s HBM roofline" << std::endl;
    if (!o.partition_loop) {      // the NUM_RUNS loop as ONE call (hs_run_batch): enqueued from the library's C loop, and replayed from a captured hipGraph
        double batch_ms[2] = {0, 0};
        for (int graph = 0; graph < 2; ++graph) {
            check(hs_set_option(ctx, "batch_graph", graph ? "1" : "0"), ctx, "hs_set_option");
            check(hs_run_batch(ctx, o.runs), ctx, "hs_run_batch");      // untimed: warm-up / capture + instantiate
            check(hs_sync(ctx), ctx, "hs_sync");
            const auto a = clock::now();
            check(hs_run_batch(ctx, o.runs), ctx, "hs_run_batch");
            check(hs_sync(ctx), ctx, "hs_sync");
            batch_ms[graph] = std::chrono::duration<double, std::milli>(clock::now() - a).count() / o.runs;
        }
        std::cout << "  hs_run_batch(" << o.runs << "), one sync at the end: " << batch_ms[0] << " ms per SpMV enqueued from a C loop, " << batch_ms[1]
                  << " ms per SpMV replayed from one hipGraph" << std::endl;
    }
    check(hs_read_result(ctx, result.data(), packets.num_rows), ctx, "hs_read_result");
    dump_words(o.dump_x, vector);
    dump_words(o.dump_y, result);
    hs_destroy(ctx);
    if (o.verify && !verify_result(o, ext_matrix, vector, result)) std::exit(3);
    return res;
}

// ---- one matrix over N GPUs of this node: row slabs, one thread, N streams, one ncclAllGather of y per SpMV ------------------
benchmark_result spmv_benchmark_multi(const Options& o, spmv::io::CSRMatrix<float>& ext_matrix, bool skip_empty_rows) {
    using clock = std::chrono::steady_clock;
    benchmark_result res{};
    const int N = o.gpus;
    const bool share = o.share_gpu;                       // dry run: every slab on device o.device, no RCCL
    auto dev_of = [&](int d) { return share ? o.device : d; };
    int visible = 0;
    hip_check(hipGetDeviceCount(&visible), "hipGetDeviceCount");
    if (share ? o.device >= visible : N > visible) {
        std::cout << "ERROR : --gpus " << N << " but only " << visible << " device(s) visible" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    std::cout << "INFO : Test started (" << N << (share ? " row slabs sharing ONE GPU: dry run of the multi-GPU path, not a measurement)" : " GPUs, row slabs)") << std::endl;
    const auto t0 = clock::now();
    hisparse::Geometry g = hisparse::make_geometry(o.impl, o.ob_bank_size, o.vb_bank_size);
    const uint32_t true_rows = ext_matrix.num_rows;
    const uint32_t padded_cols = hisparse::padded_rows(ext_matrix.num_cols, hisparse::PACK_SIZE);
    const std::vector<uint32_t> bounds = hisparse::split_rows_by_nnz(ext_matrix.adj_indptr, uint32_t(N), g.row_divisor);
    std::vector<DeviceSlab> slab(N);
    uint64_t nnz_total = 0;
    uint32_t chunk = 0;   // padded rows of the tallest slab = the all-gather count per rank
    for (int d = 0; d < N; ++d) {
        if (bounds[d + 1] == bounds[d]) {
            std::cout << "ERROR : the matrix has fewer than " << N << " x " << g.row_divisor << " rows: slab " << d << " is empty" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        spmv::io::CSRMatrix<float> part = hisparse::row_slab(ext_matrix, bounds[d], bounds[d + 1]);
        slab[d].packets = hisparse::format_matrix(part, g, skip_empty_rows);
        nnz_total += slab[d].packets.nnz;
        chunk = std::max(chunk, slab[d].packets.num_rows);
        std::cout << "  slab " << d << ": rows [" << bounds[d] << ", " << bounds[d + 1] << "), nnz " << slab[d].packets.nnz << ", "
                  << slab[d].packets.num_row_partitions << " x " << slab[d].packets.num_col_partitions << " partitions" << std::endl;
    }
    res.preprocess_time_s = std::chrono::duration<double>(clock::now() - t0).count();
    std::cout << "INFO : Matrix loading/preprocessing complete!" << std::endl;

    std::vector<uint32_t> vector = make_vector(o.impl, padded_cols);
    std::cout << "INFO : Input/result initialization complete!" << std::endl;
    std::vector<int> devlist(N);
    std::vector<hipStream_t> stream(N);
    std::vector<uint32_t*> gathered(N, nullptr);   // per device: N chunks; the device's own slab is written in place at chunk d
    std::vector<ncclComm_t> comm(N);
    for (int d = 0; d < N; ++d) devlist[d] = d;
    if (!share) nccl_check(ncclCommInitAll(comm.data(), N, devlist.data()), "ncclCommInitAll");
    for (int d = 0; d < N; ++d) {
        hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
        hip_check(hipStreamCreateWithFlags(&stream[d], hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipMalloc(reinterpret_cast<void**>(&gathered[d]), size_t(chunk) * N * 4), "hipMalloc");
        hip_check(hipMemset(gathered[d], 0, size_t(chunk) * N * 4), "hipMemset");
        load_slab(slab[d], o, dev_of(d), vector);
        check(hs_set_stream(slab[d].ctx, stream[d]), slab[d].ctx, "hs_set_stream");
        check(hs_bind_device_result(slab[d].ctx, gathered[d] + size_t(d) * chunk), slab[d].ctx, "hs_bind_device_result");
    }
    std::cout << "INFO : Host -> Device data transfer complete!" << std::endl;

    auto spmv_all = [&]() {
        for (int d = 0; d < N; ++d) run_once(slab[d], false);
    };
    auto gather_all = [&]() {   // in place: sendbuff = recvbuff + rank * count
        nccl_check(ncclGroupStart(), "ncclGroupStart");
        for (int d = 0; d < N; ++d)
            nccl_check(ncclAllGather(gathered[d] + size_t(d) * chunk, gathered[d], chunk, ncclUint32, comm[d], stream[d]), "ncclAllGather");
        nccl_check(ncclGroupEnd(), "ncclGroupEnd");
    };
    // The same gather WITHOUT a collective (--peer-gather): every device stores its slab straight into every other device's gather buffer
    // over xGMI (hs_push_result: one small kernel on the producer's stream, plain stores into peer memory); an event per device tells
    // the others' streams when its slab has been pushed.  With one device (--sharded) the "peer" is a second buffer on the same device.
    std::vector<hipEvent_t> pushed(N, nullptr);
    std::vector<uint32_t*> loopback(N, nullptr);
    const bool peer_gather = o.peer_gather || share;      // sharing one device: the peer stores ARE the gather
    if (peer_gather) {
        for (int d = 0; d < N; ++d) {
            hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
            for (int p = 0; p < N && !share; ++p)
                if (p != d) {
                    const hipError_t e = hipDeviceEnablePeerAccess(p, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_check(e, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
            hip_check(hipEventCreateWithFlags(&pushed[d], hipEventDisableTiming), "hipEventCreate");
            if (N == 1) {
                hip_check(hipMalloc(reinterpret_cast<void**>(&loopback[d]), size_t(chunk) * 4), "hipMalloc");
                hip_check(hipMemset(loopback[d], 0, size_t(chunk) * 4), "hipMemset");
            }
        }
    }
    if (peer_gather && N - 1 > 8) {      // hs_push_result: at most 8 destinations (dst[8] below)
        std::fprintf(stderr, "--peer-gather: at most 9 GPUs (a slab is pushed to 8 peers)\n");
        std::exit(2);
    }
    auto push_all = [&]() {
        for (int d = 0; d < N; ++d) {
            void* dst[8];
            uint32_t n = 0;
            for (int p = 0; p < N; ++p)
                if (p != d) dst[n++] = gathered[p] + size_t(d) * chunk;
            if (N == 1) dst[n++] = loopback[d];
            // (the slab's OWN padded rows: the chunk is the tallest slab's, and a context pushes no more than it holds -- found by the --share-gpu dry run)
            check(hs_push_result(slab[d].ctx, dst, n, slab[d].packets.num_rows), slab[d].ctx, "hs_push_result");
            hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
            hip_check(hipEventRecord(pushed[d], stream[d]), "hipEventRecord");
        }
        for (int d = 0; d < N; ++d) {      // nobody goes on (e.g. to an SpMV that reads the gathered y as its x) before every slab has arrived
            hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
            for (int p = 0; p < N; ++p)
                if (p != d) hip_check(hipStreamWaitEvent(stream[d], pushed[p], 0), "hipStreamWaitEvent");
        }
    };
    auto sync_all = [&]() {
        for (int d = 0; d < N; ++d) {
            hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
            hip_check(hipStreamSynchronize(stream[d]), "hipStreamSynchronize");
        }
    };
    auto timed = [&](int how) {      // 0: no gather, 1: ncclAllGather, 2: peer stores
        auto step = [&]() { spmv_all(); if (how == 1) gather_all(); else if (how == 2) push_all(); };
        for (int i = 0; i < 5; ++i) step();
        sync_all();
        const auto a = clock::now();
        if (how == 0) {      // no exchange between the steps: every slab's NUM_RUNS loop as one batch (hs_run_batch), the devices side by side
            for (int d = 0; d < N; ++d) check(hs_run_batch(slab[d].ctx, o.runs), slab[d].ctx, "hs_run_batch");
        } else {
            for (unsigned i = 0; i < o.runs; ++i) step();
        }
        sync_all();
        return std::chrono::duration<double, std::milli>(clock::now() - a).count() / o.runs;
    };
    std::cout << "INFO : Invoking kernel:" << std::endl;
    const double compute_ms = timed(0);
    benchmark_result compute_only = res;
    fill_result(compute_only, double(nnz_total), compute_ms);
    std::cout << "  compute only (y left sharded, like the reference leaves it in HBM): " << compute_only << std::endl;
    if (o.gather) {
        if (!share) {
            const double both_ms = timed(1);
            fill_result(res, double(nnz_total), both_ms);
            std::cout << "  compute + one all-gather of y per SpMV (RCCL, " << chunk * 4.0 / 1e3 << " kB per rank): " << res << std::endl;
            std::cout << "  all-gather cost per SpMV: " << (both_ms - compute_ms) * 1e3 << " us" << std::endl;
        } else {
            res = compute_only;
        }
        if (peer_gather) {
            const double peer_ms = timed(2);
            benchmark_result peer = res;
            fill_result(peer, double(nnz_total), peer_ms);
            std::cout << "  compute + gather by peer stores over xGMI (hs_push_result, no collective): " << peer << std::endl;
            std::cout << "  peer-store gather cost per SpMV: " << (peer_ms - compute_ms) * 1e3 << " us" << std::endl;
            // the pushed slabs must equal what the collective delivered
            spmv_all();
            push_all();
            sync_all();
            std::vector<uint32_t> a(chunk), b(chunk);
            bool same = true;
            for (int d = 0; d < N && same; ++d) {
                const int reader = N == 1 ? 0 : (d + 1) % N;
                hip_check(hipSetDevice(dev_of(reader)), "hipSetDevice");
                hip_check(hipMemcpy(a.data(), N == 1 ? loopback[d] : gathered[reader] + size_t(d) * chunk, size_t(chunk) * 4, hipMemcpyDeviceToHost), "hipMemcpy");
                hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
                hip_check(hipMemcpy(b.data(), gathered[d] + size_t(d) * chunk, size_t(chunk) * 4, hipMemcpyDeviceToHost), "hipMemcpy");
                same = a == b;
            }
            std::cout << "  peer-store gather: pushed slabs " << (same ? "identical to" : "DIFFER from") << " their sources" << std::endl;
            if (!same) std::exit(EXIT_FAILURE);
        }
    } else {
        res = compute_only;
    }
    std::cout << "  fraction of " << N << " x 8 TB/s HBM roofline: " << double(nnz_total) * 8.0 / (res.spmv_time_ms * 1e-3) / (8e12 * N) * 100 << " %" << std::endl;

    // y in natural row order from device 0's gathered buffer (or slab by slab without the gather)
    spmv_all();
    if (o.gather) { if (share) push_all(); else gather_all(); }
    sync_all();
    std::vector<uint32_t> result(true_rows, 0), tmp(chunk);
    for (int d = 0; d < N; ++d) {
        const int src = o.gather ? 0 : d;
        hip_check(hipSetDevice(dev_of(src)), "hipSetDevice");
        hip_check(hipMemcpy(tmp.data(), gathered[src] + size_t(d) * chunk, size_t(chunk) * 4, hipMemcpyDeviceToHost), "hipMemcpy");
        std::copy(tmp.begin(), tmp.begin() + (bounds[d + 1] - bounds[d]), result.begin() + bounds[d]);
    }
    dump_words(o.dump_x, vector);
    dump_words(o.dump_y, result);
    for (int d = 0; d < N; ++d) {
        hip_check(hipSetDevice(dev_of(d)), "hipSetDevice");
        check(hs_set_stream(slab[d].ctx, nullptr), slab[d].ctx, "hs_set_stream");
        hs_destroy(slab[d].ctx);
        if (!share) ncclCommDestroy(comm[d]);
        if (pushed[d]) hip_check(hipEventDestroy(pushed[d]), "hipEventDestroy");
        if (loopback[d]) hip_check(hipFree(loopback[d]), "hipFree");
        hip_check(hipFree(gathered[d]), "hipFree");
        hip_check(hipStreamDestroy(stream[d]), "hipStreamDestroy");
    }
    if (o.verify) {      // the assembled y against the float CSR loop over the UNSPLIT matrix
        std::vector<uint32_t> x_true(vector.begin(), vector.begin() + std::min<size_t>(vector.size(), padded_cols));
        if (!verify_result(o, ext_matrix, x_true, result)) std::exit(3);
    }
    return res;
}

bool parse_args(int argc, char** argv, Options& o) {
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* name) -> std::string {
            if (i + 1 >= argc) { std::cout << "ERROR : " << name << " needs a value" << std::endl; std::exit(1); }
            return argv[++i];
        };
        if (a == "--values") o.values = need("--values");
        else if (a == "--partition-loop") o.partition_loop = true;
        else if (a == "--from-csr") o.from_csr = true;
        else if (a == "--runs") o.runs = unsigned(std::max(1, std::atoi(need("--runs").c_str())));
        else if (a == "--dump-x") o.dump_x = need("--dump-x");
        else if (a == "--dump-y") o.dump_y = need("--dump-y");
        else if (a == "--gpus") o.gpus = std::max(1, std::atoi(need("--gpus").c_str()));
        else if (a == "--no-gather") o.gather = false;
        else if (a == "--peer-gather") o.peer_gather = true;
        else if (a == "--sharded") o.sharded = true;
        else if (a == "--share-gpu") o.share_gpu = true;
        else if (a == "--verify") o.verify = true;
        else if (a == "--verify-eps") { o.verify = true; o.verify_eps = std::atof(need("--verify-eps").c_str()); }
        else if (a == "--device") o.device = std::atoi(need("--device").c_str());
        else if (a.rfind("--", 0) == 0) { std::cout << "ERROR : unknown option " << a << std::endl; return false; }
        else pos.push_back(a);
    }
    if (pos.size() != 4 && pos.size() != 5) return false;
    o.impl = parse_impl(pos[0]);
    o.dataset = pos[1];
    o.vb_bank_size = unsigned(std::atoi(pos[2].c_str())) * 1024;  // benchmark.cpp:364-365
    o.ob_bank_size = unsigned(std::atoi(pos[3].c_str())) * 1024;
    if (pos.size() == 5) o.device = std::atoi(pos[4].c_str());
    if (o.impl < 0) { std::cout << "ERROR : unknown implementation " << pos[0] << std::endl; std::exit(1); }
    if (!o.values.empty() && o.values != "literal" && o.values != "intent" && o.values != "keep") {
        std::cout << "ERROR : --values must be literal, intent or keep" << std::endl;
        std::exit(1);
    }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    if (!parse_args(argc, argv, o)) {
        std::cout << "Usage: " << argv[0] << " <fixed|float_pob|float_stall> <dataset.npz | synth:kind:rows:cols:a:b:c:seed> <v> <o> [device]"
                  << " [--values literal|intent|keep] [--from-csr] [--partition-loop] [--runs K] [--dump-x FILE] [--dump-y FILE] [--gpus N [--no-gather] [--peer-gather] [--sharded] [--share-gpu]] [--verify [--verify-eps E]]" << std::endl;
        return 0;
    }
    std::cout << "------ Running benchmark on " << o.dataset << std::endl;
    spmv::io::CSRMatrix<float> mat_f;
    const bool synthetic = o.dataset.rfind("synth:", 0) == 0;
    try {
        if (synthetic) {
            // synth:kind:rows:cols:a:b:c:seed -> the seeded generators of libhisparse_host (hsf_csr_generate)
            std::vector<std::string> f;
            std::stringstream ss(o.dataset);
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
        } else {
            mat_f = spmv::io::load_csr_matrix_from_float_npz(o.dataset);
        }
    } catch (const std::exception& e) {
        std::cout << "ERROR : " << e.what() << std::endl;
        return 1;
    }
    const std::string values = !o.values.empty() ? o.values : (synthetic ? "keep" : "intent");
    if (values == "literal") {
        // `x = 1 / mat_f.num_cols` with both operands integers (sw/benchmark.cpp:411): 0 for every matrix wider than one column
        const float v = float(1 / std::max<uint32_t>(1, mat_f.num_cols));
        for (auto& x : mat_f.adj_data) x = v;
    } else if (values == "intent") {
        // the intent of that line, a small non-degenerate constant, so that the result is worth reading back
        for (auto& x : mat_f.adj_data) x = 1.0f / float(mat_f.num_cols);
    }
    std::cout << (o.gpus > 1 || o.sharded || o.share_gpu ? spmv_benchmark_multi(o, mat_f, true) : spmv_benchmark(o, mat_f, true)) << std::endl;
    std::cout << "===== Benchmark Finished =====" << std::endl;
    return 0;
}
