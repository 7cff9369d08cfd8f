// driver_capi.cpp — C ABI (include/cudasw4_amd_driver.h) around SearchDriver.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <string>

#include "../../../include/cudasw4_amd_driver.h"
#include <random>

#include "search_driver.hpp"
#include "sequence_reader.hpp"

using namespace swh;

struct swdrv_reader {
    std::unique_ptr<SequenceReader> reader;
};

struct swdrv {
    std::unique_ptr<SearchDriver> driver;
    std::shared_ptr<Database> db;
    int lastRescored = 0;
};

namespace {
thread_local std::string g_error;
template <class F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
}  // namespace

extern "C" {

const char* swdrv_last_error(void) { return g_error.c_str(); }

int swdrv_create(const int* devices, int ndev, int num_top, int matrix, int kind_single, int kind_many_small,
                 int kind_many_large, int kind_overflow, size_t max_gpu_mem, size_t max_batch_bytes,
                 size_t max_batch_sequences, size_t max_temp_bytes, int gop, int gex, int verbose, swdrv** out) {
    return guarded([&] {
        if (!out) throw std::runtime_error("null out pointer");
        std::vector<int> ids;
        if (ndev > 0 && devices) ids.assign(devices, devices + ndev);
        else {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
            for (int i = 0; i < n; i++) ids.push_back(i);
        }
        MatrixId mid;  // 45 | 50 | 62 | 80: the 21-letter tables; 4525 | 5025 | 6225 | 8025: the full 25-letter ones
        const std::string mname = matrix > 100 ? "blosum" + std::to_string(matrix / 100) + "_25" : "blosum" + std::to_string(matrix);
        if (matrix % 100 != 25 && matrix > 100) throw std::runtime_error("unknown matrix");
        if (!parse_matrix_name(mname, mid)) throw std::runtime_error("unknown matrix");
        auto kt = [](int k) {
            if (k < 0 || k > 3) throw std::runtime_error("unknown kernel type");
            return KernelType(k);
        };
        KernelTypeConfig kc{kt(kind_single), kt(kind_many_small), kt(kind_many_large), kt(kind_overflow)};
        MemoryConfig mc;
        if (max_gpu_mem) mc.maxGpuMem = max_gpu_mem;
        if (max_batch_bytes) mc.maxBatchBytes = max_batch_bytes;
        if (max_batch_sequences) mc.maxBatchSequences = max_batch_sequences;
        if (max_temp_bytes) mc.maxTempBytes = max_temp_bytes;
        auto* d = new swdrv;
        try {
            d->driver = std::make_unique<SearchDriver>(ids, num_top, mid, kc, mc, verbose != 0, gop, gex);
        } catch (...) {
            delete d;
            throw;
        }
        *out = d;
    });
}

int swdrv_destroy(swdrv* d) {
    delete d;
    return 0;
}

int swdrv_open_db(swdrv* d, const char* prefix, int prefetch) {
    return guarded([&] {
        d->db = Database::open_or_read(prefix, prefetch != 0);
        d->driver->setDatabase(d->db);
    });
}

int swdrv_pseudo_db(swdrv* d, size_t num, int32_t length) {
    return guarded([&] {
        d->db = Database::pseudo(num, length);
        d->driver->setDatabase(d->db);
    });
}

int swdrv_db_from_arrays(swdrv* d, const int8_t* chars, size_t nchars, const uint64_t* offsets, const int32_t* lengths, size_t n) {
    return guarded([&] {
        if (!chars || !offsets || !lengths) throw std::runtime_error("null array");
        if (n && offsets[n] - offsets[0] > nchars) throw std::runtime_error("chars array shorter than the last offset");
        std::vector<int8_t> c(chars, chars + nchars);
        std::vector<uint64_t> o(offsets, offsets + n + 1), ho(n + 1);
        for (size_t i = 0; i <= n; i++) { o[i] -= offsets[0]; ho[i] = i; }
        std::vector<int32_t> l(lengths, lengths + n);
        d->db = Database::from_vectors(std::move(c), std::move(o), std::move(l), std::vector<char>(n, 'S'), std::move(ho));
        d->driver->setDatabase(d->db);
    });
}

int swdrv_set_shard(swdrv* d, int rank, int world, int64_t id_base) {
    return guarded([&] { d->driver->setShard(rank, world, id_base); });
}

int swdrv_upload(swdrv* d) {
    return guarded([&] { d->driver->prefetchDBToGpus(); });
}

int64_t swdrv_num_sequences(swdrv* d) { return d && d->db ? int64_t(d->db->num_sequences()) : 0; }
int swdrv_num_gpus(swdrv* d) { return d ? d->driver->numGpus() : 0; }
int swdrv_set_num_top(swdrv* d, int k) {
    d->driver->setNumTop(k);
    return 0;
}

int swdrv_scan(swdrv* d, const char* query, int32_t qlen, int32_t* scores, int64_t* ids, int cap, int* nres,
               int* num_overflows, double* seconds, double* gcups) {
    return guarded([&] {
        ScanResult r = d->driver->scan(query, qlen);
        const int n = int(std::min<size_t>(r.scores.size(), size_t(cap > 0 ? cap : 0)));
        for (int i = 0; i < n; i++) {
            scores[i] = r.scores[size_t(i)];
            ids[i] = r.referenceIds[size_t(i)];
        }
        if (nres) *nres = n;
        if (num_overflows) *num_overflows = r.stats.numOverflows;
        d->lastRescored = r.stats.numRescored;
        if (seconds) *seconds = r.stats.seconds;
        if (gcups) *gcups = r.stats.gcups;
    });
}

int swdrv_scan_submit(swdrv* d, const char* query, int32_t qlen) {
    return guarded([&] { d->driver->submit(query, qlen); });
}

int swdrv_scan_collect(swdrv* d, int32_t* scores, int64_t* ids, int cap, int* nres, int* num_overflows, double* seconds,
                       double* gcups) {
    return guarded([&] {
        ScanResult r = d->driver->collect();
        const int n = int(std::min<size_t>(r.scores.size(), size_t(cap > 0 ? cap : 0)));
        for (int i = 0; i < n; i++) {
            scores[i] = r.scores[size_t(i)];
            ids[i] = r.referenceIds[size_t(i)];
        }
        if (nres) *nres = n;
        if (num_overflows) *num_overflows = r.stats.numOverflows;
        d->lastRescored = r.stats.numRescored;
        if (seconds) *seconds = r.stats.seconds;
        if (gcups) *gcups = r.stats.gcups;
    });
}

int swdrv_in_flight(swdrv* d) { return d ? d->driver->inFlight() : 0; }

int64_t swdrv_cached_chars(swdrv* d, int gpu) {
    int64_t v = -1;
    (void)guarded([&] { v = int64_t(d->driver->cachedChars(gpu)); });
    return v;
}

int64_t swdrv_streamed_bytes(swdrv* d) { return d ? int64_t(d->driver->streamedBytesTotal()) : 0; }

int swdrv_record_kernel_events(swdrv* d, int on) {
    return guarded([&] { d->driver->recordKernelEvents(on); });
}

int swdrv_take_kernel_events(swdrv* d, double* out, int cap) {
    int n = -1;
    const int rc = guarded([&] {
        const auto ev = d->driver->takeKernelEvents();
        n = int(ev.size());
        for (int i = 0; i < n && i < cap; i++) {
            double* o = out + size_t(i) * 15;
            const KernelEvent& e = ev[size_t(i)];
            o[0] = e.gpu; o[1] = e.kind; o[2] = e.part_id; o[3] = e.qlen; o[4] = double(e.subjects); o[5] = e.cells; o[6] = e.chars; o[7] = e.ms;
            o[8] = e.t0_ms; o[9] = e.t1_ms; o[10] = e.eff_kind; o[11] = e.rows; o[12] = e.nstripes; o[13] = e.lanes; o[14] = e.rescore;
        }
    });
    return rc == 0 ? n : -1;
}

int swdrv_shard_info(swdrv* d, int gpu, int64_t* num_local, int64_t* residues, int64_t* chars, int* resident) {
    return guarded([&] {
        if (num_local) *num_local = int64_t(d->driver->numLocal(gpu));
        if (residues) *residues = int64_t(d->driver->localResidues(gpu));
        if (chars) *chars = int64_t(d->driver->localChars(gpu));
        if (resident) *resident = d->driver->isResident(gpu) ? 1 : 0;
    });
}

int swdrv_last_scores(swdrv* d, int gpu, float* scores, int64_t* ids) {
    return guarded([&] { d->driver->lastScores(gpu, scores, ids); });
}

int swdrv_batch_intervals(swdrv* d, float* out, int cap) {
    int n = -1;
    const int rc = guarded([&] {
        const auto iv = d->driver->lastBatchIntervals();
        n = int(iv.size());
        for (int i = 0; i < n && i < cap; i++) { out[3 * i] = float(iv[size_t(i)].gpu); out[3 * i + 1] = iv[size_t(i)].begin_ms; out[3 * i + 2] = iv[size_t(i)].end_ms; }
    });
    return rc == 0 ? n : -1;
}

int swdrv_gpu_spans(swdrv* d, double* out, int cap) {
    int n = -1;
    const int rc = guarded([&] {
        const auto sp = d->driver->lastGpuSpans();
        n = int(sp.size());
        for (int i = 0; i < n && i < cap; i++) { out[2 * i] = sp[size_t(i)].begin_s; out[2 * i + 1] = sp[size_t(i)].end_s; }
    });
    return rc == 0 ? n : -1;
}

int swdrv_plan_runs(const int32_t* sorted_lengths, size_t n, int kind_single, int kind_many_small, int kind_many_large,
                    int64_t* out, int cap) {
    return swdrv_plan_runs_mode(sorted_lengths, n, kind_single, kind_many_small, kind_many_large, 0, out, cap);
}

int swdrv_plan_runs_mode(const int32_t* sorted_lengths, size_t n, int kind_single, int kind_many_small, int kind_many_large,
                         int latency_mode, int64_t* out, int cap) {
    int nruns = -1;
    const int rc = guarded([&] {
        for (int k : {kind_single, kind_many_small, kind_many_large})
            if (k < 0 || k > 3) throw std::runtime_error("unknown kernel type");
        KernelTypeConfig kc{KernelType(kind_single), KernelType(kind_many_small), KernelType(kind_many_large), KernelType::Float};
        const auto& bounds = length_partition_bounds();
        size_t partBegin[kNumLengthPartitions + 1];
        partBegin[0] = 0;
        const int32_t* first = sorted_lengths;
        for (int p = 0; p < kNumLengthPartitions; p++) {
            first = std::upper_bound(first, sorted_lengths + n, bounds[size_t(p)]);
            partBegin[p + 1] = size_t(first - sorted_lengths);
        }
        const auto runs = plan_launch_runs(kc, partBegin, 0, n, [&](size_t pos) { return sorted_lengths[pos]; },
                                           latency_mode ? SIZE_MAX : kLongPartitionMergeMin);
        nruns = int(runs.size());
        for (int i = 0; i < nruns && i < cap; i++) {
            int64_t* o = out + size_t(i) * 5;
            o[0] = int64_t(runs[size_t(i)].kind); o[1] = runs[size_t(i)].part_id; o[2] = int64_t(runs[size_t(i)].begin);
            o[3] = int64_t(runs[size_t(i)].end); o[4] = runs[size_t(i)].maxlen;
        }
    });
    return rc == 0 ? nruns : -1;
}

int swdrv_shard_ranges(const int32_t* sorted_lengths, const uint64_t* offsets, size_t n, int world, int64_t* out) {
    return guarded([&] {
        if (world < 1) throw std::runtime_error("world must be positive");
        const auto& bounds = length_partition_bounds();
        size_t partBegin[kNumLengthPartitions + 1];
        partBegin[0] = 0;
        const int32_t* first = sorted_lengths;
        for (int p = 0; p < kNumLengthPartitions; p++) {
            first = std::upper_bound(first, sorted_lengths + n, bounds[size_t(p)]);
            partBegin[p + 1] = size_t(first - sorted_lengths);
        }
        const auto sh = shard_ranges(offsets, partBegin, world);
        for (int r = 0; r < world; r++)
            for (int p = 0; p < kNumLengthPartitions; p++) {
                out[(size_t(r) * kNumLengthPartitions + size_t(p)) * 2] = int64_t(sh[size_t(r)][size_t(p)].begin);
                out[(size_t(r) * kNumLengthPartitions + size_t(p)) * 2 + 1] = int64_t(sh[size_t(r)][size_t(p)].end);
            }
    });
}

int swdrv_plan_residency(const uint64_t* local_offsets, size_t n, int32_t max_len, size_t max_gpu_mem, size_t max_batch_bytes,
                         size_t max_batch_sequences, size_t max_temp_bytes, size_t free_mem, int allow_cache, int64_t* cache_begin,
                         int64_t* cache_bytes, int64_t* batch_bytes, int64_t* batches, int cap, int64_t* temp_per_stream) {
    int nb = -1;
    const int rc = guarded([&] {
        if (!local_offsets) throw std::runtime_error("null offsets");
        MemoryConfig mc;
        if (max_gpu_mem) mc.maxGpuMem = max_gpu_mem;
        if (max_batch_bytes) mc.maxBatchBytes = max_batch_bytes;
        if (max_batch_sequences) mc.maxBatchSequences = max_batch_sequences;
        if (max_temp_bytes) mc.maxTempBytes = max_temp_bytes;
        const std::vector<uint64_t> off(local_offsets, local_offsets + n + 1);
        const ResidencyPlan rp = plan_residency(off, max_len, mc, free_mem, 3, allow_cache != 0);
        if (cache_begin) *cache_begin = int64_t(rp.cacheBegin);
        if (cache_bytes) *cache_bytes = int64_t(rp.cacheBytes);
        if (batch_bytes) *batch_bytes = int64_t(rp.batchBytes);
        if (temp_per_stream) *temp_per_stream = int64_t(std::min<size_t>(rp.tempPerStream, size_t(INT64_MAX)));
        nb = int(rp.batches.size());
        for (int i = 0; i < nb && i < cap; i++) { batches[2 * i] = int64_t(rp.batches[size_t(i)].first); batches[2 * i + 1] = int64_t(rp.batches[size_t(i)].second); }
    });
    return rc == 0 ? nb : -1;
}

int swdrv_last_rescored(swdrv* d) { return d ? d->lastRescored : 0; }

int swdrv_window_stats(swdrv* d, int64_t* launches, int64_t* windows) {
    return guarded([&] { d->driver->windowStats(launches, windows); });
}

int64_t swdrv_service_launches(swdrv* d) {
    int64_t n = -1;
    (void)guarded([&] { n = d->driver->serviceLaunches(); });
    return n;
}

int64_t swdrv_latency_scans(swdrv* d) {
    int64_t n = -1;
    (void)guarded([&] { n = d->driver->latencyScans(); });
    return n;
}


int swdrv_preferred_in_flight(swdrv* d, int32_t query_length) {
    int n = 1;
    (void)guarded([&] { n = d->driver->preferredInFlight(query_length); });
    return n;
}

int swdrv_handshake_active(swdrv* d) {
    int n = -1;
    (void)guarded([&] { n = d->driver->handshakeActive() ? 1 : 0; });
    return n;
}

int64_t swdrv_pipeline_launches(swdrv* d) {
    int64_t n = -1;
    (void)guarded([&] { n = d->driver->pipelineLaunches(); });
    return n;
}

int64_t swdrv_tail_overlaps(swdrv* d) {
    int64_t n = -1;
    (void)guarded([&] { n = d->driver->tailOverlaps(); });
    return n;
}

int swdrv_prefers_two_in_flight(swdrv* d, int32_t query_length) {
    int v = 0;
    (void)guarded([&] { v = d->driver->prefersTwoInFlight(query_length) ? 1 : 0; });
    return v;
}

int swdrv_numa_node(swdrv* d, int gpu) {
    int node = -1;
    (void)guarded([&] { node = d->driver->numaNode(gpu); });
    return node;
}

int swdrv_device_of(swdrv* d, int gpu) {
    int dev = -1;
    (void)guarded([&] { dev = d->driver->deviceOf(gpu); });
    return dev;
}

int swdrv_device_numa_node(int device) { return numa_node_of_device(device); }

int swdrv_bind_to_numa_node(int node) { return bind_thread_to_numa_node(node) ? 0 : -1; }

int32_t swdrv_reference_length(swdrv* d, int64_t id) { return d->driver->getReferenceLength(id); }

int swdrv_reference_header(swdrv* d, int64_t id, char* buf, int cap) {
    const std::string_view h = d->driver->getReferenceHeader(id);
    const int n = int(std::min<size_t>(h.size(), size_t(cap > 0 ? cap - 1 : 0)));
    if (cap > 0) {
        std::memcpy(buf, h.data(), size_t(n));
        buf[n] = 0;
    }
    return int(h.size());
}

void swdrv_encode(const char* letters, int8_t* codes, size_t n) {
    for (size_t i = 0; i < n; i++) codes[i] = encode_residue(letters[i]);
}

void swdrv_pseudo_sequence(int32_t length, int seed, int8_t* codes) {
    std::mt19937 gen(seed);
    std::uniform_int_distribution<> dist(0, 19);
    for (int32_t i = 0; i < length; i++) codes[i] = int8_t(dist(gen));
}

int swdrv_matrix(int matrix, int8_t* out441) {
    MatrixId id;
    if (!parse_matrix_name("blosum" + std::to_string(matrix), id)) return -1;
    const SubstitutionMatrix& m = substitution_matrix(id);
    std::memcpy(out441, m.m.data(), 441);
    return 0;
}

int swdrv_matrix25(int matrix, int8_t* out625) {
    MatrixId id;
    if (!parse_matrix_name("blosum" + std::to_string(matrix) + "_25", id)) return -1;
    const SubstitutionMatrix& m = substitution_matrix(id);
    std::memcpy(out625, m.m.data(), 625);
    return 0;
}

void swdrv_encode25(const char* letters, int8_t* codes, size_t n) {
    for (size_t i = 0; i < n; i++) codes[i] = encode_residue25(letters[i]);
}

int swdrv_reader_open(const char* path, swdrv_reader** out) {
    return guarded([&] {
        auto* r = new swdrv_reader;
        try {
            r->reader = std::make_unique<SequenceReader>(path);
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
    });
}

int swdrv_reader_next(swdrv_reader* r, const char** header, size_t* header_len, const char** sequence, size_t* sequence_len) {
    if (!r || !r->reader->next()) return 0;
    *header = r->reader->header().data();
    *header_len = r->reader->header().size();
    *sequence = r->reader->sequence().data();
    *sequence_len = r->reader->sequence().size();
    return 1;
}

void swdrv_reader_close(swdrv_reader* r) { delete r; }

}  // extern "C"
