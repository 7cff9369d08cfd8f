// driver_capi.cpp — C ABI (include/cudasw4_amd_driver.h) around SearchDriver.
#include <hip/hip_runtime_api.h>

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
        MatrixId mid;
        if (!parse_matrix_name("blosum" + std::to_string(matrix), mid)) throw std::runtime_error("unknown matrix");
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
        d->db = Database::open(prefix, prefetch != 0);
        d->driver->setDatabase(d->db);
    });
}

int swdrv_pseudo_db(swdrv* d, size_t num, int32_t length) {
    return guarded([&] {
        d->db = Database::pseudo(num, length);
        d->driver->setDatabase(d->db);
    });
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
        if (seconds) *seconds = r.stats.seconds;
        if (gcups) *gcups = r.stats.gcups;
    });
}

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
    std::memcpy(out441, m.m.data(), m.m.size());
    return 0;
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
