// spill_vector.hpp — an append-only array that lives in RAM until a byte budget is exceeded and then moves
// to a memory-mapped temporary file (makedb --mem / --tempdir).  Same purpose as the reference's
// FileBackedUVector (mmapbuffer.hpp:332-507, used by HybridBatch, makedb.cpp:80-103), different
// mechanism: one growable file mapping (ftruncate + mremap) instead of a split memory/file vector.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace swh {

template <class T>
class SpillVector {
    static_assert(std::is_trivially_copyable<T>::value, "SpillVector holds plain data");

public:
    // ram_budget_bytes == 0: never spill
    SpillVector(std::string spill_path, size_t ram_budget_bytes) : path_(std::move(spill_path)), budget_(ram_budget_bytes) {}
    SpillVector(const SpillVector&) = delete;
    SpillVector& operator=(const SpillVector&) = delete;
    ~SpillVector() {
        if (fd_ >= 0) {
            if (data_) munmap(data_, cap_ * sizeof(T));
            ::close(fd_);
            ::unlink(path_.c_str());
        } else {
            std::free(data_);
        }
    }

    size_t size() const { return size_; }
    T* data() { return data_; }
    const T* data() const { return data_; }
    T& operator[](size_t i) { return data_[i]; }
    const T& operator[](size_t i) const { return data_[i]; }
    bool spilled() const { return fd_ >= 0; }

    void push_back(const T& v) {
        reserve(size_ + 1);
        data_[size_++] = v;
    }
    void append(const T* src, size_t n) {
        reserve(size_ + n);
        std::memcpy(data_ + size_, src, n * sizeof(T));
        size_ += n;
    }
    void append_fill(const T& v, size_t n) {
        reserve(size_ + n);
        for (size_t i = 0; i < n; i++) data_[size_ + i] = v;
        size_ += n;
    }

private:
    void reserve(size_t want) {
        if (want <= cap_) return;
        size_t ncap = cap_ ? cap_ : 4096;
        while (ncap < want) ncap += ncap / 2 + 4096;
        const size_t bytes = ncap * sizeof(T);
        if (fd_ < 0 && budget_ && bytes > budget_) {  // move to a file mapping
            fd_ = ::open(path_.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
            if (fd_ < 0) throw std::runtime_error("Cannot create temp file " + path_);
            if (ftruncate(fd_, off_t(bytes)) != 0) throw std::runtime_error("Cannot size temp file " + path_);
            void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
            if (p == MAP_FAILED) throw std::runtime_error("Cannot map temp file " + path_);
            if (size_) std::memcpy(p, data_, size_ * sizeof(T));
            std::free(data_);
            data_ = static_cast<T*>(p);
        } else if (fd_ >= 0) {
            if (ftruncate(fd_, off_t(bytes)) != 0) throw std::runtime_error("Cannot grow temp file " + path_);
            void* p = mremap(data_, cap_ * sizeof(T), bytes, MREMAP_MAYMOVE);
            if (p == MAP_FAILED) throw std::runtime_error("Cannot remap temp file " + path_);
            data_ = static_cast<T*>(p);
        } else {
            void* p = std::realloc(data_, bytes);
            if (!p) throw std::bad_alloc();
            data_ = static_cast<T*>(p);
        }
        cap_ = ncap;
    }

    std::string path_;
    size_t budget_;
    T* data_ = nullptr;
    size_t size_ = 0, cap_ = 0;
    int fd_ = -1;
};

}  // namespace swh
