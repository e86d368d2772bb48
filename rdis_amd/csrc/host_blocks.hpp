// host_blocks.hpp -- where the host-side index arrays of a plan live.
//
// A one-shot call (rdis_hip_cgd_batch = one SubspaceOptimizer::optimize) builds some ten megabytes of
// int32 tables for ladybug as one component, uploads them and drops them.  From the C library's heap
// every such array is a fresh mapping: the kernel zeroes its pages on first touch and takes them back
// when the call ends -- measured, that was most of the host time of such a call.  Arrays of 64 KB and
// more therefore come from a small process-wide cache of blocks that were touched before (size classes
// of a quarter octave, at most CACHE_LIMIT bytes kept); smaller ones go to malloc as ever.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

namespace rdis_hip {

class HostBlocks {
public:
    static constexpr size_t MIN_CACHED = 64u << 10;
    static constexpr size_t CACHE_LIMIT = 256u << 20;

    // (never destroyed: a plan dropped by a static destructor of the caller's may still hand its blocks back;
    // the process's exit returns the memory)
    static HostBlocks& get() { static HostBlocks* b = new HostBlocks; return *b; }

    // the size class a request is rounded up to: 2^k * {1, 1.25, 1.5, 1.75}
    static size_t size_class(size_t bytes) {
        size_t p = MIN_CACHED;
        while (2 * p < bytes) p *= 2;           // p < bytes <= 2 p  (or bytes <= MIN_CACHED)
        if (bytes <= p) return p;
        const size_t q = p / 4;
        return p + (bytes - p + q - 1) / q * q;
    }
    void* take(size_t bytes) {
        if (bytes < MIN_CACHED) return std::malloc(bytes ? bytes : 1);
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t i = free_.size(); i-- > 0;)
                if (free_[i].bytes == cls) {
                    void* p = free_[i].p;
                    free_[i] = free_.back();
                    free_.pop_back();
                    held_ -= cls;
                    return p;
                }
        }
        void* p = nullptr;   // page-aligned, like the mappings the C library would hand out at this size
        return posix_memalign(&p, 4096, cls) == 0 ? p : nullptr;
    }
    size_t held_bytes() { std::lock_guard<std::mutex> g(m_); return held_; }   // what the cache holds right now
    void give(void* p, size_t bytes) {
        if (!p) return;
        if (bytes >= MIN_CACHED) {
            const size_t cls = size_class(bytes);
            std::lock_guard<std::mutex> g(m_);
            if (held_ + cls <= CACHE_LIMIT) {
                free_.push_back({p, cls});
                held_ += cls;
                return;
            }
        }
        std::free(p);
    }

private:
    struct Block { void* p; size_t bytes; };
    std::mutex m_;
    std::vector<Block> free_;
    size_t held_ = 0;
};

template <class T>
struct BlockAlloc {
    using value_type = T;
    BlockAlloc() = default;
    template <class U> BlockAlloc(const BlockAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = HostBlocks::get().take(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) { HostBlocks::get().give(p, n * sizeof(T)); }
    template <class U> bool operator==(const BlockAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const BlockAlloc<U>&) const { return false; }
};

using ivec = std::vector<int, BlockAlloc<int>>;
using cvec = std::vector<char, BlockAlloc<char>>;

}  // namespace rdis_hip
