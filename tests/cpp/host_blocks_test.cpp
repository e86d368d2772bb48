// CPU test of rdis_amd/csrc/host_blocks.hpp (no HIP in it): size classes, reuse of touched blocks, the cache's limit.
#include "../../rdis_amd/csrc/host_blocks.hpp"

#include <cstdio>
#include <cstring>

using namespace rdis_hip;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main() {
    // size classes: 2^k * {1, 1.25, 1.5, 1.75}, never below the request, at most a quarter above it
    for (size_t b = HostBlocks::MIN_CACHED; b < (size_t)300 << 20; b += b / 7 + 13) {
        const size_t c = HostBlocks::size_class(b);
        CHECK(c >= b && c <= b + b / 4 + 1);
        CHECK(HostBlocks::size_class(c) == c);
    }
    HostBlocks& H = HostBlocks::get();
    // a block given back is the block taken next for the same class (its pages are touched already), page-aligned
    void* p = H.take(1 << 20);
    CHECK(p && ((size_t)p & 4095) == 0);
    std::memset(p, 0x5a, 1 << 20);
    H.give(p, 1 << 20);
    void* q = H.take((1 << 20) - 4096);     // same class
    CHECK(q == p);
    void* r = H.take(1 << 20);              // the cache is empty again: a new block
    CHECK(r && r != q);
    H.give(q, (1 << 20) - 4096); H.give(r, 1 << 20);
    // small requests bypass the cache
    void* s = H.take(100);
    CHECK(s);
    H.give(s, 100);
    // the vectors built on it behave like vectors, and their storage comes back
    {
        ivec v((size_t)1 << 18, 7);
        CHECK(v.size() == ((size_t)1 << 18) && v[12345] == 7);
        v.resize((size_t)1 << 19, -1);
        CHECK(v[0] == 7 && v.back() == -1);
        ivec w(v.begin(), v.end());
        CHECK(w == v);
    }
    // the cache keeps at most CACHE_LIMIT bytes: blocks beyond it go back to the C library
    {
        const size_t before = H.held_bytes(), each = (size_t)48 << 20;
        void* b[8];
        for (int i = 0; i < 8; ++i) { b[i] = H.take(each); CHECK(b[i]); }
        for (int i = 0; i < 8; ++i) H.give(b[i], each);     // 384 MB given
        const size_t after = H.held_bytes();
        CHECK(after <= HostBlocks::CACHE_LIMIT && after >= before + 4 * each);
        for (int i = 0; i < 8; ++i) { b[i] = H.take(each); CHECK(b[i]); }
        CHECK(H.held_bytes() == before);                     // all cached blocks of the class handed out again
        for (int i = 0; i < 8; ++i) H.give(b[i], each);
    }
    std::printf("ok\n");
    return 0;
}
