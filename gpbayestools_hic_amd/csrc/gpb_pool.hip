// gpb_pool.hip — a small cache of freed device buffers (host code only).
//
// On this runtime hipFree hands a large buffer back to the driver and the next hipMalloc of that size maps it afresh: creating a
// context for 63 GPs at N = 1000 (three 528 MB matrices) cost 96-100 ms every other time and 4 ms in between, a context for 7 GPs
// 6.7 against 0.2 ms (tools/micro/setup_cost.py).  Trainings create and destroy contexts all the time (a fit-only context for the
// search, one per emulator afterwards; every hold-out retraining), so the buffers a context releases are kept — up to
// GPB_POOL_MB megabytes per process (default 8192; 0 switches the cache off) — and handed to the next request of about that size
// on the same device.  A buffer only enters the cache through pool_free, which the contexts call after synchronising the stream
// that used it; its contents are whatever the last owner left (as after hipMalloc).
#include "gpb_internal.h"
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace gpb {

namespace {
struct Blk { void* p; size_t bytes; int device; unsigned long long seq; };
std::mutex g_m;
std::unordered_map<void*, Blk> g_live;        // buffers handed out by pool_malloc and not yet freed
std::vector<Blk> g_cache;
size_t g_cached = 0;
unsigned long long g_seq = 0;
constexpr size_t MIN_CACHED = 256u << 10;     // smaller buffers are not worth a slot

size_t cap_bytes() {
    static const size_t cap = [] {
        const char* e = getenv("GPB_POOL_MB");
        const long long mb = e ? atoll(e) : 8192;
        return (size_t)(mb > 0 ? mb : 0) << 20;
    }();
    return cap;
}
}  // namespace

hipError_t pool_malloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (bytes >= MIN_CACHED && cap_bytes() > 0) {
        std::lock_guard<std::mutex> lk(g_m);
        int best = -1;
        for (int i = 0; i < (int)g_cache.size(); ++i) {       // smallest cached block that fits with at most a quarter to spare
            const Blk& b = g_cache[i];
            if (b.device == dev && b.bytes >= bytes && b.bytes - bytes <= bytes / 4 &&
                (best < 0 || b.bytes < g_cache[best].bytes))
                best = i;
        }
        if (best >= 0) {
            Blk b = g_cache[best];
            g_cache.erase(g_cache.begin() + best);
            g_cached -= b.bytes;
            g_live[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {                                    // out of memory with buffers in the cache: give them back and retry
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_m);
    g_live[*p] = Blk{*p, bytes, dev, 0};
    return hipSuccess;
}

void pool_free(void* p) {
    if (!p) return;
    Blk b{p, 0, 0, 0};
    std::vector<void*> evict;
    {
        std::lock_guard<std::mutex> lk(g_m);
        auto it = g_live.find(p);
        if (it == g_live.end()) {                             // not one of ours (allocated with plain hipMalloc)
            evict.push_back(p);
        } else {
            b = it->second;
            g_live.erase(it);
            if (b.bytes < MIN_CACHED || b.bytes > cap_bytes()) {
                evict.push_back(p);
            } else {
                b.seq = ++g_seq;
                g_cache.push_back(b);
                g_cached += b.bytes;
                while (g_cached > cap_bytes() && !g_cache.empty()) {      // oldest first
                    int old = 0;
                    for (int i = 1; i < (int)g_cache.size(); ++i)
                        if (g_cache[i].seq < g_cache[old].seq) old = i;
                    evict.push_back(g_cache[old].p);
                    g_cached -= g_cache[old].bytes;
                    g_cache.erase(g_cache.begin() + old);
                }
            }
        }
    }
    for (void* q : evict) (void)hipFree(q);
}

void pool_trim() {
    std::vector<Blk> all;
    {
        std::lock_guard<std::mutex> lk(g_m);
        all.swap(g_cache);
        g_cached = 0;
    }
    for (const Blk& b : all) (void)hipFree(b.p);
}

}  // namespace gpb
