// host_internal.h -- pieces the host-layer sources share (not part of the mirror's public surface, reseek_host.h):
// helper contexts with streams of their own, the pinned path-buffer pool, the team of contexts of a device list, and the
// pair-space driver.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <chrono>
#include <future>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <unordered_map>

#include "reseek_host.h"
#include "../rsk_internal.h"

namespace reseek_amd {

inline void check(int rc, const char *what)
{
    if (rc != RSK_OK) throw std::runtime_error(std::string(what) + ": " + rsk_last_error());
}

// A secondary context of this device, with a non-blocking stream of its own: its kernels run next to the primary
// context's (the long-chain job's X-drop tail under the alignment job's kernels, batch k + 1's uploads under batch k).
// Audited for this (r02): every entry point queues its copies / memsets / kernels on the context's stream and the host
// callers call rsk_ctx_sync before their own synchronous copies; chain sets are uploaded with synchronous copies before
// any context uses them.  RSK_OWN_STREAMS=0 puts every context back on the default stream.
struct SecondaryCtx {
    rsk_ctx *c = nullptr;
    hipStream_t st = nullptr;
    int device = -1;
    // Idle secondary contexts are kept per device and handed out again: a context's allocator pool holds the scratch of
    // its last job (tens of GB of X-drop trace, the SW trace blocks), and hipMalloc / hipFree of blocks that size cost
    // hundreds of ms -- per search and, with a streamed -db file, per batch.  rsk_ctx_trim() releases them.
    // A context is handed back to the ROLE it served (second alignment stage / long-chain job / -db loader): the roles'
    // scratch differs by orders of magnitude, and a 26 GB trace block that has to be allocated again costs 0.7 s on
    // some hosts.
    struct Idle { int device; rsk_ctx *c; hipStream_t st; const char *role; };
    const char *role = "";
    static std::mutex &Lock() { static std::mutex m; return m; }
    // The parked contexts keep their pools (that is the point), so two things bound what they can hold on to: the library's
    // out-of-memory ladder (rsk_dev_malloc) destroys the idle contexts of the device before any allocation fails -- the
    // hook is registered with the list -- and the list is destroyed with the process.
    struct IdleHolder {
        std::vector<Idle> v;
        IdleHolder() { rsk_set_oom_hook(&SecondaryCtx::Trim); }
        // Static destruction runs in an unspecified order relative to the HIP runtime's own teardown: no HIP call here.
        // The driver reclaims the parked contexts with the process; rsk_ctx_trim(ctx) / rsk_shutdown() release them earlier.
        ~IdleHolder() { rsk_set_oom_hook(nullptr); }
    };
    static std::vector<Idle> &IdleList() { static IdleHolder h; return h.v; }
    void Create(int dev, const char *Role)
    {
        device = dev;
        role = Role;
        {
            std::lock_guard<std::mutex> g(Lock());
            auto &v = IdleList();
            for (size_t k = 0; k < v.size(); ++k)
                if (v[k].device == dev && strcmp(v[k].role, Role) == 0) { c = v[k].c; st = v[k].st; v.erase(v.begin() + k); return; }
        }
        rsk_device_guard on(dev);                                       // the calling thread keeps ITS current device (a stream belongs to the device current at creation)
        check(rsk_ctx_create(dev, &c), "rsk_ctx_create");
        if (!(getenv("RSK_OWN_STREAMS") && atoi(getenv("RSK_OWN_STREAMS")) == 0)) {
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return; }
            rsk_ctx_set_stream(c, (void *) st);
        }
    }
    ~SecondaryCtx()
    {
        if (!c) return;
        (void) rsk_ctx_sync(c);
        std::lock_guard<std::mutex> g(Lock());
        auto &v = IdleList();
        if (v.size() < 6) { v.push_back(Idle{ device, c, st, role }); return; }
        rsk_ctx_destroy(c);
        if (st) (void) hipStreamDestroy(st);
    }
    static void Trim(int dev)
    {
        std::lock_guard<std::mutex> g(Lock());
        auto &v = IdleList();
        for (size_t k = 0; k < v.size();)
            if (dev < 0 || v[k].device == dev) {
                rsk_ctx_destroy(v[k].c);
                if (v[k].st) (void) hipStreamDestroy(v[k].st);
                v.erase(v.begin() + k);
            } else ++k;
    }
};

// Page-locked path buffers of the alignment batches (~120 MB each for 350 k pairs): ONE pool per process, kept between calls --
// hipHostMalloc / hipHostFree cost ~0.1 ms per MB, and r01-r05 built and tore down a pool (four buffers) per pass over a
// streamed database batch: ~60 ms of every 1.6-s pass of a -verysensitive search (r06 trace of configs[4]).  At most
// PINNED_IDLE_MAX buffers stay parked; rsk_ctx_trim / rsk_shutdown release them (no HIP call from a static destructor).
struct PinnedPool {
    enum { PINNED_IDLE_MAX = 6 };
    std::mutex lock;
    std::vector<std::pair<char *, size_t> > idle;
    char *Get(size_t bytes, size_t &cap)
    {
        {
            std::lock_guard<std::mutex> g(lock);
            // the smallest parked buffer that fits
            size_t best = idle.size();
            for (size_t k = 0; k < idle.size(); ++k)
                if (idle[k].second >= bytes && (best == idle.size() || idle[k].second < idle[best].second)) best = k;
            if (best < idle.size()) {
                char *p = idle[best].first;
                cap = idle[best].second;
                idle.erase(idle.begin() + (ptrdiff_t) best);
                return p;
            }
            if (!idle.empty()) { (void) hipHostFree(idle.back().first); idle.pop_back(); }      // too small: replace it
        }
        void *p = nullptr;
        cap = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) throw std::runtime_error("hipHostMalloc failed for the path buffer");
        return (char *) p;
    }
    void Put(char *p, size_t cap)
    {
        std::lock_guard<std::mutex> g(lock);
        if (idle.size() >= PINNED_IDLE_MAX) { (void) hipHostFree(p); return; }
        idle.emplace_back(p, cap);
    }
    void Trim()
    {
        std::lock_guard<std::mutex> g(lock);
        for (auto &b : idle) (void) hipHostFree(b.first);
        idle.clear();
    }
    static PinnedPool &Shared() { static PinnedPool *p = new PinnedPool; return *p; }      // never destroyed: see above
};

// One context per entry of the device list, on streams of their own; parked between calls like every helper context.
struct DeviceTeam {
    std::vector<std::unique_ptr<SecondaryCtx> > member;
    explicit DeviceTeam(const std::vector<int> &Devices)
    {
        static const char *const roles[] = { "team0", "team1", "team2", "team3", "team4", "team5", "team6", "team7" };
        for (size_t k = 0; k < Devices.size(); ++k) {
            member.emplace_back(new SecondaryCtx);
            member.back()->Create(Devices[k], roles[k % 8]);
        }
    }
    rsk_ctx *ctx(size_t k) const { return member[k]->c; }
};

// Scores the pair space S x SrcA (Mu filter, alignment batches, long-chain job) and replays the hits; defined in runpairs.cpp.
// Self with SelfOffset >= 0 is (part of) one SHARD of a self search: B = the chains [SelfOffset, SelfOffset + NB)
// of the set, A = its chains [0, NA) with NA = SelfOffset (the rectangle above the shard's triangle) or up to
// SelfOffset + NB; the pairs i <= SelfOffset + j are scored.
// One shard of a self search as a WINDOW of the set's length order (r06): the pairs whose longer member stands at positions
// [RankLo, RankHi), plus piece ShardIndex of ShardCount equal contiguous pieces of the long-chain (MKF) pair list.
struct SelfWindow { uint32_t RankLo = 0, RankHi = 0, ShardIndex = 0, ShardCount = 1; };
void RunPairs(DBSearcher &S, DBSearcher &SrcA, bool Self, int64_t SelfOffset = -1, const SelfWindow *Win = nullptr);

}   // namespace reseek_amd
