// hip_emu.cpp -- fiber scheduler of the TEST-ONLY HIP emulation (see hip_emu.h).
#include "hip_emu.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <pthread.h>
#include <thread>

namespace emu {
thread_local Fiber* cur = nullptr;
thread_local uint3_emu block_idx{0, 0, 0};
dim3 block_dim, grid_dim;
thread_local unsigned char* dyn_smem = nullptr;
thread_local uint64_t xchg[4096];
thread_local unsigned char xchg_wide[4096][64];

static thread_local ucontext_t sched_ctx;
static thread_local std::vector<Fiber> fibers;
static thread_local std::vector<unsigned char> smem;
static thread_local Group block_group;
static thread_local Group wave_groups[64];
static const std::function<void()>* body_ptr = nullptr;
static const size_t kStack = 256 * 1024;

void yield() { swapcontext(&cur->ctx, &sched_ctx); }

static void wait_group(Group& g, int size) {
    int gen = g.gen;
    if (++g.count == size) {
        g.count = 0;
        g.gen++;
    } else {
        while (g.gen == gen) yield();
    }
}
static int flat_tid() { return cur->tid.x + block_dim.x * (cur->tid.y + block_dim.y * cur->tid.z); }
int lane_id() { return flat_tid() & 63; }
void sync_block() { wait_group(block_group, (int)(block_dim.x * block_dim.y * block_dim.z)); }
void sync_wave() {
    int n = (int)(block_dim.x * block_dim.y * block_dim.z);
    int w = flat_tid() >> 6;
    int size = (n - w * 64) < 64 ? (n - w * 64) : 64;
    wait_group(wave_groups[w], size);
}

static void trampoline() {
    (*body_ptr)();
    cur->done = true;
    swapcontext(&cur->ctx, &sched_ctx);
}

// one workgroup, on the calling thread: its work-items are fibers that switch at barriers and wave exchanges
static void run_block(uint3_emu idx, dim3 block, size_t shmem) {
    const size_t n = (size_t)block.x * block.y * block.z;
    if (fibers.size() < n) {
        size_t old = fibers.size();
        fibers.resize(n);
        for (size_t i = old; i < n; i++) fibers[i].stack = (char*)malloc(kStack);
    }
    if (smem.size() < shmem + 64) smem.resize(shmem + 64);
    dyn_smem = (unsigned char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
    block_idx = idx;
    block_group = Group();
    for (auto& g : wave_groups) g = Group();
    size_t t = 0;
    for (unsigned z = 0; z < block.z; z++)
        for (unsigned y = 0; y < block.y; y++)
            for (unsigned x = 0; x < block.x; x++, t++) {
                Fiber& f = fibers[t];
                f.tid = {x, y, z};
                f.done = false;
                getcontext(&f.ctx);
                f.ctx.uc_stack.ss_sp = f.stack;
                f.ctx.uc_stack.ss_size = kStack;
                f.ctx.uc_link = &sched_ctx;
                makecontext(&f.ctx, trampoline, 0);
            }
    size_t remaining = n;
    while (remaining) {
        for (size_t i = 0; i < n; i++) {
            if (fibers[i].done) continue;
            cur = &fibers[i];
            swapcontext(&sched_ctx, &cur->ctx);
            if (fibers[i].done) remaining--;
        }
    }
    cur = nullptr;
    dyn_smem = nullptr;
}

// A pool of host threads takes the workgroups of a launch off a shared counter (TVM_EMU_THREADS, default: the
// hardware's, at most 8; 1 = everything on the caller's thread).  launch() returns when every workgroup has run.
namespace {
struct Pool {
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread> workers;
    uint64_t job = 0;           // generation of the current launch
    size_t total = 0, active = 0;
    std::atomic<size_t> next{0};
    dim3 grid, block;
    size_t shmem = 0;
    bool stop = false;

    void drain() {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= total) return;
            uint3_emu idx;
            idx.x = (unsigned)(b % grid.x);
            idx.y = (unsigned)((b / grid.x) % grid.y);
            idx.z = (unsigned)(b / ((size_t)grid.x * grid.y));
            run_block(idx, block, shmem);
        }
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            wake.wait(lock, [&] { return stop || job != seen; });
            if (stop) return;
            seen = job;
            lock.unlock();
            drain();
            lock.lock();
            if (--active == 0) done.notify_all();
        }
    }
    void run(dim3 g, dim3 b, size_t sh) {
        const size_t blocks = (size_t)g.x * g.y * g.z;
        static const int n_threads = [] {
            const char* e = getenv("TVM_EMU_THREADS");
            int n = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
            return n < 1 ? 1 : n;
        }();
        grid = g, block = b, shmem = sh, total = blocks;
        next.store(0);
        if (n_threads == 1 || blocks == 1) {
            drain();
            return;
        }
        {
            std::unique_lock<std::mutex> lock(m);
            while ((int)workers.size() < n_threads - 1) workers.emplace_back([this] { worker(); });
            active = workers.size();
            job++;
        }
        wake.notify_all();
        drain();  // the caller's thread works too
        std::unique_lock<std::mutex> lock(m);
        done.wait(lock, [&] { return active == 0; });
    }
    ~Pool() {
        {
            std::unique_lock<std::mutex> lock(m);
            stop = true;
        }
        wake.notify_all();
        for (auto& w : workers) w.join();
    }
};
Pool* g_pool = nullptr;  // never destroyed: worker threads may outlive static destruction order otherwise
Pool& pool() {
    static const bool registered = [] {
        // a forked child has none of the parent's worker threads: it starts a pool of its own (the parent's object is leaked)
        pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; });
        return true;
    }();
    (void)registered;
    if (!g_pool) g_pool = new Pool();
    return *g_pool;
}
std::mutex launch_mutex;  // launches from different host threads (two contexts) run one after the other
}  // namespace

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    size_t n = (size_t)block.x * block.y * block.z;
    if (n == 0 || n > 1024 || (size_t)grid.x * grid.y * grid.z == 0) {
        fprintf(stderr, "emu::launch: bad launch geometry (block %zu)\n", n);
        abort();
    }
    std::lock_guard<std::mutex> guard(launch_mutex);
    block_dim = block;
    grid_dim = grid;
    body_ptr = &body;
    pool().run(grid, block, shmem);
}
}  // namespace emu

hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
