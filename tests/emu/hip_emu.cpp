// hip_emu.cpp -- fiber scheduler of the TEST-ONLY HIP emulation (see hip_emu.h).
#include "hip_emu.h"

#include <chrono>

namespace emu {
Fiber* cur = nullptr;
uint3_emu block_idx{0, 0, 0};
dim3 block_dim, grid_dim;
unsigned char* dyn_smem = nullptr;
uint64_t xchg[4096];
unsigned char xchg_wide[4096][64];

static ucontext_t sched_ctx;
static std::vector<Fiber> fibers;
static Group block_group;
static Group wave_groups[64];
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

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    size_t n = (size_t)block.x * block.y * block.z;
    if (n == 0 || n > 1024 || (size_t)grid.x * grid.y * grid.z == 0) {
        fprintf(stderr, "emu::launch: bad launch geometry (block %zu)\n", n);
        abort();
    }
    if (fibers.size() < n) {
        size_t old = fibers.size();
        fibers.resize(n);
        for (size_t i = old; i < n; i++) fibers[i].stack = (char*)malloc(kStack);
    }
    std::vector<unsigned char> smem(shmem + 64);
    dyn_smem = (unsigned char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
    block_dim = block;
    grid_dim = grid;
    body_ptr = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                block_idx = {bx, by, bz};
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
            }
    cur = nullptr;
    dyn_smem = nullptr;
}
}  // namespace emu

hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
