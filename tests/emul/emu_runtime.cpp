// TEST INFRASTRUCTURE -- fiber-based workgroup emulator behind tests/emul/hip/hip_runtime.h.
// x86-64 only.  One OS thread runs one workgroup at a time: its GPU threads are fibers with
// private stacks, switched by a 14-instruction context switch; blocks of a grid are spread
// over a few OS threads.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace emu {
namespace {

constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;
constexpr size_t kLdsBytes = 64 * 1024;        // the default dynamic-LDS launch limit
constexpr size_t kLdsBytesBig = 160 * 1024;    // with the per-kernel opt-in (DS_LAUNCH_BIG_LDS)

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = false;
};

struct Wave {
    int arrived = 0;
    unsigned gen = 0;
    float buf[2][64];
};

struct Worker {
    std::vector<Fiber> fibers;
    Wave waves[kMaxThreads / 64];
    void *sched_sp = nullptr;
    int cur = 0, nthreads = 0;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    bool progressed = false;
    char *lds = nullptr;
    const std::function<void()> *body = nullptr;
    Dim3 tid{0, 0, 0}, bid{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
};

thread_local Worker *g_w = nullptr;

void yield() {
    Worker *w = g_w;
    emu_switch(&w->fibers[w->cur].sp, w->sched_sp);
}

void fiber_entry() {
    Worker *w = g_w;
    (*w->body)();
    w = g_w;
    w->fibers[w->cur].done = true;
    w->progressed = true;
    emu_switch(&w->fibers[w->cur].sp, w->sched_sp);
    abort();
}

void prepare_fiber(Fiber &f) {
    if (!f.stack) f.stack = (char *)aligned_alloc(64, kStack);
    uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;                       // fake return address of fiber_entry's "caller"
    *--sp = (void *)&fiber_entry;          // popped by emu_switch's ret
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    f.sp = (void *)sp;
    f.done = false;
}

void run_block(Worker *w, int nthreads, unsigned bx, unsigned grid) {
    w->nthreads = nthreads;
    w->bar_arrived = 0;
    w->bid = Dim3{bx, 0, 0};
    w->bdim = Dim3{(unsigned)nthreads, 1, 1};
    w->gdim = Dim3{grid, 1, 1};
    if ((int)w->fibers.size() < nthreads) w->fibers.resize(nthreads);
    for (int i = 0; i < nthreads; ++i) prepare_fiber(w->fibers[i]);
    for (auto &wv : w->waves) wv.arrived = 0;
    int remaining = nthreads;
    while (remaining) {
        w->progressed = false;
        for (int i = 0; i < nthreads; ++i) {
            Fiber &f = w->fibers[i];
            if (f.done) continue;
            w->cur = i;
            w->tid = Dim3{(unsigned)i, 0, 0};
            emu_switch(&w->sched_sp, f.sp);
            if (f.done) --remaining;
        }
        if (!w->progressed && remaining) {
            fprintf(stderr, "emu: deadlock in block %u (a thread exited before a barrier?)\n", bx);
            abort();
        }
    }
}

}  // namespace

Dim3 &Idx::tid() { return g_w->tid; }
Dim3 &Idx::bid() { return g_w->bid; }
Dim3 &Idx::bdim() { return g_w->bdim; }
Dim3 &Idx::gdim() { return g_w->gdim; }

float *dynamic_lds() { return (float *)g_w->lds; }

void syncthreads() {
    Worker *w = g_w;
    const unsigned gen = w->bar_gen;
    w->progressed = true;
    if (++w->bar_arrived == w->nthreads) {
        w->bar_arrived = 0;
        w->bar_gen = gen + 1;
    } else {
        while (g_w->bar_gen == gen) yield();
    }
}

void wave_exchange(float mine, float *all64) {
    Worker *w = g_w;
    Wave &wv = w->waves[w->cur >> 6];
    const unsigned gen = wv.gen;
    wv.buf[gen & 1][w->cur & 63] = mine;
    w->progressed = true;
    if (++wv.arrived == 64) {
        wv.arrived = 0;
        wv.gen = gen + 1;
    } else {
        while (wv.gen == gen) yield();
    }
    memcpy(all64, wv.buf[gen & 1], sizeof(float) * 64);
}

void launch(int grid, int block, size_t lds_bytes, const std::function<void()> &body, bool big_lds) {
    if (block <= 0 || block > kMaxThreads || (block & 63) || lds_bytes > (big_lds ? kLdsBytesBig : kLdsBytes) || grid <= 0) {
        fprintf(stderr, "emu: bad launch grid=%d block=%d lds=%zu\n", grid, block, lds_bytes);
        abort();
    }
    unsigned hw = std::thread::hardware_concurrency();
    int nworkers = (int)(hw ? hw : 4);
    if (nworkers > grid) nworkers = grid;
    std::atomic<int> next{0};
    auto work = [&]() {
        Worker *w = new Worker();
        w->lds = (char *)aligned_alloc(64, kLdsBytesBig);
        w->body = &body;
        g_w = w;
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= grid) break;
            // poison LDS so that reads of never-written shared memory show up as NaN
            memset(w->lds, 0xFF, lds_bytes ? lds_bytes : 64);
            run_block(w, block, (unsigned)b, (unsigned)grid);
        }
        for (auto &f : w->fibers) free(f.stack);
        free(w->lds);
        delete w;
        g_w = nullptr;
    };
    if (nworkers == 1) {
        std::thread t(work);     // always off the caller's stack/TLS
        t.join();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; ++i) ts.emplace_back(work);
        for (auto &t : ts) t.join();
    }
}

}  // namespace emu
