// simt.h -- TEST-ONLY: runs one workgroup of a kernel written against an execution-context type X (see thj_deflate_core.h) on the
// CPU.  Every thread of the workgroup is a fiber with its own stack; a fiber runs until it reaches a collective operation (a
// wave exchange or a workgroup barrier) and is resumed once all of its wave (workgroup) have arrived -- the semantics of the
// wave-wide builtins in wave-uniform control flow, which is the only place the kernels use them.  Threads run one at a time,
// so LDS / global "atomics" are plain read-modify-writes.  Never linked into libthj_hip.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace simt {

// ---- context switch: callee-saved registers and the stack pointer (x86-64 SysV)
extern "C" void simt_switch(void** save_sp, void* load_sp);
__asm__(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
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
.size simt_switch,.-simt_switch
)");

struct Block;
struct Fiber {
    void* sp = nullptr;
    std::vector<uint64_t> stack;
    bool done = false;
    int tid = 0;
    Block* blk = nullptr;
};

struct WaveState {
    uint64_t gen = 0; int arrived = 0;
    uint32_t buf[2][64];
};

struct Block {
    int nthreads = 0;
    std::vector<Fiber> fib;
    std::vector<WaveState> waves;
    uint64_t bar_gen = 0; int bar_arrived = 0;
    void* sched_sp = nullptr;
    int cur = -1;
    uint64_t events = 0;                             // arrivals and exits: a scheduler round without any is a deadlock
    std::function<void(int)> body;

    void yield() { simt_switch(&fib[(size_t)cur].sp, sched_sp); }
    // all 64 lanes of the caller's wave deposit v; returns the 64 values
    const uint32_t* exchange(int tid, uint32_t v) {
        WaveState& w = waves[(size_t)tid >> 6];
        const uint64_t g = w.gen;
        w.buf[g & 1][tid & 63] = v;
        ++events;
        if (++w.arrived == 64) { w.arrived = 0; ++w.gen; }
        else while (w.gen == g) yield();
        return w.buf[g & 1];
    }
    void barrier() {
        const uint64_t g = bar_gen;
        ++events;
        if (++bar_arrived == nthreads) { bar_arrived = 0; ++bar_gen; }
        else while (bar_gen == g) yield();
    }
};

inline void fiber_main(Fiber* f) {
    f->blk->body(f->tid);
    f->done = true;
    ++f->blk->events;
    for (;;) f->blk->yield();
}
extern "C" void simt_trampoline();
__asm__(R"(
.text
.globl simt_trampoline
.type simt_trampoline,@function
simt_trampoline:
    movq %r12, %rdi
    callq *%r13
    ud2
.size simt_trampoline,.-simt_trampoline
)");

// runs body(tid) for tid in [0, nthreads) as one workgroup (nthreads a multiple of 64)
inline void run_block(int nthreads, const std::function<void(Block&, int)>& body, size_t stack_bytes = 96 * 1024) {
    if (nthreads % 64) { fprintf(stderr, "simt: workgroup size must be a multiple of 64\n"); abort(); }
    Block b;
    b.nthreads = nthreads;
    b.fib.resize((size_t)nthreads);
    b.waves.resize((size_t)nthreads / 64);
    b.body = [&](int tid) { body(b, tid); };
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = b.fib[(size_t)t];
        f.tid = t; f.blk = &b;
        f.stack.assign(stack_bytes / 8, 0);
        // initial frame: r15 r14 r13 r12 rbx rbp, return address = trampoline; rsp 16-byte aligned at the call inside it
        uint64_t* top = f.stack.data() + f.stack.size();
        top = (uint64_t*)((uintptr_t)top & ~(uintptr_t)15);
        *--top = 0; *--top = 0;                       // after the switch's `ret` rsp is 16-byte aligned, as before a call
        *--top = (uint64_t)(uintptr_t)&simt_trampoline;
        *--top = 0;                                   // rbp
        *--top = 0;                                   // rbx
        *--top = (uint64_t)(uintptr_t)&f;             // r12 = argument
        *--top = (uint64_t)(uintptr_t)(void (*)(Fiber*))&fiber_main;   // r13 = function
        *--top = 0;                                   // r14
        *--top = 0;                                   // r15
        f.sp = top;
    }
    for (;;) {
        bool any = false;
        const uint64_t ev0 = b.events;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = b.fib[(size_t)t];
            if (f.done) continue;
            any = true;
            b.cur = t;
            simt_switch(&b.sched_sp, f.sp);
        }
        if (!any) break;
        if (b.events == ev0) { fprintf(stderr, "simt: deadlock (a collective operation some threads never reach)\n"); abort(); }
    }
}

}  // namespace simt
