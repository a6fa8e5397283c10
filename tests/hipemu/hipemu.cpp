// hipemu runtime: fiber context switch + block scheduler (TEST TOOL ONLY; see
// include/hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

#include <mutex>
#include <sys/mman.h>

// ---- minimal x86-64 SysV context switch: saves callee-saved regs on the old
// stack, stores old sp, loads new sp, restores and returns into the new fiber.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

thread_local BlockCtx* blk = nullptr;
thread_local Fiber* cur = nullptr;

static constexpr size_t STACK_BYTES = 256 * 1024;

void yield_to_main() {
    Fiber* f = cur;
    hipemu_switch(&f->sp, blk->main_sp);
}

static void fiber_entry() {
    (*blk->body)();
    cur->state = DONE;
    yield_to_main();
    fprintf(stderr, "hipemu: resumed a finished fiber\n");
    abort();
}

static void init_fiber(Fiber& f) {
    // stack top 16-aligned; after `ret` into fiber_entry rsp must be == 8 (mod 16)
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + STACK_BYTES) & ~uintptr_t(15);
    uint64_t* sp = reinterpret_cast<uint64_t*>(top - 8);     // X = top-8, X%16 == 8
    *--sp = reinterpret_cast<uint64_t>(&fiber_entry);        // return address at X-8
    for (int i = 0; i < 6; ++i) *--sp = 0;                   // rbp rbx r12 r13 r14 r15
    f.sp = sp;
}

static void release_wave_if_complete(BlockCtx& b, int w) {
    WaveBuf& wb = b.waves[w];
    if (wb.live > 0 && wb.waiting == wb.live) {
        int lo = w * WAVE, hi = std::min(b.nthreads, lo + WAVE);
        for (int i = lo; i < hi; ++i)
            if (b.fibers[i].state == WAIT_WAVE) b.fibers[i].state = RUNNABLE;
        wb.waiting = 0;
        wb.phase ^= 1;
    }
}

static void release_block_if_complete(BlockCtx& b) {
    if (b.live > 0 && b.block_waiting == b.live) {
        for (auto& f : b.fibers)
            if (f.state == WAIT_BLOCK) f.state = RUNNABLE;
        b.block_waiting = 0;
    }
}

static void run_block(BlockCtx& b) {
    const int n = b.nthreads;
    const int nw = (n + WAVE - 1) / WAVE;
    for (int w = 0; w < nw; ++w) {
        b.waves[w].phase = 0;
        b.waves[w].waiting = 0;
        b.waves[w].live = std::min(WAVE, n - w * WAVE);
    }
    b.block_waiting = 0;
    b.live = n;
    for (int i = 0; i < n; ++i) {
        Fiber& f = b.fibers[i];
        f.lin = i;
        f.tid = dim3(i % b.bdim.x, (i / b.bdim.x) % b.bdim.y, i / (b.bdim.x * b.bdim.y));
        f.state = RUNNABLE;
        init_fiber(f);
    }
    blk = &b;
    while (b.live > 0) {
        bool progressed = false;
        for (int i = 0; i < n; ++i) {
            Fiber& f = b.fibers[i];
            if (f.state != RUNNABLE) continue;
            progressed = true;
            cur = &f;
            hipemu_switch(&b.main_sp, f.sp);
            const int w = i / WAVE;
            if (f.state == WAIT_WAVE) {
                b.waves[w].waiting++;
                release_wave_if_complete(b, w);
            } else if (f.state == WAIT_BLOCK) {
                b.block_waiting++;
                release_block_if_complete(b);
            } else if (f.state == DONE) {
                b.live--;
                b.waves[w].live--;
                release_wave_if_complete(b, w);
                release_block_if_complete(b);
            }
        }
        if (!progressed) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live, %d at __syncthreads\n",
                    b.bid.x, b.bid.y, b.bid.z, b.live, b.block_waiting);
            abort();
        }
    }
    cur = nullptr;
    blk = nullptr;
}

// number of workgroups that are resident at a time (= OS worker threads; blocks are dispatched in
// linear-id order as workers free up, like the hardware dispatcher)
int worker_count() {
    static int hw = [] {
        const char* e = getenv("HIPEMU_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return std::max(1, v);
    }();
    return hw;
}

void launch_impl(dim3 grid, dim3 block, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nblocks <= 0 || nthreads <= 0) return;
    const int hw = worker_count();
    static const bool trace = getenv("HIPEMU_TRACE") != nullptr;
    if (trace) fprintf(stderr, "hipemu: launch grid (%u,%u,%u) block %d\n", grid.x, grid.y, grid.z, nthreads);
    const int nworkers = (int)std::min<long>(hw, nblocks);
    std::atomic<long> next{0};
    auto worker = [&]() {
        BlockCtx b;
        b.bdim = block;
        b.gdim = grid;
        b.nthreads = nthreads;
        b.body = &body;
        b.fibers.resize(nthreads);
        b.waves.resize((nthreads + WAVE - 1) / WAVE);
        // lazily committed: only the pages a fiber touches cost memory
        const size_t stack_bytes = (size_t)nthreads * STACK_BYTES;
        char* stacks = (char*)mmap(nullptr, stack_bytes, PROT_READ | PROT_WRITE,
                                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == (char*)MAP_FAILED) { fprintf(stderr, "hipemu: cannot map fiber stacks\n"); abort(); }
        for (int i = 0; i < nthreads; ++i) b.fibers[i].stack = stacks + (size_t)i * STACK_BYTES;
        for (;;) {
            long id = next.fetch_add(1);
            if (id >= nblocks) break;
            b.bid = dim3((unsigned)(id % grid.x), (unsigned)((id / grid.x) % grid.y),
                         (unsigned)(id / ((long)grid.x * grid.y)));
            run_block(b);
        }
        munmap(stacks, stack_bytes);
    };
    if (nworkers == 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace hipemu
