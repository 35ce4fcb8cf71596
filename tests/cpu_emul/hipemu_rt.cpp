// TEST INFRASTRUCTURE (tests/cpu_emul): the scheduler behind shim/hip/hip_runtime.h -- see that file.  One OS thread, one fiber per GPU thread
// of the block being run (its own stack, the sanitizer's fiber annotations around every switch), lanes of a wave resumed in lane order.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <mutex>
#include <vector>
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define HIPEMU_ASAN 1
#endif
#ifdef HIPEMU_ASAN
#include <sanitizer/common_interface_defs.h>
#endif

// The context switch: callee-saved registers on the old stack, stack pointers exchanged (x86-64 System V).  Not swapcontext: the sanitizer's
// interceptor for it clears the shadow of the whole fiber stack on every switch, several microseconds each, millions of times per launch.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n.globl hipemu_switch\n.type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
#if !defined(__x86_64__)
#error "hipemu_rt.cpp: the fiber switch is written for x86-64"
#endif

namespace hipemu {
dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {
enum { READY = 0, WAVE_WAIT = 1, BLOCK_WAIT = 2, DONE = 3 };
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE, line = 0, buf = 0;
    const char* what = "";
    unsigned tid = 0;
    void* fake = nullptr;
};
struct Wave { uint64_t vals[2][64]; uint64_t live[2]; unsigned gen = 0; };
constexpr size_t STACK = 512u << 10;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
std::vector<char*> g_stacks;
void* g_sched_sp = nullptr;
void* g_sched_fake = nullptr;
const void* g_sched_bottom = nullptr; size_t g_sched_size = 0;
Fiber* g_cur = nullptr;
unsigned long long g_switches = 0;
const std::function<void()>* g_body = nullptr;
char* g_lds = nullptr;
std::mutex g_launch_mutex;

char* stack_of(size_t i)
{
    while (g_stacks.size() <= i) {
        char* p = (char*)mmap(nullptr, STACK + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == (char*)MAP_FAILED) { std::perror("hipemu: mmap"); std::abort(); }
        mprotect(p, 4096, PROT_NONE);       // a guard page under the stack
        g_stacks.push_back(p + 4096);
    }
    return g_stacks[i];
}

void to_scheduler(Fiber* f, bool dying)
{
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(dying ? nullptr : &f->fake, g_sched_bottom, g_sched_size);
#endif
    hipemu_switch(&f->sp, g_sched_sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(f->fake, &g_sched_bottom, &g_sched_size);
#endif
}
void fiber_main()
{
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
    Fiber* f = g_cur;
    (*g_body)();
    f->state = DONE;
    to_scheduler(f, true);
    std::abort();       // a finished fiber is never resumed
}
void resume(Fiber* f)
{
    g_cur = f; g_switches++;
    t_threadIdx = dim3(f->tid, 0, 0);
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&g_sched_fake, f->stack, STACK);
#endif
    hipemu_switch(&g_sched_sp, f->sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#endif
}
[[noreturn]] void die_divergent(unsigned block, unsigned wave, unsigned n_threads)
{
    std::fprintf(stderr, "hipemu: divergent wave operation or deadlock in block %u, wave %u:\n", block, wave);
    for (unsigned l = 0; l < 64 && wave * 64 + l < n_threads; l++) {
        const Fiber& f = g_fibers[wave * 64 + l];
        std::fprintf(stderr, "  lane %2u: %s%s line %d\n", l, f.state == DONE ? "finished" : f.state == WAVE_WAIT ? "at " : f.state == BLOCK_WAIT ? "at __syncthreads" : "ready",
                     f.state == WAVE_WAIT ? f.what : "", f.state == DONE ? 0 : f.line);
    }
    std::abort();
}

void run_block(unsigned n_threads)
{
    const unsigned n_waves = (n_threads + 63) / 64;
    g_fibers.resize(n_threads);
    g_waves.assign(n_waves, Wave());
    for (unsigned t = 0; t < n_threads; t++) {
        Fiber& f = g_fibers[t];
        f.stack = stack_of(t); f.state = READY; f.tid = t; f.fake = nullptr; f.line = 0;
        // first switch into the fiber: six zero registers are popped, `ret` takes fiber_main's address and leaves the stack pointer where a
        // call would have (8 mod 16)
        void** top = (void**)(f.stack + STACK);
        top[-1] = nullptr; top[-2] = (void*)&fiber_main;
        for (int k = 3; k <= 8; k++) top[-k] = nullptr;
        f.sp = (void*)(top - 8);
    }
    unsigned done = 0;
    while (done < n_threads) {
        bool moved = false;
        for (unsigned t = 0; t < n_threads; t++)
            if (g_fibers[t].state == READY) { resume(&g_fibers[t]); moved = true; if (g_fibers[t].state == DONE) done++; }
        // waves whose live lanes have all arrived at the same cross-lane operation go on
        unsigned at_barrier = 0;
        for (unsigned w = 0; w < n_waves; w++) {
            unsigned waiting = 0, finished = 0, lanes = 0; int line = -1; bool same = true; uint64_t live = 0;
            for (unsigned l = 0; l < 64 && w * 64 + l < n_threads; l++) {
                const Fiber& f = g_fibers[w * 64 + l]; lanes++;
                if (f.state == DONE) finished++;
                else if (f.state == BLOCK_WAIT) at_barrier++;
                else if (f.state == WAVE_WAIT) { waiting++; live |= 1ull << l; if (line < 0) line = f.line; else if (line != f.line) same = false; }
            }
            if (waiting == 0) continue;
            if (waiting + finished != lanes) continue;      // the others are at __syncthreads: no progress possible, caught below
            if (!same) die_divergent(t_blockIdx.x, w, n_threads);
            Wave& wv = g_waves[w];
            wv.live[wv.gen & 1u] = live;
            wv.gen++;
            for (unsigned l = 0; l < 64 && w * 64 + l < n_threads; l++) if (g_fibers[w * 64 + l].state == WAVE_WAIT) g_fibers[w * 64 + l].state = READY;
            moved = true;
        }
        if (at_barrier > 0 && at_barrier + done == n_threads) {
            int line = -1;
            for (unsigned t = 0; t < n_threads; t++) if (g_fibers[t].state == BLOCK_WAIT) {
                if (line < 0) line = g_fibers[t].line;
                else if (line != g_fibers[t].line) { std::fprintf(stderr, "hipemu: threads of block %u wait at different __syncthreads (lines %d and %d)\n", t_blockIdx.x, line, g_fibers[t].line); std::abort(); }
                g_fibers[t].state = READY;
            }
            moved = true;
        }
        if (!moved && done < n_threads) {
            for (unsigned w = 0; w < n_waves; w++) for (unsigned l = 0; l < 64 && w * 64 + l < n_threads; l++) if (g_fibers[w * 64 + l].state == WAVE_WAIT) die_divergent(t_blockIdx.x, w, n_threads);
            std::fprintf(stderr, "hipemu: block %u cannot make progress\n", t_blockIdx.x); std::abort();
        }
    }
}
}  // namespace

int lane() { return (int)(g_cur->tid & 63u); }
void* dynamic_lds() { return g_lds; }

const uint64_t* wave_exchange(uint64_t v, const char* what, int line, uint64_t* live)
{
    Fiber* f = g_cur;
    Wave& w = g_waves[f->tid >> 6];
    const unsigned buf = w.gen & 1u;
    w.vals[buf][f->tid & 63u] = v;
    f->state = WAVE_WAIT; f->what = what; f->line = line;
    to_scheduler(f, false);
    *live = w.live[buf];
    return w.vals[buf];
}
void block_barrier(int line)
{
    Fiber* f = g_cur;
    f->state = BLOCK_WAIT; f->line = line;
    to_scheduler(f, false);
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body)
{
    std::lock_guard<std::mutex> lock(g_launch_mutex);
    const unsigned n_threads = block.x * block.y * block.z;
    if (n_threads == 0 || n_threads > 1024 || grid.x == 0) { std::fprintf(stderr, "hipemu: bad launch shape (%u blocks of %u threads)\n", grid.x, n_threads); std::abort(); }
    if (lds_bytes > (160u << 10)) { std::fprintf(stderr, "hipemu: %zu bytes of dynamic LDS\n", lds_bytes); std::abort(); }
    static const bool trace = std::getenv("HIPEMU_TRACE") != nullptr;
    static unsigned n_launch = 0;
    const auto t0 = std::chrono::steady_clock::now();
    g_body = &body;
    t_blockDim = block; t_gridDim = grid;
    g_lds = (char*)std::malloc(lds_bytes ? lds_bytes : 1);     // exactly what the launch asked for: an access past it is an error the sanitizer reports
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        t_blockIdx = dim3(bx, by, bz);
        run_block(n_threads);
    }
    std::free(g_lds); g_lds = nullptr; g_body = nullptr;
    if (trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "hipemu: launch #%u  %u x %u threads, %zu B LDS: %.1f ms, %llu resumes\n", ++n_launch, grid.x * grid.y * grid.z, n_threads, lds_bytes, ms, g_switches);
        g_switches = 0;
    }
}
}  // namespace hipemu
