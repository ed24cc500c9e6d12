// TEST INFRASTRUCTURE: fiber runtime behind tests/sim/hip_host_shim.h (see there).
#include <ucontext.h>
#include <atomic>
#include <thread>
#include <vector>
#include "hip_host_shim.h"

namespace storm { thread_local __attribute__((aligned(16))) char smem[160 * 1024]; }

namespace simrt {
namespace {
enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
constexpr size_t STACK = 96 * 1024;
struct Fiber { ucontext_t ctx; int state; Idx tid; };
struct Wave { int arrived, ndone, nlanes; const void* in[64]; alignas(16) char out[64][64]; };
struct Block {
    dim3 grid, block; Idx bid; int nthreads, cur, n_done, n_barrier;
    std::vector<Fiber> f; std::vector<Wave> waves; ucontext_t sched;
    const std::function<void()>* body;
};
thread_local Block* g_blk = nullptr;
thread_local std::vector<char> g_stacks;

void fiber_main() {
    Block& B = *g_blk;
    (*B.body)();
    Fiber& me = B.f[B.cur];
    me.state = DONE;
    B.n_done++;
    B.waves[B.cur / 64].ndone++;
    swapcontext(&me.ctx, &B.sched);
}

void run_block(Block& B) {
    g_blk = &B;
    if (g_stacks.size() < STACK * (size_t)B.nthreads) g_stacks.resize(STACK * (size_t)B.nthreads);
    const unsigned bx = B.block.x, by = B.block.y;
    for (int i = 0; i < B.nthreads; ++i) {
        Fiber& f = B.f[i];
        f.state = RUN;
        f.tid = Idx{(unsigned)i % bx, ((unsigned)i / bx) % by, (unsigned)i / (bx * by)};
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = g_stacks.data() + STACK * (size_t)i;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_main, 0);
    }
    const int nw = (B.nthreads + 63) / 64;
    B.waves.assign(nw, Wave());
    for (int w = 0; w < nw; ++w) { B.waves[w].arrived = 0; B.waves[w].ndone = 0; B.waves[w].nlanes = std::min(64, B.nthreads - 64 * w); }
    B.n_done = 0; B.n_barrier = 0;
    while (B.n_done < B.nthreads) {
        bool progressed = false;
        for (int i = 0; i < B.nthreads; ++i) {
            if (B.f[i].state != RUN) continue;
            B.cur = i;
            swapcontext(&B.sched, &B.f[i].ctx);
            progressed = true;
        }
        if (B.n_barrier > 0 && B.n_barrier == B.nthreads - B.n_done) {
            for (auto& f : B.f) if (f.state == WAIT_BLOCK) f.state = RUN;
            B.n_barrier = 0;
            progressed = true;
        }
        if (!progressed) { fprintf(stderr, "simrt: deadlock (divergent barrier / collective?)\n"); abort(); }
    }
    g_blk = nullptr;
}
}  // namespace

Idx thread_idx() { return g_blk->f[g_blk->cur].tid; }
Idx block_idx() { return g_blk->bid; }
dim3 block_dim() { return g_blk->block; }
dim3 grid_dim() { return g_blk->grid; }

void block_barrier() {
    Block& B = *g_blk;
    Fiber& me = B.f[B.cur];
    me.state = WAIT_BLOCK;
    B.n_barrier++;
    swapcontext(&me.ctx, &B.sched);
}

void wave_collective(const void* my_in, void* my_out, size_t out_bytes, CollFn fn, long long ctx) {
    Block& B = *g_blk;
    const int tid = B.cur, w = tid / 64, lane = tid % 64;
    Wave& W = B.waves[w];
    W.in[lane] = my_in;
    W.arrived++;
    const int active = W.nlanes - W.ndone;
    if (W.arrived == active) {
        if (active != W.nlanes) { fprintf(stderr, "simrt: wave collective with exited lanes\n"); abort(); }
        fn(W.in, W.out, W.nlanes, ctx);
        W.arrived = 0;
        for (int l = 0; l < W.nlanes; ++l) if (B.f[w * 64 + l].state == WAIT_WAVE) B.f[w * 64 + l].state = RUN;
    } else {
        Fiber& me = B.f[tid];
        me.state = WAIT_WAVE;
        swapcontext(&me.ctx, &B.sched);
    }
    memcpy(my_out, W.out[lane], out_bytes);
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    if (lds_bytes > sizeof(storm::smem)) { fprintf(stderr, "simrt: LDS request %zu too large\n", lds_bytes); abort(); }
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    unsigned hw = std::thread::hardware_concurrency();
    const int nworkers = (int)std::max(1LL, std::min<long long>(hw ? hw : 4, nblocks));
    std::atomic<long long> next(0);
    auto worker = [&]() {
        Block B;
        B.grid = grid; B.block = block; B.nthreads = nthreads; B.body = &body;
        B.f.resize(nthreads);
        for (;;) {
            const long long k = next.fetch_add(1);
            if (k >= nblocks) break;
            B.bid = Idx{(unsigned)(k % grid.x), (unsigned)((k / grid.x) % grid.y), (unsigned)(k / ((long long)grid.x * grid.y))};
            run_block(B);
        }
    };
    if (nworkers == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nworkers; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}
}  // namespace simrt
