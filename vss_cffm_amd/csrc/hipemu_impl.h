// hipemu_impl.h -- TEST INFRASTRUCTURE: definitions for hipemu.h (included once, by the -DCFFM_EMU build).
#include "hipemu.h"
namespace emu {
thread_local Block* g_blk = nullptr;

static void run_block(Block& b, dim3 bid) {
    b.bid = bid;
    b.cur = 0;
    b.live = b.nthreads;
    b.arrive = 0;
    b.gen = 0;
    b.spins = 0;
    for (auto& w : b.waves) { w.count = 0; w.gen = 0; }
    for (int t = 0; t < b.nthreads; ++t) {
        Lane& l = b.lanes[t];
        l.done = false;
        l.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = b.stacks.data() + (size_t)t * kStack;
        l.ctx.uc_stack.ss_size = kStack;
        l.ctx.uc_link = nullptr;
        makecontext(&l.ctx, (void (*)())lane_entry, 0);
    }
    swapcontext(&b.home, &b.lanes[0].ctx);
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const int nthreads = block.x * block.y * block.z;
    if (nblocks == 0) return;
    if (nthreads > 64 * kMaxWaves) { fprintf(stderr, "[hipemu] block too large\n"); abort(); }
    int nworkers = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("CFFM_EMU_THREADS")) nworkers = atoi(e);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > nblocks) nworkers = (int)nblocks;
    std::atomic<long> next{0};
    auto worker = [&]() {
        Block b;
        b.bdim = block;
        b.gdim = grid;
        b.nthreads = nthreads;
        b.lanes.resize(nthreads);
        b.stacks.resize((size_t)nthreads * kStack);
        std::vector<char> smem(shmem + 64);
        b.dynsmem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
        b.body = &body;
        g_blk = &b;
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= nblocks) break;
            run_block(b, dim3(i % grid.x, (i / grid.x) % grid.y, i / ((long)grid.x * grid.y)));
        }
        g_blk = nullptr;
    };
    if (nworkers == 1) {
        std::thread t(worker);  // own thread: thread_local __shared__ arrays + fresh stack
        t.join();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}
}  // namespace emu
