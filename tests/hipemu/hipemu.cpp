// TEST INFRASTRUCTURE ONLY -- runtime of the HIP execution-model emulation (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <ucontext.h>

#include <vector>

namespace hipemu {

enum State { READY, WAIT_BLOCK, WAIT_WAVE, DONE };

struct Fiber {
    ucontext_t ctx;
    dim3 tid;
    int linear = 0;
    State state = DONE;
    char* stack = nullptr;
};

Fiber* g_cur = nullptr;
dim3 g_block_idx, g_block_dim, g_grid_dim;
static ucontext_t g_sched;
static std::vector<Fiber> g_fibers;
static const std::function<void()>* g_body = nullptr;
static unsigned char g_xchg[64][64][16];   // [wave][lane][16 bytes]
static const size_t kStack = 256 * 1024;

static unsigned char g_dyn_lds[160 * 1024] __attribute__((aligned(64)));
// the workgroup whose fiber runs: its dynamic LDS and its shuffle exchange slots (launch(): the one static set; launch_resident(): per block)
static unsigned char* g_lds_cur = g_dyn_lds;
static unsigned char (*g_xchg_cur)[64][16] = g_xchg;
void* dynamic_lds() { return g_lds_cur; }
dim3& cur_thread_idx() { return g_cur->tid; }
int cur_lane() { return g_cur->linear & 63; }
void* wave_slot(int lane) { return g_xchg_cur[g_cur->linear >> 6][lane]; }

static void yield_with(State s) {
    Fiber* f = g_cur;
    f->state = s;
    swapcontext(&f->ctx, &g_sched);
}
void sync_block() { yield_with(WAIT_BLOCK); }
void sync_wave() { yield_with(WAIT_WAVE); }

static void fiber_main() {
    (*g_body)();
    g_cur->state = DONE;
    swapcontext(&g_cur->ctx, &g_sched);
}

static void run_block(int nthreads) {
    if ((int)g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[t];
        f.linear = t;
        f.tid = dim3(t % g_block_dim.x, (t / g_block_dim.x) % g_block_dim.y, t / (g_block_dim.x * g_block_dim.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &g_sched;
        makecontext(&f.ctx, fiber_main, 0);
        f.state = READY;
    }
    int alive = nthreads;
    while (alive > 0) {
        bool progressed = false;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = g_fibers[t];
            if (f.state != READY) continue;
            g_cur = &f;
            swapcontext(&g_sched, &f.ctx);
            progressed = true;
            if (f.state == DONE) --alive;
        }
        bool released = false;
        // block barrier: all live fibers wait at it
        bool all_block = alive > 0;
        for (int t = 0; t < nthreads && all_block; ++t)
            if (g_fibers[t].state != DONE && g_fibers[t].state != WAIT_BLOCK) all_block = false;
        if (all_block) {
            for (int t = 0; t < nthreads; ++t)
                if (g_fibers[t].state == WAIT_BLOCK) g_fibers[t].state = READY;
            released = true;
        }
        for (int w = 0; w * 64 < nthreads; ++w) {
            bool all_wave = true, any = false;
            for (int t = w * 64; t < nthreads && t < (w + 1) * 64; ++t) {
                if (g_fibers[t].state == DONE) continue;
                if (g_fibers[t].state != WAIT_WAVE) all_wave = false;
                else any = true;
            }
            if (all_wave && any) {
                for (int t = w * 64; t < nthreads && t < (w + 1) * 64; ++t)
                    if (g_fibers[t].state == WAIT_WAVE) g_fibers[t].state = READY;
                released = true;
            }
        }
        if (!progressed && !released && alive > 0) {
            fprintf(stderr, "hipemu: deadlock (divergent barrier) in block (%u,%u,%u)\n", g_block_idx.x, g_block_idx.y, g_block_idx.z);
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    g_body = &body;
    g_grid_dim = grid;
    g_block_dim = block;
    const int nthreads = block.x * block.y * block.z;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                g_block_idx = dim3(x, y, z);
                run_block(nthreads);
            }
    g_body = nullptr;
}

// ---- resident launch: EVERY workgroup of the grid is alive at the same time (persistent kernels with inter-workgroup barriers) --------------
// Each block has its own fibers, dynamic LDS and exchange slots; the scheduler visits the blocks round-robin, one pass of ready fibers + barrier
// releases per visit. A fiber that polls another workgroup's flag calls spin_yield() (device build: s_sleep) and stays READY. Kernels launched
// this way must keep ALL their LDS in the dynamic allocation (`__shared__` statics of the emulation are one array for all blocks).
struct ResBlock {
    dim3 idx;
    std::vector<Fiber> fibers;
    unsigned char* lds = nullptr;
    unsigned char (*xchg)[64][16] = nullptr;
    int alive = 0;
};
static bool g_resident = false;

void spin_yield() {
    if (!g_resident) return;          // (an ordinary launch runs one block at a time: nobody else could make progress)
    yield_with(READY);
}

void launch_resident(dim3 grid, dim3 block, const std::function<void()>& body) {
    g_body = &body;
    g_grid_dim = grid;
    g_block_dim = block;
    g_resident = true;
    const int nthreads = block.x * block.y * block.z;
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    std::vector<ResBlock> blocks(nblocks);
    size_t bi = 0;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x, ++bi) {
                ResBlock& B = blocks[bi];
                B.idx = dim3(x, y, z);
                B.lds = (unsigned char*)aligned_alloc(64, 160 * 1024);
                B.xchg = (unsigned char(*)[64][16])malloc(sizeof(g_xchg));
                B.fibers.resize(nthreads);
                B.alive = nthreads;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = B.fibers[t];
                    f.stack = (char*)malloc(kStack);
                    f.linear = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &g_sched;
                    makecontext(&f.ctx, fiber_main, 0);
                    f.state = READY;
                }
            }
    size_t alive_blocks = nblocks;
    unsigned long long idle_rounds = 0;
    while (alive_blocks > 0) {
        bool progressed = false;
        for (ResBlock& B : blocks) {
            if (B.alive == 0) continue;
            g_block_idx = B.idx;
            g_lds_cur = B.lds;
            g_xchg_cur = B.xchg;
            for (int t = 0; t < nthreads; ++t) {
                Fiber& f = B.fibers[t];
                if (f.state != READY) continue;
                g_cur = &f;
                swapcontext(&g_sched, &f.ctx);
                progressed = true;
                if (f.state == DONE) --B.alive;
            }
            bool all_block = B.alive > 0;
            for (int t = 0; t < nthreads && all_block; ++t)
                if (B.fibers[t].state != DONE && B.fibers[t].state != WAIT_BLOCK) all_block = false;
            if (all_block) {
                for (int t = 0; t < nthreads; ++t)
                    if (B.fibers[t].state == WAIT_BLOCK) B.fibers[t].state = READY;
                progressed = true;
            }
            for (int w = 0; w * 64 < nthreads; ++w) {
                bool all_wave = true, any = false;
                for (int t = w * 64; t < nthreads && t < (w + 1) * 64; ++t) {
                    if (B.fibers[t].state == DONE) continue;
                    if (B.fibers[t].state != WAIT_WAVE) all_wave = false;
                    else any = true;
                }
                if (all_wave && any) {
                    for (int t = w * 64; t < nthreads && t < (w + 1) * 64; ++t)
                        if (B.fibers[t].state == WAIT_WAVE) B.fibers[t].state = READY;
                    progressed = true;
                }
            }
            if (B.alive == 0) --alive_blocks;
        }
        if (!progressed && ++idle_rounds > 4) {
            fprintf(stderr, "hipemu: deadlock in a resident launch (divergent barrier)\n");
            abort();
        }
    }
    for (ResBlock& B : blocks) {
        for (Fiber& f : B.fibers) free(f.stack);
        free(B.lds);
        free(B.xchg);
    }
    g_lds_cur = g_dyn_lds;
    g_xchg_cur = g_xchg;
    g_resident = false;
    g_body = nullptr;
}

}  // namespace hipemu

struct hipemu_event {
    double t;
};
static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

hipError_t hipMalloc(void** p, size_t bytes) {
    // guard bytes on both sides filled with NaN patterns would hide bugs; use plain aligned allocation + poison
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes ? bytes : 16) != 0) return hipErrorOutOfMemory;
    memset(q, 0xFF, bytes);   // poison: NaN for floats
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
// free "device" memory: 0 unless HIPEMU_FREE_BYTES says otherwise (the library's workspace placement holds candidate allocations only within half of it:
// emulated solves keep their one workspace, the placement test sets the variable)
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) {
    const char* e = getenv("HIPEMU_FREE_BYTES");
    *free_bytes = e ? (size_t)atoll(e) : 0;
    *total_bytes = (size_t)1 << 34;
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = malloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) { memset(p, value, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int) {
    memset(prop, 0, sizeof(*prop));
    prop->multiProcessorCount = 2;   // small "GPU" so that planners pick many small workgroups
    strcpy(prop->name, "hipemu");
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }      // launches run synchronously here: an event has always fired
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
