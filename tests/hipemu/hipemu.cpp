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
void* dynamic_lds() { return g_dyn_lds; }
dim3& cur_thread_idx() { return g_cur->tid; }
int cur_lane() { return g_cur->linear & 63; }
void* wave_slot(int lane) { return g_xchg[g_cur->linear >> 6][lane]; }

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
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
