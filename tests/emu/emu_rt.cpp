// emu_rt.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product library).
// A small fiber-per-lane CPU emulator of the HIP execution model used by vc_rt.h under -DVC_EMU:
// blocks run one after another; the threads of a block are ucontext fibers scheduled round-robin and
// switch only at __syncthreads / wave shuffles / MFMA, so wave64 collectives and LDS semantics are
// reproduced deterministically.  MFMA follows the gfx950 operand maps quoted in vc_rt.h.
#include <ucontext.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <vector>
#include <stdint.h>

#define VC_EMU 1
#include "vc_rt.h"

namespace vcemu {

// Fiber switch.  glibc's swapcontext makes a sigprocmask syscall per switch (~0.3 us); on x86-64 a 12-instruction
// callee-saved-register switch is ~30x cheaper, which is what makes whole-engine runs under the emulator practical.
#if defined(__x86_64__)
#define VCEMU_FAST_SWITCH 1
extern "C" void vcemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl vcemu_switch
.type vcemu_switch,@function
vcemu_switch:
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
.size vcemu_switch,.-vcemu_switch
)");
#endif

struct Fiber {
#ifdef VCEMU_FAST_SWITCH
    void* sp = nullptr;
#else
    ucontext_t uc;
#endif
    Ctx ctx;
    bool done = false;
    char* stack = nullptr;
};

struct WaveState {
    int arrived = 0, gen = 0, alive = 0;
    float fslot[64];
    int islot[64];
    short a[64][8], b[64][8];
    int a32[64][8], b32[64][8]; int sa[64], sb[64];
    short tr[64][4];
    float fa[64], fb[64];
};

struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
#ifdef VCEMU_FAST_SWITCH
    void* sched_sp = nullptr;
#else
    ucontext_t sched;
#endif
    int cur = -1;
    int alive = 0;
    int bar_arrived = 0, bar_gen = 0;
    void (*tramp)(void*) = nullptr;
    void* args = nullptr;
    std::vector<unsigned char> dyn;
};

static thread_local BlockState* g_bs = nullptr;
static const size_t kStack = 256 * 1024;

Ctx* cur() { return &g_bs->fibers[g_bs->cur].ctx; }
void* dyn_shared() { return g_bs->dyn.data(); }

static void to_sched(BlockState* bs, Fiber& f) {
#ifdef VCEMU_FAST_SWITCH
    vcemu_switch(&f.sp, bs->sched_sp);
#else
    swapcontext(&f.uc, &bs->sched);
#endif
}
static void yield_() {
    BlockState* bs = g_bs;
    to_sched(bs, bs->fibers[bs->cur]);
}

static void fiber_main() {
    BlockState* bs = g_bs;
    bs->tramp(bs->args);
    Fiber& f = bs->fibers[bs->cur];
    f.done = true;
    bs->alive--;
    bs->waves[bs->cur / 64].alive--;
    to_sched(bs, f);
    abort();   // never resumed
}

void sync_block() {
    BlockState* bs = g_bs;
    int my = bs->bar_gen;
    bs->bar_arrived++;
    while (bs->bar_gen == my) {
        if (bs->bar_arrived >= bs->alive) { bs->bar_arrived = 0; bs->bar_gen++; break; }
        yield_();
    }
}

static void wave_sync() {
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    int my = w.gen;
    w.arrived++;
    while (w.gen == my) {
        if (w.arrived >= w.alive) { w.arrived = 0; w.gen++; break; }
        yield_();
    }
}

void sync_wave() { wave_sync(); }

float shfl_f(float v, int src) {
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    w.fslot[bs->cur & 63] = v;
    wave_sync();
    float r = w.fslot[src & 63];
    wave_sync();
    return r;
}
int shfl_i(int v, int src) {
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    w.islot[bs->cur & 63] = v;
    wave_sync();
    int r = w.islot[src & 63];
    wave_sync();
    return r;
}

void mfma_32x32x16_bf16(const short* a8, const short* b8, float* c16) {
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    int l = bs->cur & 63;
    memcpy(w.a[l], a8, 16);
    memcpy(w.b[l], b8, 16);
    wave_sync();
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c16[r];
        for (int k = 0; k < 16; ++k) {
            vc_bf16 av{(uint16_t)w.a[row + 32 * (k / 8)][k % 8]}, bv{(uint16_t)w.b[col + 32 * (k / 8)][k % 8]};
            acc += vc_bf16_to_f32(av) * vc_bf16_to_f32(bv);
        }
        c16[r] = acc;
    }
    wave_sync();
}

void ds_read_tr16(const void* p, short* out4) {      // semantics measured on gfx950 (tools/probe_tr16.hip)
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    int l = bs->cur & 63;
    memcpy(w.tr[l], p, 8);
    wave_sync();
    int g = l & ~15, i = l & 15;
    for (int j = 0; j < 4; ++j) out4[j] = w.tr[g + 4 * j + i / 4][i % 4];
    wave_sync();
}

void dma16(const void* gsrc, void* lds_piece) {      // global_load_lds_dwordx4: lane-linear 1 KiB piece (the emulator copies at once)
    BlockState* bs = g_bs;
    memcpy((char*)lds_piece + 16 * (bs->cur & 63), gsrc, 16);
}

void mfma_mx8_32x32x64(const int* a8, const int* b8, float* c16, int sa_byte, int sb_byte) {
    // layout probed on gfx950 (tools/probe_mx8_layout.hip): lane (row, half h): bytes 0-15 = k 16h.. of scale block 0, bytes 16-31 = k 32+16h..
    // of scale block 1; the scale of block b of a row comes from lane row + 32 b
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    int l = bs->cur & 63;
    memcpy(w.a32[l], a8, 32); memcpy(w.b32[l], b8, 32); w.sa[l] = sa_byte; w.sb[l] = sb_byte;
    wave_sync();
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c16[r];
        for (int blk = 0; blk < 2; ++blk) {
            const float sc = ldexpf(1.0f, w.sa[row + 32 * blk] - 127) * ldexpf(1.0f, w.sb[col + 32 * blk] - 127);
            float part = 0.f;
            for (int h = 0; h < 2; ++h) {
                const uint8_t* ab = (const uint8_t*)w.a32[row + 32 * h] + 16 * blk; const uint8_t* bb = (const uint8_t*)w.b32[col + 32 * h] + 16 * blk;
                for (int k = 0; k < 16; ++k) part += vc_e4m3_to_f32(ab[k]) * vc_e4m3_to_f32(bb[k]);
            }
            acc += part * sc;
        }
        c16[r] = acc;
    }
    wave_sync();
}
void mfma_32x32x2_f32(float a, float b, float* c16) {
    BlockState* bs = g_bs;
    WaveState& w = bs->waves[bs->cur / 64];
    int l = bs->cur & 63;
    w.fa[l] = a; w.fb[l] = b;
    wave_sync();
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c16[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.fa[row + 32 * k], w.fb[col + 32 * k], acc);
        c16[r] = acc;
    }
    wave_sync();
}

// Blocks of a launch are independent (the kernels use no inter-block communication), so they are spread over a few host
// threads; every thread owns its BlockState, fiber stacks and — VC_SHARED being `static thread_local` — its own LDS image.
static void run_blocks(void (*trampoline)(void*), void* args, dim3 grid, dim3 block, size_t shmem, unsigned first, unsigned stride) {
    int nthreads = block.x * block.y * block.z;
    BlockState bs;
    bs.tramp = trampoline; bs.args = args;
    bs.fibers.resize(nthreads);
    bs.waves.resize(nthreads / 64);
    bs.dyn.resize(shmem + 64);
    for (auto& f : bs.fibers) f.stack = (char*)malloc(kStack);
    g_bs = &bs;
    const unsigned total = grid.x * grid.y * grid.z;
    for (unsigned bi = first; bi < total; bi += stride) {
        const unsigned bx = bi % grid.x, by = (bi / grid.x) % grid.y, bz = bi / (grid.x * grid.y);
        bs.alive = nthreads; bs.bar_arrived = 0; bs.bar_gen = 0;
        for (auto& w : bs.waves) { w.arrived = 0; w.gen = 0; w.alive = 64; }
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = bs.fibers[t];
            f.done = false;
            f.ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.ctx.bid = dim3(bx, by, bz);
            f.ctx.bdim = block; f.ctx.gdim = grid;
#ifdef VCEMU_FAST_SWITCH
            {   // initial frame: 6 callee-saved slots + return address (16-byte aligned slot) -> fiber_main
                uintptr_t top = ((uintptr_t)f.stack + kStack - 64) & ~(uintptr_t)15;
                void** slot = (void**)top;
                slot[0] = (void*)fiber_main;
                for (int r = 1; r <= 6; ++r) slot[-r] = nullptr;
                f.sp = (void*)(slot - 6);
            }
#else
            getcontext(&f.uc);
            f.uc.uc_stack.ss_sp = f.stack; f.uc.uc_stack.ss_size = kStack; f.uc.uc_link = nullptr;
            makecontext(&f.uc, (void (*)())fiber_main, 0);
#endif
        }
        long idle_rounds = 0;
        while (bs.alive > 0) {
            for (int t = 0; t < nthreads; ++t) {
                if (bs.fibers[t].done) continue;
                bs.cur = t;
#ifdef VCEMU_FAST_SWITCH
                vcemu_switch(&bs.sched_sp, bs.fibers[t].sp);
#else
                swapcontext(&bs.sched, &bs.fibers[t].uc);
#endif
            }
            if (++idle_rounds > 100000000) { fprintf(stderr, "emu: deadlock?\n"); abort(); }
        }
    }
    for (auto& f : bs.fibers) free(f.stack);
    g_bs = nullptr;
}

void launch(void (*trampoline)(void*), void* args, dim3 grid, dim3 block, size_t shmem) {
    int nthreads = block.x * block.y * block.z;
    if (nthreads % 64 != 0) { fprintf(stderr, "emu: block size %d not a multiple of 64\n", nthreads); abort(); }
    const unsigned total = grid.x * grid.y * grid.z;
    static const unsigned hw = [] {
        const char* e = getenv("VCEMU_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return n < 1 ? 1u : (n > 8 ? 8u : n);
    }();
    const unsigned nt = total < hw ? total : hw;
    if (nt <= 1) { run_blocks(trampoline, args, grid, block, shmem, 0, 1); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(run_blocks, trampoline, args, grid, block, shmem, t, nt);
    run_blocks(trampoline, args, grid, block, shmem, 0, nt);
    for (auto& th : pool) th.join();
}

}  // namespace vcemu
