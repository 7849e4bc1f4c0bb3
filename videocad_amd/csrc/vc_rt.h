// vc_rt.h — thin runtime layer for the VideoCAD hot-path kernels.
//
// Product build (hipcc --offload-arch=gfx950): plain HIP, wave64, MFMA builtins.
// Test build (-DVC_EMU, host C++): the same kernel sources run under tests/emu/emu_rt.cpp, a
// fiber-per-lane CPU emulator used ONLY by the `-m "not gpu"` tests to check kernel indexing logic
// (it is test infrastructure, never shipped, never a fallback: the product library has no CPU path).
#pragma once
// 16-bit storage format of this build.  Default: bf16 (libvcad_hip.so).  -DVC_H16: IEEE half (libvcad_hip_f16.so) — the same kernels, the same 16-bit
// tensors and MFMA rate, 10 mantissa bits instead of 7 (the precision class of the TF32 the reference allows itself, main.py:28) at the price of
// fp16's exponent range, which the engine covers with a power-of-two gradient scale (engine.hip: grad_scale).  The type is renamed so that kernel
// names in traces say which format ran.  bf16x3 and fp8 modes exist in the bf16 build only.
#ifdef VC_H16
#define vc_bf16 vc_f16
#define VC_S16_NAME "f16"
#else
#define VC_S16_NAME "bf16"
#endif
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#ifndef VC_EMU
// ------------------------------------------------------------------------------------------ HIP
#include <hip/hip_runtime.h>

#define VC_KERNEL static __global__
#define VC_DEV __device__ __forceinline__
#define VC_HD __host__ __device__ __forceinline__
#define VC_SHARED __shared__
#define VC_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; type* name = reinterpret_cast<type*>(name##_raw)

typedef hipStream_t vc_stream_t;

#define VC_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)

static inline int vc_memset_async(void* p, int v, size_t n, vc_stream_t s) { return (int)hipMemsetAsync(p, v, n, s); }
static inline int vc_memcpy_d2d_async(void* d, const void* s_, size_t n, vc_stream_t s) { return (int)hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s); }
typedef hipEvent_t vc_event_t;
static inline int vc_stream_create(vc_stream_t* s) { return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
static inline int vc_event_create(vc_event_t* ev) { return (int)hipEventCreateWithFlags(ev, hipEventDisableTiming); }
static inline int vc_event_record(vc_event_t ev, vc_stream_t s) { return (int)hipEventRecord(ev, s); }
static inline int vc_stream_wait_event(vc_stream_t s, vc_event_t ev) { return (int)hipStreamWaitEvent(s, ev, 0); }
static inline void vc_stream_destroy(vc_stream_t s) { if (s) (void)hipStreamDestroy(s); }
static inline void vc_event_destroy(vc_event_t ev) { if (ev) (void)hipEventDestroy(ev); }
static inline bool vc_has_side_streams() { return true; }
// is p memory a kernel of this library can address?  (the bound parameter buffer of an engine: a model that still lives in host memory must fail loudly —
// there is no CPU fallback — instead of faulting inside the first kernel)
static inline bool vc_is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}
// small host -> device table upload, ordered on the stream; the (pageable) host buffer may be reused when it returns
static inline int vc_upload(void* d, const void* h, size_t n, vc_stream_t s) { int rc = (int)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); return rc ? rc : (int)hipStreamSynchronize(s); }
static inline int vc_last_launch_error() { return (int)hipGetLastError(); }
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: the launchers remember it per device (bit = current device), so a
// process that moves a model from cuda:0 to cuda:1 (or drives several GPUs) gets the large-LDS kernels on every device it uses
static inline unsigned vc_device_bit() { int d = 0; (void)hipGetDevice(&d); return 1u << (d & 31); }

VC_DEV void vc_sync() { __syncthreads(); }
// orders a wave's own LDS writes before its later LDS reads (hardware is in-order per wave; this pins the compiler)
VC_DEV void vc_wave_barrier() { __builtin_amdgcn_wave_barrier(); }
VC_DEV float vc_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
VC_DEV int vc_shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
VC_DEV float vc_shfl(float v, int src) { return __shfl(v, src, 64); }
VC_DEV int vc_shfl(int v, int src) { return __shfl(v, src, 64); }

typedef float vc_f32x16 __attribute__((ext_vector_type(16)));
typedef float vc_f32x4 __attribute__((ext_vector_type(4)));
typedef short vc_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vc_bf16x8_hw __attribute__((ext_vector_type(8)));

// D(32x32) += A(32x16) * B(16x32), bf16 in / f32 acc.  lane l: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)   (cdna_hip_programming.md §3)
// (VC_H16: the 16-bit operands are IEEE halves — v_mfma_f32_32x32x16_f16, same shape, layout and rate)
VC_DEV vc_f32x16 vc_mfma_32x32x16_bf16(vc_s16x8 a, vc_s16x8 b, vc_f32x16 c) {
#ifdef VC_H16
    typedef _Float16 vc_f16x8_hw __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vc_f16x8_hw, a), __builtin_bit_cast(vc_f16x8_hw, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vc_bf16x8_hw, a), __builtin_bit_cast(vc_bf16x8_hw, b), c, 0, 0, 0);
#endif
}
// LDS transpose read (gfx950 ds_read_b64_tr_b16), semantics measured with tools/probe_tr16.hip: within each 16-lane
// group lane i points at 4 consecutive 16-bit elements D_i[0..3]; lane i receives { D_{4j + i/4}[i%4] : j = 0..3 }.
// With lane i -> &tile[k0 + i/4][n0 + 4*(i%4)] of a row-major [k][n] tile it returns tile[k0 + j][n0 + i]: column i.
typedef short vc_s16x4 __attribute__((ext_vector_type(4)));
VC_DEV vc_s16x4 vc_ds_read_tr16(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((vc_s16x4 __attribute__((address_space(3)))*)lds_ptr);
}
// Direct-to-LDS DMA (global_load_lds_dwordx4): each lane supplies its own 16-byte global source; the 64 lanes' data lands
// at lds_piece (wave-uniform) + lane*16, i.e. one contiguous 1 KiB piece.  Tracked by vmcnt like any VMEM load.
VC_DEV void vc_dma16(const void* gsrc, void* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}
// L2 warm-up of one cache line per lane (r06, attn_mfma.h): a 4-byte direct-to-LDS DMA into a 256-byte wave-private sink nobody reads — no destination
// register, so nothing ever waits for it except a vmcnt(0); written as asm so that hipcc does not order later LDS reads behind it (it only makes the
// compiler's own counted waits conservative: the untracked operation is older than anything they count).  Respects the exec mask.
VC_DEV void vc_prefetch_line(const void* gsrc, void* lds_sink) {
    unsigned keep;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_sink);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
VC_DEV float vc_expf_fast(float x) { return __expf(x); }
// a product that is rounded on its own: hipcc contracts `a * b - c` (and __fmul_rn, which is a plain multiply in the IR) into one FMA
VC_DEV float vc_mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
VC_DEV int vc_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }     // x is known to be wave-uniform: keep it in an SGPR
// one ticket per WAVE from a device counter: scalar atomic (SMEM path, lgkmcnt) — it does not touch the vmcnt accounting of a DMA ring
// (tools/probe_satomic.hip: 4096 distinct tickets on MI355X).  Waits for the result.
VC_DEV int vc_wave_ticket(int* ctr) {
    int r;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(ctr), "0"(1) : "memory");
    return r;
}
VC_DEV uint64_t vc_uniform64(uint64_t x) {     // 64-bit value known to be wave-uniform: both halves into SGPRs
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}
template <int N> VC_DEV void vc_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that does NOT drain vmcnt (DMA stays in flight across it); LDS reads/writes are ordered around it
VC_DEV void vc_barrier_raw() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// pins the instruction order at this point (register-only MFMAs are otherwise free to float across the asm waits / barriers)
VC_DEV void vc_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
template <int P> VC_DEV void vc_setprio() { __builtin_amdgcn_s_setprio(P); }
// "hidden" register loads: issued from inline asm so that hipcc does not see them — beside LDS-DMA traffic hipcc answers the first
// use of ANY compiler-visible load with vmcnt(0), draining the whole DMA ring (probed: tools/probe_waitcnt.hip).  The caller counts
// the wave's VMEM operations itself and, before the first use, issues vc_hwait<N>() naming every destination register
// (cdna_hip_programming.md §5.7 form (ii)): address = uniform base (SGPR pair) + 32-bit per-lane byte offset.
VC_DEV uint32_t vc_hload_b32(const void* sbase, uint32_t voff) {
    uint32_t v; asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory"); return v;
}
VC_DEV uint32_t vc_hload_u16(const void* sbase, uint32_t voff) {
    uint32_t v; asm volatile("global_load_ushort %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory"); return v;
}
// stores in the same addressing form (uniform base in an SGPR pair + one 32-bit lane offset): hipcc materialises a 64-bit VGPR address
// per store for `base + uniform + lane` pointer arithmetic; from asm the per-row base is 3 SALU instructions and no VGPR
VC_DEV void vc_hstore_b32(void* sbase, uint32_t voff, uint32_t data) { asm volatile("global_store_dword %0, %1, %2" :: "v"(voff), "v"(data), "s"(sbase) : "memory"); }
VC_DEV void vc_hstore_b16(void* sbase, uint32_t voff, uint32_t data) { asm volatile("global_store_short %0, %1, %2" :: "v"(voff), "v"(data), "s"(sbase) : "memory"); }
template <int N> VC_DEV void vc_hwait16(uint32_t (&r)[16]) {
    asm volatile("s_waitcnt vmcnt(%16)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                 "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "n"(N) : "memory");
}
VC_DEV void vc_hpin2(uint32_t& a, uint32_t& b) { asm volatile("" : "+v"(a), "+v"(b) :: "memory"); }
template <int N> VC_DEV void vc_hwait2(uint32_t& a, uint32_t& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
// "this register's value is dead from here on": ends the live range of a loop-carried register (array element) at no cost — hipcc
// cannot see that e.g. MFMA fragments are never read again once the next LOAD segment starts, and would keep them allocated
template <typename T> VC_DEV void vc_undef(T& x) { asm volatile("" : "=v"(x)); }
// D(32x32) += A(32x64) * B(64x32), OCP fp8 e4m3 operands with one E8M0 scale per (row, 32-k block) — the gfx950 block-scaled ("MX") MFMA,
// twice the bf16 rate.  Layout probed on the hardware (tools/probe_mx8_layout.hip, profiles/r02_probe_mx8_layout.txt): lane l = (row l & 31,
// half h = l >> 5) holds 32 bytes (8 VGPRs): bytes 0-15 = k 16h..16h+15 (scale block 0), bytes 16-31 = k 32+16h..32+16h+15 (scale block 1);
// the scale of block b of row r is byte `opsel` of the scale dword of lane r + 32 b — i.e. a lane's scale covers 16 of its own k-values
// and 16 of its partner lane's.  D as for bf16.
typedef int vc_i32x8 __attribute__((ext_vector_type(8)));
template <int OPA, int OPB>
VC_DEV vc_f32x16 vc_mfma_mx8_32x32x64(vc_i32x8 a, vc_i32x8 b, vc_f32x16 c, int scale_a, int scale_b) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPA, scale_a, OPB, scale_b);
}
// two fp32 -> two OCP e4m3 bytes (RNE).  The instruction does NOT saturate (probed: |x| > 448 -> NaN), so clamp first.
VC_DEV uint32_t vc_cvt_pk_e4m3(float a, float b) {
    a = fminf(fmaxf(a, -448.0f), 448.0f); b = fminf(fmaxf(b, -448.0f), 448.0f);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xFFFFu;
}
// D(32x32) += A(32x2) * B(2x32), exact f32.  lane l: A[i=l&31][k=l>>5], B[k=l>>5][n=l&31]; D as above.
VC_DEV vc_f32x16 vc_mfma_32x32x2_f32(float a, float b, vc_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

#else
// ------------------------------------------------------------------------------------------ EMU
#include <string.h>
#include <algorithm>
namespace vcemu {
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct Ctx { dim3 tid, bid, bdim, gdim; };
Ctx* cur();                                  // per-fiber context
void sync_block();
void sync_wave();
float shfl_f(float v, int src_lane);         // all 64 lanes of the wave must call
int shfl_i(int v, int src_lane);
void* dyn_shared();
void mfma_32x32x16_bf16(const short* a8, const short* b8, float* c16);   // in-place on c16
void mfma_32x32x2_f32(float a, float b, float* c16);
void ds_read_tr16(const void* p, short* out4);
void dma16(const void* gsrc, void* lds_piece);
void mfma_mx8_32x32x64(const int* a8, const int* b8, float* c16, int sa_byte, int sb_byte);
void launch(void (*trampoline)(void*), void* args, dim3 grid, dim3 block, size_t shmem);
}  // namespace vcemu
using vcemu::dim3;
#define threadIdx (vcemu::cur()->tid)
#define blockIdx (vcemu::cur()->bid)
#define blockDim (vcemu::cur()->bdim)
#define gridDim (vcemu::cur()->gdim)

#define VC_KERNEL static
#define VC_DEV inline
#define VC_HD inline
#define VC_SHARED static thread_local
#define VC_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(vcemu::dyn_shared())
#define __launch_bounds__(...)
#define __restrict__

typedef void* vc_stream_t;

#include <tuple>
#include <utility>
namespace vcemu {
template <typename... KArgs, typename... Args>
void launch_k(void (*k)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
    struct Pack { void (*k)(KArgs...); std::tuple<KArgs...> a; };
    Pack p{k, std::tuple<KArgs...>(static_cast<KArgs>(args)...)};
    launch([](void* q) { Pack* pp = (Pack*)q; std::apply(pp->k, pp->a); }, &p, grid, block, shmem);
}
}  // namespace vcemu
#define VC_LAUNCH(kernel, grid, block, shmem, stream, ...) vcemu::launch_k(kernel, grid, block, shmem, __VA_ARGS__)

static inline int vc_memset_async(void* p, int v, size_t n, vc_stream_t) { memset(p, v, n); return 0; }
static inline int vc_memcpy_d2d_async(void* d, const void* s_, size_t n, vc_stream_t) { memmove(d, s_, n); return 0; }
static inline int vc_upload(void* d, const void* h, size_t n, vc_stream_t) { memcpy(d, h, n); return 0; }
typedef void* vc_event_t;                          // the emulator executes launches synchronously: one stream, no events
static inline int vc_stream_create(vc_stream_t* s) { *s = nullptr; return 0; }
static inline int vc_event_create(vc_event_t* ev) { *ev = nullptr; return 0; }
static inline int vc_event_record(vc_event_t, vc_stream_t) { return 0; }
static inline int vc_stream_wait_event(vc_stream_t, vc_event_t) { return 0; }
static inline void vc_stream_destroy(vc_stream_t) {}
static inline void vc_event_destroy(vc_event_t) {}
static inline bool vc_has_side_streams() { return false; }
static inline bool vc_is_device_ptr(const void*) { return true; }      // (the emulator's "device" is host memory)
static inline int vc_last_launch_error() { return 0; }
static inline unsigned vc_device_bit() { return 1u; }

VC_DEV void vc_sync() { vcemu::sync_block(); }
VC_DEV void vc_wave_barrier() { vcemu::sync_wave(); }
VC_DEV float vc_shfl_xor(float v, int mask) { return vcemu::shfl_f(v, (int)((threadIdx.x & 63) ^ mask)); }
VC_DEV int vc_shfl_xor(int v, int mask) { return vcemu::shfl_i(v, (int)((threadIdx.x & 63) ^ mask)); }
VC_DEV float vc_shfl(float v, int src) { return vcemu::shfl_f(v, src); }
VC_DEV int vc_shfl(int v, int src) { return vcemu::shfl_i(v, src); }

struct vc_f32x16 { float v[16]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
struct vc_f32x4 { float v[4]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
struct vc_s16x8 { short v[8]; short& operator[](int i) { return v[i]; } const short& operator[](int i) const { return v[i]; } };
VC_DEV vc_f32x16 vc_mfma_32x32x16_bf16(vc_s16x8 a, vc_s16x8 b, vc_f32x16 c) { vcemu::mfma_32x32x16_bf16(a.v, b.v, c.v); return c; }
VC_DEV vc_f32x16 vc_mfma_32x32x2_f32(float a, float b, vc_f32x16 c) { vcemu::mfma_32x32x2_f32(a, b, c.v); return c; }
struct vc_s16x4 { short v[4]; short& operator[](int i) { return v[i]; } const short& operator[](int i) const { return v[i]; } };
struct vc_i32x8 { int v[8]; int& operator[](int i) { return v[i]; } const int& operator[](int i) const { return v[i]; } };
template <int OPA, int OPB>
VC_DEV vc_f32x16 vc_mfma_mx8_32x32x64(vc_i32x8 a, vc_i32x8 b, vc_f32x16 c, int scale_a, int scale_b) {
    vcemu::mfma_mx8_32x32x64(a.v, b.v, c.v, (scale_a >> (8 * OPA)) & 0xFF, (scale_b >> (8 * OPB)) & 0xFF); return c;
}
VC_DEV vc_s16x4 vc_ds_read_tr16(const void* p) { vc_s16x4 r; vcemu::ds_read_tr16(p, r.v); return r; }
VC_DEV void vc_dma16(const void* gsrc, void* lds_piece) { vcemu::dma16(gsrc, lds_piece); }
VC_DEV float vc_expf_fast(float x) { return expf(x); }
VC_DEV float vc_mul_rn(float a, float b) { volatile float r = a * b; return r; }
VC_DEV void vc_sched_fence() {}
template <int P> VC_DEV void vc_setprio() {}
VC_DEV uint32_t vc_hload_b32(const void* sbase, uint32_t voff) { uint32_t v; memcpy(&v, (const char*)sbase + voff, 4); return v; }
VC_DEV uint32_t vc_hload_u16(const void* sbase, uint32_t voff) { uint16_t v; memcpy(&v, (const char*)sbase + voff, 2); return v; }
VC_DEV void vc_hstore_b32(void* sbase, uint32_t voff, uint32_t data) { memcpy((char*)sbase + voff, &data, 4); }
VC_DEV void vc_hstore_b16(void* sbase, uint32_t voff, uint32_t data) { uint16_t h = (uint16_t)data; memcpy((char*)sbase + voff, &h, 2); }
template <int N> VC_DEV void vc_hwait16(uint32_t (&)[16]) {}
VC_DEV void vc_hpin2(uint32_t&, uint32_t&) {}
template <int N> VC_DEV void vc_hwait2(uint32_t&, uint32_t&) {}
template <typename T> VC_DEV void vc_undef(T&) {}
VC_DEV int vc_uniform(int x) { return x; }
VC_DEV int vc_wave_ticket(int* ctr) { int t = 0; if ((threadIdx.x & 63) == 0) t = __atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED); return vcemu::shfl_i(t, 0); }
VC_DEV uint64_t vc_uniform64(uint64_t x) { return x; }
template <int N> VC_DEV void vc_wait_vmcnt() {}
VC_DEV void vc_barrier_raw() { vcemu::sync_block(); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
#endif

// ------------------------------------------------------------------------------------------ common
struct vc_bf16 { uint16_t bits; };          // one 16-bit storage element (bf16, or IEEE half under VC_H16)
VC_HD float vc_bits_f32(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
VC_HD uint32_t vc_f32_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

#ifdef VC_H16
// software codec (host side and the emulator build; the device uses v_cvt_f32_f16 / v_cvt_pk_f16_f32)
VC_HD float vc_half_bits_to_f32(uint32_t h) {
    const uint32_t sg = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    if (e == 0) return vc_bits_f32(vc_f32_bits((float)m * 5.9604644775390625e-8f) | sg);      // zero / subnormal: m * 2^-24
    if (e == 31) return vc_bits_f32(sg | 0x7F800000u | (m << 13));
    return vc_bits_f32(sg | ((e + 112u) << 23) | (m << 13));
}
VC_HD uint32_t vc_f32_to_half_bits(float f) {       // round-to-nearest-even; overflow -> inf; NaN stays NaN
    uint32_t u = vc_f32_bits(f);
    const uint32_t sg = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u > 0x7F800000u) return sg | 0x7E00u;
    if (u >= 0x477FF000u) return sg | 0x7C00u;
    if (u < 0x38800000u) {                                           // below 2^-14: subnormal half = RNE(|f| * 2^24)
        const float r = vc_bits_f32(u) * 16777216.0f + 12582912.0f;  // (1.5 * 2^23: the integer lands in the low mantissa bits)
        return sg | (vc_f32_bits(r) - 0x4B400000u);
    }
    const uint32_t r = u + 0xFFFu + ((u >> 13) & 1u);
    return sg | ((r - 0x38000000u) >> 13);
}
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 vc_h2_hw __attribute__((ext_vector_type(2)));
typedef float vc_f2_hw __attribute__((ext_vector_type(2)));
#endif
VC_HD float vc_bf16_to_f32(vc_bf16 h) {
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, h.bits);
#else
    return vc_half_bits_to_f32(h.bits);
#endif
}
VC_HD vc_bf16 vc_f32_to_bf16(float f) {
    vc_bf16 r;
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    r.bits = __builtin_bit_cast(uint16_t, (_Float16)f);
#else
    r.bits = (uint16_t)vc_f32_to_half_bits(f);
#endif
    return r;
}
// the two elements of a packed pair (element 0 in the low half) / packing two
VC_HD float vc_lo16_f32(uint32_t u) {
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(vc_h2_hw, u).x;
#else
    return vc_half_bits_to_f32(u & 0xFFFFu);
#endif
}
VC_HD float vc_hi16_f32(uint32_t u) {
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(vc_h2_hw, u).y;
#else
    return vc_half_bits_to_f32(u >> 16);
#endif
}
VC_HD uint32_t vc_pack_bf16x2(float lo, float hi) {
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    const vc_f2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vc_h2_hw));               // v_cvt_pk_f16_f32
#else
    return vc_f32_to_half_bits(lo) | (vc_f32_to_half_bits(hi) << 16);
#endif
}
#else
VC_HD float vc_bf16_to_f32(vc_bf16 h) {
    uint32_t u = ((uint32_t)h.bits) << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
VC_HD vc_bf16 vc_f32_to_bf16(float f) {     // round-to-nearest-even; NaN stays NaN
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    { __bf16 h = (__bf16)f; vc_bf16 r; __builtin_memcpy(&r, &h, 2); return r; }   // v_cvt_pk_bf16_f32 on gfx950
#endif
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    vc_bf16 r;
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) { r.bits = (uint16_t)((u >> 16) | 0x40); return r; }
    u += 0x7FFFu + ((u >> 16) & 1u);
    r.bits = (uint16_t)(u >> 16);
    return r;
}
VC_HD float vc_lo16_f32(uint32_t u) { return vc_bits_f32(u << 16); }
VC_HD float vc_hi16_f32(uint32_t u) { return vc_bits_f32(u & 0xFFFF0000u); }
VC_HD uint32_t vc_pack_bf16x2(float lo, float hi) {
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
    // one v_cvt_pk_bf16_f32 on exactly this pair (two scalar conversions leave the pairing to the vectoriser: it packed (v0, v2) / (v1, v3) of the
    // persistent GEMM's store quads and spent four more VALU per store re-sorting the halves)
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 h2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
#else
    return (uint32_t)vc_f32_to_bf16(lo).bits | ((uint32_t)vc_f32_to_bf16(hi).bits << 16);
#endif
}
#endif

// c + a.lo * b.lo + a.hi * b.hi on two packed pairs of the 16-bit storage format, fp32 accumulate: one v_dot2c_f32_{bf16,f16} on gfx950 (attn_cls.h: the
// 512-long dot products of the class-token attention are VALU work — too small for the matrix cores' tile shapes to pay)
VC_DEV float vc_dot2(uint32_t a, uint32_t b, float c) {
#if !defined(VC_EMU) && defined(__HIP_DEVICE_COMPILE__)
#ifdef VC_H16
    typedef _Float16 vc_d2_hw __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(vc_d2_hw, a), __builtin_bit_cast(vc_d2_hw, b), c, false);
#else
    typedef __bf16 vc_d2_hw __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vc_d2_hw, a), __builtin_bit_cast(vc_d2_hw, b), c, false);
#endif
#else
    return c + vc_lo16_f32(a) * vc_lo16_f32(b) + vc_hi16_f32(a) * vc_hi16_f32(b);
#endif
}

template <typename T> struct vc_cvt;
template <> struct vc_cvt<float> {
    VC_HD static float to_f32(float v) { return v; }
    VC_HD static float from_f32(float v) { return v; }
};
template <> struct vc_cvt<vc_bf16> {
    VC_HD static float to_f32(vc_bf16 v) { return vc_bf16_to_f32(v); }
    VC_HD static vc_bf16 from_f32(float v) { return vc_f32_to_bf16(v); }
};
// pre-split bf16x3 operand word (gemm.h): hi bf16 in the upper half, lo = bf16(x - hi) in the lower
struct vc_pk { uint32_t w; };
VC_HD uint32_t vc_pk_pack(float x) {
    const vc_bf16 h = vc_f32_to_bf16(x);
    const vc_bf16 l = vc_f32_to_bf16(x - vc_bf16_to_f32(h));
    return ((uint32_t)h.bits << 16) | (uint32_t)l.bits;
}
// (as a storage type: reads give hi + lo — the 16-bit-mantissa value the bf16x3 products see — writes split)
template <> struct vc_cvt<vc_pk> {
    VC_HD static float to_f32(vc_pk v) { return vc_bits_f32(v.w & 0xFFFF0000u) + vc_bits_f32(v.w << 16); }
    VC_HD static vc_pk from_f32(float v) { vc_pk r; r.w = vc_pk_pack(v); return r; }
};
template <typename T> VC_HD float vc_ld(const T* p) { return vc_cvt<T>::to_f32(*p); }
template <typename T> VC_HD void vc_st(T* p, float v) { *p = vc_cvt<T>::from_f32(v); }

VC_DEV float vc_wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += vc_shfl_xor(v, m);
    return v;
}
VC_DEV float vc_wave_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, vc_shfl_xor(v, m));
    return v;
}

struct alignas(16) vc_u32x4 { uint32_t x, y, z, w; };   // 16-byte POD for vector copies
struct alignas(8) vc_u32x2 { uint32_t x, y; };

// ---- OCP fp8 e4m3 (no infinities, NaN = 0x7f / 0xff, max 448) — software codec for the emulator build and the host
VC_HD float vc_e4m3_to_f32(uint8_t v) {
    const int e = (v >> 3) & 15, m = v & 7;
    float f = e ? ldexpf(1.0f + m * 0.125f, e - 7) : ldexpf(m * 0.125f, -6);
    if (e == 15 && m == 7) f = NAN;
    return (v & 0x80) ? -f : f;
}
VC_HD uint8_t vc_f32_to_e4m3_sw(float x) {           // round-to-nearest-even, saturating at +-448
    const uint8_t s = x < 0.0f ? 0x80 : 0x00;
    float a = fabsf(x);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0f) return s | 0x7e;
    if (a < 0.0009765625f) return s;                                     // < half of the smallest subnormal (2^-9): rounds to zero (tie -> even = 0)
    int e; const float fr = frexpf(a, &e);                               // a = fr * 2^e, fr in [0.5, 1)
    int ex = e - 1; if (ex < -6) ex = -6;                                // exponent of the representable grid (subnormals share 2^-6)
    const float q = ldexpf(a, 3 - ex);                                   // a in units of 2^(ex-3): [8, 16) for normals, [0, 8) for subnormals
    float r = rintf(q);                                                  // RNE (default rounding mode)
    int ri = (int)r;
    if (ri == 16) { ri = 8; ++ex; }
    (void)fr;
    if (ex + 7 > 15 || (ex + 7 == 15 && ri - 8 > 6)) return s | 0x7e;
    if (ri < 8) return s | (uint8_t)ri;                                  // subnormal (exponent field 0)
    return s | (uint8_t)(((ex + 7) << 3) | (ri - 8));
}
#ifdef VC_EMU
VC_DEV uint32_t vc_cvt_pk_e4m3(float a, float b) { return (uint32_t)vc_f32_to_e4m3_sw(a) | ((uint32_t)vc_f32_to_e4m3_sw(b) << 8); }
#endif

// ---- dropout: counter-based, stateless.  One 32-bit hash serves TWO consecutive elements (12-bit draws from bits 8..19 and
// 20..31): keep-multiplier of element idx at a site = (draw(hash(key, idx >> 1), idx & 1) >= thr) ? scale : 0, so the backward
// regenerates exactly the forward's mask from (key, idx) — no mask tensors in HBM — and aligned runs of elements pay half a
// hash each (the two integer multiplies of the mixer are quarter-rate VALU on CDNA).
// key = vc_drop_key(step seed, site id) (0 = disabled), thr = round(p * 4096), scale = 4096 / (4096 - thr) (unbiased for the
// effective p = thr / 4096; p = 0.1 -> 0.10010).
struct vc_drop { uint32_t key, thr; float scale; };
VC_HD uint32_t vc_hash32(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
VC_HD uint32_t vc_drop_key(uint64_t seed, uint32_t site) {
    uint32_t k = vc_hash32((uint32_t)seed ^ vc_hash32((uint32_t)(seed >> 32) + 0x9E3779B9u * (site + 1u)));
    return k | 1u;
}
VC_HD vc_drop vc_drop_make(uint32_t key, float p) {
    vc_drop d; d.key = key; d.thr = (uint32_t)(p * 4096.0f + 0.5f); d.scale = 4096.0f / (float)(4096u - d.thr);
    return d;
}
VC_HD uint32_t vc_drop_hash(const vc_drop& d, uint32_t pair) { return vc_hash32((pair * 0x9E3779B1u) ^ d.key); }          // pair = idx >> 1
VC_HD bool vc_drop_keep_lo(const vc_drop& d, uint32_t h) { return ((h >> 8) & 0xFFFu) >= d.thr; }                           // even idx
VC_HD bool vc_drop_keep_hi(const vc_drop& d, uint32_t h) { return (h >> 20) >= d.thr; }                                     // odd idx
// element indices are 32-bit: every site has < 2^32 elements (checked on the host when the workspace is planned)
VC_HD bool vc_drop_keep(const vc_drop& d, uint32_t idx) {
    const uint32_t h = vc_drop_hash(d, idx >> 1);
    return (idx & 1u) ? vc_drop_keep_hi(d, h) : vc_drop_keep_lo(d, h);
}
VC_HD float vc_drop_mul(const vc_drop& d, uint32_t idx) { return vc_drop_keep(d, idx) ? d.scale : 0.0f; }
// four consecutive elements starting at an EVEN index: two hashes
VC_HD void vc_drop_mul4(const vc_drop& d, uint32_t idx, float (&m)[4]) {
    const uint32_t h0 = vc_drop_hash(d, idx >> 1), h1 = vc_drop_hash(d, (idx >> 1) + 1u);
    m[0] = vc_drop_keep_lo(d, h0) ? d.scale : 0.0f; m[1] = vc_drop_keep_hi(d, h0) ? d.scale : 0.0f;
    m[2] = vc_drop_keep_lo(d, h1) ? d.scale : 0.0f; m[3] = vc_drop_keep_hi(d, h1) ? d.scale : 0.0f;
}

#define VC_CEIL_DIV(a, b) (((a) + (b) - 1) / (b))
#define VC_INLINE_LAMBDA __attribute__((always_inline))
