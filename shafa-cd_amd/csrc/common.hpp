// common.hpp — shared device/host helpers for libshafa_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/shafa_hip.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define WAVE 64

// ---------------------------------------------------------------------------------------------
// Tile descriptors for single-pass chained scans ("decoupled look-back").
// One 64-bit word per tile: status in the top 2 bits, value in the low 62.  The word is written and
// read with relaxed agent-scope atomics: the payload IS the flag, so no fence is needed and the
// protocol does not depend on dispatch order or XCD placement (MI355X_MICROARCH §visibility, R2).
// Every descriptor word is zeroed by a hipMemsetAsync node ahead of the launch.
// ---------------------------------------------------------------------------------------------
#define DESC_EMPTY 0ull
#define DESC_AGG 1ull     // value = this tile's own aggregate
#define DESC_PREFIX 2ull  // value = inclusive prefix up to and including this tile
#define DESC_VALUE_MASK ((1ull << 62) - 1)

__device__ __forceinline__ void desc_store(u64 *p, u64 status, u64 value)
{
    __hip_atomic_store(p, (status << 62) | (value & DESC_VALUE_MASK), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 desc_load(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bound for every inter-workgroup wait: a lost predecessor sets the block's error instead of hanging the GPU.  The bound
// is WALL-CLOCK time (s_memrealtime, the constant 100 MHz counter), not a number of polls: a predecessor that is merely
// late — two processes sharing the GPU, a profiler, a pre-empted workgroup — is waited for; only several seconds without
// progress count as lost.  (A poll count of 2^22 was about 0.15 s and turned contention into SHAFA_DEVICE_ERROR.)
#define SPIN_SECONDS 4ull
struct SpinClock {
    u32 polls;
    u64 t0;
    __device__ __forceinline__ SpinClock() : polls(0), t0(0) {}
    __device__ __forceinline__ bool expired()           // call once per unsuccessful poll
    {
        if ((++polls & 1023u) != 0) return false;       // look at the clock every 1024th poll
        const u64 now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) { t0 = now | 1ull; return false; }
        return now - t0 > SPIN_SECONDS * 100000000ull;
    }
};

// LDS byte address of a __shared__ object.  Through the LDS address space, not the flat one: the cast of a flat
// pointer carries a null check (s_cselect) that keeps the segment's base out of the ds instructions' offset field.
__device__ __forceinline__ u32 lds_addr(const void *p)
{
    return (u32)(size_t)(const __attribute__((address_space(3))) u8 *)p;
}

// ---------------------------------------------------------------------------------------------
// wave / workgroup primitives (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS only.  __syncthreads() is also a workgroup-scope fence for GLOBAL memory: the
// compiler puts s_waitcnt vmcnt(0) in front of it, so every wave first drains its loads and stores — exactly the
// prefetches and streaming stores that are meant to stay in flight across the barrier.  Use this one where the waves
// only hand each other LDS contents (global data published to other workgroups goes through the descriptor atomics).
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// workgroup-wide OR of a predicate through LDS flags (256 threads).  hip's __syncthreads_or puts 256 bytes of static
// LDS in front of the dynamic segment, whose base then costs an add in every look-up that could have used the ds
// offset field.  flags: 2 x 4 words; the sets alternate between calls (`turn`), so one barrier a call is enough.
__device__ __forceinline__ bool wg_any(bool p, u32 *flags, u32 &turn)
{
    u32 *f = flags + 4u * (turn & 1u);
    ++turn;
    const bool w = __any(p) != 0;
    if ((threadIdx.x & 63u) == 0) f[threadIdx.x >> 6] = w;
    lds_barrier();
    return (f[0] | f[1] | f[2] | f[3]) != 0;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// inclusive add-scan over the 64 lanes of a wave with DPP: four row_shr steps inside the 16-lane rows, then the row
// totals are broadcast into the following rows (row_bcast:15 -> rows 1,3; row_bcast:31 -> rows 2,3)
__device__ __forceinline__ u32 dpp_scan_add(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan_add(T v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// 32-bit sums: six DPP adds instead of six LDS-routed shuffles
template <>
__device__ __forceinline__ u32 wave_incl_scan_add<u32>(u32 v) { return dpp_scan_add(v); }

template <typename T>
__device__ __forceinline__ T wave_reduce_add(T v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ u32 bswap32(u32 x) { return __builtin_bswap32(x); }
// SWAR zero-byte tests.  zero_bytes80: 0x80 in every byte of d that is zero (exact).  The flags are gathered into
// mask bits by v_dot4_u32_u8 (flag bytes times the weights 1, 2, 4, 8 — or 16 .. 128 for the second word of a pair,
// on top of the first word's sum): 128 * mask, no shift-and-or ladder.
__device__ __forceinline__ u32 zero_bytes80(u32 d)
{
    const u32 s = (d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(s | d) & 0x80808080u;
}
// bit b = byte b of d is zero
__device__ __forceinline__ u32 zmask4(u32 d) { return __builtin_amdgcn_udot4(zero_bytes80(d), 0x08040201u, 0u, false) >> 7; }
// bit 4 i + b = byte b of d[i] is zero
__device__ __forceinline__ u32 zmask32(const u32 *d)       // eight words, fully unrolled: d stays in registers
{
    u32 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        acc[k] = __builtin_amdgcn_udot4(zero_bytes80(d[2 * k + 1]), 0x80402010u,
                                        __builtin_amdgcn_udot4(zero_bytes80(d[2 * k]), 0x08040201u, 0u, false), false);
    return (acc[0] >> 7) | (acc[1] << 1) | (acc[2] << 9) | (acc[3] << 17);
}
// a + byte C of x in one instruction (SDWA source select); the compiler only finds this form for C = 3
template <int C>
__device__ __forceinline__ u32 add_byte(u32 a, u32 x)
{
    u32 r;
    if (C == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(x));
    if (C == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(x));
    if (C == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(x));
    if (C == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(x));
    return r;
}
// every byte of x bit-reversed, bytes in place: an MSB-first bit stream loaded little-endian becomes LSB first
__device__ __forceinline__ u32 rev_bytes(u32 x) { return __builtin_bitreverse32(__builtin_bswap32(x)); }

// (hi:lo) >> sh, low 32 bits; sh in [0,31]
__device__ __forceinline__ u32 funnel_r(u32 hi, u32 lo, u32 sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// Pointers that reach a kernel inside a parameter record are "generic" to the compiler, which then
// emits flat_load/flat_store: those count on BOTH vmcnt and lgkmcnt and retire out of order, so every
// later LDS wait also waits for the HBM access (no prefetch overlap).  These helpers state the global
// address space explicitly (global_load/global_store, vmcnt only).
#define GLOBAL_AS __attribute__((address_space(1)))
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T gload(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const GLOBAL_AS T *)(unsigned long long)p;
#else
    return *(const T *)p;
#endif
}
template <>
__device__ __forceinline__ uint4 gload<uint4>(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const u32x4_t v = *(const GLOBAL_AS u32x4_t *)(unsigned long long)p;
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *(const uint4 *)p;
#endif
}
template <typename T>
__device__ __forceinline__ void gstore(void *p, T v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *(GLOBAL_AS T *)(unsigned long long)p = v;
#else
    *(T *)p = v;
#endif
}
template <>
__device__ __forceinline__ void gstore<uint4>(void *p, uint4 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    u32x4_t x;
    x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    *(GLOBAL_AS u32x4_t *)(unsigned long long)p = x;
#else
    *(uint4 *)p = v;
#endif
}

// streaming (non-temporal) forms for data that is touched once: the input symbols and the encoded output
template <typename T>
__device__ __forceinline__ T gload_nt(const void *p);
template <>
__device__ __forceinline__ uint4 gload_nt<uint4>(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const u32x4_t v = __builtin_nontemporal_load((const GLOBAL_AS u32x4_t *)(unsigned long long)p);
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *(const uint4 *)p;
#endif
}
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <>
__device__ __forceinline__ uint2 gload_nt<uint2>(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const u32x2_t v = __builtin_nontemporal_load((const GLOBAL_AS u32x2_t *)(unsigned long long)p);
    return make_uint2(v.x, v.y);
#else
    return *(const uint2 *)p;
#endif
}
template <>
__device__ __forceinline__ uint2 gload<uint2>(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const u32x2_t v = *(const GLOBAL_AS u32x2_t *)(unsigned long long)p;
    return make_uint2(v.x, v.y);
#else
    return *(const uint2 *)p;
#endif
}
template <typename T>
__device__ __forceinline__ void gstore_nt(void *p, T v);
template <>
__device__ __forceinline__ void gstore_nt<uint4>(void *p, uint4 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    u32x4_t x;
    x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    __builtin_nontemporal_store(x, (GLOBAL_AS u32x4_t *)(unsigned long long)p);
#else
    *(uint4 *)p = v;
#endif
}

// uniform base pointer + 32-bit lane offset: lets the compiler use the SGPR-base addressing form (no 64-bit VGPR address)
template <typename T>
__device__ __forceinline__ T gload_off(const void *base, u32 off) { return gload<T>((const u8 *)base + off); }
template <typename T>
__device__ __forceinline__ T gload_nt_off(const void *base, u32 off) { return gload_nt<T>((const u8 *)base + off); }
template <typename T>
__device__ __forceinline__ void gstore_off(void *base, u32 off, T v) { gstore<T>((u8 *)base + off, v); }
template <typename T>
__device__ __forceinline__ void gstore_nt_off(void *base, u32 off, T v) { gstore_nt<T>((u8 *)base + off, v); }

// first error wins per block (codes are small positive ints; keep the first non-zero)
__device__ __forceinline__ void set_error(int *err, int code)
{
    atomicCAS(err, 0, code);
}
// A block with two faults of different kinds: the sequential coder (the oracle) reports the one it meets first, and which that
// is does not depend on timing.  `code` replaces "no error" and the error `weaker`, whichever workgroup comes first.
__device__ __forceinline__ void set_error_over(int *err, int code, int weaker)
{
    const int old = atomicCAS(err, 0, code);
    if (old == weaker) atomicCAS(err, weaker, code);
}

// ---------------------------------------------------------------------------------------------
// look-back over one block's tile descriptors; wave 0 only, all 64 lanes.  Returns the exclusive
// prefix (sum of the aggregates of tiles 0..k-1).
// ---------------------------------------------------------------------------------------------
// `first` (optional): the descriptors of tiles k-1-lane, loaded earlier by the caller so that the
// round trip overlaps other work; used for the first window instead of a fresh load.
template <int STRIDE = 1, int SLEEP = 1>
__device__ __forceinline__ u64 lookback_sum(const u64 *desc, int k, int *err, bool have_first = false,
                                            u64 first = 0)
{
    const int lane = lane_id();
    u64 excl = 0;
    int j = k - 1;
    for (;;) {
        const int idx = j - lane;
        u64 d = 0;
        SpinClock spin;
        for (;;) {
            if (have_first) d = first;
            else d = (idx >= 0) ? desc_load(desc + (size_t)idx * STRIDE) : (DESC_PREFIX << 62);
            have_first = false;
            {   // entries behind the nearest inclusive prefix are not needed: do not wait for them
                const u64 pm = __ballot((d >> 62) == DESC_PREFIX), em = __ballot((d >> 62) == DESC_EMPTY);
                const u64 need = pm ? ((pm & (0 - pm)) - 1) : ~0ull;
                if (!(em & need)) break;
            }
            if (spin.expired()) {                // lost predecessor: flag instead of hanging
                if (lane == 0) set_error(err, SHAFA_DEVICE_ERROR);
                if ((d >> 62) == DESC_EMPTY) d = (DESC_PREFIX << 62);
                break;
            }
            __builtin_amdgcn_s_sleep(SLEEP);
        }
        const u64 val = d & DESC_VALUE_MASK;
        const u64 pmask = __ballot((d >> 62) == DESC_PREFIX);
        if (pmask) {
            const int pl = __ffsll((unsigned long long)pmask) - 1;
            excl += wave_reduce_add<u64>(lane <= pl ? val : 0ull);
            break;
        }
        excl += wave_reduce_add<u64>(val);
        j -= 64;
    }
    return excl;
}

// Same look-back for aggregates below 2^32 per 64-tile window (tile bit totals), without the LDS-routed shuffles of
// wave_reduce_add: the window's aggregates are summed with a DPP scan of their low dwords, the one inclusive prefix that
// ends the walk is read with v_readlane.  `first` = descriptors of tiles max(k-1-lane, 0), loaded by the caller well ahead
// (unconditionally: a select on a freshly loaded register would be a wait for the load).
__device__ __forceinline__ u64 lookback_sum_dpp(const u64 *desc, int k, int *err, u64 first)
{
    const int lane = lane_id();
    u64 excl = 0;
    int j = k - 1;
    bool have_first = true;
    for (;;) {
        const int idx = j - lane;
        u64 d = 0;
        SpinClock spin;
        for (;;) {
            if (have_first) d = (idx >= 0) ? first : (DESC_PREFIX << 62);   // the caller loaded desc[max(idx, 0)]: no select there
            else d = (idx >= 0) ? desc_load(desc + idx) : (DESC_PREFIX << 62);
            have_first = false;
            const u64 pm = __ballot((d >> 62) == DESC_PREFIX);
            const u64 em = __ballot((d >> 62) == DESC_EMPTY);
            // entries behind the nearest inclusive prefix are not needed
            const u64 need = pm ? ((pm & (0 - pm)) - 1) : ~0ull;       // lanes in front of the first prefix lane
            if (!(em & need)) break;
            if (spin.expired()) {                // lost predecessor: flag instead of hanging
                if (lane == 0) set_error(err, SHAFA_DEVICE_ERROR);
                if ((d >> 62) == DESC_EMPTY) d = (DESC_PREFIX << 62);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const u64 pmask = __ballot((d >> 62) == DESC_PREFIX);
        const int pl = pmask ? __ffsll((unsigned long long)pmask) - 1 : 64;
        const u32 agg = lane < pl ? (u32)d : 0u;                       // aggregates in front of the prefix lane
        excl += (u32)__builtin_amdgcn_readlane((int)dpp_scan_add(agg), 63);
        if (pmask) {
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)d, pl);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(d >> 32), pl);
            excl += (((u64)hi << 32) | lo) & DESC_VALUE_MASK;
            break;
        }
        j -= 64;
    }
    return excl;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return shafa_set_hip_error(_e, #expr);      \
    } while (0)

int shafa_set_hip_error(hipError_t e, const char *what);

static inline u64 ceil_div_u64(u64 a, u64 b) { return (a + b - 1) / b; }
