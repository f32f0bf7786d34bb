// mgx_device.h -- device-side helpers shared by the kernel translation units of libmgx.so (k_*.hip).
// Everything here is __device__ __forceinline__ inside an unnamed namespace: each TU gets its own copy.
#ifndef MGX_DEVICE_H
#define MGX_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ void wave_sync()
{
    // all LDS traffic of a wave is issued in program order and returns in order, so lanes of ONE wave may hand
    // data to each other through LDS without s_barrier; this only pins the compiler's ordering.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The observation stream is written once and never read back by this library: non-temporal stores keep it from
// displacing the env state (re-read every step) in L2 / Infinity Cache.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store16(uint4 *p, const uint4 &v)
{
    u32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(p));
}

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void nt_store12(uint32_t *p, uint32_t a, uint32_t b, uint32_t c)
{
    u32x3 x = {a, b, c};
    __builtin_nontemporal_store(x, reinterpret_cast<u32x3 *>(p));
}

__device__ __forceinline__ bool lane_bit(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// cell code -> (type | color<<8 | state<<16), the reference's WorldObj.encode()/Door.encode() (minigrid.py:113-115,264-275)
__device__ __forceinline__ uint32_t decode_triple(uint32_t c)
{
    const uint32_t k = c & 15u, col = (c >> 4) & 7u;
    const bool shut = k > MGX_K_AGENT; // 11 closed, 12 locked
    const uint32_t type = shut ? 4u : k;
    const uint32_t st = shut ? k - 10u : 0u;
    return type | (col << 8) | (st << 16);
}

// full-obs variant: kind 10 is the agent marker (10, 0, dir) with dir kept in the colour bits
__device__ __forceinline__ uint32_t decode_triple_full(uint32_t c)
{
    const uint32_t k = c & 15u, col = (c >> 4) & 7u;
    if (k == MGX_K_AGENT) return 10u | (col << 16);
    return decode_triple(c);
}

// Wall, or Door that is not open: see_behind() False (minigrid.py:105,233,249)
__device__ __forceinline__ bool is_opaque(uint32_t c)
{
    const uint32_t k = c & 15u;
    return k == MGX_K_WALL || k > MGX_K_AGENT;
}

// ------------------------------------------------------------------------------------------------
// global -> LDS: the tile's 64*S contiguous bytes, re-strided to LS bytes per env (LS/4 odd).
template <int CS>
__device__ __forceinline__ void stage_tile(const uint8_t *__restrict__ cells, int64_t env0, int S_rt, int LS,
                                           uint8_t *lds, int lane)
{
    const int S = CS ? CS : S_rt;
    const int SD = S >> 2, LSD = LS >> 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(cells + env0 * S);
    uint32_t *l32 = reinterpret_cast<uint32_t *>(lds);
    const int n_chunks = 4 * S; // 64*S/16
    if constexpr (CS != 0 && ((CS >> 2) & 1)) {
        // S/4 already odd (5x5, 6x6, 7x7, 9x9, 11x11): the LDS image has the layout of the HBM run, a straight 16-B copy
        uint4 *l128 = reinterpret_cast<uint4 *>(lds);
#pragma unroll 4
        for (int c = lane; c < n_chunks; c += 64) l128[c] = src[c];
        return;
    }
#pragma unroll 4
    for (int c = lane; c < n_chunks; c += 64) {
        const uint4 v = src[c]; // default cache policy on purpose: the state is re-read every step and non-temporal
                                // loads cost 9 % at 1 Mi envs (it lives in L2 / Infinity Cache between steps)
        const int d = c * 4;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dd = d + j;
            const int e = dd / SD;
            l32[e * LSD + (dd - e * SD)] = w[j];
        }
    }
}

// LDS -> global: the inverse (k_rollout once per T steps; k_dynobs every step: the moved obstacles go back as whole tiles)
template <int CS>
__device__ __forceinline__ void unstage_tile(uint8_t *__restrict__ cells, int64_t env0, int S_rt, int LS, const uint8_t *lds, int lane)
{ // inverse of stage_tile: the LDS image (LS bytes per env) back to the tile's 64*S contiguous bytes
    const int S = CS ? CS : S_rt;
    const int SD = S >> 2;
    const int LSD = LS >> 2;
    uint4 *dst = reinterpret_cast<uint4 *>(cells + env0 * S);
    const uint32_t *l32 = reinterpret_cast<const uint32_t *>(lds);
    const int n_chunks = 4 * S;
    if constexpr (CS != 0 && ((CS >> 2) & 1) != 0) {
        const uint4 *l128 = reinterpret_cast<const uint4 *>(lds);
#pragma unroll 4
        for (int c = lane; c < n_chunks; c += 64) dst[c] = l128[c];
        return;
    }
#pragma unroll 4
    for (int c = lane; c < n_chunks; c += 64) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int dd = 4 * c + j, e = dd / SD;
            w[j] = l32[e * LSD + (dd - e * SD)];
        }
        dst[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ------------------------------------------------------------------------------------------------
struct Lane {
    int ax, ay, dir;
    uint32_t carry; // cell code, MGX_CODE_EMPTY = nothing
    int steps;
    uint32_t task;  // per-env task word (16 bits), only with a task rule
    uint32_t dirty; // bit 23 of record word 0: a step has changed a cell since the episode started (cells != cells0).  An
                    // in-kernel auto-reset copies the snapshot back only then: in Empty / Crossing / LavaGap / FourRooms
                    // nothing ever changes, so their resets (40 % of LavaCrossing's waves see one per step) move no cells.
};
#define MGX_REC_DIRTY (1u << 23)
#define MGX_REC_ABSORBED (1u << 22) /* DACWrapper handles (mgx_set_dac): the env is done and waits for the wrapper's time-out; rides in Lane.dirty
                                       through pack_rec (the run-time-size step kernels only: the others never see the bit set) */

// record word 1 = step_count, or step_count | task << 16 for handles with a task rule (max_steps <= 65535 there)
__device__ __forceinline__ Lane unpack_rec(uint2 r, int has_task = 0)
{
    Lane L;
    L.ax = r.x & 255u; L.ay = (r.x >> 8) & 255u; L.dir = (r.x >> 16) & 3u; L.carry = r.x >> 24;
    L.dirty = r.x & MGX_REC_DIRTY;
    L.steps = has_task ? (int)(r.y & 0xFFFFu) : (int)r.y;
    L.task = has_task ? r.y >> 16 : 0u;
    return L;
}
__device__ __forceinline__ uint2 pack_rec(const Lane &L, int has_task = 0)
{
    return make_uint2((uint32_t)L.ax | ((uint32_t)L.ay << 8) | ((uint32_t)L.dir << 16) | L.dirty | (L.carry << 24),
                      has_task ? ((uint32_t)L.steps & 0xFFFFu) | (L.task << 16) : (uint32_t)L.steps);
}

} // namespace

#endif
