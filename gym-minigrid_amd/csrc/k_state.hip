// k_state.hip -- state I/O and small utility kernels: reference encoding <-> cell codes (k_pack_state, k_unpack_state),
// hidden object state, task words, direction, the synthetic action stream, the statistics reduction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include "mgx_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// reference encoding <-> internal codes
__global__ __launch_bounds__(256) void k_pack_state(const PackParams p)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cells = p.W * p.H;
    const int64_t total = p.n * (int64_t)p.S;
    bool bad = false;
    if (t < total) {
        const int64_t e = t / p.S;
        const int c = (int)(t - e * p.S);
        if (!p.mask || p.mask[e]) {
            uint32_t code = 0;
            if (c < cells) {
                const uint8_t *tr = p.grid + ((p.bcast ? 0 : e) * cells + c) * 3;
                const uint32_t ty = tr[0], co = tr[1], st = tr[2];
                const uint32_t ax = p.aux ? p.aux[e * cells + c] : 0u;
                uint32_t k = ty;
                if (ty < 1 || ty > 9 || co > 6) bad = true;
                // aux: bit0 Goal.overlap | (triage_color+1) << 1 | ((toggletimes-1)&15) << 4.  Goals and boxes only; without
                // object-state planes just the two goals the kernels know by themselves: default (0) and terminal (0xF1).
                if (ax != 0 && ty != 8 && ty != 7) bad = true;
                if ((ax & 1u) && ty != 8) bad = true;
                if (!p.objaux && ax != 0 && !(ty == 8 && ax == 0xF1u)) bad = true;
                if (ty == 4) { if (st > 2) bad = true; k = st == 0 ? MGX_K_DOOR_OPEN : (st == 1 ? MGX_K_DOOR_CLOSED : MGX_K_DOOR_LOCKED); }
                else if (st != 0) bad = true;
                if (ty == 1 && (co != 0 || ax != 0)) bad = true; // None encodes as exactly (1,0,0)
                code = (k & 15u) | ((co & 7u) << 4) | ((ax & 1u) << 7);
                if (p.objaux) { p.objaux[t] = (uint8_t)(ax & 0xFEu); p.objcont[t] = MGX_CODE_EMPTY; }
                if (p.objaux && p.cells0) { p.objaux0[t] = (uint8_t)(ax & 0xFEu); p.objcont0[t] = MGX_CODE_EMPTY; }
            } else if (p.objaux) {
                p.objaux[t] = 0; p.objcont[t] = MGX_CODE_EMPTY;
                if (p.cells0) { p.objaux0[t] = 0; p.objcont0[t] = MGX_CODE_EMPTY; }
            }
            p.cells[t] = (uint8_t)code;
            if (p.cells0) p.cells0[t] = (uint8_t)code; // (null: the snapshot arrays hold the NEXT level of a new_level_each_episode handle)
        }
    }
    if (t < p.n && (!p.mask || p.mask[t])) {
        const int64_t ts = p.bcast ? 0 : t;
        const int32_t x = p.agent[ts * 3], y = p.agent[ts * 3 + 1], d = p.agent[ts * 3 + 2];
        if (x < 0 || x >= p.W || y < 0 || y >= p.H || d < 0 || d > 3) bad = true;
        uint32_t cc = MGX_CODE_EMPTY;
        if (p.carry) {
            const uint32_t ty = p.carry[t * 3], co = p.carry[t * 3 + 1], st = p.carry[t * 3 + 2];
            if (ty == 1) { if (co || st) bad = true; }
            else if ((ty != 5 && ty != 6 && ty != 7) || co > 6 || st != 0) bad = true; // only can_pickup() objects
            cc = (ty & 15u) | ((co & 7u) << 4);
        }
        const int32_t sc = p.steps ? p.steps[t] : 0;
        if (sc < 0) bad = true;
        uint32_t w1 = (uint32_t)sc;
        if (p.has_task) { // the task word of the env survives a state injection
            if (sc > 0xFFFF) bad = true;
            w1 = ((uint32_t)sc & 0xFFFFu) | (p.rec[t].y & 0xFFFF0000u);
        }
        const uint2 rec = make_uint2((uint32_t)(x & 255) | ((uint32_t)(y & 255) << 8) | ((uint32_t)(d & 3) << 16) | (cc << 24), w1);
        p.rec[t] = rec;
        if (p.objaux) p.objcarry[t] = (uint16_t)(MGX_CODE_EMPTY << 8);
        // the episode start always has nothing carried and step_count 0 (reset(), minigrid.py:851-854)
        if (p.cells0) p.rec0[t] = make_uint2((rec.x & 0x00FFFFFFu) | ((uint32_t)MGX_CODE_EMPTY << 24), p.has_task ? (w1 & 0xFFFF0000u) : 0u);
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicAdd(&p.ctr->invalid_state, 1ull);
}

__global__ __launch_bounds__(256) void k_unpack_state(const PackParams p)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cells = p.W * p.H;
    const int64_t total = p.n * (int64_t)cells;
    if (t < total && (p.grid_out || p.aux_out)) {
        const int64_t e = t / cells;
        const int c = (int)(t - e * cells);
        const uint32_t code = p.cells[e * p.S + c];
        const uint32_t tr = decode_triple(code);
        if (p.grid_out) {
            uint8_t *o = p.grid_out + t * 3;
            o[0] = (uint8_t)tr; o[1] = (uint8_t)(tr >> 8); o[2] = (uint8_t)(tr >> 16);
        }
        if (p.aux_out) p.aux_out[t] = (uint8_t)((code >> 7) | (p.objaux ? p.objaux[e * p.S + c] : ((code & 15u) == MGX_K_GOAL && (code >> 7) ? 0xF0u : 0u)));
    }
    if (t < p.n) {
        const Lane L = unpack_rec(p.rec[t], p.has_task);
        if (p.agent_out) { p.agent_out[t * 3] = L.ax; p.agent_out[t * 3 + 1] = L.ay; p.agent_out[t * 3 + 2] = L.dir; }
        if (p.carry_out) {
            const uint32_t tr = decode_triple(L.carry);
            p.carry_out[t * 3] = (uint8_t)tr; p.carry_out[t * 3 + 1] = (uint8_t)(tr >> 8); p.carry_out[t * 3 + 2] = (uint8_t)(tr >> 16);
        }
        if (p.steps_out) p.steps_out[t] = L.steps;
    }
}

// counter-based action stream shared with the tests (tests/actions.py): mix(seed, env, t) -> 0..6
__device__ __host__ inline uint32_t action_of(uint64_t seed, uint64_t env, uint64_t t)
{
    uint64_t z = seed + env * 0x9E3779B97F4A7C15ull + t * 0xD1B54A32D192ED03ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(((z >> 32) * 7ull) >> 32);
}

__global__ __launch_bounds__(256) void k_fill_actions(uint8_t *out, uint64_t seed, int64_t env0, int64_t t0, int64_t n, int64_t T)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * T) return;
    const int64_t t = i / n, e = i - t * n;
    out[i] = (uint8_t)action_of(seed, (uint64_t)(env0 + e), (uint64_t)(t0 + t));
}

__global__ __launch_bounds__(64) void k_read_stats(const MgxCounters *ctr, double *out2)
{
    double ep = 0.0, rs = 0.0;
    for (int i = threadIdx.x; i < MGX_CTR_SHARDS; i += 64) { ep += (double)ctr->shard[i].episodes; rs += ctr->shard[i].reward_sum; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ep += __shfl_xor(ep, o, 64); rs += __shfl_xor(rs, o, 64); }
    if (threadIdx.x == 0) { out2[0] = ep; out2[1] = rs; }
}


} // namespace

namespace {
__global__ __launch_bounds__(256) void k_direction(const uint2 *__restrict__ rec, uint8_t *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)((rec[i].x >> 16) & 3u);
}
__global__ __launch_bounds__(256) void k_pose(const uint2 *__restrict__ rec, int32_t *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t r = rec[i].x;
        out[3 * i] = (int32_t)(r & 255u); out[3 * i + 1] = (int32_t)((r >> 8) & 255u); out[3 * i + 2] = (int32_t)((r >> 16) & 3u);
    }
}
} // namespace

namespace {
__global__ __launch_bounds__(256) void k_task(uint2 *rec, uint2 *rec0, const uint32_t *set, uint32_t *get, const uint8_t *mask, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (set && (!mask || mask[i])) {
        rec[i].y = (rec[i].y & 0xFFFFu) | (set[i] << 16);
        if (rec0) rec0[i].y = (rec0[i].y & 0xFFFFu) | (set[i] << 16);
    }
    if (get) get[i] = rec[i].y >> 16;
}
} // namespace

namespace {
// Box.contains planes / carried object's hidden pair <-> reference encoding
__global__ __launch_bounds__(256) void k_objstate(const ObjStateParams p)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cells = p.W * p.H;
    if (t < p.n * (int64_t)cells) {
        const int64_t e = t / cells;
        const int c = (int)(t - e * cells);
        if (p.contains_in) {
            const uint8_t *tr = p.contains_in + t * 3;
            const uint32_t ty = tr[0], co = tr[1], st = tr[2];
            uint32_t k = ty;
            bool bad = ty < 1 || ty > 9 || co > 6 || (ty == 1 && (co || st));
            if (ty == 4) { if (st > 2) bad = true; k = st == 0 ? MGX_K_DOOR_OPEN : (st == 1 ? MGX_K_DOOR_CLOSED : MGX_K_DOOR_LOCKED); }
            else if (st != 0) bad = true;
            if (bad) atomicAdd(&p.ctr->invalid_state, 1ull);
            const uint8_t code = (uint8_t)((k & 15u) | ((co & 7u) << 4));
            p.objcont[e * p.S + c] = code;
            if (p.objcont0) p.objcont0[e * p.S + c] = code;
        }
        if (p.contains_out) {
            const uint32_t tr = decode_triple(p.objcont[e * p.S + c]);
            uint8_t *o = p.contains_out + t * 3;
            o[0] = (uint8_t)tr; o[1] = (uint8_t)(tr >> 8); o[2] = (uint8_t)(tr >> 16);
        }
    }
    if (t < p.n) {
        uint32_t w = p.objcarry[t];
        if (p.carry_aux_in) w = (w & 0xFF00u) | (p.carry_aux_in[t] & 0xFEu);
        if (p.carry_contains_in) {
            const uint8_t *tr = p.carry_contains_in + t * 3;
            const uint32_t k = tr[0] == 4 ? (tr[2] == 0 ? MGX_K_DOOR_OPEN : (tr[2] == 1 ? MGX_K_DOOR_CLOSED : MGX_K_DOOR_LOCKED)) : tr[0];
            w = (w & 0x00FFu) | ((((k & 15u) | ((tr[1] & 7u) << 4))) << 8);
        }
        if (p.carry_aux_in || p.carry_contains_in) p.objcarry[t] = (uint16_t)w;
        if (p.carry_aux_out) p.carry_aux_out[t] = (uint8_t)(w & 0xFEu);
        if (p.carry_contains_out) {
            const uint32_t tr = decode_triple(w >> 8);
            p.carry_contains_out[t * 3] = (uint8_t)tr; p.carry_contains_out[t * 3 + 1] = (uint8_t)(tr >> 8); p.carry_contains_out[t * 3 + 2] = (uint8_t)(tr >> 16);
        }
    }
}
} // namespace

hipError_t mgx_launch_objstate(const ObjStateParams &p, hipStream_t st)
{
    const int64_t total = p.n * (int64_t)p.W * p.H;
    hipLaunchKernelGGL(k_objstate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t mgx_launch_task(uint2 *rec, uint2 *rec0, const uint32_t *set, uint32_t *get, const uint8_t *mask, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_task, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, rec0, set, get, mask, n);
    return hipGetLastError();
}

hipError_t mgx_launch_direction(const uint2 *rec, uint8_t *out, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_direction, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, out, n);
    return hipGetLastError();
}

hipError_t mgx_launch_pose(const uint2 *rec, int32_t *out, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_pose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, out, n);
    return hipGetLastError();
}

hipError_t mgx_launch_pack(const PackParams &p, hipStream_t st)
{
    const int64_t total = p.n * (int64_t)p.S;
    hipLaunchKernelGGL(k_pack_state, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t mgx_launch_unpack(const PackParams &p, hipStream_t st)
{
    const int64_t total = p.n * (int64_t)p.W * p.H;
    hipLaunchKernelGGL(k_unpack_state, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t mgx_launch_fill_actions(uint8_t *out, uint64_t seed, int64_t env0, int64_t t0, int64_t n, int64_t T, hipStream_t st)
{
    const int64_t total = n * T;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_actions, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, seed, env0, t0, n, T);
    return hipGetLastError();
}

hipError_t mgx_launch_read_stats(const MgxCounters *ctr, double *out2, hipStream_t st)
{
    hipLaunchKernelGGL(k_read_stats, dim3(1), dim3(64), 0, st, ctr, out2);
    return hipGetLastError();
}

uint32_t mgx_action_of(uint64_t seed, uint64_t env, uint64_t t) { return action_of(seed, env, t); }


hipError_t mgx_preload_state_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_pack_state));
}
