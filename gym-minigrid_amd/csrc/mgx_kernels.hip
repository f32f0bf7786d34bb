// mgx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of libmgx.so.
//
// One launch = one lockstep `env.step(a)` (or `gen_obs()`) for every env of the handle:
//   MiniGridEnv.step       /root/reference/gym_minigrid/minigrid.py:1227-1325
//   gen_obs_grid / gen_obs minigrid.py:1327-1381  (slice :453, rotate_left :439, process_vis :617, encode :571)
//   FullyObsWrapper        /root/reference/gym_minigrid/wrappers.py:326-338
//
// Mapping (wave64, no MFMA -- this is integer gather/scan work, bounded by HBM):
//   * one wavefront owns a TILE of 64 consecutive envs; lane e simulates env e of the tile.
//   * state in HBM is SoA: cells u8[N][S] (1-byte cell codes, x-major like Grid.encode, S = W*H rounded
//     up to 4) and one 8-byte agent record per env.  A tile's cells are one contiguous 64*S-byte run, so the
//     wave streams it with 16-B/lane loads and parks it in LDS with an ODD dword stride per env, which makes
//     the per-lane dynamic gathers (forward cell, 7x7 view) conflict-free up to the agents' own offsets.
//   * the transition touches the forward cell only (one LDS byte read, at most one byte written back).
//   * view: closed form  world = pos + f*(6-vy) + r*(vx-3)  (spec O1) -> 49 LDS byte reads.
//   * occlusion (process_vis) is BIT-SLICED ACROSS THE WAVE: transparency of view cell (vx,vy) for all 64
//     envs is one 64-bit ballot; the reference's two-sweep row flood becomes s_and/s_or on SGPR pairs
//     (scalar unit, off the VALU), and the result is applied with v_cndmask on the inverse ballot.
//   * the 147-byte observation is not dword aligned per env: each lane packs its 49 triples into 37 dwords,
//     funnel-shifts them by its byte phase (3*lane mod 4), merges the boundary dword with its neighbour by
//     DPP row_shr:1 and stores to the wave's LDS image of the tile's 9408 contiguous output bytes, which then
//     leaves as 16-B/lane coalesced global stores.
//   * done / fault flags reduce with wave ballots: one atomic per wave and only when something happened.
//   * auto-reset restores the episode-start snapshot for the (rare) done lanes inside the same launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include "levelgen_core.h"

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

// ------------------------------------------------------------------------------------------------
struct Lane {
    int ax, ay, dir;
    uint32_t carry; // cell code, MGX_CODE_EMPTY = nothing
    int steps;
    uint32_t task;  // per-env task word (16 bits), only with a task rule
};

// record word 1 = step_count, or step_count | task << 16 for handles with a task rule (max_steps <= 65535 there)
__device__ __forceinline__ Lane unpack_rec(uint2 r, int has_task = 0)
{
    Lane L;
    L.ax = r.x & 255u; L.ay = (r.x >> 8) & 255u; L.dir = (r.x >> 16) & 3u; L.carry = r.x >> 24;
    L.steps = has_task ? (int)(r.y & 0xFFFFu) : (int)r.y;
    L.task = has_task ? r.y >> 16 : 0u;
    return L;
}
__device__ __forceinline__ uint2 pack_rec(const Lane &L, int has_task = 0)
{
    return make_uint2((uint32_t)L.ax | ((uint32_t)L.ay << 8) | ((uint32_t)L.dir << 16) | (L.carry << 24),
                      has_task ? ((uint32_t)L.steps & 0xFFFFu) | (L.task << 16) : (uint32_t)L.steps);
}

// Task rules that env subclasses layer on MiniGridEnv.step (they run after the base step, time-out included).
template <int CH, class CellAt>
__device__ __forceinline__ void task_rule(const StepParams &p, const Lane &L, uint32_t act, float &reward, bool &done, CellAt cell_at)
{
    if (p.task == MGX_TASK_FETCH) { // envs/fetch.py:74-86
        if (L.carry != MGX_CODE_EMPTY) {
            done = true;
            reward = ((L.carry & 0x7Fu) == (L.task & 0x7Fu)) ? (float)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)) : 0.f;
        }
    } else if (p.task == MGX_TASK_GOTODOOR) { // envs/gotodoor.py:71-93: `done` next to a door; the target door is the red one
        if (act == 6) {
            const int H = CH ? CH : p.H;
            const int base = L.ax * H + L.ay;
            const uint32_t n4[4] = {cell_at(base + H), cell_at(base - H), cell_at(base + 1), cell_at(base - 1)};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t k = n4[i] & 15u;
                const bool door = k == MGX_K_DOOR_OPEN || k == MGX_K_DOOR_CLOSED || k == MGX_K_DOOR_LOCKED;
                if (door) done = true;
                if (door && ((n4[i] >> 4) & 7u) == 0u) reward = (float)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps));
            }
        }
    }
}

// MiniGridEnv.step without the observation (spec S1-S8), in two halves so that the forward cell can come from
// the LDS tile image (partial-view kernel) or straight from HBM (full-obs kernel).
// Half 1: step_count += 1, fault checks, index of the forward cell (-1: nothing to read, no transition).
// With extended_actions the target of strafe_left (7) / strafe_right (8) is the left / right cell (minigrid.py:1295-1314).
template <int CW, int CH>
__device__ __forceinline__ int transition_begin(const StepParams &p, Lane &L, uint32_t act, bool valid, bool &bad_act, bool &oob)
{
    const int W = CW ? CW : p.W, H = CH ? CH : p.H;
    L.steps += 1;
    const int dir = L.dir;
    const int td = act == 7 ? (dir + 3) & 3 : (act == 8 ? (dir + 1) & 3 : dir);
    const int dx = (td == 0) - (td == 2), dy = (td == 1) - (td == 3); // DIR_TO_VEC (minigrid.py:64-73)
    // front/left/right are all read by the reference (minigrid.py:1239-1243): any of them outside -> assert
    const uint32_t okm = (uint32_t)(L.ax + 1 < W) | ((uint32_t)(L.ay + 1 < H) << 1) | ((uint32_t)(L.ax >= 1) << 2) |
                         ((uint32_t)(L.ay >= 1) << 3);
    const uint32_t need = 0xFu & ~(1u << ((dir + 2) & 3));
    oob = valid && ((okm & need) != need);
    bad_act = valid && act >= (p.extended ? 9u : MGX_NUM_ACTIONS_K);
    if (!valid || oob || bad_act) return -1;
    return (L.ax + dx) * H + (L.ay + dy);
}

// Half 2: the action switch on the target cell code `fc`; returns the cell's new code (== fc: unchanged).
// `cell_at(idx)` reads another cell of the env (only strafe_right onto a goal needs one: the reference tests
// LEFT_cell.overlap there, minigrid.py:1310, and raises AttributeError unless the left cell is a goal as well;
// that case is counted as a fault and treated as "not terminal").
// Hidden Goal/Box state (object_state handles): per-cell byte `aux` = (toggletimes-1)&15 << 4 | (triage_color+1) << 1 and
// per-cell contents code `cont` (Box.contains), plus the same pair for the carried object.  aux == nullptr: the handle
// has no such planes and every Goal/Box is the default one (toggletimes 1, no triage colour, empty).
struct ObjRef {
    uint8_t *aux, *cont; // this env's planes
    uint16_t *carry;     // aux | cont << 8 of the carried object
};
__device__ __forceinline__ bool box_overlappable(const ObjRef &o, int idx, uint32_t code)
{ // Box.can_overlap: color == triage_color (minigrid.py:342-343)
    if (!o.aux) return false;
    const uint32_t tri = (o.aux[idx] >> 1) & 7u;
    return tri != 0u && tri - 1u == ((code >> 4) & 7u);
}

template <int CH, class CellAt>
__device__ __forceinline__ uint32_t transition_apply(const StepParams &p, Lane &L, uint32_t act, uint32_t fc, float &reward, bool &done,
                                                     CellAt cell_at, bool &refbug, int tidx, const ObjRef &o)
{
    const int dir = L.dir;
    if (act >= 7) { // strafe: only reachable with extended_actions
        const int H = CH ? CH : p.H;
        const int td = act == 7 ? (dir + 3) & 3 : (dir + 1) & 3;
        const int tx = (td == 0) - (td == 2), ty = (td == 1) - (td == 3);
        const uint32_t k = fc & 15u;
        const uint32_t OVERLAP = (1u << MGX_K_EMPTY) | (1u << MGX_K_FLOOR) | (1u << MGX_K_DOOR_OPEN) | (1u << MGX_K_GOAL) | (1u << MGX_K_LAVA);
        const int ax0 = L.ax, ay0 = L.ay;
        if (((OVERLAP >> k) & 1u) || (k == MGX_K_BOX && box_overlappable(o, tidx, fc))) { L.ax += tx; L.ay += ty; }
        if (k == MGX_K_GOAL) {
            bool ov;
            if (act == 7) ov = (fc & 0x80u) != 0;
            else {
                const int ld = (dir + 3) & 3;
                const uint32_t lc = cell_at((ax0 + (ld == 0) - (ld == 2)) * H + ay0 + (ld == 1) - (ld == 3));
                if ((lc & 15u) == MGX_K_GOAL) ov = (lc & 0x80u) != 0;
                else { ov = false; refbug = true; }
            }
            if (ov) { done = true; reward = (float)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)); }
        }
        if (k == MGX_K_LAVA) done = true; // no 'v1' special case on the strafe path (minigrid.py:1304-1305,1313-1314)
        return fc;
    }
    const int dx = (dir == 0) - (dir == 2), dy = (dir == 1) - (dir == 3);
    const uint32_t k = fc & 15u;
    uint32_t nc = fc;
    if (act == 0) L.dir = (dir + 3) & 3;
    else if (act == 1) L.dir = (dir + 1) & 3;
    else if (act == 2) {
        // None, Floor, open Door, Goal, Lava can be walked onto (minigrid.py:93,164-166,192,211,245-247)
        const uint32_t OVERLAP = (1u << MGX_K_EMPTY) | (1u << MGX_K_FLOOR) | (1u << MGX_K_DOOR_OPEN) | (1u << MGX_K_GOAL) | (1u << MGX_K_LAVA);
        if (((OVERLAP >> k) & 1u) || (k == MGX_K_BOX && box_overlappable(o, tidx, fc))) { L.ax += dx; L.ay += dy; }
        if (k == MGX_K_GOAL && (fc & 0x80u)) { // goal.overlap (minigrid.py:1259-1261)
            done = true;
            // _reward(): 1 - 0.9*(step_count/max_steps) in Python doubles (minigrid.py:933-937), then f32
            reward = (float)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps));
        }
        if (k == MGX_K_LAVA) { // minigrid.py:1262-1268
            if (p.lava_v1) { done = false; reward = -1.f; }
            else done = true;
        }
    } else if (act == 3) {
        const uint32_t PICK = (1u << MGX_K_KEY) | (1u << MGX_K_BALL) | (1u << MGX_K_BOX);
        if (((PICK >> k) & 1u) && L.carry == MGX_CODE_EMPTY) {
            L.carry = fc; nc = MGX_CODE_EMPTY;
            if (o.aux) { *o.carry = (uint16_t)(o.aux[tidx] | (o.cont[tidx] << 8)); o.aux[tidx] = 0; o.cont[tidx] = MGX_CODE_EMPTY; }
        }
    } else if (act == 4) {
        if (k == MGX_K_EMPTY && L.carry != MGX_CODE_EMPTY) {
            nc = L.carry; L.carry = MGX_CODE_EMPTY;
            if (o.aux) { o.aux[tidx] = (uint8_t)*o.carry; o.cont[tidx] = (uint8_t)(*o.carry >> 8); *o.carry = (uint16_t)(MGX_CODE_EMPTY << 8); }
        }
    } else if (act == 5 && o.aux && (k == MGX_K_GOAL || k == MGX_K_BOX)) {
        // Goal.toggle / Box.toggle with their hidden state (minigrid.py:171-181,355-364)
        uint32_t a = o.aux[tidx];
        int tt = (int)(((a >> 4) + 1u) & 15u);
        const uint32_t tri = (a >> 1) & 7u;
        const bool goal = k == MGX_K_GOAL;
        if (!goal || tt > 0) {
            tt = tt > 0 ? tt - 1 : 0; // Box counts below zero in Python; every value <= 0 behaves the same
            a = (a & 0x0Fu) | ((uint32_t)((tt - 1) & 15) << 4);
            if (tt <= 0 && tri == 0u) { // Goal: removed; Box: replaced by its contents
                nc = goal ? (uint32_t)MGX_CODE_EMPTY : (uint32_t)o.cont[tidx];
                o.cont[tidx] = MGX_CODE_EMPTY;
                a = 0;
            } else if (tt <= 0) nc = (fc & 0x8Fu) | ((tri - 1u) << 4); // self.color = self.triage_color
            o.aux[tidx] = (uint8_t)a;
        }
    } else if (act == 5) {
        if (k == MGX_K_DOOR_LOCKED) { // Door.toggle (minigrid.py:252-262)
            if ((L.carry & 15u) == MGX_K_KEY && ((L.carry >> 4) & 7u) == ((fc >> 4) & 7u)) nc = (fc & 0xF0u) | MGX_K_DOOR_OPEN;
        } else if (k == MGX_K_DOOR_OPEN) nc = (fc & 0xF0u) | MGX_K_DOOR_CLOSED;
        else if (k == MGX_K_DOOR_CLOSED) nc = (fc & 0xF0u) | MGX_K_DOOR_OPEN;
        else if (k == MGX_K_GOAL) { if (!(fc & 0x80u)) nc = MGX_CODE_EMPTY; } // Goal.toggle, toggletimes=1 (minigrid.py:171-181)
        else if (k == MGX_K_BOX) nc = MGX_CODE_EMPTY;                         // Box.toggle, contains=None (minigrid.py:355-364)
    } // act == 6 ("done"): pass (minigrid.py:1291-1293)
    return nc;
}

// auto-reset of the hidden object state: planes back to the snapshot, nothing carried
__device__ __forceinline__ void restore_objstate(const StepParams &p, int64_t env)
{
    if (!p.objaux) return;
    const uint32_t *a0 = reinterpret_cast<const uint32_t *>(p.objaux0 + env * p.S), *c0 = reinterpret_cast<const uint32_t *>(p.objcont0 + env * p.S);
    uint32_t *a = reinterpret_cast<uint32_t *>(p.objaux + env * p.S), *c = reinterpret_cast<uint32_t *>(p.objcont + env * p.S);
    for (int i = 0; i < (p.S >> 2); i++) { a[i] = a0[i]; c[i] = c0[i]; }
    p.objcarry[env] = (uint16_t)(MGX_CODE_EMPTY << 8);
}

// Restore this lane's env to its episode-start snapshot (LDS image + HBM).  Each done lane copies its own
// S bytes with all loads issued back to back, so a wave pays ONE memory latency however many of its envs
// finished (a wave-cooperative loop over done envs would pay one per env: measured 2x slower end to end on
// LavaCrossing, where 40% of the waves see a reset every step).
template <int CS>
__device__ __forceinline__ void restore_own(const StepParams &p, int64_t env, uint8_t *g)
{
    const int S = CS ? CS : p.S;
    uint32_t *l32 = reinterpret_cast<uint32_t *>(g);
    if constexpr (CS != 0 && (CS % 16) == 0) {
        const uint4 *s = reinterpret_cast<const uint4 *>(p.cells0 + env * S);
        uint4 *d = reinterpret_cast<uint4 *>(p.cells + env * S);
#pragma unroll
        for (int i = 0; i < CS / 16; i++) {
            const uint4 v = s[i];
            l32[4 * i + 0] = v.x; l32[4 * i + 1] = v.y; l32[4 * i + 2] = v.z; l32[4 * i + 3] = v.w;
            d[i] = v;
        }
    } else {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(p.cells0 + env * S);
        uint32_t *d = reinterpret_cast<uint32_t *>(p.cells + env * S);
        if constexpr (CS != 0) { // e.g. 9x9: 21 dword loads, all in flight before the first use
            uint32_t v[CS / 4];
#pragma unroll
            for (int i = 0; i < CS / 4; i++) v[i] = s[i];
#pragma unroll
            for (int i = 0; i < CS / 4; i++) { l32[i] = v[i]; d[i] = v[i]; }
        } else {
            const int SD = S >> 2;
#pragma unroll 8
            for (int i = 0; i < SD; i++) {
                const uint32_t v = s[i];
                l32[i] = v;
                d[i] = v;
            }
        }
    }
}

__device__ __forceinline__ void wave_stats(const StepParams &p, bool valid, bool done, float reward, bool bad_act, bool oob, int lane, int tile)
{
    MgxCounterShard *sh = &p.ctr->shard[tile & (MGX_CTR_SHARDS - 1)];
    const u64 md = __ballot(valid && done), mr = __ballot(valid && reward != 0.f);
    const u64 ma = __ballot(bad_act), mo = __ballot(oob);
    if (md && lane == 0) atomicAdd(&sh->episodes, (u64)__popcll(md));
    if (ma && lane == 0) atomicAdd(&p.ctr->invalid_actions, (u64)__popcll(ma));
    if (mo && lane == 0) atomicAdd(&p.ctr->out_of_bounds, (u64)__popcll(mo));
    if (mr) { // rare: wave-reduce the rewards in f64, one atomic
        double r = valid ? (double)reward : 0.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0) atomicAdd(&sh->reward_sum, r);
    }
}

// ------------------------------------------------------------------------------------------------
// Partial (VxVx3, V = agent_view_size, odd) observation of lane's env -> wave's LDS image of the tile's output -> HBM.
// B = 3*V*V bytes per env is always 3 mod 4 for odd V, so the byte-phase logic below is the same for every V.
template <int CW, int CH, int V, bool ALT>
__device__ __forceinline__ void emit_partial_obs(const StepParams &p, const Lane &L, uint8_t *lds, const uint8_t *g,
                                                 int64_t env0, int lane)
{
    constexpr int B = V * V * 3;       // bytes per observation (147 for V = 7)
    constexpr int NDW = (B + 1) / 4;   // dwords holding one observation, the last with 3 valid bytes (37)
    constexpr int NQ = (V * V) / 4;    // groups of 4 cells = 3 dwords (12), plus one last cell
    const int W = CW ? CW : p.W, H = CH ? CH : p.H;
    const int dir = L.dir;
    const int dx = (dir == 0) - (dir == 2), dy = (dir == 1) - (dir == 3);
    const int rx = -dy, ry = dx; // right_vec (minigrid.py:1102-1109)
    const int base = L.ax * H + L.ay;
    const int sf = dx * H + dy, sr = rx * H + ry;

    // in-bounds is separable: the forward coordinate depends on d = 6-vy only, the lateral one on l = vx-3 only
    bool vf[V], vl[V];
#pragma unroll
    for (int d = 0; d < V; d++) vf[d] = (unsigned)(L.ax + dx * d) < (unsigned)W && (unsigned)(L.ay + dy * d) < (unsigned)H;
#pragma unroll
    for (int k = 0; k < V; k++) vl[k] = (unsigned)(L.ax + rx * (k - V / 2)) < (unsigned)W && (unsigned)(L.ay + ry * (k - V / 2)) < (unsigned)H;

    // gather: code[vx][vy]; outside the grid -> grey wall (Grid.slice, minigrid.py:465-469)
    uint32_t code[V][V];
#pragma unroll
    for (int vy = V - 1; vy >= 0; vy--) {
        const int rowbase = base + (V - 1 - vy) * sf;
#pragma unroll
        for (int vx = 0; vx < V; vx++) {
            const bool inb = vf[V - 1 - vy] && vl[vx];
            const int idx = inb ? rowbase + (vx - V / 2) * sr : base;
            const uint32_t c = g[idx];
            code[vx][vy] = inb ? c : (uint32_t)MGX_CODE_WALL_GREY;
        }
    }

    if (ALT && !p.see_through) {
        // the fork's alternative visibility model, default_vis=False (minigrid.py:649-709), per lane on column bit
        // masks (bit j of m[i] = view cell (i, j) visible): its data-dependent `break`s become per-lane alive flags.
        constexpr int PX = V / 2, PY = V - 1;
        uint32_t m[V], oq[V];
#pragma unroll
        for (int i = 0; i < V; i++) {
            m[i] = 0u;
            uint32_t o = 0u;
#pragma unroll
            for (int j = 0; j < V; j++) o |= (uint32_t)is_opaque(code[i][j]) << j;
            oq[i] = o;
        }
        m[PX] = 1u << PY;
        bool alive = true;
#pragma unroll
        for (int i = PX + 1; i < V; i++) { if (alive) m[i] |= 1u << PY; alive = alive && !((oq[i] >> PY) & 1u); }
        alive = true;
#pragma unroll
        for (int i = PX - 1; i >= 0; i--) { if (alive) m[i] |= 1u << PY; alive = alive && !((oq[i] >> PY) & 1u); }
        alive = true;
#pragma unroll
        for (int j = V - 2; j >= 0; j--) { if (alive) m[PX] |= 1u << j; alive = alive && !((oq[PX] >> j) & 1u); }
#pragma unroll
        for (int i = PX + 1; i < V; i++) { // right side; hideside = True
            alive = true;
#pragma unroll
            for (int j = V - 2; j >= 0; j--) {
                const bool cond = alive && ((m[i] >> (j + 1)) & 1u) && ((m[i - 1] >> j) & 1u);
                const bool c = (oq[i] >> j) & 1u, ca = (oq[i] >> (j + 1)) & 1u, cb = (oq[i - 1] >> j) & 1u;
                const bool brk = cond && !c && (ca || cb);
                alive = alive && !brk;
                if (cond && !brk) m[i] |= 1u << j;
            }
        }
#pragma unroll
        for (int i = PX - 1; i >= 0; i--) { // left side
            alive = true;
#pragma unroll
            for (int j = V - 2; j >= 0; j--) {
                const bool cond = alive && ((m[i] >> (j + 1)) & 1u) && ((m[i + 1] >> j) & 1u);
                const bool c = (oq[i] >> j) & 1u, ca = (oq[i] >> (j + 1)) & 1u, cb = (oq[i + 1] >> j) & 1u;
                const bool brk = cond && !c && (ca || cb);
                alive = alive && !brk;
                if (cond && !brk) m[i] |= 1u << j;
            }
        }
#pragma unroll
        for (int i = 0; i < V; i++)
#pragma unroll
            for (int j = 0; j < V; j++) code[i][j] = ((m[i] >> j) & 1u) ? code[i][j] : 0u;
    }
    // occlusion, bit-sliced over the wave (process_vis default branch, minigrid.py:617-648; spec O4)
    if (!ALT && !p.see_through) {
        u64 vis[V];
#pragma unroll
        for (int i = 0; i < V; i++) vis[i] = (i == V / 2) ? ~0ull : 0ull;
#pragma unroll
        for (int vy = V - 1; vy >= 0; vy--) {
            u64 T[V];
#pragma unroll
            for (int i = 0; i < V; i++) T[i] = __ballot(!is_opaque(code[i][vy]));
#pragma unroll
            for (int i = 0; i < V - 1; i++) vis[i + 1] |= vis[i] & T[i]; // left-to-right sweep (:624-635)
#pragma unroll
            for (int i = V - 1; i >= 1; i--) vis[i - 1] |= vis[i] & T[i]; // right-to-left sweep (:637-648)
#pragma unroll
            for (int i = 0; i < V; i++) code[i][vy] = lane_bit(vis[i]) ? code[i][vy] : 0u; // unseen -> (0,0,0)
            if (vy > 0) {
                u64 s[V], nx[V];
#pragma unroll
                for (int i = 0; i < V; i++) s[i] = vis[i] & T[i]; // visible and transparent: lights the row above
#pragma unroll
                for (int i = 0; i < V; i++) nx[i] = s[i] | (i > 0 ? s[i - 1] : 0ull) | (i < V - 1 ? s[i + 1] : 0ull);
#pragma unroll
                for (int i = 0; i < V; i++) vis[i] = nx[i];
            }
        }
    }
    // the agent's own cell shows what it carries, after occlusion; always visible (minigrid.py:1349-1356)
    code[V / 2][V - 1] = L.carry;

    // pack 49 triples (image[vx][vy][c], vx-major) into 37 dwords; v_perm_b32 picks 4 of the 8 bytes {S0,S1}
    uint32_t D[NDW];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t c0 = decode_triple(code[(4 * q) / V][(4 * q) % V]);
        const uint32_t c1 = decode_triple(code[(4 * q + 1) / V][(4 * q + 1) % V]);
        const uint32_t c2 = decode_triple(code[(4 * q + 2) / V][(4 * q + 2) % V]);
        const uint32_t c3 = decode_triple(code[(4 * q + 3) / V][(4 * q + 3) % V]);
        D[3 * q + 0] = __builtin_amdgcn_perm(c1, c0, 0x04020100u); // c0.b0 c0.b1 c0.b2 c1.b0
        D[3 * q + 1] = __builtin_amdgcn_perm(c2, c1, 0x05040201u); // c1.b1 c1.b2 c2.b0 c2.b1
        D[3 * q + 2] = __builtin_amdgcn_perm(c3, c2, 0x06050402u); // c2.b2 c3.b0 c3.b1 c3.b2
    }
    D[NDW - 1] = decode_triple(code[V - 1][V - 1]); // 3 bytes

    // byte phase of this env inside the tile's contiguous output: B*lane = 4*P + s.  Q = D delayed by s bytes:
    // Q[k] = bytes (4-s)..(7-s) of {D[k], D[k-1]}  -> one v_perm_b32 with a per-lane selector
    const uint32_t s = (3u * (uint32_t)lane) & 3u;
    const uint32_t sel = 0x07060504u - s * 0x01010101u;
    uint32_t Q[NDW + 1];
    Q[0] = __builtin_amdgcn_perm(D[0], 0u, sel);
#pragma unroll
    for (int k = 1; k < NDW; k++) Q[k] = __builtin_amdgcn_perm(D[k], D[k - 1], sel);
    Q[NDW] = __builtin_amdgcn_perm(0u, D[NDW - 1], sel);
    // the last, partial dword belongs to the next lane's first dword
    const uint32_t tail = (s == 0u) ? Q[NDW - 1] : Q[NDW];
    const uint32_t prev_tail = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tail, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
    if (s != 0u) Q[0] |= prev_tail;

    // The tile's 64*B output bytes (9408 for V = 7) go through LDS in two halves of 32 envs (32*B = 16 * 2B bytes): the
    // LDS image is then no larger than the grid image it overlays, which doubles the resident waves per CU.
    constexpr int HALF = 32 * B, CHUNKS = 2 * B; // bytes and 16-B chunks per half
    const int64_t nv = p.n - env0; // valid envs in this tile (>= 1)
    const int lim_all = nv >= 64 ? 64 * B : (int)nv * B;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(lds) + (((uint32_t)B * (uint32_t)(lane & 31)) >> 2);
    const uint4 *l128 = reinterpret_cast<const uint4 *>(lds);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        wave_sync(); // every lane is done reading what this image overlays (grid image / previous half)
        if ((lane >> 5) == h) {
#pragma unroll
            for (int k = 0; k < NDW - 1; k++) o32[k] = Q[k];
            if (s != 0u) o32[NDW - 1] = Q[NDW - 1];
        }
        wave_sync();
        uint8_t *dst = p.obs + env0 * B + h * HALF;
        const int lim = lim_all - h * HALF; // valid bytes of this half (may be <= 0 in a tail tile)
        if (lim >= HALF) {
#pragma unroll
            for (int i = 0; i < (CHUNKS + 63) / 64; i++) {
                const int c = lane + 64 * i;
                if (c < CHUNKS) nt_store16(reinterpret_cast<uint4 *>(dst) + c, l128[c]);
            }
        } else if (lim > 0) {
            for (int c = lane; c < CHUNKS; c += 64) {
                if (16 * c + 16 <= lim) reinterpret_cast<uint4 *>(dst)[c] = l128[c];
                else
                    for (int b = 16 * c; b < lim; b++) dst[b] = lds[b];
            }
        }
    }
}

// Full-grid observation: Grid.encode() + agent marker (wrappers.py:326-338).  The wave decodes its tile's
// 64*W*H cells cooperatively: lane handles output dwords lane, lane+64, ... (coalesced 4-B stores).
template <int CW, int CH>
__device__ __forceinline__ void emit_full_obs(const StepParams &p, const Lane &L, bool valid, uint8_t *lds, uint8_t *g,
                                              int LS, int64_t env0, int lane)
{
    const int W = CW ? CW : p.W, H = CH ? CH : p.H;
    const int cells = W * H;
    if (valid) g[L.ax * H + L.ay] = (uint8_t)(MGX_K_AGENT | (L.dir << 4)); // LDS copy only
    wave_sync();
    const int64_t nv = p.n - env0;
    const int n_env = nv >= 64 ? 64 : (int)nv;
    uint8_t *dst = p.obs + env0 * cells * 3;
    if constexpr (CW != 0 && ((CW * CH) % 16) == 0) {
        // fast path: a lane decodes 16 consecutive cells of one env (4 LDS dwords) into 48 output bytes = 3 x 16-B stores
        constexpr int UPE = (CW * CH) / 16; // units per env
        const int n_units = n_env * UPE;
        const uint32_t *l32 = reinterpret_cast<const uint32_t *>(lds);
        uint8_t *xpose = lds + 64 * LS; // 3 KiB scratch behind the grid image (sized by the host: wave_lds)
        for (int u0 = 0; u0 < n_units; u0 += 64) { // wave-uniform trip count
            const int u = u0 + lane;
            const int uc = u < n_units ? u : n_units - 1;
            const int e = uc / UPE, o = uc - e * UPE;
            const uint32_t *src = l32 + e * (LS >> 2) + o * 4;
            uint32_t t[16];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t w = src[i];
#pragma unroll
                for (int j = 0; j < 4; j++) t[4 * i + j] = decode_triple_full((w >> (8 * j)) & 255u);
            }
            uint32_t D[12];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                D[3 * q + 0] = __builtin_amdgcn_perm(t[4 * q + 1], t[4 * q + 0], 0x04020100u);
                D[3 * q + 1] = __builtin_amdgcn_perm(t[4 * q + 2], t[4 * q + 1], 0x05040201u);
                D[3 * q + 2] = __builtin_amdgcn_perm(t[4 * q + 3], t[4 * q + 2], 0x06050402u);
            }
            if (u0 + 64 <= n_units) {
                // 64 lanes x 48 B = 3 KiB contiguous: transpose through LDS so each store instruction is 1 KiB contiguous
                uint4 *x4 = reinterpret_cast<uint4 *>(xpose);
                x4[3 * lane + 0] = make_uint4(D[0], D[1], D[2], D[3]);
                x4[3 * lane + 1] = make_uint4(D[4], D[5], D[6], D[7]);
                x4[3 * lane + 2] = make_uint4(D[8], D[9], D[10], D[11]);
                wave_sync();
                uint4 *o4 = reinterpret_cast<uint4 *>(dst + (size_t)u0 * 48);
                const uint4 a = x4[lane], b = x4[64 + lane], c = x4[128 + lane];
                o4[lane] = a; o4[64 + lane] = b; o4[128 + lane] = c;
                wave_sync();
            } else if (u < n_units) {
                uint4 *o4 = reinterpret_cast<uint4 *>(dst + (size_t)u * 48);
                o4[0] = make_uint4(D[0], D[1], D[2], D[3]);
                o4[1] = make_uint4(D[4], D[5], D[6], D[7]);
                o4[2] = make_uint4(D[8], D[9], D[10], D[11]);
            }
        }
        return;
    }

    const int n_dw = n_env * cells * 3 / 4;      // whole dwords
    const int n_bytes = n_env * cells * 3;
    for (int j = lane; j < n_dw; j += 64) {
        const int b = 4 * j;
        const int q = b / 3, r = b - 3 * q; // first cell of the tile's cell stream touched by this dword, byte phase
        const int e0 = q / cells, c0 = q - e0 * cells;
        int e1 = e0, c1 = c0 + 1;
        if (c1 == cells) { c1 = 0; e1 = e0 + 1; }
        const uint32_t t0 = decode_triple_full(lds[e0 * LS + c0]);
        const uint32_t t1 = (e1 < 64) ? decode_triple_full(lds[e1 * LS + c1]) : 0u;
        const u64 both = (u64)t0 | ((u64)t1 << 24);
        reinterpret_cast<uint32_t *>(dst)[j] = (uint32_t)(both >> (8 * r));
    }
    // (n_env*cells*3 is a multiple of 4 unless the tail tile has an odd cell count: finish by bytes)
    for (int b = 4 * n_dw + lane; b < n_bytes; b += 64) {
        const int q = b / 3, r = b - 3 * q;
        const int e0 = q / cells, c0 = q - e0 * cells;
        dst[b] = (uint8_t)(decode_triple_full(lds[e0 * LS + c0]) >> (8 * r));
    }
}

// ------------------------------------------------------------------------------------------------
// (device RNG + per-level generator, used by k_levelgen below)
// Word source of the lane-per-level fast path: a bounded window of the env's MT19937 block copied into LDS.  Running
// past it marks the level for the slow path (alive() == false stops the generators' rejection loops).
struct WinRng {
    const uint32_t *buf; // LDS window, indexed by the absolute position in the block
    int idx, limit;
    bool overflow;

    __device__ __forceinline__ bool alive() const { return !overflow; }
    __device__ __forceinline__ uint32_t next32()
    {
        if (idx >= limit) { overflow = true; return 0u; } // masked draws end on 0; place_obj-style loops test alive()
        return lg_temper(buf[idx++]);
    }
};

// Word source of the slow path (one lane of a wave): the env's whole block in LDS, unlimited length.  When the block is
// used up it switches to the next one -- prebuilt by the whole wave if that was foreseeable, else built here word by
// word (rare; loops kept rolled: this sits in a dozen call sites of the generators).
struct DevRng {
    uint32_t *a, *b; // current block / scratch for the next one (624 words each, LDS)
    int idx;
    bool have_b;
    int advanced;    // blocks consumed: > 0 means `a` must be written back as the env's new state

    __device__ __forceinline__ bool alive() const { return true; }
    __device__ __forceinline__ uint32_t next32()
    {
        if (idx >= 624) {
            if (!have_b) {
#pragma nounroll
                for (int k = 0; k < 227; k++) b[k] = lg_twist_word(a[k], a[k + 1], a[k + 397]);
#pragma nounroll
                for (int k = 227; k < 623; k++) b[k] = lg_twist_word(a[k], a[k + 1], b[k - 227]);
                b[623] = lg_twist_word(a[623], b[0], b[396]);
            }
            uint32_t *t = a; a = b; b = t;
            have_b = false;
            idx = 0;
            advanced++;
        }
        return lg_temper(a[idx++]);
    }
};

// One level, generated by one wave into its LDS workspace and written back coalesced.
__device__ __forceinline__ void levelgen_one(const LevelGenParams &p, int64_t env, uint8_t *base, int lane)
{
    uint32_t *cur = reinterpret_cast<uint32_t *>(base), *nxt = cur + 624;
    int *res = reinterpret_cast<int *>(base + 2 * 624 * 4); // [0]=idx after, [1]=packed agent, [2]=overflow, [3]=#cmds
    int16_t *ws = reinterpret_cast<int16_t *>(res + 4);
    LgCmd *cmds = reinterpret_cast<LgCmd *>(ws + MGX_LG_WS_WORDS);
    uint32_t *mt = p.mt + env * 624;
    { // all ten loads of the block in flight before the first LDS write (a rolled loop pays one latency per trip)
        uint32_t v[10];
#pragma unroll
        for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; v[i] = k < 624 ? mt[k] : 0u; }
#pragma unroll
        for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; if (k < 624) cur[k] = v[i]; }
    }
    const int idx0 = (int)p.mt_idx[env];
    wave_sync();
    // If the read index is within 64 words of the end of the block the level will probably run into the next block:
    // the whole wave builds it first (the recurrence is 3 data-parallel phases + 1 word).
    const bool pre = idx0 + 64 > 624;
    if (pre) {
        for (int k = lane; k < 227; k += 64) nxt[k] = lg_twist_word(cur[k], cur[k + 1], cur[k + 397]);
        wave_sync();
        for (int k = 227 + lane; k < 454; k += 64) nxt[k] = lg_twist_word(cur[k], cur[k + 1], nxt[k - 227]);
        wave_sync();
        for (int k = 454 + lane; k < 623; k += 64) nxt[k] = lg_twist_word(cur[k], cur[k + 1], nxt[k - 227]);
        wave_sync();
        if (lane == 0) nxt[623] = lg_twist_word(cur[623], nxt[0], nxt[396]);
        wave_sync();
    }
    if (lane == 0) {
        DevRng r;
        r.a = cur; r.b = nxt; r.idx = idx0; r.have_b = pre; r.advanced = 0;
        LgLevel L;
        L.cmds = cmds; L.ncmd = 0; L.W = p.cfg.width; L.H = p.cfg.height; L.ax = L.ay = -1; L.adir = 0; L.ws = ws;
        lg_generate(p.cfg, r, L);
        res[0] = r.idx;
        res[1] = (L.ax & 255) | ((L.ay & 255) << 8) | ((L.adir & 3) << 16);
        res[2] = r.advanced ? (r.a == cur ? 1 : 2) : 0; // which LDS buffer holds the env's new current block
        res[3] = L.ncmd;
        ws[MGX_LG_WS_WORDS - 1] = (int16_t)L.task; // hand the task word to the write-back below
    }
    wave_sync();
    const int idx1 = res[0];
    { // paint: every lane evaluates the command list for 4 consecutive cells and stores one dword of codes
        const int ncmd = res[3], H = p.cfg.height, cells = p.cfg.width * H;
        uint32_t *dst = reinterpret_cast<uint32_t *>(p.cells0 + env * p.S);
        for (int k = lane; k < (p.S >> 2); k += 64) {
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = 4 * k + b;
                if (c < cells) { const int x = c / H; w |= lg_cell_code(cmds, ncmd, x, c - x * H) << (8 * b); }
            }
            dst[k] = w;
        }
    }
    if (res[2]) { // moved into a later block: it becomes the env's state
        const uint32_t *blk = res[2] == 1 ? cur : nxt;
        for (int k = lane; k < 624; k += 64) mt[k] = blk[k];
    }
    if (lane == 0) {
        p.mt_idx[env] = (uint32_t)idx1;
        p.agent0[env] = make_uint2((uint32_t)res[1] | ((uint32_t)MGX_CODE_EMPTY << 24), (uint32_t)(uint16_t)ws[MGX_LG_WS_WORDS - 1] << 16);
    }
    wave_sync();
}

template <int CW, int CH, int MODE, int V, bool ALT = false>
__global__ __launch_bounds__(256) void k_step(const StepParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= p.n_tiles) return; // wave-uniform
    constexpr int CS = (CW && CH) ? ((CW * CH + 3) & ~3) : 0;
    const int S = CS ? CS : p.S;
    const int LS = p.LS;
    uint8_t *lds = smem + (size_t)wv * p.wave_lds;
    const int64_t env0 = (int64_t)tile * 64;
    const int64_t env = env0 + lane;
    const bool valid = env < p.n;

    const uint2 rec = p.agent[env]; // agent/cells arrays are padded to whole tiles
    uint32_t act = 6;
    if (p.do_step && valid) act = __builtin_nontemporal_load(&p.actions[env]);
    const bool crash = p.task == MGX_TASK_DYNOBS && (act & 0x80u); // k_dynobs' verdict rides on the folded action
    if (p.task == MGX_TASK_DYNOBS) act &= 0x7Fu;
    stage_tile<CS>(p.cells, env0, S, LS, lds, lane);
    wave_sync();

    Lane L = unpack_rec(rec, p.task);
    uint8_t *g = lds + lane * LS;
    float reward = 0.f;
    bool done = false, bad_act = false, oob = false;
    if (p.do_step) {
        const int fidx = transition_begin<CW, CH>(p, L, act, valid, bad_act, oob);
        if (fidx >= 0) {
            const uint32_t fc = g[fidx];
            // hidden object state rides only on the run-time-size kernels (mgx_launch_step routes there): pruned from the sized ones
            const bool has_obj = CW == 0 && p.objaux != nullptr;
            const ObjRef obj = {has_obj ? p.objaux + env * S : nullptr, has_obj ? p.objcont + env * S : nullptr, has_obj ? p.objcarry + env : nullptr};
            const uint32_t nc = transition_apply<CH>(p, L, act, fc, reward, done, [&](int i) -> uint32_t { return g[i]; }, oob, fidx, obj);
            if (nc != fc) g[fidx] = (uint8_t)nc;
            if (valid && L.steps >= p.max_steps) done = true; // minigrid.py:1320-1321
            if (p.task) task_rule<CH>(p, L, act, reward, done, [&](int i) -> uint32_t { return g[i]; });
            // the one cell a transition can change; skipped when the env is about to be restored anyway
            if (nc != fc && !(p.auto_reset && done)) p.cells[env * S + fidx] = (uint8_t)nc;
        } else if (valid && L.steps >= p.max_steps) done = true;
        if (crash) { reward = -1.f; done = true; } // envs/dynamicobstacles.py:83-86
        if (p.reward && valid) __builtin_nontemporal_store(reward, &p.reward[env]);
        if (p.done && valid) __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &p.done[env]);
        wave_stats(p, valid, done, reward, bad_act, oob, lane, tile);
        if (p.auto_reset && valid && done) {
            restore_own<CS>(p, env, g);
            if (CW == 0) restore_objstate(p, env);
            L = unpack_rec(p.agent0[env], p.task);
            if (p.regen) p.regen[env] = 1; // the next-level buffer was consumed: k_levelgen refills it after this launch
        }
        if (valid) p.agent[env] = pack_rec(L, p.task);
    }
    if (p.obs) {
        if (MODE == 0) emit_partial_obs<CW, CH, V, ALT>(p, L, lds, g, env0, lane);
        else emit_full_obs<CW, CH>(p, L, valid, lds, g, LS, env0, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// FullyObs, direct form (W*H a multiple of 4): no tile image in LDS, and a tile of 64 envs belongs to a whole
// 256-thread BLOCK so that the cooperative decode has 4x the waves (a wave walking a 16x16 tile alone needs 16
// dependent load->decode->store rounds and the kernel becomes latency-bound: measured 48 us wave lifetime).
//   prefetch : every thread issues the coalesced loads of the cell units it will decode, before anything else;
//   phase A  : wave 0, lane-per-env: transition with the forward cell gathered straight from HBM (its line is part
//              of the prefetch, so no extra HBM bytes), results handed over through LDS;
//   phase B  : all 4 waves: unit = 4 consecutive cells (one dword) -> 12 output bytes, stored with ONE
//              global_store_dwordx3 per lane: consecutive lanes write consecutive 12-byte records, so every store
//              instruction covers 768 contiguous, line-aligned bytes (16-cell units + 3 dwordx4 stores at a 48-byte
//              lane stride measured ~20% slower: each lane's 16 bytes was its own L1 transaction).
//              Envs that finished re-read the episode-start snapshot and write it back (a coalesced restore); the
//              one cell phase A changed and the agent marker are patched in registers, so phase B never depends
//              on phase A's global stores being visible.
template <int CW, int CH>
__global__ __launch_bounds__(256) void k_step_fulldirect(const StepParams p)
{
    __shared__ uint32_t s_info[64]; // per env: agent idx | dir<<16 | reset<<18 | (1<<19 if a cell changed)
    __shared__ uint32_t s_wr[64];   // changed cell: idx | code<<16
    __shared__ uint32_t s_lut[256]; // cell code -> (type | color<<8 | state<<16)
    constexpr int CS = CW * CH;
    constexpr int KPF = CS ? (CS / 4 * 64 + 255) / 256 : 0; // prefetched units per thread (compile-time sizes only)
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int H = CH ? CH : p.H;
    const int S = CS ? CS : p.S;
    const int UPE = S >> 2; // 4-cell units per env
    const int64_t env0 = (int64_t)tile * 64;
    const int64_t nv = p.n - env0;
    const int n_units = (nv >= 64 ? 64 : (int)nv) * UPE;
    const uint32_t *cells32 = reinterpret_cast<const uint32_t *>(p.cells + env0 * S);

    uint32_t pf[KPF ? KPF : 1];
#pragma unroll
    for (int k = 0; k < KPF; k++) {
        const int u = tid + 256 * k;
        pf[k] = cells32[u < n_units ? u : 0];
    }
    s_lut[tid] = decode_triple_full(tid);

    if (tid < 64) { // phase A: wave 0
        const int lane = tid;
        const int64_t env = env0 + lane;
        const bool valid = env < p.n;
        Lane L = unpack_rec(p.agent[env], p.task);
        uint32_t act = 6;
        if (p.do_step && valid) act = __builtin_nontemporal_load(&p.actions[env]);
        const bool crash = p.task == MGX_TASK_DYNOBS && (act & 0x80u);
        if (p.task == MGX_TASK_DYNOBS) act &= 0x7Fu;
        float reward = 0.f;
        bool done = false, bad_act = false, oob = false, reset = false;
        uint32_t wr = 0, changed = 0;
        if (p.do_step) {
            const int fidx = transition_begin<CW, CH>(p, L, act, valid, bad_act, oob);
            if (fidx >= 0) {
                const uint32_t fc = p.cells[env * S + fidx];
                // hidden object state rides only on the run-time-size kernels (mgx_launch_step routes there): pruned from the sized ones
            const bool has_obj = CW == 0 && p.objaux != nullptr;
            const ObjRef obj = {has_obj ? p.objaux + env * S : nullptr, has_obj ? p.objcont + env * S : nullptr, has_obj ? p.objcarry + env : nullptr};
                const uint32_t nc = transition_apply<CH>(p, L, act, fc, reward, done,
                                                         [&](int i) -> uint32_t { return p.cells[env * S + i]; }, oob, fidx, obj);
                if (valid && L.steps >= p.max_steps) done = true;
                if (p.task) task_rule<CH>(p, L, act, reward, done, [&](int i) -> uint32_t { return (i == fidx) ? nc : (uint32_t)p.cells[env * S + i]; });
                if (nc != fc && !(p.auto_reset && done)) {
                    p.cells[env * S + fidx] = (uint8_t)nc;
                    wr = (uint32_t)fidx | (nc << 16);
                    changed = 1;
                }
            } else if (valid && L.steps >= p.max_steps) done = true;
            if (crash) { reward = -1.f; done = true; }
            if (p.reward && valid) __builtin_nontemporal_store(reward, &p.reward[env]);
            if (p.done && valid) __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &p.done[env]);
            wave_stats(p, valid, done, reward, bad_act, oob, lane, tile);
            if (p.auto_reset && valid && done) {
                L = unpack_rec(p.agent0[env], p.task);
                if (CW == 0) restore_objstate(p, env);
                reset = true;
                if (p.regen) p.regen[env] = 1;
            }
            if (valid) p.agent[env] = pack_rec(L, p.task);
        }
        s_info[lane] = (uint32_t)(L.ax * H + L.ay) | ((uint32_t)L.dir << 16) | ((uint32_t)reset << 18) | (changed << 19);
        s_wr[lane] = wr;
    }
    __syncthreads();

    struct __attribute__((packed, aligned(4))) Out12 { uint32_t a, b, c; };
    Out12 *dst = p.obs ? reinterpret_cast<Out12 *>(p.obs + env0 * (int64_t)S * 3) : nullptr;
    constexpr int NIT = KPF ? KPF : 1;
    const int n_iter = KPF ? KPF : (n_units + 255) / 256;
#pragma unroll
    for (int kk = 0; kk < NIT; kk++)
    for (int k = kk; k < (KPF ? kk + 1 : n_iter); k++) { // compile-time sizes: fully unrolled (pf[] stays in registers)
        const int u = tid + 256 * k;
        if (u >= n_units) break;
        const int e = u / UPE, o = u - e * UPE;
        const uint32_t info = s_info[e];
        const bool rst = (info >> 18) & 1u;
        uint32_t w;
        if (KPF != 0 && !rst) w = pf[kk];
        else w = reinterpret_cast<const uint32_t *>(rst ? p.cells0 + env0 * S : p.cells + env0 * S)[u];
        if (rst) reinterpret_cast<uint32_t *>(p.cells + env0 * S)[u] = w; // restore, coalesced
        if (!dst) continue;
        if ((info >> 19) & 1u) { // the cell the transition changed (whether or not the load already saw it)
            const uint32_t x = s_wr[e], idx = x & 0xFFFFu, code = x >> 16, sh = 8u * (idx & 3u);
            if ((int)(idx >> 2) == o) w = (w & ~(0xFFu << sh)) | (code << sh);
        }
        { // agent marker (10, 0, dir): wrappers.py:329-333
            const uint32_t idx = info & 0xFFFFu, code = MGX_K_AGENT | (((info >> 16) & 3u) << 4), sh = 8u * (idx & 3u);
            if ((int)(idx >> 2) == o) w = (w & ~(0xFFu << sh)) | (code << sh);
        }
        const uint32_t t0 = s_lut[w & 255u], t1 = s_lut[(w >> 8) & 255u], t2 = s_lut[(w >> 16) & 255u], t3 = s_lut[w >> 24];
        Out12 r;
        r.a = __builtin_amdgcn_perm(t1, t0, 0x04020100u);
        r.b = __builtin_amdgcn_perm(t2, t1, 0x05040201u);
        r.c = __builtin_amdgcn_perm(t3, t2, 0x06050402u);
        nt_store12(&dst[u].a, r.a, r.b, r.c);
    }
}

// ------------------------------------------------------------------------------------------------
// On-device level generation ("new level each episode").  cells0/agent0 always hold the NEXT episode's level of
// every env; a reset consumes it inside k_step and raises regen[env]; this kernel, launched right after, refills
// the buffer by continuing the env's own numpy-RandomState stream: MT19937 block u32[624] + read index per env in
// HBM (seeded on the host by mgx_reset with gym's legacy seeding).  One wave per 64-env tile scans the flags with a
// ballot; for each flagged env the WAVE regenerates the next MT block cooperatively when the current one is nearly
// used up (the block recurrence is 3 data-parallel phases + 1 word), lane 0 runs the (tiny, sequential) generator
// of levelgen_core.h on LDS, and the wave writes level, record and RNG state back coalesced.
// A 256-thread block owns 2048 envs.  Every thread looks at 8 flags; flagged envs are compacted into an LDS queue (LDS
// atomics).  FAST PATH, one LANE per level (wave 0): the lane copies the next 32 words of its env's MT19937 block into
// its own LDS slice and runs the generator of levelgen_core.h there with small buffers (24 paint commands, 8 rivers per
// axis), paints its level into the slice command by command and stores it.  64 levels advance per wave instruction;
// the first version of this kernel used one WAVE per level with a single active lane, and its ~1,750 instructions
// per level made it cost as much as k_step itself (50 us at 8,400 levels per step).  Levels that do not fit the fast
// path -- grid rows longer than 128 bytes, the read index within 32 words of the end of the block, more than 32 draws,
// too many commands/rivers -- go to a second LDS queue and are generated afterwards by all 4 waves, one level per
// wave at a time (levelgen_one: cooperative next-block build, full-size buffers).
#define MGX_LG_LDS_PER_WAVE (2 * 624 * 4 + 16 + 2 * MGX_LG_WS_WORDS + 8 * MGX_LG_MAX_CMDS)
static_assert(MGX_LG_LDS_PER_WAVE == MGX_LG_LDS_PER_WAVE_BYTES, "keep mgx_kernels.h in sync");
#define MGX_LGF_ENVS 2048
#define MGX_LGF_WIN 32
#define MGX_LGF_CMDS 24
#define MGX_LGF_RIVERS 8
#define MGX_LGF_MAXS 128 /* largest grid row (bytes) painted in a lane slice */
#define MGX_LGF_SLICE_DW (MGX_LGF_WIN + 2 * MGX_LGF_CMDS + 3 * MGX_LGF_RIVERS + MGX_LGF_MAXS / 4 + 1) /* 137 dwords: odd */
__global__ __launch_bounds__(256) void k_levelgen(const LevelGenParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_slices[64 * MGX_LGF_SLICE_DW]; // 35 KB; reused by the slow path
    __shared__ uint16_t s_queue[MGX_LGF_ENVS], s_slow[MGX_LGF_ENVS];
    __shared__ int s_count, s_nslow, s_head;
    static_assert(sizeof(uint32_t) * 64 * MGX_LGF_SLICE_DW >= 4 * MGX_LG_LDS_PER_WAVE, "the slow path reuses the lane slices");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t env_base = (int64_t)blockIdx.x * MGX_LGF_ENVS;
    if (tid == 0) { s_count = 0; s_nslow = 0; s_head = 0; }
    __syncthreads();
    { // scan 8 flags per thread (the regen array is padded to whole tiles and 2048 is a multiple of 64)
        const int64_t e0 = env_base + (int64_t)tid * 8;
        if (e0 < p.n) {
            uint2 *f2 = reinterpret_cast<uint2 *>(p.regen + e0);
            const uint2 a = *f2;
            if (a.x | a.y) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if ((((i < 4 ? a.x : a.y) >> (8 * (i & 3))) & 255u) && e0 + i < p.n) s_queue[atomicAdd(&s_count, 1)] = (uint16_t)(tid * 8 + i);
                *f2 = make_uint2(0, 0);
            }
        }
    }
    __syncthreads();
    const int count = s_count;
    if (count == 0) return;
    const int W = p.cfg.width, H = p.cfg.height, cells = W * H;
    if (wv == 0) {
        uint32_t *slice = s_slices + (size_t)lane * MGX_LGF_SLICE_DW;
        for (int i = lane; i < ((count + 63) & ~63); i += 64) {
            if (i >= count) continue;
            const int64_t env = env_base + s_queue[i];
            const int idx0 = (int)p.mt_idx[env];
            bool ok = idx0 + MGX_LGF_WIN <= 624 && p.S <= MGX_LGF_MAXS;
            if (ok) {
                const uint32_t *mt = p.mt + env * 624 + idx0;
#pragma unroll
                for (int k = 0; k < MGX_LGF_WIN; k++) slice[k] = mt[k];
                WinRng r;
                r.buf = slice - idx0; r.idx = idx0; r.limit = idx0 + MGX_LGF_WIN; r.overflow = false;
                LgLevel L;
                L.cmds = reinterpret_cast<LgCmd *>(slice + MGX_LGF_WIN); L.ncmd = 0; L.max_cmds = MGX_LGF_CMDS;
                L.ws = reinterpret_cast<int16_t *>(slice + MGX_LGF_WIN + 2 * MGX_LGF_CMDS); L.max_rivers = MGX_LGF_RIVERS;
                L.W = W; L.H = H; L.ax = L.ay = -1; L.adir = 0;
                lg_generate(p.cfg, r, L);
                ok = !r.overflow && !L.too_big;
                if (ok) {
                    // paint command by command into the slice, then one pass of dword stores
                    uint32_t *img32 = slice + MGX_LGF_WIN + 2 * MGX_LGF_CMDS + 3 * MGX_LGF_RIVERS;
                    uint8_t *img = reinterpret_cast<uint8_t *>(img32);
                    for (int k = 0; k < (p.S >> 2); k++) img32[k] = 4 * k + 3 < cells ? 0x01010101u * MGX_CODE_EMPTY : 0u;
                    for (int c = cells & ~3; c < cells; c++) img[c] = MGX_CODE_EMPTY;
                    for (int q = 0; q < L.ncmd; q++) {
                        const LgCmd c = L.cmds[q];
                        for (int x = c.x0; x <= c.x1; x++)
                            for (int y = c.y0; y <= c.y1; y++) img[x * H + y] = c.code;
                    }
                    uint32_t *dst = reinterpret_cast<uint32_t *>(p.cells0 + env * p.S);
                    for (int k = 0; k < (p.S >> 2); k++) dst[k] = img32[k];
                    p.mt_idx[env] = (uint32_t)r.idx;
                    p.agent0[env] = make_uint2((uint32_t)((L.ax & 255) | ((L.ay & 255) << 8) | ((L.adir & 3) << 16)) | ((uint32_t)MGX_CODE_EMPTY << 24), L.task << 16);
                }
            }
            if (!ok) s_slow[atomicAdd(&s_nslow, 1)] = s_queue[i];
        }
    }
    __syncthreads();
    const int nslow = s_nslow;
    if (nslow == 0) return;
    uint8_t *base = reinterpret_cast<uint8_t *>(s_slices) + (size_t)wv * MGX_LG_LDS_PER_WAVE;
    for (;;) { // wave-uniform
        int i = 0;
        if (lane == 0) i = atomicAdd(&s_head, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= nslow) break;
        levelgen_one(p, env_base + s_slow[i], base, lane);
    }
}

// reset(): the freshly generated next-level buffer becomes the current episode (for the masked envs) and is flagged
// for regeneration
__global__ __launch_bounds__(256) void k_consume(const ConsumeParams p)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int SD = p.S >> 2;
    const int64_t total = p.n * (int64_t)SD;
    if (t < total) {
        const int64_t e = t / SD;
        if (!p.mask || p.mask[e]) reinterpret_cast<uint32_t *>(p.cells)[t] = reinterpret_cast<const uint32_t *>(p.cells0)[t];
    }
    if (t < p.n && (!p.mask || p.mask[t])) {
        p.agent[t] = p.agent0[t];
        p.regen[t] = p.flag_regen ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// One-hot epilogue: (type, color, state) triples -> NB = 11 + NC + NS bytes per cell with three ones
//   OneHotPartialObsWrapper  wrappers.py:203-243  (NC 7, NS 3 -> 21 channels: out[type] = out[11+color] = out[18+state] = 1)
//   FullyObsOneHotWrapper    wrappers.py:340-415  (NS 4 because the agent cell carries its direction; NC 7 or 0 = drop_color)
// The observation batch is one flat array of cells.  A lane takes 4 consecutive cells (12 input bytes, one
// dwordx3 load) and produces their 4*NB output bytes = NB whole dwords; every output byte is ONE compare because
// its (cell, channel) is known at compile time.  The wave's 64*NB dwords are contiguous in the output, so they are
// transposed through LDS and leave as 16-B/lane coalesced stores.
template <int NC, int NS>
__global__ __launch_bounds__(256) void k_onehot(const uint8_t *__restrict__ tri, uint8_t *__restrict__ out, int64_t n_cells)
{
    constexpr int NB = 11 + NC + NS;
    __shared__ __attribute__((aligned(16))) uint32_t s_x[4][64 * NB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t cell0 = wave * 256; // first cell of this wave
    if (cell0 >= n_cells) return;
    const int64_t c = cell0 + 4 * (int64_t)lane;
    uint32_t ty[4], co[4], st[4];
    if (c + 3 < n_cells) {
        struct __attribute__((packed, aligned(4))) In12 { uint32_t a, b, c; };
        const In12 v = *reinterpret_cast<const In12 *>(tri + c * 3);
        ty[0] = v.a & 255u; co[0] = (v.a >> 8) & 255u; st[0] = (v.a >> 16) & 255u;
        ty[1] = v.a >> 24;  co[1] = v.b & 255u;        st[1] = (v.b >> 8) & 255u;
        ty[2] = (v.b >> 16) & 255u; co[2] = v.b >> 24; st[2] = v.c & 255u;
        ty[3] = (v.c >> 8) & 255u; co[3] = (v.c >> 16) & 255u; st[3] = v.c >> 24;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool ok = c + k < n_cells;
            ty[k] = ok ? tri[(c + k) * 3] : 255u; co[k] = ok ? tri[(c + k) * 3 + 1] : 255u; st[k] = ok ? tri[(c + k) * 3 + 2] : 255u;
        }
    }
    uint32_t *x = s_x[wv] + lane * NB;
#pragma unroll
    for (int d = 0; d < NB; d++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int q = 4 * d + b, k = q / NB, ch = q % NB; // compile-time
            const bool one = ch < 11 ? ty[k] == (uint32_t)ch : (ch < 11 + NC ? co[k] == (uint32_t)(ch - 11) : st[k] == (uint32_t)(ch - 11 - NC));
            w |= (uint32_t)one << (8 * b);
        }
        x[d] = w;
    }
    wave_sync();
    const int64_t wave_cells = n_cells - cell0 < 256 ? n_cells - cell0 : 256;
    const int n_bytes = (int)wave_cells * NB;
    uint8_t *dst = out + cell0 * NB; // 256*NB bytes per wave: 16-B aligned
    const uint4 *x4 = reinterpret_cast<const uint4 *>(s_x[wv]);
    for (int i = lane; i < (n_bytes + 15) / 16; i += 64) {
        if (16 * i + 16 <= n_bytes) nt_store16(reinterpret_cast<uint4 *>(dst) + i, x4[i]);
        else
            for (int b = 16 * i; b < n_bytes; b++) dst[b] = reinterpret_cast<const uint8_t *>(s_x[wv])[b];
    }
}

// ------------------------------------------------------------------------------------------------
// env.seed(s) on the device, lane-per-env: gym's legacy key derivation (SHA-512 of str(seed), levelgen_core.h) and
// MT19937 init_by_array written into the env's state block in HBM.  The recurrences are sequential per env (each
// word depends on the previous one); 64 envs advance in lock-step per wave.  Pass 3 re-reads what pass 2 stored, 16
// words at a time so that the loads are in flight together.
__global__ __launch_bounds__(256) void k_seed(const uint64_t *__restrict__ seeds, const uint8_t *__restrict__ mask,
                                              const uint32_t *__restrict__ init, uint32_t *mt, uint32_t *mt_idx,
                                              uint8_t *regen, int64_t n)
{
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n || (mask && !mask[env])) return;
    uint32_t key[2];
    const int klen = lg_seed_key(seeds[env], key);
    uint32_t *m = mt + env * 624;
    // pass 2: mt[i] = (init[i] ^ ((mt[i-1] ^ mt[i-1] >> 30) * 1664525)) + key[j] + j for i = 1..623, then once more for i = 1
    uint32_t prev = init[0], v1 = 0;
    int j = 0;
    for (int i = 1; i < 624; i++) {
        const uint32_t x = (init[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        if (i == 1) v1 = x; else m[i] = x;
        prev = x;
        j = (j + 1 == klen) ? 0 : j + 1;
    }
    const uint32_t m1 = (v1 ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j; // mt[0] = mt[623]; i = 1 again
    prev = m1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // this lane re-reads its own stores below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // pass 3: mt[i] = (mt[i] ^ ((mt[i-1] ^ mt[i-1] >> 30) * 1566083941)) - i for i = 2..623, then for i = 1
    for (int i0 = 2; i0 < 624; i0 += 16) {
        uint32_t old[16];
#pragma unroll
        for (int k = 0; k < 16; k++) old[k] = (i0 + k < 624) ? m[i0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (i0 + k < 624) {
                const uint32_t x = (old[k] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)(i0 + k);
                m[i0 + k] = x;
                prev = x;
            }
        }
    }
    m[1] = (m1 ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u; // mt[0] = mt[623]; i = 1
    m[0] = 0x80000000u;
    mt_idx[env] = 624; // nothing drawn yet
    regen[env] = 1;
}

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
                const uint8_t *tr = p.grid + (e * cells + c) * 3;
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
                if (p.objaux) { p.objaux[t] = (uint8_t)(ax & 0xFEu); p.objaux0[t] = (uint8_t)(ax & 0xFEu); p.objcont[t] = MGX_CODE_EMPTY; p.objcont0[t] = MGX_CODE_EMPTY; }
            } else if (p.objaux) { p.objaux[t] = 0; p.objaux0[t] = 0; p.objcont[t] = MGX_CODE_EMPTY; p.objcont0[t] = MGX_CODE_EMPTY; }
            p.cells[t] = (uint8_t)code;
            p.cells0[t] = (uint8_t)code;
        }
    }
    if (t < p.n && (!p.mask || p.mask[t])) {
        const int32_t x = p.agent[t * 3], y = p.agent[t * 3 + 1], d = p.agent[t * 3 + 2];
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
        p.rec0[t] = make_uint2((rec.x & 0x00FFFFFFu) | ((uint32_t)MGX_CODE_EMPTY << 24), p.has_task ? (w1 & 0xFFFF0000u) : 0u);
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

template <int CW, int CH>
hipError_t launch_sized(const StepParams &p, int mode, dim3 grid, dim3 block, size_t shmem, hipStream_t st)
{
    if (mode == 0) hipLaunchKernelGGL((k_step<CW, CH, 0, 7>), grid, block, shmem, st, p);
    else if (mode == 1) hipLaunchKernelGGL((k_step<CW, CH, 1, 7>), grid, block, shmem, st, p);
    else hipLaunchKernelGGL((k_step_fulldirect<CW, CH>), dim3(p.n_tiles), dim3(256), 0, st, p);
    return hipGetLastError();
}

template <int CW, int CH>
hipError_t raise_lds_limit(int mode, int bytes)
{
    if (mode == 0) return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step<CW, CH, 0, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step<CW, CH, 1, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

} // namespace

#define MGX_SIZED(X) X(5, 5) X(6, 6) X(7, 7) X(8, 8) X(9, 9) X(11, 11) X(16, 16)
#define MGX_VIEWS(X) X(3) X(5) X(9) X(11)  /* agent_view_size other than 7: run-time grid size only */

hipError_t mgx_launch_step(const StepParams &p, int mode, int waves_per_block, hipStream_t st)
{
    const dim3 block(64 * waves_per_block);
    const dim3 grid((p.n_tiles + waves_per_block - 1) / waves_per_block);
    const size_t shmem = (size_t)waves_per_block * p.wave_lds;
    if (mode == 0 && p.alt_vis) { // default_vis=False: run-time grid size only
#define VCASE(v) if (p.view == v) { hipLaunchKernelGGL((k_step<0, 0, 0, v, true>), grid, block, shmem, st, p); return hipGetLastError(); }
        MGX_VIEWS(VCASE) VCASE(7)
#undef VCASE
        return hipErrorInvalidValue;
    }
    if (mode == 0 && p.view != 7) {
#define VCASE(v) if (p.view == v) { hipLaunchKernelGGL((k_step<0, 0, 0, v>), grid, block, shmem, st, p); return hipGetLastError(); }
        MGX_VIEWS(VCASE)
#undef VCASE
        return hipErrorInvalidValue;
    }
    if (p.objaux) return launch_sized<0, 0>(p, mode, grid, block, shmem, st);
#define CASE(w, h) if (p.W == w && p.H == h) return launch_sized<w, h>(p, mode, grid, block, shmem, st);
    MGX_SIZED(CASE)
#undef CASE
    return launch_sized<0, 0>(p, mode, grid, block, shmem, st);
}

hipError_t mgx_raise_lds_limit(int W, int H, int mode, int bytes, int view, int alt_vis, int object_state)
{
    if (object_state) W = H = 0;
    if (mode == 0 && alt_vis) {
#define VCASE(v) if (view == v) return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step<0, 0, 0, v, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        MGX_VIEWS(VCASE) VCASE(7)
#undef VCASE
        return hipErrorInvalidValue;
    }
    if (mode == 0 && view != 7) {
#define VCASE(v) if (view == v) return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step<0, 0, 0, v>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        MGX_VIEWS(VCASE)
#undef VCASE
        return hipErrorInvalidValue;
    }
#define CASE(w, h) if (W == w && H == h) return raise_lds_limit<w, h>(mode, bytes);
    MGX_SIZED(CASE)
#undef CASE
    return raise_lds_limit<0, 0>(mode, bytes);
}

hipError_t mgx_launch_levelgen(const LevelGenParams &p, hipStream_t st)
{
    hipLaunchKernelGGL(k_levelgen, dim3((unsigned)((p.n + MGX_LGF_ENVS - 1) / MGX_LGF_ENVS)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t mgx_launch_seed(const uint64_t *seeds, const uint8_t *mask, const uint32_t *init, uint32_t *mt, uint32_t *mt_idx,
                           uint8_t *regen, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seeds, mask, init, mt, mt_idx, regen, n);
    return hipGetLastError();
}

hipError_t mgx_launch_consume(const ConsumeParams &p, hipStream_t st)
{
    const int64_t total = p.n * (int64_t)(p.S >> 2);
    hipLaunchKernelGGL(k_consume, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t mgx_launch_onehot(const uint8_t *tri, uint8_t *out, int64_t n_cells, int nc, int ns, hipStream_t st)
{
    const dim3 grid((unsigned)((n_cells + 1023) / 1024)), block(256);
    if (nc == 7 && ns == 3) hipLaunchKernelGGL((k_onehot<7, 3>), grid, block, 0, st, tri, out, n_cells);
    else if (nc == 7 && ns == 4) hipLaunchKernelGGL((k_onehot<7, 4>), grid, block, 0, st, tri, out, n_cells);
    else if (nc == 0 && ns == 4) hipLaunchKernelGGL((k_onehot<0, 4>), grid, block, 0, st, tri, out, n_cells);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Dynamic-Obstacles (envs/dynamicobstacles.py:60-89).  The obstacle walk draws from the env's own MT19937 stream
// inside step(), so it cannot live in the streaming step kernel: k_dynobs runs before it, one lane per env.
//   RNG   : the per-env block `mt` is the one k_seed/k_levelgen left behind (words [pos, 624) not drawn yet).  Past
//           the block the next words are produced ONE AT A TIME in place -- new[k] = f(old[k], old[k+1], old[k+397] or
//           new[k-227]) is exactly the order genrand's bulk twist uses, so the stream is numpy's -- which costs three
//           loads and a store per draw instead of a 2.5 KB twist per lane.
//   reset : the in-kernel auto-reset of the step kernels raises regen[env]; the walk then first restores the obstacle
//           order and the RNG position of the episode start (ReseedWrapper: seed(s) + reset()), and the block itself
//           only if the episode ran past it (pos > 624), which random-action episodes (~6 steps) never do.
namespace {
__global__ __launch_bounds__(256) void k_dynobs_init(const DynObsParams p)
{
    // snapshot of the RNG block (coalesced: 156 uint4 per env)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < p.n * 156) {
        const int64_t e = t / 156;
        if (!p.mask || p.mask[e]) reinterpret_cast<uint4 *>(p.mt0)[t] = reinterpret_cast<const uint4 *>(p.mt)[t];
    }
    if (t >= p.n || (p.mask && !p.mask[t])) return;
    // obstacle order from the generator's marker codes; the cells become plain blue balls
    uint8_t ob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int cells = p.W * p.H;
    for (int c = 0; c < cells; c++) {
        const uint32_t code = p.cells0[t * p.S + c];
        if (MGX_IS_OBSTACLE_MARK(code)) {
            ob[(code >> 4) & 7u] = (uint8_t)(((c / p.H) << 4) | (c % p.H));
            p.cells0[t * p.S + c] = (uint8_t)MGX_CODE_BALL_BLUE;
            p.cells[t * p.S + c] = (uint8_t)MGX_CODE_BALL_BLUE;
        }
    }
    uint2 w;
    w.x = ob[0] | (ob[1] << 8) | (ob[2] << 16) | ((uint32_t)ob[3] << 24);
    w.y = ob[4] | (ob[5] << 8) | (ob[6] << 16) | ((uint32_t)ob[7] << 24);
    reinterpret_cast<uint2 *>(p.obst0)[t] = w;
    reinterpret_cast<uint2 *>(p.obst)[t] = w;
    p.pos0[t] = p.pos[t];
    p.regen[t] = 0;
}

// One wave per tile of 64 envs, lane per env, like k_step.  The walk is a chain of draw -> look at a cell -> maybe draw
// again, every link depending on the one before and diverging between lanes; taken straight from HBM each link costs a
// memory round trip for the whole wave (measured 0.55 - 1.9 ms per step at 1 Mi 8x8 envs).  So the wave first brings
// what the chain will touch into LDS with coalesced loads -- the tile's cells (as k_step stages them) and, per env, the
// next MGX_DYN_WIN words of its RNG block (two envs per 256-B load) -- and the chain then runs on LDS.
// RNG bookkeeping: `pos` counts the words drawn since the block in memory was complete (bit 31: the block is no longer
// the episode-start block).  A lane that draws past its window reads the global words, and past the block it produces
// the next block's words one at a time in place (new[k] from old[k], old[k+1], old[k+397] or new[k-227]: the order of
// genrand's bulk twist, so the stream is numpy's).  At the start of the next step the whole wave finishes such a
// half-regenerated block (words k..623, in LDS, in chunks of <= 227 independent words) so that the env is back on the
// window path; the in-kernel auto-reset of the step kernels raises regen[env], upon which the wave restores the
// obstacle order, the RNG position and -- if it was touched -- the block from the episode-start snapshot
// (ReseedWrapper: seed(s) + reset()).
#define MGX_DYN_WIN 48
#define MGX_DYN_WSTRIDE 52 /* bytes per lane: 13 dwords, odd */
// Every draw of the walk is `bounded(2)` (a 3-wide range: obstacles live in the interior, so the 3x3 box never clips):
// only the low two bits of the tempered word matter.  The window therefore keeps ONE BYTE per word (tempered at fill
// time): 3 KB per wave instead of 12, which is what bounds the occupancy of this latency-bound kernel.
struct DynRng {
    const uint8_t *win; // LDS: low byte of the tempered words [p0, lim) of the block
    uint32_t *A;
    uint32_t p, p0, lim;
    __device__ __forceinline__ uint32_t next8()
    {
        if (p < lim) return win[p++ - p0];
        uint32_t y;
        if (p < 624u) y = A[p];
        else {
            const uint32_t k = p % 624u, k1 = k + 1u == 624u ? 0u : k + 1u, km = k + 397u >= 624u ? k + 397u - 624u : k + 397u;
            y = lg_twist_word(A[k], A[k1], A[km]);
            A[k] = y;
        }
        p++;
        return lg_temper(y) & 255u;
    }
    __device__ __forceinline__ int draw3() // _rand_int(t, t + 3) - t: masked rejection on two bits
    {
        uint32_t v;
        do { v = next8() & 3u; } while (v > 2u);
        return (int)v;
    }
};

__global__ __launch_bounds__(256) void k_dynobs(const DynObsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= p.n_tiles) return; // wave-uniform
    const int W = p.W, H = p.H, S = p.S, LS = p.LS;
    uint8_t *lds = smem + (size_t)wv * p.wave_lds;
    const int cells_bytes = 64 * LS > 2496 ? ((64 * LS + 15) & ~15) : 2496;
    uint32_t *blk = reinterpret_cast<uint32_t *>(lds); // 624 words: a block being restored / finished (before the cells arrive)
    uint8_t *win = lds + cells_bytes;
    uint32_t *ps = reinterpret_cast<uint32_t *>(win + 64 * MGX_DYN_WSTRIDE);
    const int64_t env0 = (int64_t)tile * 64, env = env0 + lane;
    const bool valid = env < p.n;

    uint2 ow = reinterpret_cast<const uint2 *>(p.obst)[env]; // (all per-env arrays are padded to whole tiles)
    uint32_t pos = p.pos[env];
    const bool regen = valid && p.regen[env];
    uint32_t a = valid ? p.actions[env] : 0u;
    const uint32_t rec = p.agent[env].x;
    bool dirty = (pos >> 31) != 0u;
    pos &= 0x7FFFFFFFu;
    const bool need_restore = regen && dirty, need_finish = valid && !regen && pos >= 624u;
    if (regen) { // the previous step ended the episode: cells/agent are already the episode start
        ow = reinterpret_cast<const uint2 *>(p.obst0)[env];
        pos = p.pos0[env];
        dirty = false;
        p.regen[env] = 0;
    }
    ps[lane] = valid ? pos : 0xFFFFFFFFu;
    wave_sync();

    const unsigned long long m_restore = __ballot(need_restore);
    unsigned long long m_service = m_restore | __ballot(need_finish);
    const bool any_service = m_service != 0ull;
    while (m_service) { // wave-uniform: one env at a time, all 64 lanes on its block
        const int e = __builtin_ctzll(m_service);
        m_service &= m_service - 1;
        uint4 *dst4 = reinterpret_cast<uint4 *>(p.mt) + (env0 + e) * 156;
        uint4 *blk4 = reinterpret_cast<uint4 *>(blk);
        uint32_t pe;
        if ((m_restore >> e) & 1ull) {
            const uint4 *src4 = reinterpret_cast<const uint4 *>(p.mt0) + (env0 + e) * 156;
            for (int i = lane; i < 156; i += 64) { const uint4 v = src4[i]; dst4[i] = v; blk4[i] = v; }
            pe = ps[e];
        } else {
            for (int i = lane; i < 156; i += 64) blk4[i] = dst4[i];
            wave_sync();
            const uint32_t k0 = ps[e] % 624u; // words [0, k0) already belong to the new block
            pe = k0;
            uint32_t c0 = k0;
            while (c0 < 623u) { // chunks of <= 227 words: within one, nobody needs a word the chunk itself produces
                const uint32_t c1 = c0 + 227u < 623u ? c0 + 227u : 623u;
                uint32_t y[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t jj = c0 + (uint32_t)lane + 64u * r;
                    y[r] = 0;
                    if (jj < c1) y[r] = lg_twist_word(blk[jj], blk[jj + 1u], jj < 227u ? blk[jj + 397u] : blk[jj - 227u]);
                }
                wave_sync();
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t jj = c0 + (uint32_t)lane + 64u * r;
                    if (jj < c1) blk[jj] = y[r];
                }
                wave_sync();
                c0 = c1;
            }
            if (lane == 0) blk[623] = lg_twist_word(blk[623], blk[0], blk[396]);
            wave_sync();
            for (int i = lane; i < 156; i += 64) dst4[i] = blk4[i];
        }
        wave_sync();
        if (lane < MGX_DYN_WIN) win[e * MGX_DYN_WSTRIDE + lane] = (uint8_t)(pe + (uint32_t)lane < 624u ? lg_temper(blk[pe + lane]) : 0u);
        if (lane == 0) ps[e] = pe | 0x40000000u; // window already filled (from LDS: the global words were just written)
        wave_sync();
    }
    if (need_finish) { pos %= 624u; dirty = true; }
    if (any_service) { // blocks rewritten by the whole wave may be read word-wise by single lanes below: same CU, same
                       // L1, so the stores only have to be complete (an agent-scope fence would write back the XCD's L2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    { // RNG windows: quad q = it*64 + lane of the tile's 64 x 12 quads -> each env's 48 words are 12 consecutive
      // (4-byte aligned) dwordx4 loads; every load is issued before the first is consumed
        struct __attribute__((packed, aligned(4))) Q { uint32_t a, b, c, d; };
        Q v[MGX_DYN_WIN / 4];
#pragma unroll
        for (int it = 0; it < MGX_DYN_WIN / 4; it++) {
            const int q = it * 64 + lane, e = q / (MGX_DYN_WIN / 4), j = q - e * (MGX_DYN_WIN / 4);
            const uint32_t pe = ps[e];
            v[it] = Q{0u, 0u, 0u, 0u};
            if (pe < 624u && pe + 4u * j + 4u <= 624u) v[it] = *reinterpret_cast<const Q *>(p.mt + (env0 + e) * 624 + pe + 4 * j);
            else if (pe < 624u) { // the block ends inside this quad
                const uint32_t *src = p.mt + (env0 + e) * 624;
                const uint32_t b = pe + 4u * j;
                if (b < 624u) v[it].a = src[b];
                if (b + 1u < 624u) v[it].b = src[b + 1u];
                if (b + 2u < 624u) v[it].c = src[b + 2u];
            }
        }
        stage_tile<0>(p.cells, env0, S, LS, lds, lane);
#pragma unroll
        for (int it = 0; it < MGX_DYN_WIN / 4; it++) {
            const int q = it * 64 + lane, e = q / (MGX_DYN_WIN / 4), j = q - e * (MGX_DYN_WIN / 4);
            if (ps[e] < 624u)
                *reinterpret_cast<uint32_t *>(win + e * MGX_DYN_WSTRIDE + 4 * j) =
                    (lg_temper(v[it].a) & 255u) | ((lg_temper(v[it].b) & 255u) << 8) | ((lg_temper(v[it].c) & 255u) << 16) | (lg_temper(v[it].d) << 24);
        }
    }
    wave_sync();
    if (!valid) return;

    uint8_t *g = lds + lane * LS;
    uint8_t *gg = p.cells + env * S;
    if (a >= 3u) a = 0u; // `if action >= self.action_space.n: action = 0`
    const int ax = (int)(rec & 255u), ay = (int)((rec >> 8) & 255u), dir = (int)((rec >> 16) & 3u);
    const int fx = ax + (dir == 0) - (dir == 2), fy = ay + (dir == 1) - (dir == 3);
    bool not_clear = false; // front_cell and front_cell.type != 'goal', BEFORE the obstacles move
    if (fx >= 0 && fx < W && fy >= 0 && fy < H) {
        const uint32_t k = g[fx * H + fy] & 15u;
        not_clear = k != MGX_K_EMPTY && k != MGX_K_GOAL;
    }
    const uint32_t have = pos < 624u ? ((624u - pos) < MGX_DYN_WIN ? 624u - pos : (uint32_t)MGX_DYN_WIN) : 0u;
    DynRng r = {win + lane * MGX_DYN_WSTRIDE, p.mt + env * 624, pos, pos, pos + have};
    for (int i = 0; i < p.n_obst; i++) {
        const uint32_t o = (i < 4 ? ow.x >> (8 * i) : ow.y >> (8 * (i - 4))) & 255u; // x << 4 | y
        const int tx = (int)(o >> 4) - 1, ty = (int)(o & 15u) - 1; // top = old_pos + (-1, -1): interior, never clipped
        int nx = -1, ny = -1;
        for (int tries = 0; tries <= 100; tries++) { // num_tries > max_tries raises: 101 samples at most
            const int x = tx + r.draw3(), y = ty + r.draw3();
            if (g[x * H + y] != MGX_CODE_EMPTY) continue;
            if (x == ax && y == ay) continue;
            nx = x; ny = y;
            break;
        }
        if (nx < 0) continue; // RecursionError swallowed by the bare except: the obstacle stays
        const int n8 = nx * H + ny, o8 = (tx + 1) * H + ty + 1;
        g[n8] = (uint8_t)MGX_CODE_BALL_BLUE; gg[n8] = (uint8_t)MGX_CODE_BALL_BLUE;
        g[o8] = (uint8_t)MGX_CODE_EMPTY; gg[o8] = (uint8_t)MGX_CODE_EMPTY;
        const uint32_t nb = ((uint32_t)nx << 4) | (uint32_t)ny;
        if (i < 4) ow.x = (ow.x & ~(255u << (8 * i))) | (nb << (8 * i));
        else ow.y = (ow.y & ~(255u << (8 * (i - 4)))) | (nb << (8 * (i - 4)));
    }
    reinterpret_cast<uint2 *>(p.obst)[env] = ow;
    p.pos[env] = r.p | ((dirty || r.p > 624u) ? 0x80000000u : 0u);
    p.act_out[env] = (uint8_t)(a | ((a == 2u && not_clear) ? 0x80u : 0u));
}
} // namespace

hipError_t mgx_launch_dynobs_init(const DynObsParams &p, hipStream_t st)
{
    const int64_t total = p.n * 156;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dynobs_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

int mgx_dynobs_wave_lds(int LS) { return (64 * LS > 2496 ? ((64 * LS + 15) & ~15) : 2496) + 64 * MGX_DYN_WSTRIDE + 64 * 4; }

hipError_t mgx_launch_dynobs(const DynObsParams &p, hipStream_t st)
{
    if (p.n_tiles == 0) return hipSuccess;
    if (p.wave_lds > 65536) return hipErrorInvalidValue;
    const int wpb = 1; // one wave per block: the LDS footprint, not the wave slots, bounds the occupancy
    hipLaunchKernelGGL(k_dynobs, dim3((unsigned)((p.n_tiles + wpb - 1) / wpb)), dim3(64 * wpb), (size_t)wpb * p.wave_lds, st, p);
    return hipGetLastError();
}

// FlatObsWrapper.observation (wrappers.py:556-577): out[env] = f32(image bytes) ++ one-hot of the mission string
// (96 positions x 27 codes).  `table` holds one row of 96 character codes (0..25 letters, 26 space, 255 past the end)
// per mission of the family; only Fetch has more than one (row = (template*2 + is_ball)*8 + color of the task word).
// A pure stream of 16-B stores over the flat [n][L] output: 4 consecutive floats per lane, one divide per lane.
namespace {
__global__ __launch_bounds__(256) void k_flat(const uint8_t *__restrict__ tri, const uint2 *__restrict__ rec, const uint8_t *__restrict__ table,
                                               float *__restrict__ out, int64_t n, int img, int fetch)
{
    const int64_t L = (int64_t)img + MGX_FLAT_MISSION;
    const int64_t total = n * L;
    const int64_t g0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (g0 >= total) return;
    int64_t env = g0 / L;
    int off = (int)(g0 - env * L);
    float v[4];
    const uint8_t *row = nullptr;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float x = 0.f;
        if (g0 + j < total) {
            if (off >= L) { off = 0; env++; row = nullptr; }
            if (off < img) x = (float)tri[env * img + off];
            else {
                if (!row) {
                    int mid = 0;
                    if (fetch) {
                        const uint32_t task = rec[env].y >> 16;
                        mid = (int)((((task >> 8) & 7u) * 2u + ((task & 15u) == MGX_K_BALL ? 1u : 0u)) * 8u + ((task >> 4) & 7u));
                    }
                    row = table + mid * 96;
                }
                const int k = off - img, ch = k / 27, code = k - ch * 27;
                x = row[ch] == code ? 1.f : 0.f;
            }
            off++;
        }
        v[j] = x;
    }
    if (g0 + 3 < total) nt_store16(reinterpret_cast<uint4 *>(out + g0), make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    else
        for (int j = 0; j < 4 && g0 + j < total; j++) out[g0 + j] = v[j];
}
} // namespace

hipError_t mgx_launch_flat(const uint8_t *tri, const uint2 *rec, const uint8_t *table, float *out, int64_t n, int img, int fetch, hipStream_t st)
{
    const int64_t quads = (n * ((int64_t)img + MGX_FLAT_MISSION) + 3) / 4;
    if (quads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_flat, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, tri, rec, table, out, n, img, fetch);
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void k_direction(const uint2 *__restrict__ rec, uint8_t *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)((rec[i].x >> 16) & 3u);
}
} // namespace

namespace {
__global__ __launch_bounds__(256) void k_task(uint2 *rec, uint2 *rec0, const uint32_t *set, uint32_t *get, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (set) {
        rec[i].y = (rec[i].y & 0xFFFFu) | (set[i] << 16);
        rec0[i].y = (rec0[i].y & 0xFFFFu) | (set[i] << 16);
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
            p.objcont0[e * p.S + c] = code;
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

hipError_t mgx_launch_task(uint2 *rec, uint2 *rec0, const uint32_t *set, uint32_t *get, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_task, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, rec0, set, get, n);
    return hipGetLastError();
}

hipError_t mgx_launch_direction(const uint2 *rec, uint8_t *out, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_direction, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, out, n);
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
