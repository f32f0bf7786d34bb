// k_step.hip -- the hot path of libmgx.so for gfx950 (MI355X / CDNA4): k_step, k_step_fulldirect.
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
#include <stdio.h>
#include <stdlib.h>
#ifdef MGX_TIMELINE
#include <vector>
#endif

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include <type_traits>
#include "mgx_device.h"
#include "dynobs_device.h"

namespace {

// Task rules that env subclasses layer on MiniGridEnv.step (they run after the base step, time-out included).
// (RewardT: float -- the product kernels' reward, each rule's Python double rounded where it is assigned, exactly once -- or double: the
// run-time-size instances, which also serve handles with exploration bonuses and keep the reference's doubles up to the store)
template <int CH, class CellAt, typename RewardT>
__device__ __forceinline__ void task_rule(const StepParams &p, Lane &L, uint32_t act, RewardT &reward, bool &done, CellAt cell_at,
                                          int fidx, uint32_t fc, uint32_t carry0, bool &fault)
{
    if (p.task == MGX_TASK_FETCH) { // envs/fetch.py:74-86
        if (L.carry != MGX_CODE_EMPTY) {
            done = true;
            reward = ((L.carry & 0x7Fu) == (L.task & 0x7Fu)) ? (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)) : (RewardT)0;
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
                if (door && ((n4[i] >> 4) & 7u) == 0u) reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps));
            }
        }
    } else if (p.task == MGX_TASK_REDBLUEDOORS) { // envs/redbluedoors.py:44-66; cell_at() is the state AFTER the step
        const int H = CH ? CH : p.H;
        const int ri = (H / 2) * H + (int)(L.task & 15u), bi = (H / 2 + H - 1) * H + (int)((L.task >> 4) & 15u);
        const bool red_after = (cell_at(ri) & 15u) == MGX_K_DOOR_OPEN, blue_after = (cell_at(bi) & 15u) == MGX_K_DOOR_OPEN;
        const bool red_before = ((ri == fidx ? fc : cell_at(ri)) & 15u) == MGX_K_DOOR_OPEN;
        const bool blue_before = ((bi == fidx ? fc : cell_at(bi)) & 15u) == MGX_K_DOOR_OPEN;
        if (blue_after) { reward = red_before ? (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)) : (RewardT)0; done = true; }
        else if (red_after && blue_before) { reward = 0.f; done = true; }
    } else if (p.task == MGX_TASK_TWOGOALS) { // envs/twogoals.py:118-146 on top of the base transition; L.task = goal_count, fc = the
                                              // front cell BEFORE the step (a toggled goal is gone afterwards)
        if (act == 5) {
            if (fc == MGX_CODE_EMPTY) fault = true; // `fwd_cell.type` on None: AttributeError
            else if ((fc & 15u) == MGX_K_GOAL) {
                const uint32_t color = (fc >> 4) & 7u;
                reward = color == 1u ? 0.25f : (color == 4u ? 0.5f : 0.f);
                L.task += 1u;
            }
        }
        if (act == 6) done = true;
        if (L.task >= 2u) { // reward += 1. - 0.9 * self.step_count/self.max_steps   (NOT _reward(): the product comes first)
            reward = (RewardT)((double)reward + (1.0 - (0.9 * (double)L.steps) / (double)p.max_steps));
            done = true;
        }
    } else if (p.task == MGX_TASK_PUTNEAR) { // envs/putnear.py:91-110; carry0 = preCarrying
        const uint32_t move = (uint32_t)(MGX_K_KEY + (L.task & 3u)) | (((L.task >> 2) & 7u) << 4);
        if (act == 3 && L.carry != MGX_CODE_EMPTY && (L.carry & 0x7Fu) != move) done = true; // picked up (or holds) the wrong object
        if (act == 4 && carry0 != MGX_CODE_EMPTY) {
            if (L.carry == MGX_CODE_EMPTY) { // the drop happened: the object now lies in the front cell
                const int ox = L.ax + (L.dir == 0) - (L.dir == 2), oy = L.ay + (L.dir == 1) - (L.dir == 3);
                const int dx = ox - (int)((L.task >> 5) & 7u), dy = oy - (int)((L.task >> 8) & 7u);
                if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps));
            }
            done = true;
        }
    } else if (p.task == MGX_TASK_UNLOCK) { // envs/unlock.py:33-41: the door is at (5, task)
        const int H = CH ? CH : p.H;
        if (act == 5 && (cell_at(5 * H + (int)(L.task & 15u)) & 15u) == MGX_K_DOOR_OPEN) { reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)); done = true; }
    } else if (p.task == MGX_TASK_PICKUPBOX) { // envs/unlockpickup.py:35-43, keycorridor.py:51-59: `self.carrying == self.obj`
        if (act == 3 && (L.carry & 0x7Fu) == (L.task & 0x7Fu)) { reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)); done = true; }
    } else if (p.task == MGX_TASK_MEMORY) { // envs/memory.py:92-99 (the pickup -> toggle remap happens where the action is loaded)
        const int H = CH ? CH : p.H;
        const int tx = (int)(L.task & 15u), sy = ((L.task >> 4) & 1u) ? H / 2 - 1 : H / 2 + 1, fy = ((L.task >> 4) & 1u) ? H / 2 + 1 : H / 2 - 1;
        if (L.ax == tx && L.ay == sy) { reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)); done = true; }
        if (L.ax == tx && L.ay == fy) { reward = 0.f; done = true; }
    } else if (p.task == MGX_TASK_GOTOOBJECT) { // envs/gotoobject.py:68-84
        if (act == 5) done = true;              // "Toggle/pickup action terminates the episode"
        if (act == 6) {
            const int dx = L.ax - (int)(L.task & 15u), dy = L.ay - (int)((L.task >> 4) & 15u);
            if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps));
            done = true;
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
    bad_act = valid && (act >= (p.extended ? 9u : MGX_NUM_ACTIONS_K) || (p.task == MGX_TASK_TWOGOALS && (act == 3u || act == 4u)));
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

template <int CH, class CellAt, typename RewardT>
__device__ __forceinline__ uint32_t transition_apply(const StepParams &p, Lane &L, uint32_t act, uint32_t fc, RewardT &reward, bool &done,
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
            if (ov) { done = true; reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps)); }
        }
        if (k == MGX_K_LAVA) done = true; // no 'v1' special case on the strafe path (minigrid.py:1304-1305,1313-1314)
        return fc;
    }
    // The action switch in straight-line form: a wave of 64 independent agents takes every branch of an `if (act == ...)` chain on
    // every step, so the chain cost the sum of its arms plus an exec-mask region each (0.7 us of a lone wave's 5.5 us: wave
    // timelines, profiles/).  Each outcome is a predicate; the few results are selected at the end.  Only the rare arms that need
    // memory or f64 (hidden object state, the goal's reward) stay branches.
    const int dx = (dir == 0) - (dir == 2), dy = (dir == 1) - (dir == 3);
    const uint32_t k = fc & 15u;
    const uint32_t carry = L.carry;
    L.dir = act == 0 ? (dir + 3) & 3 : (act == 1 ? (dir + 1) & 3 : dir);
    // act 2: None, Floor, open Door, Goal, Lava can be walked onto (minigrid.py:93,164-166,192,211,245-247)
    const uint32_t OVERLAP = (1u << MGX_K_EMPTY) | (1u << MGX_K_FLOOR) | (1u << MGX_K_DOOR_OPEN) | (1u << MGX_K_GOAL) | (1u << MGX_K_LAVA);
    const bool fwd = act == 2;
    bool ovl = ((OVERLAP >> k) & 1u) != 0u;
    if (o.aux && fwd && k == MGX_K_BOX) ovl = box_overlappable(o, tidx, fc);
    const bool move = fwd && ovl;
    L.ax += move ? dx : 0;
    L.ay += move ? dy : 0;
    if (fwd && k == MGX_K_GOAL && (fc & 0x80u)) { // goal.overlap (minigrid.py:1259-1261)
        done = true;
        // _reward(): 1 - 0.9*(step_count/max_steps) in Python doubles (minigrid.py:933-937), then f32
        reward = (RewardT)(1.0 - 0.9 * ((double)L.steps / (double)p.max_steps));
    }
    const bool lava = fwd && k == MGX_K_LAVA; // minigrid.py:1262-1268 ('v1' classes: reward -1, no done)
    done = lava ? !p.lava_v1 : done;
    reward = (lava && p.lava_v1) ? (RewardT)-1 : reward;
    // act 3 / 4: pickup, drop
    const uint32_t PICK = (1u << MGX_K_KEY) | (1u << MGX_K_BALL) | (1u << MGX_K_BOX);
    const bool pick = act == 3 && ((PICK >> k) & 1u) && carry == MGX_CODE_EMPTY;
    const bool drop = act == 4 && k == MGX_K_EMPTY && carry != MGX_CODE_EMPTY;
    if (o.aux) {
        // (every write to the planes marks the env dirty: an auto-reset copies the planes' snapshot back only then, restore_objstate)
        if (pick) { *o.carry = (uint16_t)(o.aux[tidx] | (o.cont[tidx] << 8)); o.aux[tidx] = 0; o.cont[tidx] = MGX_CODE_EMPTY; L.dirty = MGX_REC_DIRTY; }
        if (drop) { o.aux[tidx] = (uint8_t)*o.carry; o.cont[tidx] = (uint8_t)(*o.carry >> 8); *o.carry = (uint16_t)(MGX_CODE_EMPTY << 8); L.dirty = MGX_REC_DIRTY; }
    }
    // act 5: toggle.  Door.toggle (minigrid.py:252-262): locked opens with a key of its colour, open <-> closed; Goal.toggle with
    // toggletimes=1 (minigrid.py:171-181) removes a goal that is not an `overlap` one; Box.toggle with contains=None (minigrid.py:355-364)
    const bool keyfits = (carry & 15u) == MGX_K_KEY && ((carry >> 4) & 7u) == ((fc >> 4) & 7u);
    const uint32_t opened = (fc & 0xF0u) | MGX_K_DOOR_OPEN, closed = (fc & 0xF0u) | MGX_K_DOOR_CLOSED;
    uint32_t tog = fc;
    tog = k == MGX_K_DOOR_LOCKED ? (keyfits ? opened : fc) : tog;
    tog = k == MGX_K_DOOR_OPEN ? closed : tog;
    tog = k == MGX_K_DOOR_CLOSED ? opened : tog;
    tog = (k == MGX_K_GOAL && !(fc & 0x80u)) ? (uint32_t)MGX_CODE_EMPTY : tog;
    tog = k == MGX_K_BOX ? (uint32_t)MGX_CODE_EMPTY : tog;
    if (act == 5 && o.aux && (k == MGX_K_GOAL || k == MGX_K_BOX)) {
        // Goal.toggle / Box.toggle with their hidden state (minigrid.py:171-181,355-364)
        tog = fc;
        uint32_t a = o.aux[tidx];
        int tt = (int)(((a >> 4) + 1u) & 15u);
        const uint32_t tri = (a >> 1) & 7u;
        const bool goal = k == MGX_K_GOAL;
        if (!goal || tt > 0) {
            tt = tt > 0 ? tt - 1 : 0; // Box counts below zero in Python; every value <= 0 behaves the same
            a = (a & 0x0Fu) | ((uint32_t)((tt - 1) & 15) << 4);
            if (tt <= 0 && tri == 0u) { // Goal: removed; Box: replaced by its contents
                tog = goal ? (uint32_t)MGX_CODE_EMPTY : (uint32_t)o.cont[tidx];
                o.cont[tidx] = MGX_CODE_EMPTY;
                a = 0;
            } else if (tt <= 0) tog = (fc & 0x8Fu) | ((tri - 1u) << 4); // self.color = self.triage_color
            o.aux[tidx] = (uint8_t)a;
            L.dirty = MGX_REC_DIRTY; // (a toggle that only counts down leaves the cell code as it was)
        }
    }
    L.carry = pick ? fc : (drop ? (uint32_t)MGX_CODE_EMPTY : carry);
    // act == 6 ("done"): pass (minigrid.py:1291-1293)
    return pick ? (uint32_t)MGX_CODE_EMPTY : (drop ? carry : (act == 5 ? tog : fc));
}

// auto-reset of the hidden object state: planes back to the snapshot -- only if the episode wrote to them (`planes`: the record's dirty
// bit, set by every plane write of transition_apply, or a next level waiting in the snapshot); nothing carried.  TwoGoals' episodes end
// on the `done` action, one env in seven per step under a random policy: copying 2 x S bytes per lane for each of them made its step
// 208 us at 524,288 envs of 16x16.
__device__ __forceinline__ void restore_objstate(const StepParams &p, int64_t env, int64_t senv, bool planes)
{
    if (!p.objaux) return;
    if (planes) {
        const uint32_t *a0 = reinterpret_cast<const uint32_t *>(p.objaux0 + senv * p.S), *c0 = reinterpret_cast<const uint32_t *>(p.objcont0 + senv * p.S);
        uint32_t *a = reinterpret_cast<uint32_t *>(p.objaux + env * p.S), *c = reinterpret_cast<uint32_t *>(p.objcont + env * p.S);
        const int SD = p.S >> 2;
        for (int i0 = 0; i0 < SD; i0 += 4) { // (loads first, clamped; then the stores)
            uint32_t va[4], vc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int i = i0 + k < SD ? i0 + k : SD - 1; va[k] = a0[i]; vc[k] = c0[i]; }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k < SD) { a[i0 + k] = va[k]; c[i0 + k] = vc[k]; }
        }
    }
    p.objcarry[env] = (uint16_t)(MGX_CODE_EMPTY << 8);
}

// Restore this lane's env to its episode-start snapshot (LDS image + HBM).  Each done lane copies its own
// S bytes with all loads issued back to back, so a wave pays ONE memory latency however many of its envs
// finished (a wave-cooperative loop over done envs would pay one per env: measured 2x slower end to end on
// LavaCrossing, where 40% of the waves see a reset every step).
// (senv: the env's index in the snapshot arrays -- env itself, or env + bank * n_pad under a seed schedule)
template <int CS>
__device__ __forceinline__ void restore_own(const StepParams &p, int64_t env, int64_t senv, uint8_t *g)
{
    const int S = CS ? CS : p.S;
    uint32_t *l32 = reinterpret_cast<uint32_t *>(g);
    if constexpr (CS != 0 && (CS % 16) == 0 && CS <= 128) {
        // (every load before the first store: written as one loop -- load, store, load, ... -- the stores may alias the next load as far
        // as the compiler knows, and the ISA was four dependent round trips: s_waitcnt vmcnt(0) behind each load.  Dynamic-Obstacles and
        // new_level_each_episode handles restore every finished env, i.e. some lane of nearly every wave on every step.)
        const uint4 *s = reinterpret_cast<const uint4 *>(p.cells0 + senv * S);
        uint4 *d = reinterpret_cast<uint4 *>(p.cells + env * S);
        uint4 v[CS / 16];
#pragma unroll
        for (int i = 0; i < CS / 16; i++) v[i] = s[i];
#pragma unroll
        for (int i = 0; i < CS / 16; i++) {
            l32[4 * i + 0] = v[i].x; l32[4 * i + 1] = v[i].y; l32[4 * i + 2] = v[i].z; l32[4 * i + 3] = v[i].w;
            d[i] = v[i];
        }
    } else if constexpr (CS != 0 && (CS % 16) == 0) { // (16x16: 64 registers would not pay; four at a time)
        const uint4 *s = reinterpret_cast<const uint4 *>(p.cells0 + senv * S);
        uint4 *d = reinterpret_cast<uint4 *>(p.cells + env * S);
#pragma unroll 1
        for (int i0 = 0; i0 < CS / 16; i0 += 4) {
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = s[i0 + i];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                l32[4 * (i0 + i) + 0] = v[i].x; l32[4 * (i0 + i) + 1] = v[i].y; l32[4 * (i0 + i) + 2] = v[i].z; l32[4 * (i0 + i) + 3] = v[i].w;
                d[i0 + i] = v[i];
            }
        }
    } else {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(p.cells0 + senv * S);
        uint32_t *d = reinterpret_cast<uint32_t *>(p.cells + env * S);
        if constexpr (CS != 0) { // e.g. 9x9: 21 dword loads, all in flight before the first use
            uint32_t v[CS / 4];
#pragma unroll
            for (int i = 0; i < CS / 4; i++) v[i] = s[i];
#pragma unroll
            for (int i = 0; i < CS / 4; i++) { l32[i] = v[i]; d[i] = v[i]; }
        } else {
            // (eight loads in flight, then eight stores: one load / store pair per trip compiled to a round trip per dword -- see above)
            const int SD = S >> 2;
            for (int i0 = 0; i0 < SD; i0 += 8) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = s[i0 + k < SD ? i0 + k : SD - 1]; // (clamped, not conditional: a load in a branch is not hoisted)
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (i0 + k < SD) { l32[i0 + k] = v[k]; d[i0 + k] = v[k]; }
            }
        }
    }
}

// ActionBonus / StateBonus (wrappers.py:87-153): a visit count per env and key -- (agent_pos, agent_dir, action) / agent_pos, the state AFTER the
// step and before any reset, kept across episodes -- and reward += 1 / math.sqrt(new_count), in the order the wrappers were stacked
// (mgx_add_bonus), in doubles like the reference.  The lane owns its env: a plain read-modify-write.
__device__ __forceinline__ double exploration_bonus(const StepParams &p, int64_t env, const Lane &L, uint32_t act, double reward)
{
    const int cell = (int)L.ax * p.H + (int)L.ay;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int kind = (p.bonus >> (4 * k)) & 15;
        if (kind == 0) break;
        uint32_t *c = kind == MGX_BONUS_ACTION ? p.bonus_action + env * (int64_t)(p.W * p.H * 4 * p.bonus_na) + ((cell * 4 + (int)L.dir) * p.bonus_na + (int)act)
                                               : p.bonus_state + env * (int64_t)(p.W * p.H) + cell;
        const uint32_t n = *c + 1u;
        *c = n;
        reward += 1.0 / sqrt((double)n);
    }
    return reward;
}

__device__ __forceinline__ void wave_stats(const StepParams &p, bool valid, bool done, float reward, bool bad_act, bool oob, int lane, int tile)
{
    MgxCounterShard *sh = &p.ctr->shard[tile & (MGX_CTR_SHARDS - 1)];
    const u64 md = __ballot(valid && done), mr = __ballot(valid && reward != 0.f);
    const u64 ma = __ballot(bad_act), mo = __ballot(oob);
    if (md && lane == 0) atomicAdd(&sh->episodes, (u64)__popcll(md));
    if (ma && lane == 0) atomicAdd(&sh->invalid_actions, (u64)__popcll(ma));
    if (mo && lane == 0) atomicAdd(&sh->out_of_bounds, (u64)__popcll(mo));
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
// WIN != 0 (the gather form's window): `g` is not the env's grid but a raw V-column x WIN-row excerpt of it (column stride WIN bytes) whose
// cell (0, 0) is world cell (wx0, wy0); indices are taken relative to that origin while the in-grid tests still use the real W x H.
// OH: the OneHotPartialObsWrapper image instead of the triples (a template parameter, not a run-time branch: with the expansion compiled
// into every instance the step kernels went from 66-70 to 176 VGPRs).
template <int CW, int CH, int V, bool ALT, bool GATHER = false, int WIN = 0, bool OH = false>
__device__ __forceinline__ void emit_partial_obs(const StepParams &p, const Lane &L, uint8_t *lds, const uint8_t *g,
                                                 int64_t env0, int lane, int pidx = -1, uint32_t pcode = 0, unsigned long long *tlv = nullptr,
                                                 int wx0 = 0, int wy0 = 0)
{
    constexpr int B = V * V * 3;       // bytes per observation (147 for V = 7)
    constexpr int NDW = (B + 1) / 4;   // dwords holding one observation, the last with 3 valid bytes (37)
    constexpr int NQ = (V * V) / 4;    // groups of 4 cells = 3 dwords (12), plus one last cell
    const int W = CW ? CW : p.W, H = CH ? CH : p.H;
    const int dir = L.dir;
    const int dx = (dir == 0) - (dir == 2), dy = (dir == 1) - (dir == 3);
    const int rx = -dy, ry = dx; // right_vec (minigrid.py:1102-1109)
    const int HS = WIN ? WIN : H; // byte stride between columns of `g`
    const int base = WIN ? (L.ax - wx0) * WIN + (L.ay - wy0) : L.ax * H + L.ay;
    const int sf = dx * HS + dy, sr = rx * HS + ry;

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
            // (LDS forms: a cell outside the grid is read where its index points -- a neighbouring env's row or the guard band,
            // StepParams.lds_guard -- and replaced by the wall below; only the global-memory form has to clamp the index)
            // (... and the window form: its V x RS excerpt is this lane's own small LDS slot, an index outside the grid would point below the
            // wave's allocation or into a neighbour's slot)
            const int idx = ((GATHER || WIN != 0) && !inb) ? base : rowbase + (vx - V / 2) * sr;
#ifdef MGX_EXP_FIXED_GATHER /* timing / counter experiment only (wrong observations): every lane reads the same offset of its own row, which is
                               conflict-free under the odd dword stride -- what the view gather would cost without LDS bank conflicts */
            uint32_t c = g[(vx * V + vy) & 63];
#else
            uint32_t c = g[idx];
#endif
            if constexpr (GATHER) c = idx == pidx ? pcode : c; // the cell this step changed is not in HBM yet for this lane's reads
            code[vx][vy] = inb ? c : (uint32_t)MGX_CODE_WALL_GREY;
        }
    }

#ifdef MGX_TIMELINE
    if (tlv) tlv[7] = __builtin_amdgcn_s_memrealtime();
#endif
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
#ifdef MGX_TIMELINE
    if (tlv) tlv[8] = __builtin_amdgcn_s_memrealtime();
#endif
    // the agent's own cell shows what it carries, after occlusion; always visible (minigrid.py:1349-1356)
    code[V / 2][V - 1] = L.carry;

    if constexpr (OH) {
        // OneHotPartialObsWrapper fused (wrappers.py:226-243: out[vx][vy][type] = out[.., 11 + color] = out[.., 18 + state] = 1, 21 channels): the
        // V*V*21 bytes per env leave straight from here -- no triples written to HBM and read back by a second kernel (2 x 147 B per env-step).
        // An env's 1,029 bytes are not 16-byte aligned, 16 envs' are: the tile goes out in four quarters.  Per quarter: the wave zeroes a
        // 16 x 1,029-byte LDS image, every lane drops the three ones of 12-13 cells into it (byte writes at type / 11 + color / 18 + state;
        // the cell codes of all 64 envs were parked in LDS first, so all 64 lanes work on every quarter), and the image leaves as 16-B/lane
        // non-temporal stores.  ~90 LDS instructions and ~200 VALU per quarter against 16 KB stored.
#ifndef MGX_OH_UNIT
#define MGX_OH_UNIT 16 /* envs per LDS image: 16 (16-byte stores) or 8 (8-byte stores, half the LDS) */
#endif
        constexpr int QE = MGX_OH_UNIT, SB = QE == 8 ? 8 : 16; // envs per image, bytes per lane and store
        constexpr int NC = V * V, CSTR = (NC + 3) & ~3, NB = 21, QB = QE * NC * NB;
        // (QE = 4: an image of 4,116 bytes is 257 16-byte stores at a 4-BYTE-aligned address -- global_store_dwordx4 only needs that -- and one dword)
        static_assert(QB % 4 == 0, "an image is a whole number of dwords");
        wave_sync(); // every lane has gathered its view: the grid image (or the window excerpts) may be overwritten
        uint32_t *cw = reinterpret_cast<uint32_t *>(lds) + lane * (CSTR / 4);
#pragma unroll
        for (int q = 0; q < CSTR / 4; q++) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = 4 * q + j;
                if (idx < NC) w |= (code[idx / V][idx % V] & 255u) << (8 * j);
            }
            cw[q] = w;
        }
        uint8_t *img = lds + 64 * CSTR;
        const int64_t nvv = p.n - env0;
        const int n_env = nvv >= 64 ? 64 : (int)nvv;
        uint8_t *dst = p.obs + env0 * (int64_t)(NC * NB);
        constexpr int NCH = QB / SB, NZ = (QB + 15) / 16, REMW = (QB - NCH * SB) / 4; // stores / 16-byte zero chunks per image / dwords behind the last store
        for (int qt = 0; qt < 64 / QE && QE * qt < n_env; qt++) { // wave-uniform
            wave_sync(); // (the codes are parked / the previous quarter's image has been read)
            // (every loop below has a compile-time trip count: rolled, with run-time bounds, the store loop was one ds_read -> s_waitcnt -> store
            // per trip with a page of tail-handling branches around it)
#pragma unroll
            for (int i = 0; i < (NZ + 63) / 64; i++) {
                const int c = lane + 64 * i;
                if (64 * i + 63 < NZ || c < NZ) reinterpret_cast<uint4 *>(img)[c] = make_uint4(0u, 0u, 0u, 0u);
            }
            wave_sync();
            {
                constexpr int NI = (QE * NC + 63) / 64;
                uint32_t cc[NI];
#pragma unroll
                for (int i = 0; i < NI; i++) { // the cell codes first, all in flight
                    const int ci = lane + 64 * i, cj = ci < QE * NC ? ci : 0;
                    const int e = cj / NC, cell = cj - e * NC; // (compile-time divisor)
                    cc[i] = lds[(QE * qt + e) * CSTR + cell];
                }
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const int ci = lane + 64 * i;
                    if (64 * i + 63 < QE * NC || ci < QE * NC) {
                        const uint32_t c = cc[i], k = c & 15u, col = (c >> 4) & 7u;
                        const bool shut = k > MGX_K_AGENT; // closed / locked door: type 4, state 1 / 2
                        uint8_t *o = img + ci * NB;
                        o[shut ? 4u : k] = 1;
                        o[11u + col] = 1;
                        o[18u + (shut ? k - 10u : 0u)] = 1;
                    }
                }
            }
            wave_sync();
            const int q_env = n_env - QE * qt >= QE ? QE : n_env - QE * qt;
            uint8_t *d = dst + (size_t)qt * QB;
            if (q_env == QE) { // a whole image: groups of six reads, then their six stores
                constexpr int NS = (NCH + 63) / 64, G = 6;
#pragma unroll
                for (int i0 = 0; i0 < NS; i0 += G) {
                    uint4 v4[G];
                    unsigned long long v8[G];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const int c = lane + 64 * (i0 + g), cr = c < NCH ? c : NCH - 1;
                        if (i0 + g < NS) { if constexpr (SB == 16) v4[g] = reinterpret_cast<const uint4 *>(img)[cr]; else v8[g] = reinterpret_cast<const unsigned long long *>(img)[cr]; }
                    }
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const int c = lane + 64 * (i0 + g);
                        if (i0 + g < NS && (64 * (i0 + g) + 63 < NCH || c < NCH)) {
                            if constexpr (SB == 16) nt_store16(reinterpret_cast<uint4 *>(d) + c, v4[g]);
                            else __builtin_nontemporal_store(v8[g], reinterpret_cast<unsigned long long *>(d) + c);
                        }
                    }
                }
                if constexpr (REMW != 0) { if (lane < REMW) reinterpret_cast<uint32_t *>(d)[NCH * (SB / 4) + lane] = reinterpret_cast<const uint32_t *>(img)[NCH * (SB / 4) + lane]; }
            } else { // the short last image of a tail tile
                const int lim = q_env * NC * NB;
                for (int c = lane; c < NCH + (REMW ? 1 : 0); c += 64) {
                    if (SB * c + SB <= lim) {
                        if constexpr (SB == 16) reinterpret_cast<uint4 *>(d)[c] = reinterpret_cast<const uint4 *>(img)[c];
                        else reinterpret_cast<unsigned long long *>(d)[c] = reinterpret_cast<const unsigned long long *>(img)[c];
                    } else
                        for (int bb = SB * c; bb < lim; bb++) d[bb] = img[bb];
                }
            }
        }
        return;
    }
    // 49 codes -> 49 (type, color, state) triples (image[vx][vy][c], vx-major) in 37 dwords, FOUR CELLS PER INSTRUCTION: the codes
    // of 4 consecutive cells are packed into one dword and decoded byte-parallel with the instructions that issue at full rate
    // on this chip (and / or / xor / add / sub / lshr: tools/ubench/issue_rate.hip; v_perm, v_cndmask, v_cmp, v_bfe, v_lshl*, SDWA run
    // at half rate, and the cell-by-cell decode was made of them: ~15 issue slots per cell, now ~8 with the packing and the
    // interleave):   k = c & 15;  col = (c >> 4) & 7;  shut = k >= 11  (k + 0x75 carries into bit 7; k <= 15: no carry between
    // bytes);  type = shut ? 4 : k;  state = shut ? (k + 2) & 3 : 0   (11 -> 1 closed, 12 -> 2 locked).
    uint32_t D[NDW];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t c0 = code[(4 * q) / V][(4 * q) % V], c1 = code[(4 * q + 1) / V][(4 * q + 1) % V];
        const uint32_t c2 = code[(4 * q + 2) / V][(4 * q + 2) % V], c3 = code[(4 * q + 3) / V][(4 * q + 3) % V];
        const uint32_t P = __builtin_amdgcn_perm(c1, c0, 0x0C0C0400u) | __builtin_amdgcn_perm(c3, c2, 0x04000C0Cu); // c0 c1 c2 c3
        const uint32_t k4 = P & 0x0F0F0F0Fu;
        const uint32_t col4 = (P >> 4) & 0x07070707u;
        const uint32_t m80 = (k4 + 0x75757575u) & 0x80808080u;      // 0x80 in the bytes of closed / locked doors
        const uint32_t m1 = m80 >> 7;                                // 0x01 there
        const uint32_t mFF = (m80 - m1) | m80;                       // 0xFF there
        const uint32_t T4 = k4 ^ ((k4 ^ 0x04040404u) & mFF);         // type: 4 for doors of any state
        const uint32_t S4 = (k4 + 0x02020202u) & 0x03030303u & mFF;  // state
        // interleave the three byte planes: t0 c0 s0 t1 | c1 s1 t2 c2 | s2 t3 c3 s3
        const uint32_t X0 = __builtin_amdgcn_perm(col4, T4, 0x010C0400u), X1 = __builtin_amdgcn_perm(col4, T4, 0x06020C05u),
                       X2 = __builtin_amdgcn_perm(col4, T4, 0x0C07030Cu);
        D[3 * q + 0] = __builtin_amdgcn_perm(S4, X0, 0x03040100u);
        D[3 * q + 1] = __builtin_amdgcn_perm(S4, X1, 0x03020500u);
        D[3 * q + 2] = __builtin_amdgcn_perm(S4, X2, 0x07020106u);
    }
    D[NDW - 1] = decode_triple(code[V - 1][V - 1]); // 3 bytes

#ifdef MGX_TIMELINE
    if (tlv) tlv[9] = __builtin_amdgcn_s_memrealtime();
#endif
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
#ifdef MGX_TIMELINE
    if (tlv) tlv[3] = __builtin_amdgcn_s_memrealtime();
#endif
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

    // general sizes: units of 4 consecutive cells of the tile's flat cell stream (cell f -> env f / cells, cell f %
    // cells) = 12 output bytes, one global_store_dwordx3 per lane, consecutive lanes consecutive records (a lane per
    // output DWORD with 4-B stores, as this path first did, measured 3.3 TB/s at 9x9; an env's 3*W*H bytes are not
    // dword aligned but the tile's are)
    const int n_flat = n_env * cells, n_u = n_flat >> 2;
    for (int u = lane; u < n_u; u += 64) {
        const int f0 = 4 * u;
        int e = f0 / cells, c = f0 - e * cells;
        uint32_t t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t[k] = decode_triple_full(lds[e * LS + c]);
            if (++c == cells) { c = 0; e++; }
        }
        nt_store12(reinterpret_cast<uint32_t *>(dst + 12 * (size_t)u), __builtin_amdgcn_perm(t[1], t[0], 0x04020100u),
                   __builtin_amdgcn_perm(t[2], t[1], 0x05040201u), __builtin_amdgcn_perm(t[3], t[2], 0x06050402u));
    }
    // (n_env*cells is a multiple of 4 unless this is the tail tile: finish cell by cell)
    for (int f = 4 * n_u + lane; f < n_flat; f += 64) {
        const int e = f / cells, c = f - e * cells;
        const uint32_t t = decode_triple_full(lds[e * LS + c]);
        dst[3 * (size_t)f] = (uint8_t)t; dst[3 * (size_t)f + 1] = (uint8_t)(t >> 8); dst[3 * (size_t)f + 2] = (uint8_t)(t >> 16);
    }
}



// OBJ: the handle keeps hidden Goal/Box state (object_state planes).  Always on for the run-time-size instances, and for the two
// sized ones ObstructedMaze needs (11x6, 16x16: round 2); pruned from every other sized instance.
// The gather form (MODE 3) lives on memory latency: record -> transition -> 7 scattered window loads -> observation.  Holding its
// instances to 64 / 72 VGPRs (8 / 7 waves per SIMD instead of the 5 the compiler's 88 VGPRs give) was measured and is NOT done: with the
// window loads all in flight at once 5, 6 and "no cap" run within 1 % of each other and 8 waves per SIMD 3 % slower (FourRooms, 1 Mi
// envs: 83.9 / 84.0 / 84.1 / 86.8 us) -- more waves in flight only deepen the queue in front of a memory system that is already busy.
#ifdef MGX_GATHER_WAVES_ALL /* (tuning builds: tools/build_variant.sh x -DMGX_GATHER_WAVES_ALL=n) */
#define MGX_GATHER_WAVES(CW) MGX_GATHER_WAVES_ALL
#else
#define MGX_GATHER_WAVES(CW) 1
#endif
// DYN (Dynamic-Obstacles, staged partial form): the obstacle walk of envs/dynamicobstacles.py:60-80 runs in front of the transition on the
// SAME staged tile (dynobs_device.h) -- one launch instead of k_dynobs + k_step, the cells read once and written back once per step, no
// folded-action buffer in between.
template <int CW, int CH, int MODE, int V, bool ALT, bool OBJ, bool DYN, bool OH = false, bool WRAP = false>
__device__ __forceinline__ void step_body(const StepParams &p, const DynObsParams *dp)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= p.n_tiles) return; // wave-uniform
    constexpr int CS = (CW && CH) ? ((CW * CH + 3) & ~3) : 0;
    const int S = CS ? CS : p.S;
    constexpr int CLS = CS + (((CS >> 2) & 1) ? 0 : 4); // the host's rule (mgx_create): odd dword stride per env in LDS
    const int LS = CS ? CLS : p.LS;
    uint8_t *lds = smem + p.lds_guard + (size_t)wv * p.wave_lds;
    const int64_t env0 = (int64_t)tile * 64;
    const int64_t env = env0 + lane;
    const bool valid = env < p.n;

    if (p.obs_mask && !__ballot(valid && p.obs_mask[env])) return; // wave-uniform: nothing in this tile was reset
    // The grid's last blocks are dispatched into a chip that is draining: at the default (equal) priority they share their SIMDs'
    // issue slots with the waves still finishing, and the launch ends one contended wave lifetime after they start.  At priority 3
    // they run through first; worth 1.6 us of 26.5 at 524,288 LavaCrossing envs and 1.1 of 39.4 at 1 Mi Empty-8x8.
#ifdef MGX_TIMELINE
    unsigned long long tlv[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    tlv[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if ((int)blockIdx.x >= p.tail_block0) __builtin_amdgcn_s_setprio(3);
    // ... and the FIRST round of blocks starts all at once: every wave's loads queue behind everyone else's, then 7 waves per SIMD
    // compute at the same time, and nothing is stored until the first of them is through (wave timelines, profiles/r02_timeline_*:
    // no store before 7 us of a 24 us launch).  The waves of the first round therefore start one after the other per SIMD: slot k of
    // a SIMD sleeps k * stagger * 256 clocks before its first load.  LavaCrossing 524,288 envs 25.0 -> 22.6 us, 1 Mi 45.3 -> 43.6.
    if (p.stagger && (int)blockIdx.x < p.round_blocks) {
        uint32_t slot;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(slot));
        for (uint32_t i = (slot < 7u ? slot : 7u) * (uint32_t)p.stagger; i; i--) __builtin_amdgcn_s_sleep(4);
    }
    uint32_t act = 6;
    if constexpr (DYN) act = dynobs_walk<CW, CH, true>(*dp, lds, lane, tile); // (stages the tile; the walk never moves the agent)
    else if (p.do_step && valid) act = __builtin_nontemporal_load(&p.actions[env]);
    const uint2 rec = p.agent[env]; // agent/cells arrays are padded to whole tiles
    const bool crash = p.task == MGX_TASK_DYNOBS && (act & 0x80u); // k_dynobs' verdict rides on the folded action
    if (p.task == MGX_TASK_DYNOBS) act &= 0x7Fu;
    if (p.task == MGX_TASK_MEMORY && act == 3u) act = 5u; // `if action == pickup: action = toggle` (envs/memory.py:89-90)
    // (gather form) the code of the cell in front of the agent as the LAST observation pass of this env saw it, 0 = unknown: a
    // coalesced byte per env instead of a dependent one-byte gather whose 128-byte line is fetched, evicted under the launch's
    // own traffic and fetched again for the view (FETCH_SIZE: 428 B per env-step at 1 Mi FourRooms envs, 633 at MultiRoom-N6)
    uint32_t front0 = 0;
    if constexpr (MODE == 3) { if (p.front && p.do_step) front0 = p.front[env]; }
    // (gather form, 7x7 view) the window excerpt of the env's last observation pass + the pose it was loaded for (StepParams.wcache): 64
    // bytes per lane, 4 KB contiguous per wave
#ifdef MGX_WC_STRICT /* (A/B builds) a miss rewrites the record only when the step left the pose alone */
#define MGX_WC_WRITE_ON_MISS(streak) false
#else
#define MGX_WC_WRITE_ON_MISS(streak) ((streak) <= 1u) /* ... or while the env has missed at most once in a row */
#endif
    constexpr bool WC = MODE == 3 && V == 7 && !ALT;
    uint4 wc[WC ? 4 : 1];
    if constexpr (WC) {
        if (p.wcache) {
            const uint4 *r4 = reinterpret_cast<const uint4 *>(p.wcache + env * 64);
#pragma unroll
            for (int i = 0; i < 4; i++) wc[i] = r4[i];
        }
    }
    bool was_reset = false;
    // MODE 3 (large grids): no tile image in LDS -- at 25x25 it would be 40 KB per wave and leave 4 waves per CU; each
    // lane gathers its forward cell and its VxV view straight from its row in HBM/L2 instead (50 byte loads).
    constexpr bool GATHER = MODE == 3;
    if constexpr (!GATHER && !DYN) {
        stage_tile<CS>(p.cells, env0, S, LS, lds, lane);
        wave_sync();
    }
#ifdef MGX_TIMELINE
    tlv[1] = __builtin_amdgcn_s_memrealtime();
#endif

    Lane L = unpack_rec(rec, p.task);
    uint8_t *g = lds + lane * LS;                       // (staged modes) this env's cells in LDS
    const uint8_t *row = GATHER ? p.cells + env * S : g; // where cells are read from
    // exploration bonuses and the DAC wrapper ride on instances of their own (WRAP: k_step_wrap / k_step_dyn_wrap, run-time grid size;
    // mgx_launch_step routes such handles there): every other kernel keeps the float reward and not one instruction of either (with the
    // double in every instance LavaCrossing at 524,288 envs measured 25.0 us per step against 22.8; in the run-time-size instances alone
    // the other view sizes measured + 2 ... 4 %)
    constexpr bool BONUS = WRAP;
    using RewardT = typename std::conditional<BONUS, double, float>::type;
    RewardT reward = 0;
    bool done = false, bad_act = false, oob = false;
    // DACWrapper (wrappers.py:35-84; mgx_set_dac, the same instances): an env that is done before its time-out is not reset but absorbed --
    // its steps only count on, reward 0, until step max_steps reports done
    bool absorbed = false, absorb_now = false;
    if constexpr (BONUS) {
        if (p.dac) { absorbed = (rec.x & MGX_REC_ABSORBED) != 0u; L.dirty |= rec.x & MGX_REC_ABSORBED; }
    }
    int pidx = -1;
    uint32_t pcode = 0;
    if (p.do_step) {
        const int fidx = transition_begin<CW, CH>(p, L, act, valid && !absorbed, bad_act, oob);
        uint32_t fc = 0, nc = 0; // forward cell before / after the transition
#ifdef MGX_TIMELINE
        tlv[5] = __builtin_amdgcn_s_memrealtime();
#endif
        if (fidx >= 0) {
            const uint32_t carry0 = L.carry;
            if (GATHER && front0 != 0u && act < 7u) fc = front0; // (strafe targets are the left / right cell: read from the row)
            else fc = row[fidx];
            // hidden object state rides only on the run-time-size kernels (mgx_launch_step routes there): pruned from the sized ones
            const bool has_obj = OBJ && p.objaux != nullptr;
            const ObjRef obj = {has_obj ? p.objaux + env * S : nullptr, has_obj ? p.objcont + env * S : nullptr, has_obj ? p.objcarry + env : nullptr};
            nc = transition_apply<CH>(p, L, act, fc, reward, done, [&](int i) -> uint32_t { return row[i]; }, oob, fidx, obj);
            if constexpr (!GATHER) { if (nc != fc) g[fidx] = (uint8_t)nc; }
            if (valid && L.steps >= p.max_steps) done = true; // minigrid.py:1320-1321
            if (p.task) task_rule<CH>(p, L, act, reward, done, [&](int i) -> uint32_t { return (GATHER && i == fidx) ? nc : (uint32_t)row[i]; }, fidx, fc, carry0, oob);
            if constexpr (BONUS) {
                if (p.dac && done && L.steps < p.max_steps) { done = false; absorb_now = true; } // (`return obs, rew, False, info`, wrappers.py:69-78)
            }
            // the one cell a transition can change; skipped when the env is about to be restored anyway
            if (nc != fc && !(p.auto_reset && done)) { p.cells[env * S + fidx] = (uint8_t)nc; pidx = fidx; pcode = nc; L.dirty = MGX_REC_DIRTY; }
        } else if (valid && L.steps >= p.max_steps) done = true;
        if constexpr (BONUS) { // (the time-out step returns last_obs as well: `obs = self.last_obs; self.env_done = True`, wrappers.py:71-76)
            if (p.dac && valid && (done || absorb_now)) L.dirty |= MGX_REC_ABSORBED;
        }
        if (crash) { reward = (RewardT)-1; done = true; } // envs/dynamicobstacles.py:83-86
        if constexpr (BONUS) {
            if (p.bonus && valid && !bad_act) // (the wrapper's key holds the action the CALLER gave: MemoryEnv.step turns pickup into toggle on its own)
                reward = exploration_bonus(p, env, L, p.task == MGX_TASK_MEMORY ? (uint32_t)p.actions[env] : act, reward);
        }
#ifdef MGX_TIMELINE
        tlv[6] = __builtin_amdgcn_s_memrealtime();
#endif
        const float reward_out = (float)reward;
        if (p.reward && valid) __builtin_nontemporal_store(reward_out, &p.reward[env]);
        if (p.done && valid) __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &p.done[env]);
        wave_stats(p, valid, done, reward_out, bad_act, oob, lane, tile);
        // the snapshot differs from the current cells only if a step changed one (dirty), if it holds the NEXT level (stream
        // mode) or if k_dynobs moved obstacles (both: p.regen)
        // (... or, under a seed schedule, the level of the NEXT seed of the env's list: ReseedWrapper.reset, wrappers.py:24-28)
        const bool needs_copy = L.dirty != 0u || p.regen != nullptr || p.bank != nullptr;
        int nb = 0; // seed schedule: the list entry the new episode runs on
        if (p.bank && p.auto_reset && valid && done) {
            if (p.ring) { nb = (int)p.bank[env]; p.bank[env] = (uint8_t)((nb + 1) & (p.ring - 1)); } // (ring of next-level buffers: this one is consumed)
            else {
                nb = (int)p.bank[env] + 1;
                nb = nb >= p.n_banks ? 0 : nb;
                p.bank[env] = (uint8_t)nb;
            }
        }
        const int64_t senv = env + (int64_t)nb * p.bank_envs; // this env in the snapshot arrays
        if constexpr (GATHER) {
            // restore global -> global, by the whole wave one finished env at a time: these rows are long (> 256 B), a lane
            // copying its own row touches 64 different lines per instruction (MultiRoom's synchronised time-outs: +15 us
            // per step on average); the observation below reads the snapshot itself
            for (u64 m = __ballot(p.auto_reset && valid && done && needs_copy); m; m &= m - 1) {
                const int el = __builtin_ctzll(m);
                const int64_t e = env0 + el;
                const int64_t se = e + (int64_t)__shfl(nb, el) * p.bank_envs;
                const uint32_t *s = reinterpret_cast<const uint32_t *>(p.cells0 + se * S);
                uint32_t *d = reinterpret_cast<uint32_t *>(p.cells + e * S);
                for (int i = lane; i < (S >> 2); i += 64) d[i] = s[i];
                if (OBJ && p.objaux) { // the hidden planes ride along (a lane copying its own 2 x S bytes dword by dword: ObstructedMaze's lock-step time-outs)
                    const uint32_t *a0 = reinterpret_cast<const uint32_t *>(p.objaux0 + se * S), *c0 = reinterpret_cast<const uint32_t *>(p.objcont0 + se * S);
                    uint32_t *a = reinterpret_cast<uint32_t *>(p.objaux + e * S), *c = reinterpret_cast<uint32_t *>(p.objcont + e * S);
                    for (int i = lane; i < (S >> 2); i += 64) { a[i] = a0[i]; c[i] = c0[i]; }
                }
            }
        }
        if (p.auto_reset && valid && done) {
            was_reset = true;
            if constexpr (GATHER) {
                row = p.cells0 + senv * S;
                pidx = -1;
            } else if (needs_copy) restore_own<CS>(p, env, senv, g);
            else if (nc != fc) g[fidx] = (uint8_t)fc; // a terminal step that changed a cell (Fetch's pickup): only the LDS image saw it
            if (OBJ) restore_objstate(p, env, senv, needs_copy && !GATHER); // (gather form: the planes were copied by the whole wave above)
            // (loaded here, by the waves that need it: fetching agent0 with the record up front takes a 3-6 us round trip under load
            // out of 40 % of LavaCrossing's waves and still measured +0.6 ... +1.5 us per launch -- 8 B per env of extra requests)
            L = unpack_rec(p.agent0[senv], p.task);
            if (p.regen) p.regen[env] = (uint8_t)(p.ring ? nb + 1 : 1); // the next-level buffer was consumed: k_levelgen refills it (ring: THAT buffer)
        }
        if (valid) p.agent[env] = pack_rec(L, p.task);
    }
    if constexpr (DYN) { // the moved obstacles (and the restored rows of finished envs) go home as the whole tile, coalesced
        wave_sync();
        unstage_tile<CS>(p.cells, env0, S, LS, lds, lane);
    }
    if constexpr (GATHER) { if (!p.obs && p.front && p.do_step && valid) p.front[env] = 0; } // (no observation pass: nothing to remember)
    if constexpr (WC) { if (!p.obs && p.wcache && p.do_step && valid) reinterpret_cast<uint32_t *>(p.wcache + env * 64)[15] = 0u; }
    if (p.obs) {
        if constexpr (GATHER) {
            // The VxV view always lies inside a world-aligned VxV window whose columns (fixed world x) are V contiguous bytes: V unaligned
            // loads of RS = 4 / 8 / 12 bytes per lane (V = 3 / 5, 7 / 9, 11) instead of V*V byte loads.  The excerpt goes to this lane's LDS
            // slot as a tiny V x RS "grid" and the ordinary closed-form gather runs on it, with indices relative to the excerpt's origin
            // and the in-grid tests on the real W x H (cells outside the grid = grey wall, which is what Grid.slice pads with).
            constexpr int RS = V <= 3 ? 4 : (V <= 7 ? 8 : 12);  // rows loaded per column
            constexpr int SLOT = (V * RS / 4) | 1;               // dwords per lane, odd: 3, 11, 15, 27, 33
            if ((CH ? CH : p.H) >= RS) {
                const int H = CH ? CH : p.H, W = CW ? CW : p.W;
                const int x0 = L.dir == 0 ? L.ax : (L.dir == 2 ? L.ax - (V - 1) : L.ax - V / 2);
                const int y0 = L.dir == 1 ? L.ay : (L.dir == 3 ? L.ay - (V - 1) : L.ay - V / 2);
                const int yc = y0 < 0 ? 0 : (y0 > H - RS ? H - RS : y0); // the RS loaded rows start at yc: every in-grid row of the view is among them
                struct __attribute__((packed)) UR { uint32_t w[RS / 4]; };
                // All V loads are issued before the first use, on a column index clamped into the grid: a load inside
                // `if (x in the grid)` is not hoisted out of its branch, and seven conditional loads were seven dependent round
                // trips (an s_waitcnt vmcnt(0) behind each one: 22 us wave lifetime, 6.6 L2 requests per env instead of 2.2 because the
                // vector cache had long lost the line when the next column asked for it).  The excerpt goes to LDS as it is -- columns
                // outside the grid hold a copy of the border column, rows outside it are not there at all -- and the observation's own
                // in-grid tests (on the real W x H) turn exactly those cells into the grey wall Grid.slice pads with.
                UR raw[V];
                // The cached excerpt belongs to a pose: a step that turned, moved or restarted the episode reloads it (7 loads in ONE branch,
                // issued together), every other step -- 4 of 7 random actions, and every blocked forward -- observes from the record.
                const uint32_t tag = (uint32_t)L.ax | ((uint32_t)L.ay << 8) | ((uint32_t)L.dir << 16) | (1u << 24);
                bool hit = false;
                uint32_t streak = 0; // consecutive misses of this env (bits 27:26 of the record's last dword, saturating at 3)
                if constexpr (WC) {
                    hit = p.wcache != nullptr && !was_reset && (wc[3].w & 0x0103FFFFu) == tag;
                    streak = p.wcache ? (wc[3].w >> 26) & 3u : 0u;
                }
                if (!hit) {
#pragma unroll
                    for (int k = 0; k < V; k++) {
                        const int x = x0 + k, xc = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
                        raw[k] = *reinterpret_cast<const UR *>(row + xc * H + yc); // (yc + RS <= H: inside the row for every column)
                    }
                }
                if constexpr (WC) {
                    if (hit) {
                        const uint32_t cw[16] = {wc[0].x, wc[0].y, wc[0].z, wc[0].w, wc[1].x, wc[1].y, wc[1].z, wc[1].w,
                                                 wc[2].x, wc[2].y, wc[2].z, wc[2].w, wc[3].x, wc[3].y, wc[3].z, wc[3].w};
#pragma unroll
                        for (int k = 0; k < V; k++) { raw[k].w[0] = cw[2 * k]; raw[k].w[1] = cw[2 * k + 1]; }
                    }
                }
                uint32_t *win32 = reinterpret_cast<uint32_t *>(lds) + lane * SLOT; // odd dword stride per lane
#pragma unroll
                for (int k = 0; k < V; k++)
#pragma unroll
                    for (int q = 0; q < RS / 4; q++) win32[(RS / 4) * k + q] = raw[k].w[q];
                uint8_t *win = reinterpret_cast<uint8_t *>(win32);
                const int fdx = (L.dir == 0) - (L.dir == 2), fdy = (L.dir == 1) - (L.dir == 3);
                if (pidx >= 0) win[(L.ax + fdx - x0) * RS + L.ay + fdy - yc] = (uint8_t)pcode; // the front cell this step changed (the agent did not move then)
                if constexpr (WC) {
                    // The record follows (4 x 16 B per lane, contiguous over the wave): rewritten when this step changed a cell of it, and when it
                    // was reloaded -- unless the env keeps missing.  An agent that walks misses on every step, and a record rewritten behind
                    // each of them would be 64 bytes per step for nothing: after two misses in a row only the last dword is kept up to date
                    // (no pose, the miss count), until a step that leaves the pose alone reloads the window -- that one writes the record
                    // again, whatever the count.
                    if (p.wcache && valid) {
                        const bool stayed = !was_reset && ((uint32_t)L.ax | ((uint32_t)L.ay << 8) | ((uint32_t)L.dir << 16)) == (rec.x & 0x3FFFFu);
                        const uint32_t streak1 = hit ? 0u : (stayed ? 0u : (streak < 3u ? streak + 1u : 3u));
                        uint32_t *w32 = reinterpret_cast<uint32_t *>(p.wcache + env * 64);
                        if ((hit && pidx >= 0) || (!hit && (MGX_WC_WRITE_ON_MISS(streak) || stayed))) {
                            uint32_t o[16];
#pragma unroll
                            for (int q = 0; q < 14; q++) o[q] = win32[q];
                            o[14] = 0u; o[15] = tag | (streak1 << 26);
                            uint4 *w4 = reinterpret_cast<uint4 *>(w32);
#pragma unroll
                            for (int i = 0; i < 4; i++) w4[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
                        } else {
                            const uint32_t last = hit ? (tag | (streak1 << 26)) : (streak1 << 26); // (a miss that does not rewrite: the record belongs to no pose)
                            if (last != wc[3].w) w32[15] = last;
                        }
                    }
                }
                // the cell in front of the (new) pose lies inside the excerpt whenever it lies inside the grid: remembered for the next step
                if (p.front && valid) {
                    const int fx = L.ax + fdx, fy = L.ay + fdy;
                    p.front[env] = ((unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H) ? win[(fx - x0) * RS + fy - yc] : (uint8_t)0;
                }
                emit_partial_obs<CW, CH, V, ALT, false, RS, OH>(p, L, lds, win, env0, lane, -1, 0, nullptr, x0, yc);
            } else { // (a grid lower than the excerpt: V*V byte loads)
                if (p.front && valid) p.front[env] = 0;
                if constexpr (WC) { if (p.wcache && valid) reinterpret_cast<uint32_t *>(p.wcache + env * 64)[15] = 0u; }
                emit_partial_obs<CW, CH, V, ALT, true, 0, OH>(p, L, lds, row, env0, lane, pidx, pcode);
            }
        }
#ifdef MGX_TIMELINE
        else if (MODE == 0) {
            tlv[2] = __builtin_amdgcn_s_memrealtime();
            emit_partial_obs<CW, CH, V, ALT>(p, L, lds, g, env0, lane, -1, 0, tlv);
            tlv[4] = __builtin_amdgcn_s_memrealtime();
            if (p.timeline && lane == 0) {
                uint32_t hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                unsigned long long *o = p.timeline + (size_t)tile * 8;
                for (int i = 0; i < 5; i++) o[i] = tlv[i];
                o[5] = hw | ((unsigned long long)xcc << 32); o[6] = (tlv[5] - tlv[1]) | ((tlv[6] - tlv[1]) << 20) | ((unsigned long long)blockIdx.x << 40);
                o[7] = (tlv[7] - tlv[2]) | ((tlv[8] - tlv[2]) << 20) | ((tlv[9] - tlv[2]) << 40); // inside the observation: gathered / occluded / decoded
            }
        }
#else
        else if (MODE == 0) emit_partial_obs<CW, CH, V, ALT, false, 0, OH>(p, L, lds, g, env0, lane);
#endif
        else emit_full_obs<CW, CH>(p, L, valid, lds, g, LS, env0, lane);
    }
}

template <int CW, int CH, int MODE, int V, bool ALT = false, bool OBJ = (CW == 0)>
__global__ __launch_bounds__(256, (MODE == 3 && V == 7 && !ALT && !OBJ) ? MGX_GATHER_WAVES(CW) : 1) void k_step(const StepParams p)
{
    step_body<CW, CH, MODE, V, ALT, OBJ, false>(p, nullptr);
}

// OneHotPartialObsWrapper expanded in the step kernel (StepParams.onehot; run-time grid size, views 3 / 5 / 7, staged or gather form)
template <int MODE, int V>
__global__ __launch_bounds__(256) void k_step_onehot(const StepParams p)
{
    step_body<0, 0, MODE, V, false, true, false, true>(p, nullptr);
}

// handles with an exploration bonus or the DAC wrapper (mgx_add_bonus, mgx_set_dac): run-time grid size, hidden object state compiled in
template <int MODE, int V, bool ALT = false>
__global__ __launch_bounds__(256) void k_step_wrap(const StepParams p)
{
    step_body<0, 0, MODE, V, ALT, true, false, false, true>(p, nullptr);
}
__global__ __launch_bounds__(256) void k_step_dyn_wrap(const StepParams p, const DynObsParams d)
{
    step_body<0, 0, 0, 7, false, false, true, false, true>(p, &d);
}

template <int CW, int CH>
__global__ __launch_bounds__(256) void k_step_dyn(const StepParams p, const DynObsParams d)
{
    step_body<CW, CH, 0, 7, false, false, true>(p, &d);
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
//   RAGGED   : W*H not a multiple of 4 (5x5, 7x7, 9x9, 11x11, 19x19, 25x25 ...): an env's 3*W*H output bytes are then
//              not dword aligned, but the TILE's are, so the units run over the tile's flat cell sequence instead
//              (cell f -> env f / cells, cell f % cells; the padded rows in HBM are read with one unaligned dword
//              load per unit, byte-wise only for the one unit per env that straddles two envs).  The output side is
//              the same stream of 12-byte records.  (Before this form these sizes went through the LDS tile image:
//              3.0-3.7 TB/s for the sized kernels, 0.8-1.2 for 19x19 / 25x25.)
template <int CW, int CH, bool RAGGED = false, bool WRAP = false>
// (19x19: 64 VGPRs = 8 resident blocks per CU, so that 131,072 envs = 2,048 blocks are ONE round: 46.7 -> 40.2 us, at the price
// of six spilled dwords)
__global__ __launch_bounds__(256, (RAGGED && CW == 19) ? 8 : 1) void k_step_fulldirect(const StepParams p)
{
    __shared__ uint32_t s_info[64]; // per env: agent idx | dir<<16 | reset<<18 | (1<<19 if a cell changed) | snapshot bank<<20 (seed schedule)
    __shared__ uint32_t s_wr[64];   // changed cell: idx | code<<16
    __shared__ uint32_t s_lut[256]; // cell code -> (type | color<<8 | state<<16)
    constexpr int CS = (CW * CH + 3) & ~3;
    constexpr int KPF = (CS && !RAGGED) ? (CS / 4 * 64 + 255) / 256 : 0; // prefetched units per thread (compile-time sizes only)
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int H = CH ? CH : p.H;
    const int S = CS ? CS : p.S;
    const int UPE = S >> 2; // 4-cell units per env
    const int64_t env0 = (int64_t)tile * 64;
    const int64_t nv = p.n - env0;
    const int n_units = (nv >= 64 ? 64 : (int)nv) * UPE;
    const uint32_t *cells32 = reinterpret_cast<const uint32_t *>(p.cells + env0 * S);
    if (p.obs_mask && !__syncthreads_or((tid < 64 && env0 + tid < p.n) ? (int)p.obs_mask[env0 + tid] : 0)) return; // block-uniform
    if ((int)blockIdx.x >= p.tail_block0) __builtin_amdgcn_s_setprio(3); // the grid's last blocks first (see k_step)

    uint32_t pf[KPF ? KPF : 1];
#pragma unroll
    for (int k = 0; k < KPF; k++) {
        const int u = tid + 256 * k;
        pf[k] = cells32[u < n_units ? u : 0];
    }
    // RAGGED with a compile-time size (19x19 FourRooms / LockedRoom / Playground, 25x25 MultiRoom): the units of the tile's flat
    // cell sequence are known per thread, so ALL of a thread's unaligned dword loads go out here, before phase A -- the
    // run-time-size form below issues one load per loop trip and waits for it (23 dependent round trips per thread at 19x19:
    // FourRooms FullyObs ran at 2.7 TB/s), and divides by a run-time cell count per unit.
    constexpr int RCELLS = CW * CH;                                       // (0 for the run-time-size instance)
    constexpr int KPR = (RAGGED && RCELLS) ? (64 * RCELLS / 4 + 255) / 256 : 0; // units per thread of a full tile
    uint32_t pr[KPR ? KPR : 1];
    __shared__ uint32_t s_first[64]; // (sized RAGGED) cells 0..3 of every env of the tile: the upper bytes of the one unit per env
                                     // that straddles into the next env
    struct __attribute__((packed)) PU4 { uint32_t v; };
    if constexpr (KPR != 0) {
        const int n_flat_pf = (nv >= 64 ? 64 : (int)nv) * RCELLS;
#pragma unroll
        for (int k = 0; k < KPR; k++) {
            const int f0 = 4 * (tid + 256 * k);
            const int e = f0 / RCELLS, c = f0 - e * RCELLS;              // compile-time divisor
            const bool in = f0 < n_flat_pf;
            const uint8_t *rowp = p.cells + (env0 + (in ? e : 0)) * S;
            pr[k] = reinterpret_cast<const PU4 *>(rowp + (in ? c : 0))->v; // c + 3 <= S - 1: inside the padded row
        }
    }
    s_lut[tid] = decode_triple_full(tid);

    if (tid < 64) { // phase A: wave 0
        const int lane = tid;
        const int64_t env = env0 + lane;
        const bool valid = env < p.n;
        Lane L = unpack_rec(p.agent[env], p.task);
        uint32_t first = 0;
        if constexpr (KPR != 0) first = *reinterpret_cast<const uint32_t *>(p.cells + env * S); // (rows are padded to whole tiles: readable
                                                                                                // for every lane; the line is part of the prefetch)
        uint32_t act = 6;
        if (p.do_step && valid) act = __builtin_nontemporal_load(&p.actions[env]);
        const bool crash = p.task == MGX_TASK_DYNOBS && (act & 0x80u);
        if (p.task == MGX_TASK_DYNOBS) act &= 0x7Fu;
        if (p.task == MGX_TASK_MEMORY && act == 3u) act = 5u;
        constexpr bool BONUS = WRAP; // (as step_body)
        using RewardT = typename std::conditional<BONUS, double, float>::type;
        RewardT reward = 0;
        bool done = false, bad_act = false, oob = false, reset = false;
        bool absorbed = false, absorb_now = false; // DACWrapper (as step_body)
        if constexpr (BONUS) {
            if (p.dac) { const uint32_t rx = p.agent[env].x; absorbed = (rx & MGX_REC_ABSORBED) != 0u; L.dirty |= rx & MGX_REC_ABSORBED; }
        }
        uint32_t wr = 0, changed = 0, nb = 0;
        if (p.do_step) {
            const int fidx = transition_begin<CW, CH>(p, L, act, valid && !absorbed, bad_act, oob);
            if (fidx >= 0) {
                const uint32_t fc = p.cells[env * S + fidx], carry0 = L.carry;
                // hidden object state rides only on the run-time-size kernels (mgx_launch_step routes there): pruned from the sized ones
            const bool has_obj = CW == 0 && p.objaux != nullptr;
            const ObjRef obj = {has_obj ? p.objaux + env * S : nullptr, has_obj ? p.objcont + env * S : nullptr, has_obj ? p.objcarry + env : nullptr};
                const uint32_t nc = transition_apply<CH>(p, L, act, fc, reward, done,
                                                         [&](int i) -> uint32_t { return p.cells[env * S + i]; }, oob, fidx, obj);
                if (valid && L.steps >= p.max_steps) done = true;
                if (p.task) task_rule<CH>(p, L, act, reward, done, [&](int i) -> uint32_t { return (i == fidx) ? nc : (uint32_t)p.cells[env * S + i]; }, fidx, fc, carry0, oob);
                if constexpr (BONUS) {
                    if (p.dac && done && L.steps < p.max_steps) { done = false; absorb_now = true; }
                }
                if (nc != fc && !(p.auto_reset && done)) {
                    p.cells[env * S + fidx] = (uint8_t)nc;
                    wr = (uint32_t)fidx | (nc << 16);
                    changed = 1;
                    L.dirty = MGX_REC_DIRTY;
                }
            } else if (valid && L.steps >= p.max_steps) done = true;
            if constexpr (BONUS) {
                if (p.dac && valid && (done || absorb_now)) L.dirty |= MGX_REC_ABSORBED;
            }
            if (crash) { reward = (RewardT)-1; done = true; }
            if constexpr (BONUS) {
                if (p.bonus && valid && !bad_act) reward = exploration_bonus(p, env, L, p.task == MGX_TASK_MEMORY ? (uint32_t)p.actions[env] : act, reward);
            }
            const float reward_out = (float)reward;
            if (p.reward && valid) __builtin_nontemporal_store(reward_out, &p.reward[env]);
            if (p.done && valid) __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &p.done[env]);
            wave_stats(p, valid, done, reward_out, bad_act, oob, lane, tile);
            if (p.auto_reset && valid && done) {
                reset = L.dirty != 0u || p.regen != nullptr || p.bank != nullptr; // else the cells already equal the snapshot: nothing to copy back
                if (p.bank) { // seed schedule: the level of the next seed of the env's list (ReseedWrapper.reset, wrappers.py:24-28)
                    nb = (uint32_t)p.bank[env] + 1u;
                    nb = nb >= (uint32_t)p.n_banks ? 0u : nb;
                    p.bank[env] = (uint8_t)nb;
                }
                const int64_t senv = env + (int64_t)nb * p.bank_envs;
                L = unpack_rec(p.agent0[senv], p.task);
                if (CW == 0) restore_objstate(p, env, senv, reset);
                if (p.regen) p.regen[env] = 1;
            }
            if (valid) p.agent[env] = pack_rec(L, p.task);
        }
        s_info[lane] = (uint32_t)(L.ax * H + L.ay) | ((uint32_t)L.dir << 16) | ((uint32_t)reset << 18) | (changed << 19) | (nb << 20);
        s_wr[lane] = wr;
        if constexpr (KPR != 0) s_first[lane] = first; // (a reset env's consumer reads the snapshot instead)
    }
    __syncthreads();

    struct __attribute__((packed, aligned(4))) Out12 { uint32_t a, b, c; };
    if constexpr (RAGGED) {
        const int cells = (CW && CH) ? CW * CH : p.W * p.H;
        const int nenv = nv >= 64 ? 64 : (int)nv;
        for (int e = 0; e < nenv; e++) { // restore the (few) envs that finished: block-uniform test, coalesced copy
            if (!((s_info[e] >> 18) & 1u)) continue;
            const uint32_t *s0 = reinterpret_cast<const uint32_t *>(p.cells0 + (env0 + e + (int64_t)(s_info[e] >> 20) * p.bank_envs) * S);
            uint32_t *d0 = reinterpret_cast<uint32_t *>(p.cells + (env0 + e) * S);
            for (int i = tid; i < UPE; i += 256) d0[i] = s0[i];
        }
        if (!p.obs) return;
        uint8_t *out = p.obs + env0 * (int64_t)cells * 3;
        const int n_flat = nenv * cells, n_u = (n_flat + 3) >> 2;
        struct __attribute__((packed)) U4 { uint32_t v; };
        if constexpr (KPR != 0) {
#pragma unroll
            for (int k = 0; k < KPR; k++) {
                const int u = tid + 256 * k;
                const int f0 = 4 * u;
                if (f0 >= n_flat) break;
                const int e = f0 / RCELLS, c = f0 - e * RCELLS;
                const int n_lo = RCELLS - c < 4 ? RCELLS - c : 4;        // cells of this unit that belong to env e (1..4)
                uint32_t w = pr[k];
                {
                    const uint32_t info = s_info[e];
                    if ((info >> 18) & 1u) w = reinterpret_cast<const U4 *>(p.cells0 + (env0 + e + (int64_t)(info >> 20) * p.bank_envs) * S + c)->v; // env e was reset: its snapshot (rare)
                    if ((info >> 19) & 1u) {
                        const uint32_t x = s_wr[e], d = (x & 0xFFFFu) - (uint32_t)c;
                        if (d < 4u) w = (w & ~(0xFFu << (8u * d))) | ((x >> 16) << (8u * d));
                    }
                    const uint32_t d = (info & 0xFFFFu) - (uint32_t)c, code = MGX_K_AGENT | (((info >> 16) & 3u) << 4);
                    if (d < 4u) w = (w & ~(0xFFu << (8u * d))) | (code << (8u * d));
                }
                if (n_lo < 4) { // the rest of the unit: the first 4 - n_lo cells of env e + 1 (or nothing past the tile's last cell)
                    uint32_t w2;
                    if (f0 + n_lo < n_flat) {
                        const uint32_t info = s_info[e + 1];
                        w2 = ((info >> 18) & 1u) ? *reinterpret_cast<const uint32_t *>(p.cells0 + (env0 + e + 1 + (int64_t)(info >> 20) * p.bank_envs) * S) : s_first[e + 1];
                        if ((info >> 19) & 1u) {
                            const uint32_t x = s_wr[e + 1], d = x & 0xFFFFu;
                            if (d < 4u) w2 = (w2 & ~(0xFFu << (8u * d))) | ((x >> 16) << (8u * d));
                        }
                        const uint32_t d = info & 0xFFFFu, code = MGX_K_AGENT | (((info >> 16) & 3u) << 4);
                        if (d < 4u) w2 = (w2 & ~(0xFFu << (8u * d))) | (code << (8u * d));
                    } else w2 = 0x01010101u * MGX_CODE_EMPTY;
                    const uint32_t keep = (1u << (8 * n_lo)) - 1u;        // n_lo in 1..3
                    w = (w & keep) | (w2 << (8 * n_lo));
                }
                const uint32_t t0 = s_lut[w & 255u], t1 = s_lut[(w >> 8) & 255u], t2 = s_lut[(w >> 16) & 255u], t3 = s_lut[w >> 24];
                const uint32_t a = __builtin_amdgcn_perm(t1, t0, 0x04020100u), b = __builtin_amdgcn_perm(t2, t1, 0x05040201u),
                               cc = __builtin_amdgcn_perm(t3, t2, 0x06050402u);
                if (f0 + 4 <= n_flat) nt_store12(reinterpret_cast<uint32_t *>(out + 12 * (int64_t)u), a, b, cc);
                else { // tail tile whose cell count is not a multiple of 4: the last unit is short
                    const uint32_t wds[3] = {a, b, cc};
                    for (int b8 = 0; b8 < 3 * (n_flat - f0); b8++) out[12 * (int64_t)u + b8] = (uint8_t)(wds[b8 >> 2] >> (8 * (b8 & 3)));
                }
            }
            return;
        }
        for (int u = tid; u < n_u; u += 256) {
            const int f0 = 4 * u;
            int e = f0 / cells, c = f0 - e * cells;
            uint32_t tr[4];
            if (c + 3 < cells) { // the whole unit inside one env: one unaligned dword
                const uint32_t info = s_info[e];
                const uint8_t *rowp = ((info >> 18) & 1u) ? p.cells0 + (env0 + e + (int64_t)(info >> 20) * p.bank_envs) * S : p.cells + (env0 + e) * S;
                uint32_t w = reinterpret_cast<const U4 *>(rowp + c)->v;
                if ((info >> 19) & 1u) {
                    const uint32_t x = s_wr[e], d = (x & 0xFFFFu) - (uint32_t)c;
                    if (d < 4u) w = (w & ~(0xFFu << (8u * d))) | ((x >> 16) << (8u * d));
                }
                {
                    const uint32_t d = (info & 0xFFFFu) - (uint32_t)c, code = MGX_K_AGENT | (((info >> 16) & 3u) << 4);
                    if (d < 4u) w = (w & ~(0xFFu << (8u * d))) | (code << (8u * d));
                }
                tr[0] = s_lut[w & 255u]; tr[1] = s_lut[(w >> 8) & 255u]; tr[2] = s_lut[(w >> 16) & 255u]; tr[3] = s_lut[w >> 24];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t code = MGX_CODE_EMPTY;
                    if (f0 + j < n_flat) {
                        const uint32_t info = s_info[e];
                        code = (((info >> 18) & 1u) ? p.cells0 + (env0 + e + (int64_t)(info >> 20) * p.bank_envs) * S : p.cells + (env0 + e) * S)[c];
                        if (((info >> 19) & 1u) && (s_wr[e] & 0xFFFFu) == (uint32_t)c) code = s_wr[e] >> 16;
                        if ((info & 0xFFFFu) == (uint32_t)c) code = MGX_K_AGENT | (((info >> 16) & 3u) << 4);
                    }
                    tr[j] = s_lut[code];
                    if (++c == cells) { c = 0; e++; }
                }
            }
            const uint32_t a = __builtin_amdgcn_perm(tr[1], tr[0], 0x04020100u), b = __builtin_amdgcn_perm(tr[2], tr[1], 0x05040201u),
                           cc = __builtin_amdgcn_perm(tr[3], tr[2], 0x06050402u);
            if (f0 + 4 <= n_flat) nt_store12(reinterpret_cast<uint32_t *>(out + 12 * (int64_t)u), a, b, cc);
            else { // tail tile whose cell count is not a multiple of 4: the last unit is short
                const uint32_t wds[3] = {a, b, cc};
                for (int b8 = 0; b8 < 3 * (n_flat - f0); b8++) out[12 * (int64_t)u + b8] = (uint8_t)(wds[b8 >> 2] >> (8 * (b8 & 3)));
            }
        }
        return;
    }
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
        else w = reinterpret_cast<const uint32_t *>(rst ? p.cells0 + (env0 + (int64_t)(info >> 20) * p.bank_envs) * S : p.cells + env0 * S)[u];
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
// mgx_rollout, fused form: T consecutive steps in ONE launch.  The caller's `for t: env.step(a[t])` loop (run_tests.py:41-68,
// benchmark.py:45-46) with the actions of all T steps given up front has no step-to-step dependency outside an env, so a wave
// keeps its tile -- the 64 grids in LDS, the 64 agent records in registers -- across the whole loop: per env-step only the
// action (1 B) comes in and the observation, reward and done flag (152 B) go out; cells and records are read once and written
// back once per launch, and there is no launch boundary between steps (the per-step launches pay ~8 us each: DESIGN.md section 4).
// The observation image gets LDS of its own behind the grid image (it overlays the grid in k_step).  Default visibility, no hidden
// object state, no epilogue, no new_level_each_episode / Dynamic-Obstacles (those interleave other kernels with the steps): partial
// views of every size on grids up to 16x16, the FullyObs observation on grids up to 13x13; everything else keeps the captured graph
// of per-step launches (mgx_rollout).
struct RolloutParams {
    const uint8_t *actions; // u8[T][n]
    uint8_t *obs;           // u8[T][n][147] or null
    float *reward;          // f32[T][n] or null
    uint8_t *done;          // u8[T][n] or null
    int64_t T;
    int grid_lds;           // bytes of the grid image per wave (the observation image follows)
};

// CW = CH = 0: run-time grid size (any size whose tile image + observation image fit the LDS); V: agent_view_size (7 for the sized
// instances, 3 / 5 / 9 / 11 on the run-time-size one: round 3 -- those handles took the captured graph before).
// FULL: the FullyObsWrapper observation (W x H x 3 per env and step, emit_full_obs on the resident tile: the agent's marker goes into the
// LDS image for the length of the emission and comes out again) instead of the partial view.
template <int CW, int CH, int V, bool FULL = false>
__global__ __launch_bounds__(256) void k_rollout(const StepParams p, const RolloutParams q)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= p.n_tiles) return; // wave-uniform
    constexpr int CS = (CW * CH + 3) & ~3;
    constexpr int CLS = CS + (((CS >> 2) & 1) ? 0 : 4);
    const int S = CS ? CS : p.S, LS = CS ? CLS : p.LS;
    uint8_t *lds = smem + p.lds_guard + (size_t)wv * p.wave_lds; // p.wave_lds = grid image + observation image (mgx_launch_rollout)
    uint8_t *img = lds + q.grid_lds;
    const int64_t env0 = (int64_t)tile * 64, env = env0 + lane;
    const bool valid = env < p.n;
    constexpr int B = 3 * V * V;

    stage_tile<CS>(p.cells, env0, S, LS, lds, lane);
    Lane L = unpack_rec(p.agent[env], p.task);
    uint8_t *g = lds + lane * LS;
    wave_sync();
    bool wrote = false; // some cell of this lane's env differs from what p.cells holds
    uint32_t act_next = valid ? (uint32_t)q.actions[env] : 6u;
    for (int64_t t = 0; t < q.T; t++) {
        uint32_t act = act_next;
        if (t + 1 < q.T && valid) act_next = q.actions[(t + 1) * p.n + env]; // in flight during this step
        if (p.task == MGX_TASK_MEMORY && act == 3u) act = 5u;
        float reward = 0.f;
        bool done = false, bad_act = false, oob = false;
        const int fidx = transition_begin<CW, CH>(p, L, act, valid, bad_act, oob);
        uint32_t fc = 0, nc = 0;
        if (fidx >= 0) {
            const uint32_t carry0 = L.carry;
            fc = g[fidx];
            const ObjRef obj = {nullptr, nullptr, nullptr};
            nc = transition_apply<CH>(p, L, act, fc, reward, done, [&](int i) -> uint32_t { return g[i]; }, oob, fidx, obj);
            if (nc != fc) g[fidx] = (uint8_t)nc;
            if (valid && L.steps >= p.max_steps) done = true;
            if (p.task) task_rule<CH>(p, L, act, reward, done, [&](int i) -> uint32_t { return (uint32_t)g[i]; }, fidx, fc, carry0, oob);
            if (nc != fc && !(p.auto_reset && done)) { L.dirty = MGX_REC_DIRTY; wrote = true; }
        } else if (valid && L.steps >= p.max_steps) done = true;
        const float reward_out = (float)reward;
        if (q.reward && valid) __builtin_nontemporal_store(reward_out, &q.reward[t * p.n + env]);
        if (q.done && valid) __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &q.done[t * p.n + env]);
        wave_stats(p, valid, done, reward, bad_act, oob, lane, tile);
        if (p.auto_reset && valid && done) {
            if (L.dirty != 0u) { // back to the episode-start snapshot (LDS only: the tile goes home once, after the last step)
                const uint32_t *s0 = reinterpret_cast<const uint32_t *>(p.cells0 + env * S);
                uint32_t *l32 = reinterpret_cast<uint32_t *>(g);
                if constexpr (CS != 0) {
                    uint32_t v[CS ? CS / 4 : 1];
#pragma unroll
                    for (int i = 0; i < CS / 4; i++) v[i] = s0[i];
#pragma unroll
                    for (int i = 0; i < CS / 4; i++) l32[i] = v[i];
                } else {
#pragma unroll 8
                    for (int i = 0; i < (S >> 2); i++) l32[i] = s0[i];
                }
                wrote = true;
            } else if (nc != fc) g[fidx] = (uint8_t)fc;
            L = unpack_rec(p.agent0[env], p.task);
        }
        if (q.obs) {
            StepParams po = p;
            if constexpr (FULL) {
                const int H = CH ? CH : p.H, W = CW ? CW : p.W;
                po.obs = q.obs + t * p.n * (int64_t)(3 * W * H);
                const int aidx = L.ax * H + L.ay;
                const uint8_t under = g[aidx]; // (what the agent stands on: an empty cell, an open door, the goal)
                emit_full_obs<CW, CH>(po, L, valid, lds, g, LS, env0, lane);
                wave_sync();
                if (valid) g[aidx] = under;
                wave_sync();
            } else {
                po.obs = q.obs + t * p.n * B;
                emit_partial_obs<CW, CH, V, false>(po, L, img, g, env0, lane);
                wave_sync(); // the image's readers are done before the next step's LDS writes (same wave: program order)
            }
        }
    }
    if (valid) p.agent[env] = pack_rec(L, p.task);
    wave_sync();
    if (__ballot(wrote)) unstage_tile<CS>(p.cells, env0, S, LS, lds, lane); // (wave-uniform; Empty / Crossing tiles never change)
}

// ---- which instantiation runs a handle's step: one selector for the launch, the LDS limit and the residency query ----------------
using StepKernel = void (*)(const StepParams);
struct StepChoice {
    StepKernel fn;
    bool block_per_tile; // the FullyObs direct forms: a 256-thread block per tile, no dynamic LDS
    const char *name;    // the instantiation, as rocprofv3 prints it (mgx_step_kernel_name: bench.py's roofline.kernel)
};
#define MGX_STR_(x) #x
#define MGX_STR(x) MGX_STR_(x)

template <int CW, int CH>
StepChoice choose_sized(const StepParams &p, int mode)
{
    struct Names { // (one per <CW, CH>, written once: a function-local static is initialised under the language's own lock)
        char n0[48], n1[48], n2[48], n3[48];
        Names()
        {
            snprintf(n0, sizeof n0, "k_step<%d,%d,0,7>", CW, CH);
            snprintf(n1, sizeof n1, "k_step<%d,%d,1,7>", CW, CH);
            snprintf(n2, sizeof n2, "k_step_fulldirect<%d,%d>", CW, CH);
            snprintf(n3, sizeof n3, "k_step_fulldirect<%d,%d,ragged>", CW, CH);
        }
    };
    static const Names nm;
    const char *n0 = nm.n0, *n1 = nm.n1, *n2 = nm.n2, *n3 = nm.n3;
    if (mode == 0) return {k_step<CW, CH, 0, 7>, false, n0};
    if (mode == 1) return {k_step<CW, CH, 1, 7>, false, n1};
    if (((CW && CH) ? CW * CH : p.W * p.H) % 4 == 0) return {k_step_fulldirect<CW, CH>, true, n2};
    if (CW == 0 && !p.objaux && p.W == 19 && p.H == 19) return {k_step_fulldirect<19, 19, true>, true, "k_step_fulldirect<19,19,ragged>"};
    if (CW == 0 && !p.objaux && p.W == 25 && p.H == 25) return {k_step_fulldirect<25, 25, true>, true, "k_step_fulldirect<25,25,ragged>"};
    return {k_step_fulldirect<CW, CH, true>, true, n3};
}

} // namespace

#define MGX_SIZED(X) X(5, 5) X(6, 6) X(7, 7) X(8, 8) X(9, 9) X(11, 11) X(16, 16) X(10, 10) X(13, 13) X(16, 8) X(12, 6) X(11, 6)
#define MGX_VIEWS(X) X(3) X(5) X(9) X(11)  /* agent_view_size other than 7: run-time grid size only */

// The step kernels are their own code object; HIP loads it on first use (18 ms measured).  mgx_create asks for it up
// front so that the first mgx_step is not the one that pays.
hipError_t mgx_preload_step_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_step<0, 0, 0, 7>));
}

// Fused T-step rollout for partial-view handles with the default visibility: sized instances for the 7x7 view, the run-time-size one
// for every other grid and for agent_view_size 3 / 5 / 9 / 11; returns hipErrorNotSupported when the two LDS images of a wave do not fit.
hipError_t mgx_launch_rollout(const StepParams &p0, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done, int64_t T, int full, hipStream_t st)
{
    StepParams p = p0;
    const int CS = (p.W * p.H + 3) & ~3, LS = CS + (((CS >> 2) & 1) ? 0 : 4);
    RolloutParams q;
    q.actions = actions; q.obs = obs; q.reward = reward; q.done = done; q.T = T;
    q.grid_lds = (64 * LS + 15) & ~15;
    p.wave_lds = (q.grid_lds + 32 * 3 * p.view * p.view + 15) & ~15; // + the half-tile observation image (4,704 B for the 7x7 view)
    if (full) p.wave_lds = q.grid_lds + 3072 + 16;                  // + emit_full_obs' 3 KiB transpose scratch behind the grid image
    const int LDS_DEFAULT = 64 * 1024, LDS_MAX = 160 * 1024;
    int wpb = (LDS_DEFAULT - p.lds_guard) / p.wave_lds;
    if (wpb > 4) wpb = 4;
    const bool raise = wpb < 1;
    if (raise) { wpb = 1; if (p.wave_lds + p.lds_guard > LDS_MAX) return hipErrorNotSupported; }
    const dim3 block(64 * wpb), grid((p.n_tiles + wpb - 1) / wpb);
    const size_t shmem = (size_t)wpb * p.wave_lds + p.lds_guard; // (guard in front; behind the last grid image lies its observation image)
#define LAUNCH(KERN)                                                                                                                       \
    do {                                                                                                                                   \
        if (raise) {                                                                                                                       \
            const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
            if (e_ != hipSuccess) return e_;                                                                                               \
        }                                                                                                                                  \
        hipLaunchKernelGGL(KERN, grid, block, shmem, st, p, q);                                                                            \
        return hipGetLastError();                                                                                                          \
    } while (0)
    if (full) { // (the view size does not enter a full-grid observation: one instance per grid size)
#define CASE(w, h) if (p.W == w && p.H == h) LAUNCH((k_rollout<w, h, 7, true>));
        MGX_SIZED(CASE)
#undef CASE
        LAUNCH((k_rollout<0, 0, 7, true>));
    }
    if (p.view == 7) {
#define CASE(w, h) if (p.W == w && p.H == h) LAUNCH((k_rollout<w, h, 7>));
        MGX_SIZED(CASE)
#undef CASE
        LAUNCH((k_rollout<0, 0, 7>));
    }
#define VCASE(v) if (p.view == v) LAUNCH((k_rollout<0, 0, v>));
    MGX_VIEWS(VCASE)
#undef VCASE
#undef LAUNCH
    return hipErrorNotSupported;
}

// handles with an exploration bonus or the DAC wrapper: the instances that carry those paths (run-time grid size)
static StepChoice choose_wrap_kernel(const StepParams &p, int mode)
{
    const StepChoice none = {nullptr, false, "none"};
    if (p.onehot) return none; // (the fused one-hot form is an opt-in of its own: mgx_add_bonus / mgx_set_dac refuse there)
    if (mode == 0 || mode == 3) {
#define WCASE(v) if (p.view == v) { \
        if (mode == 0) return p.alt_vis ? StepChoice{k_step_wrap<0, v, true>, false, "k_step_wrap<0," MGX_STR(v) ",alt>"} : StepChoice{k_step_wrap<0, v>, false, "k_step_wrap<0," MGX_STR(v) ">"}; \
        return p.alt_vis ? StepChoice{k_step_wrap<3, v, true>, false, "k_step_wrap<3," MGX_STR(v) ",alt>"} : StepChoice{k_step_wrap<3, v>, false, "k_step_wrap<3," MGX_STR(v) ">"}; }
        MGX_VIEWS(WCASE) WCASE(7)
#undef WCASE
        return none;
    }
    if (mode == 1) return {k_step_wrap<1, 7>, false, "k_step_wrap<1,7>"};
    if ((p.W * p.H) % 4 == 0) return {k_step_fulldirect<0, 0, false, true>, true, "k_step_fulldirect_wrap"};
    return {k_step_fulldirect<0, 0, true, true>, true, "k_step_fulldirect_wrap<ragged>"};
}

static StepChoice choose_step_kernel(const StepParams &p, int mode)
{
    const StepChoice none = {nullptr, false, "none"};
    if (p.bonus | p.dac) return choose_wrap_kernel(p, mode);
    if (p.onehot) { // (mgx_create sets it for partial views up to 7x7 with the default visibility only)
        if (p.alt_vis || (mode != 0 && mode != 3)) return none;
#define OCASE(v) if (p.view == v) return mode == 0 ? StepChoice{k_step_onehot<0, v>, false, "k_step_onehot<0," MGX_STR(v) ">"} : StepChoice{k_step_onehot<3, v>, false, "k_step_onehot<3," MGX_STR(v) ">"};
        OCASE(3) OCASE(5) OCASE(7)
#undef OCASE
        return none;
    }
    if (mode == 3) { // large grids: gather form (the default view and visibility by the size rule, anything else when the tile image cannot fit the LDS)
        if (p.view == 7 && !p.alt_vis && !p.objaux) { // the default view: 13x13 Memory, every 16x16 id, 17x17 Memory, FourRooms / LockedRoom / Playground 19x19, MultiRoom 25x25, any other size
#define GCASE(w, h) if (p.W == w && p.H == h) return {k_step<w, h, 3, 7, false, false>, false, "k_step<" #w "," #h ",3,7>"};
            GCASE(13, 13) GCASE(16, 16) GCASE(17, 17) GCASE(19, 19) GCASE(25, 25)
#undef GCASE
            return {k_step<0, 0, 3, 7, false, false>, false, "k_step<0,0,3,7>"};
        }
        if (p.view == 7 && !p.alt_vis) { // ... with the hidden Goal / Box planes (ObstructedMaze 2Dl / 2Dlh / 2Dlhb / 1Q / 2Q / Full are 16x16)
            if (p.W == 16 && p.H == 16) return {k_step<16, 16, 3, 7, false, true>, false, "k_step<16,16,3,7,obj>"};
            return {k_step<0, 0, 3, 7, false, true>, false, "k_step<0,0,3,7,obj>"};
        }
#define VCASE(v) if (p.view == v) return p.alt_vis ? StepChoice{k_step<0, 0, 3, v, true>, false, "k_step<0,0,3," MGX_STR(v) ",alt>"} : StepChoice{k_step<0, 0, 3, v>, false, "k_step<0,0,3," MGX_STR(v) ",obj>"};
        MGX_VIEWS(VCASE) VCASE(7)
#undef VCASE
        return none;
    }
    if (mode == 0 && p.alt_vis) { // default_vis=False: run-time grid size only
#define VCASE(v) if (p.view == v) return {k_step<0, 0, 0, v, true>, false, "k_step<0,0,0," MGX_STR(v) ",alt>"};
        MGX_VIEWS(VCASE) VCASE(7)
#undef VCASE
        return none;
    }
    if (mode == 0 && p.view != 7) {
#define VCASE(v) if (p.view == v) return {k_step<0, 0, 0, v>, false, "k_step<0,0,0," MGX_STR(v) ">"};
        MGX_VIEWS(VCASE)
#undef VCASE
        return none;
    }
    if (p.objaux && mode == 0 && p.view == 7) { // ObstructedMaze's grids with their boxed keys: sized instances that keep the plane accesses
        if (p.W == 11 && p.H == 6) return {k_step<11, 6, 0, 7, false, true>, false, "k_step<11,6,0,7,obj>"};
        if (p.W == 16 && p.H == 16) return {k_step<16, 16, 0, 7, false, true>, false, "k_step<16,16,0,7,obj>"};
    }
    if (p.objaux) return choose_sized<0, 0>(p, mode); // (hidden object state and exploration bonuses: the run-time-size instances)
#define CASE(w, h) if (p.W == w && p.H == h) return choose_sized<w, h>(p, mode);
    MGX_SIZED(CASE)
#undef CASE
    return choose_sized<0, 0>(p, mode);
}

struct StepShape { dim3 grid, block; size_t shmem; };
static StepShape step_shape(const StepParams &p, const StepChoice &c, int waves_per_block)
{
    if (c.block_per_tile) return {dim3(p.n_tiles), dim3(256), 0};
    return {dim3((p.n_tiles + waves_per_block - 1) / waves_per_block), dim3(64 * waves_per_block), (size_t)waves_per_block * p.wave_lds + 2 * (size_t)p.lds_guard};
}

// Blocks of this handle's step kernel that are resident at once on the whole chip (one "round" of the grid).
hipError_t mgx_step_round_blocks(const StepParams &p, int mode, int waves_per_block, int *blocks)
{
    const StepChoice c = choose_step_kernel(p, mode);
    if (!c.fn) return hipErrorInvalidValue;
    const StepShape sh = step_shape(p, c, waves_per_block);
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(c.fn), (int)sh.block.x, sh.shmem);
    if (e != hipSuccess) return e;
    *blocks = per_cu * cus;
    return hipSuccess;
}

// Launch shaping of a handle, fixed at mgx_create from the handle's OWN device: how many of a grid's last blocks
// run at raised wave priority -- two per CU (measured best of 0.5 / 2 / 3.5 / 7 per CU and "the last partial round") -- and the
// first-round stagger per SIMD wave slot in units of `s_sleep 4` (256 clocks).  MGX_TAIL_BLOCKS / MGX_STAGGER / MGX_STAGGER_MIN override
// them for tuning runs (0 = off; read once per handle, never needed for correctness).
hipError_t mgx_step_launch_cfg(int device, StepLaunchCfg *out)
{
    int cus = 256;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    const char *t = MGX_TUNE_ENV("MGX_TAIL_BLOCKS"), *s = MGX_TUNE_ENV("MGX_STAGGER"), *m = MGX_TUNE_ENV("MGX_STAGGER_MIN");
    out->tail_blocks = t ? atoi(t) : 2 * cus;
    out->stagger_units = s ? atoi(s) : 5;
    out->stagger_min = m ? atoi(m) : -1; // (tuning runs: the smallest grid that staggers; default: one round of resident blocks)
    return hipSuccess;
}

hipError_t mgx_launch_step(const StepParams &p0, int mode, int waves_per_block, const StepLaunchCfg &lc, hipStream_t st)
{
    const StepChoice c = choose_step_kernel(p0, mode);
    if (!c.fn) return hipErrorInvalidValue;
    const StepShape sh = step_shape(p0, c, waves_per_block);
    StepParams p = p0;
    const int blocks = (int)sh.grid.x, tb = lc.tail_blocks * 4 / (c.block_per_tile ? 4 : waves_per_block); // (counted in 4-wave blocks)
    p.tail_block0 = (tb > 0 && blocks > tb) ? blocks - tb : 0x7fffffff;
    p.stagger = (!c.block_per_tile && p.round_blocks > 0 && blocks > (lc.stagger_min >= 0 ? lc.stagger_min : p.round_blocks)) ? lc.stagger_units : 0;
#ifdef MGX_TIMELINE
    {   // launch number MGX_TL_LAUNCH (default 300) of the process records its waves; the next launch writes them to MGX_TL_FILE
        static unsigned long long *tl = nullptr;
        static int launches = 0;
        const char *f = getenv("MGX_TL_FILE");
        const int at = getenv("MGX_TL_LAUNCH") ? atoi(getenv("MGX_TL_LAUNCH")) : 300;
        if (f && !tl) { if (hipMalloc(&tl, (size_t)p.n_tiles * 64) != hipSuccess) return hipErrorOutOfMemory; (void)hipMemset(tl, 0, (size_t)p.n_tiles * 64); }
        p.timeline = (f && launches == at) ? tl : nullptr;
        if (f && launches == at + 1) {
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> hbuf((size_t)p.n_tiles * 8);
            (void)hipMemcpy(hbuf.data(), tl, hbuf.size() * 8, hipMemcpyDeviceToHost);
            FILE *fp = fopen(f, "wb");
            if (fp) { fwrite(hbuf.data(), 8, hbuf.size(), fp); fclose(fp); }
        }
        launches++;
    }
#endif
    hipLaunchKernelGGL(c.fn, sh.grid, sh.block, sh.shmem, st, p);
    return hipGetLastError();
}

// Dynamic-Obstacles, staged partial form (7x7 view, default visibility): walk + step in one launch.  p.wave_lds is the step's; the walk's
// extra LDS (position words, tape, strip) lies behind the tile image and is dead by the time the observation image overlays it.
hipError_t mgx_launch_step_dyn(const StepParams &p0, const DynObsParams &d, const StepLaunchCfg &lc, hipStream_t st)
{
    StepParams p = p0;
    if (d.wave_lds > p.wave_lds) p.wave_lds = (d.wave_lds + 15) & ~15;
    int wpb = (64 * 1024 - 2 * p.lds_guard) / p.wave_lds;
    if (wpb > 4) wpb = 4;
    if (wpb < 1) return hipErrorNotSupported;
    if (p.wave_lds > 8192) wpb = 1; // (as k_dynobs: more one-wave blocks fit a CU's 160 KB, and a block's LDS is free as soon as its wave is done)
    const int blocks = (p.n_tiles + wpb - 1) / wpb, tb = lc.tail_blocks * 4 / wpb;
    p.tail_block0 = (tb > 0 && blocks > tb) ? blocks - tb : 0x7fffffff;
    p.stagger = 0; // (the walk's own loads come first and spread the waves by themselves)
    const size_t shmem = (size_t)wpb * p.wave_lds + 2 * (size_t)p.lds_guard;
    if (p.bonus | p.dac) { hipLaunchKernelGGL(k_step_dyn_wrap, dim3(blocks), dim3(64 * wpb), shmem, st, p, d); return hipGetLastError(); }
#define CASE(w, h) if (p.W == w && p.H == h) { hipLaunchKernelGGL((k_step_dyn<w, h>), dim3(blocks), dim3(64 * wpb), shmem, st, p, d); return hipGetLastError(); }
    CASE(5, 5) CASE(6, 6) CASE(8, 8) CASE(16, 16) // the registered Dynamic-Obstacles sizes
#undef CASE
    hipLaunchKernelGGL((k_step_dyn<0, 0>), dim3(blocks), dim3(64 * wpb), shmem, st, p, d);
    return hipGetLastError();
}

const char *mgx_step_kernel_label(const StepParams &p, int mode) { return choose_step_kernel(p, mode).name; }

hipError_t mgx_raise_lds_limit(const StepParams &p, int mode, int bytes)
{
    const StepChoice c = choose_step_kernel(p, mode);
    if (!c.fn) return hipErrorInvalidValue;
    if (c.block_per_tile) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

