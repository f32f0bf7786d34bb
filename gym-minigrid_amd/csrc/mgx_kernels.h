// mgx_kernels.h -- launch interface between mgx_api.cpp (host) and the kernel files k_*.hip (device).
#ifndef MGX_KERNELS_H
#define MGX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx.h"

#define MGX_NUM_ACTIONS_K 7u
#define MGX_LG_LDS_PER_WAVE_BYTES (2 * 624 * 4 + 16 + 2 * (6 * 32) + 8 * (8 + 4 * 32)) /* == MGX_LG_LDS_PER_WAVE in k_levelgen.hip */

// Running totals.  The per-step counters are SHARDED over 256 cache lines (shard = tile & 255): with one word,
// LavaCrossing (40% of the waves see a done every step) spent >half of the step serialising ~3,300 same-address
// atomics in L2.  Readers sum the shards.
#define MGX_CTR_SHARDS 256
struct MgxCounterShard {
    unsigned long long episodes;
    double reward_sum;
    unsigned long long invalid_actions, out_of_bounds; // sharded as well: TwoGoals refuses pickup / drop, two actions in seven of a random
                                                       // policy -- every wave of every step then hit ONE word (524,288 envs: 208 us per step)
    unsigned long long pad[4]; // one 64-byte line per shard
};
struct MgxCounters {
    MgxCounterShard shard[MGX_CTR_SHARDS];
    unsigned long long invalid_actions, out_of_bounds, invalid_state; // (the first two: filled by the host from the shards, read_counters)
};

// on-device level generation (new level each episode)
struct LevelGenParams {
    mgx_config cfg;
    uint32_t *mt;      // u32[n_pad][624]  per-env MT19937 block
    uint32_t *mt2;     // new_level_each_episode handles (else null): u32[n_pad][624], always twist(mt) = the block AFTER `mt`, kept ready by
                       // whoever advances an env's block, so that a level may run across the end of `mt` on the lane-per-level path
    uint32_t *mt_idx;  // u32[n_pad]       next unread word of the block (624 = exhausted / freshly seeded)
    // Handles whose RNG state is rarely needed beyond the first level (every episode re-seeded: ReseedWrapper semantics, no stream mode,
    // no Dynamic-Obstacles; families whose levels take a few dozen draws) keep a VIRTUAL state per env after seed(): the seed itself
    // (seed0) plus the first MGX_SEED_WIN words of the first block (`win`), which is all a level reads -- 256 B per env instead of a
    // 2.5 KB block written, read back for the twist and written again.  virt[env] = 1 while that holds; whoever needs more (a level that
    // runs past the window: levelgen_one; the plain reset(): k_seed_masked<MATERIALIZE>) re-derives the block from seed0 into `mt` first.
    uint32_t *win;     // u32[n_pad][MGX_SEED_WIN] (+ slack) or null
    uint8_t *virt;     // u8[n_pad] or null
    const uint64_t *seed0;   // u64[n_pad] the seed behind a virtual state
    const uint32_t *mt_init; // init_genrand(19650218)'s 624 words
    uint8_t *regen;    // u8[n_pad]        work flags, cleared here
    uint8_t *regen_more[3]; // ... and up to three more arrays of them, later steps of the same run of a ring (n_regen arrays in all, oldest first;
    int n_regen;       //      0 / 1: just `regen`): one launch takes them together, an env's levels in array order
    uint8_t *cells0;   // next-level buffer (codes) and its agent record
    uint2 *agent0;
    uint8_t *objaux0, *objcont0; // object_state handles (else null): the next level's hidden-state planes -- aux all zero,
                                 // contains = what the generator put into boxes (ObstructedMaze's keys)
    MgxCounters *ctr;
    int64_t n;
    int n_tiles, S;
    int64_t bank_envs; // handles with a ring of next-level buffers: n_pad -- a work flag of b + 1 means buffer b, at env + b * bank_envs (else 0)
};
// layout of k_levelgen's lane slices and queues in LDS, sized per family by mgx_levelgen_layout (lanes: generating lanes per fast wave)
struct FastLayout { int cmd_cap, river_cap, img_dw, slice_dw, n_fast_waves, span, lanes, queue_off; };
struct DynObsParams;
struct StepParams {
    uint8_t *cells;        // u8[n_pad][S]   internal cell codes, x-major
    uint2 *agent;          // [n_pad]        x | y<<8 | dir<<16 | carry<<24 ; step_count
    const uint8_t *cells0; // episode-start snapshot (auto-reset)
    const uint2 *agent0;
    const uint8_t *actions; // u8[n]
    uint8_t *obs;           // u8[n][147] or u8[n][W*H*3]; may be null
    float *reward;          // may be null
    uint8_t *done;          // may be null
    MgxCounters *ctr;
    // hidden Goal/Box state (object_state handles, else null): planes, snapshots, carried object's pair
    uint8_t *objaux, *objcont;
    const uint8_t *objaux0, *objcont0;
    uint16_t *objcarry;
    uint8_t *regen;        // stream mode: set to 1 for every env that consumed its next-level buffer; Dynamic-Obstacles: the
                           // handle's own `restart` flags for k_dynobs (else null)
    uint8_t *front;        // gather form only (else null): u8[n_pad], cell code in front of the agent as of the env's last observation pass;
                           // 0 = unknown (every entry point that changes cells or poses outside the step kernel clears it)
    uint8_t *wcache;       // gather form, 7x7 view, default visibility (else null): u8[n_pad][64], the view's window excerpt as the env's last
                           // observation pass loaded it -- 7 columns x 8 rows -- and, in the last dword, the pose it belongs to (x | y << 8 |
                           // dir << 16 | 1 << 24; 0 = none).  A step that leaves the pose as it is (turn-free, move-free: 4 of 7 random actions, and
                           // every blocked forward) observes from this ONE coalesced 64-byte read instead of seven unaligned loads that cost two
                           // 128-byte lines.  Cleared wherever `front` is.
    const uint8_t *obs_mask; // observe after a masked reset: 64-env tiles without a masked env are skipped (else null)
    int onehot;            // partial view only: 1 = `obs` receives the OneHotPartialObsWrapper image (V*V*21 bytes per env) straight from the
                           // step kernel; wave_lds then holds the parked cell codes + a 16-env quarter of that image (mgx_create)
    // seed schedule (mgx_set_seed_schedule: ReseedWrapper with a list of K seeds, wrappers.py:12-28), else bank == null: every snapshot array
    // (cells0, agent0, objaux0, objcont0) is [K][n_pad][...], bank k = the episode start under the k-th seed of the env's list.  An
    // in-kernel reset of env e advances bank[e] (mod K) and restores from snapshot index e + bank[e] * bank_envs.
    uint8_t *bank;         // u8[n_pad]: list index of the env's CURRENT episode
    int n_banks;           // K
    int64_t bank_envs;     // n_pad (0 without a schedule)
    // exploration bonuses (mgx_add_bonus): up to two stacked wrappers, innermost in bits 3:0 of `bonus` (MGX_BONUS_*), the next in bits 7:4
    int dac;               // 1: DACWrapper (mgx_set_dac): envs done before their time-out are absorbed, not reset
    int bonus, bonus_na;   // bonus_na: actions per (cell, dir) in the ActionBonus counts (7, or 9 with extended_actions)
    uint32_t *bonus_action; // u32[n_pad][W*H*4*bonus_na]
    uint32_t *bonus_state;  // u32[n_pad][W*H]
    int ring;              // R > 0 (a power of two): the banks are a RING of R next-level buffers (new_level_each_episode handles whose generator runs beside
                           // the steps): bank[e] names the buffer the next reset consumes; the reset moves it on and raises regen[e] = buffer + 1
    int64_t n;
    int n_tiles;
    int W, H, S, LS, wave_lds, view;
    int max_steps, see_through, lava_v1, auto_reset, do_step, extended, alt_vis, task;
#ifdef MGX_TIMELINE     // profiling build only (tools/build_variant.sh tl -DMGX_TIMELINE=1): per-wave phase timestamps
    unsigned long long *timeline; // [n_tiles][8]: s_memrealtime at entry / tile staged / transition done / obs computed / stores issued, HW_ID, XCC_ID, block
#endif
    int lds_guard;         // bytes kept free in front of and behind the block's tile images (staged partial form): the view gather
                           // reads cells outside the grid where their index points -- up to (V-1)*H + V/2 bytes off the env's row --
                           // and replaces them by the wall afterwards; the guard keeps those reads inside the allocation
    int tail_block0;       // blocks from this index on (the last two per CU of the grid: mgx_launch_step) run at wave priority 3
    int round_blocks;      // blocks of this kernel resident at once on the chip (mgx_step_round_blocks, at create); 0 = unknown
    int stagger;           // first-round waves sleep slot * stagger * 256 clocks before their loads (set per launch; 0 = off)
};

struct PackParams {
    // inputs (reference encoding)
    const uint8_t *grid; const uint8_t *aux; const int32_t *agent; const uint8_t *carry; const int32_t *steps;
    const uint8_t *mask;
    // internal state
    uint8_t *cells; uint8_t *cells0; uint2 *rec; uint2 *rec0;
    uint8_t *objaux, *objaux0, *objcont, *objcont0; uint16_t *objcarry; // null unless object_state
    // outputs (unpack)
    uint8_t *grid_out; uint8_t *aux_out; int32_t *agent_out; uint8_t *carry_out; int32_t *steps_out;
    MgxCounters *ctr;
    int64_t n;
    int W, H, S, has_task;
    int bcast; // grid / agent hold ONE env that every (masked) env receives
};

struct ConsumeParams {
    const uint8_t *mask; // u8[n] or null
    uint8_t *cells; const uint8_t *cells0; uint2 *agent; const uint2 *agent0; uint8_t *regen;
    uint8_t *objaux, *objaux0, *objcont, *objcont0; uint16_t *objcarry; // object_state handles (else null): aux planes to 0,
                                                                        // contains <- the generated level's (objcont0), nothing carried
    uint8_t *front;     // StepParams.front of gather-form handles (else null): cleared for every env consumed here
    uint8_t *wcache;    // StepParams.wcache (else null): the pose tag of every env consumed here is cleared
    const uint8_t *bank; // seed schedule (StepParams.bank, already advanced for the envs being reset: k_bank_advance), else null
    int64_t bank_envs;
    uint8_t *restart;   // Dynamic-Obstacles handles under a seed schedule (else null): DynObsParams.regen, raised for every env consumed here
    int64_t n;
    int S, flag_regen;
};
// Dynamic-Obstacles (MGX_TASK_DYNOBS): the obstacle walk that precedes the base step, and its reset-time setup
struct DynObsParams {
    uint8_t *cells, *cells0;
    const uint2 *agent;
    const uint8_t *actions; // caller's actions (k_dynobs) ...
    uint8_t *act_out;       // ... folded to 0..2, bit 7 = "moved forward while the front cell was not clear"
    const uint8_t *mask;    // k_dynobs_init: envs that were re-seeded and get a new snapshot (null = all)
    const uint8_t *mask_reset; // k_dynobs_init: envs being reset (null = all); those not re-seeded only raise regen
    uint8_t *regen;         // the handle's `restart` flags, set by the step kernels' in-kernel reset: restore obstacle order + RNG position first
    uint8_t *obst, *obst0;  // u8[n_pad][8] position x << 4 | y of obstacle i (placement order), and at episode start
    uint32_t *mt, *mt0;     // u32[n_pad][624] MT19937 words (lazily regenerated in place past the first block) + snapshot
    uint32_t *pos, *pos0;   // u32[n_pad] words drawn since the block in `mt0` was generated (= mt_idx right after reset)
    uint32_t *tape, *tape0; // u32[n_pad][MGX_DYN_TAPE_DW] draw tape of the env's block (k_dynobs.hip) + its episode-start copy
    uint8_t *front;         // StepParams.front of a gather-form handle (16x16; else null): k_dynobs leaves the cell in front of the agent AFTER the walk
    uint32_t *sp0;          // u32[n_pad] the stream position (mt_idx) the episode started at: what `pos0`'s rank was computed from; the plain
                            // caller-side reset() of an env that has not drawn since needs it back exactly (k_dynobs_handover)
    uint32_t *mt_idx;       // k_dynobs_handover: where the stream position goes (LevelGenParams.mt_idx; the same array as `pos`)
    const uint8_t *bank;    // seed schedule (StepParams.bank), else null: obst0 / mt0 / pos0 / tape0 / sp0 are [K][n_pad][...]
    int64_t bank_envs;
    int64_t n;
    int W, H, S, n_obst;
    int n_tiles, LS, wave_lds; // k_dynobs: one wave per 64-env tile, cells + RNG windows staged in LDS
};
#define MGX_SEED_WIN 64
#define MGX_LG_RING_MAX 16 /* most next-level buffers per env where the level generator runs beside the steps (mgx_api.cpp: lg_ring) */ /* words of the first MT19937 block a virtual RNG state keeps (tools/draw_stats.cpp: what the families' levels draw) */
#define MGX_DYN_TAPE_DW 56 /* two bit planes of 848 stream positions (624 of the block + 224 of the next), 28 dwords each */
int mgx_dynobs_wave_lds(int LS);
hipError_t mgx_launch_dynobs_init(const DynObsParams &p, hipStream_t st);
hipError_t mgx_launch_dynobs(const DynObsParams &p, hipStream_t st);
// plain caller-side reset() of Dynamic-Obstacles envs (mask null = all): the walk's position (a rank on the draw tape, or a half-regenerated
// block) back into the form k_levelgen continues from -- a complete block in `mt` and a stream position in `mt_idx`
hipError_t mgx_launch_dynobs_handover(const DynObsParams &p, hipStream_t st);
hipError_t mgx_launch_levelgen(const LevelGenParams &p, hipStream_t st);
hipError_t mgx_launch_seed(const uint64_t *seeds, const uint8_t *mask, const uint32_t *init, uint32_t *mt, uint32_t *mt2, uint32_t *mt_idx,
                           uint8_t *regen, uint64_t *seed0, uint8_t *has_seed, uint8_t *reseeded, int skip_same, int64_t n, hipStream_t st);
// env.seed() of virtual-state handles (LevelGenParams.win / virt): the first MGX_SEED_WIN words only
hipError_t mgx_launch_seed_window(const uint64_t *seeds, const uint8_t *mask, const uint32_t *init, uint32_t *win, uint8_t *virt, uint32_t *mt_idx,
                                  uint8_t *regen, uint64_t *seed0, uint8_t *has_seed, uint8_t *reseeded, int skip_same, int64_t n, hipStream_t st);
// the masked envs (null = all) that still hold a virtual state get their full first block (from seed0) into `mt`; mt_idx is kept
hipError_t mgx_launch_seed_materialize(const uint8_t *mask, const uint32_t *init, uint32_t *mt, uint8_t *virt, const uint64_t *seed0, int64_t n, hipStream_t st);
hipError_t mgx_launch_consume(const ConsumeParams &p, hipStream_t st);
// mgx_set_seed_schedule: out[i] = seeds[i][b] (the b-th seed of every env's list)
hipError_t mgx_launch_seed_column(const uint64_t *seeds, int K, int b, uint64_t *out, int64_t n, hipStream_t st);
// seed schedule, caller-side reset: bank[i] = (bank[i] + 1) % K for the masked envs (mask null = all)
hipError_t mgx_launch_bank_advance(uint8_t *bank, const uint8_t *mask, int K, int64_t n, hipStream_t st);
// plain caller-side reset(): the masked envs (null = all) are flagged for k_levelgen (regen = 1); their episode-start snapshots no longer
// belong to a seed (has_seed = 0) and they count as re-seeded for k_dynobs_init (reseeded = 1)
hipError_t mgx_launch_mark_plain_reset(const uint8_t *mask, uint8_t *regen, uint8_t *has_seed, uint8_t *reseeded, int64_t n, hipStream_t st);
// launch shaping of one handle on its own device (k_step.hip: raised-priority tail blocks, first-round stagger)
struct StepLaunchCfg { int tail_blocks, stagger_units, stagger_min; };
hipError_t mgx_step_launch_cfg(int device, StepLaunchCfg *out);
hipError_t mgx_launch_step(const StepParams &p, int mode, int waves_per_block, const StepLaunchCfg &lc, hipStream_t st);
hipError_t mgx_launch_step_dyn(const StepParams &p, const DynObsParams &d, const StepLaunchCfg &lc, hipStream_t st); // (declared below: DynObsParams)
// the step + the level generator's blocks for the buffers the LAST step consumed, in one launch (g.regen null: nothing pending, the step alone)
// ring of next-level buffers, caller-side: regen[i] = bank[i] + 1 and bank[i] ^= 1 for the masked envs (a buffer was consumed) | bank[i] = 0 and
// regen[i] = 2 (buffer 0 holds the next level, buffer 1 is to be made)
// DACWrapper's `last_obs`: the observation rows of absorbed envs (record bit MGX_REC_ABSORBED) become all ones (k_epilogue.hip)
hipError_t mgx_launch_dac_obs(const uint2 *agent, uint8_t *obs, int64_t n, int64_t row_bytes, hipStream_t st);
hipError_t mgx_launch_ring_consumed(uint8_t *bank, uint8_t *regen, const uint8_t *mask, int64_t n, int ring, hipStream_t st);
hipError_t mgx_launch_ring_init(uint8_t *bank, uint8_t *regen, const uint8_t *mask, int64_t n, int flag, hipStream_t st);
const char *mgx_step_kernel_label(const StepParams &p, int mode); // the instantiation the selector picks, e.g. "k_step<8,8,0,7>"
hipError_t mgx_launch_rollout(const StepParams &p, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done, int64_t T, int full, hipStream_t st);
hipError_t mgx_preload_step_kernels();
// (every .hip file is a code object of its own; one lookup each loads it at mgx_create instead of inside the first reset / step)
hipError_t mgx_preload_levelgen_kernels();
hipError_t mgx_preload_state_kernels();
hipError_t mgx_preload_epilogue_kernels();
hipError_t mgx_preload_dynobs_kernels();
hipError_t mgx_raise_lds_limit(const StepParams &p, int mode, int bytes);
hipError_t mgx_step_round_blocks(const StepParams &p, int mode, int waves_per_block, int *blocks);
#define MGX_FLAT_MISSION (96 * 27) /* FlatObsWrapper: maxStrLen x numCharCodes (wrappers.py:534-537) */
// How the per-env task word selects the mission (a row of k_flat's pattern table), per family.  One definition for the
// host (which fills the rows from mgx_mission) and the kernel.
enum { MGX_MF_CONST = 0, MGX_MF_FETCH, MGX_MF_GOTOOBJECT, MGX_MF_PICKUP, MGX_MF_LOCKEDROOM, MGX_MF_PUTNEAR };
__host__ __device__ inline int mgx_mission_rows(int family)
{
    return family == MGX_MF_FETCH ? 80 : family == MGX_MF_GOTOOBJECT ? 24 : family == MGX_MF_PICKUP ? 8 : family == MGX_MF_LOCKEDROOM ? 64 :
           family == MGX_MF_PUTNEAR ? 576 : 1;
}
__host__ __device__ inline int mgx_mission_row(int family, uint32_t task)
{
    switch (family) {
    case MGX_MF_FETCH: return (int)((((task >> 8) & 7u) * 2u + ((task & 15u) == 6u /* ball */ ? 1u : 0u)) * 8u + ((task >> 4) & 7u)); // template, type, colour
    case MGX_MF_GOTOOBJECT: return (int)(((task >> 8) & 3u) * 8u + ((task >> 10) & 7u));                                                // type, colour
    case MGX_MF_PICKUP: return (int)((task >> 4) & 7u);                                                                                  // colour of the target
    case MGX_MF_LOCKEDROOM: return (int)(task & 63u);                                                                                    // locked colour | key room colour << 3
    case MGX_MF_PUTNEAR: return (int)((((task & 3u) * 8u + ((task >> 2) & 7u)) * 24u) + ((task >> 11) & 3u) * 8u + ((task >> 13) & 7u));
    default: return 0;
    }
}
hipError_t mgx_launch_flat(const uint8_t *tri, const uint2 *rec, const float *pattern, float *out, int64_t n, int img, int family, hipStream_t st);
hipError_t mgx_launch_onehot(const uint8_t *tri, uint8_t *out, int64_t n_cells, int nc, int ns, hipStream_t st);
struct ObjStateParams {
    const uint8_t *contains_in, *carry_aux_in, *carry_contains_in;
    uint8_t *contains_out, *carry_aux_out, *carry_contains_out;
    uint8_t *objcont, *objcont0;
    uint16_t *objcarry;
    MgxCounters *ctr;
    int64_t n;
    int W, H, S;
};
hipError_t mgx_launch_objstate(const ObjStateParams &p, hipStream_t st);
hipError_t mgx_launch_task(uint2 *rec, uint2 *rec0, const uint32_t *set, uint32_t *get, const uint8_t *mask, int64_t n, hipStream_t st);
hipError_t mgx_launch_direction(const uint2 *rec, uint8_t *out, int64_t n, hipStream_t st);
hipError_t mgx_launch_pose(const uint2 *rec, int32_t *out, int64_t n, hipStream_t st);
hipError_t mgx_launch_pack(const PackParams &p, hipStream_t st);
hipError_t mgx_launch_unpack(const PackParams &p, hipStream_t st);
hipError_t mgx_launch_fill_actions(uint8_t *out, uint64_t seed, int64_t env0, int64_t t0, int64_t n, int64_t T, hipStream_t st);
hipError_t mgx_launch_read_stats(const MgxCounters *ctr, double *out2, hipStream_t st);
uint32_t mgx_action_of(uint64_t seed, uint64_t env, uint64_t t);

#endif
