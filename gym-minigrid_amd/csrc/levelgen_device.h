// levelgen_device.h -- level generation on the device: word sources, the lane-per-level fast path, the wave-per-level slow path and the body of a
// generator block (k_levelgen.hip holds the kernel, its launch shape and the seeding kernels).
#ifndef MGX_LEVELGEN_DEVICE_H
#define MGX_LEVELGEN_DEVICE_H

#include "levelgen_core.h"
#include "mgx_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// (device RNG + per-level generator, used by k_levelgen below)
// Word source of the lane-per-level fast path: a bounded window of the env's MT19937 block copied into LDS.  Running
// past it marks the level for the slow path (alive() == false stops the generators' rejection loops).
#define MGX_LGF_WIN 32
// SLIDE = false: the window is all there is (running past it sends the level to the slow path); the cheap form, for the
// families whose levels take a few dozen draws.  SLIDE = true: the window slides along the block (draw-heavy generators:
// RoomGrid's connect_all, MultiRoom), and only the end of the block sends the level to the slow path, which builds the
// next one.  (One form with a run-time slide cost the crossing generator 10-20 %: it sits in a dozen call sites.)
template <bool SLIDE>
struct WinRng {
    uint32_t *win;       // this lane's LDS window: MGX_LGF_WIN words starting at stream position `base`
    const uint32_t *mt;  // the env's block in HBM: stream positions 0 .. 623
    const uint32_t *mt2; // the block after it (positions 624 .. 1247) or null (LevelGenParams.mt2)
    int base, idx, limit, end; // end: 624, or 1248 with a second block
    bool overflow;

    __device__ __forceinline__ bool alive() const { return !overflow; }
    // A window never straddles the two blocks (it is cut at stream position 624), so a refill is MGX_LGF_WIN loads off ONE pointer
    // with immediate offsets -- all in flight together, one round trip -- and the words it reads past the block's end (the next
    // env's, or the allocation's slack behind the last env: MGX_LGF_WIN words) are never handed out.  History: a rolled loop of single
    // words (load, s_waitcnt vmcnt(0), ds_write, next) = 32 dependent round trips, the ISA had 173 inlined copies of it; then, with
    // the second block, a pointer selected per word: 32 at once cost the kernel its registers (256 VGPRs + scratch), so 4 x 8.
    __device__ __forceinline__ int stop_of(int from) const { return (from < 624 && end > 624) ? 624 : end; }
    __device__ __forceinline__ void fill(int from)
    {
        const uint32_t *src = (mt2 && from >= 624) ? mt2 + (from - 624) : mt + from;
        uint32_t t[MGX_LGF_WIN];
#pragma unroll
        for (int k = 0; k < MGX_LGF_WIN; k++) t[k] = src[k];
#pragma unroll
        for (int k = 0; k < MGX_LGF_WIN; k++) win[k] = t[k];
    }
    // The word of the NEXT draw is read from the window right behind each draw (`ahead`): a draw is an LDS read, a tempering and a compare
    // in a rejection loop, every one waiting for the one before; read one draw early, the ~100 cycles of the LDS round trip pass under
    // the caller's work instead of in front of it.  (One word past the window's valid part is read and never used.)
    uint32_t ahead;
    __device__ __forceinline__ uint32_t next32()
    {
        if (idx >= limit) {
            if (!SLIDE || idx >= end) { overflow = true; return 0u; } // masked draws end on 0; place_obj-style loops test alive()
            base = idx;
            limit = base + MGX_LGF_WIN < stop_of(base) ? base + MGX_LGF_WIN : stop_of(base);
            fill(base);
            ahead = win[0];
        }
        const uint32_t w = ahead;
        idx++;
        ahead = win[idx - base]; // (index <= MGX_LGF_WIN: the word behind the window is this lane's own slice)
        return lg_temper(w);
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

// nxt = the MT19937 block after cur (both 624 words in LDS), by the whole wave: the recurrence is 3 data-parallel phases + 1 word
__device__ __forceinline__ void twist_to(const uint32_t *cur, uint32_t *nxt, int lane)
{
    for (int k = lane; k < 227; k += 64) nxt[k] = lg_twist_word(cur[k], cur[k + 1], cur[k + 397]);
    wave_sync();
    for (int k = 227 + lane; k < 454; k += 64) nxt[k] = lg_twist_word(cur[k], cur[k + 1], nxt[k - 227]);
    wave_sync();
    for (int k = 454 + lane; k < 623; k += 64) nxt[k] = lg_twist_word(cur[k], cur[k + 1], nxt[k - 227]);
    wave_sync();
    if (lane == 0) nxt[623] = lg_twist_word(cur[623], nxt[0], nxt[396]);
    wave_sync();
}
// one env's 624-word row: HBM -> LDS / LDS -> HBM by the whole wave (all ten loads in flight before the first LDS write)
__device__ __forceinline__ void block_load(const uint32_t *g, uint32_t *l, int lane)
{
    uint32_t v[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; v[i] = g[k < 624 ? k : 623]; }
#pragma unroll
    for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; if (k < 624) l[k] = v[i]; }
}
__device__ __forceinline__ void block_store(uint32_t *g, const uint32_t *l, int lane)
{
    for (int k = lane; k < 624; k += 64) g[k] = l[k];
}

// An env whose level (made on the lane path) ended inside its second block moves one block on: mt <- mt2, mt2 <- twist(mt2).
__device__ __forceinline__ void advance_env(const LevelGenParams &p, int64_t env, uint8_t *base, int lane)
{
    uint32_t *cur = reinterpret_cast<uint32_t *>(base), *nxt = cur + 624;
    block_load(p.mt2 + env * 624, cur, lane);
    wave_sync();
    block_store(p.mt + env * 624, cur, lane);
    twist_to(cur, nxt, lane);
    block_store(p.mt2 + env * 624, nxt, lane);
    wave_sync();
}

// one step of init_by_array's first / second pass
__device__ __forceinline__ uint32_t ib_step2(uint32_t prev, uint32_t tab, uint32_t kj) { return (tab ^ ((prev ^ (prev >> 30)) * 1664525u)) + kj; }
__device__ __forceinline__ uint32_t ib_step3(uint32_t prev, uint32_t old, uint32_t i) { return (old ^ ((prev ^ (prev >> 30)) * 1566083941u)) - i; }
// init_by_array(key of `seed`) by ONE lane into 624 words of LDS (untwisted state, as numpy's RandomState.seed leaves it)
__device__ __noinline__ void seed_block_one_lane(uint64_t seed, const uint32_t *__restrict__ init, uint32_t *m)
{
    uint32_t key[2];
    const int klen = lg_seed_key(seed, key);
    const uint32_t kj0 = key[0], kj1 = klen == 2 ? key[1] + 1u : key[0]; // key[j] + j for j = (i - 1) % klen
    uint32_t prev = init[0];
#pragma nounroll
    for (int i = 1; i < 624; i++) { prev = ib_step2(prev, init[i], (i & 1) ? kj0 : kj1); m[i] = prev; }
    prev = ib_step2(prev, m[1], kj1); // the 624th step of the first loop wraps to i = 1 (623 % klen picks key[1] + 1, or key[0] again for a one-word key)
    m[1] = prev;
#pragma nounroll
    for (int i = 2; i < 624; i++) { prev = ib_step3(prev, m[i], (uint32_t)i); m[i] = prev; }
    m[1] = ib_step3(prev, m[1], 1u);
    m[0] = 0x80000000u;
}

// One level, generated by one wave into its LDS workspace and written back coalesced.
// (senv: the env's slot in the next-level buffers -- env itself, or env + bank_envs for the second buffer of a ring, LevelGenParams.bank_envs)
__device__ __forceinline__ void levelgen_one(const LevelGenParams &p, int64_t env, uint8_t *base, int lane, int64_t senv)
{
    uint32_t *cur = reinterpret_cast<uint32_t *>(base), *nxt = cur + 624;
    int *res = reinterpret_cast<int *>(base + 2 * 624 * 4); // [0]=idx after, [1]=packed agent, [2]=overflow, [3]=#cmds
    int16_t *ws = reinterpret_cast<int16_t *>(res + 4);
    LgCmd *cmds = reinterpret_cast<LgCmd *>(ws + MGX_LG_WS_WORDS);
    uint32_t *mt = p.mt + env * 624;
    // A virtual state (seed + first words only) that a level outgrew: the env's full first block is re-derived from its seed -- ONE lane
    // runs init_by_array into LDS (the recurrences are sequential per env; rare: tools/draw_stats.cpp), the wave twists it -- and written
    // to `mt` below; from then on the env's state is materialized.
    const bool was_virtual = p.virt && p.virt[env];
    if (was_virtual) {
        if (lane == 0) seed_block_one_lane(p.seed0[env], p.mt_init, nxt);
        wave_sync();
        twist_to(nxt, cur, lane);
    } else block_load(mt, cur, lane);
    const int idx0 = (int)p.mt_idx[env];
    wave_sync();
    // The block after this one: kept in HBM by new_level_each_episode handles (p.mt2); otherwise, if the read index is within 64 words
    // of the end of the block the level will probably run into the next one and the whole wave builds it first.
    const bool pre = p.mt2 != nullptr || idx0 + 64 > 624;
    if (p.mt2) { block_load(p.mt2 + env * 624, nxt, lane); wave_sync(); }
    else if (pre) twist_to(cur, nxt, lane);
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
        uint32_t *dst = reinterpret_cast<uint32_t *>(p.cells0 + senv * p.S);
        for (int k = lane; k < (p.S >> 2); k += 64) {
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = 4 * k + b;
                if (c < cells) { const int x = c / H; w |= lg_cell_code(cmds, ncmd, x, c - x * H) << (8 * b); }
            }
            dst[k] = w;
        }
        if (p.objcont0) { // hidden planes of the next level: no aux state, boxes hold what the generator put into them
            uint32_t *da = reinterpret_cast<uint32_t *>(p.objaux0 + senv * p.S), *dc = reinterpret_cast<uint32_t *>(p.objcont0 + senv * p.S);
            for (int k = lane; k < (p.S >> 2); k += 64) {
                uint32_t w = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int c = 4 * k + b;
                    const int x = c < cells ? c / H : 0;
                    w |= (c < cells ? lg_cell_cont(cmds, ncmd, x, c - x * H) : (uint32_t)MGX_CODE_EMPTY) << (8 * b);
                }
                da[k] = 0u;
                dc[k] = w;
            }
        }
    }
    if (was_virtual && !res[2]) block_store(mt, cur, lane);
    if (was_virtual && lane == 0) p.virt[env] = 0;
    if (res[2]) { // moved into a later block: it becomes the env's state (and the one after it is made ready, where the handle keeps one)
        uint32_t *blk = res[2] == 1 ? cur : nxt, *other = res[2] == 1 ? nxt : cur;
        block_store(mt, blk, lane);
        if (p.mt2) {
            twist_to(blk, other, lane);
            block_store(p.mt2 + env * 624, other, lane);
        }
    }
    if (lane == 0) {
        p.mt_idx[env] = (uint32_t)idx1;
        p.agent0[senv] = make_uint2((uint32_t)res[1] | ((uint32_t)MGX_CODE_EMPTY << 24), (uint32_t)(uint16_t)ws[MGX_LG_WS_WORDS - 1] << 16);
    }
    wave_sync();
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
// per level made it cost as much as k_step itself (50 us at 8,400 levels per step).  Two forms (WinRng<SLIDE>): the cheap
// one (Empty / DoorKey / Crossing / LavaGap on rows <= 128 bytes) gives up when the level needs more than the 32-word
// window; the sliding one (every other family, every larger grid) refills the window from the block as it goes and
// paints rows longer than the slice straight into HBM.  What still does not fit -- the end of the MT19937 block inside
// the level, more than 40 commands / 8 rivers per axis -- goes to a second LDS queue and is generated afterwards by all
// 4 waves, one level per wave at a time (levelgen_one: cooperative next-block build, full-size buffers).
#define MGX_LG_LDS_PER_WAVE (2 * 624 * 4 + 16 + 2 * MGX_LG_WS_WORDS + 8 * MGX_LG_MAX_CMDS)
static_assert(MGX_LG_LDS_PER_WAVE == MGX_LG_LDS_PER_WAVE_BYTES, "keep mgx_kernels.h in sync");
#define MGX_LGF_ENVS 2048 /* most envs per block; FastLayout.span picks 512..2048 so that the grid is ONE round of resident blocks
   (2 per CU by LDS): 2048 for 1 Mi envs -- 512 there made the steady state of LavaCrossing 64 -> 82 us per step, four rounds of
   blocks -- and 512 for 262,144, where 2048 left a burst (every env timing out on the same step) with 512 waves on 1,024 SIMDs:
   MultiRoom-N6 654 -> 199 us per step, KeyCorridorS3R3 338 -> 213 */
#define MGX_LGF_CMDS 40
#define MGX_LGF_RIVERS 8
#ifndef MGX_LGF_MAXS
#define MGX_LGF_MAXS 384 /* largest grid row (bytes) painted in a lane slice: up to 19x19 (128 until the placement loops probed the image; us per step with a
                            new level per episode at 262,144 envs, 128 -> 384: ObstructedMaze-2Dlhb 55.8 -> 37.7, MemoryS13Random 43.9 -> 33.4, Playground 45.7 -> 40.7) */
#endif
#ifndef MGX_LGF_MAXS_CHEAP
#define MGX_LGF_MAXS_CHEAP 128 /* ... for the cheap (non-sliding) form */
#endif
// FAST PATH, one lane per level.  Returns false when the level has to go to the slow path.
// Layout of a lane's LDS slice, sized per family by the launcher (fewer dwords per lane = more lanes generating per CU):
// [0, 32) RNG window | 2*cmd_cap paint commands | 3*river_cap crossing lists | img_dw level image (rows <= 128 B) | 1 pad.
// (struct FastLayout: mgx_kernels.h -- the launcher of the fused step kernel fills one as well)

// the lane's word source and level buffers set up in its slice; false: this env's block is used up (slow path)
template <bool SLIDE>
__device__ __forceinline__ bool fast_setup(const LevelGenParams &p, const FastLayout &fl, int64_t env, uint32_t *slice, int W, int H, int cells, WinRng<SLIDE> &r, LgLevel &L)
{
    const int idx0 = (int)p.mt_idx[env];
    const bool v = p.virt && p.virt[env]; // virtual state: only the first MGX_SEED_WIN words of the first block exist (LevelGenParams.win)
    const int blk = v ? MGX_SEED_WIN : 624;
    const int end = v ? MGX_SEED_WIN : (p.mt2 ? 1248 : 624); // stream positions this env has ready: its block, and the next one where the handle keeps it
    if (idx0 >= blk || (!SLIDE && (idx0 + MGX_LGF_WIN > end || fl.img_dw == 0))) return false;
    r.win = slice; r.mt = v ? p.win + env * MGX_SEED_WIN : p.mt + env * 624; r.mt2 = (!v && p.mt2) ? p.mt2 + env * 624 : nullptr; r.end = end;
    r.base = idx0; r.idx = idx0; r.overflow = false;
    r.limit = idx0 + MGX_LGF_WIN < r.stop_of(idx0) ? idx0 + MGX_LGF_WIN : r.stop_of(idx0);
    r.fill(idx0);
    r.ahead = slice[0];
    L.cmds = reinterpret_cast<LgCmd *>(slice + MGX_LGF_WIN); L.ncmd = 0; L.max_cmds = fl.cmd_cap;
    L.ws = reinterpret_cast<int16_t *>(slice + MGX_LGF_WIN + 2 * fl.cmd_cap); L.max_rivers = fl.river_cap;
    L.W = W; L.H = H; L.ax = L.ay = -1; L.adir = 0;
    uint32_t *img32 = slice + MGX_LGF_WIN + 2 * fl.cmd_cap + 3 * fl.river_cap;
    if (fl.img_dw) { // the level image in the slice, painted by lg_rect as the level grows (LgLevel.occ: the placement loops probe it)
        uint8_t *img = reinterpret_cast<uint8_t *>(img32);
        for (int k = 0; k < (p.S >> 2); k++) img32[k] = 4 * k + 3 < cells ? 0x01010101u * MGX_CODE_EMPTY : 0u;
        for (int c = cells & ~3; c < cells; c++) img[c] = MGX_CODE_EMPTY;
        L.occ = img;
    }
    return true;
}

// the generated level into the env's next-level buffer; false: it has to be made again by the slow path
template <bool SLIDE>
__device__ __forceinline__ bool fast_store(const LevelGenParams &p, const FastLayout &fl, int64_t env, int64_t senv, uint32_t *slice, int H, int cells, const WinRng<SLIDE> &r, const LgLevel &L, bool &crossed)
{
    if (r.overflow || L.too_big) return false;
    uint32_t *img32 = slice + MGX_LGF_WIN + 2 * fl.cmd_cap + 3 * fl.river_cap;
    uint32_t *dst = reinterpret_cast<uint32_t *>(p.cells0 + senv * p.S);
    if (fl.img_dw) {
        for (int k = 0; k < (p.S >> 2); k++) dst[k] = img32[k]; // one pass of dword stores
    } else {
        // rows too long for the slice (25x25 and up): paint straight into the env's row in HBM (this lane's own stores,
        // in order)
        uint8_t *dstb = reinterpret_cast<uint8_t *>(dst);
        for (int k = 0; k < (p.S >> 2); k++) dst[k] = 4 * k + 3 < cells ? 0x01010101u * MGX_CODE_EMPTY : 0u;
        for (int c = cells & ~3; c < cells; c++) dstb[c] = MGX_CODE_EMPTY;
        for (int q = 0; q < L.ncmd; q++) {
            const LgCmd c = L.cmds[q];
            for (int x = c.x0; x <= c.x1; x++)
                for (int y = c.y0; y <= c.y1; y++) dstb[x * H + y] = c.code;
        }
    }
    if (p.objcont0) { // hidden planes of the next level (this lane's own stores, in order): defaults, then the boxes' contents
        uint32_t *da = reinterpret_cast<uint32_t *>(p.objaux0 + senv * p.S), *dc = reinterpret_cast<uint32_t *>(p.objcont0 + senv * p.S);
        for (int k = 0; k < (p.S >> 2); k++) { da[k] = 0u; dc[k] = 0x01010101u * MGX_CODE_EMPTY; }
        uint8_t *dcb = reinterpret_cast<uint8_t *>(dc);
        for (int q = 0; q < L.ncmd; q++) {
            const LgCmd c = L.cmds[q];
            if (c.x0 == c.x1 && c.y0 == c.y1) dcb[c.x0 * H + c.y0] = c.cont ? c.cont : (uint8_t)MGX_CODE_EMPTY; // (a later command over a box empties the cell's entry)
        }
    }
    // (a level that ended in the second block: the position is stored relative to it; the caller moves the env one block on)
    // (without a second block a level that used exactly the last word of its block leaves idx == 624 = "exhausted", as ever: the fuzz
    // found the first version of this line treating that as a crossing and dereferencing the null mt2)
    crossed = p.mt2 != nullptr && r.idx >= 624;
    p.mt_idx[env] = (uint32_t)(crossed ? r.idx - 624 : r.idx);
    p.agent0[senv] = make_uint2((uint32_t)((L.ax & 255) | ((L.ay & 255) << 8) | ((L.adir & 3) << 16)) | ((uint32_t)MGX_CODE_EMPTY << 24), L.task << 16);
    return true;
}

template <bool SLIDE>
__device__ __forceinline__ bool fast_level(const LevelGenParams &p, const FastLayout &fl, int64_t env, int64_t senv, uint32_t *slice, int W, int H, int cells, bool &crossed)
{
    crossed = false;
    WinRng<SLIDE> r;
    LgLevel L;
    if (!fast_setup<SLIDE>(p, fl, env, slice, W, H, cells, r, L)) return false;
    lg_generate(p.cfg, r, L);
    return fast_store<SLIDE>(p, fl, env, senv, slice, H, cells, r, L, crossed);
}

// (4 waves per SIMD = 128 VGPRs instead of 151, no spills: MultiRoom's four blocks per CU need them)
#ifndef MGX_LG_WAVES
#define MGX_LG_WAVES 4
#endif
// One block of the level generator: 256 threads, span `fl.span` of envs starting at block_id * fl.span (the body of k_levelgen).
template <bool MULTI>
__device__ __forceinline__ void levelgen_block(const LevelGenParams &p, const FastLayout &fl, int block_id)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_slices[]; // n_fast_waves x `lanes` slices; reused by the slow path (>= 4 workspaces)
    // the two queues behind them, `span` entries each (as static arrays of the largest span they cost every block 8 KB: with
    // MultiRoom's 37.6 KB of slices that was the difference between three and four blocks per CU)
    uint16_t *s_queue = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(s_slices) + fl.queue_off), *s_slow = s_queue + fl.span;
    __shared__ int s_count, s_nslow, s_head;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t env_base = (int64_t)block_id * fl.span;
    const int W = p.cfg.width, H = p.cfg.height, cells = W * H;
    // the cheap form for the families whose levels take a bounded, small number of draws on small grids (measured: the
    // sliding form costs the crossing generator 10-20 %); the sliding window and the direct paint for everything else
    const int kind = p.cfg.level_kind;
    const bool cheap = p.S <= MGX_LGF_MAXS_CHEAP && (kind == MGX_LEVEL_EMPTY || kind == MGX_LEVEL_DOORKEY || kind == MGX_LEVEL_CROSSING ||
                                               kind == MGX_LEVEL_LAVAGAP || kind == MGX_LEVEL_DISTSHIFT);
    const bool slide = !cheap || p.virt != nullptr; // (a virtual state is 64 words long: the cheap form's all-or-nothing window of 32 would send 1 % of DoorKey-8x8's levels to the slow path)
    // Several flag arrays (LevelGenParams.n_regen: the steps of one run of a ring, oldest first) in ONE launch: a pass takes, for every env, the
    // flag of the OLDEST array that has one -- its levels are independent of each other and fill the waves together -- and leaves the env's later
    // flags for the next pass: an env that finished twice within the run draws its two levels in step order, the order of its RNG stream.  One
    // array: one pass, as ever.
    // (MULTI is a template parameter: with the pass loop around it the one-array kernel spilled 83 registers instead of 46 and every launch of it
    // took 10 us longer)
    const int n_regen = MULTI ? (p.n_regen > 1 ? p.n_regen : 1) : 1;
    for (int pass = 0; pass < n_regen; pass++) {
        if (tid == 0) { s_count = 0; s_nslow = 0; s_head = 0; }
        __syncthreads();
        if (tid < fl.span / 8) { // scan 8 flags per thread and array (the regen arrays are padded to whole tiles and the span is a multiple of 64)
            const int64_t e0 = env_base + (int64_t)tid * 8;
            if (e0 < p.n) {
                uint32_t taken = 0u; // envs of this thread that have their level of this pass
                for (int a = 0; a < n_regen; a++) {
                    uint2 *f2 = reinterpret_cast<uint2 *>((a == 0 ? p.regen : p.regen_more[a - 1]) + e0);
                    uint2 f = *f2;
                    if (!(f.x | f.y)) continue;
                    bool wrote = false;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        // (a ring of next-level buffers, LevelGenParams.bank_envs != 0: the flag is the buffer's index + 1, bits 14:11 of the queue entry; a span has at most 2048 envs)
                        const uint32_t v = ((i < 4 ? f.x : f.y) >> (8 * (i & 3))) & 255u;
                        if (v && !((taken >> i) & 1u) && e0 + i < p.n) {
                            s_queue[atomicAdd(&s_count, 1)] = (uint16_t)((tid * 8 + i) | (p.bank_envs ? ((v - 1u) & 15u) << 11 : 0u));
                            taken |= 1u << i;
                            if (i < 4) f.x &= ~(255u << (8 * i)); else f.y &= ~(255u << (8 * (i - 4)));
                            wrote = true;
                        } else if (v && e0 + i >= p.n) { // (padding never carries flags; cleared all the same)
                            if (i < 4) f.x &= ~(255u << (8 * i)); else f.y &= ~(255u << (8 * (i - 4)));
                            wrote = true;
                        }
                    }
                    if (wrote) *f2 = f;
                }
            }
        }
        __syncthreads();
        const int count = s_count;
        if (count == 0) break; // (block-uniform)
        if (fl.n_fast_waves == 0) { // every level by a whole wave (levelgen_one)
            for (int i = tid; i < count; i += 256) s_slow[i] = s_queue[i];
            if (tid == 0) s_nslow = count;
        }
        if (wv < fl.n_fast_waves && lane < fl.lanes) {
            uint32_t *slice = s_slices + ((size_t)wv * fl.lanes + lane) * fl.slice_dw;
            const int stride = fl.lanes * fl.n_fast_waves;
            // The queue goes round the waves level by level (entry lane * waves + wave of each chunk), not wave by wave: a steady flow of resets
            // leaves a block a few dozen levels, and filled wave by wave they all sat in wave 0 -- one wave per block alone on its SIMD,
            // running the union of 40 lanes' control flow with nothing to hide its latencies, three waves waiting at the barrier (PutNear,
            // 262,144 envs: every launch 60-70 us for ~10 k cheap levels).
            for (int i = lane * fl.n_fast_waves + wv; i < ((count + stride - 1) / stride) * stride; i += stride) {
                if (i >= count) continue;
                const int64_t env = env_base + (s_queue[i] & 0x7FF);
                const int64_t senv = env + (int64_t)((s_queue[i] >> 11) & 15) * p.bank_envs;
                bool crossed;
                const bool ok = slide ? fast_level<true>(p, fl, env, senv, slice, W, H, cells, crossed) : fast_level<false>(p, fl, env, senv, slice, W, H, cells, crossed);
                // (the second queue holds both kinds of whole-wave work: levels the lane path gave up on, and -- bit 15 -- envs whose level is
                // done but ended in their second MT19937 block)
                if (!ok) s_slow[atomicAdd(&s_nslow, 1)] = s_queue[i];
                else if (crossed) s_slow[atomicAdd(&s_nslow, 1)] = (uint16_t)(s_queue[i] | 0x8000u);
            }
        }
        __syncthreads();
        const int nslow = s_nslow;
        if (nslow != 0) {
            uint8_t *base = reinterpret_cast<uint8_t *>(s_slices) + (size_t)wv * MGX_LG_LDS_PER_WAVE;
            for (;;) { // wave-uniform
                int i = 0;
                if (lane == 0) i = atomicAdd(&s_head, 1);
                i = __builtin_amdgcn_readfirstlane(i);
                if (i >= nslow) break;
                const uint32_t job = s_slow[i];
                if (job & 0x8000u) advance_env(p, env_base + (job & 0x7FFu), base, lane);
                else levelgen_one(p, env_base + (job & 0x7FFu), base, lane, env_base + (job & 0x7FFu) + (int64_t)((job >> 11) & 15u) * p.bank_envs);
            }
        }
        if (pass + 1 < n_regen) __syncthreads(); // (the next pass reuses the queues, the counters and the slices)
    }
}

} // namespace

#endif
