// levelgen.cpp -- host-side seeded level generation for the built-in env families.
//
// Reproduces `env.seed(s); env.reset()` of the reference for the families named in
// BASELINE.json's configs (plus their siblings), in the reference's own encoding:
//   EmptyEnv._gen_grid     /root/reference/gym_minigrid/envs/empty.py:30-57
//   DoorKeyEnv._gen_grid   /root/reference/gym_minigrid/envs/doorkey.py:15-44
//   CrossingEnv._gen_grid  /root/reference/gym_minigrid/envs/crossing.py:24-92
//   LavaGapEnv._gen_grid   /root/reference/gym_minigrid/envs/lavagap.py:21-59
//   place_obj/place_agent  /root/reference/gym_minigrid/minigrid.py:1003-1090
//   seed()                 /root/reference/gym_minigrid/minigrid.py:860-863 -> gym.utils.seeding.np_random
//
// The random stream is third-party (gym < 0.22 + numpy RandomState, neither vendored by the
// reference; setup.py:9-12 pins only lower bounds).  Restated here from their published
// algorithms:
//   gym legacy seeding : seed mod 2^64 -> SHA-512(str(seed)) -> first 8 bytes as little-endian
//                        uint32 words (trailing zero words dropped, 0 -> [0]) ->
//   numpy RandomState  : MT19937 init_by_array(words); randint/choice/shuffle all reduce to
//                        "raw 32-bit output & mask, redraw while > max" (mask = next 2^k-1 >= max;
//                        max == 0 consumes nothing); randint(lo,hi) = lo + bounded(hi-lo-1);
//                        shuffle(list) draws bounded(i) for i = n-1 .. 1; choice(seq) = randint(0,len).
// Pinned by tests/test_levelgen.py against tests/golden/levels.npz (recorded from the reference
// running on this image's NumPy under the oracle's gym stand-in).

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <thread>

#include "mgx.h"
#include "mgx_internal.h"
#include "levelgen_core.h"

namespace {

// ----------------------------------------------------------------------------- MT19937 + numpy legacy draws
struct Rng {
    uint32_t mt[624];
    int idx;

    void init_genrand(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t *key, int klen)
    {
        init_genrand(19650218U);
        int i = 1, j = 0;
        int k = 624 > klen ? 624 : klen;
        for (; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= klen) j = 0;
        }
        for (k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000U;
        idx = 624;
    }
    bool alive() const { return true; }
    uint32_t next32()
    {
        if (idx >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) {
                uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
            }
            for (; kk < 623; kk++) {
                uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
            }
            uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680U;
        y ^= (y << 15) & 0xefc60000U;
        y ^= (y >> 18);
        return y;
    }
    // uniform integer in [0, max] by masked rejection
    uint32_t bounded(uint32_t max)
    {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = next32() & mask; } while (v > max);
        return v;
    }
    int randint(int lo, int hi) { return lo + (int)bounded((uint32_t)(hi - lo - 1)); }

    // gym.utils.seeding.np_random(seed)
    void seed_gym(uint64_t seed)
    {
        uint32_t key[2];
        const int klen = lg_seed_key(seed, key);
        init_by_array(key, klen);
    }
};

// ----------------------------------------------------------------------------- glue to levelgen_core.h
// codes -> the reference's (type, color, state) triples
void codes_to_triples(const uint8_t *codes, int cells, uint8_t *out)
{
    for (int i = 0; i < cells; i++) {
        const uint32_t c = codes[i], k = c & 15u, col = (c >> 4) & 7u;
        const bool shut = k > MGX_K_AGENT;
        out[3 * i] = (uint8_t)(shut ? 4u : k);
        out[3 * i + 1] = (uint8_t)col;
        out[3 * i + 2] = (uint8_t)(shut ? k - 10u : 0u);
    }
}

int check_levelgen_cfg(const mgx_config *cfg, const char *fn)
{
    const int W = cfg->width, H = cfg->height;
    if (W < 3 || H < 3 || W > 255 || H > 255) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: grid %dx%d outside 3..255", fn, W, H);
    switch (cfg->level_kind) {
    case MGX_LEVEL_EMPTY: break;
    case MGX_LEVEL_FOURROOMS:
        if (W < 7 || H < 7) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: FourRooms needs at least 7x7", fn);
        break;
    case MGX_LEVEL_MULTIROOM: {
        const int mn = cfg->level_arg0 & 255, mx = (cfg->level_arg0 >> 8) & 255;
        if (mn < 1 || mx < mn || mx > 8 || cfg->level_arg1 < 4 || W != H)
            return mgx_fail(MGX_ERR_INVALID_ARG, "%s: MultiRoom needs 1 <= minNumRooms <= maxNumRooms <= 8, maxRoomSize >= 4, square grid", fn);
        break;
    }
    case MGX_LEVEL_DISTSHIFT:
        if (cfg->level_arg0 < 1 || cfg->level_arg0 > H - 2) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: DistShift strip2_row %d outside the room", fn, cfg->level_arg0);
        break;
    case MGX_LEVEL_FETCH:
        if (cfg->level_arg0 < 1 || cfg->level_arg0 > (W - 2) * (H - 2) - 1) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: Fetch numObjs %d does not fit", fn, cfg->level_arg0);
        break;
    case MGX_LEVEL_GOTODOOR:
        if (W < 5 || H < 5) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: GoToDoor needs at least 5x5 (envs/gotodoor.py:14)", fn);
        break;
    case MGX_LEVEL_DOORKEY:
        if (W < 5 || H < 5) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: DoorKey needs at least 5x5", fn);
        break;
    case MGX_LEVEL_LAVAGAP:
        if (W < 5 || H < 5) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: LavaGap needs at least 5x5 (envs/lavagap.py:22)", fn);
        break;
    case MGX_LEVEL_CROSSING:
        if (W % 2 == 0 || H % 2 == 0) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: crossing levels need odd sizes (envs/crossing.py:25)", fn);
        if ((W - 3) / 2 > MGX_LG_MAX_RIVERS || (H - 3) / 2 > MGX_LG_MAX_RIVERS)
            return mgx_fail(MGX_ERR_UNSUPPORTED, "%s: crossing grid %dx%d has more than %d candidate rivers per axis", fn, W, H, MGX_LG_MAX_RIVERS);
        break;
    case MGX_LEVEL_LOCKEDROOM:
        if (W != 19 || H != 19) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: LockedRoom is 19x19", fn);
        break;
    case MGX_LEVEL_TWOGOALS:
        if (W < 4 || H < 4) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: TwoGoals needs at least 4x4", fn);
        break;
    case MGX_LEVEL_PUTNEAR:
        if (W > 8 || H > 8 || cfg->level_arg0 < 2 || cfg->level_arg0 > 4) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: PutNear needs W, H <= 8 and 2..4 objects", fn);
        break;
    case MGX_LEVEL_PLAYGROUND:
        if (W < 10 || H < 10 || W % 3 != 1 || H % 3 != 1) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: Playground grids are 3k+1 wide and high (19x19)", fn);
        break;
    case MGX_LEVEL_KEYCORRIDOR: {
        const int S = cfg->level_arg0;
        if (S < 3 || S > 6 || W != 3 * (S - 1) + 1 || (H - 1) % (S - 1) != 0 || (H - 1) / (S - 1) < 1 || (H - 1) / (S - 1) > 3)
            return mgx_fail(MGX_ERR_INVALID_ARG, "%s: KeyCorridor is a RoomGrid of 3 x (1..3) rooms of size 3..6", fn);
        break;
    }
    case MGX_LEVEL_UNLOCK:
        if (W != 11 || H != 6 || cfg->level_arg0 < 0 || cfg->level_arg0 > 2) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: the Unlock family is 11x6 with variant 0..2", fn);
        break;
    case MGX_LEVEL_OBSTRUCTEDMAZE: {
        const int nq = cfg->level_arg1 & 7;
        const bool ok1d = cfg->level_arg1 == 0 && W == 11 && H == 6, ok2d = (nq == 1 || nq == 2 || nq == 4) && (cfg->level_arg1 & ~15) == 0 && W == 16 && H == 16;
        if ((!ok1d && !ok2d) || cfg->level_arg0 < 0 || cfg->level_arg0 > 3)
            return mgx_fail(MGX_ERR_INVALID_ARG, "%s: ObstructedMaze is 11x6 (level_arg1 = 0) or 16x16 (level_arg1 = num_quarters 1/2/4 [| 8]), level_arg0 = 0..3", fn);
        break;
    }
    case MGX_LEVEL_MEMORY:
        if (W != H || !(H & 1) || H < 7 || H > 17) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: Memory grids are odd squares of 7..17", fn);
        break;
    case MGX_LEVEL_REDBLUEDOORS:
        if (W != 2 * H || H < 4 || H > 16 || (H & 1))
            return mgx_fail(MGX_ERR_INVALID_ARG, "%s: RedBlueDoors grids are 2*size x size with an even size of 4..16", fn);
        break;
    case MGX_LEVEL_GOTOOBJECT:
        if (W > 16 || H > 16 || cfg->level_arg0 < 1 || cfg->level_arg0 > 8)
            return mgx_fail(MGX_ERR_INVALID_ARG, "%s: GoToObject needs W, H <= 16 and 1..8 objects", fn);
        break;
    case MGX_LEVEL_DYNOBS:
        if (W < 5 || H < 5 || W > 16 || H > 16) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: Dynamic-Obstacles grids are 5x5 .. 16x16", fn);
        if (cfg->level_arg0 < 0 || cfg->level_arg0 > 8) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: n_obstacles %d (0..8)", fn, cfg->level_arg0);
        break;
    default: return mgx_fail(MGX_ERR_NO_LEVELGEN, "%s: level_kind %d has no built-in generator", fn, cfg->level_kind);
    }
    return MGX_OK;
}

struct EnvId { const char *id; mgx_config cfg; };

mgx_config mk(int w, int h, int max_steps, int see, int v1, int kind, int a0, int a1)
{
    mgx_config c;
    memset(&c, 0, sizeof c);
    c.width = w; c.height = h; c.max_steps = max_steps; c.see_through_walls = see; c.lava_v1 = v1;
    c.level_kind = kind; c.level_arg0 = a0; c.level_arg1 = a1;
    return c;
}

mgx_config mkt(int w, int h, int max_steps, int see, int kind, int a0, int task)
{
    mgx_config c = mk(w, h, max_steps, see, 0, kind, a0, 0);
    c.task_kind = task;
    return c;
}

// ObstructedMaze: max_steps = 4 * num_rooms_visited * room_size^2 (envs/obstructedmaze.py:17-18); boxes need the contains plane
mgx_config mko(int w, int h, int rooms_visited, int flags, int layout)
{
    mgx_config c = mk(w, h, 4 * rooms_visited * 36, 0, 0, MGX_LEVEL_OBSTRUCTEDMAZE, flags, layout);
    c.task_kind = MGX_TASK_PICKUPBOX;
    c.object_state = flags & 1;
    return c;
}

mgx_config mkd(int size, int n_obst, int random_start)
{
    mgx_config c = mk(size, size, 4 * size * size, 1, 0, MGX_LEVEL_DYNOBS, n_obst, random_start);
    c.task_kind = MGX_TASK_DYNOBS;
    return c;
}

const std::vector<EnvId> &registry()
{
    static const std::vector<EnvId> R = {
        // EmptyEnv: max_steps = 4*size^2, see_through_walls=True (envs/empty.py:23-28)
        {"MiniGrid-Empty-5x5-v0", mk(5, 5, 100, 1, 0, MGX_LEVEL_EMPTY, 0, 0)},
        {"MiniGrid-Empty-6x6-v0", mk(6, 6, 144, 1, 0, MGX_LEVEL_EMPTY, 0, 0)},
        {"MiniGrid-Empty-8x8-v0", mk(8, 8, 256, 1, 0, MGX_LEVEL_EMPTY, 0, 0)},
        {"MiniGrid-Empty-16x16-v0", mk(16, 16, 1024, 1, 0, MGX_LEVEL_EMPTY, 0, 0)},
        {"MiniGrid-Empty-Random-5x5-v0", mk(5, 5, 100, 1, 0, MGX_LEVEL_EMPTY, 1, 0)},
        {"MiniGrid-Empty-Random-6x6-v0", mk(6, 6, 144, 1, 0, MGX_LEVEL_EMPTY, 1, 0)},
        {"MiniGrid-Empty-Random-8x8-v0", mk(8, 8, 256, 1, 0, MGX_LEVEL_EMPTY, 1, 0)},
        {"MiniGrid-Empty-Random-10x10-v0", mk(10, 10, 400, 1, 0, MGX_LEVEL_EMPTY, 1, 4)},
        // DoorKeyEnv: max_steps = 10*size^2 (envs/doorkey.py:9-13)
        {"MiniGrid-DoorKey-5x5-v0", mk(5, 5, 250, 0, 0, MGX_LEVEL_DOORKEY, 0, 0)},
        {"MiniGrid-DoorKey-6x6-v0", mk(6, 6, 360, 0, 0, MGX_LEVEL_DOORKEY, 0, 0)},
        {"MiniGrid-DoorKey-8x8-v0", mk(8, 8, 640, 0, 0, MGX_LEVEL_DOORKEY, 0, 0)},
        {"MiniGrid-DoorKey-16x16-v0", mk(16, 16, 2560, 0, 0, MGX_LEVEL_DOORKEY, 0, 0)},
        // CrossingEnv: max_steps = 4*size^2 (envs/crossing.py:12-22,94-172)
        {"MiniGrid-LavaCrossingS9N1-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 1, 9)},
        {"MiniGrid-LavaCrossingS9N0-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 1, 9 + 16 * 2)}, // LavaCrossingEnvVert: ori=1
        {"MiniGrid-LavaCrossingS9N2-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 2, 9)},
        {"MiniGrid-LavaCrossingS9N3-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 3, 9)},
        {"MiniGrid-LavaCrossingS11N5-v0", mk(11, 11, 484, 0, 0, MGX_LEVEL_CROSSING, 5, 9)},
        {"MiniGrid-SimpleCrossingS9N1-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 1, 2)},
        {"MiniGrid-SimpleCrossingS9N2-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 2, 2)},
        {"MiniGrid-SimpleCrossingS9N3-v0", mk(9, 9, 324, 0, 0, MGX_LEVEL_CROSSING, 3, 2)},
        {"MiniGrid-SimpleCrossingS11N5-v0", mk(11, 11, 484, 0, 0, MGX_LEVEL_CROSSING, 5, 2)},
        // FetchEnv / GoToDoorEnv: max_steps = 5*size^2, see_through_walls=True; task rules (envs/fetch.py:74-86, gotodoor.py:71-93)
        {"MiniGrid-Fetch-5x5-N2-v0", mkt(5, 5, 125, 1, MGX_LEVEL_FETCH, 2, MGX_TASK_FETCH)},
        {"MiniGrid-Fetch-6x6-N2-v0", mkt(6, 6, 180, 1, MGX_LEVEL_FETCH, 2, MGX_TASK_FETCH)},
        {"MiniGrid-Fetch-8x8-N3-v0", mkt(8, 8, 320, 1, MGX_LEVEL_FETCH, 3, MGX_TASK_FETCH)},
        {"MiniGrid-GoToDoor-5x5-v0", mkt(5, 5, 125, 1, MGX_LEVEL_GOTODOOR, 0, MGX_TASK_GOTODOOR)},
        {"MiniGrid-GoToDoor-6x6-v0", mkt(6, 6, 180, 1, MGX_LEVEL_GOTODOOR, 0, MGX_TASK_GOTODOOR)},
        {"MiniGrid-GoToDoor-8x8-v0", mkt(8, 8, 320, 1, MGX_LEVEL_GOTODOOR, 0, MGX_TASK_GOTODOOR)},
        // DynamicObstaclesEnv: max_steps = 4*size^2, see_through_walls=True; n_obstacles after the clamp of
        // envs/dynamicobstacles.py:22-26 (5x5: 2, 6x6: 3, 8x8: 4, 16x16: 8); '-Random-' = agent_start_pos=None
        {"MiniGrid-Dynamic-Obstacles-5x5-v0", mkd(5, 2, 0)},
        {"MiniGrid-Dynamic-Obstacles-Random-5x5-v0", mkd(5, 2, 1)},
        {"MiniGrid-Dynamic-Obstacles-6x6-v0", mkd(6, 3, 0)},
        {"MiniGrid-Dynamic-Obstacles-Random-6x6-v0", mkd(6, 3, 1)},
        {"MiniGrid-Dynamic-Obstacles-8x8-v0", mkd(8, 4, 0)},
        {"MiniGrid-Dynamic-Obstacles-16x16-v0", mkd(16, 8, 0)},
        // GoToObjectEnv: max_steps = 5*size^2, see_through_walls=True (envs/gotoobject.py:10-22)
        {"MiniGrid-GoToObject-6x6-N2-v0", mkt(6, 6, 180, 1, MGX_LEVEL_GOTOOBJECT, 2, MGX_TASK_GOTOOBJECT)},
        {"MiniGrid-GoToObject-8x8-N2-v0", mkt(8, 8, 320, 1, MGX_LEVEL_GOTOOBJECT, 2, MGX_TASK_GOTOOBJECT)},
        // RedBlueDoorEnv: 2*size x size, max_steps = 20*size^2 (envs/redbluedoors.py:11-18)
        {"MiniGrid-RedBlueDoors-6x6-v0", mkt(12, 6, 720, 0, MGX_LEVEL_REDBLUEDOORS, 0, MGX_TASK_REDBLUEDOORS)},
        {"MiniGrid-RedBlueDoors-8x8-v0", mkt(16, 8, 1280, 0, MGX_LEVEL_REDBLUEDOORS, 0, MGX_TASK_REDBLUEDOORS)},
        // MemoryEnv: max_steps = 5*size^2 (envs/memory.py:14-27,103-154)
        {"MiniGrid-MemoryS7-v0", mkt(7, 7, 245, 0, MGX_LEVEL_MEMORY, 0, MGX_TASK_MEMORY)},
        {"MiniGrid-MemoryS9-v0", mkt(9, 9, 405, 0, MGX_LEVEL_MEMORY, 0, MGX_TASK_MEMORY)},
        {"MiniGrid-MemoryS11-v0", mkt(11, 11, 605, 0, MGX_LEVEL_MEMORY, 0, MGX_TASK_MEMORY)},
        {"MiniGrid-MemoryS13-v0", mkt(13, 13, 845, 0, MGX_LEVEL_MEMORY, 0, MGX_TASK_MEMORY)},
        {"MiniGrid-MemoryS13Random-v0", mkt(13, 13, 845, 0, MGX_LEVEL_MEMORY, 1, MGX_TASK_MEMORY)},
        {"MiniGrid-MemoryS17Random-v0", mkt(17, 17, 1445, 0, MGX_LEVEL_MEMORY, 1, MGX_TASK_MEMORY)},
        // Unlock / UnlockPickup / BlockedUnlockPickup: RoomGrid 1x2 of 6x6 rooms, max_steps = 8 (16) * room_size^2
        {"MiniGrid-Unlock-v0", mkt(11, 6, 288, 0, MGX_LEVEL_UNLOCK, 0, MGX_TASK_UNLOCK)},
        {"MiniGrid-UnlockPickup-v0", mkt(11, 6, 288, 0, MGX_LEVEL_UNLOCK, 1, MGX_TASK_PICKUPBOX)},
        {"MiniGrid-BlockedUnlockPickup-v0", mkt(11, 6, 576, 0, MGX_LEVEL_UNLOCK, 2, MGX_TASK_PICKUPBOX)},
        // KeyCorridor: RoomGrid 3 x rows of room_size S, max_steps = 30*S^2 (envs/keycorridor.py:10-24,61-106)
        {"MiniGrid-KeyCorridorS3R1-v0", mkt(7, 3, 270, 0, MGX_LEVEL_KEYCORRIDOR, 3, MGX_TASK_PICKUPBOX)},
        {"MiniGrid-KeyCorridorS3R2-v0", mkt(7, 5, 270, 0, MGX_LEVEL_KEYCORRIDOR, 3, MGX_TASK_PICKUPBOX)},
        {"MiniGrid-KeyCorridorS3R3-v0", mkt(7, 7, 270, 0, MGX_LEVEL_KEYCORRIDOR, 3, MGX_TASK_PICKUPBOX)},
        {"MiniGrid-KeyCorridorS4R3-v0", mkt(10, 10, 480, 0, MGX_LEVEL_KEYCORRIDOR, 4, MGX_TASK_PICKUPBOX)},
        {"MiniGrid-KeyCorridorS5R3-v0", mkt(13, 13, 750, 0, MGX_LEVEL_KEYCORRIDOR, 5, MGX_TASK_PICKUPBOX)},
        {"MiniGrid-KeyCorridorS6R3-v0", mkt(16, 16, 1080, 0, MGX_LEVEL_KEYCORRIDOR, 6, MGX_TASK_PICKUPBOX)},
        // TwoGoalsEnv: max_steps = size^2, see_through_walls=True (envs/twogoals.py:9-29,148-222; '-9x9-v0' names a missing class)
        {"MiniGrid-TwoGoals-5x5-v0", mkt(5, 5, 25, 1, MGX_LEVEL_TWOGOALS, 0, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-Random-5x5-v0", mkt(5, 5, 25, 1, MGX_LEVEL_TWOGOALS, 1, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-6x6-v0", mkt(6, 6, 36, 1, MGX_LEVEL_TWOGOALS, 0, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-Random-6x6-v0", mkt(6, 6, 36, 1, MGX_LEVEL_TWOGOALS, 1, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-8x8-v0", mkt(8, 8, 64, 1, MGX_LEVEL_TWOGOALS, 0, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-Random-9x9-v0", mkt(9, 9, 81, 1, MGX_LEVEL_TWOGOALS, 1, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-16x16-v0", mkt(16, 16, 256, 1, MGX_LEVEL_TWOGOALS, 0, MGX_TASK_TWOGOALS)},
        {"MiniGrid-TwoGoals-Random-16x16-v0", mkt(16, 16, 256, 1, MGX_LEVEL_TWOGOALS, 1, MGX_TASK_TWOGOALS)},
        // ObstructedMaze (envs/obstructedmaze.py:76-223): flags bit 0 = key in box, bit 1 = door blocked; layout = quarters [| 8: agent in room (2, 1)]
        {"MiniGrid-ObstructedMaze-1Dl-v0", mko(11, 6, 2, 0, 0)},
        {"MiniGrid-ObstructedMaze-1Dlh-v0", mko(11, 6, 2, 1, 0)},
        {"MiniGrid-ObstructedMaze-1Dlhb-v0", mko(11, 6, 2, 3, 0)},
        {"MiniGrid-ObstructedMaze-2Dl-v0", mko(16, 16, 4, 0, 1 | 8)},
        {"MiniGrid-ObstructedMaze-2Dlh-v0", mko(16, 16, 4, 1, 1 | 8)},
        {"MiniGrid-ObstructedMaze-2Dlhb-v0", mko(16, 16, 4, 3, 1 | 8)},
        {"MiniGrid-ObstructedMaze-1Q-v0", mko(16, 16, 5, 3, 1)},
        {"MiniGrid-ObstructedMaze-2Q-v0", mko(16, 16, 11, 3, 2)},
        {"MiniGrid-ObstructedMaze-Full-v0", mko(16, 16, 25, 3, 4)},
        // PutNearEnv: max_steps = 5*size, see_through_walls=True (envs/putnear.py:10-22,112-126)
        {"MiniGrid-PutNear-6x6-N2-v0", mkt(6, 6, 30, 1, MGX_LEVEL_PUTNEAR, 2, MGX_TASK_PUTNEAR)},
        {"MiniGrid-PutNear-8x8-N3-v0", mkt(8, 8, 40, 1, MGX_LEVEL_PUTNEAR, 3, MGX_TASK_PUTNEAR)},
        // PlaygroundV0: 19x19, max_steps = 100 (envs/playground_v0.py:10-11)
        {"MiniGrid-Playground-v0", mk(19, 19, 100, 0, 0, MGX_LEVEL_PLAYGROUND, 0, 0)},
        // LockedRoom: 19x19, max_steps = 10*size (envs/lockedroom.py:32-35)
        {"MiniGrid-LockedRoom-v0", mkt(19, 19, 190, 0, MGX_LEVEL_LOCKEDROOM, 0, MGX_TASK_NOTE)},
        // FourRoomsEnv: 19x19, max_steps=500 (envs/fourrooms.py:14-17)
        {"MiniGrid-FourRooms-v0", mk(19, 19, 500, 0, 0, MGX_LEVEL_FOURROOMS, 0, 0)},
        // MultiRoomEnv: 25x25, max_steps = maxNumRooms*20 (envs/multiroom.py:36-39,223-246)
        {"MiniGrid-MultiRoom-N2-S4-v0", mk(25, 25, 40, 0, 0, MGX_LEVEL_MULTIROOM, 2 | (2 << 8), 4)},
        {"MiniGrid-MultiRoom-N4-S5-v0", mk(25, 25, 80, 0, 0, MGX_LEVEL_MULTIROOM, 4 | (4 << 8), 5)},
        {"MiniGrid-MultiRoom-N6-v0", mk(25, 25, 120, 0, 0, MGX_LEVEL_MULTIROOM, 6 | (6 << 8), 10)},
        // DistShiftEnv: 7x7 in this fork, max_steps = 4*W*H, see_through_walls=True; class DistShiftv1 has 'v1' in its name
        {"MiniGrid-DistShift1-v0", mk(7, 7, 196, 1, 0, MGX_LEVEL_DISTSHIFT, 2, 0)},
        {"MiniGrid-DistShift1-v1", mk(7, 7, 196, 1, 1, MGX_LEVEL_DISTSHIFT, 2, 0)},
        {"MiniGrid-DistShift2-v0", mk(7, 7, 196, 1, 0, MGX_LEVEL_DISTSHIFT, 5, 0)},
        // LavaGapEnv: max_steps = 4*size^2; 'v1' classes are const-gap AND change lava semantics
        // (envs/lavagap.py:10-19,62-85; minigrid.py:1263)
        {"MiniGrid-LavaGapS5-v0", mk(5, 5, 100, 0, 0, MGX_LEVEL_LAVAGAP, 0, 9)},
        {"MiniGrid-LavaGapS6-v0", mk(6, 6, 144, 0, 0, MGX_LEVEL_LAVAGAP, 0, 9)},
        {"MiniGrid-LavaGapS7-v0", mk(7, 7, 196, 0, 0, MGX_LEVEL_LAVAGAP, 0, 9)},
        {"MiniGrid-NormalGapS6-v0", mk(6, 6, 144, 0, 0, MGX_LEVEL_LAVAGAP, 0, 2)},
        {"MiniGrid-LavaGapS6-v1", mk(6, 6, 144, 0, 1, MGX_LEVEL_LAVAGAP, 1, 9)},
        {"MiniGrid-LavaGapS7-v1", mk(7, 7, 196, 0, 1, MGX_LEVEL_LAVAGAP, 1, 9)},
    };
    return R;
}

} // namespace

extern "C" int mgx_env_config(const char *env_id, mgx_config *cfg)
{
    if (!env_id || !cfg) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_env_config: null argument");
    for (auto &e : registry())
        if (!strcmp(e.id, env_id)) { *cfg = e.cfg; return MGX_OK; }
    return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_env_config: unknown env id '%s'", env_id);
}

extern "C" int mgx_mission(const mgx_config *cfg, uint32_t task, char *out, int cap)
{
    if (!cfg || !out || cap < 1) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: null argument");
    static const char *const colors[7] = {"red", "green", "blue", "purple", "yellow", "grey", "white"}; // COLOR_TO_IDX (minigrid.py:27-35)
    static const char *const fetch_tmpl[5] = {"get a %s %s", "go get a %s %s", "fetch a %s %s", "go fetch a %s %s", "you must fetch a %s %s"};
    const bool lava = (cfg->level_arg1 & 15) == MGX_K_LAVA;
    char buf[128];
    const char *m = "";
    switch (cfg->level_kind) {
    case MGX_LEVEL_EMPTY: case MGX_LEVEL_DISTSHIFT: m = "get to the green goal square"; break;           // envs/empty.py:57, distshift.py:52
    case MGX_LEVEL_DOORKEY: m = "use the key to open the door and then get to the goal"; break;          // envs/doorkey.py:44
    case MGX_LEVEL_CROSSING: case MGX_LEVEL_LAVAGAP:                                                        // envs/crossing.py:88-92, lavagap.py:56-60
        m = lava ? "avoid the lava and get to the green goal square" : "find the opening and get to the green goal square"; break;
    case MGX_LEVEL_MULTIROOM: m = "traverse the rooms to get to the goal"; break;                           // envs/multiroom.py:117
    case MGX_LEVEL_GOTODOOR: m = "go to the red door"; break;                                               // envs/gotodoor.py:70 (the fork's target is always red)
    case MGX_LEVEL_DYNOBS: m = "get to the green goal square"; break;                                      // envs/dynamicobstacles.py:58
    case MGX_LEVEL_REDBLUEDOORS: m = "open the red door then the blue door"; break;                        // envs/redbluedoors.py:42
    case MGX_LEVEL_MEMORY: m = "go to the matching object at the end of the hallway"; break;              // envs/memory.py:86
    case MGX_LEVEL_UNLOCK:                                                                                  // envs/unlock.py:31, unlockpickup.py:33
    case MGX_LEVEL_KEYCORRIDOR:                                                                             // keycorridor.py:49
        if (cfg->level_kind == MGX_LEVEL_UNLOCK && cfg->level_arg0 == 0) m = "open the door";
        else {
            const uint32_t kind = task & 15u, color = (task >> 4) & 7u;
            if ((kind != MGX_K_BOX && kind != MGX_K_BALL && kind != MGX_K_KEY) || color > 6)
                return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: 0x%x is not a pick-up target", task);
            snprintf(buf, sizeof buf, "pick up the %s %s", colors[color], kind == MGX_K_BOX ? "box" : (kind == MGX_K_BALL ? "ball" : "key"));
            m = buf;
        }
        break;
    case MGX_LEVEL_LOCKEDROOM: {                                                                            // envs/lockedroom.py:108-112
        const uint32_t lc = task & 7u, kc = (task >> 3) & 7u;
        if (lc > 6 || kc > 6) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: 0x%x is not a LockedRoom task word", task);
        snprintf(buf, sizeof buf, "get the %s key from the %s room, unlock the %s door and go to the goal", colors[lc], colors[kc], colors[lc]);
        m = buf;
        break;
    }
    case MGX_LEVEL_PUTNEAR: {                                                                               // envs/putnear.py:84-89
        static const char *const types[3] = {"key", "ball", "box"};
        const uint32_t mt = task & 3u, mc = (task >> 2) & 7u, tt = (task >> 11) & 3u, tc = (task >> 13) & 7u;
        if (mt > 2 || tt > 2 || mc > 6 || tc > 6) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: 0x%x is not a PutNear task word", task);
        snprintf(buf, sizeof buf, "put the %s %s near the %s %s", colors[mc], types[mt], colors[tc], types[tt]);
        m = buf;
        break;
    }
    case MGX_LEVEL_OBSTRUCTEDMAZE: m = "pick up the blue ball"; break;                                      // envs/obstructedmaze.py:40 (COLOR_NAMES[0])
    case MGX_LEVEL_TWOGOALS: m = "get to the green or red goal square"; break;                              // envs/twogoals.py:50
    case MGX_LEVEL_FOURROOMS: m = "Reach the goal"; break;                                                  // envs/fourrooms.py:69
    case MGX_LEVEL_GOTOOBJECT: {                                                                            // envs/gotoobject.py:63-64
        static const char *const types[3] = {"key", "ball", "box"};
        const uint32_t ty = (task >> 8) & 3u, color = (task >> 10) & 7u;
        if (ty > 2 || color > 6) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: 0x%x is not a GoToObject task word", task);
        snprintf(buf, sizeof buf, "go to the %s %s", colors[color], types[ty]);
        m = buf;
        break;
    }
    case MGX_LEVEL_FETCH: {                                                                                 // envs/fetch.py:57-71
        const uint32_t kind = task & 15u, color = (task >> 4) & 7u, tmpl = task >> 8;
        if ((kind != MGX_K_KEY && kind != MGX_K_BALL) || color > 6 || tmpl > 4 || (task & 0x80u))
            return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: 0x%x is not a Fetch task word", task);
        snprintf(buf, sizeof buf, fetch_tmpl[tmpl], colors[color], kind == MGX_K_KEY ? "key" : "ball");
        m = buf;
        break;
    }
    default: break;
    }
    const int len = (int)strlen(m);
    if (len + 1 > cap) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_mission: buffer of %d bytes for a %d-character mission", cap, len);
    memcpy(out, m, (size_t)len + 1);
    return len;
}

extern "C" const char *mgx_env_id(int i)
{
    auto &R = registry();
    return (i >= 0 && i < (int)R.size()) ? R[i].id : nullptr;
}

extern "C" int mgx_generate_levels(const mgx_config *cfg, int64_t n, const uint64_t *seeds, uint8_t *grid, int32_t *agent)
{
    return mgx_generate_levels_ex(cfg, n, seeds, grid, agent, nullptr);
}

extern "C" int mgx_generate_levels_ex(const mgx_config *cfg, int64_t n, const uint64_t *seeds, uint8_t *grid, int32_t *agent, uint32_t *task)
{
    return mgx_generate_levels_full(cfg, n, seeds, grid, agent, task, nullptr);
}

extern "C" int mgx_generate_levels_full(const mgx_config *cfg, int64_t n, const uint64_t *seeds, uint8_t *grid, int32_t *agent, uint32_t *task,
                                        uint8_t *contains)
{
    if (!cfg || !seeds || !grid || !agent || n < 0) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_generate_levels: null argument");
    int rc = check_levelgen_cfg(cfg, "mgx_generate_levels");
    if (rc) return rc;
    const int cells = cfg->width * cfg->height;
    std::vector<uint8_t> codes((size_t)cells);
    Rng rng;
    for (int64_t e = 0; e < n; e++) {
        LgLevel L;
        int16_t ws[MGX_LG_WS_WORDS];
        LgCmd cmds[MGX_LG_MAX_CMDS];
        L.cmds = cmds; L.ncmd = 0; L.W = cfg->width; L.H = cfg->height; L.ax = L.ay = -1; L.adir = 0; L.ws = ws;
        if (lg_uses_rng(*cfg)) rng.seed_gym(seeds[e]); // Empty with a fixed start consumes no randomness: skip the seeding
        lg_generate(*cfg, rng, L);
        lg_paint(L, codes.data());
        for (auto &cd : codes) if (MGX_IS_OBSTACLE_MARK(cd)) cd = (uint8_t)MGX_CODE_BALL_BLUE; // DynObs order markers
        codes_to_triples(codes.data(), cells, grid + (size_t)e * cells * 3);
        if (contains) {
            lg_paint_contains(L, codes.data());
            codes_to_triples(codes.data(), cells, contains + (size_t)e * cells * 3);
        }
        agent[e * 3] = L.ax; agent[e * 3 + 1] = L.ay; agent[e * 3 + 2] = L.adir;
        if (task) task[e] = L.task;
    }
    return MGX_OK;
}

// `env.seed(seed)` once, then K consecutive `env.reset()`s: the env's RNG stream continues across episodes
// (plain reference behaviour without ReseedWrapper, minigrid.py:836-839).  grid u8[K][W][H][3], agent i32[K][3].
extern "C" int mgx_generate_level_stream(const mgx_config *cfg, uint64_t seed, int64_t K, uint8_t *grid, int32_t *agent)
{
    return mgx_generate_level_stream_ex(cfg, seed, K, grid, agent, nullptr);
}

extern "C" int mgx_generate_level_stream_ex(const mgx_config *cfg, uint64_t seed, int64_t K, uint8_t *grid, int32_t *agent, uint32_t *task)
{
    return mgx_generate_level_stream_full(cfg, seed, K, grid, agent, task, nullptr);
}

extern "C" int mgx_generate_level_stream_full(const mgx_config *cfg, uint64_t seed, int64_t K, uint8_t *grid, int32_t *agent, uint32_t *task,
                                              uint8_t *contains)
{
    if (!cfg || !grid || !agent || K < 0) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_generate_level_stream: null argument");
    int rc = check_levelgen_cfg(cfg, "mgx_generate_level_stream");
    if (rc) return rc;
    const int cells = cfg->width * cfg->height;
    std::vector<uint8_t> codes((size_t)cells);
    Rng rng;
    rng.seed_gym(seed);
    for (int64_t k = 0; k < K; k++) {
        LgLevel L;
        int16_t ws[MGX_LG_WS_WORDS];
        LgCmd cmds[MGX_LG_MAX_CMDS];
        L.cmds = cmds; L.ncmd = 0; L.W = cfg->width; L.H = cfg->height; L.ax = L.ay = -1; L.adir = 0; L.ws = ws;
        lg_generate(*cfg, rng, L);
        lg_paint(L, codes.data());
        for (auto &cd : codes) if (MGX_IS_OBSTACLE_MARK(cd)) cd = (uint8_t)MGX_CODE_BALL_BLUE; // DynObs order markers
        codes_to_triples(codes.data(), cells, grid + (size_t)k * cells * 3);
        if (contains) {
            lg_paint_contains(L, codes.data());
            codes_to_triples(codes.data(), cells, contains + (size_t)k * cells * 3);
        }
        agent[k * 3] = L.ax; agent[k * 3 + 1] = L.ay; agent[k * 3 + 2] = L.adir;
        if (task) task[k] = L.task;
    }
    return MGX_OK;
}

// init_genrand(19650218): the seed-independent first pass of init_by_array, used by the device seeding kernel
void mgx_mt_init_table(uint32_t out[624])
{
    Rng r;
    r.init_genrand(19650218U);
    memcpy(out, r.mt, sizeof r.mt);
}
