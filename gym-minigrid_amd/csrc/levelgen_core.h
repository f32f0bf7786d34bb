// levelgen_core.h -- the seeded level generators, written once for host (levelgen.cpp) and device (k_levelgen in
// k_levelgen.hip).  Everything here is `__host__ __device__`, allocation-free and works on 1-byte cell codes
// (mgx_internal.h), x-major like Grid.encode().
//
// Reference: EmptyEnv._gen_grid     /root/reference/gym_minigrid/envs/empty.py:30-57
//            DoorKeyEnv._gen_grid   /root/reference/gym_minigrid/envs/doorkey.py:15-44
//            CrossingEnv._gen_grid  /root/reference/gym_minigrid/envs/crossing.py:24-92
//            LavaGapEnv._gen_grid   /root/reference/gym_minigrid/envs/lavagap.py:21-59
//            place_obj/place_agent  /root/reference/gym_minigrid/minigrid.py:1003-1090
// Random draws follow numpy's legacy RandomState on top of MT19937 (see levelgen.cpp's header): every draw is
// "raw 32-bit output & mask, redraw while > max".  `R` is any type with `uint32_t next32()` and `bool alive()`
// (false once a bounded source of words has run dry: the rejection loops then stop and the caller discards the level).
#ifndef MGX_LEVELGEN_CORE_H
#define MGX_LEVELGEN_CORE_H

#include <stdint.h>

#include "mgx.h"
#include "mgx_internal.h"

#if defined(__HIPCC__) /* hipcc: the generators also compile for the gfx950 device (k_levelgen, k_dynobs) */
#define LG_FN __host__ __device__ inline
#else
#define LG_FN inline
#endif

#define MGX_CODE_GOAL_GREEN (MGX_K_GOAL | (1 << 4))        /* Goal(): green            */
#define MGX_CODE_LAVA (MGX_K_LAVA)                         /* Lava(): red = 0          */
#define MGX_CODE_DOOR_YELLOW_LOCKED (MGX_K_DOOR_LOCKED | (4 << 4))
#define MGX_CODE_KEY_YELLOW (MGX_K_KEY | (4 << 4))
#define MGX_CODE_BALL_BLUE (MGX_K_BALL | (2u << 4))
#define MGX_LG_MAX_RIVERS 32

#define MGX_LG_WS_WORDS (6 * MGX_LG_MAX_RIVERS)
#define MGX_LG_MAX_CMDS (8 + 4 * MGX_LG_MAX_RIVERS)

// A level is described by a short list of paint commands (inclusive rectangles, later ones override earlier ones)
// instead of being drawn cell by cell: the sequential part of a generator -- the random decisions -- then touches a
// few bytes only, queries like "is this cell free?" scan the list, and the grid itself is painted afterwards, on the
// GPU by all 64 lanes in parallel (k_levelgen spent most of its time drawing walls with one lane before this).
struct alignas(4) LgCmd { uint8_t x0, y0, x1, y1, code, cont, pad[2]; }; // two dwords (never 8-byte accesses: GPU slices are 4-byte aligned)
// (cont: cell code of Box.contains for a one-cell command that paints a box, 0 = nothing inside)

struct LgLevel {
    LgCmd *cmds; // max_cmds entries (MGX_LG_MAX_CMDS always suffices)
    int ncmd;
    int W, H;
    int ax, ay, adir;
    int16_t *ws; // 6*max_rivers scratch words for the crossing generator's lists.  On the GPU cmds and ws point
                 // into LDS: dynamically indexed local arrays would live in scratch (HBM) and every access of the
                 // sequential generator would pay a memory round trip (measured 10x on k_levelgen).
    uint32_t task = 0; // per-env task word of the families with a task rule (Fetch: target code | mission template << 8)
    int max_cmds = MGX_LG_MAX_CMDS, max_rivers = MGX_LG_MAX_RIVERS; // capacities of cmds / ws (the GPU fast path has small ones)
    bool too_big = false; // a capacity was exceeded: the level is invalid and must be regenerated with full-size buffers
    uint8_t *occ = nullptr; // (GPU lane-per-level path, small grids) the level painted as it grows: W*H cell codes, x*H + y, all EMPTY at the start.
                            // The placement loops probe a cell per sample; against the command list that is a scan of every command so far
                            // (KeyCorridor: ~40 per probe), against the image one byte.  Null: the scan (host, wave-per-level path, big grids).
};

LG_FN void lg_rect(LgLevel &L, int x0, int y0, int x1, int y1, uint32_t code)
{
    if (L.ncmd >= L.max_cmds) { L.too_big = true; return; }
    LgCmd c;
    c.x0 = (uint8_t)x0; c.y0 = (uint8_t)y0; c.x1 = (uint8_t)x1; c.y1 = (uint8_t)y1; c.code = (uint8_t)code;
    c.cont = 0; c.pad[0] = c.pad[1] = 0;
    L.cmds[L.ncmd++] = c;
    if (L.occ)
        for (int x = x0; x <= x1; x++)
            for (int y = y0; y <= y1; y++) L.occ[x * L.H + y] = (uint8_t)code;
}
LG_FN void lg_set(LgLevel &L, int x, int y, uint32_t code) { lg_rect(L, x, y, x, y, code); }
// a Box with something inside (Box.contains, minigrid.py:332-364): handles created with object_state keep it in a plane of its own
LG_FN void lg_set_box(LgLevel &L, int x, int y, uint32_t code, uint32_t contains)
{
    lg_rect(L, x, y, x, y, code);
    if (!L.too_big) L.cmds[L.ncmd - 1].cont = (uint8_t)contains;
}

// code of cell (x, y) under the first n commands (an unpainted cell is empty)
LG_FN uint32_t lg_cell_code(const LgCmd *cmds, int n, int x, int y)
{
    uint32_t code = MGX_CODE_EMPTY;
    for (int i = 0; i < n; i++) {
        const LgCmd c = cmds[i];
        if (x >= c.x0 && x <= c.x1 && y >= c.y0 && y <= c.y1) code = c.code;
    }
    return code;
}
// ... of the level so far
LG_FN uint32_t lg_code_at(const LgLevel &L, int x, int y)
{
    if (L.occ) return ((unsigned)x < (unsigned)L.W && (unsigned)y < (unsigned)L.H) ? (uint32_t)L.occ[x * L.H + y] : (uint32_t)MGX_CODE_EMPTY;
    return lg_cell_code(L.cmds, L.ncmd, x, y);
}
LG_FN bool lg_empty(const LgLevel &L, int x, int y) { return lg_code_at(L, x, y) == MGX_CODE_EMPTY; }
// code of what the box in cell (x, y) contains (MGX_CODE_EMPTY: nothing, or no box there)
LG_FN uint32_t lg_cell_cont(const LgCmd *cmds, int n, int x, int y)
{
    uint32_t cont = 0;
    for (int i = 0; i < n; i++) {
        const LgCmd c = cmds[i];
        if (x >= c.x0 && x <= c.x1 && y >= c.y0 && y <= c.y1) cont = c.cont;
    }
    return cont ? cont : (uint32_t)MGX_CODE_EMPTY;
}

// serial paint (host): g[x*H + y]
LG_FN void lg_paint(const LgLevel &L, uint8_t *g)
{
    for (int x = 0; x < L.W; x++)
        for (int y = 0; y < L.H; y++) g[x * L.H + y] = (uint8_t)lg_cell_code(L.cmds, L.ncmd, x, y);
}
LG_FN void lg_paint_contains(const LgLevel &L, uint8_t *g)
{
    for (int x = 0; x < L.W; x++)
        for (int y = 0; y < L.H; y++) g[x * L.H + y] = (uint8_t)lg_cell_cont(L.cmds, L.ncmd, x, y);
}

LG_FN int lg_ctz32(uint32_t x) { int n = 0; while (!(x & 1u)) { x >>= 1; n++; } return n; } // (x != 0)

// uniform integer in [0, max] by masked rejection (numpy legacy bounded_uint32)
template <class R>
LG_FN uint32_t lg_bounded(R &r, uint32_t max)
{
    if (max == 0) return 0;
    const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(max); // smallest 2^k - 1 >= max (max != 0): the smear of max's top bit in two instructions
    uint32_t v;
    do { v = r.next32() & mask; } while (v > max);
    return v;
}
template <class R>
LG_FN int lg_randint(R &r, int lo, int hi) { return lo + (int)lg_bounded(r, (uint32_t)(hi - lo - 1)); }

// empty room inside a grey wall rectangle, green goal in the bottom-right corner
LG_FN void lg_room(LgLevel &L)
{
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_set(L, L.W - 2, L.H - 2, MGX_CODE_GOAL_GREEN);
}

// place_obj(obj=None, top=(0,0), size=(sw,sh)) rejection sampling (minigrid.py:1028-1053)
template <class R>
LG_FN void lg_sample_free(R &r, LgLevel &L, int sw, int sh, bool reject_agent, int *ox, int *oy)
{
    const int xw = sw < L.W ? sw : L.W, yh = sh < L.H ? sh : L.H;
    for (;;) {
        const int x = lg_randint(r, 0, xw);
        const int y = lg_randint(r, 0, yh);
        if (!r.alive()) { *ox = 0; *oy = 0; return; } // the word source ran dry: this level is discarded by the caller
        if (!lg_empty(L, x, y)) continue;
        if (reject_agent && x == L.ax && y == L.ay) continue;
        *ox = x; *oy = y;
        return;
    }
}

template <class R>
LG_FN void lg_gen_empty(const mgx_config &c, R &r, LgLevel &L)
{
    lg_room(L);
    if (c.level_arg0 == 0) { L.ax = 1; L.ay = 1; L.adir = 0; }
    else { // agent_start_pos=None -> place_agent(size=sizetop)
        const int sw = c.level_arg1 > 0 ? c.level_arg1 : L.W, sh = c.level_arg1 > 0 ? c.level_arg1 : L.H;
        L.ax = -1; L.ay = -1;
        lg_sample_free(r, L, sw, sh, false, &L.ax, &L.ay);
        L.adir = lg_randint(r, 0, 4);
    }
}

template <class R>
LG_FN void lg_gen_doorkey(const mgx_config &, R &r, LgLevel &L)
{
    lg_room(L);
    const int split = lg_randint(r, 2, L.W - 2);
    lg_rect(L, split, 0, split, L.H - 1, MGX_CODE_WALL_GREY);              // vert_wall(split, 0)
    L.ax = -1; L.ay = -1;
    lg_sample_free(r, L, split, L.H, false, &L.ax, &L.ay);                 // place_agent(size=(split, H))
    L.adir = lg_randint(r, 0, 4);
    const int door = lg_randint(r, 1, L.W - 2);
    lg_set(L, split, door, MGX_CODE_DOOR_YELLOW_LOCKED);                   // Door('yellow', is_locked=True)
    int kx, ky;
    lg_sample_free(r, L, split, L.H, true, &kx, &ky);                      // Key('yellow'), rejects the agent cell
    lg_set(L, kx, ky, MGX_CODE_KEY_YELLOW);
}

template <class R>
LG_FN void lg_gen_crossing(const mgx_config &c, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H, ncross = c.level_arg0;
    const uint32_t obst = (c.level_arg1 & 15) == 2 ? (uint32_t)MGX_CODE_WALL_GREY : (uint32_t)MGX_CODE_LAVA;
    const int ori = c.level_arg1 >> 4; // 0 = both (ori=2), 1 = horizontal rivers only (ori=0), 2 = vertical only (ori=1)
    lg_room(L);
    L.ax = 1; L.ay = 1; L.adir = 0;
    // candidate rivers: (v, i) for i in range(2, H-2, 2) then (h, j) for j in range(2, W-2, 2); bit 8 = vertical
    const int MR = L.max_rivers;
    if ((H - 3) / 2 > MR || (W - 3) / 2 > MR) { L.too_big = true; return; }
    int16_t *riv = L.ws, *rv = L.ws + 2 * MR, *rh = L.ws + 3 * MR, *path = L.ws + 4 * MR;
    int n = 0;
    if (ori != 1) for (int i = 2; i < H - 2; i += 2) riv[n++] = (int16_t)(0x100 | i);
    if (ori != 2) for (int j = 2; j < W - 2; j += 2) riv[n++] = (int16_t)j;
    for (int i = n - 1; i >= 1; i--) { // np_random.shuffle(list): j = bounded(i), swap
        const int j = (int)lg_bounded(r, (uint32_t)i);
        const int16_t t = riv[i]; riv[i] = riv[j]; riv[j] = t;
    }
    if (n > ncross) n = ncross;
    int nv = 0, nh = 0;
    for (int i = 0; i < n; i++) { if (riv[i] & 0x100) rv[nv++] = (int16_t)(riv[i] & 0xFF); else rh[nh++] = riv[i]; }
    for (int i = 1; i < nv; i++) { const int16_t t = rv[i]; int j = i - 1; while (j >= 0 && rv[j] > t) { rv[j + 1] = rv[j]; j--; } rv[j + 1] = t; } // sorted()
    for (int i = 1; i < nh; i++) { const int16_t t = rh[i]; int j = i - 1; while (j >= 0 && rh[j] > t) { rh[j + 1] = rh[j]; j--; } rh[j + 1] = t; }
    for (int k = 0; k < nh; k++) lg_rect(L, 1, rh[k], W - 2, rh[k], obst); // product(range(1,W-1), rivers_h)
    for (int k = 0; k < nv; k++) lg_rect(L, rv[k], 1, rv[k], H - 2, obst); // product(rivers_v, range(1,H-1))
    int np_ = 0; // path: 1 = h step (crosses a vertical river), 0 = v step
    for (int i = 0; i < nv; i++) path[np_++] = 1;
    for (int i = 0; i < nh; i++) path[np_++] = 0;
    for (int i = np_ - 1; i >= 1; i--) {
        const int j = (int)lg_bounded(r, (uint32_t)i);
        const int16_t t = path[i]; path[i] = path[j]; path[j] = t;
    }
    // limits_v = [0] + rivers_v + [H-1], limits_h = [0] + rivers_h + [W-1]
    int room_i = 0, room_j = 0;
    for (int k = 0; k < np_; k++) {
        const int lv0 = room_i == 0 ? 0 : rv[room_i - 1], lv1 = room_i < nv ? rv[room_i] : H - 1;
        const int lh0 = room_j == 0 ? 0 : rh[room_j - 1], lh1 = room_j < nh ? rh[room_j] : W - 1;
        int i, j;
        if (path[k] == 1) {
            i = lv1;
            j = lh0 + 1 + lg_randint(r, 0, lh1 - lh0 - 1); // choice(range(lh0+1, lh1))
            room_i++;
        } else {
            i = lv0 + 1 + lg_randint(r, 0, lv1 - lv0 - 1);
            j = lh1;
            room_j++;
        }
        lg_set(L, i, j, MGX_CODE_EMPTY);
    }
}

template <class R>
LG_FN void lg_gen_lavagap(const mgx_config &c, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H;
    const uint32_t obst = (c.level_arg1 & 15) == 2 ? (uint32_t)MGX_CODE_WALL_GREY : (uint32_t)MGX_CODE_LAVA;
    lg_room(L);
    L.ax = 1; L.ay = 1; L.adir = 0;
    int gx, gy;
    if (!c.level_arg0) { gx = lg_randint(r, 2, W - 2); gy = lg_randint(r, 1, H - 1); }
    else { gx = W / 2; gy = lg_randint(r, 1, H - 1); }
    lg_rect(L, gx, 1, gx, H - 2, obst); // vert_wall(gx, 1, H-2, obstacle)
    lg_set(L, gx, gy, MGX_CODE_EMPTY);
}

// DistShiftEnv._gen_grid (envs/distshift.py:30-52): goal at (W-2, 1), two lava strips of W-6 cells from x = 3 in rows 1
// and strip2_row (level_arg0), agent (1,1) facing right.  No randomness.
template <class R>
LG_FN void lg_gen_distshift(const mgx_config &c, R &, LgLevel &L)
{
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_set(L, L.W - 2, 1, MGX_CODE_GOAL_GREEN);
    if (L.W > 6) {
        lg_rect(L, 3, 1, L.W - 4, 1, MGX_CODE_LAVA);
        lg_rect(L, 3, c.level_arg0, L.W - 4, c.level_arg0, MGX_CODE_LAVA);
    }
    L.ax = 1; L.ay = 1; L.adir = 0;
}

// MultiRoomEnv._gen_grid / _placeRoom (envs/multiroom.py:40-219).  level_arg0 = minNumRooms | maxNumRooms << 8,
// level_arg1 = maxRoomSize.  The reference's recursion never removes a placed room and a parent stops at its first
// successful child, so it is a plain chain: place room 0, then up to 8 tries to attach the next room to the last one,
// and so on; a chain that ends early is retried from scratch and the longest one wins (multiroom.py:46-66).
// Rooms live in L.ws as 6 words each (topX, topY, sizeX, sizeY, entryX, entryY): two lists of up to 8 rooms.
// The room search as a state machine, one ROOM TRY per call (the body of _placeRoom's loop): the host runs it in a plain loop, the GPU's
// lane-per-level path runs 16 of them side by side and hands a lane its next level as soon as this one is done (k_levelgen.hip) -- as
// one call per level the lanes of a wave waited for the longest search among them (0.35 lane efficiency).
struct LgMultiRoom {
    int numRooms, maxSz, nbest; // rooms wanted, largest room, length of the best chain so far (its rooms: L.ws[0..6*nbest))
    int n, entryWall, ex, ey, tries; // the chain being built (its rooms: L.ws[48..)): rooms placed, the next room's entry door, exits tried from the parent
    bool fresh;                      // the next call starts a new chain (draws its first entry door)
};

template <class R>
LG_FN bool lg_multiroom_begin(const mgx_config &c, R &r, LgLevel &L, LgMultiRoom &m)
{
    const int minRooms = c.level_arg0 & 255, maxRooms = (c.level_arg0 >> 8) & 255;
    L.ncmd = 0;
    if (6 * L.max_rivers < 96 + 8 || maxRooms > 8) { L.too_big = true; return false; } // needs the full-size workspace
    m.maxSz = c.level_arg1;
    m.numRooms = lg_randint(r, minRooms, maxRooms + 1);
    m.nbest = 0; m.n = 0; m.entryWall = 2; m.ex = m.ey = 0; m.tries = 0; m.fresh = true;
    return true;
}

// one room try; true once a chain of numRooms rooms stands in L.ws[0..) (or the word source ran dry: test r.alive())
template <class R>
LG_FN bool lg_multiroom_step(R &r, LgLevel &L, LgMultiRoom &m)
{
    const int W = L.W, H = L.H;
    int16_t *best = L.ws, *rooms = L.ws + 48;
    if (m.fresh) { // a new chain: its first entry door
        m.n = 0; m.entryWall = 2; m.tries = 0; m.fresh = false;
        m.ex = lg_randint(r, 0, W - 2); m.ey = lg_randint(r, 0, W - 2);
    }
    const int n = m.n, entryWall = m.entryWall, ex = m.ex, ey = m.ey;
    // _placeRoom: size, position relative to the entry door, bounds, overlap with all rooms but the parent
    const int sx = lg_randint(r, 4, m.maxSz + 1), sy = lg_randint(r, 4, m.maxSz + 1);
    int tx, ty;
    if (n == 0) { tx = ex; ty = ey; }
    else if (entryWall == 0) { tx = ex - sx + 1; ty = lg_randint(r, ey - sy + 2, ey); }
    else if (entryWall == 1) { tx = lg_randint(r, ex - sx + 2, ex); ty = ey - sy + 1; }
    else if (entryWall == 2) { tx = ex; ty = lg_randint(r, ey - sy + 2, ey); }
    else { tx = lg_randint(r, ex - sx + 2, ex); ty = ey; }
    bool ok = !(tx < 0 || ty < 0) && !(tx + sx > W || ty + sy >= H);
    for (int k = 0; ok && k < n - 1; k++) {
        const int16_t *q = rooms + 6 * k;
        const bool nonOverlap = tx + sx < q[0] || q[0] + q[2] <= tx || ty + sy < q[1] || q[1] + q[3] <= ty;
        if (!nonOverlap) ok = false;
    }
    bool chain_over = false;
    if (!r.alive()) chain_over = true;
    else if (ok) {
        int16_t *q = rooms + 6 * n;
        q[0] = (int16_t)tx; q[1] = (int16_t)ty; q[2] = (int16_t)sx; q[3] = (int16_t)sy; q[4] = (int16_t)ex; q[5] = (int16_t)ey;
        m.n = n + 1;
        m.tries = 0;
        if (m.n == m.numRooms) chain_over = true; // numLeft == 1
    } else {
        if (n == 0) chain_over = true;             // the first room did not fit: start over
        else if (++m.tries == 8) chain_over = true; // the parent gives up after 8 exit doors
    }
    if (chain_over) { // the longest chain wins (multiroom.py:46-66)
        if (!r.alive()) return true;
        if (m.n > m.nbest) { for (int i = 0; i < 6 * m.n; i++) best[i] = rooms[i]; m.nbest = m.n; }
        m.fresh = true;
        return m.nbest >= m.numRooms;
    }
    // parent = last placed room: pick the exit wall (not its entry wall) and the exit door on it
    const int np = m.n;
    const int16_t *pr = rooms + 6 * (np - 1);
    const int pEntryWall = (np == 1) ? 2 : L.ws[96 + np - 1]; // entry wall of each placed room, kept behind the lists
    int walls[3], nw = 0;
    for (int w = 0; w < 4; w++) if (w != pEntryWall) walls[nw++] = w;
    const int exitWall = walls[lg_randint(r, 0, 3)];
    m.entryWall = (exitWall + 2) % 4;
    if (exitWall == 0) { m.ex = pr[0] + pr[2] - 1; m.ey = pr[1] + lg_randint(r, 1, pr[3] - 1); }
    else if (exitWall == 1) { m.ex = pr[0] + lg_randint(r, 1, pr[2] - 1); m.ey = pr[1] + pr[3] - 1; }
    else if (exitWall == 2) { m.ex = pr[0]; m.ey = pr[1] + lg_randint(r, 1, pr[3] - 1); }
    else { m.ex = pr[0] + lg_randint(r, 1, pr[2] - 1); m.ey = pr[1]; }
    L.ws[96 + np] = (int16_t)m.entryWall; // becomes the entry wall of room np if it gets placed
    return false;
}

template <class R>
LG_FN void lg_multiroom_finish(R &r, LgLevel &L, const LgMultiRoom &m);

template <class R>
LG_FN void lg_gen_multiroom(const mgx_config &c, R &r, LgLevel &L)
{
    LgMultiRoom m;
    if (!lg_multiroom_begin(c, r, L, m)) return;
    while (!lg_multiroom_step(r, L, m)) {}
    if (!r.alive()) return;
    lg_multiroom_finish(r, L, m);
}

template <class R>
LG_FN void lg_multiroom_finish(R &r, LgLevel &L, const LgMultiRoom &m)
{
    const int16_t *best = L.ws;
    const int nbest = m.nbest;
    // draw: walls of every room, then (from the second room on) its entry door in a colour other than the previous door's
    const int sortedColors[7] = {2, 1, 5, 3, 0, 6, 4}; // sorted(COLOR_NAMES): blue green grey purple red white yellow
    int prevColor = -1;
    for (int i = 0; i < nbest; i++) {
        const int16_t *q = best + 6 * i;
        lg_rect(L, q[0], q[1], q[0] + q[2] - 1, q[1], MGX_CODE_WALL_GREY);
        lg_rect(L, q[0], q[1] + q[3] - 1, q[0] + q[2] - 1, q[1] + q[3] - 1, MGX_CODE_WALL_GREY);
        lg_rect(L, q[0], q[1], q[0], q[1] + q[3] - 1, MGX_CODE_WALL_GREY);
        lg_rect(L, q[0] + q[2] - 1, q[1], q[0] + q[2] - 1, q[1] + q[3] - 1, MGX_CODE_WALL_GREY);
        if (i > 0) {
            int k = lg_randint(r, 0, prevColor < 0 ? 7 : 6);
            int color = -1;
            for (int j = 0; j < 7; j++) {
                if (sortedColors[j] == prevColor) continue;
                if (k-- == 0) { color = sortedColors[j]; break; }
            }
            lg_set(L, q[4], q[5], MGX_K_DOOR_CLOSED | ((uint32_t)color << 4)); // Door(color): closed, unlocked
            prevColor = color;
        }
    }
    // place_agent(top, size) in the first room, then the goal in the last room (minigrid.py:1003-1090)
    {
        const int16_t *q = best;
        for (;;) {
            const int x = lg_randint(r, q[0], (q[0] + q[2] < L.W ? q[0] + q[2] : L.W));
            const int y = lg_randint(r, q[1], (q[1] + q[3] < L.H ? q[1] + q[3] : L.H));
            if (!r.alive()) return;
            if (!lg_empty(L, x, y)) continue;
            L.ax = x; L.ay = y;
            break;
        }
        L.adir = lg_randint(r, 0, 4);
        q = best + 6 * (nbest - 1);
        for (;;) {
            const int x = lg_randint(r, q[0], (q[0] + q[2] < L.W ? q[0] + q[2] : L.W));
            const int y = lg_randint(r, q[1], (q[1] + q[3] < L.H ? q[1] + q[3] : L.H));
            if (!r.alive()) return;
            if (!lg_empty(L, x, y)) continue;
            if (x == L.ax && y == L.ay) continue;
            lg_set(L, x, y, MGX_CODE_GOAL_GREEN);
            break;
        }
    }
}

// sorted(COLOR_NAMES) (minigrid.py:24): blue green grey purple red white yellow -> COLOR_TO_IDX
LG_FN int lg_sorted_color(int k)
{
    const int sortedColors[7] = {2, 1, 5, 3, 0, 6, 4};
    return sortedColors[k];
}

// FetchEnv._gen_grid (envs/fetch.py:24-72): numObjs (level_arg0) random keys/balls anywhere, random agent, random target
// among them, random mission template.  task = target cell code | template << 8.
template <class R>
LG_FN void lg_gen_fetch(const mgx_config &c, R &r, LgLevel &L)
{
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    const int first_obj = L.ncmd, n = c.level_arg0;
    L.ax = -1; L.ay = -1;
    for (int i = 0; i < n; i++) {
        const int type = lg_randint(r, 0, 2);                 // _rand_elem(['key', 'ball'])
        const int color = lg_sorted_color(lg_randint(r, 0, 7)); // _rand_elem(COLOR_NAMES)
        int x, y;
        lg_sample_free(r, L, L.W, L.H, false, &x, &y);        // place_obj(obj): anywhere empty
        if (!r.alive()) return;
        lg_set(L, x, y, (type == 0 ? MGX_K_KEY : MGX_K_BALL) | ((uint32_t)color << 4));
    }
    lg_sample_free(r, L, L.W, L.H, false, &L.ax, &L.ay);      // place_agent()
    L.adir = lg_randint(r, 0, 4);
    const int t = lg_randint(r, 0, n);                        // target = objs[_rand_int(0, len(objs))]
    const int tmpl = lg_randint(r, 0, 5);                     // mission template
    if (L.too_big || !r.alive()) return;
    L.task = (uint32_t)L.cmds[first_obj + t].code | ((uint32_t)tmpl << 8);
}

// GoToObjectEnv._gen_grid (envs/gotoobject.py:24-66): numObjs (level_arg0) distinct (type, color) objects out of
// key / ball / box x 7 colours anywhere in the room (a duplicate pair costs its two draws and is drawn again), random
// agent, random target among them.  task = tx | ty << 4 | (type - key) << 8 | color << 10 (the rule needs the position).
template <class R>
LG_FN void lg_gen_gotoobject(const mgx_config &c, R &r, LgLevel &L)
{
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    const int first_obj = L.ncmd, n = c.level_arg0;
    L.ax = -1; L.ay = -1;
    int have = 0;
    while (have < n) {
        const int type = lg_randint(r, 0, 3);                   // _rand_elem(['key', 'ball', 'box'])
        const int color = lg_sorted_color(lg_randint(r, 0, 7)); // _rand_elem(COLOR_NAMES)
        if (!r.alive()) return;
        const uint32_t code = (uint32_t)(MGX_K_KEY + type) | ((uint32_t)color << 4);
        bool dup = false;
        for (int k = 0; k < have; k++) dup = dup || L.cmds[first_obj + k].code == code;
        if (dup) continue;
        int x, y;
        lg_sample_free(r, L, L.W, L.H, false, &x, &y);          // place_obj(obj): anywhere empty
        if (!r.alive() || L.too_big) return;
        lg_set(L, x, y, code);
        have++;
    }
    lg_sample_free(r, L, L.W, L.H, false, &L.ax, &L.ay);        // place_agent()
    L.adir = lg_randint(r, 0, 4);
    const int t = lg_randint(r, 0, n);                          // objIdx = _rand_int(0, len(objs))
    if (L.too_big || !r.alive()) return;
    const LgCmd tc = L.cmds[first_obj + t];
    L.task = (uint32_t)tc.x0 | ((uint32_t)tc.y0 << 4) | ((uint32_t)((tc.code & 15u) - MGX_K_KEY) << 8) | ((uint32_t)((tc.code >> 4) & 7u) << 10);
}

// RedBlueDoorEnv._gen_grid (envs/redbluedoors.py:20-42): a size x size room in the middle of a 2*size x size grid, random
// agent inside it, a closed red door in its left wall and a closed blue one in its right wall.
// task = red door y | blue door y << 4.
template <class R>
LG_FN void lg_gen_redbluedoors(const mgx_config &, R &r, LgLevel &L)
{
    const int s = L.H, x0 = s / 2, x1 = s / 2 + s - 1;
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);            // wall_rect(0, 0, 2*size, size)
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, x0, 0, x0, L.H - 1, MGX_CODE_WALL_GREY);          // wall_rect(size//2, 0, size, size): its two vertical walls
    lg_rect(L, x1, 0, x1, L.H - 1, MGX_CODE_WALL_GREY);
    L.ax = -1; L.ay = -1;
    for (;;) { // place_agent(top=(size//2, 0), size=(size, size))
        const int x = lg_randint(r, x0, x0 + s < L.W ? x0 + s : L.W), y = lg_randint(r, 0, s < L.H ? s : L.H);
        if (!r.alive()) return;
        if (!lg_empty(L, x, y)) continue;
        L.ax = x; L.ay = y;
        break;
    }
    L.adir = lg_randint(r, 0, 4);
    const int ry = lg_randint(r, 1, s - 1), by = lg_randint(r, 1, s - 1);
    lg_set(L, x0, ry, MGX_K_DOOR_CLOSED | (0u << 4));            // Door("red")
    lg_set(L, x1, by, MGX_K_DOOR_CLOSED | (2u << 4));            // Door("blue")
    L.task = (uint32_t)ry | ((uint32_t)by << 4);
}

// MemoryEnv._gen_grid (envs/memory.py:29-86): start room on the left with a green key or ball, a hallway to a split with a
// green key and a green ball; the matching one marks the success cell.  Draw order: hallway_end (random_length only),
// agent x, start object, order of the two objects at the split.  task = x of the success/failure cells | upper << 4.
#define MGX_CODE_KEY_GREEN (MGX_K_KEY | (1u << 4))
#define MGX_CODE_BALL_GREEN (MGX_K_BALL | (1u << 4))
template <class R>
LG_FN void lg_gen_memory(const mgx_config &c, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H, mid = H / 2, up = mid - 2, lo = mid + 2;
    L.ncmd = 0;
    lg_rect(L, 0, 0, W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, H - 1, W - 1, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, W - 1, 0, W - 1, H - 1, MGX_CODE_WALL_GREY);
    const int he = c.level_arg0 ? lg_randint(r, 4, W - 2) : W - 3; // hallway_end
    lg_rect(L, 1, up, 4, up, MGX_CODE_WALL_GREY);                  // start room
    lg_rect(L, 1, lo, 4, lo, MGX_CODE_WALL_GREY);
    lg_set(L, 4, up + 1, MGX_CODE_WALL_GREY);
    lg_set(L, 4, lo - 1, MGX_CODE_WALL_GREY);
    if (he > 5) {                                                  // horizontal hallway: i in range(5, hallway_end)
        lg_rect(L, 5, up + 1, he - 1, up + 1, MGX_CODE_WALL_GREY);
        lg_rect(L, 5, lo - 1, he - 1, lo - 1, MGX_CODE_WALL_GREY);
    }
    lg_rect(L, he, 0, he, mid - 1, MGX_CODE_WALL_GREY);            // vertical hallway, open at mid
    lg_rect(L, he, mid + 1, he, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, he + 2, 0, he + 2, H - 1, MGX_CODE_WALL_GREY);
    L.ax = lg_randint(r, 1, he + 1); L.ay = mid; L.adir = 0;
    const int start_ball = lg_randint(r, 0, 2);                    // _rand_elem([Key, Ball])
    lg_set(L, 1, mid - 1, start_ball ? MGX_CODE_BALL_GREEN : MGX_CODE_KEY_GREEN);
    const int key_first = lg_randint(r, 0, 2);                     // _rand_elem([[Ball, Key], [Key, Ball]])
    lg_set(L, he + 1, mid - 2, key_first ? MGX_CODE_KEY_GREEN : MGX_CODE_BALL_GREEN);
    lg_set(L, he + 1, mid + 2, key_first ? MGX_CODE_BALL_GREEN : MGX_CODE_KEY_GREEN);
    const bool upper = (start_ball != 0) == (key_first == 0);      // start_room_obj == other_objs[0]
    L.task = (uint32_t)(he + 1) | ((uint32_t)upper << 4);
}

// RoomGrid(num_rows=1, num_cols=2, room_size=6) (roomgrid.py:118-166) with the _gen_grid of Unlock / UnlockPickup /
// BlockedUnlockPickup (envs/unlock.py:21-31, unlockpickup.py:21-33, blockedunlockpickup.py:22-37).  variant = level_arg0.
// RoomGrid parks the agent in the middle of the grid, (8, 3), while the objects are placed: place_obj rejects that cell
// and reject_next_to (roomgrid.py:3-12) every cell at Manhattan distance < 2 of it.
template <class R>
LG_FN bool lg_roomgrid_place(R &r, LgLevel &L, int tx, int ty, int *ox, int *oy)
{
    for (;;) { // place_obj(obj, room.top, room.size, reject_fn=reject_next_to)
        const int x = lg_randint(r, tx, tx + 6 < L.W ? tx + 6 : L.W), y = lg_randint(r, ty, ty + 6 < L.H ? ty + 6 : L.H);
        if (!r.alive()) return false;
        if (!lg_empty(L, x, y)) continue;
        const int dx = x > 8 ? x - 8 : 8 - x, dy = y > 3 ? y - 3 : 3 - y;
        if (dx + dy < 2) continue; // (covers pos == agent_pos as well)
        *ox = x; *oy = y;
        return true;
    }
}

template <class R>
LG_FN void lg_gen_unlock(const mgx_config &c, R &r, LgLevel &L)
{
    const int variant = c.level_arg0;
    L.ncmd = 0;
    for (int i = 0; i < 2; i++) { // wall_rect of room (i, 0): top (5 i, 0), size 6 x 6
        const int x0 = 5 * i;
        lg_rect(L, x0, 0, x0 + 5, 0, MGX_CODE_WALL_GREY);
        lg_rect(L, x0, 5, x0 + 5, 5, MGX_CODE_WALL_GREY);
        lg_rect(L, x0, 0, x0, 5, MGX_CODE_WALL_GREY);
        lg_rect(L, x0 + 5, 0, x0 + 5, 5, MGX_CODE_WALL_GREY);
    }
    const int door_y = lg_randint(r, 1, 5); // room(0,0).door_pos[0] = (x_m, _rand_int(y_l, y_m))
    int x, y, box_color = 0;
    if (variant >= 1) { // add_object(1, 0, kind="box"): colour, then a place in the right room
        box_color = lg_sorted_color(lg_randint(r, 0, 7));
        if (!lg_roomgrid_place(r, L, 5, 0, &x, &y)) return;
        lg_set(L, x, y, MGX_K_BOX | ((uint32_t)box_color << 4));
    }
    const int door_color = lg_sorted_color(lg_randint(r, 0, 7)); // add_door(0, 0, 0, locked=True): colour
    lg_set(L, 5, door_y, MGX_K_DOOR_LOCKED | ((uint32_t)door_color << 4));
    if (variant == 2) { // the ball that blocks the door
        const int ball_color = lg_sorted_color(lg_randint(r, 0, 7));
        lg_set(L, 4, door_y, MGX_K_BALL | ((uint32_t)ball_color << 4));
    }
    if (!lg_roomgrid_place(r, L, 0, 0, &x, &y)) return;          // add_object(0, 0, 'key', door.color)
    lg_set(L, x, y, MGX_K_KEY | ((uint32_t)door_color << 4));
    for (;;) { // RoomGrid.place_agent(0, 0): a free cell of the left room whose front cell is empty or a wall
        const int ax = lg_randint(r, 0, 6), ay = lg_randint(r, 0, 6);
        if (!r.alive()) return;
        if (!lg_empty(L, ax, ay)) continue;
        const int d = lg_randint(r, 0, 4);
        const int fx = ax + (d == 0) - (d == 2), fy = ay + (d == 1) - (d == 3);
        const uint32_t fc = lg_code_at(L, fx, fy);
        L.ax = ax; L.ay = ay; L.adir = d;
        if (fc == MGX_CODE_EMPTY || (fc & 15u) == MGX_K_WALL) break;
    }
    L.task = variant == 0 ? (uint32_t)door_y : ((uint32_t)MGX_K_BOX | ((uint32_t)box_color << 4)); // door row / the target's cell code
}

// KeyCorridor._gen_grid (envs/keycorridor.py:26-52) on RoomGrid(num_cols=3, num_rows=R, room_size=S) (roomgrid.py:118-166):
// door positions drawn room by room, the middle column opened into a hallway, a locked door into a random right-hand
// room with the ball behind it, the key in a random left-hand room, the agent in the middle, then connect_all
// (roomgrid.py:306-352): random (room, side) picks until every room is reachable.  level_arg0 = S, rows = (H-1)/(S-1).
// Room bookkeeping lives in L.ws: [r] y of the right door, [9+r] x of the down door, [18+r] bits 0-3 = sides already
// connected (right, down, left, up), bit 4 = locked; r = 3*j + i.
template <class R>
LG_FN void lg_gen_keycorridor(const mgx_config &c, R &r, LgLevel &L)
{
    const int S = c.level_arg0, T = S - 1, rows = (L.H - 1) / T, W = L.W, H = L.H;
    if (6 * L.max_rivers < 27) { L.too_big = true; return; } // needs 27 words of workspace
    int16_t *dr = L.ws, *dd = L.ws + 9, *fl = L.ws + 18;
    L.ncmd = 0;
    for (int k = 0; k <= 3; k++) lg_rect(L, k * T, 0, k * T, H - 1, MGX_CODE_WALL_GREY);
    for (int k = 0; k <= rows; k++) lg_rect(L, 0, k * T, W - 1, k * T, MGX_CODE_WALL_GREY);
    for (int j = 0; j < rows; j++)
        for (int i = 0; i < 3; i++) {
            const int q = 3 * j + i, xl = i * T + 1, yl = j * T + 1, xm = i * T + S - 1, ym = j * T + S - 1;
            dr[q] = 0; dd[q] = 0; fl[q] = 0;
            if (i < 2) dr[q] = (int16_t)lg_randint(r, yl, ym);
            if (j < rows - 1) dd[q] = (int16_t)lg_randint(r, xl, xm);
        }
    if (!r.alive()) return;
    const int ax0 = T + S / 2, ay0 = (rows / 2) * T + S / 2; // RoomGrid parks the agent here while the objects are placed
    // connect_all's "every room reachable from the agent's" (a breadth-first search over the doors before every random pick,
    // roomgrid.py:306-331) is "one connected component": kept up to date as doors appear -- comp: 4 bits per room, the number of its
    // component; n_comp components left.  (The search cost ~300 instructions per pick, 30 picks per level: 40 % of a level.)
    uint64_t comp = 0;
    const int n_rooms = 3 * rows;
    for (int q = 0; q < n_rooms && q < 16; q++) comp |= (uint64_t)q << (4 * q);
    int n_comp = n_rooms;
    auto mark = [&](int i, int j, int k) { // both sides of wall k of room (i, j) are now connected
        fl[3 * j + i] |= (int16_t)(1 << k);
        const int ni = i + (k == 0) - (k == 2), nj = j + (k == 1) - (k == 3);
        fl[3 * nj + ni] |= (int16_t)(1 << ((k + 2) & 3));
        const uint32_t ca = (uint32_t)(comp >> (4 * (3 * j + i))) & 15u, cb = (uint32_t)(comp >> (4 * (3 * nj + ni))) & 15u;
        if (ca != cb && n_rooms <= 16) {
            for (int q = 0; q < n_rooms; q++)
                if (((uint32_t)(comp >> (4 * q)) & 15u) == cb) comp = (comp & ~((uint64_t)15 << (4 * q))) | ((uint64_t)ca << (4 * q));
            n_comp--;
        }
    };
    auto door_xy = [&](int i, int j, int k, int *x, int *y) { // room.door_pos[k]
        if (k == 0) { *x = i * T + S - 1; *y = dr[3 * j + i]; }
        else if (k == 1) { *x = dd[3 * j + i]; *y = j * T + S - 1; }
        else if (k == 2) { *x = i * T; *y = dr[3 * j + i - 1]; }
        else { *x = dd[3 * (j - 1) + i]; *y = j * T; }
    };
    auto place = [&](int i, int j, int *ox, int *oy) -> bool { // place_in_room: place_obj(top, size, reject_next_to)
        for (;;) {
            const int x = lg_randint(r, i * T, i * T + S), y = lg_randint(r, j * T, j * T + S);
            if (!r.alive()) return false;
            if (!lg_empty(L, x, y)) continue;
            const int dx = x > ax0 ? x - ax0 : ax0 - x, dy = y > ay0 ? y - ay0 : ay0 - y;
            if (dx + dy < 2) continue;
            *ox = x; *oy = y;
            return true;
        }
    };
    for (int j = 1; j < rows; j++) { // remove_wall(1, j, 3)
        if (S > 2) lg_rect(L, T + 1, j * T, T + S - 2, j * T, MGX_CODE_EMPTY);
        mark(1, j, 3);
    }
    const int ridx = lg_randint(r, 0, rows);
    const int door_color = lg_sorted_color(lg_randint(r, 0, 7)); // add_door(2, ridx, 2, locked=True)
    int x, y;
    door_xy(2, ridx, 2, &x, &y);
    lg_set(L, x, y, MGX_K_DOOR_LOCKED | ((uint32_t)door_color << 4));
    mark(2, ridx, 2);
    fl[3 * ridx + 2] |= 16;
    const int ball_color = lg_sorted_color(lg_randint(r, 0, 7)); // add_object(2, ridx, kind="ball")
    if (!place(2, ridx, &x, &y)) return;
    lg_set(L, x, y, MGX_K_BALL | ((uint32_t)ball_color << 4));
    const int krow = lg_randint(r, 0, rows);                     // add_object(0, _rand_int(0, rows), 'key', door.color)
    if (!place(0, krow, &x, &y)) return;
    lg_set(L, x, y, MGX_K_KEY | ((uint32_t)door_color << 4));
    const int aj = rows / 2;
    for (;;) { // place_agent(1, rows // 2)
        const int ax = lg_randint(r, T, T + S), ay = lg_randint(r, aj * T, aj * T + S);
        if (!r.alive()) return;
        if (!lg_empty(L, ax, ay)) continue;
        const int d = lg_randint(r, 0, 4);
        const uint32_t fc = lg_code_at(L, ax + (d == 0) - (d == 2), ay + (d == 1) - (d == 3));
        L.ax = ax; L.ay = ay; L.adir = d;
        if (fc == MGX_CODE_EMPTY || (fc & 15u) == MGX_K_WALL) break;
    }
    for (int it = 0; it <= 5000; it++) { // connect_all
        if (n_rooms <= 16) { if (n_comp == 1) break; } // every room reachable from the agent's
        else { // (more rooms than the packed component numbers hold -- no registered id: the search itself)
            uint32_t reach = 1u << (3 * aj + 1), frontier = reach;
            while (frontier) {
                const int q = lg_ctz32(frontier);
                frontier &= frontier - 1;
                for (int k = 0; k < 4; k++)
                    if (fl[q] & (1 << k)) {
                        const int nq = 3 * (q / 3 + (k == 1) - (k == 3)) + q % 3 + (k == 0) - (k == 2);
                        if (!(reach & (1u << nq))) { reach |= 1u << nq; frontier |= 1u << nq; }
                    }
            }
            if (reach == (1u << n_rooms) - 1u) break;
        }
        const int i = lg_randint(r, 0, 3), j = lg_randint(r, 0, rows), k = lg_randint(r, 0, 4);
        if (!r.alive()) return;
        const int ni = i + (k == 0) - (k == 2), nj = j + (k == 1) - (k == 3);
        if (ni < 0 || ni > 2 || nj < 0 || nj >= rows) continue;    // not room.door_pos[k]
        if (fl[3 * j + i] & (1 << k)) continue;                      // room.doors[k]
        if ((fl[3 * j + i] | fl[3 * nj + ni]) & 16) continue;        // room.locked or room.neighbors[k].locked
        const int color = lg_sorted_color(lg_randint(r, 0, 7));
        door_xy(i, j, k, &x, &y);
        lg_set(L, x, y, MGX_K_DOOR_CLOSED | ((uint32_t)color << 4));
        mark(i, j, k);
        if (L.too_big) return;
    }
    L.task = (uint32_t)MGX_K_BALL | ((uint32_t)ball_color << 4);
}

// ObstructedMaze (envs/obstructedmaze.py) on RoomGrid(room_size=6) (roomgrid.py:118-166): 1 x 2 rooms (1Dl / 1Dlh / 1Dlhb:
// level_arg1 = 0) or 3 x 3 (2Dl / 2Dlh / 2Dlhb / 1Q / 2Q / Full: level_arg1 = num_quarters | 8 if the agent starts in room
// (2, 1) instead of (1, 1)).  level_arg0: bit 0 = keys hidden in boxes, bit 1 = doors blocked by a ball.  A blue ball is
// the target (task word = its cell code, MGX_TASK_PICKUPBOX); blocking balls are green, boxes grey (COLOR_NAMES[0..2]).
// Draw order: door positions room by room, door_colors = _rand_subset(COLOR_NAMES, 7), then per locked door the place of
// its key / box (place_in_room: reject_next_to the parked agent), the target's room and place, the agent.
// L.ws: [q] y of the right door, [9+q] x of the down door of room q = cols*j + i.
template <class R>
LG_FN void lg_gen_obstructedmaze(const mgx_config &c, R &r, LgLevel &L)
{
    const int S = 6, T = 5, W = L.W, H = L.H, cols = (W - 1) / T, rows = (H - 1) / T;
    if (6 * L.max_rivers < 18) { L.too_big = true; return; }
    const bool in_box = (c.level_arg0 & 1) != 0, blocked = (c.level_arg0 & 2) != 0;
    int16_t *dr = L.ws, *dd = L.ws + 9;
    L.ncmd = 0;
    for (int k = 0; k <= cols; k++) lg_rect(L, k * T, 0, k * T, H - 1, MGX_CODE_WALL_GREY);
    for (int k = 0; k <= rows; k++) lg_rect(L, 0, k * T, W - 1, k * T, MGX_CODE_WALL_GREY);
    for (int j = 0; j < rows; j++)
        for (int i = 0; i < cols; i++) {
            const int q = cols * j + i;
            dr[q] = 0; dd[q] = 0;
            if (i < cols - 1) dr[q] = (int16_t)lg_randint(r, j * T + 1, j * T + S - 1);
            if (j < rows - 1) dd[q] = (int16_t)lg_randint(r, i * T + 1, i * T + S - 1);
        }
    int colors[7], pool[7]; // door_colors: _rand_elem + remove, seven times (the last one draws nothing)
    for (int k = 0; k < 7; k++) pool[k] = k;
    for (int n = 7; n > 0; n--) {
        const int idx = lg_randint(r, 0, n);
        colors[7 - n] = lg_sorted_color(pool[idx]);
        for (int k = idx; k + 1 < n; k++) pool[k] = pool[k + 1];
    }
    if (!r.alive()) return;
    const int ax0 = (cols / 2) * T + S / 2, ay0 = (rows / 2) * T + S / 2; // RoomGrid parks the agent here while objects are placed
    auto door_xy = [&](int i, int j, int k, int *x, int *y) { // room.door_pos[k]
        if (k == 0) { *x = i * T + S - 1; *y = dr[cols * j + i]; }
        else if (k == 1) { *x = dd[cols * j + i]; *y = j * T + S - 1; }
        else if (k == 2) { *x = i * T; *y = dr[cols * j + i - 1]; }
        else { *x = dd[cols * (j - 1) + i]; *y = j * T; }
    };
    auto place = [&](int i, int j, int *ox, int *oy) -> bool { // place_in_room: place_obj(room.top, room.size, reject_next_to)
        for (;;) {
            const int x = lg_randint(r, i * T, i * T + S < W ? i * T + S : W), y = lg_randint(r, j * T, j * T + S < H ? j * T + S : H);
            if (!r.alive()) return false;
            if (!lg_empty(L, x, y)) continue;
            const int dx = x > ax0 ? x - ax0 : ax0 - x, dy = y > ay0 ? y - ay0 : ay0 - y;
            if (dx + dy < 2) continue;
            *ox = x; *oy = y;
            return true;
        }
    };
    auto add_door = [&](int i, int j, int k, int color, bool locked, bool hide, bool block) -> bool { // envs/obstructedmaze.py:52-74
        int x, y;
        door_xy(i, j, k, &x, &y);
        lg_set(L, x, y, (uint32_t)(locked ? MGX_K_DOOR_LOCKED : MGX_K_DOOR_CLOSED) | ((uint32_t)color << 4));
        if (block) lg_set(L, x - ((k == 0) - (k == 2)), y - ((k == 1) - (k == 3)), MGX_K_BALL | (1u << 4)); // grid.set: whatever lay there is gone
        if (locked) {
            int kx, ky;
            if (!place(i, j, &kx, &ky)) return false;
            const uint32_t key = MGX_K_KEY | ((uint32_t)color << 4);
            if (hide) lg_set_box(L, kx, ky, MGX_K_BOX | (5u << 4), key);
            else lg_set(L, kx, ky, key);
        }
        return true;
    };
    int ai, aj, bi, bj;
    if (c.level_arg1 == 0) { // ObstructedMaze_1Dlhb._gen_grid
        if (!add_door(0, 0, 0, colors[0], true, in_box, blocked)) return;
        bi = 1; bj = 0; ai = 0; aj = 0;
    } else {                 // ObstructedMaze_Full._gen_grid
        const int nq = c.level_arg1 & 7;
        const int side_i[4] = {2, 1, 0, 1}, side_j[4] = {1, 2, 1, 0}, corner_i[4] = {2, 2, 0, 0}, corner_j[4] = {0, 2, 2, 0};
        for (int q = 0; q < nq; q++) {
            if (!add_door(1, 1, q, colors[q], false, false, false)) return;
            for (int kk = -1; kk <= 1; kk += 2)
                if (!add_door(side_i[q], side_j[q], (q + kk + 4) & 3, colors[(q + kk + 7) % 7], true, in_box, blocked)) return;
        }
        const int br = lg_randint(r, 0, nq); // _rand_elem(corners)
        bi = corner_i[br]; bj = corner_j[br];
        ai = (c.level_arg1 & 8) ? 2 : 1; aj = 1;
    }
    int x, y;
    if (!place(bi, bj, &x, &y)) return; // add_object(ball_room, "ball", blue)
    lg_set(L, x, y, MGX_CODE_BALL_BLUE);
    for (;;) { // RoomGrid.place_agent(i, j): a free cell of the room whose front cell is empty or a wall
        const int ax = lg_randint(r, ai * T, ai * T + S < W ? ai * T + S : W), ay = lg_randint(r, aj * T, aj * T + S < H ? aj * T + S : H);
        if (!r.alive()) return;
        if (!lg_empty(L, ax, ay)) continue;
        const int d = lg_randint(r, 0, 4);
        const uint32_t fc = lg_code_at(L, ax + (d == 0) - (d == 2), ay + (d == 1) - (d == 3));
        L.ax = ax; L.ay = ay; L.adir = d;
        if (fc == MGX_CODE_EMPTY || (fc & 15u) == MGX_K_WALL) break;
    }
    L.task = MGX_CODE_BALL_BLUE;
}

// LockedRoom._gen_grid (envs/lockedroom.py:37-113): a hallway between two columns of three rooms; one random room is
// locked and holds the goal, the six doors get six distinct colours, the key of the locked door lies in another room.
// task = colour of the locked room | colour of the key room << 3 (the mission names both).
template <class R>
LG_FN void lg_gen_lockedroom(const mgx_config &, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H, lw = W / 2 - 2, rw = W / 2 + 2, h3 = H / 3;
    L.ncmd = 0;
    lg_rect(L, 0, 0, W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, H - 1, W - 1, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, W - 1, 0, W - 1, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, lw, 0, lw, H - 1, MGX_CODE_WALL_GREY);            // hallway walls
    lg_rect(L, rw, 0, rw, H - 1, MGX_CODE_WALL_GREY);
    for (int n = 1; n < 3; n++) {                                // room splitting walls (n = 0 is the outer wall)
        lg_rect(L, 0, n * h3, lw - 1, n * h3, MGX_CODE_WALL_GREY);
        lg_rect(L, rw, n * h3, W - 1, n * h3, MGX_CODE_WALL_GREY);
    }
    // rooms in the reference's list order: (left, right) for n = 0, 1, 2; top = (0 | rw, n*h3), size (lw+1, h3+1), door (lw | rw, n*h3 + 3)
    const int locked = lg_randint(r, 0, 6);
    auto rand_pos = [&](int room, int *x, int *y) { // Room.rand_pos: _rand_pos(topX+1, topX+sizeX-1, topY+1, topY+sizeY-1)
        const int tx = (room & 1) ? rw : 0, ty = (room >> 1) * h3;
        *x = lg_randint(r, tx + 1, tx + lw);
        *y = lg_randint(r, ty + 1, ty + h3);
    };
    int x, y;
    rand_pos(locked, &x, &y);
    lg_set(L, x, y, MGX_CODE_GOAL_GREEN);
    uint32_t left = 0x7Fu; // colours still unassigned, as a mask over the SORTED names (blue green grey purple red white yellow)
    int color_of[6];
    for (int room = 0; room < 6; room++) {
        int n_left = 0;
        for (int k = 0; k < 7; k++) n_left += (left >> k) & 1u;
        int pick = lg_randint(r, 0, n_left), k = 0;
        for (;; k++) if ((left >> k) & 1u) { if (pick == 0) break; pick--; }
        left &= ~(1u << k);
        color_of[room] = lg_sorted_color(k);
        lg_set(L, (room & 1) ? rw : lw, (room >> 1) * h3 + 3, (room == locked ? MGX_K_DOOR_LOCKED : MGX_K_DOOR_CLOSED) | ((uint32_t)color_of[room] << 4));
    }
    int key_room;
    do { key_room = lg_randint(r, 0, 6); } while (key_room == locked && r.alive());
    rand_pos(key_room, &x, &y);
    lg_set(L, x, y, MGX_K_KEY | ((uint32_t)color_of[locked] << 4));
    L.ax = -1; L.ay = -1;
    for (;;) { // place_agent(top=(lw, 0), size=(rw - lw, H))
        const int ax = lg_randint(r, lw, rw), ay = lg_randint(r, 0, H);
        if (!r.alive()) return;
        if (!lg_empty(L, ax, ay)) continue;
        L.ax = ax; L.ay = ay;
        break;
    }
    L.adir = lg_randint(r, 0, 4);
    L.task = (uint32_t)color_of[locked] | ((uint32_t)color_of[key_room] << 3);
}

// PlaygroundV0._gen_grid (envs/playground_v0.py:13-66): 3 x 3 rooms with one randomly placed, randomly coloured door in
// every inner wall segment, random agent, 12 random objects.
template <class R>
LG_FN void lg_gen_playground(const mgx_config &, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H, rw = W / 3, rh = H / 3;
    L.ncmd = 0;
    lg_rect(L, 0, 0, W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, H - 1, W - 1, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, W - 1, 0, W - 1, H - 1, MGX_CODE_WALL_GREY);
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) {
            const int xL = i * rw, yT = j * rh, xR = xL + rw, yB = yT + rh;
            if (i + 1 < 3) {
                lg_rect(L, xR, yT, xR, yT + rh - 1, MGX_CODE_WALL_GREY);       // vert_wall(xR, yT, roomH)
                const int y = lg_randint(r, yT + 1, yB - 1);
                const int color = lg_sorted_color(lg_randint(r, 0, 7));
                lg_set(L, xR, y, MGX_K_DOOR_CLOSED | ((uint32_t)color << 4));
            }
            if (j + 1 < 3) {
                lg_rect(L, xL, yB, xL + rw - 1, yB, MGX_CODE_WALL_GREY);       // horz_wall(xL, yB, roomW)
                const int x = lg_randint(r, xL + 1, xR - 1);
                const int color = lg_sorted_color(lg_randint(r, 0, 7));
                lg_set(L, x, yB, MGX_K_DOOR_CLOSED | ((uint32_t)color << 4));
            }
            if (!r.alive() || L.too_big) return;
        }
    L.ax = -1; L.ay = -1;
    lg_sample_free(r, L, W, H, false, &L.ax, &L.ay); // place_agent()
    L.adir = lg_randint(r, 0, 4);
    for (int k = 0; k < 12; k++) {
        const int type = lg_randint(r, 0, 3);
        const int color = lg_sorted_color(lg_randint(r, 0, 7));
        int x, y;
        lg_sample_free(r, L, W, H, true, &x, &y);    // place_obj(obj): not on the agent
        if (!r.alive() || L.too_big) return;
        lg_set(L, x, y, (uint32_t)(MGX_K_KEY + type) | ((uint32_t)color << 4));
    }
}

// PutNearEnv._gen_grid (envs/putnear.py:24-89): like GoToObject, but no object may be placed within one cell of another
// (reject_fn near_obj); a random object to move and a different random target.
template <class R>
LG_FN void lg_gen_putnear(const mgx_config &c, R &r, LgLevel &L)
{
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    const int first_obj = L.ncmd, n = c.level_arg0;
    L.ax = -1; L.ay = -1;
    int have = 0;
    while (have < n) {
        const int type = lg_randint(r, 0, 3);
        const int color = lg_sorted_color(lg_randint(r, 0, 7));
        if (!r.alive()) return;
        const uint32_t code = (uint32_t)(MGX_K_KEY + type) | ((uint32_t)color << 4);
        bool dup = false;
        for (int k = 0; k < have; k++) dup = dup || L.cmds[first_obj + k].code == code;
        if (dup) continue;
        int x, y;
        for (;;) { // place_obj(obj, reject_fn=near_obj)
            x = lg_randint(r, 0, L.W); y = lg_randint(r, 0, L.H);
            if (!r.alive()) return;
            if (!lg_empty(L, x, y)) continue;
            bool near = false;
            for (int k = 0; k < have; k++) {
                const int dx = x - (int)L.cmds[first_obj + k].x0, dy = y - (int)L.cmds[first_obj + k].y0;
                near = near || (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1);
            }
            if (!near) break;
        }
        if (L.too_big) return;
        lg_set(L, x, y, code);
        have++;
    }
    lg_sample_free(r, L, L.W, L.H, false, &L.ax, &L.ay);        // place_agent()
    L.adir = lg_randint(r, 0, 4);
    const int mi = lg_randint(r, 0, n);                         // the object to move
    int ti;
    do { ti = lg_randint(r, 0, n); } while (ti == mi && r.alive()); // the target
    if (L.too_big || !r.alive()) return;
    const LgCmd mc = L.cmds[first_obj + mi], tc = L.cmds[first_obj + ti];
    L.task = (uint32_t)((mc.code & 15u) - MGX_K_KEY) | (((uint32_t)(mc.code >> 4) & 7u) << 2) | ((uint32_t)tc.x0 << 5) | ((uint32_t)tc.y0 << 8) |
             ((uint32_t)((tc.code & 15u) - MGX_K_KEY) << 11) | (((uint32_t)(tc.code >> 4) & 7u) << 13);
}

// TwoGoalsEnv._gen_grid (envs/twogoals.py:31-50): a yellow goal bottom-right, a green one bottom-left, fixed or random agent.
template <class R>
LG_FN void lg_gen_twogoals(const mgx_config &c, R &r, LgLevel &L)
{
    L.ncmd = 0;
    lg_rect(L, 0, 0, L.W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, L.H - 1, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, L.H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, L.W - 1, 0, L.W - 1, L.H - 1, MGX_CODE_WALL_GREY);
    lg_set(L, L.W - 2, L.H - 2, MGX_K_GOAL | (4u << 4)); // Goal(color1 = 'yellow')
    lg_set(L, 1, L.H - 2, MGX_K_GOAL | (1u << 4));       // Goal(color2 = 'green')
    if (c.level_arg0 == 0) { L.ax = 1; L.ay = 1; L.adir = 0; }
    else {
        L.ax = -1; L.ay = -1;
        lg_sample_free(r, L, L.W, L.H, false, &L.ax, &L.ay);
        L.adir = lg_randint(r, 0, 4);
    }
    L.task = 0; // goal_count
}

// GoToDoorEnv._gen_grid (envs/gotodoor.py:23-69, as modified by the fork): four locked doors on the four walls in four
// distinct colours, redrawn until one of them is red (the target); random agent.
template <class R>
LG_FN void lg_gen_gotodoor(const mgx_config &, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H;
    L.ncmd = 0;
    lg_rect(L, 0, 0, W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, H - 1, W - 1, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, W - 1, 0, W - 1, H - 1, MGX_CODE_WALL_GREY);
    int dx[4], dy[4], col[4];
    dx[0] = lg_randint(r, 2, W - 2); dy[0] = 0;
    dx[1] = lg_randint(r, 2, W - 2); dy[1] = H - 1;
    dx[2] = 0; dy[2] = lg_randint(r, 2, H - 2);
    dx[3] = W - 1; dy[3] = lg_randint(r, 2, H - 2);
    bool have_red = false;
    while (!have_red) { // doorIdx is None: draw the four colours again
        int n = 0;
        while (n < 4) {
            const int color = lg_sorted_color(lg_randint(r, 0, 7));
            if (!r.alive()) return;
            bool dup = false;
            for (int k = 0; k < n; k++) dup = dup || col[k] == color;
            if (dup) continue;
            if (color == 0) have_red = true;
            col[n++] = color;
        }
    }
    for (int k = 0; k < 4; k++) lg_set(L, dx[k], dy[k], MGX_K_DOOR_LOCKED | ((uint32_t)col[k] << 4));
    L.ax = -1; L.ay = -1;
    lg_sample_free(r, L, W, H, false, &L.ax, &L.ay); // place_agent(size=(W, H))
    L.adir = lg_randint(r, 0, 4);
}

// FourRoomsEnv._gen_grid (envs/fourrooms.py:19-67): outer walls, a cross of inner walls with one random gap per arm,
// random agent pose, random goal.
template <class R>
LG_FN void lg_gen_fourrooms(const mgx_config &, R &r, LgLevel &L)
{
    const int W = L.W, H = L.H, rw = W / 2, rh = H / 2;
    L.ncmd = 0;
    lg_rect(L, 0, 0, W - 1, 0, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, H - 1, W - 1, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, 0, 0, 0, H - 1, MGX_CODE_WALL_GREY);
    lg_rect(L, W - 1, 0, W - 1, H - 1, MGX_CODE_WALL_GREY);
    for (int j = 0; j < 2; j++)
        for (int i = 0; i < 2; i++) {
            const int xL = i * rw, yT = j * rh, xR = xL + rw, yB = yT + rh;
            if (i + 1 < 2) {
                lg_rect(L, xR, yT, xR, yT + rh - 1, MGX_CODE_WALL_GREY);       // vert_wall(xR, yT, room_h)
                lg_set(L, xR, lg_randint(r, yT + 1, yB), MGX_CODE_EMPTY);
            }
            if (j + 1 < 2) {
                lg_rect(L, xL, yB, xL + rw - 1, yB, MGX_CODE_WALL_GREY);       // horz_wall(xL, yB, room_w)
                lg_set(L, lg_randint(r, xL + 1, xR), yB, MGX_CODE_EMPTY);
            }
        }
    L.ax = -1; L.ay = -1;
    lg_sample_free(r, L, W, H, false, &L.ax, &L.ay); // place_agent()
    L.adir = lg_randint(r, 0, 4);
    int gx, gy;
    lg_sample_free(r, L, W, H, true, &gx, &gy);      // place_obj(Goal())
    if (!r.alive()) return;
    lg_set(L, gx, gy, MGX_CODE_GOAL_GREEN);
}

// DynamicObstaclesEnv._gen_grid (envs/dynamicobstacles.py:35-58): room + goal, fixed or random agent, then n_obstacles
// blue balls by place_obj(max_tries=100; the reference would raise after 101 failed tries -- with at most 8 balls in a
// room that is a < 1e-12 event and is not restated).  Obstacle i is painted with a marker code (bit 7 | i << 4 | ball)
// so that the reset path can recover the ORDER of the obstacles, which step() walks (k_dynobs_init turns the markers
// into plain blue balls; the host generators do the same after painting).
#define MGX_CODE_OBSTACLE(i) (0x80u | ((uint32_t)(i) << 4) | MGX_K_BALL)
#define MGX_IS_OBSTACLE_MARK(code) (((code) & 0x8Fu) == (0x80u | MGX_K_BALL))
template <class R>
LG_FN void lg_gen_dynobs(const mgx_config &c, R &r, LgLevel &L)
{
    lg_room(L);
    if (c.level_arg1 == 0) { L.ax = 1; L.ay = 1; L.adir = 0; }
    else {
        L.ax = -1; L.ay = -1;
        lg_sample_free(r, L, L.W, L.H, false, &L.ax, &L.ay); // place_agent()
        L.adir = lg_randint(r, 0, 4);
    }
    for (int i = 0; i < c.level_arg0; i++) {
        int x, y;
        lg_sample_free(r, L, L.W, L.H, true, &x, &y);
        if (!r.alive()) return;
        lg_set(L, x, y, MGX_CODE_OBSTACLE(i));
    }
}

// true if the family draws random numbers (Empty with a fixed start does not)
LG_FN bool lg_uses_rng(const mgx_config &c)
{
    return !((c.level_kind == MGX_LEVEL_EMPTY || c.level_kind == MGX_LEVEL_TWOGOALS) && c.level_arg0 == 0) && c.level_kind != MGX_LEVEL_DISTSHIFT;
}

template <class R>
LG_FN void lg_generate(const mgx_config &c, R &r, LgLevel &L)
{
    switch (c.level_kind) {
    case MGX_LEVEL_EMPTY: lg_gen_empty(c, r, L); break;
    case MGX_LEVEL_DOORKEY: lg_gen_doorkey(c, r, L); break;
    case MGX_LEVEL_CROSSING: lg_gen_crossing(c, r, L); break;
    case MGX_LEVEL_DISTSHIFT: lg_gen_distshift(c, r, L); break;
    case MGX_LEVEL_MULTIROOM: lg_gen_multiroom(c, r, L); break;
    case MGX_LEVEL_FETCH: lg_gen_fetch(c, r, L); break;
    case MGX_LEVEL_GOTODOOR: lg_gen_gotodoor(c, r, L); break;
    case MGX_LEVEL_FOURROOMS: lg_gen_fourrooms(c, r, L); break;
    case MGX_LEVEL_DYNOBS: lg_gen_dynobs(c, r, L); break;
    case MGX_LEVEL_GOTOOBJECT: lg_gen_gotoobject(c, r, L); break;
    case MGX_LEVEL_REDBLUEDOORS: lg_gen_redbluedoors(c, r, L); break;
    case MGX_LEVEL_MEMORY: lg_gen_memory(c, r, L); break;
    case MGX_LEVEL_UNLOCK: lg_gen_unlock(c, r, L); break;
    case MGX_LEVEL_KEYCORRIDOR: lg_gen_keycorridor(c, r, L); break;
    case MGX_LEVEL_LOCKEDROOM: lg_gen_lockedroom(c, r, L); break;
    case MGX_LEVEL_PLAYGROUND: lg_gen_playground(c, r, L); break;
    case MGX_LEVEL_PUTNEAR: lg_gen_putnear(c, r, L); break;
    case MGX_LEVEL_TWOGOALS: lg_gen_twogoals(c, r, L); break;
    case MGX_LEVEL_OBSTRUCTEDMAZE: lg_gen_obstructedmaze(c, r, L); break;
    default: lg_gen_lavagap(c, r, L); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// env.seed(s): gym's legacy seeding (gym.utils.seeding.np_random, gym < 0.22) on top of numpy RandomState:
//   seed mod 2^64 -> SHA-512(str(seed)) -> first 8 digest bytes as two little-endian uint32 words (a zero high word
//   is dropped, 0 -> [0]) -> MT19937 init_by_array(words).  Only the first digest word h[0] is needed.
LG_FN uint64_t lg_rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

LG_FN uint64_t lg_sha512_h0(const uint8_t *msg, int len) // len < 112: one block
{
    static const uint64_t K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
        0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
        0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
        0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
        0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
        0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
        0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
        0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
        0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
        0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
        0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
        0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
        0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
        0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
        0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    uint64_t w[16]; // rolling message schedule
#pragma unroll
    for (int t = 0; t < 16; t++) {
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int b = t * 8 + i;
            const uint64_t byte = b < len ? msg[b] : (b == len ? 0x80u : 0u);
            v = (v << 8) | byte;
        }
        w[t] = v;
    }
    w[15] = (uint64_t)len * 8; // message length in bits (len < 112, so the length field is the last word alone)
    uint64_t a = 0x6a09e667f3bcc908ULL, b = 0xbb67ae8584caa73bULL, c = 0x3c6ef372fe94f82bULL, d = 0xa54ff53a5f1d36f1ULL,
             e = 0x510e527fade682d1ULL, f = 0x9b05688c2b3e6c1fULL, g = 0x1f83d9abfb41bd6bULL, h = 0x5be0cd19137e2179ULL;
    // 5 groups of 16 rounds, the group loop ROLLED and the round constants read from the table: fully unrolled (round 3) the 80 constants
    // became 160 literal-holding VGPRs that the compiler hoisted out of the seed kernels' env loop -- k_seed ran at 186 VGPRs, two waves per
    // SIMD -- and 5,000 instructions of straight-line code.  (The schedule indices stay compile-time: the round loop inside is unrolled.)
#pragma unroll 1
    for (int grp = 0; grp < 5; grp++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (grp > 0) {
                const uint64_t w15 = w[(r + 1) & 15], w2 = w[(r + 14) & 15];
                const uint64_t s0 = lg_rotr64(w15, 1) ^ lg_rotr64(w15, 8) ^ (w15 >> 7);
                const uint64_t s1 = lg_rotr64(w2, 19) ^ lg_rotr64(w2, 61) ^ (w2 >> 6);
                w[r] = w[r] + s0 + w[(r + 9) & 15] + s1;
            }
            const uint64_t wt = w[r];
            const uint64_t S1 = lg_rotr64(e, 14) ^ lg_rotr64(e, 18) ^ lg_rotr64(e, 41);
            const uint64_t ch = (e & f) ^ (~e & g);
            const uint64_t t1 = h + S1 + ch + K[16 * grp + r] + wt;
            const uint64_t S0 = lg_rotr64(a, 28) ^ lg_rotr64(a, 34) ^ lg_rotr64(a, 39);
            const uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
            const uint64_t t2 = S0 + mj;
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
    }
    return 0x6a09e667f3bcc908ULL + a;
}

LG_FN uint32_t lg_bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xFF00u) | ((x << 8) & 0xFF0000u) | (x << 24); }

// init_by_array key of `env.seed(seed)`; returns the key length (1 or 2)
LG_FN int lg_seed_key(uint64_t seed, uint32_t key[2])
{
    uint8_t txt[20];
    int n = 0;
    { // str(seed)
        uint8_t rev[20];
        uint64_t v = seed;
        do { rev[n++] = (uint8_t)('0' + (int)(v % 10)); v /= 10; } while (v);
        for (int i = 0; i < n; i++) txt[i] = rev[n - 1 - i];
    }
    const uint64_t h0 = lg_sha512_h0(txt, n); // digest bytes 0..7 = h0 big-endian
    key[0] = lg_bswap32((uint32_t)(h0 >> 32));
    key[1] = lg_bswap32((uint32_t)h0);
    return key[1] ? 2 : 1; // _int_list_from_bigint drops a zero high word; 0 -> [0]
}

// MT19937 tempering
LG_FN uint32_t lg_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

// one word of the next MT19937 block: new[k] from (old[k], old[k+1]) and m = old[k+397] (k<227) or new[k-227]
LG_FN uint32_t lg_twist_word(uint32_t a, uint32_t b, uint32_t m)
{
    const uint32_t y = (a & 0x80000000U) | (b & 0x7fffffffU);
    return m ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
}

#endif
