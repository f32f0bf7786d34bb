/*
 * minigrid_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's hot path, written to follow the
 * reference's *sequential* algorithm step by step (slice -> rotate_left x (dir+1)
 * -> process_vis two-sweep flood -> agent-cell overwrite -> encode), so that it is
 * an independent check of the closed-form / bit-mask formulation used by the HIP
 * kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.  Parity is PINNED: tests/test_oracle_golden.py replays
 * every fixture in tests/golden/ (recorded from the reference itself by
 * oracle/gen_golden.py) through these functions and demands byte equality.
 *
 * Reference (paths relative to /root/reference/gym_minigrid/):
 *   minigrid.py:27-61     OBJECT/COLOR/STATE codebooks
 *   minigrid.py:93-107    WorldObj.can_overlap / can_pickup / see_behind defaults
 *   minigrid.py:156-181   Goal (overlap = toggletimes<=0; toggle removes a default goal)
 *   minigrid.py:184-237   Floor / Lava / Wall predicates
 *   minigrid.py:239-275   Door predicates, toggle, encode
 *   minigrid.py:301-364   Key / Ball / Box (pickup; default Box toggle -> contents = None)
 *   minigrid.py:411-419   Grid.get/set  (row-major j*W+i, bounds asserts)
 *   minigrid.py:439-473   Grid.rotate_left / Grid.slice
 *   minigrid.py:571-594   Grid.encode(vis_mask)
 *   minigrid.py:617-648   Grid.process_vis (default_vis branch); :649-709 the fork's alternative model (default_vis=False)
 *   minigrid.py:933-937   _reward
 *   minigrid.py:1112-1133 front_pos / left_pos / right_pos
 *   minigrid.py:1162-1189 get_view_exts
 *   minigrid.py:1227-1325 MiniGridEnv.step
 *   minigrid.py:1327-1381 gen_obs_grid / gen_obs
 *   wrappers.py:311-338   FullyObsWrapper.observation
 *   envs/fetch.py:74-86, envs/gotodoor.py:71-93   task rules layered on MiniGridEnv.step
 *
 * State representation used here = the reference's own encoding:
 *   grid  u8[W][H][3]   Grid.encode() layout, index [x][y][channel]; (1,0,0) = None
 *   aux   u8[W][H]      bit0 = Goal.overlap (a goal built with toggletimes<=0); bits 3:1 = triage_color + 1 (0 = None);
 *                       bits 7:4 = (toggletimes - 1) & 15, so that 0 describes the default Goal() / Box(color)
 *   contains u8[W][H][3] (optional, mgo_set_contains) encode() of Box.contains, (1,0,0) = None
 *   agent i32[3]        x, y, dir
 *   carry u8[3]         encode() of the carried object, (1,0,0) = nothing; carry_aux u8
 *   steps i32           step_count
 */
#include <stdint.h>
#include <string.h>

#define VMAX 15 /* largest agent_view_size handled (minigrid.py:776 default 7; ViewSizeWrapper wrappers.py:579-608) */

enum { T_UNSEEN = 0, T_EMPTY = 1, T_WALL = 2, T_FLOOR = 3, T_DOOR = 4, T_KEY = 5,
       T_BALL = 6, T_BOX = 7, T_GOAL = 8, T_LAVA = 9, T_AGENT = 10 };
enum { ST_OPEN = 0, ST_CLOSED = 1, ST_LOCKED = 2 };
enum { A_LEFT = 0, A_RIGHT = 1, A_FORWARD = 2, A_PICKUP = 3, A_DROP = 4, A_TOGGLE = 5, A_DONE = 6,
       A_STRAFE_LEFT = 7, A_STRAFE_RIGHT = 8 /* ExtendedActions, minigrid.py:747-764 */ };

#define MGO_OK 0
#define MGO_ERR_ACTION (-1) /* reference: assert False, "unknown action"  (minigrid.py:1318) */
#define MGO_ERR_OOB (-2)    /* reference: Grid.get bounds assert         (minigrid.py:417-418) */
#define MGO_ERR_REFBUG (-3) /* reference: AttributeError in strafe_right (minigrid.py:1310 reads left_cell.overlap) */

typedef struct { uint8_t t, c, s, a; } cell_t; /* t == T_EMPTY  <=>  Python None */

typedef struct {
    int W, H, max_steps, see_through, lava_v1;
    int view; /* agent_view_size */
    int extended; /* extended_actions (minigrid.py:774,787-789) */
    int alt_vis;  /* default_vis=False (minigrid.py:777,786,1343): the fork's own visibility model */
    int task;     /* 0 none, 1 FetchEnv.step (envs/fetch.py:74-86), 2 GoToDoorEnv.step (envs/gotodoor.py:71-93) */
} mgo_cfg;

static const int DIR_TO_VEC[4][2] = { {1, 0}, {0, 1}, {-1, 0}, {0, -1} }; /* minigrid.py:64-73 */

static cell_t NONE(void) { cell_t c = { T_EMPTY, 0, 0, 0 }; return c; }

/* aux byte accessors (Goal/Box hidden state, minigrid.py:157-161,333-337) */
static int aux_tt(uint8_t a) { return ((a >> 4) + 1) & 15; }            /* toggletimes */
static int aux_tri(uint8_t a) { return ((a >> 1) & 7) - 1; }            /* triage_color index, -1 = None */
static uint8_t aux_with_tt(uint8_t a, int tt) { return (uint8_t)((a & 0x0F) | (((tt - 1) & 15) << 4)); }

/* Box.contains planes of the batch being stepped (test-harness state, like the task words) */
static uint8_t *g_contains = 0, *g_carry_contains = 0;
void mgo_set_contains(uint8_t *contains, uint8_t *carry_contains) { g_contains = contains; g_carry_contains = carry_contains; }
static int is_none(cell_t c) { return c.t == T_EMPTY; }

static cell_t grid_get(const mgo_cfg *cf, const uint8_t *g, const uint8_t *aux, int i, int j)
{
    cell_t c;
    const uint8_t *p = g + ((size_t)i * cf->H + j) * 3;
    c.t = p[0]; c.c = p[1]; c.s = p[2]; c.a = aux ? aux[(size_t)i * cf->H + j] : 0;
    return c;
}

static void grid_set(const mgo_cfg *cf, uint8_t *g, uint8_t *aux, int i, int j, cell_t c)
{
    uint8_t *p = g + ((size_t)i * cf->H + j) * 3;
    p[0] = c.t; p[1] = c.c; p[2] = c.s;
    if (aux) aux[(size_t)i * cf->H + j] = c.a;
}

/* can_overlap: None handled by callers.  Goal/Floor/Lava True, Door iff open,
 * Box iff color == triage_color (triage_color is None for every in-scope box -> False). */
static int can_overlap(cell_t c)
{
    switch (c.t) {
    case T_GOAL: case T_FLOOR: case T_LAVA: return 1;
    case T_DOOR: return c.s == ST_OPEN;
    case T_BOX: return aux_tri(c.a) >= 0 && aux_tri(c.a) == c.c; /* Box.can_overlap: color == triage_color (minigrid.py:342-343) */
    default: return 0;
    }
}
static int can_pickup(cell_t c) { return c.t == T_KEY || c.t == T_BALL || c.t == T_BOX; }
static int see_behind(cell_t c)
{
    if (c.t == T_WALL) return 0;
    if (c.t == T_DOOR) return c.s == ST_OPEN;
    return 1;
}

/* gen_obs_grid + gen_obs (minigrid.py:1327-1381): literal slice/rotate/process_vis/encode */
static void gen_obs(const mgo_cfg *cf, const uint8_t *g, const uint8_t *aux, const int32_t *agent,
                    const uint8_t *carry, uint8_t *image /* [view][view][3] */)
{
    int ax = agent[0], ay = agent[1], dir = agent[2];
    int topX, topY, i, j, r;
    const int V = cf->view;
    cell_t a[VMAX][VMAX], b[VMAX][VMAX]; /* [i][j] = Grid.get(i, j) of the view grid */
    int mask[VMAX][VMAX];

    /* get_view_exts (minigrid.py:1162-1189) */
    if (dir == 0)      { topX = ax;             topY = ay - V / 2; }
    else if (dir == 1) { topX = ax - V / 2;     topY = ay; }
    else if (dir == 2) { topX = ax - V + 1;     topY = ay - V / 2; }
    else               { topX = ax - V / 2;     topY = ay - V + 1; }

    /* Grid.slice (minigrid.py:453-473): out of bounds -> Wall() (grey) */
    for (j = 0; j < V; j++)
        for (i = 0; i < V; i++) {
            int x = topX + i, y = topY + j;
            if (x >= 0 && x < cf->W && y >= 0 && y < cf->H) a[i][j] = grid_get(cf, g, aux, x, y);
            else { cell_t w = { T_WALL, 5, 0, 0 }; a[i][j] = w; }
        }

    /* rotate_left (minigrid.py:439-451), dir+1 times (minigrid.py:1338-1339) */
    for (r = 0; r < dir + 1; r++) {
        for (i = 0; i < V; i++)
            for (j = 0; j < V; j++)
                b[j][V - 1 - i] = a[i][j];
        memcpy(a, b, sizeof a);
    }

    if (!cf->see_through && cf->alt_vis) {
        /* process_vis, "custom visualization" branch (minigrid.py:649-709), restated loop for loop */
#define OPQ(ii, jj) (!is_none(a[ii][jj]) && !see_behind(a[ii][jj]))
        const int px = V / 2, py = V - 1;
        memset(mask, 0, sizeof mask);
        mask[px][py] = 1;
        j = py;
        for (i = px + 1; i < V; i++) { mask[i][j] = 1; if (OPQ(i, j)) break; }
        for (i = px - 1; i >= 0; i--) { mask[i][j] = 1; if (OPQ(i, j)) break; }
        i = px;
        for (j = V - 2; j >= 0; j--) { mask[i][j] = 1; if (OPQ(i, j)) break; }
        for (i = px + 1; i < V; i++)
            for (j = V - 2; j >= 0; j--) {
                int c, ca, cb;
                if (!mask[i][j + 1] || !mask[i - 1][j]) continue;
                c = OPQ(i, j); ca = OPQ(i, j + 1); cb = OPQ(i - 1, j);
                if (!c && ca) break;
                if (!c && cb) break; /* hideside = True */
                mask[i][j] = 1;
            }
        for (i = px - 1; i >= 0; i--)
            for (j = V - 2; j >= 0; j--) {
                int c, ca, cb;
                if (!mask[i][j + 1] || !mask[i + 1][j]) continue;
                c = OPQ(i, j); ca = OPQ(i, j + 1); cb = OPQ(i + 1, j);
                if (!c && ca) break;
                if (!c && cb) break;
                mask[i][j] = 1;
            }
#undef OPQ
    } else
    /* process_vis, default_vis branch (minigrid.py:617-648) */
    if (!cf->see_through) {
        memset(mask, 0, sizeof mask);
        mask[V / 2][V - 1] = 1;
        for (j = V - 1; j >= 0; j--) {
            for (i = 0; i < V - 1; i++) {
                if (!mask[i][j]) continue;
                if (!is_none(a[i][j]) && !see_behind(a[i][j])) continue;
                mask[i + 1][j] = 1;
                if (j > 0) { mask[i + 1][j - 1] = 1; mask[i][j - 1] = 1; }
            }
            for (i = V - 1; i >= 1; i--) {
                if (!mask[i][j]) continue;
                if (!is_none(a[i][j]) && !see_behind(a[i][j])) continue;
                mask[i - 1][j] = 1;
                if (j > 0) { mask[i - 1][j - 1] = 1; mask[i][j - 1] = 1; }
            }
        }
    } else {
        for (i = 0; i < V; i++) for (j = 0; j < V; j++) mask[i][j] = 1;
    }

    /* agent sees what it carries, at its own view cell (minigrid.py:1349-1356) */
    {
        cell_t c = { carry[0], carry[1], carry[2], 0 };
        a[V / 2][V - 1] = c; /* (1,0,0) == None */
    }

    /* Grid.encode(vis_mask) (minigrid.py:571-594) */
    for (i = 0; i < V; i++)
        for (j = 0; j < V; j++) {
            uint8_t *p = image + (i * V + j) * 3;
            if (mask[i][j]) { p[0] = a[i][j].t; p[1] = a[i][j].c; p[2] = a[i][j].s; }
            else { p[0] = p[1] = p[2] = 0; }
        }
}

/* FullyObsWrapper.observation (wrappers.py:326-338) */
static void full_obs(const mgo_cfg *cf, const uint8_t *g, const int32_t *agent, uint8_t *image)
{
    memcpy(image, g, (size_t)cf->W * cf->H * 3);
    uint8_t *p = image + ((size_t)agent[0] * cf->H + agent[1]) * 3;
    p[0] = T_AGENT; p[1] = 0 /* red */; p[2] = (uint8_t)agent[2];
}

/* MiniGridEnv.step (minigrid.py:1227-1325) without the trailing gen_obs */
static int step_state(const mgo_cfg *cf, uint8_t *g, uint8_t *aux, int32_t *agent, uint8_t *carry,
                      uint8_t *carry_aux, int32_t *steps, int action, double *reward, uint8_t *done,
                      uint8_t *cont /* [W][H][3] or NULL */, uint8_t *carry_cont /* [3] or NULL */)
{
    int dir = agent[2];
    int fx, fy, lx, ly, rx, ry;
    cell_t fwd;
    *steps += 1;
    *reward = 0.0;
    *done = 0;

    fx = agent[0] + DIR_TO_VEC[dir][0];           fy = agent[1] + DIR_TO_VEC[dir][1];
    lx = agent[0] + DIR_TO_VEC[(dir + 3) % 4][0]; ly = agent[1] + DIR_TO_VEC[(dir + 3) % 4][1];
    rx = agent[0] + DIR_TO_VEC[(dir + 1) % 4][0]; ry = agent[1] + DIR_TO_VEC[(dir + 1) % 4][1];
    /* grid.get(fwd/left/right) all assert bounds (minigrid.py:1239-1243) */
    if (fx < 0 || fx >= cf->W || fy < 0 || fy >= cf->H || lx < 0 || lx >= cf->W || ly < 0 || ly >= cf->H ||
        rx < 0 || rx >= cf->W || ry < 0 || ry >= cf->H) {
        if (*steps >= cf->max_steps) *done = 1;
        return MGO_ERR_OOB;
    }
    if (action < 0 || action > (cf->extended ? A_STRAFE_RIGHT : A_DONE) ||
        (cf->task == 11 && (action == A_PICKUP || action == A_DROP))) { /* TwoGoalsEnv.step has no such branches */
        /* the reference asserts here */
        if (*steps >= cf->max_steps) *done = 1;
        return MGO_ERR_ACTION;
    }
    fwd = grid_get(cf, g, aux, fx, fy);

    if (action == A_LEFT) {
        agent[2] -= 1;
        if (agent[2] < 0) agent[2] += 4;
    } else if (action == A_RIGHT) {
        agent[2] = (agent[2] + 1) % 4;
    } else if (action == A_FORWARD) {
        if (is_none(fwd) || can_overlap(fwd)) { agent[0] = fx; agent[1] = fy; }
        if (!is_none(fwd) && fwd.t == T_GOAL && (fwd.a & 1)) {
            *done = 1;
            *reward = 1.0 * (1 - 0.9 * ((double)*steps / (double)cf->max_steps));
        }
        if (!is_none(fwd) && fwd.t == T_LAVA) {
            if (cf->lava_v1) { *done = 0; *reward = -1; }
            else *done = 1;
        }
    } else if (action == A_PICKUP) {
        if (!is_none(fwd) && can_pickup(fwd)) {
            if (carry[0] == T_EMPTY) {
                carry[0] = fwd.t; carry[1] = fwd.c; carry[2] = fwd.s; *carry_aux = fwd.a;
                grid_set(cf, g, aux, fx, fy, NONE());
                if (cont) {
                    uint8_t *q = cont + ((size_t)fx * cf->H + fy) * 3;
                    memcpy(carry_cont, q, 3);
                    q[0] = T_EMPTY; q[1] = 0; q[2] = 0;
                }
            }
        }
    } else if (action == A_DROP) {
        if (is_none(fwd) && carry[0] != T_EMPTY) {
            cell_t c = { carry[0], carry[1], carry[2], *carry_aux };
            grid_set(cf, g, aux, fx, fy, c);
            if (cont) {
                uint8_t *q = cont + ((size_t)fx * cf->H + fy) * 3;
                memcpy(q, carry_cont, 3);
                carry_cont[0] = T_EMPTY; carry_cont[1] = 0; carry_cont[2] = 0;
            }
            carry[0] = T_EMPTY; carry[1] = 0; carry[2] = 0; *carry_aux = 0;
        }
    } else if (action == A_TOGGLE) {
        if (!is_none(fwd)) {
            if (fwd.t == T_DOOR) { /* Door.toggle minigrid.py:252-262 */
                if (fwd.s == ST_LOCKED) {
                    if (carry[0] == T_KEY && carry[1] == fwd.c) { fwd.s = ST_OPEN; grid_set(cf, g, aux, fx, fy, fwd); }
                } else {
                    fwd.s = (fwd.s == ST_OPEN) ? ST_CLOSED : ST_OPEN;
                    grid_set(cf, g, aux, fx, fy, fwd);
                }
            } else if (fwd.t == T_GOAL) { /* Goal.toggle minigrid.py:171-181 */
                int tt = aux_tt(fwd.a);
                if (tt > 0) {
                    tt -= 1;
                    fwd.a = aux_with_tt(fwd.a, tt);
                    if (tt <= 0 && aux_tri(fwd.a) < 0) grid_set(cf, g, aux, fx, fy, NONE());
                    else {
                        if (tt <= 0) fwd.c = (uint8_t)aux_tri(fwd.a); /* self.color = self.triage_color */
                        grid_set(cf, g, aux, fx, fy, fwd);
                    }
                }
            } else if (fwd.t == T_BOX) { /* Box.toggle minigrid.py:355-364 */
                int tt = aux_tt(fwd.a) - 1;
                if (tt < 0) tt = 0; /* Python keeps counting down; every value <= 0 behaves the same */
                fwd.a = aux_with_tt(fwd.a, tt);
                if (tt <= 0 && aux_tri(fwd.a) < 0) { /* env.grid.set(*pos, self.contains) */
                    cell_t c = NONE();
                    if (cont) {
                        uint8_t *q = cont + ((size_t)fx * cf->H + fy) * 3;
                        c.t = q[0]; c.c = q[1]; c.s = q[2];
                        q[0] = T_EMPTY; q[1] = 0; q[2] = 0;
                    }
                    grid_set(cf, g, aux, fx, fy, c);
                } else {
                    if (tt <= 0) fwd.c = (uint8_t)aux_tri(fwd.a);
                    grid_set(cf, g, aux, fx, fy, fwd);
                }
            }
        }
    } else if (action == A_STRAFE_LEFT || action == A_STRAFE_RIGHT) { /* minigrid.py:1295-1314 */
        const int tx = action == A_STRAFE_LEFT ? lx : rx, ty = action == A_STRAFE_LEFT ? ly : ry;
        cell_t tc = grid_get(cf, g, aux, tx, ty), lc = grid_get(cf, g, aux, lx, ly);
        int rc = MGO_OK;
        if (is_none(tc) || can_overlap(tc)) { agent[0] = tx; agent[1] = ty; }
        if (!is_none(tc) && tc.t == T_GOAL) {
            int ov;
            if (action == A_STRAFE_LEFT) ov = tc.a & 1;
            else if (!is_none(lc) && lc.t == T_GOAL) ov = lc.a & 1;  /* the reference reads LEFT_cell.overlap here */
            else { ov = 0; rc = MGO_ERR_REFBUG; }                   /* ... which raises AttributeError otherwise */
            if (ov) { *done = 1; *reward = 1 - 0.9 * ((double)*steps / (double)cf->max_steps); }
        }
        if (!is_none(tc) && tc.t == T_LAVA) *done = 1;              /* no 'v1' special case on the strafe path */
        if (*steps >= cf->max_steps) *done = 1;
        return rc;
    } /* A_DONE: pass */

    if (*steps >= cf->max_steps) *done = 1;
    return MGO_OK;
}

/* step() overrides that only reshape reward/done after MiniGridEnv.step.  task word: Fetch = target (type | color<<4). */
/* RedBlueDoors: is the door at (x, y) open?  (Door.encode: state 0 = open, minigrid.py:264-275) */
static int door_open(const mgo_cfg *cf, const uint8_t *g, int x, int y)
{
    const uint8_t *t = g + ((size_t)x * cf->H + y) * 3;
    return t[0] == T_DOOR && t[2] == ST_OPEN;
}

static void task_rule(const mgo_cfg *cf, const uint8_t *g, const uint8_t *aux, const int32_t *agent, const uint8_t *carry,
                      int32_t steps, uint32_t task, int action, double *reward, uint8_t *done, int pre)
{
    if (cf->task == 1) {
        if (carry[0] != T_EMPTY) { /* if self.carrying: */
            *done = 1;
            if (carry[0] == (task & 15u) && carry[1] == ((task >> 4) & 7u)) *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps);
            else *reward = 0;
        }
    } else if (cf->task == 2) {
        if (action == A_DONE) {
            static const int D4[4][2] = { {1, 0}, {-1, 0}, {0, 1}, {0, -1} };
            int k;
            for (k = 0; k < 4; k++) {
                int x = agent[0] + D4[k][0], y = agent[1] + D4[k][1];
                cell_t c;
                if (x < 0 || x >= cf->W || y < 0 || y >= cf->H) continue;
                c = grid_get(cf, g, aux, x, y);
                if (c.t != T_DOOR) continue;
                *done = 1;                                                   /* adjacent to any of the doors */
                if (c.c == 0) *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps); /* the target door is the red one */
            }
        }
    } else if (cf->task == 10) { /* PutNearEnv.step envs/putnear.py:91-110; pre = something was carried before the step */
        const uint32_t mt = T_KEY + (task & 3u), mc = (task >> 2) & 7u;
        static const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
        if (action == A_PICKUP && carry[0] != T_EMPTY && (carry[0] != mt || carry[1] != mc)) *done = 1;
        if (action == A_DROP && pre) {
            if (carry[0] == T_EMPTY) { /* grid.get(ox, oy) is preCarrying: the drop happened */
                int dx = agent[0] + DX[agent[2]] - (int)((task >> 5) & 7u), dy = agent[1] + DY[agent[2]] - (int)((task >> 8) & 7u);
                if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps);
            }
            *done = 1;
        }
    } else if (cf->task == 7) { /* Unlock.step envs/unlock.py:33-41; the door is at (5, task) */
        if (action == A_TOGGLE && door_open(cf, g, 5, (int)(task & 15u))) { *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps); *done = 1; }
    } else if (cf->task == 8) { /* UnlockPickup.step envs/unlockpickup.py:35-43, KeyCorridor.step keycorridor.py:51-59: carrying the target */
        if (action == A_PICKUP && carry[0] == (task & 15u) && carry[1] == ((task >> 4) & 7u)) { *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps); *done = 1; }
    } else if (cf->task == 6) { /* MemoryEnv.step envs/memory.py:92-99; task = x of the two end cells | (success is the upper one) << 4 */
        const int tx = (int)(task & 15u), up = (int)((task >> 4) & 1u);
        const int sy = up ? cf->H / 2 - 1 : cf->H / 2 + 1, fy = up ? cf->H / 2 + 1 : cf->H / 2 - 1;
        if (agent[0] == tx && agent[1] == sy) { *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps); *done = 1; }
        if (agent[0] == tx && agent[1] == fy) { *reward = 0; *done = 1; }
    } else if (cf->task == 5) { /* RedBlueDoorEnv.step envs/redbluedoors.py:44-66; pre = red | blue << 1 before the step */
        const int rx = cf->H / 2, bx = cf->H / 2 + cf->H - 1;
        const int red_after = door_open(cf, g, rx, (int)(task & 15u)), blue_after = door_open(cf, g, bx, (int)((task >> 4) & 15u));
        if (blue_after) { *reward = (pre & 1) ? 1 - 0.9 * ((double)steps / (double)cf->max_steps) : 0; *done = 1; }
        else if (red_after && (pre & 2)) { *reward = 0; *done = 1; }
    } else if (cf->task == 4) { /* GoToObjectEnv.step envs/gotoobject.py:68-84; task = tx | ty << 4 | ... */
        if (action == A_TOGGLE) *done = 1;
        if (action == A_DONE) {
            int dx = agent[0] - (int)(task & 15u), dy = agent[1] - (int)((task >> 4) & 15u);
            if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) *reward = 1 - 0.9 * ((double)steps / (double)cf->max_steps);
            *done = 1;
        }
    }
}

/* ------------------------------------------------------------------ exported, batched */

/* One reference `env.step(a)` per env, in place.  obs may be NULL.  full may be NULL.
 * Returns 0, or the first per-env error code (processing continues for the rest);
 * err (optional, i32[n]) receives the per-env code. */
static uint32_t *g_task = 0; /* per-env task words for the next mgo_step_batch / mgo_rollout call (test harness state;
                               TwoGoals keeps its running goal count there) */
void mgo_set_task(uint32_t *task) { g_task = task; }

int mgo_step_batch(const mgo_cfg *cf, int64_t n, uint8_t *grid, uint8_t *aux, int32_t *agent,
                   uint8_t *carry, uint8_t *carry_aux, int32_t *steps, const uint8_t *actions,
                   uint8_t *obs, uint8_t *full, double *reward, uint8_t *done, int32_t *err)
{
    const size_t cells = (size_t)cf->W * cf->H;
    int first = 0;
    for (int64_t e = 0; e < n; e++) {
        uint8_t *g = grid + e * cells * 3, *ax = aux + e * cells;
        int pre = 0;
        if (cf->task == 10) pre = carry[e * 3] != T_EMPTY;
        if (cf->task == 11) { /* TwoGoals: the front cell before the step, type | color << 4 (0 = outside the grid) */
            const int32_t *ag = agent + e * 3;
            const int fx = ag[0] + DIR_TO_VEC[ag[2]][0], fy = ag[1] + DIR_TO_VEC[ag[2]][1];
            if (fx >= 0 && fx < cf->W && fy >= 0 && fy < cf->H) { const cell_t c = grid_get(cf, g, ax, fx, fy); pre = c.t | (c.c << 4); }
        }
        if (cf->task == 5 && g_task)
            pre = door_open(cf, g, cf->H / 2, (int)(g_task[e] & 15u)) | (door_open(cf, g, cf->H / 2 + cf->H - 1, (int)((g_task[e] >> 4) & 15u)) << 1);
        const int act_e = (cf->task == 6 && actions[e] == A_PICKUP) ? A_TOGGLE : actions[e]; /* envs/memory.py:89-90 */
        int rc = step_state(cf, g, ax, agent + e * 3, carry + e * 3, carry_aux + e, steps + e,
                            act_e, reward + e, done + e,
                            g_contains ? g_contains + e * cells * 3 : 0, g_carry_contains ? g_carry_contains + e * 3 : 0);
        if (cf->task == 11 && rc == MGO_OK && g_task) { /* TwoGoalsEnv.step envs/twogoals.py:118-146 on top of the base transition */
            if (act_e == A_TOGGLE) {
                if ((pre & 15) == T_EMPTY) rc = MGO_ERR_REFBUG; /* fwd_cell.type on None: AttributeError */
                else if ((pre & 15) == T_GOAL) { reward[e] = (pre >> 4) == 1 ? 0.25 : ((pre >> 4) == 4 ? 0.5 : 0.0); g_task[e] += 1; }
            }
            if (act_e == A_DONE) done[e] = 1;
            if (g_task[e] >= 2) { reward[e] += 1. - 0.9 * steps[e] / cf->max_steps; done[e] = 1; }
        } else if (cf->task && rc == MGO_OK)
            task_rule(cf, g, ax, agent + e * 3, carry + e * 3, steps[e], g_task ? g_task[e] : 0u, act_e, reward + e, done + e, pre);
        if (err) err[e] = rc;
        if (rc && !first) first = rc;
        if (obs) gen_obs(cf, g, ax, agent + e * 3, carry + e * 3, obs + e * (cf->view * cf->view * 3));
        if (full) full_obs(cf, g, agent + e * 3, full + e * cells * 3);
    }
    return first;
}

/* reset()-time observation of the current state (minigrid.py:857) */
void mgo_obs_batch(const mgo_cfg *cf, int64_t n, const uint8_t *grid, const uint8_t *aux,
                   const int32_t *agent, const uint8_t *carry, uint8_t *obs, uint8_t *full)
{
    const size_t cells = (size_t)cf->W * cf->H;
    for (int64_t e = 0; e < n; e++) {
        if (obs) gen_obs(cf, grid + e * cells * 3, aux + e * cells, agent + e * 3, carry + e * 3, obs + e * (cf->view * cf->view * 3));
        if (full) full_obs(cf, grid + e * cells * 3, agent + e * 3, full + e * cells * 3);
    }
}

/* CPU-baseline helper for bench.py: T steps over n envs with "restore the episode's
 * initial state on done" (= ReseedWrapper(seeds=[s]) + caller-side reset, wrappers.py:24-28,
 * run_tests.py:64-66).  actions u8[T][n].  Returns the number of env-steps executed. */
int64_t mgo_rollout(const mgo_cfg *cf, int64_t n, int64_t T, uint8_t *grid, uint8_t *aux, int32_t *agent,
                    uint8_t *carry, uint8_t *carry_aux, int32_t *steps, const uint8_t *grid0,
                    const uint8_t *aux0, const int32_t *agent0, const uint8_t *actions, uint8_t *obs,
                    uint8_t *full, double *reward, uint8_t *done)
{
    const size_t cells = (size_t)cf->W * cf->H;
    for (int64_t t = 0; t < T; t++) {
        mgo_step_batch(cf, n, grid, aux, agent, carry, carry_aux, steps, actions + t * n, obs, full, reward, done, 0);
        for (int64_t e = 0; e < n; e++) {
            if (!done[e]) continue;
            memcpy(grid + e * cells * 3, grid0 + e * cells * 3, cells * 3);
            memcpy(aux + e * cells, aux0 + e * cells, cells);
            memcpy(agent + e * 3, agent0 + e * 3, 3 * sizeof(int32_t));
            carry[e * 3] = T_EMPTY; carry[e * 3 + 1] = 0; carry[e * 3 + 2] = 0; carry_aux[e] = 0;
            steps[e] = 0;
            if (obs) gen_obs(cf, grid + e * cells * 3, aux + e * cells, agent + e * 3, carry + e * 3, obs + e * (cf->view * cf->view * 3));
            if (full) full_obs(cf, grid + e * cells * 3, agent + e * 3, full + e * cells * 3);
        }
    }
    return n * T;
}
