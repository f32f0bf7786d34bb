/*
 * mgx.h -- C ABI of libmgx.so: batched MiniGrid `step -> gen_obs -> encode` on MI355X (gfx950).
 *
 * The reference (rohitrango/gym-minigrid) has no FFI / plugin layer: its only
 * interface for this path is the legacy gym.Env Python API.  Each entry point below
 * therefore cites the reference *method* it replaces (paths relative to
 * /root/reference/gym_minigrid/), batched over N independent env instances:
 *
 *   mgx_create / mgx_destroy   MiniGridEnv.__init__ kwargs        minigrid.py:767-829
 *   mgx_env_config             per-id constructor arguments       envs/empty.py:10-28, envs/doorkey.py:9-13,
 *                                                                 envs/crossing.py:12-22, envs/lavagap.py:10-19
 *   mgx_reset                  seed(s); reset()  |  reset()       minigrid.py:831-863 (caller loop: run_tests.py:64-66), wrappers.py:24-28
 *   mgx_set_seed_schedule      ReseedWrapper(env, seeds, seed_idx) wrappers.py:12-28
 *   mgx_set_state/get_state    env.grid / agent_pos / agent_dir / carrying / step_count attributes
 *                                                                 minigrid.py:816-823,851-854, Grid.encode :571-594
 *   mgx_observe                gen_obs()                          minigrid.py:1359-1381
 *   mgx_step                   step(action)                       minigrid.py:1227-1325
 *                              (+ FullyObsWrapper.observation     wrappers.py:326-338 when obs_mode = MGX_OBS_FULL)
 *   mgx_rollout                the caller's `for t: env.step(a[t])` loop as one hipGraph launch     run_tests.py:41-68
 *   mgx_generate_levels        _gen_grid of the built-in families envs/empty.py:30-57, envs/doorkey.py:15-44,
 *     (_ex, _level_stream(_ex))                                   envs/crossing.py:24-92, envs/lavagap.py:21-59,
 *                                                                 envs/distshift.py:30-52, envs/multiroom.py:40-219,
 *                                                                 envs/fourrooms.py, fetch.py, gotodoor.py, gotoobject.py,
 *                                                                 putnear.py, redbluedoors.py, memory.py, unlock*.py,
 *                                                                 keycorridor.py (+ roomgrid.py), lockedroom.py,
 *                                                                 playground_v0.py, dynamicobstacles.py, twogoals.py,
 *                                                                 obstructedmaze.py
 *   mgx_set/get_task           per-episode attributes of the task envs (targetType, target_pos, ...) as one word
 *   mgx_set/get_object_state   Goal/Box.toggletimes, triage_color, Box.contains              minigrid.py:156-181,332-364
 *   mgx_get_direction          obs['direction']                   minigrid.py:1375-1379
 *   mgx_get_pose               env.agent_pos, env.agent_dir ('pos', 'dir' of AgentExtraInfoWrapper)   minigrid.py:816-818, wrappers.py:169-187
 *   mgx_mission                obs['mission'] / env.mission       minigrid.py:1373-1379, the envs' `self.mission = ...`
 *   obs_mode one-hot / flat    OneHotPartialObsWrapper, FullyObsOneHotWrapper, FlatObsWrapper   wrappers.py:203-243,340-415,528-577
 *   task_kind                  the `step` overrides of the task envs (mgx_task_kind below cites each)
 *
 * Conventions
 *   - every function returns 0 (MGX_OK) or a negative mgx_status; mgx_last_error()
 *     returns a thread-local message for the last failure.  No C++ exception crosses.
 *   - buffers are CALLER-OWNED.  Every data pointer may be a device pointer on the
 *     handle's GPU (zero-copy; obs must be 16-byte aligned, int32/float arrays 4-byte) or a host pointer (staged
 *     through an internal buffer).  The library detects which.
 *   - calls on one handle are NOT thread-safe; work is enqueued asynchronously on the
 *     handle's HIP stream (mgx_set_stream adopts a caller stream, e.g. torch's).
 *     Host-pointer outputs are complete when the call returns; device-pointer outputs
 *     are complete after mgx_sync() or any later work on the same stream.
 *   - one handle per GPU; envs shard across GPUs by contiguous index blocks with no
 *     data-path collective.
 *
 * Encodings (identical to the reference's):
 *   grid   uint8 [N][W][H][3]  Grid.encode(): index [x][y][channel] = (type, color, state); (1,0,0) = empty
 *   aux    uint8 [N][W][H]     hidden Goal/Box state: bit0 = Goal.overlap (a goal built with toggletimes<=0: terminal,
 *                              minigrid.py:156-162,1259-1261); bits 3:1 = triage_color + 1 (0 = None); bits 7:4 =
 *                              (toggletimes - 1) & 15.  0 = the default Goal() / Box(color); 0xF1 = Goal(toggletimes=0).
 *                              Anything else needs a handle created with object_state = 1.  NULL = all zeros.
 *   agent  int32 [N][3]        x, y, dir (0 right, 1 down, 2 left, 3 up; minigrid.py:64-73)
 *   carry  uint8 [N][3]        encode() of the carried object, (1,0,0) = nothing
 *   steps  int32 [N]           step_count
 *   obs    uint8 [N][V][V][3]  obs['image'] (MGX_OBS_PARTIAL, V = agent_view_size, default 7) or uint8 [N][W][H][3] (MGX_OBS_FULL)
 *   reward float [N]           the reference's Python double narrowed to f32
 *   done   uint8 [N]           0/1
 */
#ifndef MGX_H
#define MGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_VIEW 7              /* agent_view_size (minigrid.py:776) */
#define MGX_OBS_PARTIAL_BYTES (MGX_VIEW * MGX_VIEW * 3)
#define MGX_NUM_ACTIONS 7       /* MiniGridEnv.Actions (minigrid.py:731-745) */

typedef enum {
    MGX_OK = 0,
    MGX_ERR_INVALID_ARG = -1,
    MGX_ERR_INVALID_STATE = -2,  /* set_state: a cell / agent / carry value the reference cannot produce; reset / step: a call the handle's
                                    state does not allow yet (reset() without seeds before any seeded reset; step before the reset that starts
                                    a seed schedule) */
    MGX_ERR_INVALID_ACTION = -3, /* an action >= 7 was stepped (reference: AssertionError, minigrid.py:1318) */
    MGX_ERR_OUT_OF_BOUNDS = -4,  /* agent neighbour outside the grid (reference: Grid.get assert, :417-418), or strafe_right
                                    onto a goal while the left cell is not a goal (reference: AttributeError, :1310) */
    MGX_ERR_UNSUPPORTED = -5,    /* grid too large for one wavefront's LDS tile, unknown env id, ... */
    MGX_ERR_HIP = -6,            /* HIP runtime failure (message has hipGetErrorString) */
    MGX_ERR_NO_LEVELGEN = -7     /* mgx_reset on a handle whose family has no built-in generator */
} mgx_status;

typedef enum {
    MGX_OBS_PARTIAL = 0,          /* uint8 [N][V][V][3]   obs['image']                                                  */
    MGX_OBS_FULL = 1,             /* uint8 [N][W][H][3]   FullyObsWrapper (wrappers.py:311-338)                          */
    MGX_OBS_PARTIAL_ONEHOT = 2,   /* uint8 [N][V][V][21]  OneHotPartialObsWrapper formula (wrappers.py:203-243)          */
    MGX_OBS_FULL_ONEHOT = 3,      /* uint8 [N][W][H][22]  FullyObsOneHotWrapper(flatten=False) (wrappers.py:340-415)     */
    MGX_OBS_FULL_ONEHOT_NOCOLOR = 4, /* uint8 [N][W][H][15]  ... with drop_color=True                                   */
    MGX_OBS_PARTIAL_FLAT = 5,     /* float [N][V*V*3 + 27*96]  FlatObsWrapper (wrappers.py:528-577): image bytes as f32, then
                                     the mission string one-hot (96 chars x 27 codes); the `obs` arguments then point at
                                     floats (np.concatenate of a uint8 and a float32 array is float32)                    */
    MGX_OBS_FULL_FLAT = 6         /* float [N][W*H*3 + 27*96]  FlatObsWrapper(FullyObsWrapper(env))                        */
} mgx_obs_mode;

/* level families with a built-in seeded generator */
typedef enum {
    MGX_LEVEL_NONE = 0,     /* state is injected with mgx_set_state only */
    MGX_LEVEL_EMPTY = 1,    /* EmptyEnv: level_arg0 = 1 -> random agent start (Empty-Random-*), arg1 = sizetop (0 = none) */
    MGX_LEVEL_DOORKEY = 2,  /* DoorKeyEnv */
    MGX_LEVEL_CROSSING = 3, /* CrossingEnv: level_arg0 = num_crossings, level_arg1 = obstacle type (9 lava, 2 wall)
                               + 16 * (0 both river directions, 1 horizontal only (ori=0), 2 vertical only (ori=1)) */
    MGX_LEVEL_LAVAGAP = 4,  /* LavaGapEnv:  level_arg0 = const gap column (0/1), level_arg1 = obstacle type */
    MGX_LEVEL_DISTSHIFT = 5, /* DistShiftEnv (envs/distshift.py): level_arg0 = strip2_row; no randomness */
    MGX_LEVEL_MULTIROOM = 6, /* MultiRoomEnv (envs/multiroom.py): level_arg0 = minNumRooms | maxNumRooms << 8 (<= 8),
                                level_arg1 = maxRoomSize */
    MGX_LEVEL_FETCH = 7,     /* FetchEnv (envs/fetch.py): level_arg0 = numObjs; use with task_kind = MGX_TASK_FETCH */
    MGX_LEVEL_GOTODOOR = 8,  /* GoToDoorEnv (envs/gotodoor.py); use with task_kind = MGX_TASK_GOTODOOR */
    MGX_LEVEL_FOURROOMS = 9, /* FourRoomsEnv (envs/fourrooms.py:8-70), random agent and goal */
    MGX_LEVEL_DYNOBS = 10,   /* DynamicObstaclesEnv (envs/dynamicobstacles.py): level_arg0 = n_obstacles after the
                                constructor's clamp (<= 8), level_arg1 = 1 for a random agent start; use with
                                task_kind = MGX_TASK_DYNOBS */
    MGX_LEVEL_GOTOOBJECT = 11, /* GoToObjectEnv (envs/gotoobject.py): level_arg0 = numObjs; use with task_kind = MGX_TASK_GOTOOBJECT */
    MGX_LEVEL_REDBLUEDOORS = 12, /* RedBlueDoorEnv (envs/redbluedoors.py): width = 2*height; use with task_kind = MGX_TASK_REDBLUEDOORS */
    MGX_LEVEL_MEMORY = 13,   /* MemoryEnv (envs/memory.py): odd square grids 7..17, level_arg0 = random_length; use with
                                task_kind = MGX_TASK_MEMORY */
    MGX_LEVEL_UNLOCK = 14,   /* Unlock / UnlockPickup / BlockedUnlockPickup (envs/unlock.py, unlockpickup.py, blockedunlockpickup.py:
                                RoomGrid 1x2, room_size 6 -> 11x6): level_arg0 = 0 / 1 / 2; use with MGX_TASK_UNLOCK (0) or
                                MGX_TASK_PICKUPBOX (1, 2) */
    MGX_LEVEL_KEYCORRIDOR = 15, /* KeyCorridor (envs/keycorridor.py) on RoomGrid 3 x R: level_arg0 = room_size (3..6), rows from the
                                height; use with task_kind = MGX_TASK_PICKUPBOX */
    MGX_LEVEL_LOCKEDROOM = 16, /* LockedRoom (envs/lockedroom.py), 19x19; use with task_kind = MGX_TASK_NOTE (the task word only
                                names the mission: locked colour | key room colour << 3) */
    MGX_LEVEL_PLAYGROUND = 17, /* PlaygroundV0 (envs/playground_v0.py), 19x19: nine rooms, random doors, 12 random objects, no mission */
    MGX_LEVEL_PUTNEAR = 18,  /* PutNearEnv (envs/putnear.py): level_arg0 = numObjs, grids up to 8x8; use with MGX_TASK_PUTNEAR */
    MGX_LEVEL_TWOGOALS = 19, /* TwoGoalsEnv (envs/twogoals.py): level_arg0 = 1 for a random agent start; use with MGX_TASK_TWOGOALS */
    MGX_LEVEL_OBSTRUCTEDMAZE = 20, /* ObstructedMaze (envs/obstructedmaze.py) on RoomGrid(room_size 6): level_arg0 bit 0 = keys hidden
                                in boxes (Box.contains: needs object_state = 1), bit 1 = doors blocked by a ball; level_arg1 = 0
                                for the 1 x 2 mazes (11x6: 1Dl / 1Dlh / 1Dlhb), else num_quarters (1, 2, 4) | 8 if the agent starts
                                in room (2, 1) (16x16: 2Dl / 2Dlh / 2Dlhb; 1Q / 2Q / Full start in (1, 1)); use with
                                task_kind = MGX_TASK_PICKUPBOX (target: the blue ball) */
    MGX_LEVEL_KIND_END = 21
} mgx_level_kind;

/* task rules layered on MiniGridEnv.step by env subclasses (`step` overrides that only reshape reward/done) */
typedef enum {
    MGX_TASK_NONE = 0,
    MGX_TASK_FETCH = 1,    /* envs/fetch.py:74-86: once something is carried the episode ends; reward = _reward() iff it is
                              the target object.  Per-env task word = target cell code | mission template << 8. */
    MGX_TASK_GOTODOOR = 2, /* envs/gotodoor.py:71-93: the `done` action next to any door ends the episode, next to the
                              target (red) door it also pays _reward(). */
    MGX_TASK_DYNOBS = 3,   /* envs/dynamicobstacles.py:60-89 (with level_kind = MGX_LEVEL_DYNOBS): actions >= 3 fold to 0;
                              before the base step every obstacle (blue ball) is re-placed in its 3x3 neighbourhood with
                              draws from the env's own RNG stream (place_obj, max_tries=100); moving forward while the
                              front cell was occupied by anything but the goal gives reward -1 and done.  State enters
                              through mgx_reset only (the RNG stream is part of it): mgx_set_state is refused. */
    MGX_TASK_GOTOOBJECT = 4, /* envs/gotoobject.py:68-84: `toggle` ends the episode; `done` ends it and pays _reward() when the agent
                              is within one cell (Chebyshev) of the target's INITIAL position.  Per-env task word =
                              tx | ty << 4 | (type - key) << 8 | color << 10. */
    MGX_TASK_REDBLUEDOORS = 5, /* envs/redbluedoors.py:44-66: the episode ends when the blue door is open after a step (reward
                              _reward() iff the red one was open before it) or when the red one is open after a step that
                              closed an open blue one.  Per-env task word = red door y | blue door y << 4. */
    MGX_TASK_MEMORY = 6,   /* envs/memory.py:88-101: `pickup` acts as `toggle`; standing on the success / failure cell at the
                              end of the hallway ends the episode (reward _reward() / 0).  Per-env task word = x of those
                              cells | (success is the upper one) << 4. */
    MGX_TASK_UNLOCK = 7,   /* envs/unlock.py:33-41: `toggle` with the door open afterwards ends the episode with _reward().
                              Per-env task word = door y (the door is at x = 5). */
    MGX_TASK_PICKUPBOX = 8, /* envs/unlockpickup.py:35-43, blockedunlockpickup.py:39-47, keycorridor.py:51-59, obstructedmaze.py:42-50: `pickup` while
                              carrying the target object (`self.carrying == self.obj`: the only box / ball of the level)
                              ends the episode with _reward().  Per-env task word = the target's cell code
                              (type | color << 4). */
    MGX_TASK_NOTE = 9,     /* no rule on top of MiniGridEnv.step; the per-env task word only selects the mission string */
    MGX_TASK_PUTNEAR = 10, /* envs/putnear.py:91-110: picking up anything but the object to move ends the episode; a `drop` while
                              carrying ends it too, with _reward() when the object landed within one cell of the target's
                              initial position.  Per-env task word = move type | move colour << 2 | tx << 5 | ty << 8 |
                              target type << 11 | target colour << 13 (types: 0 key, 1 ball, 2 box). */
    MGX_TASK_TWOGOALS = 11 /* envs/twogoals.py:82-146, the fork's own step: `toggle` on a goal removes it and pays 0.25 (green) /
                              0.5 (yellow); after the second one the episode ends with + 1 - 0.9*steps/max_steps; `done` ends
                              it; `pickup` / `drop` hit the reference's `assert False` (counted as invalid actions) and
                              `toggle` on an empty cell its AttributeError (counted with the out-of-bounds faults).  The
                              per-env task word is the running goal count. */
} mgx_task_kind;

typedef struct {
    int32_t width, height;      /* grid size, >= 3 */
    int32_t max_steps;
    int32_t see_through_walls;  /* 1 = skip process_vis (minigrid.py:1344-1347) */
    int32_t lava_v1;            /* 1 = class name contains 'v1': lava gives reward -1, no done (minigrid.py:1262-1268) */
    int32_t obs_mode;           /* mgx_obs_mode */
    int32_t auto_reset;         /* 0 = reference semantics (caller resets, run_tests.py:64-66: mgx_reset in one of its three forms).
                                   1 = on done the env is restored to its episode-start state (the state last given
                                       by mgx_reset / mgx_set_state: ReseedWrapper(seeds=[s]) semantics; with
                                       mgx_set_seed_schedule: the level of the NEXT seed of the env's list) inside the same
                                       step, and obs is the first observation of the new episode (VecEnv convention);
                                       reward/done still describe the terminal transition. */
    int32_t level_kind;         /* mgx_level_kind */
    int32_t level_arg0, level_arg1;
    int32_t new_level_each_episode; /* with auto_reset: 0 = every episode replays the level of mgx_reset/mgx_set_state
                                   (ReseedWrapper(seeds=[s]), wrappers.py:12-32); 1 = plain reference behaviour: the
                                   env's own RNG stream (seeded by mgx_reset) continues and every reset() draws a NEW
                                   level (minigrid.py:836-839), generated on the GPU from a per-env MT19937 state kept
                                   in HBM.  Needs a level_kind with a generator and W*H <= 4096.  Device memory per env: the
                                   MT19937 block, 2.5 KB; the draw-heavy families (everything but Empty / DoorKey / Crossing /
                                   LavaGap / DistShift) keep the following block ready as well, 5 KB in all -- dropped silently
                                   (same results, slower level bursts) when that second allocation fails.  Partial-view
                                   handles (every family but MultiRoom, FourRooms, LockedRoom, Playground, GoToObject and DoorKey / Empty
                                   on grids from 13x13 up) keep 16
                                   next levels per env ready -- 16 * (ceil4(W*H)
                                   + 8) bytes, three times the cells with object_state; one level when that allocation
                                   fails -- and generate beside the steps, on a stream of the handle's own that every entry
                                   point joins before it touches levels or RNG state from the host side; results do not
                                   depend on it. */
    int32_t agent_view_size;    /* 0 = 7 (minigrid.py:776).  ViewSizeWrapper (wrappers.py:579-608): 3, 5, 7, 9 or 11;
                                   obs is then uint8 [N][V][V][3].  Ignored by MGX_OBS_FULL. */
    int32_t extended_actions;   /* 1 = ExtendedActions (minigrid.py:747-764): 7 = strafe_left, 8 = strafe_right */
    int32_t alt_visibility;     /* 1 = default_vis=False: the fork's alternative occlusion model (minigrid.py:649-709) */
    int32_t task_kind;          /* mgx_task_kind; needs max_steps <= 65535 (the step counter shares a word with the task) */
    int32_t object_state;       /* 1 = keep the hidden Goal/Box state per cell (toggletimes, triage_color, Box.contains;
                                   minigrid.py:156-181,332-364) in extra planes; 0 = every Goal/Box is the default one
                                   (toggletimes 1, or 0 for terminal goals; no triage colour; empty boxes) */
} mgx_config;

typedef struct mgx_env_s *mgx_handle;

const char *mgx_last_error(void);
const char *mgx_version(void);

/* Fill *cfg for a registered reference id ("MiniGrid-Empty-8x8-v0", ...).  obs_mode and
 * auto_reset are left 0.  Unknown id -> MGX_ERR_UNSUPPORTED. */
int mgx_env_config(const char *env_id, mgx_config *cfg);
/* i-th supported env id, or NULL past the end. */
const char *mgx_env_id(int i);
/* env.mission (the `mission` entry of gen_obs, minigrid.py:1373-1379) of a level of this family; `task` is the per-env
 * task word (mgx_get_task; only Fetch missions depend on it: envs/fetch.py:60-71).  Writes a NUL-terminated string,
 * returns its length, or a negative status (MGX_ERR_INVALID_ARG: cap too small / bad task word). */
int mgx_mission(const mgx_config *cfg, uint32_t task, char *out, int cap);

int mgx_create(const mgx_config *cfg, int64_t n_envs, int device, mgx_handle *out);
int mgx_destroy(mgx_handle h);
int mgx_set_stream(mgx_handle h, void *hip_stream); /* adopt a caller stream (hipStream_t); NULL = the null stream */
int mgx_use_own_stream(mgx_handle h);               /* back to the handle's own non-blocking stream (the default) */
int mgx_sync(mgx_handle h); /* waits for the stream; returns MGX_ERR_INVALID_ACTION / MGX_ERR_OUT_OF_BOUNDS
                               if a fault was recorded since the last mgx_clear_faults */
int mgx_clear_faults(mgx_handle h);
int mgx_obs_bytes(mgx_handle h, int64_t *per_env);

/* Host-side level generation (pure CPU, no handle, no GPU): seeds -> initial states in the
 * reference's encoding.  Reproduces `env.seed(s); env.reset()` of the family. */
int mgx_generate_levels(const mgx_config *cfg, int64_t n, const uint64_t *seeds,
                        uint8_t *grid, int32_t *agent);

/* Same, also returning the per-env task word (uint32 [n], may be NULL) of the families that have one (Fetch). */
int mgx_generate_levels_ex(const mgx_config *cfg, int64_t n, const uint64_t *seeds, uint8_t *grid, int32_t *agent, uint32_t *task);
/* Same, also returning Box.contains of every cell (uint8 [n][W][H][3], encode() of the contents, (1,0,0) = nothing; may be
 * NULL): the keys ObstructedMaze hides in boxes.  The format of mgx_set_object_state's `contains`. */
int mgx_generate_levels_full(const mgx_config *cfg, int64_t n, const uint64_t *seeds, uint8_t *grid, int32_t *agent, uint32_t *task,
                             uint8_t *contains);

/* Plain reference behaviour without ReseedWrapper (pure CPU): `env.seed(seed)` once, then K consecutive
 * `env.reset()`s -- the env's RNG stream continues, every episode gets a new level (minigrid.py:836-839).
 * grid uint8 [K][W][H][3], agent int32 [K][3]. */
int mgx_generate_level_stream(const mgx_config *cfg, uint64_t seed, int64_t K, uint8_t *grid, int32_t *agent);
/* Same, also returning the per-level task words (uint32 [K], may be NULL). */
int mgx_generate_level_stream_ex(const mgx_config *cfg, uint64_t seed, int64_t K, uint8_t *grid, int32_t *agent, uint32_t *task);
/* Same, also returning Box.contains of every cell of every level (uint8 [K][W][H][3], may be NULL). */
int mgx_generate_level_stream_full(const mgx_config *cfg, uint64_t seed, int64_t K, uint8_t *grid, int32_t *agent, uint32_t *task,
                                   uint8_t *contains);

/* The reference's reset() of every env with mask[i] != 0 (mask NULL = all), in its three forms:
 *
 *   seeds != NULL             env.seed(seeds[i]); env.reset()                     minigrid.py:831-863 (ReseedWrapper(seeds=[s]), wrappers.py:24-28)
 *   seeds == NULL, no schedule  env.reset()  -- the plain caller-side reset of `if done: env.reset()` (run_tests.py:64-66): the env's own
 *                             RNG stream CONTINUES from where the last reset (and, for Dynamic-Obstacles, the obstacle walks since)
 *                             left it, and the next level is drawn from it (minigrid.py:836-839).  The env must have been seeded by an
 *                             earlier mgx_reset(seeds) (MiniGridEnv.__init__ does `self.seed(1337); self.reset()`, minigrid.py:824-829:
 *                             a binding that mirrors gym.make() issues that first call itself), else MGX_ERR_INVALID_STATE.
 *   seeds == NULL, schedule   ReseedWrapper(seeds=[s0..sK-1]).reset(): env i is re-seeded with the next entry of ITS seed list
 *                             (mgx_set_seed_schedule below), cyclically                wrappers.py:19-28
 *
 * For the families that draw random numbers the whole reset runs on the GPU (k_seed: SHA-512 key + MT19937
 * init_by_array per env, then k_levelgen); an env whose seed is the one it already has is restored from its
 * episode-start snapshot instead (same result: the level and the RNG state are functions of the seed).
 * Families whose level does not depend on the seed (Empty with a fixed start, DistShift, fixed TwoGoals) generate it
 * once on the host at the first full reset; every later reset, of any form, is a restore on the device.  Grids beyond 64x64 cells
 * with a random family are generated per env on the host (seeds may still be a device pointer; the seeds == NULL forms are
 * MGX_ERR_UNSUPPORTED there: their RNG stream lives on the host only for the length of a call).
 * With auto_reset = 1 the state a reset leaves behind is also the new episode-start snapshot (plain form: the level just drawn).
 * obs (optional) receives the current observation of all envs; with a mask and a DEVICE obs buffer only the 64-env
 * tiles that contain a reset env are rewritten (pass the buffer the last mgx_step wrote, as the reference's
 * `if done: obs = env.reset()` loop does, and the other entries are already right). */
int mgx_reset(mgx_handle h, const uint64_t *seeds, const uint8_t *mask, uint8_t *obs);

/* ReseedWrapper(env_i, seeds = seeds[i][0 .. K-1], seed_idx = idx0) for every env (wrappers.py:12-28): from now on every reset of
 * env i -- the in-kernel one of auto_reset = 1 handles and mgx_reset(h, NULL, mask, obs) -- is `env.seed(seeds[i][j]); env.reset()`
 * with j = idx0, idx0 + 1, ... mod K, counted per env.  seeds: uint64 [N][K] (host or device pointer), 1 <= K <= 255, 0 <= idx0 < K.
 * The K levels of every env are functions of its K seeds: they are generated here, once, on the GPU (K x (k_seed + k_levelgen)) into
 * K episode-start snapshots per env that stay resident in HBM (K x (W*H + 8) bytes per env; Dynamic-Obstacles: + 2.7 KB of RNG
 * state each), so a reset at run time -- in the step kernel or in mgx_reset -- is a copy, whatever K is.
 * The wrapper's constructor does not reset, and neither does this call: the live episodes are INVALID afterwards (their RNG state was
 * used to generate the snapshots) and mgx_step / mgx_observe return MGX_ERR_INVALID_STATE until mgx_reset(h, NULL, NULL, obs) has
 * started every env on seeds[i][idx0].  K = 0 (seeds may be NULL) removes the schedule; so do mgx_reset with seeds != NULL and
 * mgx_set_state, which define the episode start themselves.  Not for new_level_each_episode handles (the two settings are the two
 * alternatives of the reference: with or without the wrapper) and not for grids beyond 64x64 cells: MGX_ERR_UNSUPPORTED. */
int mgx_set_seed_schedule(mgx_handle h, const uint64_t *seeds, int32_t K, int32_t idx0);

/* env = ActionBonus(env) / env = StateBonus(env) (wrappers.py:87-153): every step adds 1 / math.sqrt(count) to the reward, count =
 * the visits of this env to (agent_pos, agent_dir, action) / to agent_pos -- the state AFTER the step (before an auto-reset), this
 * step included -- kept across episodes (the wrappers' reset() leaves self.counts alone).  The sum is formed in doubles like the
 * reference's and rounded to the float the caller gets once.  Calls stack in call order, innermost first, at most one of each kind
 * (StateBonus(ActionBonus(env)) = MGX_BONUS_ACTION, then MGX_BONUS_STATE); kind 0 removes both and frees the counts.  Every call
 * zeroes the counts.  Device memory: uint32 [N][W][H][4][A] (A = 7 actions, 9 with extended_actions) for the action bonus,
 * uint32 [N][W][H] for the state bonus.  Steps that hit the reference's `unknown action` assertion count nothing.
 * mgx_rollout of such a handle is the captured graph of per-step launches.  The step kernel is then a `k_step_wrap` / `k_step_dyn_wrap`
 * / wrap-`k_step_fulldirect` instance (mgx_step_kernel_name): the only kernels that carry this code. */
typedef enum { MGX_BONUS_ACTION = 1, MGX_BONUS_STATE = 2 } mgx_bonus_kind;
int mgx_add_bonus(mgx_handle h, int32_t kind);
/* The counts of one of the wrappers (its self.counts as a dense array, layout above), device or host memory. */
int mgx_get_bonus_counts(mgx_handle h, int32_t kind, uint32_t *counts);

/* env = DACWrapper(env) (wrappers.py:35-84, the fork's absorbing-state wrapper): an env that is done before step max_steps of its
 * episode is not reset; from that step on its observation is the wrapper's `last_obs` -- the image with every byte 1 --, its reward 0
 * (the terminal step still reports the env's reward) and its actions are ignored, until the step that is the max_steps-th since the
 * reset reports done = 1 (and, with auto_reset, starts the next episode in the same step).  Every episode is max_steps long.
 * on = 0 removes the wrapper.  uint8 image observations only (MGX_OBS_PARTIAL / MGX_OBS_FULL); not for Dynamic-Obstacles handles.
 * mgx_get_state of an absorbed env: the env's state after its last step, with `steps` = the wrapper's count; mgx_get_direction: the
 * direction after that step (the reference's last_obs carries the direction of the episode's first observation).  With a bonus
 * wrapper as well the handle is ActionBonus / StateBonus(DACWrapper(env)): absorbed steps count visits of the frozen pose. */
int mgx_set_dac(mgx_handle h, int32_t on);

/* Inject / read back the full simulator state.  set_state also records the state as the
 * episode start used by auto_reset.  aux, carry, steps may be NULL (zeros / nothing / 0).
 * get_state: any pointer may be NULL.
 * new_level_each_episode handles (after a seeded reset of every env): the injected state replaces the CURRENT episode only, like
 * assignments to env.grid / env.agent_pos in the reference, which draw nothing: the next episode is the level the env's RNG
 * stream gives next, as if set_state had not been called.  The same holds for mgx_set_task and mgx_set_object_state there. */
int mgx_set_state(mgx_handle h, const uint8_t *grid, const uint8_t *aux, const int32_t *agent,
                  const uint8_t *carry, const int32_t *steps);
int mgx_get_state(mgx_handle h, uint8_t *grid, uint8_t *aux, int32_t *agent, uint8_t *carry, int32_t *steps);

/* Per-env task word (uint32 [N], 16 bits used; handles with task_kind != MGX_TASK_NONE): Fetch target etc.
 * set_task also records it for the episode-start snapshot. */
int mgx_set_task(mgx_handle h, const uint32_t *task);
int mgx_get_task(mgx_handle h, uint32_t *task);

/* object_state handles: Box.contains of every cell (uint8 [N][W][H][3], encode() of the contents, (1,0,0) = None) and the
 * hidden state of the carried object (carry_aux uint8 [N] in the aux format above, carry_contains uint8 [N][3]).
 * Any pointer may be NULL (set: leave unchanged; get: skip).  mgx_set_state resets all of them to "empty / default";
 * call this after it.  set also records the planes for the episode-start snapshot. */
int mgx_set_object_state(mgx_handle h, const uint8_t *contains, const uint8_t *carry_aux, const uint8_t *carry_contains);
int mgx_get_object_state(mgx_handle h, uint8_t *contains, uint8_t *carry_aux, uint8_t *carry_contains);

/* gen_obs() of the current state, no transition. */
int mgx_observe(mgx_handle h, uint8_t *obs);
/* T consecutive mgx_step calls in one host call (the reference caller's `for t: env.step(a[t])` loop, run_tests.py:41-68):
 * actions uint8 [T][N], obs [T][N][obs_bytes] (or NULL), reward float [T][N] (or NULL), done uint8 [T][N] (or NULL), all
 * DEVICE memory.  Handles with the default visibility, no object_state, plain uint8 observations and no new_level_each_episode /
 * Dynamic-Obstacles run all T steps in ONE kernel launch with the env state resident on chip (k_rollout): partial views of every
 * size on grids up to 16x16, the FullyObs observation on grids up to 13x13; every other handle captures its per-step launches into
 * a hipGraph on first use and replays it while T and the buffers stay the same.  Same results as T mgx_step calls either way.  Asynchronous on the handle's stream like mgx_step. */
int mgx_rollout(mgx_handle h, int64_t T, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done);

/* obs['direction'] (minigrid.py:1375-1379): agent_dir of every env, uint8 [N]. */
int mgx_get_direction(mgx_handle h, uint8_t *direction);

/* env.agent_pos and env.agent_dir of every env (minigrid.py:816-818; the 'pos' and 'dir' entries AgentExtraInfoWrapper
 * adds to the observation, wrappers.py:169-187): int32 [N][3] = (x, y, dir).  Asynchronous for a device buffer (unlike
 * mgx_get_state, which synchronises): meant to be read next to the observations of every step. */
int mgx_get_pose(mgx_handle h, int32_t *pose);

/* One lockstep env.step(actions[i]) for all N envs.  reward / done may be NULL. */
int mgx_step(mgx_handle h, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done);

/* Counters accumulated on the device since creation (for logging): env-steps executed,
 * episodes finished (done=1 transitions), sum of rewards, faults.  Synchronises the stream. */
typedef struct {
    int64_t steps, episodes;
    double reward_sum;
    int64_t invalid_actions, out_of_bounds;
} mgx_stats;
int mgx_get_stats(mgx_handle h, mgx_stats *out);

/* Asynchronous variant for multi-GPU logging: enqueues a copy of (episodes, reward_sum) as two doubles into
 * caller DEVICE memory (e.g. a torch tensor that is then all-reduced over RCCL).  No host synchronisation. */
int mgx_read_stats_async(mgx_handle h, double *out2_dev);

/* Bench helper: fill actions[T][N] (device or host pointer) with the counter-based stream
 * a = mix(seed, env0 + i, t0 + t) % 7  (same function as oracle-side tests use). */
int mgx_fill_actions(mgx_handle h, uint64_t seed, int64_t env0, int64_t t0, int64_t T, uint8_t *actions);

/* The same stream as a plain function (no handle, no GPU): action of global env `env` at step `t`. */
uint32_t mgx_action_at(uint64_t seed, int64_t env, int64_t t);

/* Timing with HIP events on the handle's stream, two measurements at once:
 *  - the SPAN: one event at mgx_profile_begin(), one at mgx_profile_end(); `span_ms` covers everything the handle
 *    enqueued in between (k_dynobs, the step kernel, one-hot / flat epilogues, k_levelgen, launch gaps) and `launches` is
 *    the number of step-kernel launches inside it;
 *  - the STEP KERNEL ALONE: every `stride`-th launch of the step kernel (k_step / k_step_fulldirect, nothing else) is
 *    bracketed by its own event pair, up to 256 pairs per span (later launches are not sampled).  mgx_profile_kernel()
 *    returns how many were sampled and the sum of their durations; valid after mgx_profile_end().  Launches recorded
 *    into mgx_rollout's graph are counted in `launches` but not sampled.
 * mgx_profile_begin(h) == mgx_profile_begin_sampled(h, 8).  stride > the number of launches samples the first launch only.
 * mgx_profile_stop() (optional) enqueues the span's end marker without waiting, so that a caller who synchronises the stream anyway
 * (bench.py's timed region) pays for one wait instead of two; mgx_profile_end() then only reads the times. */
/* Name of the step-kernel instantiation this handle launches, as a profiler prints it without the argument list
 * ("k_step<8,8,0,7>", "k_step_fulldirect<19,19,ragged>", "k_step<0,0,3,7>" = the gather form of large grids, ...): lets a
 * bench line and a rocprofv3 row be matched by name.  Writes a NUL-terminated string, returns its length or a negative status. */
int mgx_step_kernel_name(mgx_handle h, char *out, int cap);

int mgx_profile_begin(mgx_handle h);
int mgx_profile_begin_sampled(mgx_handle h, int stride);
int mgx_profile_stop(mgx_handle h);
int mgx_profile_end(mgx_handle h, int64_t *launches, double *span_ms);
int mgx_profile_kernel(mgx_handle h, int64_t *samples, double *sum_ms);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
