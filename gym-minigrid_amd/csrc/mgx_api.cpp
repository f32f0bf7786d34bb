// mgx_api.cpp -- host side of libmgx.so: the C ABI declared in include/mgx.h.
//
// Owns the per-GPU handle (SoA state in HBM, episode-start snapshot, counters, staging buffers for
// host-pointer callers, HIP stream/events) and enqueues the kernels of k_*.hip.  There is NO CPU
// fallback for any compute entry point: without a usable HIP device mgx_create fails loudly.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mgx.h"
#include "mgx_internal.h"
#include "mgx_kernels.h"

#define MGX_PROF_MAX_SAMPLES 256

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

int mgx_fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

extern "C" const char *mgx_last_error(void) { return g_err; }
#ifdef MGX_TUNING
extern "C" const char *mgx_version(void) { return "mgx 0.3 (gfx950, tuning build)"; }
#else
extern "C" const char *mgx_version(void) { return "mgx 0.3 (gfx950)"; }
#endif

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return mgx_fail(MGX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------------ handle
struct Staging { // grow-only device scratch + pinned host mirror for host-pointer callers
    void *dev = nullptr;
    size_t cap = 0;
};

struct mgx_env_s {
    mgx_config cfg;
    int64_t n = 0, n_pad = 0;
    int device = 0;
    int W = 0, H = 0, cells = 0, S = 0, LS = 0, wave_lds = 0, wpb = 4, view = 7;
    int lds_guard = 0;    // StepParams.lds_guard
    int staged_guard = 0; // ... as the staged partial form needs it (k_rollout keeps the tile in LDS whatever form the single step takes)
    int round_blocks = 0; // blocks of the step kernel resident at once on the chip (first-round stagger, k_step)
    StepLaunchCfg launch_cfg = {0, 0, -1}; // raised-priority tail / stagger of THIS handle's device (mgx_step_launch_cfg, at create)
    bool partial = true;   // the simulator emits the VxV view (else the full grid)
    int oh_nc = -1, oh_ns = 0; // one-hot epilogue channels (oh_nc < 0: none)
    bool oh_fused = false;     // ... expanded inside the step kernel (partial views up to 7x7: StepParams.onehot), no k_onehot pass
    uint8_t *tri_d = nullptr;  // triples scratch feeding the one-hot / flat epilogue
    bool flat = false;         // FlatObsWrapper epilogue (obs is float)
    float *mission_d = nullptr;   // [missions][96*27] one-hot mission blocks for k_flat
    int64_t tri_bytes = 0;     // per env
    int kernel_mode = 0; // 0 partial view, 1 full obs via the LDS tile image, 2 full obs direct (W*H % 4 == 0),
                         // 3 partial view gathered straight from HBM (grids beyond 16x16)
    int64_t obs_bytes = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint8_t *cells_d = nullptr, *cells0_d = nullptr;
    uint8_t *front_d = nullptr; // StepParams.front (gather form: kernel_mode 3)
    uint8_t *wcache_d = nullptr; // StepParams.wcache (gather form, 7x7 view, default visibility, no Dynamic-Obstacles)
    uint2 *agent_d = nullptr, *agent0_d = nullptr;
    MgxCounters *ctr_d = nullptr;
    // new level each episode: per-env MT19937 block + read index, regeneration flags
    bool stream_mode = false; // new level each episode
    bool device_levels = false; // the family draws random numbers: mgx_reset seeds and generates on the GPU
    bool one_level = false;          // the family draws none (Empty with a fixed start, DistShift, fixed TwoGoals): its level does not depend on the seed
    bool snapshot_is_level = false;  // ... and every env's episode-start snapshot holds it: mgx_reset is a restore (k_consume)
    uint32_t *mt_d = nullptr, *mt_idx_d = nullptr, *mt_init_d = nullptr;
    uint32_t *mt2_d = nullptr; // LevelGenParams.mt2 (new_level_each_episode handles)
    uint8_t *regen_d = nullptr;
    uint64_t *seed0_d = nullptr;                  // seed of the level the episode-start snapshot holds ...
    uint8_t *has_seed_d = nullptr, *reseeded_d = nullptr; // ... if any; envs the last mgx_reset really re-seeded
    // mgx_rollout: the captured T-step graph and the arguments it was captured for
    hipGraphExec_t roll_exec = nullptr;
    hipStream_t cap_stream = nullptr; // capture happens here (the caller's stream may be the null stream, which cannot capture)
    int64_t roll_T = 0;
    const void *roll_args[4] = {nullptr, nullptr, nullptr, nullptr};
    bool assume_device = false; // inside the capture: arguments were classified up front
    // Dynamic-Obstacles: obstacle order (+ episode-start copy), RNG block snapshot, folded actions
    bool dynobs = false;
    bool dyn_fused = false; // ... and its walk runs inside the step kernel (k_step_dyn: staged partial form, 7x7 view, default visibility)
    uint8_t *obst_d = nullptr, *obst0_d = nullptr, *act_d = nullptr;
    uint8_t *restart_d = nullptr; // u8[n_pad]: the in-kernel auto-reset restarted this env's episode; the next k_dynobs restores obstacle
                                  // order + RNG position first (NOT regen_d: those flags mean "k_levelgen, make this env a new level")
    uint32_t *mt0_d = nullptr, *pos0_d = nullptr, *tape_d = nullptr, *tape0_d = nullptr;
    uint32_t *sp0_d = nullptr;    // DynObsParams.sp0
    // seed schedule (mgx_set_seed_schedule): the snapshot arrays (cells0, agent0, objaux0, objcont0, obst0, mt0, pos0, tape0, sp0) hold
    // snap_banks episode starts per env, bank-major; sched_K > 0 while a schedule is installed
    int sched_K = 0, snap_banks = 1;
    uint8_t *bank_d = nullptr;    // StepParams.bank
    bool needs_full_reset = false; // a schedule was installed and the live episodes are invalid until mgx_reset(h, NULL, NULL, ...)
    bool seeded = false;          // every env has been through mgx_reset(seeds) at least once: its RNG stream exists (plain reset())
    // virtual RNG states (LevelGenParams.win / virt): seed() leaves the seed + the first MGX_SEED_WIN words of the stream, not the 2.5 KB block
    bool virt_mode = false;
    uint32_t *win_d = nullptr;
    uint8_t *virt_d = nullptr;
    bool maybe_virtual = false;   // some env may still hold a virtual state (a plain reset() materializes the masked ones first)
    // new_level_each_episode with the generator running BESIDE the steps (cheap families on the staged kernels): a ring of R next-level buffers
    // per env (snapshot banks; bank_d = the one the env's next reset consumes), the flags of step s in lg_flags[s % R], and every R/4 steps
    // k_levelgen for the flags of those steps -- in step order, the order of the env's RNG stream -- on a stream of its own: forked behind the
    // last of them, joined 3R/4 steps later, before the first of its buffers can be needed again (an env finishes at most one episode per step)
    int lg_ring = 0;                               // R buffers (a power of two, 0: one buffer, the generator behind every step)
    uint8_t *lg_flags[MGX_LG_RING_MAX] = {};       // ([0] = regen_d)
    int lg_s = 0;                                  // the flag array the NEXT step raises
    bool lg_dirty[MGX_LG_RING_MAX] = {};           // raised by a step, not handed to a generator yet
    bool lg_merge = false;                         // one k_levelgen launch per run (its arrays together) instead of one per array
    int lg_groups = 4;                             // the arrays in lg_groups runs of lg_ring / lg_groups steps, one fork / join per run
    bool lg_unjoined[4] = {};                      // the generator launches of a run: not waited for by the caller's stream yet
    hipStream_t lg_stream = nullptr;
    hipEvent_t lg_fork = nullptr, lg_join[4] = {};
    // ActionBonus / StateBonus (mgx_add_bonus): the stacking order (innermost in bits 3:0) and the wrappers' counts
    int bonus = 0;
    uint32_t *bonus_action_d = nullptr, *bonus_state_d = nullptr;
    bool dac = false; // DACWrapper (mgx_set_dac)
    // object_state: hidden Goal/Box planes (+ episode-start snapshots) and the carried object's pair
    uint8_t *objaux_d = nullptr, *objaux0_d = nullptr, *objcont_d = nullptr, *objcont0_d = nullptr;
    uint16_t *objcarry_d = nullptr;
    Staging st_in[6], st_out[4];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profiling = false, prof_stopped = false;
    int64_t prof_launches = 0, steps_total = 0;
    // per-launch samples of the step kernel alone (mgx_profile_begin_sampled): event pairs around every stride-th launch
    std::vector<hipEvent_t> prof_ev;   // 2 per sample, created on first use
    int prof_stride = 8, prof_samples = 0;
    double prof_kernel_ms = 0.0;
    int64_t prof_kernel_n = 0;
    unsigned long long base_bad_act = 0, base_oob = 0;
};

namespace {

int ensure(Staging &s, size_t bytes)
{
    if (bytes <= s.cap) return MGX_OK;
    if (s.dev) (void)hipFree(s.dev);
    s.dev = nullptr; s.cap = 0;
    HIP_TRY(hipMalloc(&s.dev, bytes));
    s.cap = bytes;
    return MGX_OK;
}

// Is `p` device memory we can hand to a kernel directly?  Asked on every call (hipPointerGetAttributes, ~0.15 us per
// pointer: 7.3 -> 7.8 us per mgx_step call from Python): remembering the answer would be wrong the day an address is
// freed as device memory and comes back as host memory, and a kernel reading a host pointer faults the GPU.
// 0: host memory, 1: memory of the current device (DeviceGuard made it the handle's), 2: memory of another GPU
int ptr_kind(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; } // plain host memory
    if (a.type == hipMemoryTypeManaged) return 1;
    if (a.type != hipMemoryTypeDevice) return 0;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return 2;
    return a.device == cur ? 1 : 2;
}
bool is_device_ptr(const void *p) { return ptr_kind(p) == 1; }
int wrong_device(const void *p)
{
    return mgx_fail(MGX_ERR_INVALID_ARG, "pointer %p is memory of another GPU than the handle's (one handle, one device: pass buffers of that device)", p);
}

// Input argument: returns a device pointer holding `bytes` of *src (staged if src is host memory).
int dev_in(mgx_handle h, int slot, const void *src, size_t bytes, const void **out, unsigned align = 1)
{
    if (!src) { *out = nullptr; return MGX_OK; }
    const int kind = h->assume_device ? 1 : ptr_kind(src);
    if (kind == 2) return wrong_device(src);
    if (kind == 1) {
        if ((uintptr_t)src & (align - 1)) return mgx_fail(MGX_ERR_INVALID_ARG, "device pointer %p is not %u-byte aligned", src, align);
        *out = src;
        return MGX_OK;
    }
    int rc = ensure(h->st_in[slot], bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->st_in[slot].dev, src, bytes, hipMemcpyHostToDevice, h->stream));
    *out = h->st_in[slot].dev;
    return MGX_OK;
}

struct OutArg { void *user = nullptr; void *dev = nullptr; size_t bytes = 0; bool staged = false; };

int dev_out(mgx_handle h, int slot, void *dst, size_t bytes, OutArg *o, unsigned align = 1)
{
    o->user = dst; o->bytes = bytes; o->staged = false; o->dev = nullptr;
    if (!dst) return MGX_OK;
    const int kind = h->assume_device ? 1 : ptr_kind(dst);
    if (kind == 2) return wrong_device(dst);
    if (kind == 1) {
        if ((uintptr_t)dst & (align - 1)) return mgx_fail(MGX_ERR_INVALID_ARG, "device pointer %p is not %u-byte aligned", dst, align);
        o->dev = dst;
        return MGX_OK;
    }
    int rc = ensure(h->st_out[slot], bytes);
    if (rc) return rc;
    o->dev = h->st_out[slot].dev;
    o->staged = true;
    return MGX_OK;
}

// copies staged outputs back; synchronises only if something was staged
int finish_out(mgx_handle h, OutArg *outs, int n_outs)
{
    bool any = false;
    for (int i = 0; i < n_outs; i++)
        if (outs[i].staged) {
            HIP_TRY(hipMemcpyAsync(outs[i].user, outs[i].dev, outs[i].bytes, hipMemcpyDeviceToHost, h->stream));
            any = true;
        }
    if (any) HIP_TRY(hipStreamSynchronize(h->stream));
    return MGX_OK;
}

StepParams base_params(mgx_handle h)
{
    StepParams p;
    memset(&p, 0, sizeof p);
    p.cells = h->cells_d; p.agent = h->agent_d; p.cells0 = h->cells0_d; p.agent0 = h->agent0_d;
    p.ctr = h->ctr_d;
    p.n = h->n; p.n_tiles = (int)(h->n_pad / 64);
    p.W = h->W; p.H = h->H; p.S = h->S; p.LS = h->LS; p.wave_lds = h->wave_lds; p.view = h->view;
    p.round_blocks = h->round_blocks; p.lds_guard = h->lds_guard;
    p.max_steps = h->cfg.max_steps; p.see_through = h->cfg.see_through_walls; p.lava_v1 = h->cfg.lava_v1;
    p.auto_reset = h->cfg.auto_reset;
    p.extended = h->cfg.extended_actions ? 1 : 0;
    p.alt_vis = h->cfg.alt_visibility ? 1 : 0;
    p.task = h->cfg.task_kind;
    p.regen = h->dynobs ? h->restart_d : (h->stream_mode ? h->regen_d : nullptr);
    p.objaux = h->objaux_d; p.objcont = h->objcont_d; p.objaux0 = h->objaux0_d; p.objcont0 = h->objcont0_d; p.objcarry = h->objcarry_d;
    p.front = h->front_d; // (Dynamic-Obstacles: k_dynobs moves cells between two steps and rewrites the entry itself)
    p.wcache = h->wcache_d;
    p.onehot = h->oh_fused ? 1 : 0;
    p.bank = h->sched_K ? h->bank_d : nullptr; p.n_banks = h->sched_K ? h->sched_K : 1; p.bank_envs = h->sched_K ? h->n_pad : 0;
    if (h->lg_ring) { p.bank = h->bank_d; p.n_banks = h->lg_ring; p.bank_envs = h->n_pad; p.ring = h->lg_ring; }
    p.dac = h->dac ? 1 : 0;
    p.bonus = h->bonus; p.bonus_na = h->cfg.extended_actions ? 9 : MGX_NUM_ACTIONS;
    p.bonus_action = h->bonus_action_d; p.bonus_state = h->bonus_state_d;
    return p;
}

// The gather form's per-env "cell in front" cache describes the state the last observation pass saw: whoever changes cells or
// poses behind the step kernel's back (set_state, reset, set_task's snapshot, ...) marks every entry unknown.
int forget_front(mgx_handle h)
{
    if (h->front_d) HIP_TRY(hipMemsetAsync(h->front_d, 0, (size_t)h->n_pad, h->stream));
    if (h->wcache_d) HIP_TRY(hipMemsetAsync(h->wcache_d, 0, (size_t)h->n_pad * 64, h->stream)); // (every pose tag: no record belongs to a pose any more)
    return MGX_OK;
}

// which of the families of mgx_mission_row (mgx_kernels.h) the handle's missions belong to
int mission_family(const mgx_config *cfg)
{
    switch (cfg->level_kind) {
    case MGX_LEVEL_FETCH: return MGX_MF_FETCH;
    case MGX_LEVEL_GOTOOBJECT: return MGX_MF_GOTOOBJECT;
    case MGX_LEVEL_UNLOCK: return cfg->level_arg0 ? MGX_MF_PICKUP : MGX_MF_CONST;
    case MGX_LEVEL_KEYCORRIDOR: return MGX_MF_PICKUP;
    case MGX_LEVEL_LOCKEDROOM: return MGX_MF_LOCKEDROOM;
    case MGX_LEVEL_PUTNEAR: return MGX_MF_PUTNEAR;
    default: return MGX_MF_CONST;
    }
}

LevelGenParams levelgen_params(mgx_handle h)
{
    LevelGenParams g;
    memset(&g, 0, sizeof g);
    g.cfg = h->cfg;
    g.mt = h->mt_d; g.mt2 = h->mt2_d; g.mt_idx = h->mt_idx_d; g.regen = h->regen_d; g.cells0 = h->cells0_d; g.agent0 = h->agent0_d;
    g.objaux0 = h->objaux0_d; g.objcont0 = h->objcont0_d;
    g.ctr = h->ctr_d;
    g.n = h->n; g.n_tiles = (int)(h->n_pad / 64); g.S = h->S;
    g.win = h->win_d; g.virt = h->virt_d; g.seed0 = h->seed0_d; g.mt_init = h->mt_init_d;
    g.bank_envs = h->lg_ring ? h->n_pad : 0;
    return g;
}

DynObsParams dynobs_params(mgx_handle h)
{
    DynObsParams d;
    memset(&d, 0, sizeof d);
    d.cells = h->cells_d; d.cells0 = h->cells0_d; d.agent = h->agent_d; d.act_out = h->act_d; d.regen = h->restart_d;
    d.obst = h->obst_d; d.obst0 = h->obst0_d; d.mt = h->mt_d; d.mt0 = h->mt0_d; d.pos = h->mt_idx_d; d.pos0 = h->pos0_d;
    d.tape = h->tape_d; d.tape0 = h->tape0_d; d.front = h->front_d; d.sp0 = h->sp0_d; d.mt_idx = h->mt_idx_d;
    d.bank = h->sched_K ? h->bank_d : nullptr; d.bank_envs = h->sched_K ? h->n_pad : 0;
    d.n = h->n; d.W = h->W; d.H = h->H; d.S = h->S; d.n_obst = h->cfg.level_arg0;
    d.n_tiles = (int)(h->n_pad / 64); d.LS = h->LS; d.wave_lds = mgx_dynobs_wave_lds(h->LS);
    return d;
}

// env.seed(seeds[i]) of the masked envs on the device, in the handle's form (full block / virtual state)
hipError_t launch_seed(mgx_handle h, const uint64_t *seeds_dev, const uint8_t *mask_dev, int skip_same)
{
    if (h->virt_mode) {
        h->maybe_virtual = true;
        return mgx_launch_seed_window(seeds_dev, mask_dev, h->mt_init_d, h->win_d, h->virt_d, h->mt_idx_d, h->regen_d, h->seed0_d, h->has_seed_d, h->reseeded_d,
                                      skip_same, h->n, h->stream);
    }
    return mgx_launch_seed(seeds_dev, mask_dev, h->mt_init_d, h->mt_d, h->mt2_d, h->mt_idx_d, h->regen_d, h->seed0_d, h->has_seed_d, h->reseeded_d, skip_same,
                           h->n, h->stream);
}

int launch_levelgen(mgx_handle h)
{
    HIP_TRY(mgx_launch_levelgen(levelgen_params(h), h->stream));
    return MGX_OK;
}

// new_level_each_episode handles, once every env has been seeded: the snapshot arrays hold the NEXT level of each env, already drawn from its RNG
// stream.  State injected from outside (set_state, set_task, set_object_state: attribute assignments in the reference, which draw nothing) then
// changes the current episode only and leaves that level where it is.
bool next_level_waiting(mgx_handle h) { return h->stream_mode && h->seeded && h->device_levels; }

// Ring handles: generators may still be running beside the caller's stream, and the last step's flags may not have been handed to one yet.
// Every entry point that reads or rewrites next-level buffers, RNG state or flags from the host side first brings all of that onto the caller's stream.
int lg_drain(mgx_handle h)
{
    if (!h->lg_ring) return MGX_OK;
    for (int pr = 0; pr < 4; pr++)
        if (h->lg_unjoined[pr]) { HIP_TRY(hipStreamWaitEvent(h->stream, h->lg_join[pr], 0)); h->lg_unjoined[pr] = false; }
    for (int k = 0; k < h->lg_ring; k++) { // (the steps of a run whose generators have not been launched)
        const int a = (h->lg_s + k) & (h->lg_ring - 1); // oldest first
        if (!h->lg_dirty[a]) continue;
        LevelGenParams g = levelgen_params(h);
        g.regen = h->lg_flags[a];
        HIP_TRY(mgx_launch_levelgen(g, h->stream));
        h->lg_dirty[a] = false;
    }
    h->lg_s = 0;
    return MGX_OK;
}

// ... and after a host-side reset of the single-buffer kind (buffer 0 = the masked envs' next level) the ring is completed behind it
int lg_ring_init(mgx_handle h, const uint8_t *mask_dev)
{
    if (!h->lg_ring) return MGX_OK;
    for (int b = 1; b < h->lg_ring; b++) {
        HIP_TRY(mgx_launch_ring_init(h->bank_d, h->regen_d, mask_dev, h->n, b + 1, h->stream));
        const int rc = launch_levelgen(h);
        if (rc) return rc;
    }
    return MGX_OK;
}

int resize_snapshots(mgx_handle h, int K);

// Every entry point runs with the handle's device current and puts the caller's device back on the way out (a process
// that drives several GPUs -- or torch with another current device -- must not find it changed behind its back).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    int enter_device(int device, const char *fn)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev == device) return MGX_OK;
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return mgx_fail(MGX_ERR_HIP, "%s: hipSetDevice(%d): %s", fn, device, hipGetErrorString(e));
        changed = prev >= 0;
        return MGX_OK;
    }
    int enter(mgx_handle h, const char *fn)
    {
        if (!h) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: null handle", fn);
        return enter_device(h->device, fn);
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

int read_counters(mgx_handle h, MgxCounters *c)
{
    HIP_TRY(hipMemcpyAsync(c, h->ctr_d, sizeof *c, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < MGX_CTR_SHARDS; i++) { c->invalid_actions += c->shard[i].invalid_actions; c->out_of_bounds += c->shard[i].out_of_bounds; }
    return MGX_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------ lifecycle
extern "C" int mgx_create(const mgx_config *cfg, int64_t n_envs, int device, mgx_handle *out)
{
    if (!cfg || !out) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: null argument");
    *out = nullptr;
    if (n_envs <= 0) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: n_envs must be positive");
    if (cfg->width < 3 || cfg->height < 3 || cfg->width > 255 || cfg->height > 255)
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: grid %dx%d outside 3..255 (Grid.__init__ asserts >= 3)", cfg->width, cfg->height);
    if (cfg->max_steps <= 0) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: max_steps must be positive");
    if (cfg->level_kind == MGX_LEVEL_OBSTRUCTEDMAZE && (cfg->level_arg0 & 1) && !cfg->object_state)
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: ObstructedMaze levels with keys hidden in boxes need object_state = 1 (Box.contains)");
    if (cfg->task_kind < MGX_TASK_NONE || cfg->task_kind > MGX_TASK_TWOGOALS)
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: bad task_kind %d", cfg->task_kind);
    if ((cfg->task_kind == MGX_TASK_DYNOBS) != (cfg->level_kind == MGX_LEVEL_DYNOBS))
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: MGX_TASK_DYNOBS and MGX_LEVEL_DYNOBS go together (the obstacle walk continues the level's RNG stream)");
    if (cfg->task_kind == MGX_TASK_DYNOBS) {
        if (cfg->new_level_each_episode || cfg->object_state || cfg->extended_actions)
            return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: Dynamic-Obstacles handles do not take new_level_each_episode / object_state / extended_actions");
        if (cfg->width > 16 || cfg->height > 16 || cfg->level_arg0 < 0 || cfg->level_arg0 > 8)
            return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: Dynamic-Obstacles needs W, H <= 16 and 0..8 obstacles");
    }
    if (cfg->task_kind != MGX_TASK_NONE && cfg->max_steps > 65535)
        return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: task rules need max_steps <= 65535");
    if (cfg->obs_mode < MGX_OBS_PARTIAL || cfg->obs_mode > MGX_OBS_FULL_FLAT)
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: bad obs_mode %d", cfg->obs_mode);
    const int view = cfg->agent_view_size ? cfg->agent_view_size : MGX_VIEW;
    if (view != 3 && view != 5 && view != 7 && view != 9 && view != 11)
        return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: agent_view_size %d (supported: 3, 5, 7, 9, 11)", view);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return mgx_fail(MGX_ERR_HIP, "mgx_create: no HIP device (%s); libmgx has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_create: device %d of %d", device, ndev);
    DeviceGuard dev_guard;
    { int rc_ = dev_guard.enter_device(device, "mgx_create"); if (rc_) return rc_; }

    mgx_env_s *h = new (std::nothrow) mgx_env_s();
    if (!h) return mgx_fail(MGX_ERR_HIP, "mgx_create: out of host memory");
    h->cfg = *cfg;
    h->n = n_envs;
    h->n_pad = (n_envs + 63) / 64 * 64;
    h->device = device;
    h->W = cfg->width; h->H = cfg->height;
    h->cells = h->W * h->H;
    h->S = (h->cells + 3) & ~3;
    h->LS = h->S + (((h->S >> 2) & 1) ? 0 : 4); // odd dword stride per env in LDS
    h->view = view;
    h->partial = cfg->obs_mode == MGX_OBS_PARTIAL || cfg->obs_mode == MGX_OBS_PARTIAL_ONEHOT || cfg->obs_mode == MGX_OBS_PARTIAL_FLAT;
    h->flat = cfg->obs_mode == MGX_OBS_PARTIAL_FLAT || cfg->obs_mode == MGX_OBS_FULL_FLAT;
    if (cfg->obs_mode == MGX_OBS_PARTIAL_ONEHOT) { h->oh_nc = 7; h->oh_ns = 3; }
    else if (cfg->obs_mode == MGX_OBS_FULL_ONEHOT) { h->oh_nc = 7; h->oh_ns = 4; }
    else if (cfg->obs_mode == MGX_OBS_FULL_ONEHOT_NOCOLOR) { h->oh_nc = 0; h->oh_ns = 4; }
    const int obs_img = h->partial ? 32 * view * view * 3 : 0; // half-tile output image
    int need = 64 * h->LS;
    if (obs_img > need) need = obs_img;
    if (!h->partial) need = ((need + 15) & ~15) + 3072; // + transpose scratch of emit_full_obs
    // FullyObs: the direct kernels unless W*H is small AND not a multiple of 4 (5x5, 7x7, 9x9, 11x11: a block per tile
    // has too little to do there and the wave-per-tile LDS form measures faster: 110 vs 138 us at 1 Mi 9x9 envs).
    // MGX_FULL_KERNEL=lds / direct overrides the rule (tests, tuning).
    const char *ff = getenv("MGX_FULL_KERNEL");
    const bool direct_ok = h->cells <= 65535;
    const bool direct = direct_ok && (ff ? !strcmp(ff, "direct") : (h->cells % 4 == 0 || h->cells > 128));
    h->kernel_mode = h->partial ? 0 : (direct ? 2 : 1);
    if (h->kernel_mode == 2) need = 16; // k_step_fulldirect keeps no tile image in LDS
    const char *force = getenv("MGX_PARTIAL_KERNEL"); // "staged" / "gather": override the size rule (tests, tuning)
    // From 13x13 up each lane gathers its V x 4 / 8 / 12-byte window straight from its row (k_step MODE 3; V = 7 in the figures) instead of
    // the wave staging the whole tile in LDS.  Measured with all seven window loads in flight at once (round 3): 8x8 41.7 vs 39.6 us
    // staged, 9x9 48.3 vs 43.2, 11x11 50.9 vs 50.0, 16x8 32.9 vs 28.7 -- but 13x13 32.3 vs 34.3, 16x16 32.0 vs 43.7 (the 16.6 KB tile
    // image left two waves per SIMD), and past 16x16 the image does not pay at all.
    const bool gather_ok = h->partial; // (every view size, both visibility rules, with or without the hidden-object planes)
    if (gather_ok && (force ? !strcmp(force, "gather") : h->S >= 160)) { h->kernel_mode = 3; need = obs_img; }
    const int guard = ((view - 1) * h->H + view / 2 + 15) & ~15; // StepParams.lds_guard (staged partial form only)
    // (a forced staged form whose tile image cannot fit the LDS -- past ~50x50 -- gathers after all)
    if (h->partial && h->kernel_mode == 0 && need + 2 * guard > 160 * 1024) { h->kernel_mode = 3; need = obs_img; }
    if (h->kernel_mode == 3) { // the lanes' window excerpts (V columns x 4 / 8 / 12 rows, odd dword stride): read before the output image overlays them
        const int rs = view <= 3 ? 4 : (view <= 7 ? 8 : 12), slot = ((view * rs / 4) | 1) * 4;
        if (64 * slot > need) need = 64 * slot;
    }
    // OneHotPartialObsWrapper inside the step kernel (views up to 7x7): the wave parks its 64 x V*V cell codes and builds the image a quarter
    // tile (16 envs x V*V*21 bytes) at a time.  Built and pinned (tests run both forms), and it moves 1,115 instead of 1,409 bytes per env-step --
    // but it is the SLOWER form (256 us against 39 + 208 for k_step + k_onehot at 1 Mi Empty-8x8 envs: its 19.4 KB of LDS per wave leave two
    // waves per SIMD, and eight waves queue on one LDS pipe for 3,500 cycles of image traffic per tile; profiles/r04_onehot_fused.txt), so
    // the default stays the two-kernel form and MGX_ONEHOT=fused selects this one.
    const char *ohf = getenv("MGX_ONEHOT");
    h->oh_fused = cfg->obs_mode == MGX_OBS_PARTIAL_ONEHOT && view <= 7 && !cfg->alt_visibility && cfg->task_kind != MGX_TASK_DYNOBS && ohf && !strcmp(ohf, "fused");
    if (h->oh_fused) {
#ifndef MGX_OH_UNIT
#define MGX_OH_UNIT 16
#endif
        const int oh_need = 64 * ((view * view + 3) & ~3) + ((MGX_OH_UNIT * view * view * 21 + 15) & ~15);
        if (oh_need > need) need = oh_need;
    }
    h->lds_guard = h->kernel_mode == 0 ? guard : 0;
    h->staged_guard = guard;
    h->wave_lds = (need + 15) & ~15;
    if (const char *e = MGX_TUNE_ENV("MGX_EXTRA_LDS")) h->wave_lds += atoi(e) & ~15; // (tuning builds: what the step costs with fewer resident blocks)
    // a family that draws no random numbers (Empty with a fixed start) has only one level: nothing to generate
    const bool uses_rng = cfg->level_kind != MGX_LEVEL_NONE && cfg->level_kind != MGX_LEVEL_DISTSHIFT &&
                          !((cfg->level_kind == MGX_LEVEL_EMPTY || cfg->level_kind == MGX_LEVEL_TWOGOALS) && cfg->level_arg0 == 0);
    h->device_levels = uses_rng && h->cells <= 4096;
    h->one_level = cfg->level_kind != MGX_LEVEL_NONE && !uses_rng;
    h->stream_mode = cfg->new_level_each_episode && uses_rng;
    h->tri_bytes = h->partial ? view * view * 3 : (int64_t)h->cells * 3;
    h->obs_bytes = h->oh_nc < 0 ? h->tri_bytes : h->tri_bytes / 3 * (11 + h->oh_nc + h->oh_ns);
    if (h->flat) h->obs_bytes = (h->tri_bytes + MGX_FLAT_MISSION) * (int64_t)sizeof(float);
    const int LDS_DEFAULT = 64 * 1024, LDS_MAX = 160 * 1024;
    if (h->wave_lds + 2 * h->lds_guard > LDS_MAX) {
        int rc = mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: a %dx%d tile (64 envs) needs %d B of LDS > %d", h->W, h->H, h->wave_lds, LDS_MAX);
        delete h;
        return rc;
    }
    h->wpb = (LDS_DEFAULT - 2 * h->lds_guard) / h->wave_lds;
    if (h->wpb > 4) h->wpb = 4;
    if (h->oh_fused && h->wpb > 1 && MGX_OH_UNIT == 16) h->wpb = 1; // (19.4 KB per wave at the 7x7 view: eight one-wave blocks fit a CU's LDS, two three-wave blocks would)
    if (const char *e = MGX_TUNE_ENV("MGX_WPB")) { const int w = atoi(e); if (w >= 1 && w < h->wpb) h->wpb = w; } // (tuning runs: waves per block of k_step)
    const bool raise_lds = h->wpb < 1; // (done below, once the buffers that select the kernel instantiation exist)
    if (raise_lds) h->wpb = 1;
#define CREATE_TRY(expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            int rc_ = mgx_fail(MGX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            mgx_destroy(h);                                                                  \
            return rc_;                                                                      \
        }                                                                                    \
    } while (0)
    if (cfg->new_level_each_episode) {
        const char *why = nullptr;
        if (!cfg->auto_reset) why = "needs auto_reset = 1";
        else if (cfg->level_kind == MGX_LEVEL_NONE) why = "needs a level_kind with a built-in generator";
        else if (h->cells > 4096) why = "supports grids up to W*H = 4096";
        if (why) {
            int rc = mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: new_level_each_episode %s", why);
            delete h;
            return rc;
        }
    }
    CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    const size_t cb = (size_t)h->n_pad * h->S, ab = (size_t)h->n_pad * sizeof(uint2);
    CREATE_TRY(hipMalloc((void **)&h->cells_d, cb));
    CREATE_TRY(hipMalloc((void **)&h->cells0_d, cb));
    CREATE_TRY(hipMalloc((void **)&h->agent_d, ab));
    CREATE_TRY(hipMalloc((void **)&h->agent0_d, ab));
    CREATE_TRY(hipMalloc((void **)&h->ctr_d, sizeof(MgxCounters)));
    if ((h->oh_nc >= 0 && !h->oh_fused) || h->flat) CREATE_TRY(hipMalloc((void **)&h->tri_d, (size_t)h->n * h->tri_bytes + 16));
    if (h->flat) { // the family's mission strings as character codes (wrappers.py:563-571)
        const int family = mission_family(cfg), rows = mgx_mission_rows(family);
        std::vector<float> tab((size_t)rows * MGX_FLAT_MISSION, 0.f);
        std::vector<char> filled((size_t)rows, 0);
        for (uint32_t task = 0; task < 65536u; task++) { // every task word that names a mission fills its row (once)
            const int r = mgx_mission_row(family, task);
            if (r < 0 || r >= rows || filled[(size_t)r]) continue;
            if (family == MGX_MF_CONST && task) break;
            // (the row of a pick-up target is its colour; its type is the family's: the box of UnlockPickup, the ball of KeyCorridor)
            if (family == MGX_MF_PICKUP && (task & 15u) != (uint32_t)(cfg->level_kind == MGX_LEVEL_UNLOCK ? MGX_K_BOX : MGX_K_BALL)) continue;
            char m[128];
            const int len = mgx_mission(cfg, task, m, (int)sizeof m);
            if (len < 0) continue; // not a task word of this family
            if (len > 96) { // assert len(mission) <= self.maxStrLen
                int rc = mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: mission string too long (%d chars)", len);
                mgx_destroy(h);
                return rc;
            }
            filled[(size_t)r] = 1;
            int code = -1;
            for (int i = 0; i < len; i++) {
                const char ch = (char)(m[i] >= 'A' && m[i] <= 'Z' ? m[i] - 'A' + 'a' : m[i]);
                if (ch >= 'a' && ch <= 'z') code = ch - 'a';
                else if (ch == ' ') code = 26;
                // any other character re-uses the previous character's code (chNo keeps its value, wrappers.py:565-570)
                if (code < 0) { int rc = mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: mission starts with a character the wrapper cannot encode"); mgx_destroy(h); return rc; }
                tab[(size_t)r * MGX_FLAT_MISSION + (size_t)i * 27 + code] = 1.f; // strArray[idx, chNo] = 1
            }
        }
        CREATE_TRY(hipMalloc((void **)&h->mission_d, tab.size() * sizeof(float)));
        CREATE_TRY(hipMemcpy(h->mission_d, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (cfg->object_state) {
        for (uint8_t **pp : {&h->objaux_d, &h->objaux0_d}) { CREATE_TRY(hipMalloc((void **)pp, cb)); CREATE_TRY(hipMemsetAsync(*pp, 0, cb, h->stream)); }
        for (uint8_t **pp : {&h->objcont_d, &h->objcont0_d}) { CREATE_TRY(hipMalloc((void **)pp, cb)); CREATE_TRY(hipMemsetAsync(*pp, MGX_CODE_EMPTY, cb, h->stream)); }
        CREATE_TRY(hipMalloc((void **)&h->objcarry_d, (size_t)h->n_pad * sizeof(uint16_t)));
        CREATE_TRY(hipMemsetD16Async((hipDeviceptr_t)h->objcarry_d, (unsigned short)(MGX_CODE_EMPTY << 8), (size_t)h->n_pad, h->stream));
    }
    CREATE_TRY(hipMemsetAsync(h->cells_d, 0, cb, h->stream));
    CREATE_TRY(hipMemsetAsync(h->cells0_d, 0, cb, h->stream));
    CREATE_TRY(hipMemsetAsync(h->agent_d, 0, ab, h->stream));
    CREATE_TRY(hipMemsetAsync(h->agent0_d, 0, ab, h->stream));
    CREATE_TRY(hipMemsetAsync(h->ctr_d, 0, sizeof(MgxCounters), h->stream));
    if (h->kernel_mode == 3) {
        CREATE_TRY(hipMalloc((void **)&h->front_d, (size_t)h->n_pad));
        CREATE_TRY(hipMemsetAsync(h->front_d, 0, (size_t)h->n_pad, h->stream));
        // (MGX_GATHER_CACHE=off: the gather form without its per-env window records -- tests and A/B runs)
        const char *gc = getenv("MGX_GATHER_CACHE");
        // (not for TwoGoals: its episodes end on the `done` action, one env in seven per step under a random policy, and a record per pose that is
        // thrown away that often measured 11 % slower -- profiles/r04_gather_window_cache.txt; MGX_GATHER_CACHE=on forces it)
        const bool wc_rule = cfg->task_kind != MGX_TASK_TWOGOALS;
        if (view == 7 && !cfg->alt_visibility && cfg->task_kind != MGX_TASK_DYNOBS && (gc ? !strcmp(gc, "on") : wc_rule)) {
            CREATE_TRY(hipMalloc((void **)&h->wcache_d, (size_t)h->n_pad * 64));
            CREATE_TRY(hipMemsetAsync(h->wcache_d, 0, (size_t)h->n_pad * 64, h->stream));
        }
    }
    if (h->device_levels) {
        uint32_t init[624];
        mgx_mt_init_table(init);
        CREATE_TRY(hipMalloc((void **)&h->mt_init_d, sizeof init));
        CREATE_TRY(hipMemcpyAsync(h->mt_init_d, init, sizeof init, hipMemcpyHostToDevice, h->stream));
        CREATE_TRY(hipStreamSynchronize(h->stream)); // `init` is a stack array
        CREATE_TRY(hipMalloc((void **)&h->mt_d, ((size_t)h->n_pad * 624 + 64) * sizeof(uint32_t))); // (+ slack: k_levelgen's window refills read whole windows)
        CREATE_TRY(hipMalloc((void **)&h->mt_idx_d, (size_t)h->n_pad * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc((void **)&h->regen_d, (size_t)h->n_pad));
        CREATE_TRY(hipMemsetAsync(h->mt_d, 0, ((size_t)h->n_pad * 624 + 64) * sizeof(uint32_t), h->stream));
        // new_level_each_episode handles of the draw-heavy families keep the NEXT block ready as well (LevelGenParams.mt2): a level may then
        // run across the end of its block on the lane-per-level path instead of being generated again by a whole wave (MultiRoom-N6 uses
        // 287 of a block's 624 words per level: every second level crossed).  The cheap families (tens of words per level) cross rarely
        // and measured 5 % slower with the second block's upkeep (LavaCrossing, 1 Mi envs: 60.0 -> 63.1 us per step).
        const bool cheap_draws = cfg->level_kind == MGX_LEVEL_EMPTY || cfg->level_kind == MGX_LEVEL_DOORKEY || cfg->level_kind == MGX_LEVEL_CROSSING ||
                                 cfg->level_kind == MGX_LEVEL_LAVAGAP || cfg->level_kind == MGX_LEVEL_DISTSHIFT;
        if (h->stream_mode && !(cfg->task_kind == MGX_TASK_DYNOBS) && !cheap_draws) {
            // (2.5 KB per env on top of the first block's 2.5 KB -- include/mgx.h, new_level_each_episode.  An optimisation only: a handle that
            // cannot get it keeps the single-block rule, where a level that runs past its block is generated again by a whole wave.)
            if (hipMalloc((void **)&h->mt2_d, ((size_t)h->n_pad * 624 + 64) * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); h->mt2_d = nullptr; }
            else CREATE_TRY(hipMemsetAsync(h->mt2_d, 0, ((size_t)h->n_pad * 624 + 64) * sizeof(uint32_t), h->stream));
        }
        CREATE_TRY(hipMemsetAsync(h->mt_idx_d, 0, (size_t)h->n_pad * sizeof(uint32_t), h->stream));
        CREATE_TRY(hipMemsetAsync(h->regen_d, 0, (size_t)h->n_pad, h->stream));
        CREATE_TRY(hipMalloc((void **)&h->seed0_d, (size_t)h->n_pad * sizeof(uint64_t)));
        CREATE_TRY(hipMalloc((void **)&h->has_seed_d, (size_t)h->n_pad));
        CREATE_TRY(hipMalloc((void **)&h->reseeded_d, (size_t)h->n_pad));
        CREATE_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream));
        CREATE_TRY(hipMemsetAsync(h->reseeded_d, 0, (size_t)h->n_pad, h->stream));
        // Virtual RNG states: handles that re-seed at every episode boundary (no stream mode; Dynamic-Obstacles draws inside step()) of the
        // families whose levels take a few dozen draws (tools/draw_stats.cpp: the share of levels past 64 words is 0 for most of them, 2 %
        // for DoorKey-5x5, 6 % for Fetch-5x5; KeyCorridor, Playground, MultiRoom and the 16x16 ObstructedMaze ids draw 50 ... 300 and keep
        // the full block).  MGX_SEED_FORM=full / window overrides the rule (tests, tuning).
        const int k = cfg->level_kind;
        const bool short_draws = k == MGX_LEVEL_EMPTY || k == MGX_LEVEL_DOORKEY || k == MGX_LEVEL_CROSSING || k == MGX_LEVEL_LAVAGAP || k == MGX_LEVEL_FETCH ||
                                 k == MGX_LEVEL_GOTODOOR || k == MGX_LEVEL_FOURROOMS || k == MGX_LEVEL_GOTOOBJECT || k == MGX_LEVEL_REDBLUEDOORS ||
                                 k == MGX_LEVEL_MEMORY || k == MGX_LEVEL_UNLOCK || k == MGX_LEVEL_LOCKEDROOM || k == MGX_LEVEL_PUTNEAR || k == MGX_LEVEL_TWOGOALS ||
                                 (k == MGX_LEVEL_OBSTRUCTEDMAZE && cfg->level_arg1 == 0);
        const char *sf = getenv("MGX_SEED_FORM");
        h->virt_mode = !h->stream_mode && cfg->task_kind != MGX_TASK_DYNOBS && (sf ? !strcmp(sf, "window") : short_draws);
        if (h->virt_mode) {
            CREATE_TRY(hipMalloc((void **)&h->win_d, ((size_t)h->n_pad * MGX_SEED_WIN + 64) * sizeof(uint32_t))); // (+ slack: see mt_d)
            CREATE_TRY(hipMalloc((void **)&h->virt_d, (size_t)h->n_pad));
            CREATE_TRY(hipMemsetAsync(h->win_d, 0, ((size_t)h->n_pad * MGX_SEED_WIN + 64) * sizeof(uint32_t), h->stream));
            CREATE_TRY(hipMemsetAsync(h->virt_d, 0, (size_t)h->n_pad, h->stream));
        }
    }
    if (cfg->task_kind == MGX_TASK_DYNOBS) {
        if (!h->device_levels) { int rc = mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_create: Dynamic-Obstacles needs the on-device level generator"); mgx_destroy(h); return rc; }
        h->dynobs = true;
        // (form selection, read in every build: MGX_DYNOBS=split keeps the walk a kernel of its own -- tests run both)
        const char *df = getenv("MGX_DYNOBS");
        // (16x16: the single step gathers, mode 3; the fused kernel stages the tile -- the walk needs it in LDS anyway)
        h->dyn_fused = (h->kernel_mode == 0 || (h->kernel_mode == 3 && h->partial)) && view == 7 && !cfg->alt_visibility && !(df && !strcmp(df, "split")) &&
                       mgx_dynobs_wave_lds(h->LS) + 2 * h->staged_guard <= 64 * 1024;
        CREATE_TRY(hipMalloc((void **)&h->obst_d, (size_t)h->n_pad * 8));
        CREATE_TRY(hipMalloc((void **)&h->obst0_d, (size_t)h->n_pad * 8));
        CREATE_TRY(hipMalloc((void **)&h->act_d, (size_t)h->n_pad));
        CREATE_TRY(hipMalloc((void **)&h->restart_d, (size_t)h->n_pad));
        CREATE_TRY(hipMemsetAsync(h->restart_d, 0, (size_t)h->n_pad, h->stream));
        CREATE_TRY(hipMalloc((void **)&h->mt0_d, (size_t)h->n_pad * 624 * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc((void **)&h->pos0_d, (size_t)h->n_pad * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc((void **)&h->tape_d, (size_t)h->n_pad * MGX_DYN_TAPE_DW * sizeof(uint32_t)));
        CREATE_TRY(hipMalloc((void **)&h->tape0_d, (size_t)h->n_pad * MGX_DYN_TAPE_DW * sizeof(uint32_t)));
        CREATE_TRY(hipMemsetAsync(h->tape_d, 0, (size_t)h->n_pad * MGX_DYN_TAPE_DW * sizeof(uint32_t), h->stream));
        CREATE_TRY(hipMemsetAsync(h->tape0_d, 0, (size_t)h->n_pad * MGX_DYN_TAPE_DW * sizeof(uint32_t), h->stream));
        CREATE_TRY(hipMemsetAsync(h->obst_d, 0, (size_t)h->n_pad * 8, h->stream));
        CREATE_TRY(hipMemsetAsync(h->obst0_d, 0, (size_t)h->n_pad * 8, h->stream));
        CREATE_TRY(hipMemsetAsync(h->mt0_d, 0, (size_t)h->n_pad * 624 * sizeof(uint32_t), h->stream));
        CREATE_TRY(hipMemsetAsync(h->pos0_d, 0, (size_t)h->n_pad * sizeof(uint32_t), h->stream));
        CREATE_TRY(hipMalloc((void **)&h->sp0_d, (size_t)h->n_pad * sizeof(uint32_t)));
        CREATE_TRY(hipMemsetAsync(h->sp0_d, 0, (size_t)h->n_pad * sizeof(uint32_t), h->stream));
    }
    {   // The generator beside the steps (new_level_each_episode, partial views).  k_levelgen behind every step is compute (MT19937 draws, rejection
        // loops: 17 us for the ~50 k levels a step of 1 Mi LavaCrossing envs ends, 10 % of the chip's issue slots: one wave's dependent chain) in
        // front of a memory-bound step; beside the steps it fills the step's stalls.  Two things decide the form, both measured
        // (profiles/README.md, round 4): a kernel with both kinds of block runs every step wave at the generator's 128 VGPRs (74 us against 62
        // for the two launches), and an event pair between two streams costs 10-20 us on this stack, so the coupling is loose: a fork every
        // R/4 steps, its join due 3R/4 steps later (R < 8: halves).  us per step at 1 Mi / 512 Ki / 64 Ki LavaCrossingS9N1 envs: one buffer 59.0 / 39.5 / 21.2,
        // R = 4: 58.7 / 38.5 / 19.1, R = 8: 53.8 / 33.4 / 16.5, R = 16: 51.6 / 31.0 / 15.3 (replay 43.1 / 23.0 / 8.4).  Other families at R = 16
        // (profiles/r04_levelgen_ring_families.txt): Unlock 62.6 -> 47.6, Fetch 71.9 -> 60.3, GoToDoor 97.4 -> 82.1, RedBlueDoors 66.4 -> 54.8,
        // MemoryS13Random 47.4 -> 38.1, KeyCorridorS3R3 74.5 -> 66.7; those whose episodes end rarely (DoorKey, Empty-Random, SimpleCrossing,
        // FourRooms, LockedRoom) within 1 us either way; MultiRoom-N6, whose 262,144 levels of one step are a 5.7 ms launch that any join
        // waits for, 68.7 -> 74.0: it keeps one buffer.  (R * (S + 8) B per env, three planes with object_state: 1.7 KB at 9x9.)
        // MGX_LG_RING=off | 2 | 4 | 8 | 16: one buffer and k_levelgen behind every step on the caller's stream (tests, A/B), or another depth.
        const char *lf = getenv("MGX_LG_RING");
        const int R = (lf && strcmp(lf, "on")) ? atoi(lf) : 16; // ("off" -> 0)
        // (every partial-view step kernel -- staged, gather, with or without hidden object state -- is step_body, which knows the ring; the
        // FullyObs kernels do not)
        // (the families it costs 1-4 % instead: their episodes end together, at the time-out, and a burst of N levels is as long beside the
        // steps as behind one -- MultiRoom-N6 68.7 -> 74.0, FourRooms 36.4 -> 37.2, LockedRoom 20.5 -> 22.0, Playground 37.9 -> 38.5 -- or there are
        // more levels per step than steps' worth of work: GoToObject, whose `done` action ends two episodes in seven, 251 -> 260.  They keep
        // one buffer; MGX_LG_RING set to a depth overrides the rule)
        const int lk = cfg->level_kind;
        const bool ring_pays = lk != MGX_LEVEL_MULTIROOM && lk != MGX_LEVEL_FOURROOMS && lk != MGX_LEVEL_LOCKEDROOM && lk != MGX_LEVEL_PLAYGROUND &&
                               lk != MGX_LEVEL_GOTOOBJECT &&
                               !((lk == MGX_LEVEL_DOORKEY || lk == MGX_LEVEL_EMPTY) && h->kernel_mode == 3); // (DoorKey-16x16: 33.0 -> 34.2, time-outs only)
        if (h->stream_mode && h->partial && (h->kernel_mode == 0 || h->kernel_mode == 3) && (ring_pays || lf) &&
            (R == 2 || R == 4 || R == 8 || R == 16)) h->lg_ring = R;
        // R next-level buffers (+ hidden planes) per env; a handle too large for them keeps one buffer and the generator behind every step
        if (h->lg_ring && resize_snapshots(h, R) != MGX_OK) { h->lg_ring = 0; (void)hipGetLastError(); }
        if (h->lg_ring) {
            int prio_lo = 0, prio_hi = 0;
            CREATE_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            CREATE_TRY(hipStreamCreateWithPriority(&h->lg_stream, hipStreamNonBlocking, prio_hi));
            CREATE_TRY(hipEventCreateWithFlags(&h->lg_fork, hipEventDisableTiming));
            for (int a = 0; a < 4; a++) CREATE_TRY(hipEventCreateWithFlags(&h->lg_join[a], hipEventDisableTiming));
            // One generator launch per run of the ring (its four steps' flag arrays together, k_levelgen<true>) where a step ends few episodes
            // and every launch is as long as its slowest level -- us per step, a launch per array -> one per run: KeyCorridorS3R3 66.8 -> 41.7,
            // GoToDoor 85.9 -> 56.6, Fetch 59.6 -> 52.3, PutNear 120.6 -> 98.2, ObstructedMaze-1Dlhb 22.2 -> 21.7; where a step ends many, the
            // one launch is one wave's work four times over and its pass loop costs the kernel 37 more spilled registers: LavaCrossing 52.1 ->
            // 56.1, Unlock 46.0 -> 49.4, RedBlueDoors 55.7 -> 59.1, MemoryS13 37.3 -> 42.9 (profiles/r04_levelgen_ring_families.txt).
            // MGX_LG_MERGE=0|1 overrides the family rule.
            h->lg_merge = lk == MGX_LEVEL_KEYCORRIDOR || lk == MGX_LEVEL_GOTODOOR || lk == MGX_LEVEL_FETCH || lk == MGX_LEVEL_PUTNEAR || lk == MGX_LEVEL_OBSTRUCTEDMAZE;
            if (const char *e = getenv("MGX_LG_MERGE")) h->lg_merge = atoi(e) != 0;
            h->lg_groups = R >= 8 ? 4 : 2; // (quarters: the join is due three quarters of a turn after the fork -- 50.9 / 31.2 us per step at 1 Mi / 512 Ki
                                           // LavaCrossing envs against 51.3 / 31.9 with halves, whose generators end about when their join is due)
            CREATE_TRY(hipMalloc((void **)&h->bank_d, (size_t)h->n_pad));
            CREATE_TRY(hipMemsetAsync(h->bank_d, 0, (size_t)h->n_pad, h->stream));
            h->lg_flags[0] = h->regen_d;
            for (int a = 1; a < R; a++) {
                CREATE_TRY(hipMalloc((void **)&h->lg_flags[a], (size_t)h->n_pad));
                CREATE_TRY(hipMemsetAsync(h->lg_flags[a], 0, (size_t)h->n_pad, h->stream));
            }
        }
    }
    CREATE_TRY(mgx_preload_step_kernels());
    {
        const StepParams sp = base_params(h);
        if (raise_lds) {
            hipError_t e2 = mgx_raise_lds_limit(sp, h->kernel_mode, h->wave_lds + 2 * h->lds_guard);
            if (e2 != hipSuccess) {
                int rc = mgx_fail(MGX_ERR_HIP, "mgx_create: cannot raise dynamic LDS to %d B: %s", h->wave_lds, hipGetErrorString(e2));
                mgx_destroy(h);
                return rc;
            }
        }
        CREATE_TRY(mgx_step_round_blocks(sp, h->kernel_mode, h->wpb, &h->round_blocks));
        CREATE_TRY(mgx_step_launch_cfg(device, &h->launch_cfg));
    }
    CREATE_TRY(mgx_preload_state_kernels());
    if (h->device_levels || h->one_level) CREATE_TRY(mgx_preload_levelgen_kernels());
    if (h->oh_nc >= 0 || h->flat) CREATE_TRY(mgx_preload_epilogue_kernels());
    if (h->dynobs) CREATE_TRY(mgx_preload_dynobs_kernels());
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    *out = h;
    return MGX_OK;
}

extern "C" int mgx_destroy(mgx_handle h)
{
    if (!h) return MGX_OK;
    DeviceGuard dev_guard;
    (void)dev_guard.enter_device(h->device, "mgx_destroy");
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->lg_stream) (void)hipStreamSynchronize(h->lg_stream);
    (void)hipFree(h->cells_d); (void)hipFree(h->cells0_d); (void)hipFree(h->agent_d); (void)hipFree(h->agent0_d);
    (void)hipFree(h->ctr_d); (void)hipFree(h->tri_d); (void)hipFree(h->mission_d); (void)hipFree(h->front_d); (void)hipFree(h->wcache_d);
    (void)hipFree(h->objaux_d); (void)hipFree(h->objaux0_d); (void)hipFree(h->objcont_d); (void)hipFree(h->objcont0_d); (void)hipFree(h->objcarry_d);
    (void)hipFree(h->mt_d); (void)hipFree(h->mt2_d); (void)hipFree(h->mt_idx_d); (void)hipFree(h->regen_d); (void)hipFree(h->mt_init_d);
    (void)hipFree(h->seed0_d); (void)hipFree(h->has_seed_d); (void)hipFree(h->reseeded_d);
    if (h->roll_exec) (void)hipGraphExecDestroy(h->roll_exec);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->lg_stream) (void)hipStreamDestroy(h->lg_stream);
    if (h->lg_fork) (void)hipEventDestroy(h->lg_fork);
    for (int a = 0; a < 4; a++) if (h->lg_join[a]) (void)hipEventDestroy(h->lg_join[a]);
    for (int a = 1; a < MGX_LG_RING_MAX; a++) (void)hipFree(h->lg_flags[a]);
    (void)hipFree(h->obst_d); (void)hipFree(h->obst0_d); (void)hipFree(h->act_d); (void)hipFree(h->restart_d); (void)hipFree(h->mt0_d); (void)hipFree(h->pos0_d); (void)hipFree(h->tape_d); (void)hipFree(h->tape0_d);
    (void)hipFree(h->bonus_action_d); (void)hipFree(h->bonus_state_d);
    (void)hipFree(h->sp0_d); (void)hipFree(h->bank_d); (void)hipFree(h->win_d); (void)hipFree(h->virt_d);
    for (auto &s : h->st_in) if (s.dev) (void)hipFree(s.dev);
    for (auto &s : h->st_out) if (s.dev) (void)hipFree(s.dev);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t ev : h->prof_ev) (void)hipEventDestroy(ev);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return MGX_OK;
}

extern "C" int mgx_set_stream(mgx_handle h, void *hip_stream)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_set_stream");
    if (rc) return rc;
    if ((rc = lg_drain(h))) return rc; // (generators beside the steps join the old stream first)
    HIP_TRY(hipStreamSynchronize(h->stream)); // hand-over point: everything enqueued so far is complete
    h->stream = (hipStream_t)hip_stream; // NULL is a real stream: the device's default (null) stream
    return MGX_OK;
}

extern "C" int mgx_use_own_stream(mgx_handle h)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_use_own_stream");
    if (rc) return rc;
    if ((rc = lg_drain(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->stream = h->own_stream;
    return MGX_OK;
}

extern "C" int mgx_obs_bytes(mgx_handle h, int64_t *per_env)
{
    if (!h || !per_env) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_obs_bytes: null argument");
    *per_env = h->obs_bytes;
    return MGX_OK;
}

extern "C" int mgx_sync(mgx_handle h)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_sync");
    if (rc) return rc;
    MgxCounters c;
    if ((rc = lg_drain(h))) return rc; // ("everything enqueued is complete" includes the generators running beside the caller's stream)
    rc = read_counters(h, &c);
    if (rc) return rc;
    if (c.invalid_actions > h->base_bad_act)
        return mgx_fail(MGX_ERR_INVALID_ACTION, "%llu env-steps were given an action >= %d (reference: AssertionError 'unknown action')",
                        c.invalid_actions - h->base_bad_act, h->cfg.extended_actions ? 9 : MGX_NUM_ACTIONS);
    if (c.out_of_bounds > h->base_oob)
        return mgx_fail(MGX_ERR_OUT_OF_BOUNDS, "%llu env-steps had a front/left/right cell outside the grid (reference: Grid.get assert) or hit the "
                                                "reference's strafe_right-onto-goal AttributeError (minigrid.py:1310)",
                        c.out_of_bounds - h->base_oob);
    return MGX_OK;
}

extern "C" int mgx_clear_faults(mgx_handle h)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_clear_faults");
    if (rc) return rc;
    MgxCounters c;
    rc = read_counters(h, &c);
    if (rc) return rc;
    h->base_bad_act = c.invalid_actions;
    h->base_oob = c.out_of_bounds;
    return MGX_OK;
}

extern "C" int mgx_get_stats(mgx_handle h, mgx_stats *out)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_get_stats");
    if (rc) return rc;
    if (!out) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_get_stats: null argument");
    MgxCounters c;
    rc = read_counters(h, &c);
    if (rc) return rc;
    out->steps = h->steps_total;
    unsigned long long ep = 0;
    double rs = 0.0;
    for (int i = 0; i < MGX_CTR_SHARDS; i++) { ep += c.shard[i].episodes; rs += c.shard[i].reward_sum; }
    out->episodes = (int64_t)ep;
    out->reward_sum = rs;
    out->invalid_actions = (int64_t)c.invalid_actions;
    out->out_of_bounds = (int64_t)c.out_of_bounds;
    return MGX_OK;
}

// Enqueue (no host sync) a copy of the running totals into caller DEVICE memory, e.g. two elements of a torch
// tensor that is then all-reduced over RCCL for logging: out[0] = episodes finished, out[1] = reward sum.
extern "C" int mgx_read_stats_async(mgx_handle h, double *out2_dev)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_read_stats_async");
    if (rc) return rc;
    if (!out2_dev || !is_device_ptr(out2_dev) || ((uintptr_t)out2_dev & 7))
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_read_stats_async: need an 8-byte aligned device pointer");
    HIP_TRY(mgx_launch_read_stats(h->ctr_d, out2_dev, h->stream));
    return MGX_OK;
}

// ------------------------------------------------------------------------------------------------ state I/O
static int set_state_impl(mgx_handle h, const uint8_t *grid, const uint8_t *aux, const int32_t *agent,
                          const uint8_t *carry, const int32_t *steps, const uint8_t *mask_host, bool bcast = false)
{
    if (!grid || !agent) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_set_state: grid and agent are required");
    const size_t n = (size_t)h->n, cells = (size_t)h->cells, n_src = bcast ? 1 : n; // bcast: grid / agent hold one env for all
    PackParams p;
    memset(&p, 0, sizeof p);
    const void *d;
    int rc;
    if ((rc = lg_drain(h))) return rc;
    h->snapshot_is_level = false;
    h->sched_K = 0; h->needs_full_reset = false; // (the injected state is the episode start from now on: a seed schedule ends here)
    if ((rc = forget_front(h))) return rc;
    p.bcast = bcast ? 1 : 0;
    if ((rc = dev_in(h, 0, grid, n_src * cells * 3, &d))) return rc;
    p.grid = (const uint8_t *)d;
    if ((rc = dev_in(h, 1, aux, n * cells, &d))) return rc;
    p.aux = (const uint8_t *)d;
    if ((rc = dev_in(h, 2, agent, n_src * 3 * sizeof(int32_t), &d, 4))) return rc;
    p.agent = (const int32_t *)d;
    if ((rc = dev_in(h, 3, carry, n * 3, &d))) return rc;
    p.carry = (const uint8_t *)d;
    if ((rc = dev_in(h, 4, steps, n * sizeof(int32_t), &d, 4))) return rc;
    p.steps = (const int32_t *)d;
    if ((rc = dev_in(h, 5, mask_host, n, &d))) return rc;
    p.mask = (const uint8_t *)d;
    p.cells = h->cells_d; p.cells0 = next_level_waiting(h) ? nullptr : h->cells0_d; p.rec = h->agent_d; p.rec0 = h->agent0_d;
    p.objaux = h->objaux_d; p.objaux0 = h->objaux0_d; p.objcont = h->objcont_d; p.objcont0 = h->objcont0_d; p.objcarry = h->objcarry_d;
    p.ctr = h->ctr_d;
    p.n = h->n; p.W = h->W; p.H = h->H; p.S = h->S; p.has_task = h->cfg.task_kind != MGX_TASK_NONE;
    MgxCounters before, after;
    if ((rc = read_counters(h, &before))) return rc;
    HIP_TRY(mgx_launch_pack(p, h->stream));
    if (h->has_seed_d) HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream)); // the snapshots no longer belong to seeds
    if ((rc = read_counters(h, &after))) return rc;
    // new_level_each_episode: the injected state is the CURRENT episode (the reference's env.grid / agent_pos assignments draw nothing); the next
    // level is the one already waiting in the buffer, drawn from the env's stream where its last reset left it.  Only a handle whose envs were
    // never all seeded has no such level yet: it is drawn here.
    if (h->stream_mode && !next_level_waiting(h)) {
        if (h->lg_ring) HIP_TRY(mgx_launch_ring_init(h->bank_d, h->regen_d, mask_host ? p.mask : nullptr, h->n, 1, h->stream));
        else if (mask_host) HIP_TRY(hipMemcpyAsync(h->regen_d, p.mask, n, hipMemcpyDeviceToDevice, h->stream));
        else HIP_TRY(hipMemsetAsync(h->regen_d, 1, n, h->stream));
        if ((rc = launch_levelgen(h))) return rc;
        if ((rc = lg_ring_init(h, mask_host ? p.mask : nullptr))) return rc;
    }
    if (after.invalid_state != before.invalid_state)
        return mgx_fail(MGX_ERR_INVALID_STATE, "mgx_set_state: input holds a cell/agent/carry value the reference cannot produce "
                                               "(type 1..9, color 0..6, state 0 or door 0..2, agent inside the grid, dir 0..3, carry key/ball/box)");
    return MGX_OK;
}

// (the handle moves between the sized step kernel of its grid and the run-time-size instance, which carries the bonus code: what mgx_create
// derived from the kernel function is derived again)
static int bonus_kernel_changed(mgx_handle h)
{
    const StepParams sp = base_params(h);
    if (h->wave_lds + 2 * h->lds_guard > 64 * 1024) HIP_TRY(mgx_raise_lds_limit(sp, h->kernel_mode, h->wave_lds + 2 * h->lds_guard));
    HIP_TRY(mgx_step_round_blocks(sp, h->kernel_mode, h->wpb, &h->round_blocks));
    return MGX_OK;
}

// ActionBonus / StateBonus (wrappers.py:87-153) as a property of the handle: the step kernels count and add (exploration_bonus, k_step.hip)
extern "C" int mgx_add_bonus(mgx_handle h, int32_t kind)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_add_bonus");
    if (rc) return rc;
    if (kind != 0 && kind != MGX_BONUS_ACTION && kind != MGX_BONUS_STATE)
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_add_bonus: kind %d is not 0, MGX_BONUS_ACTION or MGX_BONUS_STATE", kind);
    if (kind != 0 && h->oh_fused) return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_add_bonus: not with MGX_ONEHOT=fused (the two-kernel one-hot form carries it)");
    HIP_TRY(hipStreamSynchronize(h->stream)); // (steps in flight still count into the arrays)
    if (h->roll_exec) { (void)hipGraphExecDestroy(h->roll_exec); h->roll_exec = nullptr; } // (a captured rollout holds the old parameters)
    if (kind == 0) {
        (void)hipFree(h->bonus_action_d); (void)hipFree(h->bonus_state_d);
        h->bonus_action_d = nullptr; h->bonus_state_d = nullptr; h->bonus = 0;
        return bonus_kernel_changed(h);
    }
    if ((h->bonus & 15) == kind || ((h->bonus >> 4) & 15) == kind)
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_add_bonus: this handle already carries that wrapper");
    if (h->bonus >> 4) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_add_bonus: two wrappers are stacked already");
    uint32_t **arr = kind == MGX_BONUS_ACTION ? &h->bonus_action_d : &h->bonus_state_d;
    const size_t per_env = (size_t)h->W * h->H * (kind == MGX_BONUS_ACTION ? 4u * (h->cfg.extended_actions ? 9u : (unsigned)MGX_NUM_ACTIONS) : 1u);
    const hipError_t e = hipMalloc((void **)arr, (size_t)h->n_pad * per_env * sizeof(uint32_t));
    if (e != hipSuccess) {
        *arr = nullptr; (void)hipGetLastError();
        return mgx_fail(MGX_ERR_HIP, "mgx_add_bonus: %zu bytes of counts: %s", (size_t)h->n_pad * per_env * sizeof(uint32_t), hipGetErrorString(e));
    }
    h->bonus = h->bonus ? (h->bonus | (kind << 4)) : kind;
    // "every call zeroes the counts": the wrapper objects are new
    if (h->bonus_action_d) HIP_TRY(hipMemsetAsync(h->bonus_action_d, 0, (size_t)h->n_pad * h->W * h->H * 4u * (h->cfg.extended_actions ? 9u : (unsigned)MGX_NUM_ACTIONS) * sizeof(uint32_t), h->stream));
    if (h->bonus_state_d) HIP_TRY(hipMemsetAsync(h->bonus_state_d, 0, (size_t)h->n_pad * h->W * h->H * sizeof(uint32_t), h->stream));
    return bonus_kernel_changed(h);
}

// DACWrapper(env) (wrappers.py:35-84) as a property of the handle
extern "C" int mgx_set_dac(mgx_handle h, int32_t on)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_set_dac");
    if (rc) return rc;
    if (on && (h->oh_nc >= 0 || h->flat))
        return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_set_dac: the wrapper's last_obs is defined on the uint8 image (obs_mode MGX_OBS_PARTIAL or MGX_OBS_FULL)");
    if (on && h->dynobs) return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_set_dac: not for Dynamic-Obstacles handles");
    if (on && h->oh_fused) return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_set_dac: not with MGX_ONEHOT=fused");
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->roll_exec) { (void)hipGraphExecDestroy(h->roll_exec); h->roll_exec = nullptr; }
    h->dac = on != 0;
    return bonus_kernel_changed(h);
}

extern "C" int mgx_get_bonus_counts(mgx_handle h, int32_t kind, uint32_t *counts)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_get_bonus_counts");
    if (rc) return rc;
    const uint32_t *src = kind == MGX_BONUS_ACTION ? h->bonus_action_d : (kind == MGX_BONUS_STATE ? h->bonus_state_d : nullptr);
    if (!src || !counts) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_get_bonus_counts: the handle does not carry that wrapper (or counts is NULL)");
    const size_t bytes = (size_t)h->n * h->W * h->H * (kind == MGX_BONUS_ACTION ? 4u * (h->cfg.extended_actions ? 9u : (unsigned)MGX_NUM_ACTIONS) : 1u) * sizeof(uint32_t);
    HIP_TRY(hipMemcpyAsync(counts, src, bytes, hipMemcpyDefault, h->stream));
    if (!is_device_ptr(counts)) HIP_TRY(hipStreamSynchronize(h->stream));
    return MGX_OK;
}

extern "C" int mgx_set_state(mgx_handle h, const uint8_t *grid, const uint8_t *aux, const int32_t *agent,
                             const uint8_t *carry, const int32_t *steps)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_set_state");
    if (rc) return rc;
    if (h->dynobs)
        return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_set_state: a Dynamic-Obstacles state includes the env's RNG stream and the obstacle order; use mgx_reset(seeds, mask)");
    return set_state_impl(h, grid, aux, agent, carry, steps, nullptr);
}

extern "C" int mgx_get_state(mgx_handle h, uint8_t *grid, uint8_t *aux, int32_t *agent, uint8_t *carry, int32_t *steps)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_get_state");
    if (rc) return rc;
    const size_t n = (size_t)h->n, cells = (size_t)h->cells;
    // get_state is a debugging / checkpoint path: stage everything through fresh device buffers
    struct Tmp { void *dev = nullptr; void *user = nullptr; size_t bytes = 0; bool direct = false; } t[5];
    void *users[5] = {grid, aux, agent, carry, steps};
    size_t sizes[5] = {n * cells * 3, n * cells, n * 3 * sizeof(int32_t), n * 3, n * sizeof(int32_t)};
    for (int i = 0; i < 5; i++) {
        t[i].user = users[i]; t[i].bytes = sizes[i];
        if (!users[i]) continue;
        if (is_device_ptr(users[i])) { t[i].dev = users[i]; t[i].direct = true; }
        else {
            hipError_t e = hipMalloc(&t[i].dev, sizes[i]);
            if (e != hipSuccess) {
                for (int j = 0; j < i; j++) if (t[j].dev && !t[j].direct) (void)hipFree(t[j].dev);
                return mgx_fail(MGX_ERR_HIP, "mgx_get_state: hipMalloc(%zu): %s", sizes[i], hipGetErrorString(e));
            }
        }
    }
    PackParams p;
    memset(&p, 0, sizeof p);
    p.cells = h->cells_d; p.rec = h->agent_d; p.objaux = h->objaux_d;
    p.grid_out = (uint8_t *)t[0].dev; p.aux_out = (uint8_t *)t[1].dev; p.agent_out = (int32_t *)t[2].dev;
    p.carry_out = (uint8_t *)t[3].dev; p.steps_out = (int32_t *)t[4].dev;
    p.n = h->n; p.W = h->W; p.H = h->H; p.S = h->S; p.has_task = h->cfg.task_kind != MGX_TASK_NONE;
    hipError_t e = mgx_launch_unpack(p, h->stream);
    for (int i = 0; i < 5 && e == hipSuccess; i++)
        if (t[i].dev && !t[i].direct) e = hipMemcpyAsync(t[i].user, t[i].dev, t[i].bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    for (int i = 0; i < 5; i++) if (t[i].dev && !t[i].direct) (void)hipFree(t[i].dev);
    if (e != hipSuccess) return mgx_fail(MGX_ERR_HIP, "mgx_get_state: %s", hipGetErrorString(e));
    return MGX_OK;
}

// ------------------------------------------------------------------------------------------------ step / observe
static int run_step(mgx_handle h, bool do_step, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done, const uint8_t *obs_mask_dev = nullptr)
{
    if (h->needs_full_reset)
        return mgx_fail(MGX_ERR_INVALID_STATE, "%s: a seed schedule was installed; start every env on it with mgx_reset(h, NULL, NULL, obs) first", do_step ? "mgx_step" : "mgx_observe");
    StepParams p = base_params(h);
    p.do_step = do_step ? 1 : 0;
    p.obs_mask = do_step ? nullptr : obs_mask_dev;
    const void *d = nullptr;
    int rc;
    if (do_step) {
        if (!actions) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_step: actions is required");
        if ((rc = dev_in(h, 0, actions, (size_t)h->n, &d))) return rc;
        p.actions = (const uint8_t *)d;
        if (h->dynobs && !h->dyn_fused) { // the obstacle walk precedes the base step and folds the actions (envs/dynamicobstacles.py:60-80)
            DynObsParams dp = dynobs_params(h);
            dp.actions = p.actions;
            HIP_TRY(mgx_launch_dynobs(dp, h->stream));
            p.actions = h->act_d;
        }
    }
    OutArg o[3];
    if ((rc = dev_out(h, 0, obs, (size_t)h->n * h->obs_bytes, &o[0], 16))) return rc;
    if ((rc = dev_out(h, 1, reward, (size_t)h->n * sizeof(float), &o[1], 4))) return rc;
    if ((rc = dev_out(h, 2, done, (size_t)h->n, &o[2]))) return rc;
    p.obs = (uint8_t *)o[0].dev; p.reward = (float *)o[1].dev; p.done = (uint8_t *)o[2].dev;
    if (((h->oh_nc >= 0 && !h->oh_fused) || h->flat) && p.obs) p.obs = h->tri_d; // the simulator writes triples; the epilogue below expands them
    if (do_step && h->lg_ring) {
        // first step of a run of the ring: the generators forked three runs ago refilled the buffers consumed a whole turn of the ring ago -- the first
        // of which this step may need again -- and cleared this run's flag arrays (ahead of the profiling events: the wait is not the kernel)
        const int a = h->lg_s, half = h->lg_ring / h->lg_groups, grp = a / half;
        if (a == grp * half && h->lg_unjoined[grp]) { HIP_TRY(hipStreamWaitEvent(h->stream, h->lg_join[grp], 0)); h->lg_unjoined[grp] = false; }
    }
    // profiling: this launch alone between its own two events (not while a graph is being captured)
    const bool sample = do_step && h->profiling && !h->assume_device && h->prof_samples < MGX_PROF_MAX_SAMPLES &&
                        (h->prof_launches % h->prof_stride) == 0;
    if (sample) {
        while ((int)h->prof_ev.size() < 2 * (h->prof_samples + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            h->prof_ev.push_back(ev);
        }
        HIP_TRY(hipEventRecord(h->prof_ev[2 * h->prof_samples], h->stream));
    }
    if (do_step && h->lg_ring) {
        const int a = h->lg_s, half = h->lg_ring / h->lg_groups, grp = a / half;
        p.regen = h->lg_flags[a];
        HIP_TRY(mgx_launch_step(p, h->kernel_mode, h->wpb, h->launch_cfg, h->stream));
        h->lg_dirty[a] = true;
        if (a == grp * half + half - 1) { // last step of a run: its flags, in step order (the env's RNG stream), beside the steps of the other runs
            HIP_TRY(hipEventRecord(h->lg_fork, h->stream));
            HIP_TRY(hipStreamWaitEvent(h->lg_stream, h->lg_fork, 0));
            LevelGenParams g = levelgen_params(h);
            if (h->lg_merge) { // one launch for the run's arrays (half <= 4)
                g.regen = h->lg_flags[grp * half];
                g.n_regen = half;
                for (int b = grp * half + 1; b <= a; b++) g.regen_more[b - grp * half - 1] = h->lg_flags[b];
                HIP_TRY(mgx_launch_levelgen(g, h->lg_stream));
            } else {
                for (int b = grp * half; b <= a; b++) {
                    g.regen = h->lg_flags[b];
                    HIP_TRY(mgx_launch_levelgen(g, h->lg_stream));
                }
            }
            for (int b = grp * half; b <= a; b++) h->lg_dirty[b] = false;
            HIP_TRY(hipEventRecord(h->lg_join[grp], h->lg_stream));
            h->lg_unjoined[grp] = true;
        }
        h->lg_s = (a + 1) & (h->lg_ring - 1);
    } else if (do_step && h->dyn_fused) { // Dynamic-Obstacles, staged partial form: walk + step in ONE launch on the staged tile
        DynObsParams dp = dynobs_params(h);
        dp.actions = p.actions;
        dp.front = nullptr;
        StepParams pf = p;
        pf.lds_guard = h->staged_guard; // (a handle whose single step gathers: this kernel stages)
        if (h->kernel_mode == 3) { const int img = 32 * h->view * h->view * 3; pf.wave_lds = ((64 * h->LS > img ? 64 * h->LS : img) + 15) & ~15; pf.front = nullptr; }
        HIP_TRY(mgx_launch_step_dyn(pf, dp, h->launch_cfg, h->stream));
    } else HIP_TRY(mgx_launch_step(p, h->kernel_mode, h->wpb, h->launch_cfg, h->stream));
    if (sample) {
        HIP_TRY(hipEventRecord(h->prof_ev[2 * h->prof_samples + 1], h->stream));
        h->prof_samples++;
    }
    if (h->dac && o[0].dev) HIP_TRY(mgx_launch_dac_obs(h->agent_d, (uint8_t *)o[0].dev, h->n, h->obs_bytes, h->stream));
    if (h->oh_nc >= 0 && !h->oh_fused && o[0].dev)
        HIP_TRY(mgx_launch_onehot(h->tri_d, (uint8_t *)o[0].dev, h->n * (h->tri_bytes / 3), h->oh_nc, h->oh_ns, h->stream));
    if (h->flat && o[0].dev)
        HIP_TRY(mgx_launch_flat(h->tri_d, h->agent_d, h->mission_d, (float *)o[0].dev, h->n, (int)h->tri_bytes, mission_family(&h->cfg), h->stream));
    if (do_step) {
        h->steps_total += h->n;
        if (h->profiling) h->prof_launches++;
        if (h->stream_mode && !h->lg_ring && (rc = launch_levelgen(h))) return rc; // refill the next-level buffers this step consumed
    }
    return finish_out(h, o, 3);
}

extern "C" int mgx_step(mgx_handle h, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_step");
    if (rc) return rc;
    return run_step(h, true, actions, obs, reward, done);
}

extern "C" int mgx_observe(mgx_handle h, uint8_t *obs)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_observe");
    if (rc) return rc;
    if (!obs) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_observe: obs is required");
    return run_step(h, false, nullptr, obs, nullptr, nullptr);
}

// T consecutive mgx_step calls in one host call: step t reads actions[t][N] and writes obs[t], reward[t], done[t].
// The T x (k_dynobs, k_step, epilogue, k_levelgen) launches are captured into a hipGraph the first time and replayed
// afterwards (same T and buffers), so small batches are not bound by one host launch per kernel.
extern "C" int mgx_rollout(mgx_handle h, int64_t T, const uint8_t *actions, uint8_t *obs, float *reward, uint8_t *done)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_rollout");
    if (rc) return rc;
    if (T <= 0 || !actions) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_rollout: T > 0 and actions are required");
    if (h->needs_full_reset)
        return mgx_fail(MGX_ERR_INVALID_STATE, "mgx_rollout: a seed schedule was installed; start every env on it with mgx_reset(h, NULL, NULL, obs) first");
    const void *args[4] = {actions, obs, reward, done};
    for (const void *a : args)
        if (a && !is_device_ptr(a)) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_rollout: buffers must be device memory (host buffers: call mgx_step per step)");
    if (obs && (((uintptr_t)obs | (uintptr_t)((size_t)h->n * h->obs_bytes)) & 15u))
        return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_rollout: obs slices must stay 16-byte aligned (n_envs * obs_bytes = %lld)", (long long)(h->n * h->obs_bytes));
    if ((reward && ((uintptr_t)reward & 3u))) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_rollout: reward is not 4-byte aligned");
    // Fused form: one launch for all T steps, the tile resident in LDS (k_rollout).  Handles whose steps interleave other kernels
    // (k_levelgen, k_dynobs, one-hot / flat epilogues), other views / visibility / hidden object state and grids without a
    // sized instance take the captured graph of per-step launches below.  MGX_ROLLOUT=graph forces that form (tests, tuning).
    const char *rf = getenv("MGX_ROLLOUT");
    // (grids whose single step takes the gather form: fused while the tile is no larger than 16x16 -- sized k_rollout instances exist for
    // 13x13 and 16x16 -- and the graph beyond: measured at 262,144 envs, T = 64, the run-time-size k_rollout runs FourRooms 19x19 at
    // 20.7 us per step against 19.6 for the graph of gather steps and MultiRoom 25x25 at 27.5 against 21.8: a 23-40 KB tile image
    // leaves one or two waves per block)
    // (FullyObs handles, round 3: the same kernel with emit_full_obs on the resident tile, grids up to 13x13 -- us per step at 1 Mi envs, graph of
    // direct-form steps -> fused: DoorKey-8x8 54.6 -> 34.7, LavaCrossingS9N1 64.4 -> 50.0, S11N5 105 -> 82, RedBlueDoors-8x8 95 -> 65, MemoryS13
    // 151 -> 138; at 16x16 the 19.7 KB of LDS per wave leave six waves per CU and the direct form wins, 44 against 51 at 262,144 envs)
    const bool fused_ok = (h->partial ? (h->kernel_mode == 0 || (h->kernel_mode == 3 && h->S <= 256)) : h->S <= 192) &&
                          !h->cfg.alt_visibility && !h->objaux_d && !h->stream_mode &&
                          !h->dynobs && h->oh_nc < 0 && !h->flat && !h->sched_K && !h->bonus && !h->dac && !(rf && !strcmp(rf, "graph"));
    if (fused_ok) {
        StepParams p = base_params(h);
        p.do_step = 1;
        p.lds_guard = h->staged_guard; // (k_rollout stages the tile in LDS even where the single step gathers: sized grids up to 16x16)
        if (h->front_d) { int rc2 = forget_front(h); if (rc2) return rc2; } // k_rollout moves agents and cells without keeping the gather form's "cell in front"
        const hipError_t e = mgx_launch_rollout(p, actions, obs, reward, done, T, h->partial ? 0 : 1, h->stream);
        if (e == hipSuccess) {
            h->steps_total += T * h->n;
            if (h->profiling) h->prof_launches += T;
            return MGX_OK;
        }
        if (e != hipErrorNotSupported) return mgx_fail(MGX_ERR_HIP, "mgx_rollout: k_rollout launch failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
    }
    if ((rc = lg_drain(h))) return rc; // (the captured steps start from "no flags waiting", and end there: the graph drains behind its last step)
    const bool cached = h->roll_exec && h->roll_T == T && !memcmp(h->roll_args, args, sizeof args);
    if (!cached) {
        if (h->roll_exec) { (void)hipGraphExecDestroy(h->roll_exec); h->roll_exec = nullptr; }
        const int64_t steps_before = h->steps_total, launches_before = h->prof_launches;
        if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
        hipStream_t user_stream = h->stream;
        HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        h->stream = h->cap_stream; // run_step enqueues on h->stream: recorded, not executed
        h->assume_device = true;
        for (int64_t t = 0; t < T && !rc; t++)
            rc = run_step(h, true, actions + t * h->n, obs ? obs + t * h->n * h->obs_bytes : nullptr,
                          reward ? reward + t * h->n : nullptr, done ? done + t * h->n : nullptr);
        if (!rc) rc = lg_drain(h);
        h->assume_device = false;
        h->stream = user_stream;
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
        h->steps_total = steps_before; h->prof_launches = launches_before; // nothing ran yet
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return mgx_fail(MGX_ERR_HIP, "mgx_rollout: stream capture failed: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&h->roll_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { h->roll_exec = nullptr; return mgx_fail(MGX_ERR_HIP, "mgx_rollout: hipGraphInstantiate: %s", hipGetErrorString(e)); }
        h->roll_T = T;
        memcpy(h->roll_args, args, sizeof args);
    }
    HIP_TRY(hipGraphLaunch(h->roll_exec, h->stream));
    h->steps_total += T * h->n;
    if (h->profiling) h->prof_launches += T;
    return MGX_OK;
}

static int objstate_io(mgx_handle h, const char *fn, const uint8_t *ci, const uint8_t *ai, const uint8_t *cci, uint8_t *co, uint8_t *ao, uint8_t *cco)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, fn);
    if (rc) return rc;
    if (!h->objaux_d) return mgx_fail(MGX_ERR_INVALID_ARG, "%s: the handle was not created with object_state = 1", fn);
    const size_t n = (size_t)h->n, cells = (size_t)h->cells;
    ObjStateParams p;
    memset(&p, 0, sizeof p);
    const void *d;
    if ((rc = dev_in(h, 0, ci, n * cells * 3, &d))) return rc;
    p.contains_in = (const uint8_t *)d;
    if ((rc = dev_in(h, 1, ai, n, &d))) return rc;
    p.carry_aux_in = (const uint8_t *)d;
    if ((rc = dev_in(h, 2, cci, n * 3, &d))) return rc;
    p.carry_contains_in = (const uint8_t *)d;
    OutArg o[3];
    if ((rc = dev_out(h, 0, co, n * cells * 3, &o[0]))) return rc;
    if ((rc = dev_out(h, 1, ao, n, &o[1]))) return rc;
    if ((rc = dev_out(h, 2, cco, n * 3, &o[2]))) return rc;
    p.contains_out = (uint8_t *)o[0].dev; p.carry_aux_out = (uint8_t *)o[1].dev; p.carry_contains_out = (uint8_t *)o[2].dev;
    p.objcont = h->objcont_d; p.objcont0 = next_level_waiting(h) ? nullptr : h->objcont0_d; p.objcarry = h->objcarry_d; p.ctr = h->ctr_d;
    p.n = h->n; p.W = h->W; p.H = h->H; p.S = h->S;
    MgxCounters before, after;
    if ((rc = read_counters(h, &before))) return rc;
    HIP_TRY(mgx_launch_objstate(p, h->stream));
    if ((rc = read_counters(h, &after))) return rc;
    if (after.invalid_state != before.invalid_state)
        return mgx_fail(MGX_ERR_INVALID_STATE, "%s: a contained object is not a (type, color, state) the reference can produce", fn);
    return finish_out(h, o, 3);
}

extern "C" int mgx_set_object_state(mgx_handle h, const uint8_t *contains, const uint8_t *carry_aux, const uint8_t *carry_contains)
{
    int rc = objstate_io(h, "mgx_set_object_state", contains, carry_aux, carry_contains, nullptr, nullptr, nullptr);
    if (rc || !contains) return rc;
    // the snapshot's contains plane is no longer the generated level's: the next mgx_reset regenerates instead of restoring
    DeviceGuard dev_guard;
    if ((rc = dev_guard.enter(h, "mgx_set_object_state"))) return rc;
    h->snapshot_is_level = false;
    if (h->has_seed_d) HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream));
    return MGX_OK;
}

extern "C" int mgx_get_object_state(mgx_handle h, uint8_t *contains, uint8_t *carry_aux, uint8_t *carry_contains)
{
    return objstate_io(h, "mgx_get_object_state", nullptr, nullptr, nullptr, contains, carry_aux, carry_contains);
}

// (mask: only those envs -- the word of TwoGoals is a running count that an unmasked env must keep)
static int set_task_impl(mgx_handle h, const uint32_t *task, const uint8_t *mask)
{
    int rc;
    const void *d, *dm;
    if ((rc = dev_in(h, 4, task, (size_t)h->n * sizeof(uint32_t), &d, 4))) return rc;
    if ((rc = dev_in(h, 5, mask, (size_t)h->n, &dm))) return rc;
    HIP_TRY(mgx_launch_task(h->agent_d, next_level_waiting(h) ? nullptr : h->agent0_d, (const uint32_t *)d, nullptr, (const uint8_t *)dm, h->n, h->stream));
    return MGX_OK;
}

extern "C" int mgx_set_task(mgx_handle h, const uint32_t *task)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_set_task");
    if (rc) return rc;
    if (!task) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_set_task: null argument");
    if (h->cfg.task_kind == MGX_TASK_NONE) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_set_task: this handle has no task rule");
    if ((rc = set_task_impl(h, task, nullptr))) return rc;
    h->snapshot_is_level = false; // the snapshot's task word changed
    if (h->has_seed_d) HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream)); // the snapshot's task word changed
    return MGX_OK;
}

extern "C" int mgx_get_task(mgx_handle h, uint32_t *task)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_get_task");
    if (rc) return rc;
    if (!task) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_get_task: null argument");
    if (h->cfg.task_kind == MGX_TASK_NONE) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_get_task: this handle has no task rule");
    OutArg o;
    if ((rc = dev_out(h, 3, task, (size_t)h->n * sizeof(uint32_t), &o, 4))) return rc;
    HIP_TRY(mgx_launch_task(h->agent_d, h->agent0_d, nullptr, (uint32_t *)o.dev, nullptr, h->n, h->stream));
    return finish_out(h, &o, 1);
}

extern "C" int mgx_get_direction(mgx_handle h, uint8_t *direction)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_get_direction");
    if (rc) return rc;
    if (!direction) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_get_direction: null argument");
    OutArg o;
    if ((rc = dev_out(h, 3, direction, (size_t)h->n, &o))) return rc;
    HIP_TRY(mgx_launch_direction(h->agent_d, (uint8_t *)o.dev, h->n, h->stream));
    return finish_out(h, &o, 1);
}

extern "C" int mgx_get_pose(mgx_handle h, int32_t *pose)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_get_pose");
    if (rc) return rc;
    if (!pose) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_get_pose: null argument");
    OutArg o;
    if ((rc = dev_out(h, 3, pose, (size_t)h->n * 3 * sizeof(int32_t), &o, 4))) return rc;
    HIP_TRY(mgx_launch_pose(h->agent_d, (int32_t *)o.dev, h->n, h->stream));
    return finish_out(h, &o, 1);
}

namespace {

ConsumeParams consume_params(mgx_handle h, const uint8_t *mask_dev)
{
    ConsumeParams c;
    memset(&c, 0, sizeof c);
    c.mask = mask_dev;
    c.cells = h->cells_d; c.cells0 = h->cells0_d; c.agent = h->agent_d; c.agent0 = h->agent0_d; c.regen = h->regen_d;
    c.objaux = h->objaux_d; c.objaux0 = h->objaux0_d; c.objcont = h->objcont_d; c.objcont0 = h->objcont0_d; c.objcarry = h->objcarry_d;
    c.front = h->front_d; // (the reset envs' "cell in front" is unknown until their next observation pass)
    c.wcache = h->wcache_d;
    c.n = h->n; c.S = h->S; c.flag_regen = 0;
    return c;
}

// The snapshot arrays for K episode starts per env (bank-major).  Contents are lost: the caller regenerates every bank.
int resize_snapshots(mgx_handle h, int K)
{
    if (K == h->snap_banks) return MGX_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->n_pad * (size_t)K;
    struct Arr { void **pp; size_t bytes; int fill; };
    const Arr arrs[] = {{(void **)&h->cells0_d, n * h->S, 0}, {(void **)&h->agent0_d, n * sizeof(uint2), 0},
                        {(void **)&h->objaux0_d, h->objaux_d ? n * h->S : 0, 0}, {(void **)&h->objcont0_d, h->objaux_d ? n * h->S : 0, MGX_CODE_EMPTY},
                        {(void **)&h->obst0_d, h->dynobs ? n * 8 : 0, 0}, {(void **)&h->mt0_d, h->dynobs ? n * 624 * sizeof(uint32_t) : 0, 0},
                        {(void **)&h->pos0_d, h->dynobs ? n * sizeof(uint32_t) : 0, 0}, {(void **)&h->tape0_d, h->dynobs ? n * MGX_DYN_TAPE_DW * sizeof(uint32_t) : 0, 0},
                        {(void **)&h->sp0_d, h->dynobs ? n * sizeof(uint32_t) : 0, 0}};
    // (new arrays first, the old ones go only when every allocation succeeded: a failed resize leaves the handle as it was)
    void *fresh[sizeof arrs / sizeof arrs[0]] = {};
    for (size_t i = 0; i < sizeof arrs / sizeof arrs[0]; i++) {
        if (!arrs[i].bytes) continue;
        const hipError_t e = hipMalloc(&fresh[i], arrs[i].bytes);
        if (e != hipSuccess) {
            for (size_t j = 0; j < i; j++) if (fresh[j]) (void)hipFree(fresh[j]);
            (void)hipGetLastError();
            return mgx_fail(MGX_ERR_HIP, "mgx_set_seed_schedule: %d episode-start snapshots per env need %zu more bytes: %s", K, arrs[i].bytes, hipGetErrorString(e));
        }
    }
    for (size_t i = 0; i < sizeof arrs / sizeof arrs[0]; i++) {
        if (!arrs[i].bytes) continue;
        (void)hipFree(*arrs[i].pp);
        *arrs[i].pp = fresh[i];
        HIP_TRY(hipMemsetAsync(fresh[i], arrs[i].fill, arrs[i].bytes, h->stream));
    }
    h->snap_banks = K;
    h->snapshot_is_level = false;
    if (h->has_seed_d) HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream));
    if (h->roll_exec) { (void)hipGraphExecDestroy(h->roll_exec); h->roll_exec = nullptr; } // (a captured rollout holds the old pointers)
    return MGX_OK;
}

} // namespace

// ReseedWrapper(env_i, seeds = seeds[i][0..K-1], seed_idx = idx0) (wrappers.py:12-28): the K levels of every env generated once into K
// resident episode-start snapshots; resets then copy from the bank the env's list index names.
extern "C" int mgx_set_seed_schedule(mgx_handle h, const uint64_t *seeds, int32_t K, int32_t idx0)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_set_seed_schedule");
    if (rc) return rc;
    if (K == 0) { // remove the schedule: the current episode-start snapshot (bank 0 is NOT it in general) is undefined until the next seeded reset
        if (h->sched_K) {
            h->sched_K = 0; h->needs_full_reset = false;
            if (h->has_seed_d) HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream));
            if (h->roll_exec) { (void)hipGraphExecDestroy(h->roll_exec); h->roll_exec = nullptr; }
        }
        return MGX_OK;
    }
    if (K < 1 || K > 255) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_set_seed_schedule: K = %d outside 0..255", K);
    if (idx0 < 0 || idx0 >= K) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_set_seed_schedule: seed_idx %d outside 0..%d (the wrapper would raise IndexError)", idx0, K - 1);
    if (!seeds) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_set_seed_schedule: seeds is required");
    if (h->cfg.level_kind == MGX_LEVEL_NONE)
        return mgx_fail(MGX_ERR_NO_LEVELGEN, "mgx_set_seed_schedule: this handle has no built-in level generator; use mgx_set_state");
    if (h->cfg.new_level_each_episode)
        return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_set_seed_schedule: a new_level_each_episode handle is the env WITHOUT ReseedWrapper; create it with new_level_each_episode = 0");
    if (h->one_level) return MGX_OK; // every seed gives the family's one level: mgx_reset(h, NULL, ...) is a restore with or without the wrapper
    if (!h->device_levels)
        return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_set_seed_schedule: grids beyond 64x64 cells generate their levels on the host (mgx_reset with seeds)");
    const size_t n = (size_t)h->n;
    const void *ds = nullptr;
    if ((rc = dev_in(h, 4, seeds, n * (size_t)K * sizeof(uint64_t), &ds, 8))) return rc;
    if (!h->bank_d) {
        HIP_TRY(hipMalloc((void **)&h->bank_d, (size_t)h->n_pad));
    }
    h->sched_K = 0;
    if ((rc = resize_snapshots(h, K))) return rc;
    if ((rc = forget_front(h))) return rc;
    uint64_t *col = nullptr;
    HIP_TRY(hipMalloc((void **)&col, n * sizeof(uint64_t)));
    hipError_t e = hipSuccess;
    for (int b = 0; b < K && e == hipSuccess && !rc; b++) {
        const size_t off = (size_t)b * (size_t)h->n_pad;
        e = mgx_launch_seed_column((const uint64_t *)ds, K, b, col, h->n, h->stream);
        if (e == hipSuccess) e = launch_seed(h, col, nullptr, 0);
        if (e != hipSuccess) break;
        LevelGenParams g = levelgen_params(h);
        g.cells0 += off * h->S; g.agent0 += off;
        if (g.objaux0) { g.objaux0 += off * h->S; g.objcont0 += off * h->S; }
        e = mgx_launch_levelgen(g, h->stream);
        if (e == hipSuccess && h->dynobs) { // obstacle order + the RNG state right behind this seed's reset(), into bank b
            DynObsParams dp = dynobs_params(h);
            dp.cells0 += off * h->S; dp.obst0 += off * 8; dp.mt0 += off * 624; dp.pos0 += off; dp.tape0 += off * MGX_DYN_TAPE_DW; dp.sp0 += off;
            dp.bank = nullptr; dp.bank_envs = 0; // (the offsets above select the bank)
            dp.mask = nullptr; dp.mask_reset = nullptr;
            e = mgx_launch_dynobs_init(dp, h->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(col);
    if (e != hipSuccess) return mgx_fail(MGX_ERR_HIP, "mgx_set_seed_schedule: %s", hipGetErrorString(e));
    // the first reset takes entry idx0: the list index of the "current" episode is the one before it
    HIP_TRY(hipMemsetAsync(h->bank_d, (idx0 + K - 1) % K, (size_t)h->n_pad, h->stream));
    HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream)); // (k_seed's bookkeeping describes the last bank only)
    h->sched_K = K;
    h->needs_full_reset = true;
    h->seeded = true;
    return MGX_OK;
}

extern "C" int mgx_reset(mgx_handle h, const uint64_t *seeds, const uint8_t *mask, uint8_t *obs)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_reset");
    if (rc) return rc;
    if (h->cfg.level_kind == MGX_LEVEL_NONE)
        return mgx_fail(MGX_ERR_NO_LEVELGEN, "mgx_reset: this handle has no built-in level generator; use mgx_set_state");
    const size_t n = (size_t)h->n, cells = (size_t)h->cells;
    if ((rc = lg_drain(h))) return rc;
    if (seeds && h->sched_K) { // explicit seeds define the episode start themselves: the schedule ends (include/mgx.h)
        h->sched_K = 0; h->needs_full_reset = false;
        if (h->has_seed_d) HIP_TRY(hipMemsetAsync(h->has_seed_d, 0, (size_t)h->n_pad, h->stream));
        if (h->roll_exec) { (void)hipGraphExecDestroy(h->roll_exec); h->roll_exec = nullptr; }
    }
    if (!seeds && h->sched_K) {
        // ReseedWrapper.reset() (wrappers.py:24-28): every masked env moves on to the next seed of its list; the level -- and for
        // Dynamic-Obstacles the RNG state behind it -- is the resident snapshot of that list entry
        if (h->needs_full_reset && mask)
            return mgx_fail(MGX_ERR_INVALID_STATE, "mgx_reset: the first reset after mgx_set_seed_schedule must cover every env (mask = NULL)");
        const void *dm = nullptr;
        if ((rc = dev_in(h, 5, mask, n, &dm))) return rc;
        HIP_TRY(mgx_launch_bank_advance(h->bank_d, (const uint8_t *)dm, h->sched_K, h->n, h->stream));
        ConsumeParams c = consume_params(h, (const uint8_t *)dm);
        c.regen = nullptr;
        c.bank = h->bank_d; c.bank_envs = h->n_pad;
        c.restart = h->dynobs ? h->restart_d : nullptr; // the next k_dynobs restores obstacle order, RNG position, block and tape from the bank
        HIP_TRY(mgx_launch_consume(c, h->stream));
        h->needs_full_reset = false;
        if (obs) return run_step(h, false, nullptr, obs, nullptr, nullptr, (dm && is_device_ptr(obs)) ? (const uint8_t *)dm : nullptr);
        return MGX_OK;
    }
    if (!seeds && !h->one_level) {
        // The plain reset() (minigrid.py:831-858; the caller loop of run_tests.py:64-66): the env's RNG stream continues and the next
        // level is drawn from it, on the GPU.
        if (!h->device_levels)
            return mgx_fail(MGX_ERR_UNSUPPORTED, "mgx_reset: grids beyond 64x64 cells keep no RNG stream on the device; pass seeds");
        if (!h->seeded)
            return mgx_fail(MGX_ERR_INVALID_STATE, "mgx_reset: reset() without seeds continues the env's RNG stream, which starts with a seeded reset "
                                                   "(MiniGridEnv.__init__ calls seed(1337), minigrid.py:824-826): call mgx_reset with seeds for every env first");
        const void *dm = nullptr;
        if ((rc = dev_in(h, 5, mask, n, &dm))) return rc;
        ConsumeParams c = consume_params(h, (const uint8_t *)dm);
        if (h->stream_mode) { // the next level is waiting in the buffer: make it current, refill behind it
            c.flag_regen = 1;
            if (h->lg_ring) { c.regen = nullptr; c.bank = h->bank_d; c.bank_envs = h->n_pad; } // (the buffer the env's ring position names)
            HIP_TRY(mgx_launch_consume(c, h->stream));
            if (h->lg_ring) HIP_TRY(mgx_launch_ring_consumed(h->bank_d, h->regen_d, (const uint8_t *)dm, h->n, h->lg_ring, h->stream));
            if ((rc = launch_levelgen(h))) return rc;
        } else {
            if (h->dynobs) { // the obstacle walks' place in the stream back into (block, stream position)
                DynObsParams dp = dynobs_params(h);
                dp.mask_reset = (const uint8_t *)dm;
                HIP_TRY(mgx_launch_dynobs_handover(dp, h->stream));
            }
            if (h->maybe_virtual) { // the stream goes on past the first level: the masked envs that still hold seed + first words get their block
                HIP_TRY(mgx_launch_seed_materialize((const uint8_t *)dm, h->mt_init_d, h->mt_d, h->virt_d, h->seed0_d, h->n, h->stream));
                if (!mask) h->maybe_virtual = false;
            }
            HIP_TRY(mgx_launch_mark_plain_reset((const uint8_t *)dm, h->regen_d, h->has_seed_d, h->reseeded_d, h->n, h->stream));
            if ((rc = launch_levelgen(h))) return rc;
            HIP_TRY(mgx_launch_consume(c, h->stream));
            if (h->dynobs) {
                DynObsParams dp = dynobs_params(h);
                dp.mask = h->reseeded_d;
                dp.mask_reset = (const uint8_t *)dm;
                HIP_TRY(mgx_launch_dynobs_init(dp, h->stream));
            }
        }
        if (obs) return run_step(h, false, nullptr, obs, nullptr, nullptr, (dm && is_device_ptr(obs)) ? (const uint8_t *)dm : nullptr);
        return MGX_OK;
    }
    std::vector<uint64_t> zero_seeds;
    if (!seeds && !(h->snapshot_is_level || !mask)) { // (a one-level family's first reset under a mask, without seeds: any seed gives its level)
        zero_seeds.assign(n, 0);
        seeds = zero_seeds.data();
    }
    if (h->device_levels) {
        // Everything on the GPU: env.seed(s_i) (k_seed: SHA-512 key + MT19937 init_by_array per env), level 1 into the
        // next-level buffer (k_levelgen), make it current (k_consume); with new_level_each_episode level 2 is
        // generated behind it, otherwise the buffer keeps level 1 as the episode-start snapshot.
        const void *ds = nullptr, *dm = nullptr;
        if ((rc = dev_in(h, 4, seeds, n * sizeof(uint64_t), &ds, 8))) return rc;
        if ((rc = dev_in(h, 5, mask, n, &dm))) return rc;
        // (k_seed clears the regeneration flags of the envs being reset before it raises those of the envs it really seeds)
        // (an env that keeps its seed keeps its level: with the snapshot still holding it -- not in stream mode, where the
        // buffer holds the NEXT level -- seeding and generation are skipped for it and k_consume alone restores it)
        HIP_TRY(launch_seed(h, (const uint64_t *)ds, (const uint8_t *)dm, h->stream_mode ? 0 : 1));
        if ((rc = launch_levelgen(h))) return rc;
        ConsumeParams c = consume_params(h, (const uint8_t *)dm);
        c.flag_regen = h->stream_mode ? 1 : 0;
        HIP_TRY(mgx_launch_consume(c, h->stream));
        if (h->stream_mode && (rc = launch_levelgen(h))) return rc;
        if ((rc = lg_ring_init(h, (const uint8_t *)dm))) return rc;
        if (!mask) h->seeded = true; // every env's RNG stream exists from here on: reset() without seeds may continue it
        if (h->dynobs) { // obstacle order out of the generator's markers + snapshot of the RNG right after reset()
            DynObsParams dp = dynobs_params(h);
            dp.mask = h->reseeded_d;
            dp.mask_reset = (const uint8_t *)dm;
            HIP_TRY(mgx_launch_dynobs_init(dp, h->stream));
        }
        // (with a mask only the 64-env tiles that hold a reset env are re-observed; the rest of `obs` is left as it is)
        if (obs) return run_step(h, false, nullptr, obs, nullptr, nullptr, is_device_ptr(obs) ? (const uint8_t *)dm : nullptr);
        return MGX_OK;
    }
    if (h->one_level && (h->snapshot_is_level || !mask)) {
        // One level for every seed.  The first full reset generates it once on the host and k_pack_state hands it to
        // every env (the first version generated and uploaded N copies: 300 ms per reset at 1 Mi envs, during which
        // the GPU went idle); from then on a reset -- full or masked -- is a restore of the snapshot on the device.
        const void *dm = nullptr;
        if (!h->snapshot_is_level) {
            std::vector<uint8_t> g1(cells * 3);
            int32_t a1[3];
            uint32_t t1 = 0;
            const uint64_t s0 = 0;
            if ((rc = mgx_generate_levels_ex(&h->cfg, 1, &s0, g1.data(), a1, h->cfg.task_kind != MGX_TASK_NONE ? &t1 : nullptr))) return rc;
            if ((rc = set_state_impl(h, g1.data(), nullptr, a1, nullptr, nullptr, nullptr, true))) return rc;
            if (h->cfg.task_kind != MGX_TASK_NONE) {
                std::vector<uint32_t> tw(n, t1);
                if ((rc = set_task_impl(h, tw.data(), nullptr))) return rc;
                HIP_TRY(hipStreamSynchronize(h->stream)); // `tw` is about to go away
            }
            HIP_TRY(hipStreamSynchronize(h->stream)); // so are `g1` / `a1`
            h->snapshot_is_level = true;
        } else {
            if ((rc = dev_in(h, 5, mask, n, &dm))) return rc;
            ConsumeParams c = consume_params(h, (const uint8_t *)dm);
            c.regen = nullptr;
            HIP_TRY(mgx_launch_consume(c, h->stream));
        }
        if (obs) return run_step(h, false, nullptr, obs, nullptr, nullptr, (dm && is_device_ptr(obs)) ? (const uint8_t *)dm : nullptr);
        return MGX_OK;
    }
    // per-seed levels generated on the host (grids beyond 64x64, or the first reset of a one-level family under a mask)
    if (!seeds) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_reset: seeds is required here");
    std::vector<uint64_t> seeds_host;
    if (is_device_ptr(seeds)) { // the host generators read them
        seeds_host.resize(n);
        HIP_TRY(hipMemcpyAsync(seeds_host.data(), seeds, n * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        seeds = seeds_host.data();
    }
    std::vector<uint8_t> grid(n * cells * 3);
    std::vector<int32_t> agent(n * 3);
    // (generation for unmasked envs is wasted work but keeps the code simple; k_pack_state ignores them)
    std::vector<uint32_t> task(h->cfg.task_kind != MGX_TASK_NONE ? n : 0);
    rc = mgx_generate_levels_ex(&h->cfg, h->n, seeds, grid.data(), agent.data(), task.empty() ? nullptr : task.data());
    if (rc) return rc;
    rc = set_state_impl(h, grid.data(), nullptr, agent.data(), nullptr, nullptr, mask);
    if (rc) return rc;
    if (!task.empty() && (rc = set_task_impl(h, task.data(), mask))) return rc; // (masked: TwoGoals' word is a running count an unmasked env must keep)
    HIP_TRY(hipStreamSynchronize(h->stream)); // the host vectors above are about to go away
    if (obs) return run_step(h, false, nullptr, obs, nullptr, nullptr);
    return MGX_OK;
}

// ------------------------------------------------------------------------------------------------ bench helpers
extern "C" int mgx_fill_actions(mgx_handle h, uint64_t seed, int64_t env0, int64_t t0, int64_t T, uint8_t *actions)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_fill_actions");
    if (rc) return rc;
    if (!actions || T < 0) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_fill_actions: bad argument");
    if (is_device_ptr(actions)) {
        HIP_TRY(mgx_launch_fill_actions(actions, seed, env0, t0, h->n, T, h->stream));
        return MGX_OK;
    }
    for (int64_t t = 0; t < T; t++)
        for (int64_t e = 0; e < h->n; e++)
            actions[t * h->n + e] = (uint8_t)mgx_action_of(seed, (uint64_t)(env0 + e), (uint64_t)(t0 + t));
    return MGX_OK;
}

extern "C" uint32_t mgx_action_at(uint64_t seed, int64_t env, int64_t t) { return mgx_action_of(seed, (uint64_t)env, (uint64_t)t); }

extern "C" int mgx_step_kernel_name(mgx_handle h, char *out, int cap)
{
    if (!h || !out || cap < 1) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_step_kernel_name: null argument");
    char dyn_name[32];
    const char *name = mgx_step_kernel_label(base_params(h), h->kernel_mode); // (host-side table lookup: no device call)
    if (h->dyn_fused) { // walk + step in one launch (mgx_launch_step_dyn: sized instances for the registered Dynamic-Obstacles grids)
        const bool sized = (h->W == h->H) && (h->W == 5 || h->W == 6 || h->W == 8 || h->W == 16);
        if (h->bonus || h->dac) snprintf(dyn_name, sizeof dyn_name, "k_step_dyn_wrap");
        else snprintf(dyn_name, sizeof dyn_name, "k_step_dyn<%d,%d>", sized ? h->W : 0, sized ? h->H : 0);
        name = dyn_name;
    }
    const int len = (int)strlen(name);
    if (len + 1 > cap) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_step_kernel_name: need %d bytes", len + 1);
    memcpy(out, name, (size_t)len + 1);
    return len;
}

extern "C" int mgx_profile_begin_sampled(mgx_handle h, int stride)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_profile_begin");
    if (rc) return rc;
    if (stride < 1) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_profile_begin_sampled: stride must be >= 1");
    h->profiling = true;
    h->prof_stopped = false;
    h->prof_launches = 0;
    h->prof_stride = stride;
    h->prof_samples = 0;
    h->prof_kernel_ms = 0.0;
    h->prof_kernel_n = 0;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return MGX_OK;
}

extern "C" int mgx_profile_begin(mgx_handle h) { return mgx_profile_begin_sampled(h, 8); }

extern "C" int mgx_profile_stop(mgx_handle h)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_profile_stop");
    if (rc) return rc;
    if (!h->profiling) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_profile_stop without mgx_profile_begin");
    if (h->prof_stopped) return mgx_fail(MGX_ERR_INVALID_STATE, "mgx_profile_stop: already stopped");
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->prof_stopped = true;
    return MGX_OK;
}

extern "C" int mgx_profile_end(mgx_handle h, int64_t *launches, double *span_ms)
{
    DeviceGuard dev_guard;
    int rc = dev_guard.enter(h, "mgx_profile_end");
    if (rc) return rc;
    if (!h->profiling) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_profile_end without mgx_profile_begin");
    if (!h->prof_stopped) HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->prof_stopped = false;
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->profiling = false;
    for (int i = 0; i < h->prof_samples; i++) {
        float k = 0.f;
        HIP_TRY(hipEventElapsedTime(&k, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
        h->prof_kernel_ms += (double)k;
    }
    h->prof_kernel_n = h->prof_samples;
    h->prof_samples = 0;
    if (launches) *launches = h->prof_launches;
    if (span_ms) *span_ms = (double)ms;
    return MGX_OK;
}

extern "C" int mgx_profile_kernel(mgx_handle h, int64_t *samples, double *sum_ms)
{
    if (!h) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_profile_kernel: null handle");
    if (h->profiling) return mgx_fail(MGX_ERR_INVALID_ARG, "mgx_profile_kernel: call mgx_profile_end first");
    if (samples) *samples = h->prof_kernel_n;
    if (sum_ms) *sum_ms = h->prof_kernel_ms;
    return MGX_OK;
}
