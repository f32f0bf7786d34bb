// k_levelgen.hip -- env.seed() and level generation on the device: k_seed, k_levelgen, k_consume.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include "levelgen_core.h"
#include "mgx_device.h"
#include "levelgen_device.h"

namespace {

#ifndef MGX_LG_WAVES
#define MGX_LG_WAVES 4
#endif
// (the generator itself -- word sources, lane-per-level fast path, wave-per-level slow path -- is levelgen_device.h)
template <bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MGX_LG_WAVES, MGX_LG_WAVES))) void k_levelgen(const LevelGenParams p, const FastLayout fl)
{
    levelgen_block<MULTI>(p, fl, (int)blockIdx.x);
}

// reset(): the freshly generated next-level buffer becomes the current episode (for the masked envs) and is flagged
// for regeneration
__global__ __launch_bounds__(256) void k_consume(const ConsumeParams p)
{
    // (no mask: every env) a thread per (env, 16-byte chunk of its row)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int CPE = (p.S + 15) >> 4; // chunks per env
    if (t < p.n * (int64_t)CPE) {
        const int64_t e = t / CPE;
        const int c = (int)(t - e * CPE);
        const int64_t se = p.bank ? e + (int64_t)p.bank[e] * p.bank_envs : e; // seed schedule: the snapshot of the list entry this episode runs on
        const uint32_t *s = reinterpret_cast<const uint32_t *>(p.cells0 + se * p.S) + 4 * c;
        uint32_t *d = reinterpret_cast<uint32_t *>(p.cells + e * p.S) + 4 * c;
        const int nd = (p.S >> 2) - 4 * c < 4 ? (p.S >> 2) - 4 * c : 4;
        for (int k = 0; k < nd; k++) d[k] = s[k];
        if (p.objaux) { // a generated level has no aux state; its boxes hold what the generator put into them (objcont0)
            uint32_t *pl[3] = {reinterpret_cast<uint32_t *>(p.objaux + e * p.S) + 4 * c, reinterpret_cast<uint32_t *>(p.objaux0 + se * p.S) + 4 * c,
                               reinterpret_cast<uint32_t *>(p.objcont + e * p.S) + 4 * c};
            const uint32_t *c0 = reinterpret_cast<const uint32_t *>(p.objcont0 + se * p.S) + 4 * c;
            for (int k = 0; k < nd; k++) { pl[0][k] = 0u; pl[1][k] = 0u; pl[2][k] = c0[k]; }
        }
    }
    if (t < p.n) {
        p.agent[t] = p.agent0[p.bank ? t + (int64_t)p.bank[t] * p.bank_envs : t];
        if (p.regen) p.regen[t] = p.flag_regen ? 1 : 0;
        if (p.objaux) p.objcarry[t] = (uint16_t)(MGX_CODE_EMPTY << 8);
        if (p.front) p.front[t] = 0;
        if (p.wcache) reinterpret_cast<uint32_t *>(p.wcache + t * 64)[15] = 0u;
        if (p.restart) p.restart[t] = 1;
    }
}

// (with a mask: the caller-side reset of the finished envs, ~1 % of them) a thread per env reads its mask byte -- one
// coalesced pass over the mask instead of one per chunk -- and the wave copies the rows of its masked envs together,
// one env at a time (14 -> 5 us at 1 Mi LavaCrossing envs)
__global__ __launch_bounds__(256) void k_consume_masked(const ConsumeParams p)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool mine = t < p.n && p.mask[t];
    const int bk = (mine && p.bank) ? (int)p.bank[t] : 0; // seed schedule: the list entry this episode runs on (k_bank_advance has moved it on)
    if (mine) {
        p.agent[t] = p.agent0[t + (int64_t)bk * p.bank_envs];
        if (p.regen) p.regen[t] = p.flag_regen ? 1 : 0;
        if (p.objaux) p.objcarry[t] = (uint16_t)(MGX_CODE_EMPTY << 8);
        if (p.front) p.front[t] = 0;
        if (p.wcache) reinterpret_cast<uint32_t *>(p.wcache + t * 64)[15] = 0u;
        if (p.restart) p.restart[t] = 1;
    }
    const int SD = p.S >> 2;
    for (u64 m = __ballot(mine); m; m &= m - 1) {
        const int el = __builtin_ctzll(m);
        const int64_t e = t - lane + el;
        const int64_t se = e + (int64_t)__shfl(bk, el) * p.bank_envs;
        const uint32_t *s = reinterpret_cast<const uint32_t *>(p.cells0 + se * p.S);
        uint32_t *d = reinterpret_cast<uint32_t *>(p.cells + e * p.S);
        for (int i = lane; i < SD; i += 64) d[i] = s[i];
        if (p.objaux) {
            uint32_t *a = reinterpret_cast<uint32_t *>(p.objaux + e * p.S), *a0 = reinterpret_cast<uint32_t *>(p.objaux0 + se * p.S);
            uint32_t *c = reinterpret_cast<uint32_t *>(p.objcont + e * p.S), *c0 = reinterpret_cast<uint32_t *>(p.objcont0 + se * p.S);
            for (int i = lane; i < SD; i += 64) { a[i] = 0u; a0[i] = 0u; c[i] = c0[i]; }
        }
    }
}

// ---- small kernels of the seed schedule (mgx_set_seed_schedule) and of the plain caller-side reset()
__global__ __launch_bounds__(256) void k_seed_column(const uint64_t *__restrict__ seeds, int K, int b, uint64_t *__restrict__ out, int64_t n)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = seeds[t * K + b];
}
__global__ __launch_bounds__(256) void k_bank_advance(uint8_t *bank, const uint8_t *__restrict__ mask, int K, int64_t n)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || (mask && !mask[t])) return;
    const int b = bank[t] + 1; // self.seed_idx = (self.seed_idx + 1) % len(self.seeds)   (wrappers.py:26)
    bank[t] = (uint8_t)(b >= K ? 0 : b);
}
// ring of two next-level buffers (StepParams.ring), caller-side: the masked envs consumed the buffer their bank names ...
__global__ __launch_bounds__(256) void k_ring_consumed(uint8_t *bank, uint8_t *regen, const uint8_t *__restrict__ mask, int64_t n, int ring)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || (mask && !mask[t])) return;
    const int b = bank[t] & (ring - 1);
    regen[t] = (uint8_t)(b + 1);
    bank[t] = (uint8_t)((b + 1) & (ring - 1));
}
// ... | buffer 0 holds the masked envs' next level (the single-buffer reset paths put it there) and buffer 1 is to be made behind it
// (flag 1: make buffer 0 first -- mgx_set_state, whose mask bytes are any non-zero value)
__global__ __launch_bounds__(256) void k_ring_init(uint8_t *bank, uint8_t *regen, const uint8_t *__restrict__ mask, int64_t n, int flag)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || (mask && !mask[t])) return;
    bank[t] = 0;
    regen[t] = (uint8_t)flag;
}
__global__ __launch_bounds__(256) void k_mark_plain_reset(const uint8_t *__restrict__ mask, uint8_t *regen, uint8_t *has_seed, uint8_t *reseeded, int64_t n)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || (mask && !mask[t])) return;
    regen[t] = 1;
    has_seed[t] = 0;
    reseeded[t] = 1;
}


// ------------------------------------------------------------------------------------------------
// env.seed(s) on the device, lane-per-env: gym's legacy key derivation (SHA-512 of str(seed), levelgen_core.h),
// MT19937 init_by_array written into the env's state block in HBM, and the first block twist, so that the block is
// ready to be read (mt_idx = 0) and k_levelgen's lane-per-level fast path can take it.  The recurrences of
// init_by_array are sequential per env (each word depends on the previous one); the envs of a wave advance in
// lock-step.  Passes re-read what earlier passes stored, 16 words at a time so that the loads are in flight together.
// A block first COMPACTS the envs of its span that really need seeding into a queue (mask, and envs that already hold
// this seed), so only ceil(count / 64) waves per block do the work.
#define MGX_SEED_SPAN 512
// Re-seeding with the seed an env already has (ReseedWrapper's seed(s); reset() at every episode boundary, or a caller's
// `if done: reset()` loop) reproduces the level and the RNG state the episode-start snapshot already holds: with
// `skip_same` those envs are left to k_consume alone (reseeded[env] = 0) -- no SHA-512, no init_by_array, no generator.
struct SeedBook { uint64_t *seed0; uint8_t *has_seed; uint8_t *reseeded; int skip_same; };

__device__ __forceinline__ bool seed_needed(const SeedBook &b, const uint64_t *seeds, int64_t env)
{
    const uint8_t hs = b.has_seed[env]; // (all three loads in flight together: `&&` on the loads themselves made them three round trips)
    const uint64_t s0 = b.seed0[env], sd = seeds[env];
    const bool same = b.skip_same && hs && s0 == sd;
    b.reseeded[env] = same ? 0 : 1;
    return !same;
}

// The first version walked HBM three times per env (pass 2 store, pass 3 load + store, twist load + store, every access a
// 4-byte word of a 2,496-byte-strided row: 7.1 ms per 1 Mi envs).  Here
//   * pass 2 is never stored: it is run once for its last word (pass 3 starts from it) and then RECOMPUTED in lockstep
//     with pass 3 -- two interleaved dependent chains per lane, one multiply each per word;
//   * pass 3's words leave as whole 64-byte segments (16 words collected in registers, 4 x 16-byte stores per lane);
//   * the first block twist is done by the whole wave, one env at a time: 624 words in (coalesced), three rounds in LDS
//     (new[k] needs old[k], old[k+1] and old[k+397] or new[k-227]: k < 227, k < 454, k < 623 are independent sets), 624 out.
// The table of init_genrand(19650218) is read at wave-uniform addresses (scalar loads).  7.1 -> 4.1 ms with these three;
// a 512-env span (4x the waves in flight) and loading the next env's row during the current twist: 2.05 ms per 1 Mi envs
// (7.5 KB of traffic per env: 3.8 TB/s).

__device__ __forceinline__ void seed_passes(uint64_t seed, const uint32_t *__restrict__ init, uint32_t *m)
{
    uint32_t key[2];
    const int klen = lg_seed_key(seed, key);
    const uint32_t kj0 = key[0], kj1 = klen == 2 ? key[1] + 1u : key[0]; // key[j] + j for j = (i - 1) % klen
    // pass 2 for its last word: x_i = f(x_{i-1}, init[i]) for i = 1..623, then once more at i = 1 on top of x_1
    uint32_t prev = init[0], v1 = 0;
#pragma unroll 1
    for (int i0 = 0; i0 < 624; i0 += 16) {
        uint32_t tab[16];
#pragma unroll
        for (int k = 0; k < 16; k++) tab[k] = init[i0 + k];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (i0 + k == 0) continue;
            prev = ib_step2(prev, tab[k], (k & 1) ? kj0 : kj1); // i = i0 + k, i0 even: j = (i - 1) & 1 picks kj0 for odd i
            if (i0 + k == 1) v1 = prev;
        }
    }
    const uint32_t m1 = ib_step2(prev, v1, kj1); // (623 % klen: kj1 either way)
    // pass 3 on top of pass 2, recomputed: y_i = g(y_{i-1}, x_i) for i = 2..623 starting from y_1' = m1, then y_1 = g(y_623, m1)
    uint32_t px = init[0], py = m1;
    uint32_t head[16]; // words 0..15 leave last (word 1 is the last one known)
    uint4 *m4 = reinterpret_cast<uint4 *>(m);
#pragma unroll 1
    for (int i0 = 0; i0 < 624; i0 += 16) {
        uint32_t tab[16], out[16];
#pragma unroll
        for (int k = 0; k < 16; k++) tab[k] = init[i0 + k];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            out[k] = 0u;
            if (i0 + k == 0) continue;
            px = ib_step2(px, tab[k], (k & 1) ? kj0 : kj1);
            if (i0 + k == 1) continue;
            py = ib_step3(py, px, (uint32_t)(i0 + k));
            out[k] = py;
        }
        if (i0 == 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) head[k] = out[k];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) m4[(i0 >> 2) + q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        }
    }
    head[0] = 0x80000000u;
    head[1] = ib_step3(py, m1, 1u);
#pragma unroll
    for (int q = 0; q < 4; q++) m4[q] = make_uint4(head[4 * q], head[4 * q + 1], head[4 * q + 2], head[4 * q + 3]);
}

// first genrand block twist of one env by the whole wave (s: 624 words of this wave's LDS; r: the env's row, loaded by
// twist_load -- the caller loads the NEXT env's row while this one is in LDS, so the global round trip is off the path)
__device__ __forceinline__ void twist_load(const uint32_t *m, uint32_t r[10], int lane)
{
#pragma unroll
    for (int j = 0; j < 10; j++) r[j] = (lane + 64 * j < 624) ? m[lane + 64 * j] : 0u;
}
__device__ __forceinline__ void twist_to_lds(uint32_t *s, const uint32_t r[10], int lane)
{
#pragma unroll
    for (int j = 0; j < 10; j++) if (lane + 64 * j < 624) s[lane + 64 * j] = r[j];
    wave_sync();
}
// m2 (or null): the block after that one as well (new_level_each_episode handles keep it ready: LevelGenParams.mt2)
__device__ __forceinline__ void twist_rounds(uint32_t *s, int lane);
__device__ __forceinline__ void twist_block_wave(uint32_t *m, uint32_t *m2, uint32_t *s, int lane)
{
    twist_rounds(s, lane);
#pragma unroll
    for (int j = 0; j < 10; j++) if (lane + 64 * j < 624) m[lane + 64 * j] = s[lane + 64 * j];
    if (m2) {
        twist_rounds(s, lane);
#pragma unroll
        for (int j = 0; j < 10; j++) if (lane + 64 * j < 624) m2[lane + 64 * j] = s[lane + 64 * j];
    }
    wave_sync();
}
__device__ __forceinline__ void twist_rounds(uint32_t *s, int lane)
{
    uint32_t v[4];
    // k in [0, 227): old[k], old[k+1], old[k+397]
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = lane + 64 * j; v[j] = k < 227 ? lg_twist_word(s[k], s[k + 1], s[k + 397]) : 0u; }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = lane + 64 * j; if (k < 227) s[k] = v[j]; }
    wave_sync();
    // k in [227, 454): old[k], old[k+1], new[k-227]
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = 227 + lane + 64 * j; v[j] = k < 454 ? lg_twist_word(s[k], s[k + 1], s[k - 227]) : 0u; }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = 227 + lane + 64 * j; if (k < 454) s[k] = v[j]; }
    wave_sync();
    // k in [454, 623): old[k], old[k+1], new[k-227]
#pragma unroll
    for (int j = 0; j < 3; j++) { const int k = 454 + lane + 64 * j; v[j] = k < 623 ? lg_twist_word(s[k], s[k + 1], s[k - 227]) : 0u; }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 3; j++) { const int k = 454 + lane + 64 * j; if (k < 623) s[k] = v[j]; }
    wave_sync();
    if (lane == 0) s[623] = lg_twist_word(s[623], s[0], s[396]);
    wave_sync();
}

// The first MGX_SEED_WIN words of the first twisted block, without the block: new[k] = old[k + 397] ^ g(old[k], old[k + 1]) for k < 227, so
// of init_by_array's 624 words only old[0 .. WIN] and old[397 .. 397 + WIN) are ever looked at -- but every one of them stands at the end of
// the two sequential passes, so the chains run in full (pass 2 once for its last word, then again in lockstep with pass 3, as in
// seed_passes) and simply keep nothing else: g(old[k], old[k + 1]) goes into a register as soon as old[k + 1] is known (k = 2 .. WIN - 1; old[1] is
// the chain's LAST word), and is folded with old[k + 397] when that one comes by.  No LDS, no HBM traffic but the 256-byte result.
__device__ __forceinline__ uint32_t tw_g(uint32_t a, uint32_t b)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ void seed_window(uint64_t seed, const uint32_t *__restrict__ init, uint32_t *win)
{
    constexpr int WIN = MGX_SEED_WIN;
    static_assert(WIN % 16 == 0 && WIN <= 96, "chunks of 16; old[0 .. WIN] must be complete before old[397] comes by");
    uint32_t key[2];
    const int klen = lg_seed_key(seed, key);
    const uint32_t kj0 = key[0], kj1 = klen == 2 ? key[1] + 1u : key[0];
    uint32_t prev = init[0], v1 = 0;
#pragma unroll 1
    for (int i0 = 0; i0 < 624; i0 += 16) {
        uint32_t tab[16];
#pragma unroll
        for (int k = 0; k < 16; k++) tab[k] = init[i0 + k];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (i0 + k == 0) continue;
            prev = ib_step2(prev, tab[k], (k & 1) ? kj0 : kj1);
            if (i0 + k == 1) v1 = prev;
        }
    }
    const uint32_t m1 = ib_step2(prev, v1, kj1);
    uint32_t px = init[0], py = m1;
    uint32_t t[WIN];          // t[k] = g(old[k], old[k + 1]), then new[k]
    uint32_t y2 = 0, ylast = 0, y397 = 0, y398 = 0;
#define MGX_SW_BARRIER __builtin_amdgcn_sched_barrier(0); /* (unrolled chunks: keeps the scheduler from running the x chain far ahead of the y chain) */
    // one chunk of 16 steps; EARLY: i0 <= WIN (old[2 .. WIN] come by), LATE: the chunks that hold old[397 .. 397 + WIN)
#define MGX_SW_CHUNK(I0, EARLY, LATE)                                                                       \
    {                                                                                                        \
        uint32_t tab[16];                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 16; k++) tab[k] = init[(I0) + k];                             \
        _Pragma("unroll") for (int k = 0; k < 16; k++) {                                                    \
            const int i = (I0) + k;                                                                          \
            if (i == 0) continue;                                                                            \
            px = ib_step2(px, tab[k], (k & 1) ? kj0 : kj1);                                                  \
            if (i == 1) continue;                                                                            \
            py = ib_step3(py, px, (uint32_t)i);                                                              \
            if (EARLY) {                                                                                     \
                if (i == 2) y2 = py;                                                                         \
                if (i >= 3 && i <= WIN) { uint32_t gv = tw_g(ylast, py); asm volatile("" : "+v"(gv)); t[(i - 1) < WIN ? (i - 1) : 0] = gv; } \
                ylast = py;                                                                                  \
            }                                                                                                \
            if (LATE) {                                                                                      \
                if (i == 397) y397 = py;                                                                     \
                if (i == 398) y398 = py;                                                                     \
                if (i >= 399 && i < 397 + WIN) { uint32_t nv = t[(i - 397) < WIN ? (i - 397) : 0] ^ py; asm volatile("" : "+v"(nv)); t[(i - 397) < WIN ? (i - 397) : 0] = nv; } \
            }                                                                                                \
        }                                                                                                    \
        MGX_SW_BARRIER \
    }
    // (the index clamps above only keep the compiler from seeing an out-of-range constant in dead arms; the empty asm statements pin each
    // folded word where it is computed -- left alone the compiler sinks all 64 folds to the end of the kernel and keeps the raw chain words,
    // split into three masked pieces each, alive until then: 249 VGPRs)
#pragma unroll
    for (int c = 0; c <= WIN / 16; c++) MGX_SW_CHUNK(16 * c, true, false)
#pragma unroll 1
    for (int i0 = WIN + 16; i0 < 384; i0 += 16) MGX_SW_CHUNK(i0, false, false)
#pragma unroll
    for (int c = 0; c < (397 + WIN - 384 + 15) / 16; c++) MGX_SW_CHUNK(384 + 16 * c, false, true)
#pragma unroll 1
    for (int i0 = 384 + 16 * ((397 + WIN - 384 + 15) / 16); i0 < 624; i0 += 16) MGX_SW_CHUNK(i0, false, false)
#undef MGX_SW_CHUNK
    const uint32_t y1 = ib_step3(py, m1, 1u); // old[1]; old[0] = 0x80000000
    t[0] = y397 ^ tw_g(0x80000000u, y1);
    t[1] = y398 ^ tw_g(y1, y2);
    uint4 *w4 = reinterpret_cast<uint4 *>(win);
#pragma unroll
    for (int q = 0; q < WIN / 4; q++) w4[q] = make_uint4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}

// up to 64 queued envs of a span by one wave, virtual form: every lane its env's window
__device__ __forceinline__ void seed_queue_wave_window(const uint64_t *__restrict__ seeds, const uint32_t *__restrict__ init, uint32_t *win, uint8_t *virt,
                                                       uint32_t *mt_idx, uint8_t *regen, const SeedBook &book, int64_t base, const uint16_t *s_queue,
                                                       int q0, int count, int lane)
{
    const int q = q0 + lane;
    if (q >= count) return;
    const int64_t env = base + s_queue[q];
    const uint64_t sd = seeds[env];
    seed_window(sd, init, win + env * MGX_SEED_WIN);
    mt_idx[env] = 0;
    regen[env] = 1;
    virt[env] = 1;
    book.seed0[env] = sd;
    book.has_seed[env] = 1;
}

// up to 64 queued envs of a span by one wave: the two passes lane-per-env, then the block twists together
__device__ __forceinline__ void seed_queue_wave(const uint64_t *__restrict__ seeds, const uint32_t *__restrict__ init, uint32_t *mt, uint32_t *mt2, uint32_t *mt_idx,
                                                uint8_t *regen, const SeedBook &book, int64_t base, const uint16_t *s_queue, int q0, int count,
                                                uint32_t *s_blk, int lane)
{
    const int q = q0 + lane;
    const bool have = q < count;
    const int64_t env = have ? base + s_queue[q] : 0;
    if (have) seed_passes(seeds[env], init, mt + env * 624);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the rows are read back by the whole wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int nq = count - q0 < 64 ? count - q0 : 64;
    uint32_t r[10];
    twist_load(mt + (base + s_queue[q0]) * 624, r, lane);
    for (int l = 0; l < nq; l++) {
        twist_to_lds(s_blk, r, lane);
        if (l + 1 < nq) twist_load(mt + (base + s_queue[q0 + l + 1]) * 624, r, lane); // in flight during the rounds below
        twist_block_wave(mt + (base + s_queue[q0 + l]) * 624, mt2 ? mt2 + (base + s_queue[q0 + l]) * 624 : nullptr, s_blk, lane);
    }
    if (have) {
        mt_idx[env] = 0; // the first block is ready
        regen[env] = 1;
        book.seed0[env] = seeds[env];
        book.has_seed[env] = 1;
    }
}

// (MATERIALIZE) as seed_queue_wave, for envs whose seed is on record: block into `mt`, virt cleared, nothing else touched
__device__ __forceinline__ void seed_queue_wave_materialize(const uint64_t *__restrict__ seed0, const uint32_t *__restrict__ init, uint32_t *mt, uint8_t *virt,
                                                            int64_t base, const uint16_t *s_queue, int q0, int count, uint32_t *s_blk, int lane)
{
    const int q = q0 + lane;
    const bool have = q < count;
    const int64_t env = have ? base + s_queue[q] : 0;
    if (have) seed_passes(seed0[env], init, mt + env * 624);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int nq = count - q0 < 64 ? count - q0 : 64;
    uint32_t r[10];
    twist_load(mt + (base + s_queue[q0]) * 624, r, lane);
    for (int l = 0; l < nq; l++) {
        twist_to_lds(s_blk, r, lane);
        if (l + 1 < nq) twist_load(mt + (base + s_queue[q0 + l + 1]) * 624, r, lane);
        twist_block_wave(mt + (base + s_queue[q0 + l]) * 624, nullptr, s_blk, lane);
    }
    if (have) virt[env] = 0;
}

// FORM 0: the full state (block in `mt`, first twist done).  FORM 1: the virtual state (`mt` = LevelGenParams.win, `mt2` = LevelGenParams.virt as
// bytes): no LDS block, twice the waves per SIMD.
#ifndef MGX_SEEDW_WAVES
#define MGX_SEEDW_WAVES 4 /* waves per SIMD the window form is held to (its 64 pending words live in registers) */
#endif
template <int FORM>
__device__ __forceinline__ void seed_body(const uint64_t *__restrict__ seeds, const uint8_t *__restrict__ mask,
                                              const uint32_t *__restrict__ init, uint32_t *mt, uint32_t *mt2, uint32_t *mt_idx,
                                              uint8_t *regen, SeedBook book, int64_t n)
{
    __shared__ uint16_t s_queue[MGX_SEED_SPAN];
    __shared__ int s_count;
    __shared__ uint32_t s_blk[FORM == 0 ? 4 : 1][FORM == 0 ? 624 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * MGX_SEED_SPAN;
    if (tid == 0) s_count = 0;
    __syncthreads();
    for (int k = tid; k < MGX_SEED_SPAN; k += 256) {
        const int64_t env = base + k;
        const bool mine = env < n && (!mask || mask[env]);
        // only the envs being reset: on Dynamic-Obstacles handles the flag of another env means "the in-kernel auto-reset
        // restarted this episode, restore obstacle order + RNG position first" (k_dynobs) and must survive a masked reset
        if (mine) regen[env] = 0; // (of these, only the ones seeded below get a new level)
        if (mine && seed_needed(book, seeds, env)) s_queue[atomicAdd(&s_count, 1)] = (uint16_t)k;
    }
    __syncthreads();
    const int count = s_count;
    if constexpr (FORM == 1) {
        for (int q0 = wv * 64; q0 < count; q0 += 256) seed_queue_wave_window(seeds, init, mt, reinterpret_cast<uint8_t *>(mt2), mt_idx, regen, book, base, s_queue, q0, count, lane);
    } else {
        for (int q0 = wv * 64; q0 < count; q0 += 256) seed_queue_wave(seeds, init, mt, mt2, mt_idx, regen, book, base, s_queue, q0, count, s_blk[wv], lane); // wave-uniform trip count
    }
}

// Masked form (caller-side `reset(mask = done)`: ~1 % of the envs): one wave per 512-env span compacts the masked envs
// whose seed really changed and runs the same wave body on them.  (The first version kept 16 state blocks in LDS and ran
// the three passes there: ~100 us per group, 320 us per launch with ~8 k new seeds at 1 Mi envs; this: 88 us, which is
// about the latency of one env's ~1,900 dependent multiply steps -- every wave has a few envs and nothing to overlap
// them with; staging the table in LDS changed nothing.  A dense mask is still correct here, only slower than k_seed's
// four waves per span.)
#define MGX_SEEDM_SPAN 512
// FORM 0 / 1 as k_seed.  FORM 2, MATERIALIZE (the plain reset() of envs that still hold a virtual state): the masked envs (mask null = all)
// with virt[env] set get their full first block from seed0 (= `seeds`) into `mt`; flags, seed book and mt_idx stay as they are.
#define MGX_SEED_ARGS const uint64_t *__restrict__ seeds, const uint8_t *__restrict__ mask, const uint32_t *__restrict__ init, uint32_t *mt, uint32_t *mt2, \
                      uint32_t *mt_idx, uint8_t *regen, SeedBook book, int64_t n
#ifndef MGX_SEED_WAVES
#define MGX_SEED_WAVES 3 /* waves per SIMD of the full form: it is bound by its 7.5 KB of row traffic per env, and more waves only deepen the queue
                            (1 Mi envs, us per launch: 2 waves 1,993 | 3: 1,975 | 4: 2,179 | the 5 its 94 VGPRs allow: 2,111) */
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MGX_SEED_WAVES, MGX_SEED_WAVES))) void k_seed(MGX_SEED_ARGS) { seed_body<0>(seeds, mask, init, mt, mt2, mt_idx, regen, book, n); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MGX_SEEDW_WAVES, MGX_SEEDW_WAVES))) void k_seed_window(MGX_SEED_ARGS)
{
    seed_body<1>(seeds, mask, init, mt, mt2, mt_idx, regen, book, n);
}

template <int FORM>
__device__ __forceinline__ void seed_masked_body(const uint64_t *__restrict__ seeds, const uint8_t *__restrict__ mask,
                                                    const uint32_t *__restrict__ init, uint32_t *mt, uint32_t *mt2, uint32_t *mt_idx,
                                                    uint8_t *regen, SeedBook book, int64_t n)
{
    __shared__ uint32_t s_blk[FORM == 1 ? 1 : 624];
    __shared__ uint16_t s_queue[MGX_SEEDM_SPAN], s_masked[MGX_SEEDM_SPAN];
    __shared__ int s_count;
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * MGX_SEEDM_SPAN;
    if (lane == 0) s_count = 0;
    wave_sync();
    // masked envs of the span first (LDS only), then the seed comparison for all of them at once: a lane that loaded its
    // operands inside the scan paid one round trip per mask byte position that any lane of the wave had set
    uint8_t *virt = reinterpret_cast<uint8_t *>(mt2); // (FORM 1 / 2)
    if constexpr (FORM == 2) {
        for (int k = lane; k < MGX_SEEDM_SPAN; k += 64)
            if (base + k < n && (!mask || mask[base + k]) && virt[base + k]) s_queue[atomicAdd(&s_count, 1)] = (uint16_t)k;
        wave_sync();
        const int cnt = s_count;
        // (the full-state body with its own seeds: mt_idx / regen / book writes of seed_queue_wave are redirected to scratch-free no-ops below)
        for (int q0 = 0; q0 < cnt; q0 += 64) seed_queue_wave_materialize(seeds, init, mt, virt, base, s_queue, q0, cnt, s_blk, lane);
        return;
    } else
    if (base + MGX_SEEDM_SPAN <= n && ((uintptr_t)(mask + base) & 7u) == 0) {
        // the span's 512 mask bytes in ONE round trip, 8 per lane (a byte per lane and round was 8 dependent round trips)
        static_assert(MGX_SEEDM_SPAN == 512, "8 mask bytes per lane");
        const u64 mb = *reinterpret_cast<const u64 *>(mask + base + 8 * lane);
        // flags of the masked envs only (see k_seed): bytes of `mb` that are non-zero -> 0xFF
        const u64 nz = ((((mb & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | mb) & 0x8080808080808080ull) >> 7;
        if (nz) *reinterpret_cast<u64 *>(regen + base + 8 * lane) &= ~(nz * 0xFFull); // (of these, only the ones seeded below get a new level)
#pragma unroll
        for (int b = 0; b < 8; b++)
            if ((mb >> (8 * b)) & 0xFFull) s_masked[atomicAdd(&s_count, 1)] = (uint16_t)(8 * lane + b);
    } else {
        for (int k = lane; k < MGX_SEEDM_SPAN; k += 64) {
            if (base + k < n && mask[base + k]) regen[base + k] = 0;
            if (base + k < n && mask[base + k]) s_masked[atomicAdd(&s_count, 1)] = (uint16_t)k;
        }
    }
    wave_sync();
    const int n_masked = s_count;
    wave_sync();
    if (lane == 0) s_count = 0;
    wave_sync();
    for (int q = lane; q < n_masked; q += 64) {
        const int k = s_masked[q];
        if (seed_needed(book, seeds, base + k)) s_queue[atomicAdd(&s_count, 1)] = (uint16_t)k;
    }
    wave_sync();
    const int count = s_count;
    if (count == 0) return;
    if constexpr (FORM == 1) {
        for (int q0 = 0; q0 < count; q0 += 64) seed_queue_wave_window(seeds, init, mt, virt, mt_idx, regen, book, base, s_queue, q0, count, lane);
    } else {
        for (int q0 = 0; q0 < count; q0 += 64) seed_queue_wave(seeds, init, mt, mt2, mt_idx, regen, book, base, s_queue, q0, count, s_blk, lane);
    }
}

__global__ __launch_bounds__(64) void k_seed_masked(MGX_SEED_ARGS) { seed_masked_body<0>(seeds, mask, init, mt, mt2, mt_idx, regen, book, n); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MGX_SEEDW_WAVES, MGX_SEEDW_WAVES))) void k_seed_masked_window(MGX_SEED_ARGS)
{
    seed_masked_body<1>(seeds, mask, init, mt, mt2, mt_idx, regen, book, n);
}
__global__ __launch_bounds__(64) void k_seed_materialize(MGX_SEED_ARGS) { seed_masked_body<2>(seeds, mask, init, mt, mt2, mt_idx, regen, book, n); }

} // namespace

static hipError_t mgx_levelgen_layout(const LevelGenParams &p, FastLayout *flp, unsigned *blocks, size_t *shmem_out)
{
    // slice layout per family: generous command capacities (a level that needs more goes to the slow path), crossing lists
    // only where rivers exist, the level image only for rows that fit
    FastLayout &fl = *flp;
    const int kind = p.cfg.level_kind, n_obj = p.cfg.level_arg0 > 0 && p.cfg.level_arg0 < 16 ? p.cfg.level_arg0 : 8;
    // 6*cap int16 words of workspace: the crossing lists; RoomGrid's door bookkeeping; MultiRoom's two room lists + entry walls (104 words:
    // rounds 1 and 2 gave it none, so every MultiRoom level was "too big" for the lane path and took the wave-per-level one).
    fl.river_cap = (kind == MGX_LEVEL_CROSSING) ? MGX_LGF_RIVERS : ((kind == MGX_LEVEL_KEYCORRIDOR || kind == MGX_LEVEL_OBSTRUCTEDMAZE) ? 5 : (kind == MGX_LEVEL_MULTIROOM ? 18 : 0));
    switch (kind) {
    case MGX_LEVEL_TWOGOALS: case MGX_LEVEL_EMPTY: case MGX_LEVEL_DOORKEY: case MGX_LEVEL_LAVAGAP: case MGX_LEVEL_DISTSHIFT: case MGX_LEVEL_GOTODOOR: case MGX_LEVEL_REDBLUEDOORS: fl.cmd_cap = 12; break;
    case MGX_LEVEL_FETCH: case MGX_LEVEL_GOTOOBJECT: case MGX_LEVEL_PUTNEAR: case MGX_LEVEL_DYNOBS: fl.cmd_cap = 6 + n_obj; break;
    case MGX_LEVEL_CROSSING: fl.cmd_cap = 24; break;
    case MGX_LEVEL_MEMORY: case MGX_LEVEL_UNLOCK: case MGX_LEVEL_FOURROOMS: fl.cmd_cap = 20; break;
    case MGX_LEVEL_MULTIROOM: { const int rooms = (p.cfg.level_arg0 >> 8) & 255; fl.cmd_cap = rooms >= 1 && rooms <= 8 ? 5 * rooms : MGX_LGF_CMDS; break; } // 4 walls + a door per room, the goal
    default: fl.cmd_cap = MGX_LGF_CMDS; break; // KeyCorridor, LockedRoom, Playground
    }
    fl.img_dw = p.S <= MGX_LGF_MAXS ? p.S >> 2 : 0;
    fl.slice_dw = (MGX_LGF_WIN + 2 * fl.cmd_cap + 3 * fl.river_cap + fl.img_dw) | 1; // odd dword stride
    const int slow_bytes = 4 * MGX_LG_LDS_PER_WAVE;
    // MultiRoom (a room chain restarted until it fits) and KeyCorridor (connect_all: a breadth-first search over the rooms per random door)
    // have long-tailed, branchy generators.  64 of them per wave run for the slowest lane through the union of all paths, and their
    // 36-43 KB of slices per wave leave one wave per SIMD: measured SLOWER than a whole wave per level (one active lane, but 20 waves per
    // CU).  262,144 envs all timing out together, us per step averaged over 600 steps (profiles/r03_levelgen_paths.txt):
    //   KeyCorridorS3R3  64 lanes per wave 238 | 16 lanes 225 | a wave per level 201 (512-env blocks) -> 188 (128-env blocks)   <- rule
    //   MultiRoom-N6     64 lanes 237-340      | 16 lanes 152 (64-env blocks)  <- rule | a wave per level 203 (512) -> 155 (64)
    //   MultiRoom-N4-S5  (round 2: 128)        | 16 lanes 62                   <- rule | a wave per level 101
    // Small blocks matter as much as the path: in a burst every env of a block's span has a level to make, and a 512-env span per
    // 4-wave block left the chip with 512 blocks of 128 sequential levels each.
    const bool heavy = kind == MGX_LEVEL_MULTIROOM || kind == MGX_LEVEL_KEYCORRIDOR;
    fl.lanes = 64;
    if (kind == MGX_LEVEL_MULTIROOM) fl.lanes = 16;
    if (kind == MGX_LEVEL_KEYCORRIDOR && p.mt2) fl.lanes = 16; // (with the second block: 155 us per step -- 8 lanes 160, 12: 157, 20: 158, 24 / 32: 162 -- against 178 for a wave per level)
    // (tuning builds only, read once per process and clamped: a negative MGX_LG_FAST_WAVES used to leave every flagged level ungenerated)
    struct LgTune { int lanes, fast_waves, span; };
    static const LgTune tune = [] {
        LgTune t = {0, -1, 0};
        if (const char *e = MGX_TUNE_ENV("MGX_LG_LANES")) { const int v = atoi(e); if (v >= 1 && v <= 64) t.lanes = v; }
        if (const char *e = MGX_TUNE_ENV("MGX_LG_FAST_WAVES")) { const int v = atoi(e); if (v >= 0 && v <= 4) t.fast_waves = v; }
        if (const char *e = MGX_TUNE_ENV("MGX_LG_SPAN")) { const int v = atoi(e); if (v >= 64 && v <= MGX_LGF_ENVS && (v & 63) == 0) t.span = v; }
        return t;
    }();
    if (tune.lanes) fl.lanes = tune.lanes;
    const int slice_bytes = fl.lanes * fl.slice_dw * 4;
    fl.n_fast_waves = 60 * 1024 / slice_bytes;
    if (fl.n_fast_waves > 4) fl.n_fast_waves = 4;
    if (fl.n_fast_waves < 1) fl.n_fast_waves = 1;
    if (kind == MGX_LEVEL_KEYCORRIDOR && !p.mt2) fl.n_fast_waves = 0; // a wave per level
    if (tune.fast_waves >= 0) fl.n_fast_waves = tune.fast_waves; // (0 = every level takes the wave-per-level path)
    fl.span = heavy ? 128 : 512;
    while (fl.span < MGX_LGF_ENVS && (p.n + fl.span - 1) / fl.span > (heavy ? 8192 : 512)) fl.span *= 2;
    if (tune.span) fl.span = tune.span;
    size_t shmem = (size_t)(fl.n_fast_waves > 0 ? fl.n_fast_waves : 1) * slice_bytes;
    if (shmem < (size_t)slow_bytes) shmem = slow_bytes;
    fl.queue_off = (int)((shmem + 15) & ~(size_t)15);
    shmem = (size_t)fl.queue_off + 4 * (size_t)fl.span; // the two queues: `span` 16-bit entries each
    *shmem_out = shmem;
    *blocks = (unsigned)((p.n + fl.span - 1) / fl.span);
    return hipSuccess;
}

hipError_t mgx_launch_levelgen(const LevelGenParams &p, hipStream_t st)
{
    FastLayout fl;
    unsigned blocks = 0;
    size_t shmem = 0;
    const hipError_t el = mgx_levelgen_layout(p, &fl, &blocks, &shmem);
    if (el != hipSuccess) return el;
    if (shmem > 60 * 1024) { // more dynamic LDS than a kernel gets unasked (up to 60 KB of slices + 8 KB of queues; tuning runs: up to 148 KB)
        static bool raised[64]; // per device (the attribute belongs to the function on the current device; racing callers set the same value)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!raised[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_levelgen<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_levelgen<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            if (e != hipSuccess) return e;
            raised[dev] = true;
        }
    }
    if (p.n_regen > 1) hipLaunchKernelGGL(k_levelgen<true>, dim3(blocks), dim3(256), shmem, st, p, fl);
    else hipLaunchKernelGGL(k_levelgen<false>, dim3(blocks), dim3(256), shmem, st, p, fl);
    return hipGetLastError();
}

hipError_t mgx_launch_seed(const uint64_t *seeds, const uint8_t *mask, const uint32_t *init, uint32_t *mt, uint32_t *mt2, uint32_t *mt_idx,
                           uint8_t *regen, uint64_t *seed0, uint8_t *has_seed, uint8_t *reseeded, int skip_same, int64_t n, hipStream_t st)
{
    const SeedBook book = {seed0, has_seed, reseeded, skip_same};
    if (mask) hipLaunchKernelGGL(k_seed_masked, dim3((unsigned)((n + MGX_SEEDM_SPAN - 1) / MGX_SEEDM_SPAN)), dim3(64), 0, st, seeds, mask, init, mt, mt2, mt_idx, regen, book, n);
    else hipLaunchKernelGGL(k_seed, dim3((unsigned)((n + MGX_SEED_SPAN - 1) / MGX_SEED_SPAN)), dim3(256), 0, st, seeds, mask, init, mt, mt2, mt_idx, regen, book, n);
    return hipGetLastError();
}

hipError_t mgx_launch_seed_window(const uint64_t *seeds, const uint8_t *mask, const uint32_t *init, uint32_t *win, uint8_t *virt, uint32_t *mt_idx,
                                  uint8_t *regen, uint64_t *seed0, uint8_t *has_seed, uint8_t *reseeded, int skip_same, int64_t n, hipStream_t st)
{
    const SeedBook book = {seed0, has_seed, reseeded, skip_same};
    uint32_t *virt32 = reinterpret_cast<uint32_t *>(virt); // (rides in the kernels' `mt2` slot)
    if (mask) hipLaunchKernelGGL(k_seed_masked_window, dim3((unsigned)((n + MGX_SEEDM_SPAN - 1) / MGX_SEEDM_SPAN)), dim3(64), 0, st, seeds, mask, init, win, virt32, mt_idx, regen, book, n);
    else hipLaunchKernelGGL(k_seed_window, dim3((unsigned)((n + MGX_SEED_SPAN - 1) / MGX_SEED_SPAN)), dim3(256), 0, st, seeds, mask, init, win, virt32, mt_idx, regen, book, n);
    return hipGetLastError();
}

hipError_t mgx_launch_seed_materialize(const uint8_t *mask, const uint32_t *init, uint32_t *mt, uint8_t *virt, const uint64_t *seed0, int64_t n, hipStream_t st)
{
    const SeedBook book = {nullptr, nullptr, nullptr, 0};
    hipLaunchKernelGGL(k_seed_materialize, dim3((unsigned)((n + MGX_SEEDM_SPAN - 1) / MGX_SEEDM_SPAN)), dim3(64), 0, st, seed0, mask, init, mt,
                       reinterpret_cast<uint32_t *>(virt), (uint32_t *)nullptr, (uint8_t *)nullptr, book, n);
    return hipGetLastError();
}

hipError_t mgx_launch_seed_column(const uint64_t *seeds, int K, int b, uint64_t *out, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_seed_column, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seeds, K, b, out, n);
    return hipGetLastError();
}

hipError_t mgx_launch_bank_advance(uint8_t *bank, const uint8_t *mask, int K, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_bank_advance, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bank, mask, K, n);
    return hipGetLastError();
}

hipError_t mgx_launch_ring_consumed(uint8_t *bank, uint8_t *regen, const uint8_t *mask, int64_t n, int ring, hipStream_t st)
{
    hipLaunchKernelGGL(k_ring_consumed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bank, regen, mask, n, ring);
    return hipGetLastError();
}

hipError_t mgx_launch_ring_init(uint8_t *bank, uint8_t *regen, const uint8_t *mask, int64_t n, int flag, hipStream_t st)
{
    hipLaunchKernelGGL(k_ring_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bank, regen, mask, n, flag);
    return hipGetLastError();
}

hipError_t mgx_launch_mark_plain_reset(const uint8_t *mask, uint8_t *regen, uint8_t *has_seed, uint8_t *reseeded, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_mark_plain_reset, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask, regen, has_seed, reseeded, n);
    return hipGetLastError();
}

hipError_t mgx_launch_consume(const ConsumeParams &p, hipStream_t st)
{
    if (p.mask) {
        hipLaunchKernelGGL(k_consume_masked, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    const int64_t total = p.n * (int64_t)((p.S + 15) >> 4);
    hipLaunchKernelGGL(k_consume, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}


hipError_t mgx_preload_levelgen_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_consume));
}
