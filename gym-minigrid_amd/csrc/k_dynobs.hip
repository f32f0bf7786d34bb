// k_dynobs.hip -- Dynamic-Obstacles: the obstacle walk that precedes the base step (k_dynobs, k_dynobs_init).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include "levelgen_core.h"
#include "mgx_device.h"

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

