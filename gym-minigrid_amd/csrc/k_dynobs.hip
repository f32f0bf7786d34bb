// k_dynobs.hip -- Dynamic-Obstacles: the obstacle walk that precedes the base step (k_dynobs, k_dynobs_init).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include "levelgen_core.h"
#include "mgx_device.h"
#include "dynobs_device.h"

// ------------------------------------------------------------------------------------------------
// Dynamic-Obstacles (envs/dynamicobstacles.py:60-89).  The obstacle walk draws from the env's own MT19937 stream
// inside step(), so it cannot live in the streaming step kernel: k_dynobs runs before it, one lane per env.
//   RNG   : the per-env block `mt` is the one k_seed/k_levelgen left behind (words [pos, 624) not drawn yet); `tape` holds the
//           accepted two-bit draws of its tempered words and of the first 224 words of the next block (see "the draw tape" below).
//   reset : the in-kernel auto-reset of the step kernels raises restart[env] (DynObsParams.regen); the walk then first restores
//           the obstacle order and the RNG position of the episode start (ReseedWrapper: seed(s) + reset()), and block + tape
//           only if the episode twisted or touched the block (bit 31 of the stored position).
//   -DMGX_EXP_DYN=<bits> (tools/build_variant.sh; never in the product build) are measurement aids, most of them with WRONG results:
//           1 no block service, 2 no walk, 4 ask for 8 waves per SIMD, 8 every placement through the draw-by-draw loop (a check aid:
//           results stay right), 16 nobody takes that loop, 32 count the lanes that do (mgx_debug_dyn_count).  profiles/r03_dynobs_steps.txt
//           holds what they measured.
namespace {
__global__ __launch_bounds__(256) void k_dynobs_init(const DynObsParams p)
{
    // (the snapshot of the RNG block of the envs this reset really re-seeded is taken by k_dynobs_tape, which has the block in registers
    // anyway: as 156 threads per env here -- 163 M threads at 1 Mi envs whatever the mask -- it made every caller-side reset(mask) of these
    // handles a 100 us launch)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n || (p.mask_reset && !p.mask_reset[t])) return;
    if (p.mask && !p.mask[t]) { // reset with the seed it already has: the next k_dynobs restores order, RNG position and block
        p.regen[t] = 1;
        return;
    }
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
    p.regen[t] = 0; // (k_dynobs_tape, next, turns the stream position k_levelgen left in `pos` into a rank and writes pos0)
}

// (the walk itself -- draw tape, straight-line placements, block service -- is dynobs_device.h: shared with the fused step kernel of k_step.hip)
// Reset time: tapes (and their episode-start copies) of the envs a reset really re-seeded, and their position as a rank; one wave
// per 64 envs.
__global__ __launch_bounds__(256) void k_dynobs_tape(const DynObsParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_blk[4][624 + MGX_DYN_TAPE_DW + MGX_DYN_STRIP / 4 + 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (env0 >= p.n) return; // wave-uniform
    const int64_t env = env0 + lane;
    const bool mine = env < p.n && (!p.mask_reset || p.mask_reset[env]) && (!p.mask || p.mask[env]);
    const uint32_t raw = mine ? p.pos[env] : 0u; // mt_idx as k_seed / k_levelgen left it: a stream position <= 624
    uint32_t *blk = s_blk[wv], *tp = blk + 624, *slot = tp + MGX_DYN_TAPE_DW + MGX_DYN_STRIP / 4;
    uint8_t *strip = reinterpret_cast<uint8_t *>(tp + MGX_DYN_TAPE_DW);
    for (dyn_u64 m = __ballot(mine); m; m &= m - 1) { // wave-uniform
        const int el = __builtin_ctzll(m);
        const int64_t e = env0 + el;
        const uint32_t *src = p.mt + e * 624;
        uint32_t v[10];
#pragma unroll
        for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; v[i] = src[k < 624 ? k : 623]; }
#pragma unroll
        for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; if (k < 624) { blk[k] = v[i]; p.mt0[e * 624 + k] = v[i]; } } // (+ the episode-start snapshot)
        wave_sync();
        uint32_t r624, rtot;
        const uint32_t want = (uint32_t)__shfl((int)raw, el);
        const uint32_t rank = dyn_build_tape<true>(blk, tp, strip, slot, lane, want < 624u ? want : 624u, r624, rtot);
        if (lane < MGX_DYN_TAPE_DW) { p.tape[e * MGX_DYN_TAPE_DW + lane] = tp[lane]; p.tape0[e * MGX_DYN_TAPE_DW + lane] = tp[lane]; }
        if (lane == el) { const uint32_t w = rank | (r624 << 10) | (rtot << 20); p.pos[e] = w; p.pos0[e] = w; p.sp0[e] = want; }
        wave_sync();
    }
}

#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 4)
#define MGX_DYN_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define MGX_DYN_OCC
#endif
template <int CW, int CH>
__global__ __launch_bounds__(256) MGX_DYN_OCC void k_dynobs(const DynObsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= p.n_tiles) return; // wave-uniform
    (void)dynobs_walk<CW, CH, false>(p, smem + (size_t)wv * p.wave_lds, lane, tile);
}


// The plain caller-side reset() of a Dynamic-Obstacles env (minigrid.py:831-858 without a seed() in front: run_tests.py:64-66): the next level
// is drawn from the stream where the obstacle walks left it.  The walk keeps that place as a RANK on the env's draw tape (or as a
// half-regenerated block); k_levelgen wants a complete block in `mt` and a stream position in `mt_idx`.  One wave per 64 envs, one masked
// env at a time, all 64 lanes on its block:
//   restart pending (the in-kernel auto-reset ended the episode, the next k_dynobs would have restored it): block, position = the episode start's;
//   stream position p (bit 30):   words [0, p % 624) of the block are the next block's already: finish it; position = p % 624;
//   rank r, nothing drawn since the episode start: the start's stream position (sp0) -- r alone does not say where in a run of rejected
//           words the stream stands, and the generator's next draw has another mask;
//   rank r otherwise:  the walk's last draw was accepted word r - 1 of the tape's 848 positions, at stream index k: position k + 1; past
//           624 the block is twisted on first.
__global__ __launch_bounds__(256) void k_dynobs_handover(const DynObsParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_blk[4][624 + 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (env0 >= p.n) return; // wave-uniform
    const int64_t env = env0 + lane;
    const bool mine = env < p.n && (!p.mask_reset || p.mask_reset[env]);
    const uint32_t pw = mine ? p.pos[env] : 0u;
    const bool restart = mine && p.regen[env];
    const int bk = (mine && p.bank) ? (int)p.bank[env] : 0;
    const int64_t senv = env + (int64_t)bk * p.bank_envs;
    const uint32_t pw0 = mine ? p.pos0[senv] : 0u, sp0 = mine ? p.sp0[senv] : 0u;
    uint32_t *blk = s_blk[wv], *slot = blk + 624;
    for (dyn_u64 m = __ballot(mine); m; m &= m - 1) { // wave-uniform
        const int el = __builtin_ctzll(m);
        const int64_t e = env0 + el, se = env0 + el + (int64_t)__shfl(bk, el) * p.bank_envs;
        const uint32_t w = (uint32_t)__shfl((int)pw, el), w0 = (uint32_t)__shfl((int)pw0, el), s0 = (uint32_t)__shfl((int)sp0, el);
        const bool rs = __shfl((int)restart, el) != 0;
        const bool dirty = (w & MGX_DYN_DIRTY) != 0u, raw = (w & MGX_DYN_INPLACE) != 0u;
        const bool at_start = rs || (!dirty && !raw && (w & 0x3FFFFFFFu) == (w0 & 0x3FFFFFFFu));
        uint32_t *dst = p.mt + e * 624;
        if (at_start) { // (the live block is the episode start's unless the episode twisted or touched it -- or belongs to another seed)
            if (rs && (dirty || p.bank)) {
                const uint32_t *src = p.mt0 + se * 624;
                uint32_t v[10];
#pragma unroll
                for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; v[i] = src[k < 624 ? k : 623]; }
#pragma unroll
                for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; if (k < 624) dst[k] = v[i]; }
            }
            if (lane == 0) p.mt_idx[e] = s0;
            continue;
        }
        {
            uint32_t v[10];
#pragma unroll
            for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; v[i] = dst[k < 624 ? k : 623]; }
#pragma unroll
            for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; if (k < 624) blk[k] = v[i]; }
        }
        wave_sync();
        uint32_t k0 = 0, position = 0; // words [0, k0) already belong to the next block
        bool twist = false;
        if (raw) { k0 = (w & 0x3FFFFFFFu) % 624u; position = k0; twist = true; }
        else {
            const uint32_t r = w & 1023u; // (> 0: a rank of 0 is the episode start's, handled above; defensively treated as position 0)
            uint32_t base = 0;
            if (lane == 0) *slot = 0xFFFFFFFFu;
            wave_sync();
#pragma unroll 1
            for (int rd = 0; rd < MGX_DYN_PLANE_DW / 2; rd++) {
                const int k = 64 * rd + lane;
                uint32_t y = 0;
                if (k < 624) y = blk[k];
                else if (k < MGX_DYN_POSITIONS) { const int j = k - 624; y = lg_twist_word(blk[j], blk[j + 1], blk[j + 397]); }
                const bool acc = k < MGX_DYN_POSITIONS && temper2(y) != 3u;
                const dyn_u64 mv = __ballot(acc);
                const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mv, 0u));
                if (acc && r > 0u && rank == r - 1u) *slot = (uint32_t)k;
                base += (uint32_t)__builtin_popcountll(mv);
            }
            wave_sync();
            const uint32_t k = *slot;
            position = k == 0xFFFFFFFFu ? 0u : k + 1u;
            if (position > 624u) { position -= 624u; twist = true; }
        }
        if (twist) { // the rest of the block, in chunks of <= 227 words (within one nobody needs a word the chunk itself produces), then word 623
            uint32_t c0 = k0;
            while (c0 < 623u) {
                const uint32_t c1 = c0 + 227u < 623u ? c0 + 227u : 623u;
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t jj = c0 + (uint32_t)lane + 64u * q;
                    y[q] = 0;
                    if (jj < c1) y[q] = lg_twist_word(blk[jj], blk[jj + 1u], jj < 227u ? blk[jj + 397u] : blk[jj - 227u]);
                }
                wave_sync();
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t jj = c0 + (uint32_t)lane + 64u * q;
                    if (jj < c1) blk[jj] = y[q];
                }
                wave_sync();
                c0 = c1;
            }
            if (lane == 0) blk[623] = lg_twist_word(blk[623], blk[0], blk[396]);
            wave_sync();
            for (int k = lane; k < 624; k += 64) dst[k] = blk[k];
        }
        if (lane == 0) p.mt_idx[e] = position;
        wave_sync();
    }
}
} // namespace

hipError_t mgx_launch_dynobs_handover(const DynObsParams &p, hipStream_t st)
{
    if (p.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dynobs_handover, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t mgx_launch_dynobs_init(const DynObsParams &p, hipStream_t st)
{
    if (p.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dynobs_init, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dynobs_tape, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p); // tapes of the re-seeded envs' blocks
    return hipGetLastError();
}

int mgx_dynobs_wave_lds(int LS) { return (64 * LS > 2496 ? ((64 * LS + 15) & ~15) : 2496) + 64 * 4 + MGX_DYN_TAPE_DW * 4 + MGX_DYN_STRIP + 16; } // (dynobs_device.h: dynobs_walk's layout)

hipError_t mgx_launch_dynobs(const DynObsParams &p, hipStream_t st)
{
    if (p.n_tiles == 0) return hipSuccess;
    if (p.wave_lds > 65536) return hipErrorInvalidValue;
    // waves per block: four, or one where a wave's LDS is large (16x16: 18 KB per wave; 1 Mi envs, us per launch: three-wave blocks
    // 255-281, two-wave 201, single-wave blocks 196 -- more of them fit a CU's 160 KB, and a block's LDS is free again as soon as its
    // one wave is done)
    int wpb = p.wave_lds > 8192 ? 1 : 4;
    static const int wpb_env = MGX_TUNE_ENV("MGX_DYN_WPB") ? atoi(MGX_TUNE_ENV("MGX_DYN_WPB")) : 0; // (tuning builds)
    if (wpb_env > 0 && wpb_env <= 4 && wpb_env * p.wave_lds <= 65536) wpb = wpb_env;
    const dim3 grid((unsigned)((p.n_tiles + wpb - 1) / wpb)), block(64 * wpb);
    const size_t shmem = (size_t)wpb * p.wave_lds;
#define CASE(w, h) if (p.W == w && p.H == h) { hipLaunchKernelGGL((k_dynobs<w, h>), grid, block, shmem, st, p); return hipGetLastError(); }
    CASE(5, 5) CASE(6, 6) CASE(8, 8) CASE(16, 16) // the registered Dynamic-Obstacles sizes
#undef CASE
    hipLaunchKernelGGL((k_dynobs<0, 0>), grid, block, shmem, st, p);
    return hipGetLastError();
}

#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 32)
extern "C" void mgx_debug_dyn_count(unsigned long long *out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dyn_count), 32); }
#endif

hipError_t mgx_preload_dynobs_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_dynobs_init));
}
