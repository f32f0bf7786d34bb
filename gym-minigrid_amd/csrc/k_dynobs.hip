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
//   RNG   : the per-env block `mt` is the one k_seed/k_levelgen left behind (words [pos, 624) not drawn yet); `tape` holds the
//           low two bits of its tempered words and of the first 224 words of the next block (see "the draw tape" below).
//   reset : the in-kernel auto-reset of the step kernels raises restart[env] (DynObsParams.regen); the walk then first restores
//           the obstacle order and the RNG position of the episode start (ReseedWrapper: seed(s) + reset()), and block + tape
//           only if the episode twisted or touched the block (bit 31 of the stored position).
namespace {
__global__ __launch_bounds__(256) void k_dynobs_init(const DynObsParams p)
{
    // snapshot of the RNG block (coalesced: 156 uint4 per env) of the envs this reset really re-seeded
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < p.n * 156) {
        const int64_t e = t / 156;
        if ((!p.mask_reset || p.mask_reset[e]) && (!p.mask || p.mask[e])) reinterpret_cast<uint4 *>(p.mt0)[t] = reinterpret_cast<const uint4 *>(p.mt)[t];
    }
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
    p.pos0[t] = p.pos[t];
    p.regen[t] = 0;
}

// One wave per tile of 64 envs, lane per env, like k_step.  The walk is a chain of draw -> look at a cell -> maybe draw again,
// every link depending on the one before and diverging between lanes.
//   * every draw of the walk is `bounded(2)` (a 3-wide range: obstacles live in the interior, so the 3x3 box never clips):
//     masked rejection on the low two bits of the tempered word.  The accepted draws are therefore simply the words whose
//     two bits are not 3, in order: a window of 64 stream positions is three 64-bit masks (valid, bit 0, bit 1), and a draw
//     is ffbl + clear-lowest-bit + two bit extracts: no loop, no memory access.
//   * THE DRAW TAPE.  Each block has a tape: two bit planes (bit k of plane 0 / 1 = bit 0 / 1 of tempered word k) over 848
//     stream positions -- the 624 words of the block and the first 224 of the NEXT one, which depend on the old block only
//     (new[k] = twist(old[k], old[k+1], old[k+397]) for k < 227).  The whole wave builds it once per block (k_dynobs_tape at
//     reset; the service loop of k_dynobs when a block is finished or restored: 14 rounds of 64 positions, two ballots
//     each); a step reads its window as two unaligned 12-byte loads per lane, and again for every further 64 positions a
//     long placement needs.  (Round 1 rebuilt every env's window every step with 64 coalesced loads + 192 ballots per wave --
//     ~1,700 of its 4,400 VALU instructions and 256 B/env of reads for the ~11 words a step consumes -- and every env within 64
//     words of its block's end, i.e. some lane of nearly every wave, fell through to one dependent global load per draw.)
//   * positions 624..847 are consumed from the tape WITHOUT touching the block; the stored position then is >= 624 and the next
//     step's service loop twists the whole block (in LDS, chunks of <= 227 independent words), rebuilds the tape and takes
//     624 off the position.  Only a lane that runs off the tape altogether falls back to DynRng's word-by-word source, which
//     first catches the block up in place (bit 30 of the stored position: words [0, pos - 624) already belong to the new block).
//   * the tile's cells are staged in LDS as k_step stages them; moved obstacles are written through to HBM.
//   * history (1 Mi 8x8 envs, us per launch): lane-per-env straight from HBM 550-1,900; byte-wide windows in LDS 280; ballot-built
//     register windows 254 (round 1); the tape 186; + look-ahead 224, further windows, the hopeless-box skip, one flat
//     (obstacle, try) loop per lane: 143.
typedef unsigned long long dyn_u64;
#define MGX_DYN_PLANE_DW (MGX_DYN_TAPE_DW / 2)
#define MGX_DYN_POSITIONS 848 /* 624 + 224: the head of the next block depends on the old block only for k < 227 */
static_assert(MGX_DYN_POSITIONS <= 32 * MGX_DYN_PLANE_DW && (MGX_DYN_POSITIONS - 64) / 32 + 2 < MGX_DYN_PLANE_DW,
              "the last window advance() may take starts at 848 - 64 and reads three dwords of each plane: they must lie inside the plane");

// low two bits of genrand's tempering of y (checked against the full tempering on 1e6 random words)
__device__ __forceinline__ uint32_t temper2(uint32_t y)
{
    const uint32_t y1 = y ^ (y >> 11);
    return (y1 ^ (y1 >> 18) ^ (y1 >> 11) ^ ((y1 >> 3) & 1u)) & 3u;
}

struct DynRng {
    dyn_u64 valid, lo, hi; // window: word pos+i is an accepted draw / its bit 0 / its bit 1
    uint32_t *A;           // the env's block in HBM
    uint32_t pos, c;       // window start, words of it consumed
    uint32_t p;            // absolute position once the window is used up (0xFFFFFFFF while inside it)
    bool inplace;          // words [0, p - 624) of the next block have been generated in place (else the block is untouched)
    __device__ __forceinline__ int draw3() // _rand_int(t, t + 3) - t
    {
        if (p == 0xFFFFFFFFu) {
            const dyn_u64 m = c < 64u ? valid >> c : 0ull;
            if (m) {
                const uint32_t idx = c + (uint32_t)__builtin_ctzll(m);
                c = idx + 1u;
                return (int)(((lo >> idx) & 1ull) | (((hi >> idx) & 1ull) << 1));
            }
            p = pos + 64u; // first word behind the window (a window always has 64 positions: the tape looks 224 words ahead)
            if (p > 624u && !inplace) { // the tape supplied positions 624 .. p-1 without touching the block: catch the block up first
#pragma nounroll
                for (uint32_t k = 0; k < p - 624u; k++) A[k] = lg_twist_word(A[k], A[k + 1u], A[k + 397u]); // (k < 224 < 227)
            }
            if (p >= 624u) inplace = true;
        }
        for (;;) {
            uint32_t y;
            if (p < 624u) y = A[p];
            else {
                inplace = true;
                const uint32_t k = p % 624u, k1 = k + 1u == 624u ? 0u : k + 1u, km = k + 397u >= 624u ? k + 397u - 624u : k + 397u;
                y = lg_twist_word(A[k], A[k1], A[km]);
                A[k] = y;
            }
            p++;
            const uint32_t v = temper2(y);
            if (v != 3u) return (int)v;
        }
    }
    __device__ __forceinline__ uint32_t end_pos() const { return p == 0xFFFFFFFFu ? pos + c : p; }
};


// The tape of a complete block `blk` (624 words in LDS) into `tp` (MGX_DYN_TAPE_DW dwords in LDS), by the whole wave:
// MGX_DYN_PLANE_DW / 2 = 14 rounds of 64 stream positions, two ballots each.  Position k >= 624 is word k - 624 of the next block;
// the plane's last 48 positions (848 .. 895) do not exist and are marked as rejected draws (both bits set), so a window that
// reached them would skip them instead of reading zeros as draws (advance() stops at 848 anyway).
__device__ __forceinline__ void dyn_build_tape(const uint32_t *blk, uint32_t *tp, int lane)
{
#pragma unroll
    for (int r = 0; r < MGX_DYN_PLANE_DW / 2; r++) {
        const int k = 64 * r + lane;
        uint32_t y = 0;
        if (k < 624) y = blk[k];
        else if (k < MGX_DYN_POSITIONS) { const int j = k - 624; y = lg_twist_word(blk[j], blk[j + 1], blk[j + 397]); } // (j < 224 < 227: old words only)
        const uint32_t v = k < MGX_DYN_POSITIONS ? temper2(y) : 3u;
        const dyn_u64 ml = __ballot((v & 1u) != 0u), mh = __ballot((v & 2u) != 0u);
        if (lane == 0) { tp[2 * r] = (uint32_t)ml; tp[2 * r + 1] = (uint32_t)(ml >> 32); }
        if (lane == 1) { tp[MGX_DYN_PLANE_DW + 2 * r] = (uint32_t)mh; tp[MGX_DYN_PLANE_DW + 2 * r + 1] = (uint32_t)(mh >> 32); }
    }
}

// 64 positions of a tape starting at `pos` (< 624): unaligned 12-byte reads of the two planes -> (lo, hi)
__device__ __forceinline__ void dyn_window(const uint32_t *tp, uint32_t pos, dyn_u64 &lo, dyn_u64 &hi)
{
    const uint32_t d = pos >> 5, sh = pos & 31u;
    const uint32_t a0 = tp[d], a1 = tp[d + 1], a2 = tp[d + 2];
    const uint32_t b0 = tp[MGX_DYN_PLANE_DW + d], b1 = tp[MGX_DYN_PLANE_DW + d + 1], b2 = tp[MGX_DYN_PLANE_DW + d + 2];
    lo = (dyn_u64)__builtin_amdgcn_alignbit(a1, a0, sh) | ((dyn_u64)__builtin_amdgcn_alignbit(a2, a1, sh) << 32);
    hi = (dyn_u64)__builtin_amdgcn_alignbit(b1, b0, sh) | ((dyn_u64)__builtin_amdgcn_alignbit(b2, b1, sh) << 32);
}

// Reset time: tapes (and their episode-start copies) of the envs a reset really re-seeded; one wave per 64 envs.
__global__ __launch_bounds__(256) void k_dynobs_tape(const DynObsParams p)
{
    __shared__ uint32_t s_blk[4][624 + MGX_DYN_TAPE_DW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (env0 >= p.n) return; // wave-uniform
    const int64_t env = env0 + lane;
    const bool mine = env < p.n && (!p.mask_reset || p.mask_reset[env]) && (!p.mask || p.mask[env]);
    uint32_t *blk = s_blk[wv], *tp = blk + 624;
    for (dyn_u64 m = __ballot(mine); m; m &= m - 1) { // wave-uniform
        const int64_t e = env0 + __builtin_ctzll(m);
        const uint32_t *src = p.mt + e * 624;
        uint32_t v[10];
#pragma unroll
        for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; v[i] = k < 624 ? src[k] : 0u; }
#pragma unroll
        for (int i = 0; i < 10; i++) { const int k = lane + 64 * i; if (k < 624) blk[k] = v[i]; }
        wave_sync();
        dyn_build_tape(blk, tp, lane);
        wave_sync();
        if (lane < MGX_DYN_TAPE_DW) { p.tape[e * MGX_DYN_TAPE_DW + lane] = tp[lane]; p.tape0[e * MGX_DYN_TAPE_DW + lane] = tp[lane]; }
        wave_sync();
    }
}

template <int CW, int CH>
__global__ __launch_bounds__(256) void k_dynobs(const DynObsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= p.n_tiles) return; // wave-uniform
    constexpr int CS = (CW && CH) ? ((CW * CH + 3) & ~3) : 0;
    const int H = CH ? CH : p.H, S = CS ? CS : p.S, LS = p.LS;
    uint8_t *lds = smem + (size_t)wv * p.wave_lds;
    const int cells_bytes = 64 * LS > 2496 ? ((64 * LS + 15) & ~15) : 2496;
    uint32_t *blk = reinterpret_cast<uint32_t *>(lds); // 624 words: a block being restored / finished (before the cells arrive)
    uint32_t *ps = reinterpret_cast<uint32_t *>(lds + cells_bytes);
    uint32_t *tp = ps + 64;                            // the tape of the block in `blk` (MGX_DYN_TAPE_DW dwords)
    const int64_t env0 = (int64_t)tile * 64, env = env0 + lane;
    const bool valid = env < p.n;

    uint2 ow = reinterpret_cast<const uint2 *>(p.obst)[env]; // (all per-env arrays are padded to whole tiles)
    uint32_t pos = p.pos[env];
    const bool regen = valid && p.regen[env];
    uint32_t a = valid ? p.actions[env] : 0u;
    const uint32_t rec = p.agent[env].x;
    bool dirty = (pos >> 31) != 0u;          // the block in memory is no longer the episode-start block
    bool inplace = ((pos >> 30) & 1u) != 0u; // words [0, pos - 624) of the next block were generated in place (window overrun)
    pos &= 0x3FFFFFFFu;
    const bool need_restore = regen && dirty, need_finish = valid && !regen && pos >= 624u;
    if (regen) { // the previous step ended the episode: cells/agent are already the episode start
        ow = reinterpret_cast<const uint2 *>(p.obst0)[env];
        pos = p.pos0[env];
        dirty = false;
        inplace = false;
        p.regen[env] = 0;
    }
    ps[lane] = pos | (inplace ? 0x40000000u : 0u);
    wave_sync();

    dyn_u64 w_lo = 0, w_hi = 0;
    const dyn_u64 m_restore = __ballot(need_restore);
    const dyn_u64 m_serviced = m_restore | __ballot(need_finish);
    for (dyn_u64 m = m_serviced; m; m &= m - 1) { // wave-uniform: one env at a time, all 64 lanes on its block
        const int e = __builtin_ctzll(m);
        uint4 *dst4 = reinterpret_cast<uint4 *>(p.mt) + (env0 + e) * 156;
        uint4 *blk4 = reinterpret_cast<uint4 *>(blk);
        uint32_t *tape_e = p.tape + (env0 + e) * MGX_DYN_TAPE_DW;
        uint32_t pe;
        if ((m_restore >> e) & 1ull) {
            const uint4 *src4 = reinterpret_cast<const uint4 *>(p.mt0) + (env0 + e) * 156;
            { // (the three loads together, on clamped indices: a load / store pair per trip compiled to three dependent round trips)
                uint4 v[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { const int i = lane + 64 * k; v[k] = src4[i < 156 ? i : 155]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { const int i = lane + 64 * k; if (i < 156) dst4[i] = v[k]; }
            }
            if (lane < MGX_DYN_TAPE_DW) { const uint32_t v = p.tape0[(env0 + e) * MGX_DYN_TAPE_DW + lane]; tape_e[lane] = v; tp[lane] = v; }
            pe = ps[e] & 0x3FFFFFFFu;
        } else {
            {
                uint4 v[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { const int i = lane + 64 * k; v[k] = dst4[i < 156 ? i : 155]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { const int i = lane + 64 * k; if (i < 156) blk4[i] = v[k]; }
            }
            wave_sync();
            const uint32_t pv = ps[e];
            const uint32_t k0 = (pv & 0x40000000u) ? (pv & 0x3FFFFFFFu) % 624u : 0u; // words [0, k0) already belong to the new block
            pe = (pv & 0x3FFFFFFFu) % 624u;
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
            dyn_build_tape(blk, tp, lane); // (reads blk only: the block is complete)
            wave_sync();
            if (lane < MGX_DYN_TAPE_DW) tape_e[lane] = tp[lane];
        }
        wave_sync();
        if (lane == e) dyn_window(tp, pe, w_lo, w_hi); // (from LDS: the global tape was only just written)
        wave_sync();
    }
    if (need_finish) { pos %= 624u; dirty = true; inplace = false; }
    if (valid && !((m_serviced >> lane) & 1ull)) { // this lane's window straight from its tape: two unaligned 12-byte reads
        struct __attribute__((packed, aligned(4))) T3 { uint32_t a, b, c; };
        const uint32_t *tpe = p.tape + env * MGX_DYN_TAPE_DW;
        const uint32_t d = pos >> 5, sh = pos & 31u;
        const T3 x = *reinterpret_cast<const T3 *>(tpe + d), y = *reinterpret_cast<const T3 *>(tpe + MGX_DYN_PLANE_DW + d);
        w_lo = (dyn_u64)__builtin_amdgcn_alignbit(x.b, x.a, sh) | ((dyn_u64)__builtin_amdgcn_alignbit(x.c, x.b, sh) << 32);
        w_hi = (dyn_u64)__builtin_amdgcn_alignbit(y.b, y.a, sh) | ((dyn_u64)__builtin_amdgcn_alignbit(y.c, y.b, sh) << 32);
    }
    const dyn_u64 w_valid = ~(w_lo & w_hi); // masked rejection: the word is redrawn when its two bits are 3
    if (m_serviced) { // blocks rewritten by the whole wave may be read word-wise by single lanes below (a window overrun): same
                      // CU, same L1, so the stores only have to be complete (an agent-scope fence would write back the XCD's L2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    stage_tile<CS>(p.cells, env0, S, LS, lds, lane);
    wave_sync();
    if (!valid) return;

    uint8_t *g = lds + lane * LS;
    uint8_t *gg = p.cells + env * S;
    if (a >= 3u) a = 0u; // `if action >= self.action_space.n: action = 0`
    const int W = CW ? CW : p.W;
    const int ax = (int)(rec & 255u), ay = (int)((rec >> 8) & 255u), dir = (int)((rec >> 16) & 3u);
    const int fx = ax + (dir == 0) - (dir == 2), fy = ay + (dir == 1) - (dir == 3);
    bool not_clear = false; // front_cell and front_cell.type != 'goal', BEFORE the obstacles move
    if (fx >= 0 && fx < W && fy >= 0 && fy < H) {
        const uint32_t k = g[fx * H + fy] & 15u;
        not_clear = k != MGX_K_EMPTY && k != MGX_K_GOAL;
    }
    DynRng r = {w_valid, w_lo, w_hi, p.mt + env * 624, pos, 0u, 0xFFFFFFFFu, inplace};
    // The window as two 32-bit halves (fv/fl/fh: valid positions left / bit 0 / bit 1 of the current half): a draw is
    // ffbl + clear-lowest-bit + two bit extracts, all 32-bit and branch-free but for the switch to the upper half.  The generic
    // source `r` takes over behind the window (a step that needs more than its ~48 valid draws: rare).
    uint32_t fv = (uint32_t)w_valid, fl = (uint32_t)w_lo, fh = (uint32_t)w_hi, fbase = 0u, flast = 0u;
    uint32_t fv1 = (uint32_t)(w_valid >> 32), fl1 = (uint32_t)(w_lo >> 32), fh1 = (uint32_t)(w_hi >> 32);
    uint32_t wstart = pos; // stream position of the window's first entry
    const uint32_t *tape_l = p.tape + env * MGX_DYN_TAPE_DW;
    // next half / next window; false once the tape of this block is used up (positions >= 624 + 64: the generic source goes on)
    auto advance = [&]() -> bool {
        if (fv1 != 0u) { fv = fv1; fl = fl1; fh = fh1; fv1 = 0u; fbase = 32u; return true; }
        if (r.p != 0xFFFFFFFFu || wstart + 128u > MGX_DYN_POSITIONS) return false;
        wstart += 64u; // a long placement (an obstacle with few free neighbours): the next 64 positions of the tape
        struct __attribute__((packed, aligned(4))) T3 { uint32_t a, b, c; };
        const uint32_t d = wstart >> 5, sh = wstart & 31u;
        const T3 x = *reinterpret_cast<const T3 *>(tape_l + d), y = *reinterpret_cast<const T3 *>(tape_l + MGX_DYN_PLANE_DW + d);
        fl = __builtin_amdgcn_alignbit(x.b, x.a, sh); fl1 = __builtin_amdgcn_alignbit(x.c, x.b, sh);
        fh = __builtin_amdgcn_alignbit(y.b, y.a, sh); fh1 = __builtin_amdgcn_alignbit(y.c, y.b, sh);
        fv = ~(fl & fh); fv1 = ~(fl1 & fh1);
        fbase = 0u; flast = 0u;
        r.pos = wstart; // (the generic source, should it take over, continues behind THIS window)
        if (fv == 0u) { fv = fv1; fl = fl1; fh = fh1; fv1 = 0u; fbase = 32u; }
        return fv != 0u;
    };
    auto draw3 = [&]() -> int {
        if (fv == 0u && !advance()) { r.c = 64u; return r.draw3(); }
        const uint32_t idx = (uint32_t)__builtin_ctz(fv);
        fv &= fv - 1u;
        flast = fbase + idx + 1u;
        return (int)(((fl >> idx) & 1u) | (((fh >> idx) & 1u) << 1));
    };
    // n accepted draws whose values nobody looks at (a placement that cannot succeed still draws 2 x 101 times)
    auto skip_draws = [&](int n) {
        while (n > 0) {
            if (fv == 0u && !advance()) { r.c = 64u; for (; n > 0; n--) (void)r.draw3(); return; }
            const int have = __builtin_popcount(fv);
            if (have <= n) { flast = fbase + 32u - (uint32_t)__builtin_clz(fv); fv = 0u; n -= have; } // the whole half
            else { for (; n > 0; n--) { flast = fbase + (uint32_t)__builtin_ctz(fv) + 1u; fv &= fv - 1u; } }
        }
    };
    // One loop over (obstacle, try) per lane, not a try loop per obstacle: with nested loops every lane waits, obstacle by
    // obstacle, for the wave's unluckiest placement (an obstacle with one free neighbour needs ~9 samples, some lane's 30+),
    // i.e. the sum over obstacles of the per-obstacle maxima; flattened, the wave runs for the lane with the most samples
    // in total.  (SQ counters before: 21 of 64 lanes active on average.)
    for (int i = 0, tries = 0; i < p.n_obst;) {
        const uint32_t o = (i < 4 ? ow.x >> (8 * i) : ow.y >> (8 * (i - 4))) & 255u; // x << 4 | y
        const int tx = (int)(o >> 4) - 1, ty = (int)(o & 15u) - 1; // top = old_pos + (-1, -1): interior, never clipped
        bool give_up = false;
        if (tries == 8) { // eight misses: look at the 3x3 box once -- with no free cell in it the remaining 93 samples are
                          // known to fail too, and all that is left of them is their 186 draws
            bool any = false;
#pragma unroll
            for (int dxy = 0; dxy < 9; dxy++) {
                const int x = tx + dxy / 3, y = ty + dxy % 3;
                any = any || (g[x * H + y] == MGX_CODE_EMPTY && !(x == ax && y == ay));
            }
            if (!any) { skip_draws(2 * (101 - tries)); give_up = true; }
        }
        if (!give_up) {
            const int x = tx + draw3(), y = ty + draw3();
            if (g[x * H + y] == MGX_CODE_EMPTY && !(x == ax && y == ay)) {
                const int n8 = x * H + y, o8 = (tx + 1) * H + ty + 1;
                g[n8] = (uint8_t)MGX_CODE_BALL_BLUE; gg[n8] = (uint8_t)MGX_CODE_BALL_BLUE;
                g[o8] = (uint8_t)MGX_CODE_EMPTY; gg[o8] = (uint8_t)MGX_CODE_EMPTY;
                const uint32_t nb = ((uint32_t)x << 4) | (uint32_t)y;
                if (i < 4) ow.x = (ow.x & ~(255u << (8 * i))) | (nb << (8 * i));
                else ow.y = (ow.y & ~(255u << (8 * (i - 4)))) | (nb << (8 * (i - 4)));
                i++; tries = 0;
                continue;
            }
            give_up = ++tries > 100; // num_tries > max_tries raises (101 samples at most); the RecursionError is swallowed
                                     // by the bare except: the obstacle stays
        }
        if (give_up) { i++; tries = 0; }
    }
    if (r.p == 0xFFFFFFFFu) r.c = flast; // (else the generic source holds the position; r.pos = start of the last window)
    reinterpret_cast<uint2 *>(p.obst)[env] = ow;
    const uint32_t pe = r.end_pos(); // (>= 624: the next step's service loop twists the block first)
    p.pos[env] = pe | ((dirty || r.inplace) ? 0x80000000u : 0u) | (r.inplace ? 0x40000000u : 0u);
    p.act_out[env] = (uint8_t)(a | ((a == 2u && not_clear) ? 0x80u : 0u));
}
} // namespace

hipError_t mgx_launch_dynobs_init(const DynObsParams &p, hipStream_t st)
{
    const int64_t total = p.n * 156;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dynobs_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_dynobs_tape, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p); // tapes of the re-seeded envs' blocks
    return hipGetLastError();
}

int mgx_dynobs_wave_lds(int LS) { return (64 * LS > 2496 ? ((64 * LS + 15) & ~15) : 2496) + 64 * 4 + ((MGX_DYN_TAPE_DW * 4 + 15) & ~15); }

hipError_t mgx_launch_dynobs(const DynObsParams &p, hipStream_t st)
{
    if (p.n_tiles == 0) return hipSuccess;
    if (p.wave_lds > 65536) return hipErrorInvalidValue;
    int wpb = 65536 / p.wave_lds;
    if (wpb > 4) wpb = 4;
    const dim3 grid((unsigned)((p.n_tiles + wpb - 1) / wpb)), block(64 * wpb);
    const size_t shmem = (size_t)wpb * p.wave_lds;
#define CASE(w, h) if (p.W == w && p.H == h) { hipLaunchKernelGGL((k_dynobs<w, h>), grid, block, shmem, st, p); return hipGetLastError(); }
    CASE(5, 5) CASE(6, 6) CASE(8, 8) CASE(16, 16) // the registered Dynamic-Obstacles sizes
#undef CASE
    hipLaunchKernelGGL((k_dynobs<0, 0>), grid, block, shmem, st, p);
    return hipGetLastError();
}

hipError_t mgx_preload_dynobs_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_dynobs_init));
}
