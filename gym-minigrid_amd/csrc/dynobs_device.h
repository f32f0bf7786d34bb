// dynobs_device.h -- the obstacle walk of Dynamic-Obstacles (envs/dynamicobstacles.py:60-89) as device code shared by k_dynobs.hip (the walk as a
// kernel of its own, in front of whatever step kernel the handle takes) and k_step.hip (fused into the staged partial-view step: one launch, the
// tile staged once).
#ifndef MGX_DYNOBS_DEVICE_H
#define MGX_DYNOBS_DEVICE_H

#include "levelgen_core.h"
#include "mgx_device.h"

namespace {

// One wave per tile of 64 envs, lane per env, like k_step.  The walk is a chain of draw -> look at a cell -> maybe draw again,
// every link depending on the one before and diverging between lanes; written as that loop it ran for the wave's unluckiest lane
// with 25 of 64 lanes active and was bound by its chain of LDS round trips and branches (108 of the kernel's 143 us at 1 Mi 8x8 envs
// went to it, measured by leaving it out).  Now:
//   * every draw of the walk is `bounded(2)` (a 3-wide range: obstacles live in the interior, so the 3x3 box never clips):
//     masked rejection on the low two bits of the tempered word.  The accepted draws are therefore simply the words whose
//     two bits are not 3, in order.
//   * THE DRAW TAPE is that sequence, per block: the ACCEPTED draws of 848 stream positions -- the 624 words of the block and the
//     first 224 of the NEXT one, which depend on the old block only (new[k] = twist(old[k], old[k+1], old[k+397]) for k < 227) --
//     as two bit planes indexed by RANK (bit j of plane 0 / 1 = bit 0 / 1 of the j-th accepted draw).  The whole wave builds it once
//     per block (k_dynobs_tape at reset; the service loop of k_dynobs when a block is finished or restored): 14 rounds of 64
//     positions, each lane with an accepted word drops its two bits at byte `rank` of an LDS strip (rank = accepted words before it:
//     a running count + v_mbcnt of the round's ballot), then 14 rounds of two ballots over the strip.  An env's position is a rank,
//     and the per-env position word carries the block's two constants beside it: R624 (accepted draws among the block's own 624
//     words) and Rtot (among all 848).  A step reads 64 draws = 32 (dx, dy) samples as two unaligned 12-byte loads per lane.
//   * A PLACEMENT IS STRAIGHT-LINE CODE.  A sample is the pair (draw 2k, draw 2k+1); an obstacle moves to the first sample whose cell
//     is free.  For 16 samples at once: the even bits of the two planes say which dx each sample has (three masks), the odd bits which
//     dy; with the nine cells of the 3x3 box read from the LDS image, hit = OR over dx of (Xdx & OR over the free dy of that column
//     (Ydy)) is the mask of successful samples, and its lowest set bit the one the reference's loop stops at.  No loop, every lane
//     busy, same cost for the first sample as for the sixteenth.  (The agent's cell is marked in the LDS image, which is private to
//     this kernel, so "free" is one compare.)  A lane whose obstacle finds no free cell among its (at most 16) samples at hand, or
//     sits in a box without a free cell, takes that obstacle and the rest of its walk through the reference's loop as written
//     (`slow` below: draws one at a time, further windows of the tape, the hopeless-box skip, 101 samples at most).
//   * ranks >= R624 are consumed from the tape WITHOUT touching the block; the next step's service loop then twists the whole block
//     (in LDS, chunks of <= 227 independent words), rebuilds the tape and takes R624 off the position.  Only a lane that runs off the
//     tape altogether falls back to DynRng's word-by-word source at stream position 848, which first catches the block up in place
//     (bit 30 of the stored position: it then holds a STREAM position, and words [0, pos - 624) already belong to the new block).
//   * the tile's cells are staged in LDS as k_step stages them; moved obstacles are written through to HBM.
//   * history (1 Mi 8x8 envs, us per launch): lane-per-env straight from HBM 550-1,900; byte-wide windows in LDS 280; ballot-built
//     register windows 254 (round 1); a tape of all 848 positions + validity mask 186; + look-ahead 224, further windows, the
//     hopeless-box skip, one flat (obstacle, try) loop per lane: 143 (round 2); rank tape + straight-line placement: see DESIGN.md.
typedef unsigned long long dyn_u64;
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 32)
__device__ unsigned long long g_dyn_count[4];
#endif
#define MGX_DYN_PLANE_DW (MGX_DYN_TAPE_DW / 2)
#define MGX_DYN_POSITIONS 848 /* 624 + 224: the head of the next block depends on the old block only for k < 227 */
#define MGX_DYN_STRIP 896     /* bytes of the LDS strip the accepted draws are compacted into (ranks < 848; the last byte takes the rejected words) */
static_assert(MGX_DYN_POSITIONS <= 32 * MGX_DYN_PLANE_DW && MGX_DYN_STRIP == 32 * MGX_DYN_PLANE_DW && MGX_DYN_POSITIONS < 1024, "plane size / 10-bit ranks");
// the per-env position word: rank | R624 << 10 | Rtot << 20, or (bit 30) a stream position; bit 31 = the block in memory is no
// longer the episode-start block
#define MGX_DYN_INPLACE 0x40000000u
#define MGX_DYN_DIRTY 0x80000000u

// low two bits of genrand's tempering of y (checked against the full tempering on 1e6 random words)
__device__ __forceinline__ uint32_t temper2(uint32_t y)
{
    const uint32_t y1 = y ^ (y >> 11);
    return (y1 ^ (y1 >> 18) ^ (y1 >> 11) ^ ((y1 >> 3) & 1u)) & 3u;
}

// The word-by-word source behind the tape (stream position p >= 848 when it takes over).
struct DynRng {
    uint32_t *A;  // the env's block in HBM
    uint32_t p;   // stream position (0xFFFFFFFF: not in use, the tape still supplies the draws)
    bool inplace; // words [0, p - 624) of the next block have been generated in place (else the block is untouched)
    __device__ __forceinline__ void take_over()
    {
        p = MGX_DYN_POSITIONS; // the tape supplied positions 624 .. 847 without touching the block: catch the block up first
#pragma nounroll
        for (uint32_t k = 0; k < MGX_DYN_POSITIONS - 624u; k++) A[k] = lg_twist_word(A[k], A[k + 1u], A[k + 397u]); // (k < 224 < 227)
        inplace = true;
    }
    __device__ __forceinline__ int draw3() // _rand_int(t, t + 3) - t
    {
        for (;;) {
            const uint32_t k = p % 624u, k1 = k + 1u == 624u ? 0u : k + 1u, km = k + 397u >= 624u ? k + 397u - 624u : k + 397u;
            const uint32_t y = lg_twist_word(A[k], A[k1], A[km]);
            A[k] = y;
            p++;
            const uint32_t v = temper2(y);
            if (v != 3u) return (int)v;
        }
    }
};

// The tape of a complete block `blk` (624 words in LDS) into `tp` (MGX_DYN_TAPE_DW dwords in LDS: dword 2j = bit 0 of draws
// 32j .. 32j+31, dword 2j+1 = their bit 1, so that a window is ONE run of 24 bytes), by the whole wave; `strip` is MGX_DYN_STRIP
// bytes of LDS (16-byte aligned), `slot` one dword.  Returns the block's constants and, with WANT, the rank of stream position
// `want` <= 848 (= accepted draws in front of it) to every lane.
//   pass 1, 14 rounds of 64 positions: a lane with an accepted word drops its two bits at byte `rank` of the strip (rank = running
//           count + v_mbcnt of the round's ballot; the others write to the strip's last byte, which no rank reaches);
//   pass 2, lanes 0..27: 32 strip bytes -> one dword of each plane (bit k of four bytes at once: (w & 0x01010101) * 0x10204080 >> 28).
//           (First version: 14 more rounds of two ballots over the strip and four v_writelane each; 120 instructions more per block.)
template <bool WANT>
__device__ __forceinline__ uint32_t dyn_build_tape(const uint32_t *blk, uint32_t *tp, uint8_t *strip, uint32_t *slot, int lane, uint32_t want,
                                                   uint32_t &r624, uint32_t &rtot)
{
    uint32_t base = 0; // (wave-uniform)
    r624 = 0;
#pragma unroll
    for (int r = 0; r < MGX_DYN_PLANE_DW / 2; r++) {
        const int k = 64 * r + lane;
        uint32_t y = 0;
        if (k < 624) y = blk[k];
        else if (k < MGX_DYN_POSITIONS) { const int j = k - 624; y = lg_twist_word(blk[j], blk[j + 1], blk[j + 397]); } // (j < 224 < 227: old words only)
        const uint32_t v = k < MGX_DYN_POSITIONS ? temper2(y) : 3u;
        const dyn_u64 mv = __ballot(v != 3u);
        const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mv, 0u));
        strip[v != 3u ? rank : MGX_DYN_STRIP - 1u] = (uint8_t)v;
        if (WANT && (uint32_t)k == want) *slot = rank;
        if (r == 9) r624 = base + (uint32_t)__builtin_popcountll(mv & 0xFFFFFFFFFFFFull); // 624 = 64 * 9 + 48
        base += (uint32_t)__builtin_popcountll(mv);
    }
    rtot = base;
    wave_sync();
    if (lane < MGX_DYN_PLANE_DW) {
        const uint4 q0 = reinterpret_cast<const uint4 *>(strip)[2 * lane], q1 = reinterpret_cast<const uint4 *>(strip)[2 * lane + 1];
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        uint32_t p0 = 0, p1 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            p0 |= (((w[i] & 0x01010101u) * 0x10204080u) >> 28) << (4 * i);
            p1 |= ((((w[i] >> 1) & 0x01010101u) * 0x10204080u) >> 28) << (4 * i);
        }
        const uint32_t lo = 32u * (uint32_t)lane;
        const uint32_t keep = base >= lo + 32u ? 0xFFFFFFFFu : base > lo ? (1u << (base - lo)) - 1u : 0u; // (draws that exist)
        reinterpret_cast<uint2 *>(tp)[lane] = make_uint2(p0 & keep, p1 & keep);
    }
    wave_sync();
    return WANT ? *slot : 0u;
}

// 64 draws of a tape starting at rank `pos`: one unaligned 24-byte read -> (lo, hi).  Dwords past the tape's end are not read (a
// window that starts in the last 64 ranks is shifted in from the dwords that exist).
struct __attribute__((packed, aligned(4))) DynT6 { uint32_t a0, b0, a1, b1, a2, b2; };
template <typename P>
__device__ __forceinline__ void dyn_window(P tp, uint32_t pos, dyn_u64 &lo, dyn_u64 &hi)
{
    const uint32_t d = pos >> 5, sh = pos & 31u;
    DynT6 t;
    if (d + 2u < MGX_DYN_PLANE_DW) t = *reinterpret_cast<const DynT6 *>(tp + 2u * d);
    else {
        t.a0 = tp[2u * d]; t.b0 = tp[2u * d + 1u];
        t.a1 = d + 1u < MGX_DYN_PLANE_DW ? tp[2u * d + 2u] : 0u; t.b1 = d + 1u < MGX_DYN_PLANE_DW ? tp[2u * d + 3u] : 0u;
        t.a2 = 0u; t.b2 = 0u;
    }
    lo = (dyn_u64)__builtin_amdgcn_alignbit(t.a1, t.a0, sh) | ((dyn_u64)__builtin_amdgcn_alignbit(t.a2, t.a1, sh) << 32);
    hi = (dyn_u64)__builtin_amdgcn_alignbit(t.b1, t.b0, sh) | ((dyn_u64)__builtin_amdgcn_alignbit(t.b2, t.b1, sh) << 32);
}

// LDS of one wave: [0, cells_bytes) the tile image (before it arrives: 624 words of a block being finished / restored), then 64 position
// words, the tape of the block in service, the strip, one slot -- mgx_dynobs_wave_lds(LS) bytes in all.
__device__ __forceinline__ int dynobs_cells_bytes(int LS) { return 64 * LS > 2496 ? ((64 * LS + 15) & ~15) : 2496; }

// The walk of one tile by one wave (lane = env): block service, the tile staged into `lds`, every obstacle re-placed; obstacle order and RNG
// position written back.  Returns the lane's folded action (0..2) with bit 7 = "moved forward while the front cell was not clear".
// FUSED = false: the kernel of its own -- also writes that byte to act_out, the gather form's front entry, and the tile back to HBM.
// FUSED = true: the caller is the step kernel, goes on with the tile in LDS and writes it back itself.
template <int CW, int CH, bool FUSED>
__device__ __forceinline__ uint32_t dynobs_walk(const DynObsParams &p, uint8_t *lds, int lane, int tile)
{
    constexpr int CS = (CW && CH) ? ((CW * CH + 3) & ~3) : 0;
    const int H = CH ? CH : p.H, S = CS ? CS : p.S, LS = p.LS;
    const int cells_bytes = dynobs_cells_bytes(LS);
    uint32_t *blk = reinterpret_cast<uint32_t *>(lds); // 624 words: a block being restored / finished (before the cells arrive)
    uint32_t *ps = reinterpret_cast<uint32_t *>(lds + cells_bytes);
    uint32_t *tp = ps + 64;                            // the tape of the block in `blk` (MGX_DYN_TAPE_DW dwords)
    uint8_t *strip = reinterpret_cast<uint8_t *>(tp + MGX_DYN_TAPE_DW);
    uint32_t *slot = tp + MGX_DYN_TAPE_DW + MGX_DYN_STRIP / 4;
    const int64_t env0 = (int64_t)tile * 64, env = env0 + lane;
    const bool valid = env < p.n;

    uint2 ow = reinterpret_cast<const uint2 *>(p.obst)[env]; // (all per-env arrays are padded to whole tiles)
    uint32_t pos = p.pos[env];
    const bool regen = valid && p.regen[env];
    uint32_t a = valid ? p.actions[env] : 0u;
    const uint32_t rec = p.agent[env].x;
    bool dirty = (pos & MGX_DYN_DIRTY) != 0u;     // the block in memory is no longer the episode-start block
    bool inplace = (pos & MGX_DYN_INPLACE) != 0u; // `pos` is a stream position and words [0, pos - 624) of the next block were generated in place
    pos &= 0x3FFFFFFFu;
    // (under a seed schedule the new episode runs on another seed than the last one: its block always comes from the snapshot)
    const int bk = (p.bank && valid) ? (int)p.bank[env] : 0; // the list entry of the CURRENT episode (k_step / k_bank_advance moved it on)
    const int64_t senv = env + (int64_t)bk * p.bank_envs;
    const bool need_restore = regen && (dirty || p.bank != nullptr);
    // A block is finished -- twisted on, the position re-based -- once the walk has drawn PAST its last accepted word (rank > R624; `>=` until
    // round 4).  At rank == R624 the stream stands somewhere in the old block's tail of rejected words; the plain caller-side reset()
    // (k_dynobs_handover) needs that position back exactly -- the level generator's next draw has another mask and may accept those words --
    // and a re-based rank of 0 would have lost it.  The look-ahead the walk is guaranteed is the same (Rtot - R624 draws).  Without
    // obstacles nothing is ever drawn and nothing needs finishing.
    const bool need_finish = valid && !regen && p.n_obst > 0 && (inplace || (pos & 1023u) > ((pos >> 10) & 1023u));
    if (regen) { // the previous step ended the episode: cells/agent are already the episode start
        ow = reinterpret_cast<const uint2 *>(p.obst0)[senv];
        pos = p.pos0[senv];
        dirty = false;
        inplace = false;
        p.regen[env] = 0;
    }
    ps[lane] = pos | (inplace ? MGX_DYN_INPLACE : 0u);
    wave_sync();

    dyn_u64 w_lo = 0, w_hi = 0;
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 1) /* timing only (wrong results): no block service */
    const dyn_u64 m_restore = 0, m_serviced = 0;
#else
    const dyn_u64 m_restore = __ballot(need_restore);
    const dyn_u64 m_serviced = m_restore | __ballot(need_finish);
#endif
    for (dyn_u64 m = m_serviced; m; m &= m - 1) { // wave-uniform: one env at a time, all 64 lanes on its block
        const int e = __builtin_ctzll(m);
        uint4 *dst4 = reinterpret_cast<uint4 *>(p.mt) + (env0 + e) * 156;
        uint4 *blk4 = reinterpret_cast<uint4 *>(blk);
        uint32_t *tape_e = p.tape + (env0 + e) * MGX_DYN_TAPE_DW;
        uint32_t pw; // env e's position word afterwards (flags aside)
        if ((m_restore >> e) & 1ull) {
            const int64_t se = env0 + e + (int64_t)__shfl(bk, e) * p.bank_envs;
            const uint4 *src4 = reinterpret_cast<const uint4 *>(p.mt0) + se * 156;
            { // (the three loads together, on clamped indices: a load / store pair per trip compiled to three dependent round trips)
                uint4 v0 = src4[lane], v1 = src4[lane + 64], v2 = src4[lane + 128 < 156 ? lane + 128 : 155];
                dst4[lane] = v0; dst4[lane + 64] = v1;
                if (lane + 128 < 156) dst4[lane + 128] = v2;
            }
            if (lane < MGX_DYN_TAPE_DW) { const uint32_t v = p.tape0[se * MGX_DYN_TAPE_DW + lane]; tape_e[lane] = v; tp[lane] = v; }
            pw = ps[e] & 0x3FFFFFFFu;
        } else {
            {
                uint4 v0 = dst4[lane], v1 = dst4[lane + 64], v2 = dst4[lane + 128 < 156 ? lane + 128 : 155];
                blk4[lane] = v0; blk4[lane + 64] = v1;
                if (lane + 128 < 156) blk4[lane + 128] = v2;
            }
            wave_sync();
            const uint32_t pv = ps[e];
            const bool raw = (pv & MGX_DYN_INPLACE) != 0u;
            const uint32_t k0 = raw ? (pv & 0x3FFFFFFFu) % 624u : 0u; // words [0, k0) already belong to the new block
            uint32_t r624, rtot, rank = 0;
            if (!raw) { // the whole block: three chunks of 192 words (three full rounds each; within a chunk nobody needs a word the
                        // chunk itself produces: 192 < 227), then words 576 .. 622, then 623
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    uint32_t y[3];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const uint32_t jj = 192u * c + 64u * r + (uint32_t)lane;
                        y[r] = lg_twist_word(blk[jj], blk[jj + 1u], jj < 227u ? blk[jj + 397u] : blk[jj - 227u]);
                    }
                    wave_sync();
#pragma unroll
                    for (int r = 0; r < 3; r++) blk[192u * c + 64u * r + (uint32_t)lane] = y[r];
                    wave_sync();
                }
                const uint32_t jj = 576u + (uint32_t)lane;
                uint32_t y = 0;
                if (jj < 623u) y = lg_twist_word(blk[jj], blk[jj + 1u], blk[jj - 227u]);
                wave_sync();
                if (jj < 623u) blk[jj] = y;
                wave_sync();
            } else {
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
            }
            if (lane == 0) blk[623] = lg_twist_word(blk[623], blk[0], blk[396]);
            wave_sync();
            for (int i = lane; i < 156; i += 64) dst4[i] = blk4[i];
            // (the tape reads blk only: the block is complete)
            if (raw) rank = dyn_build_tape<true>(blk, tp, strip, slot, lane, k0, r624, rtot);
            else (void)dyn_build_tape<false>(blk, tp, strip, slot, lane, 0u, r624, rtot);
            if (lane < MGX_DYN_TAPE_DW) tape_e[lane] = tp[lane];
            pw = (raw ? rank : (pv & 1023u) - ((pv >> 10) & 1023u)) | (r624 << 10) | (rtot << 20);
        }
        wave_sync();
        if (lane == e) { pos = pw; dyn_window(tp, pw & 1023u, w_lo, w_hi); } // (from LDS: the global tape was only just written)
        wave_sync();
    }
    if (need_finish) { dirty = true; inplace = false; }
    if (valid && !((m_serviced >> lane) & 1ull)) { // this lane's window straight from its tape
        dyn_window(p.tape + env * MGX_DYN_TAPE_DW, pos & 1023u, w_lo, w_hi); // (not serviced: rank <= R624 <= 624)
    }
    // (round 4: asking for the tile's cells -- into registers -- and for these windows BEFORE the service loop, so that the wave's dependent
    // round trips overlap, measured slower: k_dynobs<8,8> 84 -> 89.6 us, the fused k_step_dyn<8,8> 118.9 -> 126.4 at 104 instead of 87 VGPRs)
    if (m_serviced) { // blocks rewritten by the whole wave may be read word-wise by single lanes below (off the tape): same
                      // CU, same L1, so the stores only have to be complete (an agent-scope fence would write back the XCD's L2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    stage_tile<CS>(p.cells, env0, S, LS, lds, lane);
    wave_sync();
    uint32_t folded = 6u; // (a lane past the batch's end: "done", no transition)
    if (valid) {
        uint8_t *g = lds + lane * LS;
        if (a >= 3u) a = 0u; // `if action >= self.action_space.n: action = 0`
        const int W = CW ? CW : p.W;
        const int ax = (int)(rec & 255u), ay = (int)((rec >> 8) & 255u), dir = (int)((rec >> 16) & 3u);
        const int fx = ax + (dir == 0) - (dir == 2), fy = ay + (dir == 1) - (dir == 3);
        bool not_clear = false; // front_cell and front_cell.type != 'goal', BEFORE the obstacles move
        if (fx >= 0 && fx < W && fy >= 0 && fy < H) {
            const uint32_t k = g[fx * H + fy] & 15u;
            not_clear = k != MGX_K_EMPTY && k != MGX_K_GOAL;
        }
        // The window: wl / wh hold the next `navail` draws of the tape from bit 0 up; `rpos` is the rank of the next draw.
        const uint32_t r624 = (pos >> 10) & 1023u, rtot = (pos >> 20) & 1023u;
        uint32_t rpos = pos & 1023u;
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 1)
        if (r624 && rpos >= r624) rpos %= r624; // (timing only: with no service the position wraps instead)
#endif
        dyn_u64 wl = w_lo, wh = w_hi;
        uint32_t navail = rtot - rpos < 64u ? rtot - rpos : 64u;
        const uint32_t *tape_l = p.tape + env * MGX_DYN_TAPE_DW;
        DynRng r = {p.mt + env * 624, 0xFFFFFFFFu, false};
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 2) /* timing only (wrong results): no placement */
        const int n_obst = 0;
#else
        const int n_obst = p.n_obst;
#endif
        // ---- straight-line placements (all lanes together, obstacle by obstacle)
        const uint8_t under_agent = g[ax * H + ay];
        g[ax * H + ay] = 0xFFu; // the agent's cell is not free: marked in the LDS image for the length of the walk
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 8) /* check aid: every placement through the loop */
        int slow_from = 0;
#else
        int slow_from = n_obst; // first obstacle this lane takes through the loop below
#endif
        int slow_tries = 0;     // ... and the samples that obstacle has already missed
        for (int i = 0; i < n_obst; i++) { // (wave-uniform trip count)
            const bool mine = slow_from == n_obst;
            const uint32_t o = (i < 4 ? ow.x >> (8 * i) : ow.y >> (8 * (i - 4))) & 255u; // x << 4 | y
            const int tx = (int)(o >> 4) - 1, ty = (int)(o & 15u) - 1; // top = old_pos + (-1, -1): interior, never clipped
            const uint8_t *b = g + tx * H + ty;
            uint32_t c00 = 0, c01 = 0, c02 = 0, c10 = 0, c11 = 0, c12 = 0, c20 = 0, c21 = 0, c22 = 0;
            if (mine) { c00 = b[0]; c01 = b[1]; c02 = b[2]; c10 = b[H]; c11 = b[H + 1]; c12 = b[H + 2]; c20 = b[2 * H]; c21 = b[2 * H + 1]; c22 = b[2 * H + 2]; }
            const uint32_t F = MGX_CODE_EMPTY;
            const bool f00 = c00 == F, f01 = c01 == F, f02 = c02 == F, f10 = c10 == F, f11 = c11 == F, f12 = c12 == F, f20 = c20 == F, f21 = c21 == F, f22 = c22 == F;
            bool pending = mine;
            int tries = 0;
            if (pending && !(f00 || f01 || f02 || f10 || f11 || f12 || f20 || f21 || f22)) {
                // no free cell in the box: all 101 samples fail, and all that is left of them is their 202 draws
                if (rtot - rpos >= 202u) { rpos += 202u; navail = 0u; } // (the next sample loads the window at the new rank)
                else { slow_from = i; slow_tries = 0; }                // (the tape ends first)
                pending = false;
            }
            // rounds of up to 16 samples (one is the rule; a box with one or two free cells may take more)
            while (__ballot(pending)) { // wave-uniform
                if (pending) {
                    if (navail < 2u) { // the window is used up: the next 64 draws of the tape -- or, at the tape's end, the loop below
                        if (rtot - rpos < 2u) { slow_from = i; slow_tries = tries; pending = false; }
                        else { dyn_window(tape_l, rpos, wl, wh); navail = rtot - rpos < 64u ? rtot - rpos : 64u; }
                    }
                }
                if (pending) {
                    const uint32_t E = 0x55555555u, l = (uint32_t)wl, h = (uint32_t)wh;
                    const uint32_t x1 = l & E, x2 = h & E, y1 = (l >> 1) & E, y2 = (h >> 1) & E;
                    const uint32_t x0 = E & ~(x1 | x2), y0 = E & ~(y1 | y2);
                    const uint32_t s0 = (f00 ? y0 : 0u) | (f01 ? y1 : 0u) | (f02 ? y2 : 0u);
                    const uint32_t s1 = (f10 ? y0 : 0u) | (f11 ? y1 : 0u) | (f12 ? y2 : 0u);
                    const uint32_t s2 = (f20 ? y0 : 0u) | (f21 ? y1 : 0u) | (f22 ? y2 : 0u);
                    uint32_t hit = (x0 & s0) | (x1 & s1) | (x2 & s2);
                    uint32_t ns = navail >> 1; // whole samples at hand, 16 looked at, 101 at most for one obstacle
                    ns = ns < 16u ? ns : 16u;
                    ns = ns < (uint32_t)(101 - tries) ? ns : (uint32_t)(101 - tries);
                    if (ns < 16u) hit &= (1u << (2u * ns)) - 1u;
                    if (hit != 0u) {
                        const uint32_t bp = (uint32_t)__builtin_ctz(hit); // even: the sample's dx draw
                        const int x = tx + (int)(((l >> bp) & 1u) | (((h >> bp) & 1u) << 1)), y = ty + (int)(((l >> (bp + 1u)) & 1u) | (((h >> (bp + 1u)) & 1u) << 1));
                        const uint32_t used = bp + 2u; // (<= 32)
                        wl >>= used; wh >>= used; navail -= used; rpos += used;
                        const int n8 = x * H + y, o8 = (tx + 1) * H + ty + 1;
                        g[n8] = (uint8_t)MGX_CODE_BALL_BLUE;
                        g[o8] = (uint8_t)MGX_CODE_EMPTY;
                        const uint32_t nb = ((uint32_t)x << 4) | (uint32_t)y;
                        if (i < 4) ow.x = (ow.x & ~(255u << (8 * i))) | (nb << (8 * i));
                        else ow.y = (ow.y & ~(255u << (8 * (i - 4)))) | (nb << (8 * (i - 4)));
                        pending = false;
                    } else { // ns samples missed
                        const uint32_t used = 2u * ns;
                        wl >>= used; wh >>= used; navail -= used; rpos += used;
                        tries += (int)ns;
                        if (tries >= 101) pending = false; // num_tries > max_tries raises (101 samples at most); the RecursionError is
                                                           // swallowed by the bare except: the obstacle stays
                    }
                }
            }
        }
        // ---- the reference's loop, draw by draw, for a lane whose walk reaches the end of the tape within this step
        // next window of the tape; false once the tape is used up (the word-by-word source goes on behind stream position 848)
        auto refill = [&]() -> bool {
            if (r.p != 0xFFFFFFFFu) return false;
            if (rpos >= rtot) { r.take_over(); return false; }
            dyn_window(tape_l, rpos, wl, wh);
            navail = rtot - rpos < 64u ? rtot - rpos : 64u;
            return true;
        };
        auto draw3 = [&]() -> int {
            if (navail == 0u && !refill()) return r.draw3();
            const int v = (int)(((uint32_t)wl & 1u) | (((uint32_t)wh & 1u) << 1));
            wl >>= 1; wh >>= 1; navail--; rpos++;
            return v;
        };
        // n accepted draws whose values nobody looks at (a placement that cannot succeed still draws 2 x 101 times)
        auto skip_draws = [&](uint32_t n) {
            if (r.p == 0xFFFFFFFFu) {
                const uint32_t take = rtot - rpos < n ? rtot - rpos : n;
                rpos += take; n -= take; navail = 0u; // (the next draw loads the window at the new rank)
                if (n == 0u) return;
                r.take_over();
            }
            for (; n > 0u; n--) (void)r.draw3();
        };
        // One loop over (obstacle, try) per lane, not a try loop per obstacle: flattened, the wave runs for the lane with the most
        // samples in total instead of the sum over obstacles of the per-obstacle maxima.
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 32) /* count: lanes / waves that take the loop, lane-steps */
        if (slow_from != n_obst) atomicAdd(&g_dyn_count[0], 1ull);
        if (__ballot(slow_from != n_obst) && lane == __builtin_ctzll(__ballot(1))) atomicAdd(&g_dyn_count[1], 1ull);
        atomicAdd(&g_dyn_count[2], 1ull);
#endif
#if defined(MGX_EXP_DYN) && (MGX_EXP_DYN & 16) /* timing only (wrong results): nobody takes the loop */
        slow_from = n_obst;
#endif
        bool look = true;
        for (int i = slow_from, tries = slow_tries; i < n_obst;) {
            const uint32_t o = (i < 4 ? ow.x >> (8 * i) : ow.y >> (8 * (i - 4))) & 255u; // x << 4 | y
            const int tx = (int)(o >> 4) - 1, ty = (int)(o & 15u) - 1;
            bool give_up = false;
            if (look) { // before an obstacle's first sample here: look at the 3x3 box -- with no free cell in it the remaining samples are
                        // known to fail, and all that is left of them is their draws
                look = false;
                bool any = false;
#pragma unroll
                for (int dxy = 0; dxy < 9; dxy++) {
                    const int x = tx + dxy / 3, y = ty + dxy % 3;
                    any = any || g[x * H + y] == MGX_CODE_EMPTY;
                }
                if (!any) { skip_draws(2u * (uint32_t)(101 - tries)); give_up = true; }
            }
            if (!give_up) {
                const int x = tx + draw3(), y = ty + draw3();
                if (g[x * H + y] == MGX_CODE_EMPTY) { // (the agent's cell carries the mark)
                    const int n8 = x * H + y, o8 = (tx + 1) * H + ty + 1;
                    g[n8] = (uint8_t)MGX_CODE_BALL_BLUE;
                    g[o8] = (uint8_t)MGX_CODE_EMPTY;
                    const uint32_t nb = ((uint32_t)x << 4) | (uint32_t)y;
                    if (i < 4) ow.x = (ow.x & ~(255u << (8 * i))) | (nb << (8 * i));
                    else ow.y = (ow.y & ~(255u << (8 * (i - 4)))) | (nb << (8 * (i - 4)));
                    i++; tries = 0; look = true;
                    continue;
                }
                give_up = ++tries > 100; // (101 samples at most)
            }
            if (give_up) { i++; tries = 0; look = true; }
        }
        reinterpret_cast<uint2 *>(p.obst)[env] = ow;
        // (rank >= R624: the next step's service loop twists the block first)
        p.pos[env] = r.p == 0xFFFFFFFFu ? (rpos | (r624 << 10) | (rtot << 20) | (dirty ? MGX_DYN_DIRTY : 0u)) : (r.p | MGX_DYN_INPLACE | MGX_DYN_DIRTY);
        folded = a | ((a == 2u && not_clear) ? 0x80u : 0u);
        if constexpr (!FUSED) p.act_out[env] = (uint8_t)folded;
        // the gather form of k_step (16x16) keeps the cell in front of the agent from its last observation pass; the walk has just made that
        // stale, and this kernel has the image at hand (0 = unknown: outside the grid)
        if constexpr (!FUSED) { if (p.front) p.front[env] = (fx >= 0 && fx < W && fy >= 0 && fy < H) ? g[fx * H + fy] : (uint8_t)0; }
        g[ax * H + ay] = under_agent;
    }
    // The moved obstacles go back as the whole tile, coalesced (64 x S bytes per wave), not cell by cell (two byte stores per moved obstacle
    // and lane, each to a line of its own: 111 -> 99 us per launch at 1 Mi 8x8 envs).
    wave_sync();
    if constexpr (!FUSED) unstage_tile<CS>(p.cells, env0, S, LS, lds, lane); // (fused into the step kernel: the tile goes home once, behind the step)
    return folded;
}

} // namespace

#endif
