// k_epilogue.hip -- observation epilogues behind k_step: one-hot wrappers (k_onehot) and FlatObsWrapper (k_flat).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgx_internal.h"
#include "mgx_kernels.h"
#include "mgx_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// One-hot epilogue: (type, color, state) triples -> NB = 11 + NC + NS bytes per cell with three ones
//   OneHotPartialObsWrapper  wrappers.py:203-243  (NC 7, NS 3 -> 21 channels: out[type] = out[11+color] = out[18+state] = 1)
//   FullyObsOneHotWrapper    wrappers.py:340-415  (NS 4 because the agent cell carries its direction; NC 7 or 0 = drop_color)
// The observation batch is one flat array of cells.  A lane takes 4 consecutive cells (12 input bytes, one
// dwordx3 load) and produces their 4*NB output bytes = NB whole dwords; every output byte is ONE compare because
// its (cell, channel) is known at compile time.  The wave's 64*NB dwords are contiguous in the output, so they are
// transposed through LDS and leave as 16-B/lane coalesced stores.
template <int NC, int NS>
__global__ __launch_bounds__(256) void k_onehot(const uint8_t *__restrict__ tri, uint8_t *__restrict__ out, int64_t n_cells)
{
    constexpr int NB = 11 + NC + NS;
    __shared__ __attribute__((aligned(16))) uint32_t s_x[4][64 * NB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
    const int64_t cell0 = wave * 256; // first cell of this wave
    if (cell0 >= n_cells) return;
    const int64_t c = cell0 + 4 * (int64_t)lane;
    uint32_t ty[4], co[4], st[4];
    if (c + 3 < n_cells) {
        struct __attribute__((packed, aligned(4))) In12 { uint32_t a, b, c; };
        const In12 v = *reinterpret_cast<const In12 *>(tri + c * 3);
        ty[0] = v.a & 255u; co[0] = (v.a >> 8) & 255u; st[0] = (v.a >> 16) & 255u;
        ty[1] = v.a >> 24;  co[1] = v.b & 255u;        st[1] = (v.b >> 8) & 255u;
        ty[2] = (v.b >> 16) & 255u; co[2] = v.b >> 24; st[2] = v.c & 255u;
        ty[3] = (v.c >> 8) & 255u; co[3] = (v.c >> 16) & 255u; st[3] = v.c >> 24;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool ok = c + k < n_cells;
            ty[k] = ok ? tri[(c + k) * 3] : 255u; co[k] = ok ? tri[(c + k) * 3 + 1] : 255u; st[k] = ok ? tri[(c + k) * 3 + 2] : 255u;
        }
    }
    uint32_t *x = s_x[wv] + lane * NB;
#pragma unroll
    for (int d = 0; d < NB; d++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int q = 4 * d + b, k = q / NB, ch = q % NB; // compile-time
            const bool one = ch < 11 ? ty[k] == (uint32_t)ch : (ch < 11 + NC ? co[k] == (uint32_t)(ch - 11) : st[k] == (uint32_t)(ch - 11 - NC));
            w |= (uint32_t)one << (8 * b);
        }
        x[d] = w;
    }
    wave_sync();
    const int64_t wave_cells = n_cells - cell0 < 256 ? n_cells - cell0 : 256;
    const int n_bytes = (int)wave_cells * NB;
    uint8_t *dst = out + cell0 * NB; // 256*NB bytes per wave: 16-B aligned
    const uint4 *x4 = reinterpret_cast<const uint4 *>(s_x[wv]);
    for (int i = lane; i < (n_bytes + 15) / 16; i += 64) {
        if (16 * i + 16 <= n_bytes) nt_store16(reinterpret_cast<uint4 *>(dst) + i, x4[i]);
        else
            for (int b = 16 * i; b < n_bytes; b++) dst[b] = reinterpret_cast<const uint8_t *>(s_x[wv])[b];
    }
}


} // namespace

namespace {
// DACWrapper.step's `return self.last_obs, ...` (wrappers.py:58-78; last_obs['image'] = obs['image'] * 0 + 1, :51): a wave looks at 64 envs'
// records and fills the observation row of every absorbed one with ones, 64 bytes per instruction (rows are `row_bytes` apart, any alignment).
__global__ __launch_bounds__(256) void k_dac_obs(const uint2 *__restrict__ agent, uint8_t *__restrict__ obs, int64_t n, int64_t row_bytes)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool mine = t < n && (agent[t].x & (1u << 22)); // MGX_REC_ABSORBED (mgx_device.h)
    for (unsigned long long m = __ballot(mine); m; m &= m - 1) {
        const int64_t e = t - lane + __builtin_ctzll(m);
        uint8_t *row = obs + e * row_bytes;
        for (int64_t i = lane; i < row_bytes; i += 64) row[i] = 1;
    }
}
} // namespace

hipError_t mgx_launch_dac_obs(const uint2 *agent, uint8_t *obs, int64_t n, int64_t row_bytes, hipStream_t st)
{
    hipLaunchKernelGGL(k_dac_obs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, agent, obs, n, row_bytes);
    return hipGetLastError();
}

hipError_t mgx_launch_onehot(const uint8_t *tri, uint8_t *out, int64_t n_cells, int nc, int ns, hipStream_t st)
{
    const dim3 grid((unsigned)((n_cells + 1023) / 1024)), block(256);
    if (nc == 7 && ns == 3) hipLaunchKernelGGL((k_onehot<7, 3>), grid, block, 0, st, tri, out, n_cells);
    else if (nc == 7 && ns == 4) hipLaunchKernelGGL((k_onehot<7, 4>), grid, block, 0, st, tri, out, n_cells);
    else if (nc == 0 && ns == 4) hipLaunchKernelGGL((k_onehot<0, 4>), grid, block, 0, st, tri, out, n_cells);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}


// FlatObsWrapper.observation (wrappers.py:556-577): out[env] = f32(image bytes) ++ one-hot of the mission string
// (96 positions x 27 codes, 95 % of the row).  `pattern` holds that one-hot block as floats, one row of 2592 per
// mission of the family (built by the host at create; mgx_mission_row in mgx_kernels.h maps the task word to the row), so the kernel is a copy: a block owns a group of 4 envs (4*L floats is a whole number of
// 16-B quads, L itself is not), a lane one quad; inside the mission block a quad is one 4-byte-aligned dwordx4 read
// of the L2-resident pattern and one non-temporal 16-B store.
namespace {
struct __attribute__((packed, aligned(4))) FlatQuad { uint32_t a, b, c, d; };

__device__ __forceinline__ int flat_row(const uint2 *rec, int64_t env, int family)
{
    return family == MGX_MF_CONST ? 0 : mgx_mission_row(family, rec[env].y >> 16);
}

__global__ __launch_bounds__(256) void k_flat(const uint8_t *__restrict__ tri, const uint2 *__restrict__ rec, const float *__restrict__ pattern,
                                               float *__restrict__ out, int64_t n, int img, int family)
{
    const int L = img + MGX_FLAT_MISSION;
    const int64_t env_base = (int64_t)blockIdx.y * 4;
    const int n_here = n - env_base < 4 ? (int)(n - env_base) : 4;
    const int q = blockIdx.x * 256 + threadIdx.x; // quad of this group's n_here*L floats
    const int g0 = 4 * q;
    if (g0 >= n_here * L) return;
    float *dst = out + env_base * L + g0;
    const int e = (g0 >= L) + (g0 >= 2 * L) + (g0 >= 3 * L);
    const int off = g0 - e * L;
    if (off >= img && off + 4 <= L) { // the whole quad lies in one env's mission block
        const int row = flat_row(rec, env_base + e, family);
        const FlatQuad v = *reinterpret_cast<const FlatQuad *>(pattern + (size_t)row * MGX_FLAT_MISSION + (off - img));
        nt_store16(reinterpret_cast<uint4 *>(dst), make_uint4(v.a, v.b, v.c, v.d));
        return;
    }
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { // image bytes, or a quad that straddles two envs
        const int g = g0 + j;
        const int ej = (g >= L) + (g >= 2 * L) + (g >= 3 * L), oj = g - ej * L;
        float x = 0.f;
        if (g < n_here * L) {
            if (oj < img) x = (float)tri[(env_base + ej) * img + oj];
            else x = pattern[(size_t)flat_row(rec, env_base + ej, family) * MGX_FLAT_MISSION + (oj - img)];
        }
        v[j] = x;
    }
    if (g0 + 3 < n_here * L) nt_store16(reinterpret_cast<uint4 *>(dst), make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    else
        for (int j = 0; j < 4 && g0 + j < n_here * L; j++) dst[j] = v[j];
}
} // namespace

hipError_t mgx_launch_flat(const uint8_t *tri, const uint2 *rec, const float *pattern, float *out, int64_t n, int img, int family, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const int quads = (4 * (img + MGX_FLAT_MISSION) + 3) / 4;
    const int64_t groups = (n + 3) / 4;
    if (groups > 65535 * 1024ll) return hipErrorInvalidValue;
    // grid.y is limited to 65535: fold larger batches into several launches
    for (int64_t g0 = 0; g0 < groups; g0 += 65535) {
        const int64_t gy = groups - g0 < 65535 ? groups - g0 : 65535;
        const int64_t e0 = g0 * 4;
        hipLaunchKernelGGL(k_flat, dim3((unsigned)((quads + 255) / 256), (unsigned)gy), dim3(256), 0, st, tri + e0 * img, rec + e0, pattern,
                           out + e0 * (img + MGX_FLAT_MISSION), n - e0 < gy * 4 ? n - e0 : gy * 4, img, family);
    }
    return hipGetLastError();
}

hipError_t mgx_preload_epilogue_kernels()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_flat));
}
