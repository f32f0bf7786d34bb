// Micro-benchmark (tuning aid, not product): the memory skeleton of k_step's GATHER form (partial view on grids past 16x16) with no
// compute.  Per env: record 8 B + action 1 B in (coalesced), the 7x8-byte view window out of the env's S-byte row (x-major: 7 columns,
// H bytes apart, at a per-env offset), 147 B observation + reward + done + record out.  Variants of the WINDOW read only:
//   0  seven unaligned 8-byte loads per lane (what k_step<0,0,3,7> does)
//   1  the lane reads its window's whole span [off, off + 6H + 8) as 16-byte aligned dwordx4 loads
//   2  four lanes per 64-byte segment: a wave instruction reads 16 envs x one aligned 64-B segment each; via LDS to the owning lane
//   3  the whole row streamed, coalesced (the staged form's traffic: floor of "read everything")
//   4  no window read at all (floor of the other streams)
//   hipcc --offload-arch=gfx950 -O3 -o gather_window gather_window.hip && ./gather_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct __attribute__((packed)) U8 { uint32_t a, b; };

template <int H, int S, int VAR>
__global__ __launch_bounds__(256) void kg(const uint8_t *__restrict__ cells, const uint32_t *__restrict__ offs, uint2 *__restrict__ rec, const uint8_t *__restrict__ act,
                                          u4 *__restrict__ obs, float *__restrict__ reward, uint8_t *__restrict__ done, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= n_tiles) return;
    constexpr int SPAN = 6 * H + 8, NSEG = (SPAN + 63 + 63) / 64, SLOT = NSEG * 64 + 16; // LDS slot per env (VAR 2), odd multiple of 16 B
    uint8_t *lds = smem + (size_t)wv * 64 * SLOT;
    const size_t env = (size_t)tile * 64 + lane;
    uint2 r = rec[env];
    uint32_t a = __builtin_nontemporal_load(&act[env]);
    const uint32_t off = offs[env] + (r.x & 1u); // (depends on the record, as the view depends on the pose)
    const uint8_t *row = cells + env * S;
    u64 acc = r.x ^ ((u64)r.y << 32) ^ a;
    if (VAR == 0) {
#pragma unroll
        for (int k = 0; k < 7; k++) { const U8 v = *reinterpret_cast<const U8 *>(row + off + k * H); acc ^= (u64)v.a | ((u64)v.b << 32); }
    } else if (VAR == 1) {
        const uint32_t lo = (uint32_t)(env * S + off) & ~15u;
        constexpr int NL = (SPAN + 15 + 15) / 16;
        const u4 *p = reinterpret_cast<const u4 *>(cells + lo);
        u4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < NL; k++) acc ^= (u64)(v[k].x ^ v[k].z) | ((u64)(v[k].y ^ v[k].w) << 32);
    } else if (VAR == 2) {
        const uint32_t lo = (uint32_t)(env * S + off) & ~63u; // first aligned 64-B segment of this env's span (byte offset in `cells`)
        u4 v[NSEG * 4];
#pragma unroll
        for (int j = 0; j < NSEG * 4; j++) {
            const int e = (j & 3) * 16 + (lane >> 2), seg = j >> 2;
            const uint32_t lo_e = __shfl(lo, e);
            v[j] = *reinterpret_cast<const u4 *>(cells + lo_e + seg * 64 + (lane & 3) * 16);
        }
#pragma unroll
        for (int j = 0; j < NSEG * 4; j++) {
            const int e = (j & 3) * 16 + (lane >> 2), seg = j >> 2;
            *reinterpret_cast<u4 *>(lds + e * SLOT + seg * 64 + (lane & 3) * 16) = v[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t d = (uint32_t)(env * S + off) - lo;
#pragma unroll
        for (int k = 0; k < 7; k++) { const U8 w = *reinterpret_cast<const U8 *>(lds + lane * SLOT + d + k * H); acc ^= (u64)w.a | ((u64)w.b << 32); }
    } else if (VAR == 3) {
        const u4 *src = reinterpret_cast<const u4 *>(cells + (size_t)tile * 64 * S);
        constexpr int RQ = 4 * S;
        u4 x = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < (RQ + 63) / 64; i++) { const int q = i * 64 + lane; if (q < RQ) x ^= src[q]; }
        acc ^= (u64)(x.x ^ x.z) | ((u64)(x.y ^ x.w) << 32);
    }
    uint32_t ax = (uint32_t)acc, ay = (uint32_t)(acc >> 32);
    ax ^= __shfl_xor(ax, 1); ay ^= __shfl_xor(ay, 2);
    __builtin_nontemporal_store(__uint_as_float(ax & 0x3fffffffu), &reward[env]);
    __builtin_nontemporal_store((uint8_t)(ay & 1u), &done[env]);
    rec[env] = make_uint2(r.x ^ (ax & 1u), r.y);
    u4 *dst = obs + (size_t)tile * 588;
#pragma unroll
    for (int i = 0; i < 10; i++) { const int q = i * 64 + lane; if (q < 588) { u4 v = {ax, ay, (uint32_t)i, 0}; __builtin_nontemporal_store(v, &dst[q]); } }
}

template <int H, int S, int VAR>
static void run(const char *name, int n_envs, uint8_t *cells, uint32_t *offs, uint2 *rec, uint8_t *act, u4 *obs, float *reward, uint8_t *done)
{
    const int n_tiles = n_envs / 64, iters = 200;
    constexpr int SPAN = 6 * H + 8, NSEG = (SPAN + 63 + 63) / 64, SLOT = NSEG * 64 + 16;
    const size_t shmem = VAR == 2 ? (size_t)4 * 64 * SLOT : 0;
    if (shmem > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kg<H, S, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    dim3 grid((n_tiles + 3) / 4), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((kg<H, S, VAR>), grid, block, shmem, 0, cells, offs, rec, act, obs, reward, done, n_tiles);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((kg<H, S, VAR>), grid, block, shmem, 0, cells, offs, rec, act, obs, reward, done, n_tiles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s envs %8d  H %2d S %3d  %8.2f us/launch\n", name, n_envs, H, S, ms * 1e3 / iters);
    fflush(stdout);
}

template <int H, int S>
static void all(int n_envs, uint8_t *cells, uint32_t *offs, uint2 *rec, uint8_t *act, u4 *obs, float *reward, uint8_t *done)
{
    // window offsets: column x0 in [0, W-7], row y0 in [0, H-8] -> x0 * H + y0 (the env's row holds W = H columns)
    std::vector<uint32_t> h((size_t)n_envs);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n_envs; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t x0 = (uint32_t)(s % (H - 6)), y0 = (uint32_t)((s >> 20) % (H - 7)); h[i] = x0 * H + y0; }
    CK(hipMemcpy(offs, h.data(), (size_t)n_envs * 4, hipMemcpyHostToDevice));
    run<H, S, 4>("no window read (other streams only)", n_envs, cells, offs, rec, act, obs, reward, done);
    run<H, S, 0>("7 unaligned 8-byte loads per lane", n_envs, cells, offs, rec, act, obs, reward, done);
    run<H, S, 1>("span as aligned dwordx4 loads per lane", n_envs, cells, offs, rec, act, obs, reward, done);
    run<H, S, 2>("4 lanes per 64-B segment, via LDS", n_envs, cells, offs, rec, act, obs, reward, done);
    run<H, S, 3>("whole row streamed (coalesced)", n_envs, cells, offs, rec, act, obs, reward, done);
}

int main()
{
    const int NMAX = 1048576;
    uint8_t *cells, *act, *done; uint32_t *offs; uint2 *rec; u4 *obs; float *reward;
    CK(hipMalloc(&cells, (size_t)NMAX * 628 + 4096)); CK(hipMemset(cells, 1, (size_t)NMAX * 628 + 4096));
    CK(hipMalloc(&offs, (size_t)NMAX * 4)); CK(hipMalloc(&rec, (size_t)NMAX * 8)); CK(hipMemset(rec, 0, (size_t)NMAX * 8));
    CK(hipMalloc(&act, NMAX)); CK(hipMemset(act, 2, NMAX));
    CK(hipMalloc(&obs, (size_t)NMAX / 64 * 588 * 16)); CK(hipMalloc(&reward, (size_t)NMAX * 4)); CK(hipMalloc(&done, NMAX));
    for (int n : {262144, 1048576}) {
        all<19, 364>(n, cells, offs, rec, act, obs, reward, done);
        all<25, 628>(n, cells, offs, rec, act, obs, reward, done);
        all<16, 256>(n, cells, offs, rec, act, obs, reward, done);
    }
    return 0;
}
