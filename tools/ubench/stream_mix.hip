// Micro-benchmark (tuning aid, not product): the memory skeleton of k_step with no compute.  One wave per 64-env tile reads RB bytes
// per env (state: re-read every launch, so cache-resident where it fits) and writes WB bytes per env (observation: non-temporal,
// dependent on the loads), in k_step's launch shape.  Prints the span per launch over back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// RQ / WQ: 16-byte quads per tile read / written.  NT: 0 plain stores, 1 non-temporal.  WAVES: waves per block.
template <int RQ, int WQ, int NT>
__global__ __launch_bounds__(256) void k(const u4 *__restrict__ in, u4 *__restrict__ out, int n_tiles)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= n_tiles) return;
    const u4 *src = in + (size_t)tile * RQ;
    u4 *dst = out + (size_t)tile * WQ;
    u4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < (RQ + 63) / 64; i++) {
        const int q = i * 64 + lane;
        if (q < RQ) acc ^= src[q];
    }
    // every lane's stores depend on every lane's loads, as the view gather does
    acc.x ^= __shfl_xor(acc.x, 1); acc.y ^= __shfl_xor(acc.y, 2); acc.z ^= __shfl_xor(acc.z, 4); acc.w ^= __shfl_xor(acc.w, 8);
#pragma unroll
    for (int i = 0; i < (WQ + 63) / 64; i++) {
        const int q = i * 64 + lane;
        if (q < WQ) {
            u4 v = acc; v.x += i;
            if (NT) __builtin_nontemporal_store(v, &dst[q]); else dst[q] = v;
        }
    }
}

// the FullyObs direct form's shape: a 256-thread block per tile
template <int RQ, int WQ>
__global__ __launch_bounds__(256) void kb(const u4 *__restrict__ in, u4 *__restrict__ out, int n_tiles)
{
    const int tile = blockIdx.x, t = threadIdx.x;
    const u4 *src = in + (size_t)tile * RQ;
    u4 *dst = out + (size_t)tile * WQ;
    u4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < (RQ + 255) / 256; i++) { const int q = i * 256 + t; if (q < RQ) acc ^= src[q]; }
    __shared__ uint32_t sh[4];
    if ((t & 63) == 0) sh[t >> 6] = acc.x;
    __syncthreads();
    acc.y ^= sh[0]; // phase B depends on phase A (wave 0)
#pragma unroll
    for (int i = 0; i < (WQ + 255) / 256; i++) { const int q = i * 256 + t; if (q < WQ) { u4 v = acc; v.x += i; __builtin_nontemporal_store(v, &dst[q]); } }
}
template <int RQ, int WQ>
static void runb(const char *name, int n_envs, u4 *in, u4 *out)
{
    const int n_tiles = n_envs / 64, iters = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((kb<RQ, WQ>), dim3(n_tiles), dim3(256), 0, 0, in, out, n_tiles);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((kb<RQ, WQ>), dim3(n_tiles), dim3(256), 0, 0, in, out, n_tiles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, bytes = (double)n_tiles * (RQ + WQ) * 16;
    printf("%-34s envs %8d  block/tile  %7.2f us/launch  %6.0f GB/s (R %5.1f MB + W %5.1f MB)\n", name, n_envs, us, bytes / us * 1e-3, n_tiles * RQ * 16e-6, n_tiles * WQ * 16e-6);
    fflush(stdout);
}

// k_step's real streams for an 8x8 / 9x9 tile: reads = cells (CQ quads, 16 B per lane) + record (8 B per lane) + action (1 B per lane);
// writes = observation (588 quads, non-temporal) + reward (4 B per lane, nt) + done (1 B per lane, nt) + record (8 B per lane, plain)
// VAR: 0 as k_step; 1 done stored plain (not nt); 2 no done stream; 3 no reward stream; 4 no record write; 5 reward + done plain
template <int CQ, int VAR = 0>
__global__ __launch_bounds__(256) void ks(const u4 *__restrict__ cells, const uint2 *__restrict__ rec_in, const uint8_t *__restrict__ act,
                                          u4 *__restrict__ obs, float *__restrict__ reward, uint8_t *__restrict__ done, uint2 *__restrict__ rec_out, int n_tiles)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= n_tiles) return;
    const size_t env = (size_t)tile * 64 + lane;
    uint2 r = rec_in[env];
    uint32_t a = __builtin_nontemporal_load(&act[env]);
    const u4 *src = cells + (size_t)tile * CQ;
    u4 acc = {r.x, r.y, a, 0};
#pragma unroll
    for (int i = 0; i < (CQ + 63) / 64; i++) { const int q = i * 64 + lane; if (q < CQ) acc ^= src[q]; }
    acc.x ^= __shfl_xor(acc.x, 1); acc.y ^= __shfl_xor(acc.y, 2);
    if (VAR == 5) reward[env] = __uint_as_float(acc.x & 0x3fffffffu);
    else if (VAR != 3) __builtin_nontemporal_store(__uint_as_float(acc.x & 0x3fffffffu), &reward[env]);
    if (VAR == 1 || VAR == 5) done[env] = (uint8_t)(acc.y & 1u);
    else if (VAR != 2) __builtin_nontemporal_store((uint8_t)(acc.y & 1u), &done[env]);
    if (VAR != 4) rec_out[env] = make_uint2(acc.x, acc.y);
    u4 *dst = obs + (size_t)tile * 588;
#pragma unroll
    for (int i = 0; i < 10; i++) { const int q = i * 64 + lane; if (q < 588) { u4 v = acc; v.x += i; __builtin_nontemporal_store(v, &dst[q]); } }
}
template <int CQ, int VAR = 0>
static void runs(const char *name, int n_envs, u4 *in, u4 *out)
{
    const int n_tiles = n_envs / 64, iters = 300;
    // carve the streams out of the two big buffers
    const u4 *cells = in; const uint2 *rec_in = reinterpret_cast<const uint2 *>(in + (size_t)n_tiles * CQ + 64);
    const uint8_t *act = reinterpret_cast<const uint8_t *>(rec_in + (size_t)n_envs + 64);
    u4 *obs = out; float *reward = reinterpret_cast<float *>(out + (size_t)n_tiles * 588 + 64);
    uint8_t *done = reinterpret_cast<uint8_t *>(reward + (size_t)n_envs + 64);
    uint2 *rec_out = const_cast<uint2 *>(rec_in); // the record is updated in place, as in k_step
    dim3 grid((n_tiles + 3) / 4), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; i++) hipLaunchKernelGGL((ks<CQ, VAR>), grid, block, 0, 0, cells, rec_in, act, obs, reward, done, rec_out, n_tiles);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((ks<CQ, VAR>), grid, block, 0, 0, cells, rec_in, act, obs, reward, done, rec_out, n_tiles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s envs %8d  k_step's own streams  %7.2f us/launch\n", name, n_envs, ms * 1e3 / iters);
    fflush(stdout);
}

template <int RQ, int WQ, int NT>
static void run(const char *name, int n_envs, int wpb, u4 *in, u4 *out)
{
    const int n_tiles = n_envs / 64, iters = 300;
    dim3 grid((n_tiles + wpb - 1) / wpb), block(wpb * 64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; i++) hipLaunchKernelGGL((k<RQ, WQ, NT>), grid, block, 0, 0, in, out, n_tiles);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<RQ, WQ, NT>), grid, block, 0, 0, in, out, n_tiles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, bytes = (double)n_tiles * (RQ + WQ) * 16;
    printf("%-34s envs %8d  wpb %d  %7.2f us/launch  %6.0f GB/s (R %5.1f MB + W %5.1f MB)\n", name, n_envs, wpb, us, bytes / us * 1e-3,
           n_tiles * RQ * 16e-6, n_tiles * WQ * 16e-6);
    fflush(stdout);
}

int main()
{
    const size_t cap = (size_t)4 << 30;
    u4 *in, *out;
    CK(hipMalloc(&in, cap)); CK(hipMalloc(&out, cap));
    CK(hipMemset(in, 1, cap)); CK(hipMemset(out, 0, cap));
    // LavaCrossing 9x9: read 84 (cells, padded) + 8 (record) + 1 (action) ~ 92 B/env = 368 quads per tile; write 147 + 8 + 4 + 1 = 160 B/env = 640 quads
    const int sizes[] = {524288, 1048576};
    for (int n : sizes) {
        run<368, 640, 1>("lava 9x9 skeleton, nt stores", n, 4, in, out);
        run<368, 640, 0>("lava 9x9 skeleton, plain stores", n, 4, in, out);
        run<368, 640, 1>("lava 9x9 skeleton, nt, 1 wave/blk", n, 1, in, out);
        run<0, 640, 1>("writes only (160 B/env), nt", n, 4, in, out);
        run<368, 4, 1>("reads only (92 B/env)", n, 4, in, out);
        run<1008, 1008, 1>("copy 252 B/env each way, nt", n, 4, in, out);
    }
    // FullyObs 16x16: read 256 + 8 + 1 B/env ~ 272 B = 1088 quads per tile; write 768 + 8 + 5 ~ 784 B = 3136 quads
    for (int n : {262144, 1048576}) runb<1088, 3136>("FullyObs 16x16 skeleton", n, in, out);
    // FullyObs 19x19: read 364 + 9, write 1083 + 13
    for (int n : {131072, 524288}) runb<1492, 4384>("FullyObs 19x19 skeleton", n, in, out);
    // Empty-8x8 partial: read 64 + 8 + 1 = 73 B/env ~ 292 quads, write 640
    for (int n : {524288, 1048576}) run<292, 640, 1>("empty 8x8 skeleton, nt stores", n, 4, in, out);
    // the same bytes as k_step moves, in k_step's own seven streams (3 in, 4 out; 147-B observations: a tile's 9,408 B straddle lines)
    for (int n : {524288, 1048576}) { runs<256>("8x8 tile, real streams", n, in, out); runs<336>("9x9 tile, real streams", n, in, out); }
    runs<256, 1>("8x8, done stored plain", 1048576, in, out);
    runs<256, 2>("8x8, no done stream", 1048576, in, out);
    runs<256, 3>("8x8, no reward stream", 1048576, in, out);
    runs<256, 4>("8x8, no record write", 1048576, in, out);
    runs<256, 5>("8x8, reward + done plain", 1048576, in, out);
    runs<256, 0>("8x8 tile, real streams (again)", 1048576, in, out);
    return 0;
}
