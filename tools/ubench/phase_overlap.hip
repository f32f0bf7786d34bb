// Micro-benchmark (tuning aid, not product): k_step's skeleton = [load 92 B/env] -> [~1,100 VALU instructions] -> [store 160 B/env],
// one wave per 64-env tile, to see which arrangement lets the three phases of different waves overlap.
//   hipcc --offload-arch=gfx950 -O3 -o phase_overlap phase_overlap.hip && ./phase_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int RQ = 368, WQ = 640; // 16-byte quads per tile in / out

__device__ __forceinline__ u4 compute(u4 a, int iters, uint32_t sel)
{
    // per iteration: 8 full-rate + 8 half-rate VALU instructions in 4 independent chains
    for (int it = 0; it < iters; it++) {
        asm volatile("v_add_u32 %0, %0, %1\n v_perm_b32 %2, %2, %3, %4\n v_add_u32 %1, %1, %2\n v_perm_b32 %3, %3, %0, %4\n"
                     "v_xor_b32 %0, %0, %3\n v_perm_b32 %2, %2, %1, %4\n v_add_u32 %1, %1, %0\n v_perm_b32 %3, %3, %2, %4\n"
                     "v_add_u32 %0, %0, %1\n v_perm_b32 %2, %2, %3, %4\n v_add_u32 %1, %1, %2\n v_perm_b32 %3, %3, %0, %4\n"
                     "v_xor_b32 %0, %0, %3\n v_perm_b32 %2, %2, %1, %4\n v_add_u32 %1, %1, %0\n v_perm_b32 %3, %3, %2, %4"
                     : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) : "v"(sel));
    }
    return a;
}

__device__ __forceinline__ u4 load_tile(const u4 *src, int lane)
{
    u4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < (RQ + 63) / 64; i++) { const int q = i * 64 + lane; if (q < RQ) acc ^= src[q]; }
    return acc;
}
__device__ __forceinline__ void store_tile(u4 *dst, int lane, u4 acc, int i0, int i1)
{
#pragma unroll
    for (int i = i0; i < i1; i++) { const int q = i * 64 + lane; if (q < WQ) { u4 v = acc; v.x += i; __builtin_nontemporal_store(v, &dst[q]); } }
}

// VARIANT 0: load | compute | store.  1: s_setprio from the wave's slot on its SIMD (older slot = higher).  2: compute in two halves,
// half of the stores after each.  3: two tiles per wave, software-pipelined (both loads first; stores of A drain under compute of B).
// 4: four tiles per wave, pipelined with a one-tile prefetch.
template <int VARIANT>
__global__ __launch_bounds__(256) void k(const u4 *__restrict__ in, u4 *__restrict__ out, int n_tiles, int iters, uint32_t sel, int tail_block0)
{
    extern __shared__ uint32_t dyn_lds[]; // only to bound the resident blocks per CU as k_step's tile images do
    if ((int)blockIdx.x >= tail_block0) asm volatile("s_setprio 3");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * (blockDim.x >> 6) + wv;
    constexpr int TPW = VARIANT == 3 ? 2 : VARIANT == 4 ? 4 : 1;
    if (w * TPW >= n_tiles) return;
    if (VARIANT == 1) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw)); // wave slot on its SIMD
        if ((hw & 7u) < 2u) asm volatile("s_setprio 3");
        else if ((hw & 7u) < 4u) asm volatile("s_setprio 2");
        else if ((hw & 7u) < 6u) asm volatile("s_setprio 1");
    }
    if (TPW == 1) {
        u4 acc = load_tile(in + (size_t)w * RQ, lane);
        acc.x ^= __shfl_xor(acc.x, 1);
        if (VARIANT == 2) {
            acc = compute(acc, iters / 2, sel);
            store_tile(out + (size_t)w * WQ, lane, acc, 0, 5);
            acc = compute(acc, iters - iters / 2, sel);
            store_tile(out + (size_t)w * WQ, lane, acc, 5, 10);
        } else {
            acc = compute(acc, iters, sel);
            store_tile(out + (size_t)w * WQ, lane, acc, 0, 10);
        }
    } else {
        // interleaved tile assignment so that concurrently running waves touch neighbouring tiles
        const int n_w = (n_tiles + TPW - 1) / TPW;
        u4 nxt = load_tile(in + (size_t)w * RQ, lane);
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int tile = w + t * n_w;
            u4 acc = nxt;
            if (t + 1 < TPW && tile + n_w < n_tiles) nxt = load_tile(in + (size_t)(tile + n_w) * RQ, lane);
            if (tile < n_tiles) {
                acc.x ^= __shfl_xor(acc.x, 1);
                acc = compute(acc, iters, sel);
                store_tile(out + (size_t)tile * WQ, lane, acc, 0, 10);
            }
        }
    }
}

template <int VARIANT>
static void run(const char *name, int n_envs, int wpb, int citers, u4 *in, u4 *out, int shmem = 0, int tail = 0)
{
    const int n_tiles = n_envs / 64, iters = 300;
    constexpr int TPW = VARIANT == 3 ? 2 : VARIANT == 4 ? 4 : 1;
    const int n_w = (n_tiles + TPW - 1) / TPW;
    dim3 grid((n_w + wpb - 1) / wpb), block(wpb * 64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; i++) hipLaunchKernelGGL((k<VARIANT>), grid, block, shmem, 0, in, out, n_tiles, citers, 0x03020104u, tail ? (int)grid.x - tail : 0x7fffffff);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<VARIANT>), grid, block, shmem, 0, in, out, n_tiles, citers, 0x03020104u, tail ? (int)grid.x - tail : 0x7fffffff);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s envs %8d wpb %d valu %5d lds %5d tail %4d  %7.2f us/launch\n", name, n_envs, wpb, citers * 16, shmem, tail, ms * 1e3 / iters);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const size_t cap = (size_t)2 << 30;
    u4 *in, *out;
    CK(hipMalloc(&in, cap)); CK(hipMalloc(&out, cap));
    CK(hipMemset(in, 1, cap)); CK(hipMemset(out, 0, cap));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int n : {458752, 524288, 1048576}) { // k_step<9,9>'s residency: 7 blocks of 4 waves per CU (21.5 KB of LDS each)
        run<0>("7 blocks/CU", n, 4, 70, in, out, 21504);
        run<0>("7 blocks/CU, tail 512 at priority 3", n, 4, 70, in, out, 21504, 512);
        run<0>("7 blocks/CU, tail 256 at priority 3", n, 4, 70, in, out, 21504, 256);
        run<0>("7 blocks/CU, no compute", n, 4, 0, in, out, 21504);
        run<0>("4 blocks/CU", n, 4, 70, in, out, 36000);
    }
    const int sizes[] = {524288, 1048576};
    for (int n : sizes) {
        for (int c : {0, 35, 70, 140}) run<0>("load|compute|store", n, 4, c, in, out);
        run<0>("load|compute|store, 1 wave/blk", n, 1, 70, in, out);
        run<0>("load|compute|store, 2 waves/blk", n, 2, 70, in, out);
        run<1>("+ s_setprio by SIMD slot", n, 4, 70, in, out);
        run<2>("compute/store in two halves", n, 4, 70, in, out);
        run<3>("2 tiles/wave pipelined", n, 4, 70, in, out);
        run<4>("4 tiles/wave pipelined", n, 4, 70, in, out);
        run<4>("4 tiles/wave pipelined, 2 waves/blk", n, 2, 70, in, out);
    }
    return 0;
}
