// Micro-benchmark (tuning aid, not product): per-SIMD issue rate of the instruction kinds k_step is made of, at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t lds[256 * 33];
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b ^ c, f = d * 5;
    unsigned long long m = (unsigned long long)seed * 0x9E3779B97F4A7C15ull, m2 = ~m;
    for (int i = 0; i < 33; i++) lds[threadIdx.x * 33 + i] = a + i;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { // plain VOP2 adds, 4 independent chains
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 1) { // VOP3, 3 VGPR sources: v_perm_b32
                asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
            } else if (KIND == 2) { // v_cndmask with an SGPR-pair mask (VOP3)
                asm volatile("v_cndmask_b32 %0, %0, %1, %4\n v_cndmask_b32 %2, %2, %3, %4\n v_cndmask_b32 %1, %1, %2, %4\n v_cndmask_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));
            } else if (KIND == 3) { // VOP3 encoding of a 2-source op
                asm volatile("v_add_u32_e64 %0, %0, %1\n v_add_u32_e64 %2, %2, %3\n v_add_u32_e64 %1, %1, %2\n v_add_u32_e64 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 4) { // pure SALU
                asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0" : "+s"(m), "+s"(m2) : : "scc");
            } else if (KIND == 5) { // ds_read_u8, lane stride 33 dwords (conflict free), 4 in flight
                uint32_t r0, r1, r2, r3;
                asm volatile("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:5\n ds_read_u8 %2, %4 offset:10\n ds_read_u8 %3, %4 offset:15\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"((threadIdx.x * 33 * 4) & 0xFFFF));
                a += r0 + r1; b += r2 + r3;
            } else if (KIND == 6) { // mix: 3 VALU + 1 SALU
                asm volatile("v_add_u32 %0, %0, %1\n s_and_b64 %4, %4, %5\n v_add_u32 %2, %2, %3\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(m) : "s"(m2) : "scc");
            } else if (KIND == 7) { // v_perm with an SGPR selector (2 VGPR sources)
                asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
            } else if (KIND == 8) { // v_cmp_e64 -> sgpr then v_cndmask_e64 from that sgpr
                unsigned long long t0;
                asm volatile("v_cmp_gt_u32 %0, %1, %2\n v_cndmask_b32 %1, %1, %2, %0\n v_cmp_gt_u32 %0, %3, %4\n v_cndmask_b32 %3, %3, %4, %0" : "=&s"(t0), "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 9) { // VOPC to VCC + VOP2 cndmask on VCC
                asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
            } else if (KIND == 10) { // v_bfe_u32 with inline constants (1 VGPR source, VOP3)
                asm volatile("v_bfe_u32 %0, %1, 4, 3\n v_bfe_u32 %1, %2, 4, 3\n v_bfe_u32 %2, %3, 4, 3\n v_bfe_u32 %3, %0, 4, 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 11) { // v_lshl_or_b32 with a constant shift (2 VGPR sources, VOP3)
                asm volatile("v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %2, %2, 8, %3\n v_lshl_or_b32 %1, %1, 8, %2\n v_lshl_or_b32 %3, %3, 8, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 12) { // v_or3_b32, 3 VGPR sources
                asm volatile("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 13) { // v_and_b32 with a 32-bit literal (VOP2 + literal = 8 bytes)
                asm volatile("v_and_b32 %0, 0x0f0f0f0f, %1\n v_and_b32 %1, 0x07070707, %2\n v_and_b32 %2, 0x0f0f0f0f, %3\n v_and_b32 %3, 0x07070707, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 14) { // SDWA byte select
                asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 15) { // v_bfi_b32 with an SGPR mask (2 VGPR)
                asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
            } else if (KIND == 16) { // v_mad_u32_u24 3 VGPR
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 17) { // v_mul_u32_u24 VOP2
                asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 18) { // ds_read_b32 x4 (conflict free)
                uint32_t r0, r1, r2, r3;
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:8\n ds_read_b32 %2, %4 offset:16\n ds_read_b32 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"((threadIdx.x * 33 * 4) & 0xFFFF));
                a += r0 + r1; b += r2 + r3;
            } else if (KIND == 19) { // ds_write_b32 x4 (conflict free)
                asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:8\n ds_write_b32 %0, %3 offset:16\n ds_write_b32 %0, %4 offset:24\n s_waitcnt lgkmcnt(0)" : : "v"((threadIdx.x * 33 * 4) & 0xFFFF), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
            } else if (KIND == 20) { // v_alignbyte_b32 3 VGPR
                asm volatile("v_alignbyte_b32 %0, %0, %1, %4\n v_alignbyte_b32 %2, %2, %3, %4\n v_alignbyte_b32 %1, %1, %2, %4\n v_alignbyte_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
            } else if (KIND == 21) { // DPP row_shr mov
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
            if (KIND == 22) {
                asm volatile("v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 5, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 7, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 23) {
                asm volatile("v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %1, 5, %2\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %3, 7, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 24) {
                asm volatile("v_or_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_or_b32 %1, %1, %2\n v_xor_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 25) { // v_cmp to vcc only
                asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %1, %2\n v_cmp_eq_u32 vcc, %3, %0" : : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            } else if (KIND == 26) { // v_cndmask_e32 only
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 27) {
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 28) {
                asm volatile("v_and_or_b32 %0, %0, 15, %1\n v_and_or_b32 %2, %2, 15, %3\n v_and_or_b32 %1, %1, 15, %2\n v_and_or_b32 %3, %3, 15, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 29) {
                asm volatile("v_sub_u32 %0, %0, %1\n v_min_u32 %2, %2, %3\n v_sub_u32 %1, %1, %2\n v_max_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 30) { // v_cmp_e64 to sgpr only
                unsigned long long t0, t1;
                asm volatile("v_cmp_gt_u32 %0, %2, %3\n v_cmp_gt_u32 %1, %4, %5\n v_cmp_eq_u32 %0, %3, %4\n v_cmp_eq_u32 %1, %5, %2" : "=&s"(t0), "=&s"(t1) : "v"(a), "v"(b), "v"(c), "v"(d));
            } else if (KIND == 31) { // add with sgpr / inline operands
                asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, 7, %1\n v_and_b32 %2, %4, %2\n v_lshl_add_u32 %3, %3, 2, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + (uint32_t)m + (uint32_t)m2;
}
template <int KIND> void run(const char *name, uint32_t *out, int per_instr_valu, int per_instr_salu, int lds_ops)
{
    for (int bpc = 2; bpc <= 8; bpc *= 4) { // blocks of 4 waves per CU = waves per SIMD
        const int iters = 200, grid = 256 * bpc;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 10, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double groups = (double)iters * 16 * bpc; // 4-instruction groups per SIMD
        fflush(stdout); printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per 4-instr group per SIMD (= %.2f cycles @2.1GHz per instr)\n", name, bpc, ms, ms * 1e6 / groups, ms * 1e6 / groups * 2.1 / 4);
    }
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    printf("start\n");
    run<22>("v_lshlrev_b32", out, 4, 0, 0);
    run<23>("v_lshrrev_b32", out, 4, 0, 0);
    run<24>("v_or/xor_b32", out, 4, 0, 0);
    run<25>("v_cmp -> vcc", out, 4, 0, 0);
    run<30>("v_cmp -> sgpr", out, 4, 0, 0);
    run<26>("v_cndmask_e32 vcc", out, 4, 0, 0);
    run<27>("v_mov_b32", out, 4, 0, 0);
    run<28>("v_and_or_b32", out, 4, 0, 0);
    run<29>("v_sub/min/max", out, 4, 0, 0);
    run<31>("add sgpr/inline, lshl_add", out, 4, 0, 0);
    run<4>("s_and/s_or b64 x4", out, 0, 4, 0);
    run<6>("3 v_add + 1 s_and", out, 3, 1, 0);
    run<5>("ds_read_u8 x4 + wait", out, 0, 0, 4);
    run<18>("ds_read_b32 x4 + wait", out, 0, 0, 4);
    run<19>("ds_write_b32 x4 + wait", out, 0, 0, 4);
    return 0;
}
