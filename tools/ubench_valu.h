// ubench_valu.h -- VALU issue-rate probe (developer tool): cycles per wave64 instruction on one SIMD
// for the opcodes the Sobel core is made of, at 1..8 resident waves per SIMD.
#pragma once

namespace ubv {

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void k_valu(unsigned *sink, int iters, unsigned long long *cycles)
{
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    unsigned b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    const unsigned two = 0x00020002u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 1) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %5, %6\n v_perm_b32 %2, %2, %6, %7\n v_perm_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 2) { REP16(asm volatile("v_pk_mad_u16 %0, %0, %8, %4\n v_pk_mad_u16 %1, %1, %8, %5\n v_pk_mad_u16 %2, %2, %8, %6\n v_pk_mad_u16 %3, %3, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(two));) }
        if (OP == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %7, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 4) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 5) { REP16(asm volatile("v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %2, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 6) { REP16(asm volatile("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %5, 1, %1\n v_cvt_pk_u8_f32 %2, %6, 2, %2\n v_cvt_pk_u8_f32 %3, %7, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        if (OP == 7) {   // packed f32 fma: 2 results per lane per instruction (64-bit register pairs)
            typedef float fl2 __attribute__((ext_vector_type(2)));
            fl2 p0 = {f0, f1}, p1 = {f2, f3}, q0 = {(float)b0, (float)b1}, q1 = {(float)b2, (float)b3};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %3, %2\n v_pk_fma_f32 %0, %0, %3, %2\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(q0), "v"(q1));)
            f0 = p0.x; f1 = p0.y; f2 = p1.x; f3 = p1.y;
        }
        if (OP == 8) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 9) { REP16(asm volatile("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %5, %6, %1\n v_dot2_f32_f16 %2, %6, %7, %2\n v_dot2_f32_f16 %3, %7, %4, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 10) { REP16(asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %5, %6\n v_pk_fma_f16 %2, %2, %6, %7\n v_pk_fma_f16 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 11) { REP16(asm volatile("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 12) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %5\n v_cvt_f32_ubyte2 %2, %6\n v_cvt_f32_ubyte3 %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 13) { REP16(asm volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %5\n v_sub_f32 %2, %2, %6\n v_sub_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 14) { REP16(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 15) { REP16(asm volatile("v_cvt_u32_f32 %0, %4\n v_cvt_u32_f32 %1, %5\n v_cvt_u32_f32 %2, %6\n v_cvt_u32_f32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        if (OP == 16) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %5, %6\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 17) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 8, %4\n v_lshl_or_b32 %1, %1, 8, %5\n v_lshl_or_b32 %2, %2, 8, %6\n v_lshl_or_b32 %3, %3, 8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 18) { REP16(asm volatile("v_bfe_u32 %0, %4, 8, 8\n v_bfe_u32 %1, %5, 16, 8\n v_bfe_u32 %2, %6, 8, 8\n v_bfe_u32 %3, %7, 16, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 19) { REP16(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %6\n v_min_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 20) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 21) { REP16(asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %5\n v_mul_i32_i24 %2, %2, %6\n v_mul_i32_i24 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 22) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 8\n v_alignbit_b32 %1, %1, %5, 8\n v_alignbit_b32 %2, %2, %6, 8\n v_alignbit_b32 %3, %3, %7, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 23) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %6\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 24) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa %2, %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_add_u32_sdwa %3, %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ __builtin_bit_cast(unsigned, f0 + f1 + f2 + f3);
}

template <int OP>
static void run_valu(const char *name)
{
    const int iters = 2000;   // x 64 instructions per iteration per wave
    unsigned *sink; unsigned long long *cyc;
    CK(hipMalloc(&sink, 256 * 8 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 4 * 8));
    printf("%-22s", name);
    for (int wps : {1, 2, 4}) {      // waves per SIMD = blocks per CU (a 256-thread block puts one wave on each SIMD)
        const int grid = 256 * wps;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k_valu<OP><<<grid, 256>>>(sink, 10, cyc);
        CK(hipEventRecord(a));
        k_valu<OP><<<grid, 256>>>(sink, iters, cyc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h(grid * 4);
        CK(hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double ninstr = (double)iters * 64;
        // s_memtime ticks at 100 MHz-derived shader clock; report both: wall ns per instr per SIMD, and ticks
        const double ns_per_instr_simd = ms * 1e6 / (ninstr * wps);
        printf("  w%d: %.2f", wps, ns_per_instr_simd);
    }
    printf("\n");
    CK(hipFree(sink)); CK(hipFree(cyc));
}

}  // namespace ubv

static void bench_valu()
{
    using namespace ubv;
    printf("\n== VALU issue rate: wall ns per wave64 instruction per SIMD (1 cycle = 0.42 ns at 2.4 GHz) ==\n");
    run_valu<8>("v_add_u32");
    run_valu<0>("v_pk_add_u16");
    run_valu<2>("v_pk_mad_u16");
    run_valu<1>("v_perm_b32");
    run_valu<3>("v_fma_f32");
    run_valu<7>("v_pk_fma_f32");
    run_valu<4>("v_sqrt_f32");
    run_valu<5>("v_cvt_f32_i32_sdwa");
    run_valu<6>("v_cvt_pk_u8_f32");
    run_valu<9>("v_dot2_f32_f16");
    run_valu<10>("v_pk_fma_f16");
    run_valu<11>("v_mov_b32_dpp wave_sh");
    run_valu<12>("v_cvt_f32_ubyteN");
    run_valu<13>("v_sub_f32");
    run_valu<14>("v_cvt_f32_i32");
    run_valu<15>("v_cvt_u32_f32");
    run_valu<16>("v_mad_u32_u24");
    run_valu<17>("v_lshl_or_b32");
    run_valu<18>("v_bfe_u32");
    run_valu<19>("v_min_f32");
    run_valu<20>("v_mul_f32");
    run_valu<21>("v_mul_i32_i24");
    run_valu<22>("v_alignbit_b32");
    run_valu<23>("v_and_or_b32");
    run_valu<24>("v_add_u32_sdwa(byte)");
}
