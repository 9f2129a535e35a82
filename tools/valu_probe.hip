// tools/valu_probe.hip -- measures the integer-VALU issue rate of gfx950 for the instruction kinds the scan kernels
// are made of, at 1, 2, 4 and 8 waves per SIMD (measurement tool, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip && tools/valu_probe
// Every kernel runs ITER iterations of 32 instructions over 8 independent register chains (an instruction
// depends on the one 8 places back, far beyond the ALU latency), stamped with s_memtime around the loop.
// Output per (instruction, waves/SIMD): cycles per wave-instruction as one wave sees it, cycles per wave-instruction
// per SIMD (= the issue cost: wave cycles / (waves per SIMD x instructions)), and chip-wide lane-ops/s from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define PROBE_KERNEL(NAME, ASM8)                                                                               \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *cycles, unsigned *sink, unsigned seed) {  \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, \
                 a6 = a0 * 17u, a7 = a0 * 19u, b = seed * 2654435761u + 1u, c = seed ^ 0x00010001u;            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                            \
        for (int i = 0; i < ITER; ++i) {                                                                        \
            asm volatile(ASM8 ASM8 ASM8 ASM8                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                         : "v"(b), "v"(c));                                                                     \
        }                                                                                                       \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                            \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                   \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678u) sink[0] = a0;                               \
    }

#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define OP_ADD(r) "v_add_u32 " r ", " r ", %8\n\t"
#define OP_PKADD(r) "v_pk_add_u16 " r ", " r ", %8\n\t"
#define OP_PKMIN(r) "v_pk_min_u16 " r ", " r ", %8\n\t"
#define OP_AND(r) "v_and_b32 " r ", " r ", %8\n\t"
#define OP_MIN3(r) "v_min3_u32 " r ", " r ", %8, %9\n\t"
#define OP_BFE(r) "v_bfe_u32 " r ", " r ", 3, 13\n\t"
#define OP_ALIGNBIT(r) "v_alignbit_b32 " r ", %8, " r ", 6\n\t"
#define OP_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 1, %8\n\t"
#define OP_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n\t"
#define OP_BITOP3(r) "v_bitop3_b32 " r ", " r ", %8, %9 bitop3:0x96\n\t"
#define OP_DPP(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define OP_DOT4(r) "v_dot4_u32_u8 " r ", " r ", %8, %9\n\t"
#define OP_MULU24(r) "v_mul_u32_u24 " r ", " r ", %8\n\t"
#define OP_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n\t"
#define OP_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8\n\t"
#define OP_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 3, %8\n\t"
#define OP_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n\t"
#define OP_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8\n\t"
#define OP_ADDC(r) "v_add_co_u32 " r ", vcc, " r ", %8\n\t"
#define OP_LSHR(r) "v_lshrrev_b32 " r ", 3, " r "\n\t"
#define OP_LSHL(r) "v_lshlrev_b32 " r ", 1, " r "\n\t"
#define OP_LSHRV(r) "v_lshrrev_b32 " r ", %8, " r "\n\t"
#define OP_OR(r) "v_or_b32 " r ", " r ", %8\n\t"
#define OP_XOR(r) "v_xor_b32 " r ", " r ", %8\n\t"
#define OP_SUB(r) "v_sub_u32 " r ", " r ", %8\n\t"
#define OP_MINU(r) "v_min_u32 " r ", " r ", %8\n\t"
#define OP_MOV(r) "v_mov_b32 " r ", %8\n\t"
#define OP_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n\t"
#define OP_BFI(r) "v_bfi_b32 " r ", %8, " r ", %9\n\t"
#define OP_OR3(r) "v_or3_b32 " r ", " r ", %8, %9\n\t"
#define OP_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n\t"
#define OP_CMPSEL(r) "v_cmp_eq_u32 vcc, " r ", %8\n\tv_cndmask_b32 " r ", " r ", %9, vcc\n\t"
#define OP_PKLSHR(r) "v_pk_lshrrev_b16 " r ", 1, " r "\n\t"
#define OP_ALIGNBYTE(r) "v_alignbyte_b32 " r ", %8, " r ", 1\n\t"
#define OP_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", %8, " r "\n\t"

PROBE_KERNEL(k_add, R8(OP_ADD))
PROBE_KERNEL(k_pkadd, R8(OP_PKADD))
PROBE_KERNEL(k_pkmin, R8(OP_PKMIN))
PROBE_KERNEL(k_and, R8(OP_AND))
PROBE_KERNEL(k_min3, R8(OP_MIN3))
PROBE_KERNEL(k_bfe, R8(OP_BFE))
PROBE_KERNEL(k_alignbit, R8(OP_ALIGNBIT))
PROBE_KERNEL(k_lshlor, R8(OP_LSHLOR))
PROBE_KERNEL(k_andor, R8(OP_ANDOR))
PROBE_KERNEL(k_bitop3, R8(OP_BITOP3))
PROBE_KERNEL(k_dpp, R8(OP_DPP))
PROBE_KERNEL(k_dot4, R8(OP_DOT4))
PROBE_KERNEL(k_mulu24, R8(OP_MULU24))
PROBE_KERNEL(k_mad24, R8(OP_MAD24))
PROBE_KERNEL(k_mullo, R8(OP_MULLO))
PROBE_KERNEL(k_lshladd, R8(OP_LSHLADD))
PROBE_KERNEL(k_perm, R8(OP_PERM))
PROBE_KERNEL(k_bcnt, R8(OP_BCNT))
PROBE_KERNEL(k_addc, R8(OP_ADDC))
PROBE_KERNEL(k_lshr, R8(OP_LSHR))
PROBE_KERNEL(k_lshl, R8(OP_LSHL))
PROBE_KERNEL(k_lshrv, R8(OP_LSHRV))
PROBE_KERNEL(k_or, R8(OP_OR))
PROBE_KERNEL(k_xor, R8(OP_XOR))
PROBE_KERNEL(k_sub, R8(OP_SUB))
PROBE_KERNEL(k_minu, R8(OP_MINU))
PROBE_KERNEL(k_mov, R8(OP_MOV))
PROBE_KERNEL(k_cndmask, R8(OP_CNDMASK))
PROBE_KERNEL(k_bfi, R8(OP_BFI))
PROBE_KERNEL(k_or3, R8(OP_OR3))
PROBE_KERNEL(k_add3, R8(OP_ADD3))
PROBE_KERNEL(k_cmpsel, R8(OP_CMPSEL))
PROBE_KERNEL(k_pklshr, R8(OP_PKLSHR))
PROBE_KERNEL(k_alignbyte, R8(OP_ALIGNBYTE))
PROBE_KERNEL(k_mbcnt, R8(OP_MBCNT))

// random LDS reads out of a 32 KiB table (the presence-bitmap access pattern): 16 reads in flight, then a wait
template <int BYTES>
__global__ __launch_bounds__(256) void k_lds(unsigned long long *cycles, unsigned *sink, unsigned seed) {
    __shared__ unsigned tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = i * 2654435761u + seed;
    __syncthreads();
    unsigned x = (seed + threadIdx.x) * 2246822519u, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER / 4; ++i) {
        unsigned v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 1664525u + 1013904223u;
            const unsigned a = (x >> 9) & 32767u;
            if constexpr (BYTES == 1) v[j] = reinterpret_cast<const unsigned char *>(tab)[a];
            else v[j] = tab[a >> 2];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

typedef void (*kern_t)(unsigned long long *, unsigned *, unsigned);
struct Probe { const char *name; kern_t fn; };

int main() {
    int dev = 0, ncu = 0;
    CHECK(hipSetDevice(dev));
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    int clk_khz = 0;
    CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev));
    printf("# device 0: %d CUs, max clock %.0f MHz; %d iterations x 32 instructions per wave\n", ncu, clk_khz / 1000.0, ITER);
    printf("# %-14s %10s %22s %26s %20s\n", "instruction", "waves/SIMD", "cyc/instr (one wave)", "cyc/wave-instr per SIMD", "chip lane-ops/s");
    const Probe probes[] = {{"v_add_u32", k_add}, {"v_pk_add_u16", k_pkadd}, {"v_pk_min_u16", k_pkmin}, {"v_and_b32", k_and},
                            {"v_min3_u32", k_min3}, {"v_bfe_u32", k_bfe}, {"v_alignbit_b32", k_alignbit}, {"v_lshl_or_b32", k_lshlor},
                            {"v_and_or_b32", k_andor}, {"v_bitop3_b32", k_bitop3}, {"v_mov_dpp shr1", k_dpp}, {"v_dot4_u32_u8", k_dot4},
                            {"v_mul_u32_u24", k_mulu24}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_lshl_add_u32", k_lshladd},
                            {"v_perm_b32", k_perm}, {"v_bcnt_u32_b32", k_bcnt}, {"v_add_co_u32", k_addc},
                            {"v_lshrrev imm", k_lshr}, {"v_lshlrev imm", k_lshl}, {"v_lshrrev vgpr", k_lshrv}, {"v_or_b32", k_or},
                            {"v_xor_b32", k_xor}, {"v_sub_u32", k_sub}, {"v_min_u32", k_minu}, {"v_mov_b32", k_mov},
                            {"v_cndmask_b32", k_cndmask}, {"v_bfi_b32", k_bfi}, {"v_or3_b32", k_or3}, {"v_add3_u32", k_add3},
                            {"v_cmp+cndmask/2", k_cmpsel}, {"v_pk_lshrrev_b16", k_pklshr}, {"v_alignbyte_b32", k_alignbyte},
                            {"v_mbcnt_lo", k_mbcnt}};
    unsigned long long *d_cyc;
    unsigned *d_sink;
    const int max_blocks = ncu * 8;
    CHECK(hipMalloc(&d_cyc, (size_t)max_blocks * 4 * 8));
    CHECK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Probe &p : probes) {
        for (int w : {1, 2, 4, 8}) {
            const int nblk = ncu * w; // 256-thread blocks: one wave per SIMD each
            p.fn<<<nblk, 256>>>(d_cyc, d_sink, 1u); // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            p.fn<<<nblk, 256>>>(d_cyc, d_sink, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t)nblk * 4);
            CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            const double n_instr = (double)ITER * 32.0;
            printf("  %-14s %10d %22.3f %26.3f %20.4e\n", p.name, w, med / n_instr, med / n_instr / w,
                   (double)nblk * 4 * 64 * n_instr / (ms * 1e-3));
        }
    }
    printf("# random LDS reads from a 32 KiB table (per read: 1 LCG mul+add, 1 bfe/shift, the read): cycles per read per wave, reads/s chip-wide\n");
    for (int bytes : {1, 4}) {
        for (int w : {1, 2, 4}) {
            const int nblk = ncu * w;
            auto fn = bytes == 1 ? k_lds<1> : k_lds<4>;
            fn<<<nblk, 256>>>(d_cyc, d_sink, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            fn<<<nblk, 256>>>(d_cyc, d_sink, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t)nblk * 4);
            CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double n_reads = (double)(ITER / 4) * 16.0;
            printf("  ds_read_%s random %2d waves/SIMD: %8.2f cycles per wave-read, %.3e wave-reads/s per CU, %.3e lane-reads/s chip\n",
                   bytes == 1 ? "u8 " : "b32", w, (double)h[h.size() / 2] / n_reads, (double)w * 4 * n_reads / (ms * 1e-3),
                   (double)nblk * 256 * n_reads / (ms * 1e-3));
        }
    }
    return 0;
}
