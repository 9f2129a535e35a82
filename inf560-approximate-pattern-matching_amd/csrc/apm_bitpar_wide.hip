/*
 * apm_bitpar_wide.hip -- the BITPAR kernel for patterns of 129 .. 512 bytes (bit-vector columns of 8 and 16 words) and the
 * kernel for their truncated tail windows (/root/reference/src/sequential.c:131-134).  Same code as the narrow forms
 * (apm_bitpar.h, apm_core.h); a translation unit of its own so that it compiles beside apm_kernels.hip.
 */
#include "apm_internal.h"
#include "apm_core.h"
#include "apm_bitpar.h"

hipError_t apm_launch_bitpar_wide(const ApmScanArgs &a, unsigned n_tiles, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(apm_bitpar_kernel<true>, dim3(n_tiles), dim3(APM_BLOCK), lds_bytes, s, a);
    return hipGetLastError();
}

// The same for 128 < m <= 512: 16-word columns, up to 511 truncated windows per pattern, one per thread of a 512-thread
// workgroup; Eq table of 256 x 16 words in LDS.  A launch of its own (its registers would cost the sieve kernels, where
// the short tails ride, their occupancy); it replaces the one-column GENERIC kernel, whose serial global-memory DP
// took 8 - 30 ms per call for these few windows.
template <int W>
__device__ __forceinline__ void apm_tail_wide_body(const ApmTailArgs &a, const ApmPatDesc &d, uint32_t *s_eq, uint8_t *s_txt, int tid) {
    constexpr int NT = 512; // text bytes staged (the workgroup's threads)
    const int m = (int)d.m;
    const uint8_t *pat = a.bytes + d.byte_off;
    for (int i = tid; i < 256 * W; i += 512) s_eq[i] = 0u;
    const int64_t first_trunc = max(a.jb, a.nrel - m + 1);
    const int64_t t0 = a.nrel - NT > 0 ? a.nrel - NT : 0;
    s_txt[tid] = (t0 + tid < a.nrel) ? a.text[t0 + tid] : (uint8_t)0;
    __syncthreads();
    if (tid < m) atomicOr(&s_eq[(int)pat[tid] * W + (tid >> 5)], 1u << (tid & 31));
    __syncthreads();
    const int64_t j = first_trunc + tid;
    const bool valid = j < a.je; // (j < nrel - k <= nrel: at least one byte)
    const int size = valid ? (int)(a.nrel - j) : 0; // 1 .. m-1
    const int lo = valid ? (int)(j - t0) : 0;
    uint32_t pv[W], mv[W];
    bp_init<W>(pv, mv);
    for (int x = 0; x < m - 1; ++x) {
        if (x < size) {
            uint32_t eq[W];
            const uint4 *row = reinterpret_cast<const uint4 *>(s_eq + (int)s_txt[lo + x] * W);
#pragma unroll
            for (int q = 0; q < W / 4; ++q) {
                const uint4 v = row[q];
                eq[4 * q] = v.x; eq[4 * q + 1] = v.y; eq[4 * q + 2] = v.z; eq[4 * q + 3] = v.w;
            }
            bp_step<W>(pv, mv, eq);
        }
    }
    const bool hit = valid && bp_distance<W>(pv, mv, size, size) <= a.k;
    if (a.pos.out && hit) apm_push_pos(a.pos, j);
    const uint32_t cnt = apm_wave_count(hit);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counts[d.index], (unsigned long long)cnt);
}

__global__ __launch_bounds__(512) void apm_tail_wide_kernel(ApmTailArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_eq[256 * 16];
    __shared__ uint8_t s_txt[512 + 16]; // the last <= 512 text bytes
    const ApmPatDesc d = a.pats[blockIdx.x];
    if (d.m <= 256) apm_tail_wide_body<8>(a, d, s_eq, s_txt, (int)threadIdx.x); // (8-word columns: half the work per text byte)
    else apm_tail_wide_body<16>(a, d, s_eq, s_txt, (int)threadIdx.x);
}

hipError_t apm_launch_tail_wide(const ApmTailArgs &a, int n_pats, hipStream_t s) {
    if (n_pats <= 0 || a.je <= a.jb) return hipSuccess;
    hipLaunchKernelGGL(apm_tail_wide_kernel, dim3((unsigned)n_pats), dim3(512), 0, s, a);
    return hipGetLastError();
}

