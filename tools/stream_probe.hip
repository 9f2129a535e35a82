// tools/stream_probe.hip -- measurement tool: HBM read rate of a 1 GiB buffer under the access shapes the scan kernels could use
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip && tools/stream_probe
// Shapes: grid-stride 16 B per lane (A); the sieve's (B: persistent 512-thread workgroups, a wave takes 4 KiB -- four 1 KiB
// chunks, the next four in flight -- then jumps by the whole grid); the same with 8 / 16 KiB per wave and round (C, D); one
// workgroup per 32 KiB, not persistent (E); a contiguous region per workgroup (F); B plus a write of 1/16 of the bytes (G).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void shape_a(const uint4 *buf, unsigned long long n16, unsigned *sink) {
    unsigned acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// KPW = KiB per wave and round; WRITE: one dword per lane and 4 KiB goes back out
template <int KPW, int WRITE, bool CONTIG>
__global__ __launch_bounds__(512, 8) void shape_b(const unsigned char *buf, unsigned long long n, unsigned *sink, unsigned *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long W = (unsigned long long)gridDim.x * 8, nround = n / (1024ull * KPW);
    const unsigned long long w = (unsigned long long)blockIdx.x * 8 + wv;
    unsigned long long c = CONTIG ? w * (nround / W) : w;
    const unsigned long long c_end = CONTIG ? (w + 1) * (nround / W) : nround, step = CONTIG ? 1 : W;
    unsigned acc = 0;
    unsigned wacc[4] = {0, 0, 0, 0};
    u32x4 r[KPW];
    auto load = [&](unsigned long long cc) {
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(buf) + (cc < c_end ? (cc * KPW + j) * 1024 : 0), 0, cc < c_end ? 1024 : 0, 0x00020000);
            r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, 0, 0);
        }
    };
    load(c);
    for (; c < c_end; c += step) {
        u32x4 v[KPW];
#pragma unroll
        for (int j = 0; j < KPW; ++j) v[j] = r[j];
        load(c + step);
        unsigned x = 0;
#pragma unroll
        for (int j = 0; j < KPW; ++j) x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        acc ^= x;
        if (WRITE == 1) {
#pragma unroll
            for (int j = 0; j < KPW / 4; ++j) out[(c * (KPW / 4) + j) * 64 + lane] = x;
        } else if (WRITE == 2) { // a quarter of the bytes: 16 lanes write per round
            if (lane < 16) out[c * 16 + lane] = x;
        } else if (WRITE == 3) {
            __builtin_nontemporal_store(x, &out[c * 64 + lane]);
        } else if (WRITE == 4) { // one 1 KiB store per wave and four rounds
            wacc[(c / step) & 3] = x;
            if (((c / step) & 3) == 3) reinterpret_cast<uint4 *>(out)[(c / step / 4 * W + w) * 64 + lane] = make_uint4(wacc[0], wacc[1], wacc[2], wacc[3]);
        } else if (WRITE == 5) { // 1/64 of the bytes through a quarter-wave store of 16 B... one dword per lane every 4th round
            if (((c / step) & 3) == 3) out[(c / step / 4 * W + w) * 64 + lane] = x;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(512) void shape_e(const unsigned char *buf, unsigned *sink) { // one workgroup per 32 KiB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned char *p = buf + ((unsigned long long)blockIdx.x * 8 + wv) * 4096;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(p), 0, 4096, 0x00020000);
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, 1024 * j + 16 * lane, 0, 0); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const unsigned long long n = 1ull << 30;
    unsigned char *buf; unsigned *sink, *out;
    CHECK(hipMalloc(&buf, n + 4096)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&out, n / 16 + 4096));
    CHECK(hipMemset(buf, 1, n + 4096));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto &&launch) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-58s %.4f ms  %.0f GB/s\n", name, ms, n / ms / 1e6);
    };
    time("A grid-stride 16 B/lane, 4096 x 256", [&] { shape_a<<<4096, 256>>>((const uint4 *)buf, n / 16, sink); });
    time("A grid-stride 16 B/lane, 16384 x 256", [&] { shape_a<<<16384, 256>>>((const uint4 *)buf, n / 16, sink); });
    time("B sieve shape: 4 KiB per wave and round, 4 wg/CU", [&] { shape_b<4, 0, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("B' the same, 2 wg/CU", [&] { shape_b<4, 0, false><<<ncu * 2, 512>>>(buf, n, sink, out); });
    time("C 8 KiB per wave and round, 4 wg/CU", [&] { shape_b<8, 0, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("C' 8 KiB per wave and round, 2 wg/CU", [&] { shape_b<8, 0, false><<<ncu * 2, 512>>>(buf, n, sink, out); });
    time("E one 512-thread workgroup per 32 KiB", [&] { shape_e<<<(unsigned)(n / 32768), 512>>>(buf, sink); });
    time("F contiguous region per wave, 4 KiB steps, 4 wg/CU", [&] { shape_b<4, 0, true><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("G = B + 1/16 written back", [&] { shape_b<4, 1, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("H = C + 1/16 written back", [&] { shape_b<8, 1, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("I = B + 1/64 written back (16 lanes per round)", [&] { shape_b<4, 2, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("J = G with nontemporal stores", [&] { shape_b<4, 3, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("K = B + 1/16 written as 1 KiB per wave and 4 rounds", [&] { shape_b<4, 4, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    time("L = B + 1/64 written as 256 B per wave and 4 rounds", [&] { shape_b<4, 5, false><<<ncu * 4, 512>>>(buf, n, sink, out); });
    return 0;
}
