// tools/fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access shapes of the verify kernel (measurement
// tool): gathers of 24 bytes per lane (one 16-byte + one 8-byte buffer load) at known positions of a 1 GiB buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p -- tools/fetch_calib
// Kernels: gather_sparse (one 24-byte read per 256 bytes, never straddling a 128-byte line: N distinct lines),
// gather_dense (one per 44 bytes), gather_158 (one per 158 bytes: cfg3's candidate density), stream16 (the wide
// coalesced read whose FETCH_SIZE the guide says is halved).  The program prints the bytes each kernel really asks for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int STRIDE>
__global__ __launch_bounds__(256) void gather(const unsigned char *buf, unsigned long long n_req, unsigned *sink) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(buf), 0, (int)0xfffffff0u, 0x00020000);
    unsigned acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n_req; i += (unsigned long long)gridDim.x * 256) {
        unsigned pos = (unsigned)(i * STRIDE);
        if (STRIDE == 256) pos += (unsigned)((i * 2654435761ull) >> 20) % 96u; // inside the first 128-byte line of its 256
        pos &= ~3u;
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)pos, 0, 0);
        const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(pos + 16u), 0, 0);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void stream16(const uint4 *buf, unsigned long long n16, unsigned *sink) {
    unsigned acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const unsigned long long n = 1ull << 30;
    unsigned char *buf; unsigned *sink;
    CHECK(hipMalloc(&buf, n + 4096)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, n + 4096));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        gather<256><<<4096, 256>>>(buf, n / 256, sink);
        gather<44><<<4096, 256>>>(buf, n / 44, sink);
        gather<158><<<4096, 256>>>(buf, n / 158, sink);
        stream16<<<4096, 256>>>((const uint4 *)buf, n / 16, sink);
        CHECK(hipDeviceSynchronize());
    }
    printf("requests: gather<256> %llu (distinct 128-B lines: the same), gather<44> %llu (distinct lines %llu), gather<158> %llu (distinct lines ~%llu), stream16 bytes %llu\n",
           n / 256, n / 44, n / 128, n / 158, (unsigned long long)((n / 158) * 1.15), n);
    return 0;
}
