// Probe: semantics of __builtin_amdgcn_global_load_lds (16 B/lane) on gfx950:
//  (1) LDS destination = wave-uniform base + lane*16 (per-lane SOURCE is arbitrary)
//  (2) exec-masked lanes are skipped (no write), destination still base + lane*16
//  (3) counted s_waitcnt vmcnt + raw s_barrier ordering
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void probe(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const int* __restrict__ perm, int active) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0xEE;
    __syncthreads();
    // each wave fills its own 1 KB: lane reads 16 B from a PERMUTED source chunk
    const unsigned char* g = src + (size_t)perm[wave * 64 + lane] * 16;
    if (lane < active)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_ptr_t)(lds + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) dst[i] = lds[i];
}

int main() {
    std::vector<unsigned char> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)(i / 16);   // chunk id as content
    std::vector<int> perm(256);
    for (int i = 0; i < 256; ++i) perm[i] = (i * 37 + 11) % 256;
    unsigned char *ds, *dd; int* dp;
    hipMalloc(&ds, 4096); hipMalloc(&dd, 4096); hipMalloc(&dp, 1024);
    hipMemcpy(ds, h.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(dp, perm.data(), 1024, hipMemcpyHostToDevice);
    for (int active : {64, 20}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, ds, dd, dp, active);
        std::vector<unsigned char> o(4096);
        hipMemcpy(o.data(), dd, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int b = 0; b < 16; ++b) {
            unsigned char want = (l < active) ? (unsigned char)perm[w * 64 + l] : 0xEE;
            if (o[w * 1024 + l * 16 + b] != want) ++bad;
        }
        printf("active=%d mismatches=%d\n", active, bad);
    }
    return 0;
}
