// micro-benchmark: cost of wave-uniform stores (all 64 lanes, same address) vs single-lane stores,
// to LDS and to global memory, and of a dependent LDS load->readfirstlane chain.  One wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void k(uint32_t *g, unsigned long long *out, int mode)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    uint32_t x = 1, idx = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 4096; i++) {
        if (mode == 0) { lds[idx & 4095] = x; }                                   // all lanes, same address, LDS b32
        else if (mode == 1) { if (threadIdx.x == 0) lds[idx & 4095] = x; }       // lane 0 only
        else if (mode == 2) { g[idx & 4095] = x; }                                // all lanes, same address, global
        else if (mode == 3) { if (threadIdx.x == 0) g[idx & 4095] = x; }
        else if (mode == 4) { idx = __builtin_amdgcn_readfirstlane(lds[idx & 4095]); }   // dependent LDS load chain
        else if (mode == 5) { ((uint2 *)lds)[idx & 2047] = make_uint2(x, idx); }  // all lanes same address b64
        else if (mode == 6) { idx = __builtin_amdgcn_readfirstlane(g[idx & 4095]); }     // dependent global load chain (L2 hit)
        else if (mode == 7) { lds[(idx & 4095)] = x; idx = __builtin_amdgcn_readfirstlane(lds[(idx + 7) & 4095]); }   // store + dependent load
        x = x * 3 + 1; idx += 17;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[mode] = t1 - t0; g[4096 + mode] = x + idx + lds[5]; }
}
int main()
{
    uint32_t *g; unsigned long long *out;
    hipMalloc(&g, 8192 * 4); hipMalloc(&out, 64 * 8);
    hipMemset(g, 0, 8192 * 4);
    unsigned long long h[16];
    const char *names[] = {"lds b32 all-lanes same addr", "lds b32 lane0", "global all-lanes same addr", "global lane0", "dependent lds load chain",
                           "lds b64 all-lanes same addr", "dependent global load chain", "lds store + dependent load"};
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 8; m++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, out, m); hipDeviceSynchronize(); }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 8; m++) printf("%-32s %8.1f cycles/iter\n", names[m], h[m] / 4096.0);
    return 0;
}
