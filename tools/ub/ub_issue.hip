// micro-benchmark: what one lone wavefront pays per instruction kind (gfx950).  Each kernel runs a
// 4096-iteration loop around a small body; cycles/iter from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *g, unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t s = RFL(seed), a = RFL(seed * 3), acc = 0;
    uint32_t v = threadIdx.x + seed;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 4096; i++) {
        if (MODE == 0) { s = s * 1664525u + 1013904223u; }                                   // 1 dependent scalar mul-add
        if (MODE == 1) { s = s * 1664525u + 1013904223u; s ^= s >> 7; s += a; s ^= s << 3; s = s * 3 + 1; s ^= s >> 11; s += 5; s ^= a; }   // ~12 dependent SALU
        if (MODE == 2) { s = s * 1664525u + 1013904223u; if (s & 0x10000) acc += s; }        // uniform branch, 50% taken, tiny body
        if (MODE == 3) { s = s * 1664525u + 1013904223u; acc += (s & 0x10000) ? s : 0; }     // the same as a select
        if (MODE == 4) { s = s * 1664525u + 1013904223u; if (s & 0x10000) { acc += s; acc ^= acc >> 3; acc *= 5; acc += a; } else { acc -= s; acc ^= acc << 2; acc *= 7; acc -= a; } }  // if/else bodies
        if (MODE == 5) { v = v * 1664525u + 1013904223u; }                                   // 1 dependent VALU mul-add
        if (MODE == 6) { v = v * 1664525u + 1013904223u; s = RFL(v) + s; v += s; }           // VALU -> readfirstlane -> SALU -> VALU round trip
        if (MODE == 7) { s = RFL(lds[s & 4095]); }                                           // dependent LDS load chain
        if (MODE == 8) { lds[s & 4095] = s; s = s * 1664525u + 1013904223u; }                // all-lane same-address LDS store
        if (MODE == 9) { if (threadIdx.x == 0) lds[s & 4095] = s; s = s * 1664525u + 1013904223u; }   // lane-0 LDS store
        if (MODE == 10) { s = RFL(g[s & 4095]); }                                            // dependent global load chain (L2)
        if (MODE == 11) { g[s & 4095] = s; s = s * 1664525u + 1013904223u; }                 // all-lane same-address global store
        if (MODE == 12) { s = s * 1664525u + 1013904223u; uint32_t t = __builtin_amdgcn_readlane((int)v, (int)(s & 63)); acc += t; }   // readlane with SGPR index
        if (MODE == 13) { s = s * 1664525u + 1013904223u; switch (s >> 30) { case 0: acc += 1; break; case 1: acc ^= s; break; case 2: acc *= 3; break; default: acc -= s; } }  // 4-way uniform switch
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[MODE] = t1 - t0; g[4096 + MODE] = s + acc + v + lds[7]; }
}
int main()
{
    uint32_t *g; unsigned long long *out;
    (void)hipMalloc(&g, 8192 * 4); (void)hipMalloc(&out, 64 * 8);
    (void)hipMemset(g, 0, 8192 * 4);
    unsigned long long h[16];
    const char *names[] = {"1 dependent SALU mul-add", "12 dependent SALU", "uniform branch 50% tiny body", "same as select", "if/else 4-instr bodies",
                           "1 dependent VALU mul-add", "VALU->readfirstlane->SALU->VALU", "dependent LDS load chain", "LDS store all lanes same addr",
                           "LDS store lane 0", "dependent global load chain", "global store all lanes same addr", "readlane SGPR index", "4-way switch"};
    for (int rep = 0; rep < 2; rep++) {
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, g, out, 12345u + rep); (void)hipDeviceSynchronize();
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    }
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 14; m++) printf("%-36s %8.1f cycles/iter\n", names[m], h[m] / 4096.0);
    return 0;
}
