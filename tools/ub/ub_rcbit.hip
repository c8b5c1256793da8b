// micro-benchmark: what ONE binary decision of the decoder's range-coder chain costs a lone wavefront (gfx950), piece by piece.
// An iteration walks an 8-level tree (probabilities in a vector register, lane = node) like a literal; cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/ub/ub_rcbit.hip -o tools/ub/ub_rcbit && gpurun -- tools/ub/ub_rcbit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
struct S { uint32_t range, code, feed; };
__device__ __forceinline__ void renorm(S &s) { s.range = RFL(s.range << 8); s.feed = RFL(s.feed * 1664525u + 1013904223u); s.code = RFL((s.code << 8) | (s.feed >> 24)); }
template <int CMP>
__device__ __forceinline__ uint32_t step(S &s, uint32_t p, uint32_t v)
{
    uint32_t bound, vn, r2, c2;
    if (CMP)
        asm("s_lshr_b32 %0, %4, 12\n\ts_mul_i32 %0, %0, %6\n\ts_sub_u32 %2, %4, %0\n\ts_sub_u32 %3, %5, %0\n\ts_cmp_lt_u32 %5, %0\n\t"
            "s_cselect_b32 %2, %0, %2\n\ts_cselect_b32 %3, %5, %3\n\ts_addc_u32 %1, %7, %7"
            : "=&s"(bound), "=&s"(vn), "=&s"(r2), "=&s"(c2) : "s"(s.range), "s"(s.code), "s"(p), "s"(v) : "scc");
    else
        asm("s_lshr_b32 %0, %4, 12\n\ts_mul_i32 %0, %0, %6\n\ts_sub_u32 %2, %4, %0\n\ts_sub_u32 %3, %5, %0\n\t"
            "s_cselect_b32 %2, %0, %2\n\ts_cselect_b32 %3, %5, %3\n\ts_addc_u32 %1, %7, %7"
            : "=&s"(bound), "=&s"(vn), "=&s"(r2), "=&s"(c2) : "s"(s.range), "s"(s.code), "s"(p), "s"(v) : "scc");
    s.range = r2; s.code = c2;
    return vn;
}
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *g, unsigned long long *out, uint32_t seed)
{
    const uint32_t pv = 1024u + ((threadIdx.x * 2654435761u) >> 21);      // a probability per lane (node)
    S s; s.range = RFL(0xFFFFFFFFu - seed); s.code = RFL(seed * 2654435761u); s.feed = RFL(seed);
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 4096; it++) {
        uint32_t v = 1;
#pragma unroll
        for (int b = 0; b < 6; b++) {     // six levels: node indices stay below 64
            if (MODE == 0) { v = step<1>(s, 2048u + (uint32_t)b, v); }                                             // the 8-instruction chain alone (no renormalisation: wrong, but timed)
            if (MODE == 1) { v = step<0>(s, 2048u + (uint32_t)b, v); }                                             // the 7-instruction chain alone
            if (MODE == 2) { if (__builtin_expect(s.range < (1u << 24), 0)) renorm(s); v = step<0>(s, 2048u + (uint32_t)b, v); }   // + renormalisation test
            if (MODE == 3) { if (__builtin_expect(s.range < (1u << 24), 0)) renorm(s); v = step<0>(s, rl(pv, v), v); }             // + readlane of the node's probability: the decoder's step
            if (MODE == 4) { v = step<0>(s, rl(pv, v), v); }                                                       // readlane, no renormalisation test
            if (MODE == 5) { const uint32_t p0 = rl(pv, 2 * v), p1 = rl(pv, 2 * v + 1); (void)p0; (void)p1;        // both children fetched ahead, chosen by a select
                             if (__builtin_expect(s.range < (1u << 24), 0)) renorm(s); v = step<0>(s, rl(pv, v), v); }
            if (MODE == 6) { if (__builtin_expect(s.range < (1u << 24), 0)) renorm(s); v = step<0>(s, rl(pv, v), v); acc += v; acc ^= acc >> 3; }   // + two independent SALU a bit
            if (MODE == 7) { if (__builtin_expect(s.range < (1u << 24), 0)) renorm(s); v = step<0>(s, rl(pv, v), v);
                             asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(pv)); }   // + two independent VALU a bit
        }
        acc += v;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[MODE] = t1 - t0; g[MODE] = acc + s.range + s.code; }
}
int main()
{
    uint32_t *g; unsigned long long *out;
    (void)hipMalloc(&g, 64 * 4); (void)hipMalloc(&out, 64 * 8);
    unsigned long long h[16];
    const char *names[] = {"chain, 8 SALU (with s_cmp)", "chain, 7 SALU (borrow = bit)", "7 SALU + renormalisation test", "7 SALU + test + v_readlane (the decoder's step)",
                           "7 SALU + v_readlane, no test", "as 3 + two more readlanes", "as 3 + 2 independent SALU", "as 3 + 2 independent VALU"};
    for (int rep = 0; rep < 2; rep++) {
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, g, out, 12345u + rep); (void)hipDeviceSynchronize();
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    }
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 8; m++) printf("%-52s %8.1f cycles per binary decision\n", names[m], h[m] / 4096.0 / 6.0);
    return 0;
}
