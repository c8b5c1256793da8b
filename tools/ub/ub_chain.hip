// micro-benchmark: latency of the instruction kinds the DP master chains together (gfx950, one lone wavefront).
// Each mode runs N dependent copies of a pattern per loop iteration; cycles per copy from s_memtime, loop overhead measured apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define RDL(x, l) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (int)(l)))
#define DPP(old, src, ctl) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), ctl, 0xF, 0xF, false))
constexpr int IT = 2048;
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *g, unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    __shared__ uint4 lds4[256];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (i * 2654435761u) & 4095u;
    for (int i = threadIdx.x; i < 256; i += 64) lds4[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    uint32_t s = RFL(seed), a = RFL(seed * 3) | 1, acc = 0;
    uint32_t v = threadIdx.x + seed, w = threadIdx.x * 7 + 1, lane = threadIdx.x & 63;
    uint32_t rI0 = lane, rI1 = lane * 3, n_tg = 0, n_hs = 0, n_w = 0, n_lp = 0, n_fw = 0;
    uint4 n_en = make_uint4(5, 0, 0, 0);
    uint16_t *ldsh = (uint16_t *)lds;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < IT; i++) {
        if (MODE == 0) { s += a; }                                                              // loop overhead + 1 SALU
#define R8(X) X X X X X X X X
        if (MODE == 1) { R8(v = v * 3 + w;) }                                                   // 8 dependent VALU mad
        if (MODE == 2) { R8(v = (v > w) ? v + 1 : w;) }                                         // 8 dependent v_cmp + v_cndmask
        if (MODE == 3) { R8(v = max(v, DPP(0, v, 0x111)) + 1;) }                                // 8 dependent DPP row_shr:1 max (+1)
        if (MODE == 4) { R8(s = RFL(v) + s; v += s;) }                                          // 8 x (readfirstlane -> SALU -> VALU)
        if (MODE == 5) { R8(s = RDL(v, 5) + s; v += s;) }                                       // 8 x (readlane const -> SALU -> VALU)
        if (MODE == 6) { R8(s = RDL(v, s & 63) + s; v += s;) }                                  // 8 x (readlane SGPR idx -> SALU -> VALU)
        if (MODE == 7) { R8({ uint64_t b = __ballot(v > w); s = (uint32_t)__builtin_ctzll(b | (1ull << 63)) + s; v += s; }) }   // ballot -> ff1 -> VALU
        if (MODE == 8) { R8(v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((v & 63) * 4), (int)w) + 1;) }   // 8 dependent ds_bpermute
        if (MODE == 9) { R8(v = lds[v & 4095];) }                                               // 8 dependent ds_read_b32 (per-lane address)
        if (MODE == 10) { R8({ uint4 q = lds4[v & 255]; v = q.x + q.w; }) }                     // 8 dependent ds_read_b128
        if (MODE == 11) { R8(v = DPP(0, v, 0x134); w = DPP(0, w, 0x134);) v += w; }             // 16 wave_rol:1 (2 chains)
        if (MODE == 12) { R8(acc += (uint32_t)__builtin_readcyclecounter();) }                  // 8 s_memtime
        if (MODE == 13) { R8(s = s * 1664525u + 1013904223u; if (s & 0x80000000u) acc += s;) } // 8 x (SALU + uniform branch ~50%)
        if (MODE == 14) { R8(asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(s), "s"(s & 63) : "m0"); s += a;) }
        if (MODE == 15) { R8(v = __builtin_amdgcn_perm(v, w, 0x01000504u) + 1;) }               // 8 dependent v_perm
        if (MODE == 16) { R8({ uint64_t m = (((uint64_t)v << 32) | w) >> (v & 63); v = (uint32_t)__builtin_ctzll(~m | (1ull << 63)) + w; }) }   // 64-bit shift + ctz chain
        if (MODE == 17) { R8(lds[(s & 1023)] = s; s = RFL(lds[(s + 1) & 1023]) + s;) }          // uniform LDS store + dependent uniform load
        if (MODE == 18) { R8(v += 1; w += 2; acc += 3; s += a;) }                               // 8 x (2 independent VALU + 2 SALU)
        if (MODE == 20) {
            // replica of the DP master's straight-line node (csc_kernels_dp4.inc): label read, mask -> rep length, rare tests,
            // one vector relax from a prefetched length table, literal edge, requests for the next node, ring rotate
            const uint32_t s_price = RFL(v), s_bs = RFL(w), s_i0 = RFL(rI0), s_i1 = RFL(rI1);
            const uint32_t state = (s_bs >> 16) & 63;
            const uint32_t row = ((volatile uint16_t *)ldsh)[state * 16 + (lane & 15)];
            const uint32_t off = (uint32_t)i - n_en.y;
            const uint64_t m = (((uint64_t)n_en.w << 32) | n_en.z) >> (off & 63u);
            const uint32_t room = 64u - (off & 63u);
            const uint32_t mm = (uint32_t)__builtin_ctzll(~m | (1ull << 63));
            const uint32_t rl = mm < room ? mm : room;
            const bool rare_v = lane < 4 && (rl >= 2u || off >= 64u || rl >= room || n_fw != 0);
            const uint32_t hs = RFL(n_hs);
            const bool rare_s = RFL(n_tg) != (uint32_t)i + 7 || ((hs >> 8) & 255) >= 16;
            const uint32_t hw = lane < 16 ? n_w : 0u;
            const uint32_t hd = hw & 0x3FFFFFFFu;
            const uint32_t slt = hd < 4 ? hd - 1 : 33u - (uint32_t)__builtin_clz(hd - 2);
            const uint32_t hcost = (slt > 2u ? slt + 2u : 2u) * 128u;
            const uint32_t cand = acc + hcost + (s_price + RDL(row, 4));
            const bool better = hw != 0 && cand < v;
            uint32_t nP = better ? cand : v, nB = better ? (s_bs + 1) : w, n0 = better ? (hw >> 30 | s_i0 << 16) : rI0, n1 = better ? (s_i0 >> 16 | s_i1 << 16) : rI1;
            const uint32_t c1 = s_price + RFL(n_lp) + RDL(row, 8);
            const bool b1 = lane == 1 && c1 < nP;
            nP = b1 ? c1 : nP; nB = b1 ? s_bs : nB; n0 = b1 ? s_i0 : n0; n1 = b1 ? s_i1 : n1;
            if (!rare_s && __ballot(rare_v) == 0) { v = nP; w = nB; rI0 = n0; rI1 = n1; s += hs >> 24; }
            const uint32_t rs = ((uint32_t)i + 1) & 255;
            n_tg = ((volatile uint32_t *)lds)[rs]; n_hs = ((volatile uint32_t *)lds)[256 + rs]; n_w = ((volatile uint32_t *)lds)[512 + rs * 8 + (lane & 7)];
            n_lp = ((volatile uint16_t *)ldsh)[2048 + rs];
            const uint32_t q0 = RDL(rI0, 1), q1 = RDL(rI1, 1);
            const uint32_t rid = lane < 4 ? (__builtin_amdgcn_perm(q1, q0, 0x0C0C0100u) & 255) : 0;
            { const volatile uint32_t *e4 = (const volatile uint32_t *)&lds4[rid]; n_en.x = e4[0]; n_en.y = e4[1]; n_en.z = e4[2]; n_en.w = e4[3]; } n_fw = ((volatile uint32_t *)lds)[3000 + rid];
            v = DPP(0, v, 0x134); w = DPP(0, w, 0x134); rI0 = DPP(0, rI0, 0x134); rI1 = DPP(0, rI1, 0x134);
        }
        if (MODE == 19) { R8(s = RDL(v, 3) + RDL(w, 4) + RDL(v, 9) + RDL(w, 11) + s;) v += s; } // 32 independent readlanes feeding SALU adds
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[MODE] = t1 - t0; }
    g[4096 + MODE * 64 + threadIdx.x] = s + acc + v + w + lds[7] + lane;
}
int main()
{
    uint32_t *g; unsigned long long *out;
    (void)hipMalloc(&g, 16384 * 4); (void)hipMalloc(&out, 64 * 8);
    (void)hipMemset(g, 0, 16384 * 4);
    unsigned long long h[32];
    const char *names[] = {"loop overhead + 1 SALU", "dependent VALU mad", "dependent v_cmp + v_cndmask", "dependent DPP row_shr max (+add)",
                           "readfirstlane -> SALU -> VALU", "readlane const -> SALU -> VALU", "readlane SGPR idx -> SALU -> VALU", "ballot -> ff1 -> SALU -> VALU",
                           "dependent ds_bpermute (+add)", "dependent ds_read_b32", "dependent ds_read_b128 (+add)", "2 x wave_rol:1 (2 chains)", "s_memtime",
                           "SALU + uniform branch ~50%", "m0 writelane + SALU", "dependent v_perm (+add)", "64-bit shift + ctz chain", "uniform LDS store + load",
                           "2 VALU + 2 SALU independent", "4 const readlanes + SALU adds", "REPLICA of the DP straight-line node"};
    const int NM = 21;
    for (int rep = 0; rep < 2; rep++) {
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, g, out, 12345u + rep); (void)hipDeviceSynchronize();
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20)
    }
    (void)hipMemcpy(h, out, sizeof(unsigned long long) * NM, hipMemcpyDeviceToHost);
    const double base = h[0] / (double)IT;
    printf("%-40s %8.1f cycles/iter\n", names[0], base);
    for (int m = 1; m < NM - 1; m++) printf("%-40s %8.1f cycles per copy (8 copies per iteration)\n", names[m], (h[m] / (double)IT - base) / 8.0);
    printf("%-40s %8.1f cycles per node\n", names[NM - 1], h[NM - 1] / (double)IT);
    return 0;
}
