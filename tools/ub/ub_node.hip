// micro-benchmark: ablation of the DP master's straight-line node (csc_kernels_dp4.inc) on one lone wavefront.
// OFF bits switch pieces off to see what each costs; cycles per node from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef volatile __attribute__((address_space(3))) uint32_t lv32;
typedef volatile __attribute__((address_space(3))) uint16_t lv16;
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define RDL(x, l) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (int)(l)))
#define DPP(old, src, ctl) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), ctl, 0xF, 0xF, false))
constexpr int IT = 4096;
enum { NO_PREFETCH = 1, NO_RECSCALARS = 2, NO_DISTSLOT = 4, NO_LITERAL = 8, NO_BRANCH = 16, NO_REPMASK = 32, NO_ROWREAD = 64, NO_RELAX = 128, NO_ROTATE = 256, NO_LABELRFL = 512 };
template <int OFF>
__global__ __launch_bounds__(64) void k(uint32_t *g, unsigned long long *out, uint32_t seed, int slot)
{
    __shared__ uint32_t lds[8192];
    __shared__ uint4 lds4[256];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 0;
    for (int i = threadIdx.x; i < 256; i += 64) { lds[i] = i + 7; lds[256 + i] = (1u << 8) | (2u << 24) | 1u; lds4[i] = make_uint4(5, 0, 0, 0); }
    for (int i = threadIdx.x; i < 2048; i += 64) lds[512 + i] = (i & 7) >= 2 && (i & 7) < 4 ? 1000u + i : 0u;
    __syncthreads();
    uint16_t *ldsh = (uint16_t *)lds;
    uint32_t s = RFL(seed), acc = 100;
    uint32_t v = threadIdx.x == 0 ? 0 : 0xFFFFFFFFu, w = 0, lane = threadIdx.x & 63;
    uint32_t rI0 = 1 | (2 << 16), rI1 = 3 | (4 << 16), n_tg = 7, n_hs = 0, n_w = 0, n_lp = 0, n_fw = 0;
    uint4 n_en = make_uint4(5, 0, 0, 0);
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < IT; i++) {
        uint32_t s_price, s_bs, s_i0, s_i1;
        if (OFF & NO_LABELRFL) { s_price = s; s_bs = s * 3; s_i0 = s + 1; s_i1 = s + 2; }
        else { s_price = RFL(v); s_bs = RFL(w); s_i0 = RFL(rI0); s_i1 = RFL(rI1); }
        const uint32_t state = (s_bs >> 16) & 63;
        const uint32_t row = (OFF & NO_ROWREAD) ? lane * 3 : (uint32_t)((lv16 *)ldsh)[4096 + state * 16 + (lane & 15)];
        bool rare_v = false;
        if (!(OFF & NO_REPMASK)) {
            const uint32_t off = (uint32_t)i - n_en.y;
            const uint64_t m = (((uint64_t)n_en.w << 32) | n_en.z) >> (off & 63u);
            const uint32_t room = 64u - (off & 63u);
            const uint32_t mm = (uint32_t)__builtin_ctzll(~m | (1ull << 63));
            const uint32_t rl = mm < room ? mm : room;
            rare_v = lane < 4 && (rl >= 2u || n_fw != 0);
        }
        uint32_t hs = 0x02000101u;
        bool rare_s = false;
        if (!(OFF & NO_RECSCALARS)) { hs = RFL(n_hs); rare_s = RFL(n_tg) != (uint32_t)((i & 255) + 7) || ((hs >> 8) & 255) >= 16; }
        const uint32_t hw = lane < 16 ? n_w : 0u;
        const uint32_t hd = hw & 0x3FFFFFFFu;
        uint32_t hcost = 256;
        if (!(OFF & NO_DISTSLOT)) { const uint32_t slt = hd < 4 ? hd - 1 : 33u - (uint32_t)__builtin_clz(hd - 2); hcost = (slt > 2u ? slt + 2u : 2u) * 128u; }
        uint32_t nP = v, nB = w, n0 = rI0, n1 = rI1;
        if (!(OFF & NO_RELAX)) {
            const uint32_t cand = acc + hcost + (s_price + RDL(row, 4));
            const bool better = hw != 0 && cand < v;
            nP = better ? cand : v; nB = better ? (s_bs + 1) : w; n0 = better ? (hw >> 30 | s_i0 << 16) : rI0; n1 = better ? (s_i0 >> 16 | s_i1 << 16) : rI1;
        }
        if (!(OFF & NO_LITERAL)) {
            const uint32_t c1 = s_price + ((OFF & NO_PREFETCH) ? 17u : RFL(n_lp)) + RDL(row, 8);
            const bool b1 = lane == 1 && c1 < nP;
            nP = b1 ? c1 : nP; nB = b1 ? s_bs : nB; n0 = b1 ? s_i0 : n0; n1 = b1 ? s_i1 : n1;
        }
        if (OFF & NO_BRANCH) { v = nP; w = nB; rI0 = n0; rI1 = n1; s += hs >> 24; }
        else if (!rare_s && __ballot(rare_v) == 0) { v = nP; w = nB; rI0 = n0; rI1 = n1; s += hs >> 24; }
        if (!(OFF & NO_PREFETCH)) {
            const uint32_t rs = ((uint32_t)i + 1) & 255;
            n_tg = ((lv32 *)lds)[rs]; n_hs = ((lv32 *)lds)[256 + rs]; n_w = ((lv32 *)lds)[512 + rs * 8 + (lane & 7)];
            n_lp = ((lv16 *)ldsh)[6000 + rs];
            const uint32_t q0 = RDL(rI0, 1), q1 = RDL(rI1, 1);
            const uint32_t rid = lane < 4 ? (__builtin_amdgcn_perm(q1, q0, 0x0C0C0100u) & 255) : 0;
            { typedef uint32_t __attribute__((ext_vector_type(4))) rv4; const rv4 q = *(volatile __attribute__((address_space(3))) rv4 *)&lds4[rid]; n_en.x = q.x; n_en.y = q.y; n_en.z = q.z; n_en.w = q.w; }
            n_fw = ((lv32 *)lds)[7000 + rid];
        }
        if (!(OFF & NO_ROTATE)) { v = DPP(0, v, 0x134); w = DPP(0, w, 0x134); rI0 = DPP(0, rI0, 0x134); rI1 = DPP(0, rI1, 0x134); }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[slot] = t1 - t0;
    g[threadIdx.x] = s + acc + v + w + rI0 + rI1 + n_tg + n_hs + n_w + n_lp + n_fw + n_en.x;
}
int main()
{
    uint32_t *g; unsigned long long *out;
    (void)hipMalloc(&g, 4096); (void)hipMalloc(&out, 64 * 8);
    unsigned long long h[32];
    int n = 0;
#define RUN(F, name) { names[n] = name; hipLaunchKernelGGL(k<F>, dim3(1), dim3(64), 0, 0, g, out, 12345u, n); (void)hipDeviceSynchronize(); n++; }
    const char *names[32];
    for (int rep = 0; rep < 2; rep++) {
        n = 0;
        RUN(0, "full node")
        RUN(NO_PREFETCH, "- prefetch loads (tail)")
        RUN(NO_RECSCALARS, "- record scalars (2 readfirstlane + tests)")
        RUN(NO_DISTSLOT, "- dist_slot chain")
        RUN(NO_LITERAL, "- literal edge")
        RUN(NO_BRANCH, "- rare branch (always commit)")
        RUN(NO_REPMASK, "- rep mask -> length")
        RUN(NO_ROWREAD, "- table row read")
        RUN(NO_RELAX, "- vector relax")
        RUN(NO_ROTATE, "- ring rotate")
        RUN(NO_LABELRFL, "- label readfirstlanes")
        RUN(NO_PREFETCH | NO_RECSCALARS | NO_ROWREAD, "- all LDS reads")
        RUN(NO_PREFETCH | NO_RECSCALARS | NO_ROWREAD | NO_BRANCH | NO_LABELRFL, "- all LDS reads, branch, label rfl")
        RUN(1023, "nothing (loop only)")
    }
    (void)hipMemcpy(h, out, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
    for (int m = 0; m < n; m++) printf("%-50s %8.1f cycles per node\n", names[m], h[m] / (double)IT);
    return 0;
}
