// csc_dec_kernels.hip -- gfx950 kernel for the libcsc DECODE path (reference: src/libcsc/csc_dec.cpp).
//
// One wavefront per stream, like the encoder: decoding is a bit-serial chain (every bit depends on
// the adapted probabilities), so control flow is wave-uniform and the lanes are used for the byte
// work -- match copies into the window, copy-out of the decoded run, delta / E8E9 inverse filters.
//
// Input arrives in RC/BC blocks that the reference pulls from the caller's ISeqInStream strictly on
// demand (csc_memio.cpp:5-81), and only the calling host thread may run that callback.  So the
// kernel is RESUMABLE: k_decode_run decodes one CSCDecoder::Decompress call (csc_dec.cpp:586-682);
// when a block is exhausted and the next one of that kind has not been uploaded yet it rolls the
// current packet back (probability updates are journalled since the last checkpoint), stores its
// state and returns NEED_RC / NEED_BC; the host reads blocks exactly like MemIO::ReadBlock would,
// uploads them and relaunches.  The read pattern seen by the caller is the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csc_device.h"

namespace cscmi {

#define DDEV __device__ __forceinline__
#define DUNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// A taken scalar branch costs a lone wavefront ~35 cycles (tools/ub/ub_issue.hip), a dependent SALU op ~4: rare paths
// are marked so that the hot path is the fall-through one, and short alternatives are written as selects.
#define LIKELY(x) __builtin_expect(!!(x), 1)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)
typedef __attribute__((address_space(1))) uint8_t dgu8;
typedef __attribute__((address_space(1))) uint32_t dgu32;
typedef __attribute__((address_space(1))) DecState gDecState;   // the stream state, known to live in global memory

// LDS image of one stream (138 KiB: one stream per CU).  The order-1 literal table --
// 65 536 twelve-bit probabilities, the table every literal walks 8 levels of -- lives here as u16 for the whole
// launch (HBM image: DecState::p_lit reinterpreted as u16[65536]); so do the small tables and the head of the
// undo journal.  p_delta (DT_DLT runs only) stays in HBM.
constexpr uint32_t kUjLds = 256;
// hand-over block between the Decompress state machine and the fast packet loop (dlz_fast)
struct DecHand {
    uint32_t d_lo, d_hi;                 // the stream's DecState
    uint32_t rd[2], fill[2], slot[2];    // read offsets / payload sizes / ring slots of the current BC and RC blocks
    uint32_t range, code, bc_bits, bc_val, state, ctx, rep[4], wnd_pos, i, limit, ret;
#ifdef CSCMI_TIMERS
    unsigned long long tm[16];
#endif
};
struct DecLds {
    uint32_t P[P_COUNT + 4];
    DecHand hand;
    uint32_t wtab[128];   // the dictionary filter's words, 4 chars packed per entry (0-terminated)
    uint2 uj[kUjLds + 1]; // undo journal, first kUjLds entries of a packet: (table | index, old value); the last entry is a dump
    uint16_t plit[65536];
    uint16_t plit_dump[8];   // where the lanes that hold no node of a literal's path put their "update" (see tree_update)
};
constexpr uint32_t kDecLdsBytes = sizeof(DecLds);
constexpr uint32_t kPDump = P_COUNT + 3;      // the same for the small tables (P[] has four words of slack)
// A file-scope image, not dynamic shared memory: the functions outside the kernel body (dlz_fast, the filters) then address it
// with constants -- the base of a dynamic allocation is looked up in memory (an s_load + wait inside the packet loop).
__shared__ __attribute__((aligned(16))) DecLds g_dec_lds;
#define DEC_LDS (&g_dec_lds)

#ifdef CSCMI_TIMERS
#define DTM_DECL unsigned long long dtm__ = __builtin_readcyclecounter()
#define DTM_ADD(c, k) do { unsigned long long n__ = __builtin_readcyclecounter(); (c).tm[k] += n__ - dtm__; (c).tm[8 + (k)]++; dtm__ = n__; } while (0)
#else
#define DTM_DECL do {} while (0)
#define DTM_ADD(c, k) do {} while (0)
#endif

struct Dc {
#ifdef CSCMI_TIMERS
    unsigned long long tm[16];
#endif
    gDecState *D;
    DecLds *L;
    dgu8 *wnd, *out, *q[2];
    dgu8 *blk[2];       // fast loop only: the current BC / RC block
    uint32_t fast;      // 1 inside dlz_fast (a constant of the function it is set in: the tests on it fold away)
    dgu32 *p_delta, *qsize[2], *undo_addr, *undo_val;
    uint32_t wnd_size, bsize, qslots, lane;
    uint32_t avail[2], taken[2], rd[2], fill[2];
    uint32_t range, code, bc_bits, bc_val, state, ctx, rep[4], wnd_pos;
    uint64_t consumed;
    uint32_t need;      // 0, or DEC_NEED_RC / DEC_NEED_BC once a block ran out with no successor uploaded
    uint32_t err;       // DECODE_ERROR-class failure inside a packet
    uint32_t undo_n;    // bits decoded in this packet (= journal entries when `careful`)
    uint32_t careful;   // journal the probability updates of this packet (it may run out of input and be rolled back)
    uint32_t bv[2], woff[2];   // 64 bytes of the current RC / BC block, one per lane, and the read offset inside them (64 = refetch)
    // resume variables of the Decompress call in flight (mirrors of DecState fields)
    uint32_t phase, type, run_size, i, copied, copied_from, status, out_size, p_delta_ready;
};

// channel counts of the delta types: {1, 2, 3, 4, 8}, csc_typedef.h:36 (as arithmetic at the one place that needs it)

// scalar snapshot taken at every packet / symbol / int boundary
struct Ck {
    uint32_t taken[2], rd[2], fill[2], range, code, bc_bits, bc_val, state, ctx, rep[4], wnd_pos;
    uint64_t consumed;
};

DDEV void ck_take(Dc &c, Ck &k)
{
    for (int i = 0; i < 2; i++) { k.taken[i] = c.taken[i]; k.rd[i] = c.rd[i]; k.fill[i] = c.fill[i]; }
    k.range = c.range; k.code = c.code; k.bc_bits = c.bc_bits; k.bc_val = c.bc_val;
    k.state = c.state; k.ctx = c.ctx; k.wnd_pos = c.wnd_pos; k.consumed = c.consumed;
    for (int i = 0; i < 4; i++) k.rep[i] = c.rep[i];
    c.undo_n = 0;
    c.careful = 1;
}
// undo the probability updates made since the checkpoint and restore the scalars.
// journal address: bits 31:30 = table (0 small tables in LDS, 1 p_lit in LDS, 2 p_delta in HBM)
DDEV void ck_rollback(Dc &c, const Ck &k)
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t n = c.undo_n; n > 0; n--) {
        uint32_t a, v;
        if (n - 1 < kUjLds) { uint2 e = c.L->uj[n - 1]; a = DUNI(e.x); v = DUNI(e.y); }
        else { a = DUNI(c.undo_addr[n - 1 - kUjLds]); v = DUNI(c.undo_val[n - 1 - kUjLds]); }
        const uint32_t sp = a >> 30, idx = a & 0x3FFFFFFFu;
        if (sp == 0) c.L->P[idx] = v;
        else if (sp == 1) c.L->plit[idx] = (uint16_t)v;
        else c.p_delta[idx] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    c.undo_n = 0;
    for (int i = 0; i < 2; i++) { c.taken[i] = k.taken[i]; c.rd[i] = k.rd[i]; c.fill[i] = k.fill[i]; c.woff[i] = 64; }
    c.range = DUNI(k.range); c.code = DUNI(k.code); c.bc_bits = k.bc_bits; c.bc_val = k.bc_val;
    c.state = k.state; c.ctx = k.ctx; c.wnd_pos = k.wnd_pos; c.consumed = k.consumed;
    for (int i = 0; i < 4; i++) c.rep[i] = k.rep[i];
}

// one byte of the RC (kind 1) or BC (kind 0) stream; refills as soon as a block is exhausted, like
// DecodeBit / coder_decode_direct do (csc_dec.cpp:14-21, 70-76)
DDEV uint32_t next_byte(Dc &c, int kind)
{
    if (!c.fast && UNLIKELY(c.need)) return 0;
    // bytes are served from a 64-byte register window over the current block (one HBM fetch per 64 bytes)
    if (UNLIKELY(c.woff[kind] >= 64)) {
        if (c.fast) c.bv[kind] = c.blk[kind][c.rd[kind] + c.lane];
        else {
            uint32_t slot = (c.taken[kind] - 1) % c.qslots;
            c.bv[kind] = c.q[kind][(size_t)slot * c.bsize + c.rd[kind] + c.lane];   // ring has 64 bytes of slack
        }
        c.woff[kind] = 0;
        // The wait for these 64 bytes belongs HERE, once per refill: left to the compiler it lands in front of the v_readlane below, on the path every byte takes
        // (the refill joins it there), as s_waitcnt vmcnt(0) -- and on this target a wavefront's window STORES count in vmcnt until the L2 has acknowledged them:
        // every renormalisation then waited for the last literal's store to come back.  The asm takes the loaded register in and out: what the path below reads
        // is no longer the result of a load.
#ifndef DEC_NO_REFILL_WAIT
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(c.bv[kind]));
#endif
    }
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)c.bv[kind], (int)c.woff[kind]);
    c.woff[kind]++;
    c.rd[kind]++;
    if (!c.fast && UNLIKELY(c.rd[kind] >= c.fill[kind])) {    // (the fast loop stays clear of a block's end)
        if (c.taken[kind] < c.avail[kind]) {
            c.consumed += c.rd[kind];
            uint32_t ns = c.taken[kind] % c.qslots;
            c.fill[kind] = DUNI(c.qsize[kind][ns]);
            c.taken[kind]++;
            c.rd[kind] = 0;
            c.woff[kind] = 64;
        } else {
            c.need = kind ? DEC_NEED_RC : DEC_NEED_BC;
        }
    }
    return b;
}

// ---- bit decoding ---------------------------------------------------------------------------------------------
// DecodeBit (csc_dec.cpp:10-35) is split in two.  The SERIAL part -- range, code, the bit -- runs on scalars and is
// all that sits on the dependency chain (rc_bit, ~12 instructions).  The probability UPDATE of the nodes a symbol
// visited does not feed the chain (a tree visits each node at most once per symbol), so it is done afterwards by the
// lanes that hold those nodes: one vector update + one LDS store per tree instead of one per bit (tree_update).
DDEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
DDEV uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// range / code into scalar registers: at the top of every packet (what flows around the packet loop may count as per-lane data)
DDEV void rc_scalar(Dc &c) { c.range = DUNI(c.range); c.code = DUNI(c.code); }
template <int SITE>
DDEV uint32_t rc_bit_at(Dc &c, uint32_t p)
{
    if (UNLIKELY(c.range < (1u << 24))) { c.range = DUNI(c.range << 8); c.code = DUNI((c.code << 8) + next_byte(c, 1)); }
    // The chain as SCALAR instructions, spelled out: left to itself the compiler keeps range / code / the tree index in vector
    // registers (a quarter-rate v_mul_lo_u32, v_cmp + v_cndmask pairs, a v_readfirstlane before every v_readlane: rounds 1-4
    // measured 177 cycles a binary decision).  Seven SALU operations: bound, the two differences (the second one's borrow is the
    // bit), three selects.
    uint32_t bound, bit, r2, c2;
    asm("; rc_bit site %7\n\t"
        "s_lshr_b32 %0, %4, 12\n\t"
        "s_mul_i32 %0, %0, %6\n\t"
        "s_sub_u32 %2, %4, %0\n\t"
        "s_sub_u32 %3, %5, %0\n\t"          // SCC = the borrow = (code < bound) = the bit
        "s_cselect_b32 %2, %0, %2\n\t"
        "s_cselect_b32 %3, %5, %3\n\t"
        "s_cselect_b32 %1, 1, 0"
        : "=&s"(bound), "=&s"(bit), "=&s"(r2), "=&s"(c2)
        : "s"(c.range), "s"(c.code), "s"(p), "n"(SITE)
        : "scc");
    c.range = r2;
    c.code = c2;
    return bit;
}
#define rc_bit(c, p) rc_bit_at<__LINE__>((c), (p))
// one level of a binary tree walk: the node index moves on to 2 v + bit -- s_addc_u32 v, v, v takes the bit straight from SCC
DDEV uint32_t rc_step(Dc &c, uint32_t p, uint32_t v)
{
    if (UNLIKELY(c.range < (1u << 24))) { c.range = DUNI(c.range << 8); c.code = DUNI((c.code << 8) + next_byte(c, 1)); }
    uint32_t bound, vn, r2, c2;
    asm("s_lshr_b32 %0, %4, 12\n\t"
        "s_mul_i32 %0, %0, %6\n\t"
        "s_sub_u32 %2, %4, %0\n\t"
        "s_sub_u32 %3, %5, %0\n\t"          // SCC = the borrow = (code < bound) = the bit
        "s_cselect_b32 %2, %0, %2\n\t"
        "s_cselect_b32 %3, %5, %3\n\t"
        "s_addc_u32 %1, %7, %7"
        : "=&s"(bound), "=&s"(vn), "=&s"(r2), "=&s"(c2)
        : "s"(c.range), "s"(c.code), "s"(p), "s"(v)
        : "scc");
    c.range = r2;
    c.code = c2;
    return vn;
}
// p += (0xFFF - p) >> 5 or p -= p >> 5 (csc_dec.cpp:24-33; p is a 12-bit number)
// as ONE expression for both bits: p + ((T - p) >> 5) with T = 4095 / 31 and an arithmetic shift -- floor((31 - p) / 32) = -(p >> 5)
DDEV uint32_t p_update(uint32_t p, uint32_t bit)
{
    const int32_t t = bit ? 4095 : 31;
    return p + (uint32_t)((t - (int32_t)p) >> 5);
}
// ---- lane-divergent work lives OUTSIDE the packet loop's control flow --------------------------------------------------
// Rounds 1-5 kept the whole stream context in VECTOR registers (a 215-VGPR loop with thirty v_mov phi copies at its head,
// an exec-mask branch for every scalar test): the compiler's uniformity analysis marks EVERY phi of a block in which the two
// sides of a lane-dependent branch meet as divergent, and after block merging those blocks were the packet loop's own
// latches -- one `if (lane == 0)` or `for (j = lane; ...)` made range, code, positions and counters per-lane data.  So:
// inside the loops that carry the stream's scalars a lane-dependent choice is a SELECT (of an address: lanes with nothing
// to store write to a dump slot), and lane-strided loops are functions of their own (`__noinline__`: their joins are not
// the caller's).  tools/dec_uniformity.sh prints what the analysis says about the kernel.
#define DNOINL __device__ __noinline__
DNOINL void d_copy_bytes(dgu8 *dst, const dgu8 *src, uint32_t n)
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t t = threadIdx.x & 63u; t < n; t += 64) dst[t] = src[t];
}
DNOINL void d_fill_bytes(dgu8 *dst, uint32_t n, uint32_t v)
{
    for (uint32_t t = threadIdx.x & 63u; t < n; t += 64) dst[t] = (uint8_t)v;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
DNOINL void d_fill_words(dgu32 *dst, uint32_t n, uint32_t v)
{
    for (uint32_t t = threadIdx.x & 63u; t < n; t += 64) dst[t] = v;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
// journal address: bits 31:30 = table (0 small tables in LDS, 1 p_lit in LDS, 2 p_delta in HBM)
// entries behind the LDS part of the journal (a packet of more than kUjLds bits: long-length chains only)
DNOINL void d_journal_far(dgu32 *ua, dgu32 *uv, bool on, uint32_t slot, uint32_t addr, uint32_t old)
{
    if (on && slot >= kUjLds && slot < kUjLds + kDecUndoCap) { ua[slot - kUjLds] = addr; uv[slot - kUjLds] = old; }
}
DNOINL void d_store_word(dgu32 *p, bool on, uint32_t v) { if (on) *p = v; }
// one entry per lane that holds a node of the path (`on`), at journal slot `slot` (per lane)
DDEV void journal_lanes(Dc &c, bool on, uint32_t slot, uint32_t addr, uint32_t old, uint32_t slot_max)
{
    c.L->uj[on && slot < kUjLds ? slot : kUjLds] = make_uint2(addr, old);
    if (UNLIKELY(slot_max >= kUjLds)) d_journal_far(c.undo_addr, c.undo_val, on, slot, addr, old);
}
// one entry, the same in every lane
DDEV void journal_put(Dc &c, uint32_t slot, uint32_t addr, uint32_t old)
{
    if (LIKELY(slot < kUjLds)) c.L->uj[slot] = make_uint2(addr, old);
    else if (slot < kUjLds + kDecUndoCap) { c.undo_addr[slot - kUjLds] = addr; c.undo_val[slot - kUjLds] = old; }
}
// After an NB-bit tree has been decoded to `v` (leading 1 included): lane-parallel update of the NB nodes on the
// path.  Each lane holds in `g` the probability of tree node `node` (heap numbering, 1 = root; anything outside
// [1, 2^NB) = not a node of this tree) stored at table index `idx`.  Lanes off the path store to a dump slot.
template <int NB, int SPACE>
DDEV void tree_update(Dc &c, uint32_t g, uint32_t v, uint32_t node, uint32_t idx)
{
    const uint32_t depth = 31u - (uint32_t)__builtin_clz(node | 1u);       // 0 for the root
    const uint32_t sh = (uint32_t)NB - depth;                              // v >> sh is the path's node at that depth
    const bool on = node >= 1u && node < (1u << NB) && (v >> (sh & 31u)) == node;
    const uint32_t bit = (v >> ((sh - 1u) & 31u)) & 1u;
    if (UNLIKELY(c.careful)) journal_lanes(c, on, c.undo_n + depth, idx | ((uint32_t)SPACE << 30), g, c.undo_n + NB);
    const uint32_t np = p_update(g, bit);
    if (SPACE == 0) c.L->P[on ? idx : kPDump] = np;
    else if (SPACE == 1) ((uint16_t *)c.L->plit)[on ? idx : 65536u + (c.lane & 7u)] = (uint16_t)np;   // (plit_dump follows plit)
    else d_store_word(c.p_delta + idx, on, np);
    c.undo_n += NB;
}
// one stand-alone bit under a small-table probability (packet flags, length-slot bits, the long-length escape)
template <int SITE>
DDEV uint32_t dbit_p_at(Dc &c, uint32_t idx, uint32_t p)
{
    const uint32_t bit = rc_bit_at<SITE>(c, p);
    if (UNLIKELY(c.careful)) journal_put(c, c.undo_n, idx, p);
    c.undo_n++;
    c.L->P[idx] = p_update(p, bit);
    return bit;
}
#define dbit_p(c, idx, p) dbit_p_at<__LINE__>((c), (idx), (p))
// A packet flag out of the gather `fl` (lane k holds flag k, read from table index `own`; k's index is idx): the update is made
// by the lane that holds the flag -- three vector operations and a store instead of eight scalar ones, two moves and a store.
template <int SITE>
DDEV uint32_t dflag_at(Dc &c, uint32_t fl, uint32_t own, uint32_t k, uint32_t idx)
{
    const uint32_t p = rl(fl, k);
    const uint32_t bit = rc_bit_at<SITE>(c, p);
    if (UNLIKELY(c.careful)) journal_put(c, c.undo_n, idx, p);
    c.undo_n++;
    c.L->P[c.lane == k ? own : kPDump] = p_update(fl, bit);
    return bit;
}
#define dflag(c, fl, own, k, idx) dflag_at<__LINE__>((c), (fl), (own), (k), (idx))
DDEV uint32_t dflag_index(const Dc &c, uint32_t st3) { return (c.lane < 3 ? P_STATE : P_REPDIST - 3) + st3 + (c.lane < 6 ? c.lane : 5); }
#define dbit(c, idx) dbit_p_at<__LINE__>((c), (idx), DUNI((c).L->P[(idx)]))

DDEV uint32_t ddirect16(Dc &c, uint32_t len)   // coder_decode_direct, csc_dec.cpp:65-88
{
    while (c.bc_bits < len && !c.need) { c.bc_val = (c.bc_val << 8) | next_byte(c, 0); c.bc_bits += 8; }
    if (c.need) return 0;
    uint32_t r = (c.bc_val >> (c.bc_bits - len)) & ((1u << len) - 1);
    c.bc_bits -= len;
    return r;
}
DDEV uint32_t ddirect(Dc &c, uint32_t l) { return l <= 16 ? ddirect16(c, l) : ((ddirect16(c, l - 16) << 16) | ddirect16(c, 16)); }
DDEV uint32_t dget_int(Dc &c)                  // decode_int, csc_dec.cpp:90-97
{
    uint32_t slot = ddirect(c, 5);
    uint32_t num = ddirect(c, slot == 0 ? 1 : slot);
    return slot ? num + (1u << slot) : num;
}

// 8 bits under an order-1 row.  The node of bit k depends on the bits before it, so instead of 8 dependent
// fetches the 15 nodes of the top 4 levels are fetched at once (heap order: lane L = node L), then the 15
// nodes of the 4-level subtree under the node reached.  `top` = the first fetch, issued by the caller.
// The update of the lower subtree's path is left to the caller (dbyte_low_update): the literal loop first asks for what the
// NEXT packet starts with and lets this update ride on those loads' latency.
struct LowTree { uint32_t sub, h, sidx; };
template <int SPACE>
DDEV uint32_t dbyte_bits(Dc &c, uint32_t row, uint32_t top, LowTree &lo)
{
    const uint32_t L = c.lane & 15;
    uint32_t v = 1;
#pragma unroll
    for (int k = 0; k < 4; k++) v = rc_step(c, rl(top, v), v);
    // subtree under node v (16..31): lane L = 2^j + t  ->  node (v << j) + t
    // (tried: asking for it a bit early under both nodes the fourth bit can lead to and choosing afterwards -- the second gather and
    // the two selects cost what the hidden round trip saves: text 5.48 -> 5.38 MB/s, exe 2.37 -> 2.29)
    const uint32_t j = 31u - (uint32_t)__builtin_clz(L | 1u);
    lo.sidx = row + (v << j) + (L - (1u << j));
    lo.sub = SPACE == 1 ? (uint32_t)c.L->plit[lo.sidx] : (uint32_t)c.p_delta[lo.sidx];
    tree_update<4, SPACE>(c, top, v, c.lane < 16 ? L : 0u, row + L);
    uint32_t h = 1;
#pragma unroll
    for (int k = 0; k < 4; k++) h = rc_step(c, rl(lo.sub, h), h);
    lo.h = h;
    return ((v << 4) | (h & 15u)) & 0xFF;
}
template <int SPACE>
DDEV void dbyte_low_update(Dc &c, const LowTree &lo) { tree_update<4, SPACE>(c, lo.sub, lo.h, c.lane < 16 ? (c.lane & 15) : 0u, lo.sidx); }
template <int SPACE>
DDEV uint32_t dbyte_tree_from(Dc &c, uint32_t row, uint32_t top)
{
    LowTree lo;
    const uint32_t b = dbyte_bits<SPACE>(c, row, top, lo);
    dbyte_low_update<SPACE>(c, lo);
    return b;
}
DDEV uint32_t dbyte_tree_l(Dc &c, uint32_t row) { return dbyte_tree_from<1>(c, row, c.L->plit[row + (c.lane & 15)]); }
DDEV uint32_t dbyte_tree_g(Dc &c, uint32_t row) { return dbyte_tree_from<2>(c, row, c.p_delta[row + (c.lane & 15)]); }   // p_delta row (HBM)

// decode_matchlen_1 (csc_dec.cpp:187-220).  P_LEN_SLOT[2], P_LEN_X1[8], P_LEN_X2[8], P_LEN_X3[128] are contiguous: one lane-parallel
// LDS read holds the two slot flags, both 3-bit trees and the top three levels of the 7-bit tree; its lower four levels (rare:
// lengths from 18) are a second gather under the node reached, as in a literal's tree.
DDEV uint32_t dmatchlen_1(Dc &c)
{
    const uint32_t i0 = P_LEN_SLOT + c.lane;
    const uint32_t g0 = c.L->P[i0];
    rc_scalar(c);          // (this bit follows joins of the packet's control flow: see rc_scalar)
    if (LIKELY(dbit_p(c, P_LEN_SLOT, rl(g0, 0)) == 0)) {                      // 3-bit tree, lengths 0..7
        uint32_t i = 1;
#pragma unroll
        for (int k = 0; k < 3; k++) i = rc_step(c, rl(g0, (P_LEN_X1 - P_LEN_SLOT) + i), i);
        tree_update<3, 0>(c, g0, i, i0 - P_LEN_X1, i0);
        return i & 7u;
    }
    if (dbit_p(c, P_LEN_SLOT + 1, rl(g0, 1)) == 0) {                          // 3-bit tree, lengths 8..15
        uint32_t i = 1;
#pragma unroll
        for (int k = 0; k < 3; k++) i = rc_step(c, rl(g0, (P_LEN_X2 - P_LEN_SLOT) + i), i);
        tree_update<3, 0>(c, g0, i, i0 - P_LEN_X2, i0);
        return 8u + (i & 7u);
    }
    static_assert(P_LEN_X3 - P_LEN_SLOT + 7 < 64, "the 7-bit tree's nodes 1-7 lie in the first gather");
    uint32_t i = 1;                                                           // 7-bit tree, lengths 16..143: nodes 1-7 from g0
#pragma unroll
    for (int k = 0; k < 3; k++) i = rc_step(c, rl(g0, (P_LEN_X3 - P_LEN_SLOT) + i), i);
    // the subtree under node i (8..15): lane L = 2^j + t (L = 1..15)  ->  node (i << j) + t
    const uint32_t L = c.lane & 15;
    const uint32_t j = 31u - (uint32_t)__builtin_clz(L | 1u);
    const uint32_t sidx = P_LEN_X3 + (i << j) + (L - (1u << j));
    const uint32_t sub = c.L->P[sidx];
    tree_update<3, 0>(c, g0, i, i0 - P_LEN_X3, i0);
    uint32_t h = 1;
#pragma unroll
    for (int k = 0; k < 4; k++) h = rc_step(c, rl(sub, h), h);
    tree_update<4, 0>(c, sub, h, c.lane < 16 ? L : 0u, sidx);
    return 16u + (((i << 4) | (h & 15u)) & 127u);
}
DDEV uint32_t dmatchlen_2(Dc &c)               // csc_dec.cpp:222-234
{
    uint32_t len = dmatchlen_1(c);
    if (LIKELY(len != 143)) return len;
    for (;;) {
        rc_scalar(c);          // (what flows around this loop counts as per-lane data otherwise)
        if (c.need || c.err || dbit(c, P_LONGLEN)) break;
        len += 143;
        if (c.undo_n > kDecUndoCap) c.err = 1;   // a packet longer than any the encoder can produce
    }
    return len + dmatchlen_1(c);
}
DDEV void dmatch(Dc &c, uint32_t &dist, uint32_t &len)   // decode_match, csc_dec.cpp:236-283
{
    len = dmatchlen_2(c);
    uint32_t pos, sbits;
    if (len == 0) { pos = 0; sbits = 3; }
    else if (len <= 2) { pos = 16 * (len - 1) + 8; sbits = 4; }
    else if (len <= 5) { pos = 32 * (len - 3) + 8 + 32; sbits = 5; }
    else { pos = 32 * 3 + 8 + 32; sbits = 5; }
    const uint32_t si = P_DIST + pos + (c.lane & 31);
    const uint32_t gs = c.L->P[si];                                           // the whole slot tree (<= 31 nodes)
    uint32_t i = 1;
#pragma unroll
    for (int k = 0; k < 3; k++) i = rc_step(c, rl(gs, i), i);
    if (sbits == 3) tree_update<3, 0>(c, gs, i, c.lane < 32 ? c.lane : 0u, si);
    else if (sbits == 4) { i = rc_step(c, rl(gs, i), i); tree_update<4, 0>(c, gs, i, c.lane < 32 ? c.lane : 0u, si); }
    else {
        i = rc_step(c, rl(gs, i), i);
        i = rc_step(c, rl(gs, i), i);
        tree_update<5, 0>(c, gs, i, c.lane < 32 ? c.lane : 0u, si);
    }
    uint32_t slot = i & ((1u << sbits) - 1);
    if (slot <= 2) dist = slot;
    else {
        const uint32_t ebits = slot - 2;
        const uint32_t ei = P_DIST_EXTRA + (ebits - 1) * 16 + (c.lane & 15);
        const uint32_t ge = c.L->P[ei];                                       // the 15 nodes of the low-bits tree
        const uint32_t elen = ebits > 4 ? ddirect(c, ebits - 4) : 0;
        i = 1;
        rc_scalar(c);
#pragma unroll
        for (int k = 0; k < 4; k++) i = rc_step(c, rl(ge, i), i);
        tree_update<4, 0>(c, ge, i, c.lane < 16 ? c.lane : 0u, ei);
        dist = ((1u << ebits) + 1) + (elen << 4) + (__brev(i & 0x0Fu) >> 28);   // dist_table_[slot] + ... + rev16_table_
    }
    c.state = (c.state * 4 + 1) & 0x3F;
}

// window copy of a match: dst[j] = src[j] in increasing j, i.e. a pattern repeat when the regions
// overlap (csc_dec.cpp:513-518) -- lane-parallel with the modulo made explicit.  Earlier window stores of THIS wavefront
// are ordered before these loads by the hardware (one wavefront's vector memory operations execute in order through one
// L1); nothing waits behind the stores.  Returns the last byte copied (the next literal context) -- in every lane, but as
// the result of a call: the caller passes it through v_readfirstlane.  A function of its own: see DNOINL above.
DNOINL uint32_t d_copy_match(dgu8 *w, uint32_t from, uint32_t to, uint32_t dist, uint32_t len)
{
    const uint32_t lane = threadIdx.x & 63u;
    const bool overlap = from < to && from + len > to;   // then dist = to - from < len
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    uint32_t last = 0;
    if (!overlap) {
        for (uint32_t j = lane; j < len; j += 64) { last = w[from + j]; w[to + j] = (uint8_t)last; }
    } else {
        // every source byte is one of the `dist` bytes in front of `to`: fetch those once, then replicate
        for (uint32_t j = lane; j < len; j += 64) { last = w[from + j % dist]; w[to + j] = (uint8_t)last; }
    }
    return rl(last, (len - 1) & 63u);
}
DDEV uint32_t dcopy_match(Dc &c, uint32_t from, uint32_t dist, uint32_t len) { return DUNI(d_copy_match(c.wnd, from, c.wnd_pos, dist, len)); }

// One packet of CSCDecoder::lz_decode (csc_dec.cpp:476-571): the bits, the model updates, the new rep distances / state; the
// window write it asks for comes back in `w` (the caller does it once the packet is known to stand).
struct Wr { uint32_t kind, from, dist, len, byte; };     // kind 0 nothing (the terminator), 1 a literal, 2 a copy
// what a packet can start with -- the three packet-kind flags, the two rep-index flags that can follow (lanes 0-2 / 3-5) and the
// top of the literal tree under the current context -- in one LDS round trip; then the first flag.  True: a literal follows.
DDEV bool dlz_packet_head(Dc &c, uint32_t &st3, uint32_t &fl, uint32_t &lrow, uint32_t &ltop)
{
    st3 = c.state * 3;
    fl = c.L->P[dflag_index(c, st3)];
    lrow = c.ctx * 256;
    ltop = c.L->plit[lrow + (c.lane & 15)];
    return dflag(c, fl, dflag_index(c, st3), 0, P_STATE + st3) == 0;
}
DDEV uint32_t dlz_literal(Dc &c, uint32_t lrow, uint32_t ltop)
{
    const uint32_t b = dbyte_tree_from<1>(c, lrow, ltop);
    c.ctx = b;
    c.state = (c.state * 4) & 0x3F;
    return b;
}
// the packet kinds behind a set first flag
DDEV void dlz_packet_rest(Dc &c, uint32_t st3, uint32_t fl, uint32_t i, uint32_t limit, uint32_t &ni, bool &end, Wr &w)
{
    ni = i; end = false;
    w.kind = 0; w.from = 0; w.dist = 0; w.len = 0; w.byte = 0;
    const uint32_t own = dflag_index(c, st3);
    if (dflag(c, fl, own, 1, P_STATE + st3 + 1) == 1) {
        uint32_t dist, len;
        dmatch(c, dist, len);
        if (len == 0 && dist == 64) end = true;
        else {
            dist++; len += 2;
            c.rep[3] = c.rep[2]; c.rep[2] = c.rep[1]; c.rep[1] = c.rep[0]; c.rep[0] = dist;
            uint32_t from = c.wnd_pos >= dist ? c.wnd_pos - dist : c.wnd_pos + c.wnd_size - dist;
            if (from >= c.wnd_size || from + len > c.wnd_size || len + i > limit || c.wnd_pos + len > c.wnd_size) c.err = 1;
            w.kind = 2; w.from = from; w.dist = dist; w.len = len; ni = i + len;
        }
    } else if (dflag(c, fl, own, 2, P_STATE + st3 + 2) == 0) {
        c.state = (c.state * 4 + 2) & 0x3F;
        uint32_t from = c.wnd_pos > c.rep[0] ? c.wnd_pos - c.rep[0] : c.wnd_pos + c.wnd_size - c.rep[0];
        if (from > c.wnd_size) c.err = 1;   // the reference reads out of bounds here on corrupt input
        w.kind = 2; w.from = from; w.dist = c.rep[0]; w.len = 1; ni = i + 1;
    } else {
        uint32_t kk = 2u + dflag(c, fl, own, 3, P_REPDIST + st3);
        kk = kk + kk + dflag(c, fl, own, 3 + kk - 1, P_REPDIST + st3 + kk - 1);
        uint32_t idx = kk & 3, len = dmatchlen_2(c) + 2;
        c.state = (c.state * 4 + 3) & 0x3F;
        if (len + i > limit) c.err = 1;
        // move-to-front as selects: a dynamic index would push the whole stream context into scratch memory
        const uint32_t r0 = c.rep[0], r1 = c.rep[1], r2 = c.rep[2], r3 = c.rep[3];
        uint32_t dist = idx == 0 ? r0 : idx == 1 ? r1 : idx == 2 ? r2 : r3;
        c.rep[3] = idx >= 3 ? r2 : r3;
        c.rep[2] = idx >= 2 ? r1 : r2;
        c.rep[1] = idx >= 1 ? r0 : r1;
        c.rep[0] = dist;
        uint32_t from = c.wnd_pos >= dist ? c.wnd_pos - dist : c.wnd_pos + c.wnd_size - dist;
        if (from >= c.wnd_size || from + len > c.wnd_size || len + i > limit || c.wnd_pos + len > c.wnd_size) c.err = 1;
        w.kind = 2; w.from = from; w.dist = dist; w.len = len; ni = i + len;
    }
}
DDEV void dlz_packet(Dc &c, uint32_t i, uint32_t limit, uint32_t &ni, bool &end, Wr &w)
{
    uint32_t st3, fl, lrow, ltop;
    if (dlz_packet_head(c, st3, fl, lrow, ltop)) {
        end = false; ni = i + 1;
        w.kind = 1; w.from = 0; w.dist = 0; w.len = 0; w.byte = dlz_literal(c, lrow, ltop);
    } else dlz_packet_rest(c, st3, fl, i, limit, ni, end, w);
}
// the window write of a packet that stands
DDEV void dlz_write(Dc &c, const Wr &w)
{
    if (w.kind == 1) { c.wnd[c.wnd_pos] = (uint8_t)w.byte; c.wnd_pos++; }   // (every lane the same byte to the same place)
    else if (w.kind == 2) {
        c.ctx = dcopy_match(c, w.from, w.dist, w.len);
        c.wnd_pos += w.len;
    }
}
// A packet is journalled (and the scalars checkpointed) only if the input could run out inside it: fewer than
// kDecUndoCap/8 + 64 coded bytes left in the current RC block, or fewer than 16 in the BC block.  Everywhere
// else -- 93 % of a 64 KiB block -- a packet cannot exhaust its block before the kDecUndoCap-bit limit stops it.
DDEV bool dlz_careful_zone(const Dc &c) { return c.rd[1] + kDecUndoCap / 8 + 64 > c.fill[1] || c.rd[0] + 16 > c.fill[0]; }

// The packets outside the careful zones -- nothing to journal, no block can end -- run in a loop, and a FUNCTION, of their
// own: what it carries from packet to packet is a dozen scalars, not the Decompress state machine's context with its
// checkpoint, and its registers are its own (no argument: it finds the stream through the hand-over block in LDS and
// puts it back there).  hand.ret: 0 = went as far as it may (a careful zone, the end of the run, the end of the window:
// the caller looks), 1 = the terminator packet was decoded, 2 = a packet that cannot be (DECODE_ERROR).
DNOINL void dlz_fast()
{
    DecLds *const L = DEC_LDS;
    DecHand &H = L->hand;
    Dc c;
    gDecState *const D = (gDecState *)(((uint64_t)DUNI(H.d_hi) << 32) | DUNI(H.d_lo));
    auto U64 = [](uint64_t v) { return ((uint64_t)DUNI((uint32_t)(v >> 32)) << 32) | DUNI((uint32_t)v); };
    c.D = D; c.L = L; c.lane = threadIdx.x & 63u;
    c.wnd = (dgu8 *)U64((uint64_t)D->wnd);
    c.wnd_size = DUNI(D->wnd_size);
    const uint32_t bsize = DUNI(D->bsize);
    c.blk[0] = (dgu8 *)U64((uint64_t)D->q[0]) + (size_t)DUNI(H.slot[0]) * bsize;
    c.blk[1] = (dgu8 *)U64((uint64_t)D->q[1]) + (size_t)DUNI(H.slot[1]) * bsize;
    c.fast = 1; c.careful = 0; c.need = 0; c.err = 0; c.undo_n = 0;
    c.rd[0] = DUNI(H.rd[0]); c.rd[1] = DUNI(H.rd[1]); c.fill[0] = DUNI(H.fill[0]); c.fill[1] = DUNI(H.fill[1]);
    c.range = DUNI(H.range); c.code = DUNI(H.code); c.bc_bits = DUNI(H.bc_bits); c.bc_val = DUNI(H.bc_val);
    c.state = DUNI(H.state); c.ctx = DUNI(H.ctx); c.wnd_pos = DUNI(H.wnd_pos);
    for (int r = 0; r < 4; r++) c.rep[r] = DUNI(H.rep[r]);
    c.bv[0] = c.bv[1] = 0; c.woff[0] = c.woff[1] = 64;
    uint32_t i = DUNI(H.i), ret = 0;
    const uint32_t limit = DUNI(H.limit);
#ifdef CSCMI_TIMERS
    for (int t = 0; t < 16; t++) c.tm[t] = 0;
#endif
    // A copy of up to 64 bytes is TWO instructions a packet apart: its load is issued when the packet is decoded, its store -- and
    // the wait in front of it -- when the next packet needs the window or the context (a literal: the byte it is coded under; a
    // copy: before its own load, a wavefront's memory operations are carried out in order).  The bytes wait in `pv`, one a lane;
    // lanes >= p_n have nothing to store (their buffer offset is out of range: the hardware drops them), lane p_last holds the
    // byte that follows as context.  With no copy pending pv is the context byte in every lane.
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)c.wnd, 0, (int)(c.wnd_size + 256u), 0x00020000);
    uint32_t pv = c.ctx, p_n = 0, p_to = 0, p_last = 0;
    // Development A/B (tools/ab_variant_dec.sh, profiles/r06_dec_two_wave.md): DEC_AB_COPY2 (= DEC_AB_LIT2 + DEC_AB_MATCH2: the literals' stores / the copies' loads and stores) issues every window store / copy load of this loop a second
    // time (same bytes, same places: the output stands) -- what the window work costs THIS wavefront in issue slots, i.e. the most a second wavefront that owned
    // the window could take off it; DEC_AB_TOKEN adds what such a split would add instead: a token and a count into LDS per packet.
#ifdef DEC_AB_COPY2
#define DEC_AB_LIT2
#define DEC_AB_MATCH2
#endif
#ifdef DEC_AB_TOKEN
    __shared__ uint32_t ab_tok[128][2];
    __shared__ uint32_t ab_head;
    uint32_t ab_n = 0;
#define DEC_AB_PUSH(x, y) do { *(volatile __attribute__((address_space(3))) uint32_t *)&ab_tok[ab_n & 127u][0] = (x); *(volatile __attribute__((address_space(3))) uint32_t *)&ab_tok[ab_n & 127u][1] = (y); ab_n++; *(volatile __attribute__((address_space(3))) uint32_t *)&ab_head = ab_n; } while (0)
#else
#define DEC_AB_PUSH(x, y) do {} while (0)
#endif
    auto flush = [&]() {
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pv, wr, (int)(c.lane < p_n ? p_to + c.lane : 0xFFFFFFFFu), 0, 0);
#if defined(DEC_AB_MATCH2) || defined(DEC_AB_CLOB)
        asm volatile("" ::: "memory");
#endif
#ifdef DEC_AB_MATCH2
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pv, wr, (int)(c.lane < p_n ? p_to + c.lane : 0xFFFFFFFFu), 0, 0);
#endif
        p_n = 0;
    };
    for (;;) {
        // the same tests as the caller's
        if (UNLIKELY(i > limit || c.wnd_pos >= c.wnd_size || c.rd[1] + kDecUndoCap / 8 + 64 > c.fill[1] || c.rd[0] + 16 > c.fill[0])) break;
        DTM_DECL;
        // the packet's flags hang on the state only: the first of them is decoded while a pending copy's load is still under way
        uint32_t st3 = c.state * 3;
        uint32_t fl = c.L->P[dflag_index(c, st3)];
        // A run of literals is a loop of its own: nothing of a literal can fail, it moves only the coder's words, the context, the
        // state and the two positions -- the rep distances and the bit coder's words are not carried round this loop.
        if (LIKELY(dflag(c, fl, dflag_index(c, st3), 0, P_STATE + st3) == 0)) {
          c.ctx = rl(pv, p_last);
          flush();
          uint32_t lrow = c.ctx * 256;
          uint32_t ltop = c.L->plit[lrow + (c.lane & 15)];
          // the run ends with the run of the block / the window, or in front of a careful zone: one count, one comparison
          const uint32_t n0 = umin(limit + 1 - i, c.wnd_size - c.wnd_pos), rd_stop = c.fill[1] - (kDecUndoCap / 8 + 64);
          uint32_t n_left = n0;
          bool stop;
          do {
            LowTree lo;
            const uint32_t b = dbyte_bits<1>(c, lrow, ltop, lo);
            c.ctx = b;
            c.state = (c.state * 4) & 0x3F;
            DTM_ADD(c, 0);
            // what the next packet starts with is asked for NOW -- it hangs on the byte and the state only; the rest of this
            // literal (its lower subtree's update: nodes 16-255, the next packet's gather reads nodes 1-15; the window store, the
            // loop's tests) rides on the latency of these loads
            // (volatile: the compiler would sink the flags' load behind the loop's tests, to where they are used)
            st3 = c.state * 3;
            fl = *(volatile __attribute__((address_space(3))) uint32_t *)&c.L->P[dflag_index(c, st3)];
            lrow = b * 256;
            ltop = *(volatile __attribute__((address_space(3))) uint16_t *)&c.L->plit[lrow + (c.lane & 15)];
            dbyte_low_update<1>(c, lo);
            c.wnd[c.wnd_pos] = (uint8_t)b;          // (every lane the same byte to the same place)
#if defined(DEC_AB_LIT2) || defined(DEC_AB_CLOB)
            asm volatile("" ::: "memory");
#endif
#ifdef DEC_AB_LIT2
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)b, wr, (int)c.wnd_pos, 0, 0);          // (not a volatile store: that one would wait for its own acknowledgement)
#endif
            DEC_AB_PUSH(b, c.wnd_pos);
            c.wnd_pos++;
            n_left--;
            DTM_ADD(c, 2);
            stop = n_left == 0 || c.rd[1] > rd_stop;
          } while (LIKELY(!stop) && LIKELY(dflag(c, fl, dflag_index(c, st3), 0, P_STATE + st3) == 0));
          i += n0 - n_left;
          pv = c.ctx; p_last = 0;
          if (UNLIKELY(stop)) break;     // (nothing of the next packet has been decoded)
        }
        c.undo_n = 1;
        uint32_t ni; bool end; Wr w;
        dlz_packet_rest(c, st3, fl, i, limit, ni, end, w);
        DTM_ADD(c, 1);
        if (UNLIKELY(((c.undo_n > kDecUndoCap ? 1u : 0u) | c.err | (end ? 1u : 0u)) != 0)) { ret = (c.err || c.undo_n > kDecUndoCap) ? 2u : 1u; break; }
        // the packet's copy (every kind that comes here has one)
        flush();
        const bool overlap = w.from < c.wnd_pos && w.from + w.len > c.wnd_pos;      // then dist = wnd_pos - from < len
#ifdef DEC_DBG_NODEFER
        if (false) {                     // (development: every copy at once, none deferred)
#else
        if (LIKELY(w.len <= 64 && !overlap)) {
#endif
            pv = __builtin_amdgcn_raw_buffer_load_b8(wr, (int)(w.from + c.lane), 0, 0);
#if defined(DEC_AB_MATCH2) || defined(DEC_AB_CLOB)
            asm volatile("" ::: "memory");
#endif
#ifdef DEC_AB_MATCH2
            pv |= __builtin_amdgcn_raw_buffer_load_b8(wr, (int)(w.from + c.lane), 0, 0);
#endif
            DEC_AB_PUSH(w.from, w.len | 0x80000000u);
            p_n = w.len; p_to = c.wnd_pos; p_last = w.len - 1;
        } else {
            pv = dcopy_match(c, w.from, w.dist, w.len);
            p_last = 0;
        }
        c.wnd_pos += w.len;
        i = ni;
        DTM_ADD(c, 2);
    }
    c.ctx = rl(pv, p_last);
    flush();
    H.rd[0] = c.rd[0]; H.rd[1] = c.rd[1];
    H.range = c.range; H.code = c.code; H.bc_bits = c.bc_bits; H.bc_val = c.bc_val;
    H.state = c.state; H.ctx = c.ctx; H.wnd_pos = c.wnd_pos;
    for (int r = 0; r < 4; r++) H.rep[r] = c.rep[r];
    H.i = i; H.ret = ret;
#ifdef CSCMI_TIMERS
    for (int t = 0; t < 3; t++) { H.tm[t] += c.tm[t]; H.tm[8 + t] += c.tm[8 + t]; }
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// CSCDecoder::lz_decode, csc_dec.cpp:476-571: one packet per loop turn; checkpointed per packet inside the careful zones,
// handed to dlz_fast outside them
DDEV void dlz_decode(Dc &c, uint32_t limit)
{
    uint32_t i = c.i, copied = c.copied, copied_from = c.copied_from;
    Ck k;
    for (; i <= limit;) {
#ifdef DEC_DBG_CAREFUL
        const bool zone = true;          // (development: every packet the checkpointed way, dlz_fast never runs)
#else
        const bool zone = dlz_careful_zone(c);
#endif
        bool end = false;
        if (LIKELY(!zone)) {
            DecHand &H = c.L->hand;
            H.rd[0] = c.rd[0]; H.rd[1] = c.rd[1]; H.fill[0] = c.fill[0]; H.fill[1] = c.fill[1];
            H.slot[0] = (c.taken[0] - 1) % c.qslots; H.slot[1] = (c.taken[1] - 1) % c.qslots;
            H.range = c.range; H.code = c.code; H.bc_bits = c.bc_bits; H.bc_val = c.bc_val;
            H.state = c.state; H.ctx = c.ctx; H.wnd_pos = c.wnd_pos;
            for (int r = 0; r < 4; r++) H.rep[r] = c.rep[r];
            H.i = i; H.limit = limit;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            dlz_fast();
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            c.rd[0] = DUNI(H.rd[0]); c.rd[1] = DUNI(H.rd[1]);
            c.range = DUNI(H.range); c.code = DUNI(H.code); c.bc_bits = DUNI(H.bc_bits); c.bc_val = DUNI(H.bc_val);
            c.state = DUNI(H.state); c.ctx = DUNI(H.ctx); c.wnd_pos = DUNI(H.wnd_pos);
            for (int r = 0; r < 4; r++) c.rep[r] = DUNI(H.rep[r]);
            c.woff[0] = c.woff[1] = 64;
            i = DUNI(H.i);
            const uint32_t ret = DUNI(H.ret);
            if (UNLIKELY(ret == 2)) { c.status = DEC_ERR_DECODE; return; }
            end = ret == 1;
        } else {
            DTM_DECL;
            c.careful = 1;
            c.undo_n = 0;
            rc_scalar(c);
            ck_take(c, k);
            uint32_t ni; Wr w;
            dlz_packet(c, i, limit, ni, end, w);
            if (w.kind == 2) DTM_ADD(c, 1); else DTM_ADD(c, 0);
            // the rare outcomes of a packet behind ONE test
            if (UNLIKELY(((c.undo_n > kDecUndoCap ? 1u : 0u) | c.err | c.need) != 0)) {
                if (c.undo_n > kDecUndoCap) c.err = 1;
                if (c.err) { c.status = DEC_ERR_DECODE; return; }
                // ran out of input inside this packet: take it back, resume here later
                ck_rollback(c, k);
                c.i = i; c.copied = copied; c.copied_from = copied_from;
                return;
            }
            if (!end) { dlz_write(c, w); i = ni; }
            DTM_ADD(c, 2);
        }
        if (UNLIKELY(end)) break;                 // the terminator
        if (UNLIKELY(c.wnd_pos >= c.wnd_size)) {
            if (c.wnd_pos > c.wnd_size) { c.status = DEC_ERR_DECODE; return; }
            d_copy_bytes(c.out + copied, c.wnd + copied_from, i - copied);
            c.wnd_pos = 0;
            copied_from = 0;
            copied = i;
        }
    }
    {
        DTM_DECL;
        d_copy_bytes(c.out + copied, c.wnd + copied_from, i - copied);
        DTM_ADD(c, 3);
    }
    c.out_size = i;
    c.i = 0; c.copied = 0; c.copied_from = 0;
    c.phase = DEC_PH_POST;
}

DDEV void dcopy2dict(Dc &c, uint32_t size)   // lz_copy2dict, csc_dec.cpp:573-584
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t i = 0; i < size;) {
        uint32_t cur = c.wnd_size - c.wnd_pos < size - i ? c.wnd_size - c.wnd_pos : size - i;
        if (cur > kMinBlock) cur = kMinBlock;
        d_copy_bytes(c.wnd + c.wnd_pos, c.out + i, cur);
        c.wnd_pos += cur;
        if (c.wnd_pos >= c.wnd_size) c.wnd_pos = 0;
        i += cur;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// ---- inverse filters on the decoded run in `out` ----
// Inverse_E89 (csc_filters.cpp:560-575,600-610), unrolled like the forward filter: opcode candidates by
// ballot, the skip chain over the sparse hits is serial, operands are rewritten in place.
DNOINL void dinverse_e89(dgu8 *b, uint32_t size)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t next_ok = 0;
    for (uint32_t base = 0; base + 5 < size; base += 64) {
        uint32_t j0 = base + lane;
        bool cand = false;
        if (j0 + 5 < size) cand = (b[j0] & 0xFEu) == 0xE8u;
        uint64_t m = __ballot(cand);
        while (m) {
            uint32_t bit = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            uint32_t j = base + bit;
            if (j < next_ok) continue;
            next_ok = j + 4;
            uint32_t x0 = DUNI((uint32_t)b[j + 1]) | (DUNI((uint32_t)b[j + 2]) << 8) | (DUNI((uint32_t)b[j + 3]) << 16) | (DUNI((uint32_t)b[j + 4]) << 24);
            uint32_t x = x0 - 0xFF000000u;
            if (x < 0x02000000u) {
                x = ((x >> 24) << 7) | (((x >> 16) & 0xFF) << 8) | (((x >> 8) & 0xFF) << 16) | (x << 24);   // E89yswap
                x >>= 7;
                x = ((x - (j + 5)) & 0x01FFFFFFu) + 0xFF000000u;
                b[j + 1] = (uint8_t)x; b[j + 2] = (uint8_t)(x >> 8); b[j + 3] = (uint8_t)(x >> 16); b[j + 4] = (uint8_t)(x >> 24);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// Inverse_Delta (csc_filters.cpp:371-399): a running byte sum over the de-interleaved order
DNOINL void dinverse_delta(dgu8 *src, dgu8 *copy, uint32_t size, uint32_t chn)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (size < 512) return;
    for (uint32_t t = lane; t < size; t += 64) copy[t] = src[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    uint32_t carry = 0;
    for (uint32_t base = 0; base < size; base += 64) {
        uint32_t n = base + lane;
        uint32_t v = n < size ? copy[n] : 0;
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o); if ((int)lane >= o) v += t; }   // inclusive scan
        v = (v + carry) & 0xFF;
        if (n < size) {
            // n-th element of the de-interleaved order is position j = ch + idx * chn
            uint32_t ch = 0, acc = 0, idx = n;
            for (uint32_t k = 0; k < chn; k++) {
                uint32_t cnt = (size - k + chn - 1) / chn;
                if (n >= acc && n < acc + cnt) { ch = k; idx = n - acc; }
                acc += cnt;
            }
            src[ch + idx * chn] = (uint8_t)v;
        }
        carry = DUNI((uint32_t)__shfl(v, 63));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// Inverse_Dict (csc_filters.cpp:337-369): expand word symbols; 254 escapes a byte >= 0x82.
// 64 source bytes per step, one per lane.  Which bytes are escaped follows from the parity of the run of 254s in
// front of them (a 254 is an escape marker iff it is not itself escaped and the next byte is >= 0x82, so inside a
// run markers and escaped bytes alternate); every lane then knows how many bytes it emits (0 marker, 1, or a
// 2-4 letter word), an exclusive wave scan gives its output offset, and it stores its bytes.  The last 66 source
// positions -- where the reference's `i + 1 < size` guard matters -- are walked serially like the reference does.
DNOINL void dinverse_dict(dgu8 *src, dgu8 *dst, uint32_t size)
{
    const uint32_t lane = threadIdx.x & 63u;
    DecLds *const lds = DEC_LDS;
    uint32_t i = 0, o = 0;
    uint32_t carry = 0;                                  // parity of the run of 254s that ends just before position i
    while (o < size && i + 66 < size) {
        const uint32_t b = src[i + lane];              // `out` has slack behind `size`
        const uint32_t nb_lane63 = DUNI((uint32_t)src[i + 64]);
        const uint64_t m254 = __ballot(b == 254);
        // r = number of consecutive 254s immediately below this lane (run reaching the chunk start continues the carry)
        const uint64_t below = lane ? (m254 << (64 - lane)) : 0ull;            // bits [0, lane) moved to the top
        uint32_t r = lane ? (uint32_t)__builtin_clzll(~below | ((1ull << (64 - lane)) - 1ull)) : 0u;
        if (r > lane) r = lane;
        const uint32_t par = (r + (r == lane ? carry : 0u)) & 1u;
        const bool ge82 = b >= 0x82;
        const bool esc = par && ge82;                                                 // swallowed by the marker before it
        uint32_t nb = (uint32_t)__shfl_down((int)b, 1);
        if (lane == 63) nb = nb_lane63;
        const bool marker = b == 254 && !esc && nb >= 0x82;
        const bool word = !esc && b >= 0x82 && b < 0x82 + 122;
        const uint32_t w = word ? lds->wtab[b - 0x82] : b;
        const uint32_t wl = word ? ((w >> 16) & 0xFF ? ((w >> 24) ? 4u : 3u) : 2u) : 1u;   // words have 2..4 letters
        const uint32_t n = marker ? 0u : wl;
        uint32_t x = n;                                                               // inclusive scan over the wave
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)x, off); if ((int)lane >= off) x += t; }
        const uint32_t at = o + x - n;
        for (uint32_t t = 0; t < n; t++) if (at + t < size) dst[at + t] = (uint8_t)(w >> (8 * t));
        // carry for the next chunk: parity of the run of 254s at the top of this one
        const uint32_t lead = m254 == ~0ull ? 64u : (uint32_t)__builtin_clzll(~m254);
        carry = (lead == 64 ? carry + 64u : lead) & 1u;
        // a trailing marker in lane 63 swallows the first byte of the next chunk: that is what carry expresses
        o += DUNI((uint32_t)__shfl((int)x, 63));
        i += 64;
    }
    // serial tail (and the whole run if it is shorter than 66 bytes); `carry` odd = position i is an escaped byte
    {
        bool escaped = false;
        if (carry & 1u) {
            // the run of 254s before i has odd length: its last 254 is a marker iff src[i] >= 0x82
            escaped = DUNI((uint32_t)src[i]) >= 0x82;
        }
        while (o < size) {
            const uint32_t base = i;
            uint32_t v = src[base + lane];
            uint32_t vn = DUNI((uint32_t)src[base + 64]);
            uint32_t j = 0;
            while (j < 64 && o < size) {
                uint32_t b = rl(v, j);
                if (escaped) { dst[o++] = (uint8_t)b; escaped = false; }
                else if (b >= 0x82 && b < 0x82 + 122) {
                    uint32_t w = DUNI(lds->wtab[b - 0x82]);
                    for (uint32_t t = 0; t < 4 && o < size; t++) {
                        uint32_t ch = (w >> (8 * t)) & 0xFF;
                        if (!ch) break;
                        dst[o++] = (uint8_t)ch;
                    }
                } else {
                    uint32_t nb = j + 1 < 64 ? rl(v, j + 1) : vn;
                    if (b == 254 && base + j + 1 < size && nb >= 0x82) { j++; dst[o++] = (uint8_t)nb; }
                    else dst[o++] = (uint8_t)b;
                }
                j++;
            }
            i = base + j;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t t = lane; t < size; t += 64) src[t] = dst[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// re-prime the arithmetic decoder from the NEXT RC and BC blocks (csc_dec.cpp:336-345, 657-680)
DDEV bool dprime(Dc &c)
{
    if (c.taken[1] >= c.avail[1]) { c.need = DEC_NEED_RC; return false; }
    if (c.taken[0] >= c.avail[0]) { c.need = DEC_NEED_BC; return false; }
#pragma unroll
    for (int kind = 1; kind >= 0; kind--) {
        uint32_t ns = c.taken[kind] % c.qslots;
        c.fill[kind] = DUNI(c.qsize[kind][ns]);
        c.taken[kind]++;
        c.rd[kind] = 0;
    }
    c.range = DUNI(0xFFFFFFFFu); c.bc_bits = c.bc_val = 0;
    const dgu8 *p = c.q[1] + (size_t)((c.taken[1] - 1) % c.qslots) * c.bsize;
    c.code = DUNI((DUNI((uint32_t)p[1]) << 24) | (DUNI((uint32_t)p[2]) << 16) | (DUNI((uint32_t)p[3]) << 8) | DUNI((uint32_t)p[4]));
    c.rd[1] = 5;
    return true;
}

DDEV void decode_stream(gDecState *D, DecLds &lds)
{
    Dc c;
    c.D = D; c.L = &lds; c.lane = threadIdx.x;
    // Every field below is the same in all lanes, and the compiler is TOLD so (v_readfirstlane): a value that merely arrives by a
    // vector load stays in a vector register, and the bit chain -- bound = (range >> 12) * p, the compare, the two selects, the
    // renormalisation test -- then runs on the vector unit with a quarter-rate v_mul_lo_u32 and an exec-mask branch per bit
    // (rounds 1-4: 177 cycles a binary decision).  In scalar registers it is s_mul_i32 / s_cmp / s_cselect and a scalar branch.
    auto U64 = [](uint64_t v) { return ((uint64_t)DUNI((uint32_t)(v >> 32)) << 32) | DUNI((uint32_t)v); };
    c.wnd = (dgu8 *)U64((uint64_t)D->wnd); c.out = (dgu8 *)U64((uint64_t)D->out); c.q[0] = (dgu8 *)U64((uint64_t)D->q[0]); c.q[1] = (dgu8 *)U64((uint64_t)D->q[1]);
    c.p_delta = (dgu32 *)U64((uint64_t)D->p_delta);
    c.qsize[0] = (dgu32 *)U64((uint64_t)D->qsize[0]); c.qsize[1] = (dgu32 *)U64((uint64_t)D->qsize[1]);
    c.undo_addr = (dgu32 *)U64((uint64_t)D->undo_addr); c.undo_val = (dgu32 *)U64((uint64_t)D->undo_val);
    c.wnd_size = DUNI(D->wnd_size); c.bsize = DUNI(D->bsize); c.qslots = DUNI(D->qslots);
    for (int i = 0; i < 2; i++) { c.avail[i] = DUNI(D->avail[i]); c.taken[i] = DUNI(D->taken[i]); c.rd[i] = DUNI(D->rd[i]); c.fill[i] = DUNI(D->fill[i]); }
    c.range = DUNI(D->range); c.code = DUNI(D->code); c.bc_bits = DUNI(D->bc_bits); c.bc_val = DUNI(D->bc_val);
    c.state = DUNI(D->state); c.ctx = DUNI(D->ctx); c.wnd_pos = DUNI(D->wnd_pos); c.consumed = U64(D->consumed);
    for (int i = 0; i < 4; i++) c.rep[i] = DUNI(D->rep[i]);
#ifdef CSCMI_TIMERS
    for (int i = 0; i < 16; i++) c.tm[i] = 0;
    unsigned long long tk0 = __builtin_readcyclecounter();
#endif
    c.need = 0; c.err = 0; c.undo_n = 0; c.careful = 1; c.fast = 0; c.blk[0] = c.blk[1] = nullptr;
    c.bv[0] = c.bv[1] = 0; c.woff[0] = c.woff[1] = 64;
    c.phase = DUNI(D->phase); c.type = DUNI(D->type); c.run_size = DUNI(D->run_size); c.i = DUNI(D->i); c.copied = DUNI(D->copied); c.copied_from = DUNI(D->copied_from);
    c.out_size = DUNI(D->out_size); c.p_delta_ready = DUNI(D->p_delta_ready); c.status = DEC_RUNNING;
    lds.hand.d_lo = (uint32_t)(uint64_t)D; lds.hand.d_hi = (uint32_t)((uint64_t)D >> 32);
#ifdef CSCMI_TIMERS
    for (int i = 0; i < 16; i++) lds.hand.tm[i] = 0;
#endif
    for (uint32_t i = c.lane; i < P_COUNT; i += 64) lds.P[i] = D->probs[i];
    for (uint32_t i = c.lane; i < 122; i += 64) {
        const dgu8 *w = (const dgu8 *)D->words + i * 8;
        lds.wtab[i] = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) | ((uint32_t)w[3] << 24);
    }
    {   // p_lit image (u16[65536], 128 KiB) HBM -> LDS, 16 bytes per lane and step
        typedef uint32_t __attribute__((ext_vector_type(4))) v4u;
        const __attribute__((address_space(1))) v4u *src = (const __attribute__((address_space(1))) v4u *)D->p_lit;
        v4u *dst = (v4u *)lds.plit;
        for (uint32_t i = c.lane; i < 65536 * 2 / 16; i += 64) dst[i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t max = DUNI(D->raw_blocksize);
    Ck k;
    bool done = false;
    while (!done && !c.need && c.status == DEC_RUNNING) {
        switch (c.phase) {
        case DEC_PH_PRIME0:
            if (dprime(c)) c.phase = DEC_PH_TYPE;
            break;
        case DEC_PH_TYPE: {
            ck_take(c, k);
            uint32_t type = dget_int(c);
            if (c.need) { ck_rollback(c, k); break; }
            c.type = type; c.i = 0; c.copied = 0; c.copied_from = c.wnd_pos;
            if (type == DT_NORMAL || type == DT_EXE) c.phase = DEC_PH_LZ;
            else if (type == DT_ENGTXT || type == DT_BAD || type == DT_ENTROPY || (type >= DT_DLT && type < DT_DLT + 5)) c.phase = DEC_PH_SIZE;
            else if (type == SIG_EOF) { c.out_size = 0; c.phase = DEC_PH_TAIL; }
            else c.status = DEC_ERR_DECODE;
        } break;
        case DEC_PH_SIZE: {
            ck_take(c, k);
            uint32_t sz = dget_int(c);
            if (c.need) { ck_rollback(c, k); break; }
            c.run_size = sz;
            if (c.type == DT_ENGTXT) { c.copied_from = c.wnd_pos; c.phase = DEC_PH_LZ; }
            else if (sz > max) c.status = DEC_ERR_MINUS1;
            else c.phase = DEC_PH_RAW;
        } break;
        case DEC_PH_LZ:
            dlz_decode(c, max);
            break;
        case DEC_PH_RAW: {   // decode_bad :98-108, decode_literals :169-185, decode_rle :110-153
            const uint32_t type = c.type, size = c.run_size;
            uint32_t i = c.i, sctx = c.copied;   // `copied` doubles as the RLE context between launches
            if (type >= DT_DLT && !c.p_delta_ready) {
                d_fill_words(c.p_delta, 256 * 256, 2048);
                c.p_delta_ready = 1;
            }
            while (i < size) {
                rc_scalar(c);
                ck_take(c, k);
                uint32_t ni = i + 1, byte = 0, runlen = 0;
                if (type == DT_BAD) byte = ddirect16(c, 8);
                else if (type == DT_ENTROPY) { byte = dbyte_tree_l(c, c.ctx * 256); c.ctx = byte; }
                else if (dbit(c, P_RLE_FLAG) == 0) byte = dbyte_tree_g(c, sctx * 256);
                else { runlen = dmatchlen_2(c) + 11; if (i == 0) c.err = 2; }
                if (c.need) { ck_rollback(c, k); break; }
                if (c.err) { c.status = c.err == 2 ? DEC_ERR_MINUS1 : DEC_ERR_DECODE; break; }
                if (runlen) {
                    uint32_t n = runlen < size - i ? runlen : size - i;
                    uint32_t prev = DUNI((uint32_t)c.out[i - 1]);
                    d_fill_bytes(c.out + i, n, prev);
                    ni = i + n;
                    sctx = prev;
                } else {
                    c.out[i] = (uint8_t)byte;
                    if (type >= DT_DLT) sctx = byte;
                }
                i = ni;
            }
            c.i = i; c.copied = sctx;
            if (!c.need && c.status == DEC_RUNNING) { c.out_size = size; c.i = 0; c.copied = 0; c.phase = DEC_PH_POST; }
        } break;
        case DEC_PH_POST: {
            const uint32_t type = c.type, size = c.out_size;
            DTM_DECL;
            if (type == DT_EXE) dinverse_e89(c.out, size);
            else if (type == DT_ENGTXT) dinverse_dict(c.out, (dgu8 *)c.D->swap, size);
            else if (type >= DT_DLT) { dinverse_delta(c.out, (dgu8 *)c.D->swap, size, type - DT_DLT < 4 ? type - DT_DLT + 1 : 8u); dcopy2dict(c, size); }
            else if (type == DT_BAD || type == DT_ENTROPY) dcopy2dict(c, size);
            DTM_ADD(c, 4);
            c.phase = DEC_PH_TAIL;
        } break;
        case DEC_PH_TAIL: {
            ck_take(c, k);
            uint32_t t = dget_int(c);
            if (c.need) { ck_rollback(c, k); break; }
            if (t == 1) { c.consumed += c.rd[0] + c.rd[1]; c.phase = DEC_PH_PRIME; }
            else { c.phase = DEC_PH_TYPE; done = true; }
        } break;
        case DEC_PH_PRIME:
            if (dprime(c)) { c.phase = DEC_PH_TYPE; done = true; }
            break;
        default:
            c.status = DEC_ERR_DECODE;
        }
    }
    // store the stream state
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t i = c.lane; i < P_COUNT; i += 64) D->probs[i] = lds.P[i];
    {
        typedef uint32_t __attribute__((ext_vector_type(4))) v4u;
        __attribute__((address_space(1))) v4u *dst = (__attribute__((address_space(1))) v4u *)D->p_lit;
        const v4u *src = (const v4u *)lds.plit;
        for (uint32_t i = c.lane; i < 65536 * 2 / 16; i += 64) dst[i] = src[i];
    }
#ifdef CSCMI_TIMERS
    if (c.lane == 0) { for (int i = 0; i < 16; i++) D->dbg[i] += c.tm[i] + lds.hand.tm[i]; D->dbg[7] += __builtin_readcyclecounter() - tk0; D->dbg[15]++; }
#endif
    if (c.lane == 0) {
        for (int i = 0; i < 2; i++) { D->taken[i] = c.taken[i]; D->rd[i] = c.rd[i]; D->fill[i] = c.fill[i]; }
        D->range = c.range; D->code = c.code; D->bc_bits = c.bc_bits; D->bc_val = c.bc_val;
        D->state = c.state; D->ctx = c.ctx; D->wnd_pos = c.wnd_pos; D->consumed = c.consumed;
        for (int i = 0; i < 4; i++) D->rep[i] = c.rep[i];
        D->phase = c.phase; D->type = c.type; D->run_size = c.run_size; D->i = c.i; D->copied = c.copied; D->copied_from = c.copied_from;
        D->out_size = c.out_size; D->p_delta_ready = c.p_delta_ready;
        D->status = c.status != DEC_RUNNING ? c.status : (c.need ? c.need : DEC_DONE);
    }
}

// one stream: CSCDec_Decode
__global__ __launch_bounds__(64) void k_decode_run(DecState *D)
{
    decode_stream((gDecState *)D, g_dec_lds);
}
// many independent streams (the tasks of an archive): workgroup b advances stream b until it needs input or its
// Decompress call is complete; one stream per CU (the literal table fills most of the LDS)
__global__ __launch_bounds__(64) void k_decode_run_multi(DecState *const *states)
{
    // the pointer is the same in every lane; say so, or everything loaded through it is treated as per-lane data
    const uint64_t a = (uint64_t)states[blockIdx.x];
    const uint64_t u = ((uint64_t)DUNI((uint32_t)(a >> 32)) << 32) | DUNI((uint32_t)a);
    decode_stream((gDecState *)u, g_dec_lds);
}

__global__ void k_decode_init(DecState *D)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (uint32_t i = tid; i < 256 * 256 / 2; i += nth) D->p_lit[i] = 2048u | (2048u << 16);   // u16 image of p_lit
    for (uint32_t i = tid; i < P_COUNT; i += nth) D->probs[i] = 2048;
}

void launch_decode_init(DecState *D, hipStream_t st) { hipLaunchKernelGGL(k_decode_init, dim3(64), dim3(256), 0, st, D); }
void launch_decode_run(DecState *D, hipStream_t st)
{
    hipLaunchKernelGGL(k_decode_run, dim3(1), dim3(64), 0, st, D);
}
void launch_decode_run_multi(DecState *const *states, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_decode_run_multi, dim3(n), dim3(64), 0, st, states);
}

}  // namespace cscmi
