// csc_kernels.hip -- gfx950 (CDNA4) kernels for the libcsc encode path.
//
// Design (DESIGN.md has the long form):
//  * k_analyze / k_dup_check are wide kernels: one workgroup per 8 KiB block
//     -- these are the embarrassingly parallel
//    stages.
//  * k_encode_runs is ONE WORKGROUP per stream.  A libcsc stream is a strictly
//    serial dependency chain (adaptive probabilities -> prices -> parse ->
//    match-finder state), so a wavefront executes that protocol with
//    wave-uniform control flow while its 64 lanes cooperate on the byte work
//    inside each step: all match candidates of a position are extended at once
//    (4 lanes x 8 bytes per candidate, __ballot + ctz), hash-bucket gathers and
//    shift-inserts are one vector memory op, price tables and the parser's DP
//    relaxation are filled one length per lane, and the order-1 literal coder
//    fetches/updates its 8 probabilities in 8 lanes before the (serial)
//    range-coder arithmetic.  Where the protocol leaves room, more wavefronts
//    of the workgroup take part: the lazy levels run a parse wavefront feeding
//    a coder wavefront through an LDS token ring (csc_kernels_lz.inc), the
//    advanced parser of the hash-table levels runs up to four parse wavefronts
//    that take the DP nodes of a window in turn (csc_kernels_dp2.inc).  Small
//    adaptive tables, the DP nodes and the word trie live in LDS; window, hash
//    tables / binary tree and p_lit live in HBM.  Independent streams (the
//    archiver's -p / per-extension tasks) run as independent workgroups on
//    other CUs / GPUs.
//
// No MFMA here: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "csc_device.h"
#include "csc_tables.h"

namespace cscmi {

#define DEV __device__ __forceinline__
// Everything the kernels touch outside LDS is HBM: typing those pointers as address_space(1)
// makes the compiler emit global_* instead of flat_* memory instructions (flat ones also tick
// lgkmcnt and cannot be told apart from LDS traffic by the waitcnt logic).
#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS uint8_t gu8;
typedef GLOBAL_AS uint16_t gu16;
typedef GLOBAL_AS uint32_t gu32;
typedef uint32_t __attribute__((ext_vector_type(4))) raw_vec4;
typedef GLOBAL_AS raw_vec4 gvec4;
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

// The kernels are built as two code objects (see the Makefile): CSCMI_TU 1 = the advanced parser of the hash-table levels (the
// headline path) with everything shared, CSCMI_TU 2 = the other encode kernels; 0 = everything in one (development builds).
// Same ISA either way, but the level-3 kernels run 7 % faster out of the small code object (measured, reproducibly).
#ifndef CSCMI_TU
#define CSCMI_TU 0
#endif
static __device__ __constant__ uint32_t d_p2bits[512];
static __device__ __constant__ uint32_t d_logtable[513];

__device__ static const uint32_t kDltIndex[5] = {1, 2, 3, 4, 8};   // csc_typedef.h:36

// ------------------------------------------------------------------------------------------
// LDS image of one stream while k_encode_runs is resident (< 40 KiB: four streams per CU fit in 160 KiB)
constexpr uint32_t kTokRing = 2048;
constexpr uint32_t kDpWaves = 4;          // parse wavefronts per stream the advanced parser can use (csc_kernels_dp2.inc)
// Advanced parser on two parse wavefronts (csc_kernels_dp2.inc): what the wavefront that visited DP node k leaves for the
// visitor of node k + 1, and the control words of a DP window.
struct DpEdge {
    uint32_t lit, p1;          // price of reaching node k + 1 by a literal / by rep0len1 from node k (0xFFFFFFFF = no such edge)
    uint32_t state, reach;     // coder state of node k; max over visited j of j + longest candidate (the reference's apend - 1)
    uint32_t a0l, a0d;         // the longest candidate at node k (what a match exit encodes)
    uint32_t rep[4];           // rep distances of node k
};
struct DpShared {
    uint32_t cmd, idle;        // master -> helper: window number (kPipeQuit at kernel end); helper -> master: window left
    uint32_t dec;              // ((k + 1) << 2) | way out: decision of node k (0 = the window goes on)
    uint32_t price_done, relax_done;   // k + 1 once node k's length pricing / relaxation is complete (both are ordered across nodes)
    uint32_t ins_done, erep_done;      // k + 1 once position k's table insert has completed / node k's rep distances + state are in er[]
    uint32_t lp_int;           // GetMatchLenPrice's refresh countdown while a window is open
    uint32_t abort, gate;      // gate: see dp_undo
    uint32_t wpos0, pos0, aplimit, limit0, stage_base, stage_end;   // the open window
    uint32_t bstat[2];
    DpEdge e[2];               // mailbox of node k at [k & 1]
    uint16_t st_lit[64], st_p1[64];   // per coder state: price of the literal flag / of the rep0len1 flags, while a window is open
    uint32_t e2[4][8];         // chain variant: node k's length-2 edge at [k & 3] (price, label, state, rep distances; [7] = k + 1 when written)
    uint32_t er[4][8];         // rep distances [0..3] and coder state [4] of node k at [k & 3], published as soon as its label is final
};
struct EncLds {
    uint32_t P[P_COUNT + 4];                  // small adaptive probability tables
    uint16_t p2b[512];                        // probability -> price (1/128 bit; at most 12 * 128)
    uint32_t len_price[32], len_price_old[32];
    uint32_t rep[4];                          // rep_dist_[4] (live)
    union {
        uint32_t cd2[kDpWaves][32];               // candidate distances of the position being searched: 0-3 rep, 4 HT2, 5 HT3, 6 BT head, 7.. bucket
                                                  // ([wavefront]: the advanced parser of the hash-table configurations runs several parse wavefronts)
        struct { uint32_t cd2_row0[32]; uint32_t cmp_pos[16], cmp_lim[16], cmp_res[16]; };   // wide-bucket / binary-tree configurations run one
                                                  // wavefront: their compare scratch lies over the unused rows
    };
    uint32_t rp42[kDpWaves][4];                     // the four rep-index prices of the position being priced ([wavefront])
    // The parser's DP nodes (APUnit, csc_lz.h:33-41) as a 256-slot ring + a per-node log.  A node is
    // relabelled only from nodes before it and only up to good_len - 1 <= 254 positions ahead, so
    // the live frontier (price, label, coder state, rep distances) fits a ring indexed by node & 255;
    // what the back-trace needs (label, back pointer, literal) is logged when the node is visited.
    uint32_t rg_price[256], rg_dist[256];
    uint32_t rg_rep[256 * 4];
    uint16_t rg_back[256];
    uint8_t rg_state[256];
    union {
        struct {
            uint32_t fin_dist[kAPLimit + 1];
            uint16_t fin_back[kAPLimit + 2];
            union {
                uint16_t fin_next[kAPLimit + 2];                              // back-trace only (on the way out)
                struct { uint32_t appt_price[256], appt_dist[256]; };        // price table beyond 64 lengths (custom good_len > 66)
            };
            uint8_t fin_lit[kAPLimit + 3];
        };
        uint32_t fin2[(kAPLimit + 1) * 2];                                  // level-3 pipeline form (csc_kernels_dp4.inc): its log, {distance code, back | state << 16} per node
        struct { uint16_t trie_next[300 * 26]; uint8_t trie_sym[304]; };     // word trie, only while the dictionary filter runs
        struct {   // lazy levels: parse wavefront -> coder wavefront (see "token pipe"); the control words sit behind the
                   // ring, i.e. beyond the 15.9 KiB the word trie uses, because the coder wavefront polls them while the
                   // dictionary filter owns the front of this union
            uint2 tok[kTokRing];
            uint32_t pipe_cmd, pipe_ack, tok_head, tok_tail;
            uint32_t hand[16];                // coder-side scalars handed between the two wavefronts
        };
    };
    DpShared dp;                              // advanced parser, two parse wavefronts: control words + edge mailboxes
    // the sub-block being parsed: stage[j] = wnd[stage_base + j - 16] (16 bytes of history, 48 of look-ahead)
    uint32_t stage[(kMinBlock + 64) / 4];
};

// wave-uniform scalar state of the stream (SGPR/VGPR resident; spilled back to EncState at exit)
struct Sc {
    EncState *S;
    EncLds *L;
    void *X;                         // Dp4X (csc_kernels_dp4.inc) when the kernel has one
    struct CoderQ *Q;                // pipeline form: range / bit coder arithmetic runs on a wavefront of its own, fed through this queue
    uint32_t q_head, q_room;
    gu8 *wnd;
    uint32_t wnd_size, vld_rge;
    gu32 *ht2, *ht3, *ht6, *bt_head, *bt_nodes, *p_lit, *p_delta, *mfbuf;
    uint32_t ht6_off, bth_off;       // word offsets of ht6 / bt_head inside mfbuf (ht2 at 0, ht3 at kHT2Size)
    bool fast_ht;                    // hash-table-only configuration: vector replay path
    uint32_t pr_off, pr_rmask, pr_pol;   // per-lane constants of the 12-term state price gather
    uint32_t gm6, gm2, gm3, gmb, gslot;   // per-lane constants of the entry gather (lane masks, destination slot)
    uint32_t ht_bits, ht_width, ht_low, ht_cyc, bt_bits, bt_size, bt_cyc, good_len;
    uint32_t lz_good_len, lz_bt_cyc, lz_ht_cyc;
    uint32_t bt_pos, pos, wnd_curpos;
    uint32_t state, ctx, lp_rebuild_int;
    uint64_t rc_low;
    uint32_t rc_range, rc_cache, rc_cachesize, rc_size, bc_size, bc_curbits, bc_curval, bsize;
    gu8 *rc_buf, *bc_buf, *arena, *swapbuf;
    uint32_t arena_used, arena_cap, error;
    uint32_t lane;
    uint32_t stage_base, stage_end;   // window positions covered by L->stage: [stage_base - 16, stage_end)
    uint32_t cand_len_v, cand_dist_v; // mfcand_[1..]: candidate j lives in lane j of these two VGPRs
    uint32_t wv, dp_seq, dp_idle, dp_seq_seen;              // which parse wavefront this is (0 master, 1 helper: scratch row in cd2 / rp42); DP windows opened so far
    uint32_t st_find, st_slide, st_bt, st_lit, st_match;
    uint32_t pipe_ok, piped, tok_headl, tok_tail_seen, pipe_seq, pair_ok;   // token pipe (parse side): active?, local head, last tail seen, hand-over number
#ifdef CSCMI_TIMERS
    unsigned long long tm[16];
#endif
};

// development-only section timers (s_memtime); compiled out of the product build
#ifdef CSCMI_TIMERS
#define TM_DECL unsigned long long tm__ = __builtin_readcyclecounter()
#define TM_ADD(c, k) do { unsigned long long n__ = __builtin_readcyclecounter(); (c).tm[k] += n__ - tm__; tm__ = n__; } while (0)
#define TM_RESET tm__ = __builtin_readcyclecounter()
// (-DCSCMI_TIMERS_FINE on top: stamps per packet / per DP node as well -- some hundred cycles a node of their own, which is why the default development build
// leaves them out: its sections are per window, a dozen s_memtime in ~80 k cycles, and its speed is the product's to a per cent or two)
#ifdef CSCMI_TIMERS_FINE
#define TMF_ADD(c, k) TM_ADD(c, k)
#else
#define TMF_ADD(c, k) do {} while (0)
#endif
#else
#define TMF_ADD(c, k) do {} while (0)
#define TM_DECL do {} while (0)
#define TM_ADD(c, k) do {} while (0)
#define TM_RESET do {} while (0)
#endif

DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// a 64-bit value / a pointer that is the same in every lane, told to the compiler (two v_readfirstlane)
DEV uint64_t uni64(uint64_t v) { return ((uint64_t)UNI((uint32_t)(v >> 32)) << 32) | UNI((uint32_t)v); }
template <class T> DEV T *uni_gptr(T *p) { return (T *)(uintptr_t)uni64((uint64_t)(uintptr_t)p); }
// Every scalar of the context through v_readfirstlane: whatever a lane-dependent branch somewhere upstream (a filter's lane-strided
// loop, the block walk) has made "per-lane data" in the compiler's eyes is wave-uniform again from here on.  The serial loops
// call it at their head (a sub-block, a way out): some seventy VALU moves against thousands of instructions -- and everything
// computed from these values inside the loop stays on the scalar unit (see emit_block).  Per-lane fields (cand_*_v, pr_*, gm*) are
// left alone.
DEV void sc_uniform(Sc &c)
{
    c.S = uni_gptr(c.S); c.wnd = uni_gptr(c.wnd); c.wnd_size = UNI(c.wnd_size); c.vld_rge = UNI(c.vld_rge);
    c.ht2 = uni_gptr(c.ht2); c.ht3 = uni_gptr(c.ht3); c.ht6 = uni_gptr(c.ht6); c.bt_head = uni_gptr(c.bt_head); c.bt_nodes = uni_gptr(c.bt_nodes);
    c.p_lit = uni_gptr(c.p_lit); c.p_delta = uni_gptr(c.p_delta); c.mfbuf = uni_gptr(c.mfbuf);
    c.ht6_off = UNI(c.ht6_off); c.bth_off = UNI(c.bth_off);
    c.ht_bits = UNI(c.ht_bits); c.ht_width = UNI(c.ht_width); c.ht_low = UNI(c.ht_low); c.ht_cyc = UNI(c.ht_cyc);
    c.bt_bits = UNI(c.bt_bits); c.bt_size = UNI(c.bt_size); c.bt_cyc = UNI(c.bt_cyc); c.good_len = UNI(c.good_len);
    c.lz_good_len = UNI(c.lz_good_len); c.lz_bt_cyc = UNI(c.lz_bt_cyc); c.lz_ht_cyc = UNI(c.lz_ht_cyc);
    c.bt_pos = UNI(c.bt_pos); c.pos = UNI(c.pos); c.wnd_curpos = UNI(c.wnd_curpos);
    c.state = UNI(c.state); c.ctx = UNI(c.ctx); c.lp_rebuild_int = UNI(c.lp_rebuild_int);
    c.rc_low = uni64(c.rc_low); c.rc_range = UNI(c.rc_range); c.rc_cache = UNI(c.rc_cache); c.rc_cachesize = UNI(c.rc_cachesize);
    c.rc_size = UNI(c.rc_size); c.bc_size = UNI(c.bc_size); c.bc_curbits = UNI(c.bc_curbits); c.bc_curval = UNI(c.bc_curval); c.bsize = UNI(c.bsize);
    c.rc_buf = uni_gptr(c.rc_buf); c.bc_buf = uni_gptr(c.bc_buf); c.arena = uni_gptr(c.arena); c.swapbuf = uni_gptr(c.swapbuf);
    c.arena_used = UNI(c.arena_used); c.arena_cap = UNI(c.arena_cap); c.error = UNI(c.error);
    c.stage_base = UNI(c.stage_base); c.stage_end = UNI(c.stage_end);
    c.wv = UNI(c.wv); c.dp_seq = UNI(c.dp_seq); c.dp_idle = UNI(c.dp_idle); c.dp_seq_seen = UNI(c.dp_seq_seen);
    c.st_find = UNI(c.st_find); c.st_slide = UNI(c.st_slide); c.st_bt = UNI(c.st_bt); c.st_lit = UNI(c.st_lit); c.st_match = UNI(c.st_match);
    c.q_head = UNI(c.q_head); c.q_room = UNI(c.q_room);
}

DEV void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
DEV uint32_t rdlane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// write a uniform value into one lane of a per-lane register (v_cmp + v_cndmask)
DEV uint32_t wrlane(uint32_t val, uint32_t lane, uint32_t old) { return (threadIdx.x & 63u) == lane ? val : old; }

// 8 bytes at an arbitrary byte address through three aligned dword loads + v_alignbyte
DEV uint64_t load8u(const gu8 *p)
{
    uintptr_t a = (uintptr_t)p;
    uint32_t sh = (uint32_t)a & 3u;
    const gu32 *q = (const gu32 *)(a & ~(uintptr_t)3);
    uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
}
DEV uint32_t ldb(const gu8 *p) { return UNI((uint32_t)*p); }

// the same 8-byte fetch from the LDS copy of the current sub-block (window position wpos)
DEV uint64_t stage_load8(const uint32_t *stage, uint32_t byte_idx)
{
    uint32_t sh = byte_idx & 3u;
    const uint32_t *q = stage + (byte_idx >> 2);
    uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
}

// ==========================================================================================
// output arena + coder (csc_coder.cpp, csc_memio.cpp:83-108 is finished on the host)

// Lane-dependent branches and the scalars of a serial loop do not go together: the compiler's uniformity analysis makes per-lane
// data of EVERY value that is merged in a block where the two sides of such a branch meet, and after block merging those are the
// latches of the loops around it -- one `if (lane == 0)` in a coder loop put low / range / cache and the buffer positions into
// vector registers and made exec-mask branches of its tests (round 5, tools/enc_uniformity.py; csc_dec_kernels.hip has the longer
// story).  So inside such loops: every lane stores the same word to the same place, a lane with nothing to store stores to a dump
// slot, and lane-strided loops are functions of their own.
__device__ __noinline__ void lanes_copy16(gvec4 *d4, const gvec4 *s4, uint32_t n16)
{
    for (uint32_t i = threadIdx.x & 63u; i < n16; i += 64) d4[i] = s4[i];
}
__device__ __noinline__ void lanes_copy8(gu8 *dst, const gu8 *src, uint32_t n)
{
    for (uint32_t t = threadIdx.x & 63u; t < n; t += 64) dst[t] = src[t];
}
__device__ __noinline__ void lanes_fill32(gu32 *dst, uint32_t n, uint32_t v)
{
    for (uint32_t t = threadIdx.x & 63u; t < n; t += 64) dst[t] = v;
}
// bytes [first, end) of the sub-block's LDS stage from the window (src[t] is the byte of stage position t)
__device__ __noinline__ void lanes_stage(uint8_t *sb, const gu8 *src, uint32_t first, uint32_t end)
{
    for (uint32_t t = first + (threadIdx.x & 63u); t < end; t += 64) sb[t] = src[t];
}
// hand a finished RC/BC buffer to the host: header + 16-byte-lane copy
DEV void emit_block(Sc &c, uint32_t kind, const gu8 *buf, uint32_t size)
{
    uint32_t need = 16 + ((size + 15) & ~15u);
    if (c.arena_used + need > c.arena_cap) { c.error = ERR_ARENA_FULL; return; }
    gu8 *dst = c.arena + c.arena_used;
    wave_fence();   // the byte stores into buf came from uniform code; the copy below is per-lane
    ((gu32 *)dst)[0] = kind; ((gu32 *)dst)[1] = size;       // (every lane the same words)
    lanes_copy16((gvec4 *)(dst + 16), (const gvec4 *)buf, (size + 15) >> 4);
    c.arena_used += need;
}

// ------------------------------------------------------------------------------------------
// Coder queue (pipeline form, csc_kernels_dp4.inc).  What the adaptive model needs from a coded bit is its probability
// update; the carry-less range arithmetic (csc_coder.h:67-81, csc_coder.cpp:76-112) needs only (probability before the
// update, bit) and is a serial chain of its own.  So the master wavefront only updates probabilities and queues
//   0x80000000 | bit << 12 | p          EncodeBit's arithmetic
//   0x40000000 | nbits << 16 | value    EncDirect16
//   0x20000000                          Coder::Flush
// in coding order; the coder wavefront owns low / range / cache / the two block buffers and the output arena.
constexpr uint32_t kCoderQ = 4096;
constexpr uint32_t kPDumpE = P_COUNT + 3;     // EncLds::P has four words of slack: where lanes with nothing to store store
struct CoderQ {
    uint32_t pub, tail, done, pad;
    uint32_t e[kCoderQ];
};
DEV void q_publish(Sc &c)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    *(volatile __attribute__((address_space(3))) uint32_t *)&c.Q->pub = c.q_head;
}
// wait until at most kCoderQ - `need` entries are outstanding; -> entries outstanding.  Bounded: a coder wavefront that has stopped
// consuming turns into an error code (0x40200000 | ...) instead of a hung kernel
constexpr uint32_t ERR_QUEUE_STALL = 0x40200000u;
DEV uint32_t q_wait_room(Sc &c, uint32_t need)
{
    uint32_t used, spins = 0;
    if (__builtin_expect(c.error != 0, 0)) return 0;      // already gave up once: do not spin for every later entry (the caller ends the kernel at its next check)
    while ((used = c.q_head - UNI(*(volatile __attribute__((address_space(3))) uint32_t *)&c.Q->tail)) > kCoderQ - need) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) { if (!c.error) c.error = ERR_QUEUE_STALL | (used & 0xFFFFu); return 0; }
    }
    return used;
}
DEV void q_push(Sc &c, uint32_t e)
{
    if (__builtin_expect(c.q_room == 0, 0)) {
        q_publish(c);
        c.q_room = umin(kCoderQ - 64u - q_wait_room(c, 128u), 256u);
    }
    c.Q->e[c.q_head & (kCoderQ - 1)] = e;
    c.q_head++; c.q_room--;
}

// `n` EncodeBit entries at once: lane from + k holds decision k (its probability before the update, its bit)
DEV void q_push_bits(Sc &c, uint32_t pold, uint32_t bit, uint32_t n, uint32_t from = 0)
{
    if (__builtin_expect(c.q_room < n, 0)) {
        q_publish(c);
        c.q_room = umin(kCoderQ - 64u - q_wait_room(c, 128u), 256u);
    }
    // (no branch: the lanes that hold no decision write to the entry 63 behind the head -- q_room always leaves 64 entries spare,
    // and every entry is written again before the head passes it)
    const uint32_t k = c.lane - from;
    c.Q->e[(c.q_head + (k < n ? k : 63u)) & (kCoderQ - 1)] = 0x80000000u | (bit ? 0x1000u : 0u) | pold;
    c.q_head += n; c.q_room -= n;
}

DEV void rc_put(Sc &c, uint32_t byte)
{
    c.rc_buf[c.rc_size++] = (uint8_t)byte;
    if (c.rc_size == c.bsize) { emit_block(c, 1, c.rc_buf, c.bsize); c.rc_size = 0; }
}

// Coder::RC_ShiftLow, csc_coder.cpp:89-112
DEV void rc_shift_low(Sc &c)
{
    uint32_t low32 = (uint32_t)c.rc_low, top = (uint32_t)(c.rc_low >> 32);
    if (low32 < 0xFF000000u || top != 0) {
        uint32_t temp = c.rc_cache;
        do {
            rc_put(c, (temp + top) & 0xFF);
            temp = 0xFF;
        } while (--c.rc_cachesize != 0);
        c.rc_cache = low32 >> 24;
    }
    c.rc_cachesize++;
    c.rc_low = (uint64_t)(low32 << 8);
}

// arithmetic half of EncodeBit (csc_coder.h:67-81); p is the probability BEFORE its update
DEV void rc_code(Sc &c, uint32_t v, uint32_t p)
{
    if (c.Q) { q_push(c, 0x80000000u | (v ? 0x1000u : 0u) | p); return; }
    // selects, not branches: a taken scalar branch costs a lone wavefront ~35 cycles (tools/ub/ub_issue.hip)
    const uint32_t bound = (c.rc_range >> 12) * p;
    c.rc_low += v ? 0u : bound;
    c.rc_range = v ? bound : c.rc_range - bound;
    if (__builtin_expect(c.rc_range < (1u << 24), 0)) { c.rc_range <<= 8; rc_shift_low(c); }
}
DEV uint32_t p_update(uint32_t v, uint32_t p) { return v ? p + ((0xFFFu - p) >> 5) : p - (p >> 5); }

// EncodeBit on an LDS-resident probability
DEV void enc_bit_lds(Sc &c, uint32_t v, uint32_t idx)
{
    uint32_t p = UNI(c.L->P[idx]);
    c.L->P[idx] = p_update(v, p);
    rc_code(c, v, p);
}

DEV void bc_put(Sc &c, uint32_t byte)
{
    c.bc_buf[c.bc_size++] = (uint8_t)byte;
    if (c.bc_size == c.bsize) { emit_block(c, 0, c.bc_buf, c.bsize); c.bc_size = 0; }   // BCWCheckBound
}

// Coder::EncDirect16, csc_coder.cpp:76-87
DEV void enc_direct16(Sc &c, uint32_t val, uint32_t len)
{
    if (c.Q) { q_push(c, 0x40000000u | (len << 16) | (val & 0xFFFFu)); return; }
    c.bc_curval = (c.bc_curval << len) | val;
    c.bc_curbits += len;
    while (c.bc_curbits >= 8) {
        bc_put(c, (c.bc_curval >> (c.bc_curbits - 8)) & 0xFF);
        c.bc_curbits -= 8;
    }
}
DEV void enc_direct(Sc &c, uint32_t v, uint32_t l)   // EncodeDirect, csc_coder.h:83-88
{
    if (l <= 16) enc_direct16(c, v, l);
    else { enc_direct16(c, v >> 16, l - 16); enc_direct16(c, v & 0xFFFF, 16); }
}

// Coder::Flush, csc_coder.cpp:40-74: the byte at rc_buf[rc_size] is NOT stored -- it keeps
// whatever the persistent buffer held; rc_buf is zero-initialised once.
DEV void coder_flush(Sc &c)
{
    if (c.Q) { q_push(c, 0x20000000u); q_publish(c); return; }
    for (int i = 0; i < 5; i++) rc_shift_low(c);
    c.rc_size++;
    bc_put(c, (c.bc_curval << (8 - c.bc_curbits)) & 0xFF);
    bc_put(c, 0);
    emit_block(c, 1, c.rc_buf, c.rc_size);
    emit_block(c, 0, c.bc_buf, c.bc_size);
    c.rc_low = 0; c.rc_range = 0xFFFFFFFFu; c.rc_cachesize = 1; c.rc_cache = 0;
    c.rc_size = c.bc_size = 0; c.bc_curbits = c.bc_curval = 0;
}

// ==========================================================================================
// Model (csc_model.cpp)

DEV uint32_t bit_price(const Sc &c, uint32_t v, uint32_t p)   // FEncodeBit, csc_model.cpp:161-167
{
    return UNI(c.L->p2b[(v ? p : 4096u - p) >> 3]);
}
DEV uint32_t lds_price(const Sc &c, uint32_t v, uint32_t idx) { return bit_price(c, v, UNI(c.L->P[idx])); }

// Model::EncodeInt, csc_model.cpp:389-414
DEV void encode_int(Sc &c, uint32_t num)
{
    uint32_t slot = num ? 31u - (uint32_t)__builtin_clz(num) : 0u;
    enc_direct(c, slot, 5);
    if (slot == 0) enc_direct(c, num, 1);
    else enc_direct(c, num - (1u << slot), slot);
}

// Model::encode_matchlen_1, csc_model.cpp:113-145
DEV void encode_matchlen_1(Sc &c, uint32_t len)
{
    if (len < 16) {
        uint32_t base;
        if (len < 8) { enc_bit_lds(c, 0, P_LEN_SLOT); base = P_LEN_X1; }
        else { enc_bit_lds(c, 1, P_LEN_SLOT); enc_bit_lds(c, 0, P_LEN_SLOT + 1); len -= 8; base = P_LEN_X2; }
        uint32_t cc = len | 0x08;
        do { enc_bit_lds(c, (cc >> 2) & 1, base + (cc >> 3)); cc <<= 1; } while (cc < 0x40);
    } else {
        enc_bit_lds(c, 1, P_LEN_SLOT);
        enc_bit_lds(c, 1, P_LEN_SLOT + 1);
        len -= 16;
        uint32_t cc = len | 0x80;
        do { enc_bit_lds(c, (cc >> 6) & 1, P_LEN_X3 + (cc >> 7)); cc <<= 1; } while (cc < 0x4000);
    }
}

// Model::encode_matchlen_2, csc_model.cpp:147-159
DEV void encode_matchlen_2(Sc &c, uint32_t len)
{
    if (len >= 143) {
        encode_matchlen_1(c, 143);
        len -= 143;
        while (len >= 143) { len -= 143; enc_bit_lds(c, 0, P_LONGLEN); }
        enc_bit_lds(c, 1, P_LONGLEN);
    }
    encode_matchlen_1(c, len);
}

// 8 binary decisions of one byte under an order-1 row in HBM.  The 8 tree nodes are distinct
// and known up front, so 8 lanes fetch + update them in one round trip; only the range-coder
// arithmetic is serial.  (csc_model.cpp:176-183, 452-459, 504-509)
DEV void encode_byte_tree(Sc &c, gu32 *row, uint32_t sym)
{
    uint32_t cc = sym | 0x100;
    uint32_t k = c.lane & 7;
    uint32_t idx = cc >> (8 - k), bit = (cc >> (7 - k)) & 1;
    uint32_t pold = row[idx];
    *(c.lane < 8 ? row + idx : (gu32 *)c.S->pf_sink + c.lane) = p_update(bit, pold);      // (lanes 8-63 to the dump: no lane-dependent branch)
#pragma unroll
    for (int j = 0; j < 8; j++) rc_code(c, (cc >> (7 - j)) & 1, rdlane(pold, j));
}

// Model::EncodeLiteral, csc_model.cpp:169-183.  The eight tree probabilities are requested first (an HBM / L2 round trip) and
// the flag bit is coded while they travel.
DEV void encode_literal(Sc &c, uint32_t sym)
{
    gu32 *row = c.p_lit + c.ctx * 256;
    const uint32_t cc = sym | 0x100;
    if (c.Q) {
        // the flag (lane 0) and the eight decisions of the symbol's tree walk (lanes 1..8) as one gather / update / scatter and
        // one store into the coder wavefront's queue
        const uint32_t k = (c.lane - 1u) & 7u;
        const uint32_t idx = cc >> (8 - k), bit = c.lane == 0 ? 0u : (cc >> (7 - k)) & 1;
        const uint32_t fidx = P_STATE + c.state * 3;
        // (loads by every lane, stores to a dump slot from the lanes that hold nothing: no lane-dependent branch, see emit_block)
        const uint32_t pf = c.L->P[fidx], pr = row[idx];
        const uint32_t pold = c.lane == 0 ? pf : c.lane < 9 ? pr : 0u;
        const uint32_t pnew = p_update(bit, pold);
        c.L->P[c.lane == 0 ? fidx : kPDumpE] = pnew;
        gu32 *const dst = (c.lane - 1u < 8u) ? row + idx : (gu32 *)c.S->pf_sink + c.lane;
        *dst = pnew;
        q_push_bits(c, pold, bit, 9);
        c.state = (c.state * 4) & 0x3F;
        c.ctx = sym;
        c.st_lit++;
        return;
    }
    const uint32_t k = c.lane & 7;
    const uint32_t idx = cc >> (8 - k), bit = (cc >> (7 - k)) & 1;
    const uint32_t pold = row[idx];
    enc_bit_lds(c, 0, P_STATE + c.state * 3);
    c.state = (c.state * 4) & 0x3F;
    c.ctx = sym;
    *(c.lane < 8 ? row + idx : (gu32 *)c.S->pf_sink + c.lane) = p_update(bit, pold);      // (lanes 8-63 to the dump: no lane-dependent branch)
#pragma unroll
    for (int j = 0; j < 8; j++) rc_code(c, (cc >> (7 - j)) & 1, rdlane(pold, j));
    c.st_lit++;
}

// Model::GetLiteralPrice, csc_model.cpp:185-196 -- 8 lanes gather, 3 xor-shuffles reduce
DEV uint32_t literal_price(const Sc &c, uint32_t fstate, uint32_t fctx, uint32_t sym)
{
    uint32_t cc = sym | 0x100;
    uint32_t k = c.lane & 7;
    uint32_t p = c.p_lit[fctx * 256 + (cc >> (8 - k))];
    uint32_t bit = (cc >> (7 - k)) & 1;
    uint32_t pr = c.L->p2b[(bit ? p : 4096u - p) >> 3];
    pr += __shfl_xor(pr, 1);
    pr += __shfl_xor(pr, 2);
    pr += __shfl_xor(pr, 4);
    return UNI(pr) + lds_price(c, 0, P_STATE + fstate * 3);
}

// ------------------------------------------------------------------------------------------
// A packet's binary decisions under the small (LDS) tables are all known before the first one is coded, and --
// lengths below 143 -- no probability is used twice inside one packet.  So they are laid out one per lane
// (decision k in lane k: table index, bit), fetched with ONE LDS gather, updated by one vector operation and
// stored with one scatter; only the range arithmetic stays serial, fed from registers.  The order of the
// decisions, and where the direct bits of a distance fall between them, is the reference's.
struct Decisions {
    uint32_t idx, bit;   // per lane
    uint32_t n;          // uniform: decisions so far
};
DEV void dec_one(const Sc &c, Decisions &d, uint32_t idx, uint32_t bit)
{
    d.idx = wrlane(idx, d.n, d.idx);
    d.bit = wrlane(bit, d.n, d.bit);
    d.n++;
}
// the nb decisions of an MSB-first binary tree walk of the nb-bit value x; node numbering 1, 2|b, 4|bb, ...; table index = base + node
DEV void dec_tree(const Sc &c, Decisions &d, uint32_t base, uint32_t x, uint32_t nb)
{
    const uint32_t j = c.lane - d.n;                 // level of this lane (wraps for lanes below d.n)
    const bool in = j < nb;
    const uint32_t node = (1u << (j & 31u)) | (x >> ((nb - j) & 31u));
    const uint32_t b = (x >> ((nb - 1u - j) & 31u)) & 1u;
    d.idx = in ? base + node : d.idx;
    d.bit = in ? b : d.bit;
    d.n += nb;
}
// encode_matchlen_1 (csc_model.cpp:113-145) as decisions; len < 143 + 1
DEV void dec_matchlen_1(const Sc &c, Decisions &d, uint32_t len)
{
    if (len < 8) { dec_one(c, d, P_LEN_SLOT, 0); dec_tree(c, d, P_LEN_X1, len, 3); }
    else if (len < 16) { dec_one(c, d, P_LEN_SLOT, 1); dec_one(c, d, P_LEN_SLOT + 1, 0); dec_tree(c, d, P_LEN_X2, len - 8, 3); }
    else { dec_one(c, d, P_LEN_SLOT, 1); dec_one(c, d, P_LEN_SLOT + 1, 1); dec_tree(c, d, P_LEN_X3, len - 16, 7); }
}
// gather + update + scatter; returns the probabilities as they were
DEV uint32_t dec_apply(Sc &c, const Decisions &d)
{
    // (gather by every lane, scatter to the dump slot from the lanes that hold no decision: no lane-dependent branch)
    const bool on = c.lane < d.n;
    const uint32_t i = on ? d.idx : kPDumpE;
    const uint32_t p = c.L->P[i];
    c.L->P[i] = p_update(d.bit, p);
    return on ? p : 0u;
}
// range-code decisions [0, to) from registers (static lane numbers), and the four of a distance's low-bits tree
DEV void dec_code(Sc &c, uint32_t pold, uint64_t bits, uint32_t to)
{
    if (c.Q) { q_push_bits(c, pold, (uint32_t)(bits >> c.lane) & 1u, to); return; }
#pragma unroll
    for (uint32_t k = 0; k < 24; k++) {
        if (k >= to) break;
        rc_code(c, (uint32_t)(bits >> k) & 1u, rdlane(pold, k));
    }
}
DEV void dec_code4(Sc &c, uint32_t pold, uint64_t bits, uint32_t from)
{
    if (c.Q) { q_push_bits(c, pold, (uint32_t)(bits >> c.lane) & 1u, 4, from); return; }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) rc_code(c, (uint32_t)(bits >> (from + k)) & 1u, rdlane(pold, from + k));
}

// Model::EncodeRep0Len1, csc_model.cpp:198-207
DEV void encode_rep0len1(Sc &c)
{
    enc_bit_lds(c, 1, P_STATE + c.state * 3 + 0);
    enc_bit_lds(c, 0, P_STATE + c.state * 3 + 1);
    enc_bit_lds(c, 0, P_STATE + c.state * 3 + 2);
    c.ctx = 0;
    c.state = (c.state * 4 + 2) & 0x3F;
    c.st_match++;
}
DEV uint32_t rep0len1_price(const Sc &c, uint32_t fs)   // csc_model.cpp:209-216
{
    return lds_price(c, 1, P_STATE + fs * 3) + lds_price(c, 0, P_STATE + fs * 3 + 1) + lds_price(c, 0, P_STATE + fs * 3 + 2);
}

// Model::EncodeRepDistMatch, csc_model.cpp:218-232
DEV void encode_rep_match_seq(Sc &c, uint32_t rep_idx, uint32_t match_len);
DEV void encode_rep_match(Sc &c, uint32_t rep_idx, uint32_t match_len)
{
    if (__builtin_expect(match_len >= 143, 0)) { encode_rep_match_seq(c, rep_idx, match_len); return; }
    Decisions d; d.idx = 0; d.bit = 0; d.n = 0;
    const uint32_t s3 = c.state * 3;
    dec_one(c, d, P_STATE + s3, 1); dec_one(c, d, P_STATE + s3 + 1, 0); dec_one(c, d, P_STATE + s3 + 2, 1);
    dec_tree(c, d, P_REPDIST + s3 - 1, rep_idx, 2);
    dec_matchlen_1(c, d, match_len);
    const uint32_t pold = dec_apply(c, d);
    dec_code(c, pold, __ballot(d.bit != 0), d.n);
    c.state = (c.state * 4 + 3) & 0x3F;
    c.st_match++;
}
DEV void encode_rep_match_seq(Sc &c, uint32_t rep_idx, uint32_t match_len)
{
    enc_bit_lds(c, 1, P_STATE + c.state * 3 + 0);
    enc_bit_lds(c, 0, P_STATE + c.state * 3 + 1);
    enc_bit_lds(c, 1, P_STATE + c.state * 3 + 2);
    uint32_t i = 1, j;
    j = (rep_idx >> 1) & 1; enc_bit_lds(c, j, P_REPDIST + c.state * 3 + i - 1); i += i + j;
    j = rep_idx & 1;        enc_bit_lds(c, j, P_REPDIST + c.state * 3 + i - 1);
    encode_matchlen_2(c, match_len);
    c.state = (c.state * 4 + 3) & 0x3F;
    c.st_match++;
}
DEV uint32_t rep_dist_price(const Sc &c, uint32_t fs, uint32_t rep_idx)   // csc_model.cpp:273-284
{
    uint32_t ret = lds_price(c, 1, P_STATE + fs * 3) + lds_price(c, 0, P_STATE + fs * 3 + 1) + lds_price(c, 1, P_STATE + fs * 3 + 2);
    uint32_t i = 1, j;
    j = (rep_idx >> 1) & 1; ret += lds_price(c, j, P_REPDIST + fs * 3 + i - 1); i += i + j;
    j = rep_idx & 1;        ret += lds_price(c, j, P_REPDIST + fs * 3 + i - 1);
    return ret;
}

// dist_table_ slot: largest l with dist_table_[l] <= dist (binary search of csc_model.cpp:328-337;
// dist_table_[l] = 2^(l-2)+1 for l >= 2, so it is a clz)
DEV uint32_t dist_slot(uint32_t dist) { return dist < 3 ? dist : 33u - (uint32_t)__builtin_clz(dist - 1); }

// Model::EncodeMatch, csc_model.cpp:301-366
DEV void encode_match_seq(Sc &c, uint32_t dist, uint32_t len);
DEV void encode_match(Sc &c, uint32_t dist, uint32_t len)
{
    if (__builtin_expect(len >= 143, 0)) { encode_match_seq(c, dist, len); return; }
    Decisions d; d.idx = 0; d.bit = 0; d.n = 0;
    const uint32_t s3 = c.state * 3;
    dec_one(c, d, P_STATE + s3, 1); dec_one(c, d, P_STATE + s3 + 1, 1);
    dec_matchlen_1(c, d, len);
    uint32_t pdist_pos, sbits;
    if (len == 0) { pdist_pos = 0; sbits = 3; }
    else if (len <= 2) { pdist_pos = 16 * (len - 1) + 8; sbits = 4; }
    else if (len <= 5) { pdist_pos = 32 * (len - 3) + 8 + 16 * 2; sbits = 5; }
    else { pdist_pos = 32 * 3 + 8 + 16 * 2; sbits = 5; }
    const uint32_t slot = dist_slot(dist);
    const uint32_t extra_bits = slot > 2 ? slot - 2 : 0;
    dec_tree(c, d, P_DIST + pdist_pos, slot, sbits);
    const uint32_t n_head = d.n;                       // the distance's direct bits go between these and the low-bits tree
    uint32_t extra_len = 0;
    if (extra_bits) {
        extra_len = dist - (1u << extra_bits) - 1;
        dec_tree(c, d, P_DIST_EXTRA + (extra_bits - 1) * 16, __brev(extra_len & 0x0Fu) >> 28, 4);   // rev16_table_, csc_model.cpp:57-62
    }
    const uint32_t pold = dec_apply(c, d);
    const uint64_t bits = __ballot(d.bit != 0);
    dec_code(c, pold, bits, n_head);
    if (extra_bits) {
        if (extra_bits > 4) enc_direct(c, extra_len >> 4, extra_bits - 4);
        dec_code4(c, pold, bits, n_head);
    }
    c.state = (c.state * 4 + 1) & 0x3F;
    c.st_match++;
}
DEV void encode_match_seq(Sc &c, uint32_t dist, uint32_t len)
{
    enc_bit_lds(c, 1, P_STATE + c.state * 3 + 0);
    enc_bit_lds(c, 1, P_STATE + c.state * 3 + 1);
    encode_matchlen_2(c, len);
    uint32_t pdist_pos, sbits;
    if (len == 0) { pdist_pos = 0; sbits = 3; }
    else if (len <= 2) { pdist_pos = 16 * (len - 1) + 8; sbits = 4; }
    else if (len <= 5) { pdist_pos = 32 * (len - 3) + 8 + 16 * 2; sbits = 5; }
    else { pdist_pos = 32 * 3 + 8 + 16 * 2; sbits = 5; }
    uint32_t slot = dist_slot(dist), cc = slot | (1u << sbits);
    uint32_t extra_bits = slot > 2 ? slot - 2 : 0;
    do { enc_bit_lds(c, (cc >> (sbits - 1)) & 1, P_DIST + pdist_pos + (cc >> sbits)); cc <<= 1; } while (cc < (1u << (sbits * 2)));
    if (extra_bits) {
        uint32_t extra_len = dist - (1u << extra_bits) - 1;
        if (extra_bits > 4) enc_direct(c, extra_len >> 4, extra_bits - 4);
        cc = (__brev(extra_len & 0x0Fu) >> 28) | 0x10;   // rev16_table_, csc_model.cpp:57-62
        uint32_t base = P_DIST_EXTRA + (extra_bits - 1) * 16;
        do { enc_bit_lds(c, (cc >> 3) & 1, base + (cc >> 4)); cc <<= 1; } while (cc < (1u << 8));
    }
    c.state = (c.state * 4 + 1) & 0x3F;
    c.st_match++;
}
DEV uint32_t match_dist_price(const Sc &c, uint32_t fs, uint32_t dist)   // csc_model.cpp:368-387
{
    uint32_t l = dist_slot(dist);
    return lds_price(c, 1, P_STATE + fs * 3) + lds_price(c, 1, P_STATE + fs * 3 + 1) + (l > 2 ? l + 2 : 2) * 128;
}

// Model::len_price_rebuild, csc_model.cpp:234-270 -- one length per lane.  A function of its own: its per-lane loops must not meet
// the callers' scalars (see emit_block); it runs once in 4096 price calls.
__device__ __noinline__ void len_price_rebuild_lanes(EncLds *L)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < 32) {
        uint32_t len = lane, ret = 0, cc, base;
        uint32_t s0 = L->P[P_LEN_SLOT], s1 = L->P[P_LEN_SLOT + 1];
        if (len < 16) {
            if (len < 8) { ret += L->p2b[(4096u - s0) >> 3]; base = P_LEN_X1; }
            else { ret += L->p2b[s0 >> 3] + L->p2b[(4096u - s1) >> 3]; len -= 8; base = P_LEN_X2; }
            cc = len | 0x08;
            do {
                uint32_t p = L->P[base + (cc >> 3)];
                ret += L->p2b[(((cc >> 2) & 1) ? p : 4096u - p) >> 3];
                cc <<= 1;
            } while (cc < 0x40);
        } else {
            ret += L->p2b[s0 >> 3] + L->p2b[s1 >> 3];
            len -= 16;
            cc = len | 0x80;
            do {
                uint32_t p = L->P[P_LEN_X3 + (cc >> 7)];
                ret += L->p2b[(((cc >> 6) & 1) ? p : 4096u - p) >> 3];
                cc <<= 1;
            } while (cc < 0x4000);
        }
        L->len_price[lane] = ret;
    }
}
DEV void len_price_rebuild(Sc &c)
{
    len_price_rebuild_lanes(c.L);
    c.lp_rebuild_int = 4096;
}

// Model::CompressLiterals, csc_model.cpp:448-461
DEV void compress_literals(Sc &c, const gu8 *src, uint32_t size)
{
    encode_int(c, size);
    for (uint32_t i = 0; i < size; i++) {
        uint32_t sym = ldb(src + i);
        gu32 *row = c.p_lit + c.ctx * 256;
        c.ctx = sym;
        encode_byte_tree(c, row, sym);
    }
}

// Model::CompressBad, csc_model.cpp:463-469
DEV void compress_bad(Sc &c, const gu8 *src, uint32_t size)
{
    encode_int(c, size);
    for (uint32_t i = 0; i < size; i++) enc_direct16(c, ldb(src + i), 8);
}

// Model::CompressRLE, csc_model.cpp:471-513
DEV void compress_rle(Sc &c, const gu8 *src, uint32_t size)
{
    EncState *S = c.S;
    uint32_t sctx = 0;
    encode_int(c, size);
    if (!UNI(S->p_delta_ready)) {
        lanes_fill32(c.p_delta, 256 * 256, 2048);
        wave_fence();
        S->p_delta_ready = 1;
    }
    for (uint32_t i = 0; i < size;) {
        uint32_t cur = ldb(src + i);
        if (i > 0 && size - i > 3 && ldb(src + i - 1) == cur && cur == ldb(src + i + 1) && cur == ldb(src + i + 2)) {
            uint32_t j = i + 3, len = 3;
            while (j < size && ldb(src + j) == cur) { len++; j++; }
            if (len > 10) {
                sctx = cur;
                len -= 11;
                enc_bit_lds(c, 1, P_RLE_FLAG);
                encode_matchlen_2(c, len);
                i = j;
                continue;
            }
        }
        enc_bit_lds(c, 0, P_RLE_FLAG);
        encode_byte_tree(c, c.p_delta + sctx * 256, cur);
        sctx = cur;
        i++;
    }
}

// level-5 form (csc_kernels_bt.inc): the finder wavefront (bt_finder) was measured slower; kept for the record, not dispatched
constexpr bool kBtFinder = false;
constexpr uint32_t kBtThreads = kBtFinder ? 448 : 256;
DEV void sc_load_regs(Sc &c, EncState *S, EncLds *L);   // csc_kernels_blocks.inc: the register part of a context (the roles' own functions build theirs with it)
#include "csc_kernels_mf.inc"
#include "csc_kernels_lz.inc"
#include "csc_kernels_dp2.inc"
#if CSCMI_TU != 1
#include "csc_kernels_bt.inc"
#include "csc_kernels_hp.inc"
#else
DEV bool hp_ok(const Sc &) { return false; }
DEV void lz_compress_normal_hp(Sc &, uint32_t, bool) {}
DEV void hp_init(Sc &) {}
DEV void hp_inserter(Sc &) {}
DEV void hp_quit(Sc &) {}
DEV void hp_parser_call(Sc &, uint32_t, bool) {}
DEV void hp_inserter_call(Sc &) {}
DEV void hp_coder_call(Sc &) {}
DEV bool bt_ok(const Sc &, uint32_t) { return false; }
DEV void lz_compress_advanced_bt(Sc &, uint32_t) {}
DEV void bt_parser_call(Sc &, uint32_t) {}
DEV void bt_inserter_call(Sc &) {}
DEV void bt_post_call(Sc &) {}
DEV void bt_coder_call(Sc &) {}
DEV void bt_finder_call(Sc &) {}
DEV void bt_init(Sc &, uint32_t) {}
DEV void bt_inserter(Sc &) {}
DEV void bt_post(Sc &) {}
DEV void bt_quit(Sc &) {}
DEV void bt_coder(Sc &) {}
DEV CoderQ *bt_coderq(Sc &) { return nullptr; }
#endif
#if CSCMI_TU != 2
#include "csc_kernels_dp4.inc"
#else
DEV void lz_compress_advanced_dp4(Sc &, uint32_t) {}
DEV void d4_master_call(Sc &, uint32_t) {}
DEV void d4_init(Sc &, uint32_t) {}
DEV void d4_worker(Sc &) {}
DEV void d4_worker_call(Sc &) {}
DEV void d4_quit(Sc &) {}
DEV CoderQ *d4_coderq(Sc &) { return nullptr; }
#endif
#include "csc_kernels_blocks.inc"

}  // namespace cscmi
