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
typedef __attribute__((address_space(1))) uint8_t dgu8;
typedef __attribute__((address_space(1))) uint32_t dgu32;

struct DecLds {
    uint32_t P[P_COUNT + 4];
    uint32_t wtab[128];   // the dictionary filter's words, 4 chars packed per entry (0-terminated)
};

struct Dc {
    DecState *D;
    DecLds *L;
    dgu8 *wnd, *out, *q[2];
    dgu32 *p_lit, *p_delta, *qsize[2], *undo_addr, *undo_val;
    uint32_t wnd_size, bsize, qslots, lane;
    uint32_t avail[2], taken[2], rd[2], fill[2];
    uint32_t range, code, bc_bits, bc_val, state, ctx, rep[4], wnd_pos;
    uint64_t consumed;
    uint32_t need;      // 0, or DEC_NEED_RC / DEC_NEED_BC once a block ran out with no successor uploaded
    uint32_t err;       // DECODE_ERROR-class failure inside a packet
    uint32_t undo_n;
    uint32_t bv[2], bbase[2], btag[2];   // 64 bytes of the current RC / BC block, one per lane: block bytes [bbase, bbase+64) of block #btag
    // resume variables of the Decompress call in flight (mirrors of DecState fields)
    uint32_t phase, type, run_size, i, copied, copied_from, status, out_size, p_delta_ready;
};

__device__ static const uint32_t kDltIndexD[5] = {1, 2, 3, 4, 8};   // csc_typedef.h:36

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
}
// undo the probability updates made since the checkpoint and restore the scalars
DDEV void ck_rollback(Dc &c, const Ck &k)
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t n = c.undo_n; n > 0; n--) {
        uint32_t a = DUNI(c.undo_addr[n - 1]), v = DUNI(c.undo_val[n - 1]);
        if (a & 0x80000000u) c.p_lit[a & 0x7FFFFFFFu] = v;   // p_lit and p_delta are one allocation
        else c.L->P[a] = v;
    }
    c.undo_n = 0;
    for (int i = 0; i < 2; i++) { c.taken[i] = k.taken[i]; c.rd[i] = k.rd[i]; c.fill[i] = k.fill[i]; c.btag[i] = 0xFFFFFFFFu; }
    c.range = k.range; c.code = k.code; c.bc_bits = k.bc_bits; c.bc_val = k.bc_val;
    c.state = k.state; c.ctx = k.ctx; c.wnd_pos = k.wnd_pos; c.consumed = k.consumed;
    for (int i = 0; i < 4; i++) c.rep[i] = k.rep[i];
}

// one byte of the RC (kind 1) or BC (kind 0) stream; refills as soon as a block is exhausted, like
// DecodeBit / coder_decode_direct do (csc_dec.cpp:14-21, 70-76)
DDEV uint32_t next_byte(Dc &c, int kind)
{
    if (c.need) return 0;
    // bytes are served from a 64-byte register window over the current block (one HBM fetch per 64 bytes)
    uint32_t off = c.rd[kind] - c.bbase[kind];
    if (c.btag[kind] != c.taken[kind] || off >= 64) {
        uint32_t slot = (c.taken[kind] - 1) % c.qslots;
        c.bbase[kind] = c.rd[kind];
        c.btag[kind] = c.taken[kind];
        c.bv[kind] = c.q[kind][(size_t)slot * c.bsize + c.rd[kind] + c.lane];   // ring has 64 bytes of slack
        off = 0;
    }
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)c.bv[kind], (int)off);
    c.rd[kind]++;
    if (c.rd[kind] >= c.fill[kind]) {
        if (c.taken[kind] < c.avail[kind]) {
            c.consumed += c.rd[kind];
            uint32_t ns = c.taken[kind] % c.qslots;
            c.fill[kind] = DUNI(c.qsize[kind][ns]);
            c.taken[kind]++;
            c.rd[kind] = 0;
        } else {
            c.need = kind ? DEC_NEED_RC : DEC_NEED_BC;
        }
    }
    return b;
}

// DecodeBit (csc_dec.cpp:10-35): space 0 = small tables in LDS, 1 = p_lit / p_delta words in HBM
DDEV uint32_t dbit_p(Dc &c, uint32_t v, uint32_t space, uint32_t idx, uint32_t p)
{
    if (c.range < (1u << 24)) { c.range <<= 8; c.code = (c.code << 8) + next_byte(c, 1); }
    uint32_t bound = (c.range >> 12) * p, np, bit;
    if (c.code < bound) { c.range = bound; np = p + ((0xFFFu - p) >> 5); bit = 1; }
    else { c.range -= bound; c.code -= bound; np = p - (p >> 5); bit = 0; }
    if (!c.need) {
        if (c.undo_n < kDecUndoCap) {
            c.undo_addr[c.undo_n] = idx | (space << 31);
            c.undo_val[c.undo_n] = p;
            c.undo_n++;
        } else {
            c.err = 1;   // a packet longer than any the encoder can produce
        }
        if (space) c.p_lit[idx] = np; else c.L->P[idx] = np;
    }
    return v + v + bit;
}
DDEV uint32_t dbit(Dc &c, uint32_t v, uint32_t space, uint32_t idx)
{
    uint32_t p = space ? DUNI(c.p_lit[idx]) : DUNI(c.L->P[idx]);
    return dbit_p(c, v, space, idx, p);
}

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
// 8 bits under an order-1 row of p_lit / p_delta.  The node of bit k depends on the bits before it,
// so instead of 8 dependent HBM fetches the 15 nodes of the top 4 levels are fetched at once (heap
// order: lane L = node L), then the 15 nodes of the 4-level subtree under the node reached.
DDEV uint32_t dbyte_tree(Dc &c, uint32_t row_word)
{
    const uint32_t L = c.lane & 15;
    uint32_t pv = c.p_lit[row_word + L];
    uint32_t v = 1;
#pragma unroll
    for (int k = 0; k < 4; k++) v = dbit_p(c, v, 1, row_word + v, (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)v));
    // subtree under node v (16..31): lane L = 2^j + t  ->  node (v << j) + t
    const uint32_t j = 31u - (uint32_t)__builtin_clz(L | 1u);
    uint32_t pv2 = c.p_lit[row_word + (v << j) + (L - (1u << j))];
    uint32_t h = 1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pv2, (int)h);
        uint32_t nv = dbit_p(c, v, 1, row_word + v, p);
        h = h + h + (nv & 1);
        v = nv;
    }
    return v & 0xFF;
}
DDEV uint32_t dmatchlen_1(Dc &c)               // csc_dec.cpp:187-220
{
    uint32_t base, tab, i = 1;
    if (dbit(c, 0, 0, P_LEN_SLOT) == 0) { tab = P_LEN_X1; base = 0; }
    else if (dbit(c, 0, 0, P_LEN_SLOT + 1) == 0) { tab = P_LEN_X2; base = 8; }
    else { tab = P_LEN_X3; base = 16; }
    uint32_t top = base == 16 ? 0x80 : 0x08;
    do { i = dbit(c, i, 0, tab + i); } while (i < top);
    return base + (i & (top - 1));
}
DDEV uint32_t dmatchlen_2(Dc &c)               // csc_dec.cpp:222-234
{
    uint32_t len = dmatchlen_1(c);
    if (len != 143) return len;
    while (!c.need && !c.err && !dbit(c, 0, 0, P_LONGLEN)) len += 143;
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
    uint32_t i = 1;
    do { i = dbit(c, i, 0, P_DIST + pos + i); } while (i < (1u << sbits));
    uint32_t slot = i & ((1u << sbits) - 1);
    if (slot <= 2) dist = slot;
    else {
        uint32_t ebits = slot - 2, elen = ebits > 4 ? ddirect(c, ebits - 4) : 0;
        i = 1;
        do { i = dbit(c, i, 0, P_DIST_EXTRA + (ebits - 1) * 16 + i); } while (i < 0x10);
        dist = ((1u << ebits) + 1) + (elen << 4) + (__brev(i & 0x0Fu) >> 28);   // dist_table_[slot] + ... + rev16_table_
    }
    c.state = (c.state * 4 + 1) & 0x3F;
}

// window copy of a match: dst[j] = src[j] in increasing j, i.e. a pattern repeat when the regions
// overlap (csc_dec.cpp:513-518) -- lane-parallel with the modulo made explicit
DDEV void dcopy_match(Dc &c, uint32_t from, uint32_t dist, uint32_t len)
{
    dgu8 *w = c.wnd;
    const uint32_t to = c.wnd_pos;
    const bool overlap = from < to && from + len > to;   // then dist = to - from < len
    for (uint32_t j = c.lane; j < len; j += 64) w[to + j] = w[from + (overlap ? j % dist : j)];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// CSCDecoder::lz_decode body, csc_dec.cpp:476-571, one packet per loop turn, checkpointed per packet
DDEV void dlz_decode(Dc &c, uint32_t limit)
{
    uint32_t i = c.i, copied = c.copied, copied_from = c.copied_from;
    Ck k;
    for (; i <= limit;) {
        ck_take(c, k);
        uint32_t ni = i;
        bool end = false;
        uint32_t wr_kind = 0, wr_from = 0, wr_dist = 0, wr_len = 0, wr_byte = 0;   // deferred window write of this packet
        if (dbit(c, 0, 0, P_STATE + c.state * 3) == 0) {
            uint32_t b = dbyte_tree(c, c.ctx * 256);
            c.ctx = b;
            c.state = (c.state * 4) & 0x3F;
            wr_kind = 1; wr_byte = b; ni = i + 1;
        } else if (dbit(c, 0, 0, P_STATE + c.state * 3 + 1) == 1) {
            uint32_t dist, len;
            dmatch(c, dist, len);
            if (len == 0 && dist == 64) end = true;
            else {
                dist++; len += 2;
                c.rep[3] = c.rep[2]; c.rep[2] = c.rep[1]; c.rep[1] = c.rep[0]; c.rep[0] = dist;
                uint32_t from = c.wnd_pos >= dist ? c.wnd_pos - dist : c.wnd_pos + c.wnd_size - dist;
                if (from >= c.wnd_size || from + len > c.wnd_size || len + i > limit || c.wnd_pos + len > c.wnd_size) c.err = 1;
                wr_kind = 2; wr_from = from; wr_dist = dist; wr_len = len; ni = i + len;
            }
        } else if (dbit(c, 0, 0, P_STATE + c.state * 3 + 2) == 0) {
            c.state = (c.state * 4 + 2) & 0x3F;
            uint32_t from = c.wnd_pos > c.rep[0] ? c.wnd_pos - c.rep[0] : c.wnd_pos + c.wnd_size - c.rep[0];
            if (from > c.wnd_size) c.err = 1;   // the reference reads out of bounds here on corrupt input
            wr_kind = 2; wr_from = from; wr_dist = c.rep[0]; wr_len = 1; ni = i + 1;
        } else {
            uint32_t kk = 1;
            do { kk = dbit(c, kk, 0, P_REPDIST + c.state * 3 + kk - 1); } while (kk < 4);
            uint32_t idx = kk & 3, len = dmatchlen_2(c) + 2;
            c.state = (c.state * 4 + 3) & 0x3F;
            if (len + i > limit) c.err = 1;
            uint32_t dist = c.rep[idx];
            for (uint32_t j = idx; j > 0; j--) c.rep[j] = c.rep[j - 1];
            c.rep[0] = dist;
            uint32_t from = c.wnd_pos >= dist ? c.wnd_pos - dist : c.wnd_pos + c.wnd_size - dist;
            if (from >= c.wnd_size || from + len > c.wnd_size || len + i > limit || c.wnd_pos + len > c.wnd_size) c.err = 1;
            wr_kind = 2; wr_from = from; wr_dist = dist; wr_len = len; ni = i + len;
        }
        if (c.need) {          // ran out of input inside this packet: take it back, resume here later
            ck_rollback(c, k);
            c.i = i; c.copied = copied; c.copied_from = copied_from;
            return;
        }
        if (c.err) { c.status = DEC_ERR_DECODE; return; }
        if (end) break;
        if (wr_kind == 1) { c.wnd[c.wnd_pos] = (uint8_t)wr_byte; c.wnd_pos++; }
        else if (wr_kind == 2) {
            dcopy_match(c, wr_from, wr_dist, wr_len);
            c.wnd_pos += wr_len;
            c.ctx = DUNI((uint32_t)c.wnd[c.wnd_pos - 1]);
        }
        i = ni;
        if (c.wnd_pos > c.wnd_size) { c.status = DEC_ERR_DECODE; return; }
        if (c.wnd_pos == c.wnd_size) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            for (uint32_t t = c.lane; t < i - copied; t += 64) c.out[copied + t] = c.wnd[copied_from + t];
            c.wnd_pos = 0;
            copied_from = 0;
            copied = i;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t t = c.lane; t < i - copied; t += 64) c.out[copied + t] = c.wnd[copied_from + t];
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
        for (uint32_t t = c.lane; t < cur; t += 64) c.wnd[c.wnd_pos + t] = c.out[i + t];
        c.wnd_pos += cur;
        if (c.wnd_pos >= c.wnd_size) c.wnd_pos = 0;
        i += cur;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// ---- inverse filters on the decoded run in `out` ----
// Inverse_E89 (csc_filters.cpp:560-575,600-610), unrolled like the forward filter: opcode candidates by
// ballot, the skip chain over the sparse hits is serial, operands are rewritten in place.
DDEV void dinverse_e89(Dc &c, uint32_t size)
{
    dgu8 *b = c.out;
    uint32_t next_ok = 0;
    for (uint32_t base = 0; base + 5 < size; base += 64) {
        uint32_t j0 = base + c.lane;
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
DDEV void dinverse_delta(Dc &c, uint32_t size, uint32_t chn)
{
    if (size < 512) return;
    dgu8 *src = c.out, *copy = (dgu8 *)c.D->swap;
    for (uint32_t t = c.lane; t < size; t += 64) copy[t] = src[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    uint32_t carry = 0;
    for (uint32_t base = 0; base < size; base += 64) {
        uint32_t n = base + c.lane;
        uint32_t v = n < size ? copy[n] : 0;
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o); if ((int)c.lane >= o) v += t; }   // inclusive scan
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

// Inverse_Dict (csc_filters.cpp:337-369): expand word symbols; 254 escapes a byte >= 0x82.  64 source
// bytes are fetched per step (one per lane) and walked with readlane; the walk itself is serial
// because an escape swallows the byte after it.
DDEV void dinverse_dict(Dc &c, uint32_t size)
{
    dgu8 *src = c.out, *dst = (dgu8 *)c.D->swap;
    uint32_t i = 0, o = 0;
    while (o < size) {
        const uint32_t base = i;
        uint32_t v = src[base + c.lane];                 // `out` has 128 bytes of slack
        uint32_t vn = DUNI((uint32_t)src[base + 64]);
        uint32_t j = 0;
        while (j < 64 && o < size) {
            uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)j);
            if (b >= 0x82 && b < 0x82 + 122) {
                uint32_t w = DUNI(c.L->wtab[b - 0x82]);
                for (uint32_t t = 0; t < 4 && o < size; t++) {
                    uint32_t ch = (w >> (8 * t)) & 0xFF;
                    if (!ch) break;
                    dst[o++] = (uint8_t)ch;
                }
            } else {
                uint32_t nb = j + 1 < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(j + 1)) : vn;
                if (b == 254 && base + j + 1 < size && nb >= 0x82) { j++; dst[o++] = (uint8_t)nb; }
                else dst[o++] = (uint8_t)b;
            }
            j++;
        }
        i = base + j;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t t = c.lane; t < size; t += 64) src[t] = dst[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

// re-prime the arithmetic decoder from the NEXT RC and BC blocks (csc_dec.cpp:336-345, 657-680)
DDEV bool dprime(Dc &c)
{
    if (c.taken[1] >= c.avail[1]) { c.need = DEC_NEED_RC; return false; }
    if (c.taken[0] >= c.avail[0]) { c.need = DEC_NEED_BC; return false; }
    for (int kind = 1; kind >= 0; kind--) {
        uint32_t ns = c.taken[kind] % c.qslots;
        c.fill[kind] = DUNI(c.qsize[kind][ns]);
        c.taken[kind]++;
        c.rd[kind] = 0;
    }
    c.range = 0xFFFFFFFFu; c.bc_bits = c.bc_val = 0;
    const dgu8 *p = c.q[1] + (size_t)((c.taken[1] - 1) % c.qslots) * c.bsize;
    c.code = (DUNI((uint32_t)p[1]) << 24) | (DUNI((uint32_t)p[2]) << 16) | (DUNI((uint32_t)p[3]) << 8) | DUNI((uint32_t)p[4]);
    c.rd[1] = 5;
    return true;
}

__global__ __launch_bounds__(64) void k_decode_run(DecState *D)
{
    __shared__ DecLds lds;
    Dc c;
    c.D = D; c.L = &lds; c.lane = threadIdx.x;
    c.wnd = (dgu8 *)D->wnd; c.out = (dgu8 *)D->out; c.q[0] = (dgu8 *)D->q[0]; c.q[1] = (dgu8 *)D->q[1];
    c.p_lit = (dgu32 *)D->p_lit; c.p_delta = (dgu32 *)D->p_delta;
    c.qsize[0] = (dgu32 *)D->qsize[0]; c.qsize[1] = (dgu32 *)D->qsize[1];
    c.undo_addr = (dgu32 *)D->undo_addr; c.undo_val = (dgu32 *)D->undo_val;
    c.wnd_size = D->wnd_size; c.bsize = D->bsize; c.qslots = D->qslots;
    for (int i = 0; i < 2; i++) { c.avail[i] = D->avail[i]; c.taken[i] = D->taken[i]; c.rd[i] = D->rd[i]; c.fill[i] = D->fill[i]; }
    c.range = D->range; c.code = D->code; c.bc_bits = D->bc_bits; c.bc_val = D->bc_val;
    c.state = D->state; c.ctx = D->ctx; c.wnd_pos = D->wnd_pos; c.consumed = D->consumed;
    for (int i = 0; i < 4; i++) c.rep[i] = D->rep[i];
    c.need = 0; c.err = 0; c.undo_n = 0;
    c.bv[0] = c.bv[1] = 0; c.bbase[0] = c.bbase[1] = 0; c.btag[0] = c.btag[1] = 0xFFFFFFFFu;
    c.phase = D->phase; c.type = D->type; c.run_size = D->run_size; c.i = D->i; c.copied = D->copied; c.copied_from = D->copied_from;
    c.out_size = D->out_size; c.p_delta_ready = D->p_delta_ready; c.status = DEC_RUNNING;
    for (uint32_t i = c.lane; i < P_COUNT; i += 64) lds.P[i] = D->probs[i];
    for (uint32_t i = c.lane; i < 122; i += 64) {
        const dgu8 *w = (const dgu8 *)D->words + i * 8;
        lds.wtab[i] = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) | ((uint32_t)w[3] << 24);
    }
    const uint32_t max = D->raw_blocksize;
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
                for (uint32_t t = c.lane; t < 256 * 256; t += 64) c.p_delta[t] = 2048;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                c.p_delta_ready = 1;
            }
            while (i < size) {
                ck_take(c, k);
                uint32_t ni = i + 1, byte = 0, runlen = 0;
                if (type == DT_BAD) byte = ddirect16(c, 8);
                else if (type == DT_ENTROPY) { byte = dbyte_tree(c, c.ctx * 256); c.ctx = byte; }
                else if (dbit(c, 0, 0, P_RLE_FLAG) == 0) byte = dbyte_tree(c, 65536 + sctx * 256);   // p_delta follows p_lit
                else { runlen = dmatchlen_2(c) + 11; if (i == 0) c.err = 2; }
                if (c.need) { ck_rollback(c, k); break; }
                if (c.err) { c.status = c.err == 2 ? DEC_ERR_MINUS1 : DEC_ERR_DECODE; break; }
                if (runlen) {
                    uint32_t n = runlen < size - i ? runlen : size - i;
                    uint32_t prev = DUNI((uint32_t)c.out[i - 1]);
                    for (uint32_t t = c.lane; t < n; t += 64) c.out[i + t] = (uint8_t)prev;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
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
            if (type == DT_EXE) dinverse_e89(c, size);
            else if (type == DT_ENGTXT) dinverse_dict(c, size);
            else if (type >= DT_DLT) { dinverse_delta(c, size, kDltIndexD[type - DT_DLT]); dcopy2dict(c, size); }
            else if (type == DT_BAD || type == DT_ENTROPY) dcopy2dict(c, size);
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
    for (uint32_t i = c.lane; i < P_COUNT; i += 64) D->probs[i] = lds.P[i];
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

__global__ void k_decode_init(DecState *D)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (uint32_t i = tid; i < 256 * 256; i += nth) D->p_lit[i] = 2048;
    for (uint32_t i = tid; i < P_COUNT; i += nth) D->probs[i] = 2048;
}

void launch_decode_init(DecState *D, hipStream_t st) { hipLaunchKernelGGL(k_decode_init, dim3(64), dim3(256), 0, st, D); }
void launch_decode_run(DecState *D, hipStream_t st) { hipLaunchKernelGGL(k_decode_run, dim3(1), dim3(64), 0, st, D); }

}  // namespace cscmi
