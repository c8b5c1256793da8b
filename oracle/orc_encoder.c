/*
 * oracle/orc_encoder.c -- CPU restatement of the libcsc ENCODE path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  This file is the checker the HIP path is
 * compared against.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (csc_amd/) never links or calls it.
 *
 * It is a from-scratch plain-C restatement (explicit state, no classes) of the
 * algorithm in /root/reference/src/libcsc; each function cites the reference
 * file:line it follows.  Parity is PINNED: oracle/_ref (the reference compiled
 * from its own sources by oracle/Makefile) is compared byte-for-byte with this
 * restatement by tools/make_golden.py and tests/test_oracle_vs_ref.py, and the
 * resulting vectors are committed under tests/golden/.
 */
#include <math.h>
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

#include "orc_api.h"

#define KB 1024u
#define MB 1048576u
#define MIN_BLOCK (8u * KB)       /* csc_typedef.h:9 MinBlockSize */
#define UMIN(a, b) ((a) < (b) ? (a) : (b))

/* block types, csc_typedef.h:20-40 */
enum {
    DT_NORMAL = 1, DT_ENGTXT = 2, DT_EXE = 3, DT_FAST = 4, DT_NO_LZ = 5,
    DT_ENTROPY = 7, DT_BAD = 8, SIG_EOF = 9, DT_DLT = 0x10, DT_SKIP = 0x1E
};
static const uint32_t kDltIndex[5] = {1, 2, 3, 4, 8};

#define HT2_SIZE (16u * KB)       /* csc_mf.h:18 */
#define HT3_SIZE (64u * KB)       /* csc_mf.h:17 */
#define MF_CAND_LIMIT 32          /* csc_mf.h:34 */
#define AP_LIMIT 2048             /* csc_lz.h:43 */

typedef struct { uint32_t len; uint32_t dist; } MFUnit; /* len doubles as price, csc_mf.h:8-14 */

typedef struct {
    uint32_t dist, state;
    int back_pos, next_pos;
    uint32_t price, lit;
    uint32_t rep_dist[4];
} APUnit; /* csc_lz.h:33-41 */

typedef struct {
    uint32_t next[26];
    uint8_t symbol;
} TrieNode; /* csc_filters.h:30-33 */

typedef struct OrcEnc {
    ISzAlloc *alloc;
    ISeqOutStream *os;
    CSCProps props;
    jmp_buf on_error;             /* replaces `throw (int)` of csc_coder.h:11 */

    /* ---- MemIO + Coder, csc_coder.h:15-64 ---- */
    uint32_t bsize;
    uint8_t *rc_buf, *bc_buf;
    uint32_t rc_size, bc_size;
    uint64_t rc_low, rc_cachesize;
    uint32_t rc_range;
    uint8_t rc_cache;
    uint32_t bc_curbits, bc_curval;
    int64_t outsize;

    /* ---- Model, csc_model.h:58-122 ---- */
    uint32_t p_state[64 * 3];
    uint32_t state, ctx;
    uint32_t p_rle_flag;
    uint32_t *p_lit, *p_delta;
    uint32_t p_repdist[64 * 4];
    uint32_t p_dist[8 + 16 * 2 + 32 * 4];
    uint32_t p_longlen;
    uint32_t p_2_bits[512];
    uint32_t p_len_slot[2], p_len_x1[8], p_len_x2[8], p_len_x3[128];
    uint32_t p_dist_extra[29 * 16];
    uint32_t len_price[32];
    uint32_t lp_rebuild_int;

    /* ---- MatchFinder, csc_mf.h:16-53 ---- */
    uint8_t *wnd;
    uint32_t wnd_size, vld_rge;
    uint32_t *mfbuf, *ht2, *ht3, *ht6, *bt_head, *bt_nodes;
    uint64_t mf_size;
    uint32_t ht_bits, ht_width, ht_low;
    uint32_t bt_bits, bt_size, bt_pos;
    uint32_t ht_cyc, bt_cyc, good_len;
    uint32_t pos;
    MFUnit mfcand[MF_CAND_LIMIT];

    /* ---- LZ, csc_lz.h:19-56 ---- */
    uint32_t wnd_curpos;
    uint32_t rep_dist[4];
    uint32_t lz_good_len, lz_bt_cyc, lz_ht_cyc;
    MFUnit *appt;
    APUnit *ap;

    /* ---- Analyzer, csc_analyzer.h:19 ---- */
    uint32_t log_table[(MIN_BLOCK >> 4) + 1];

    /* ---- Filters, csc_filters.h:26-62 ---- */
    TrieNode trie[300];
    uint8_t *swap_buf;
    uint32_t swap_size;
    uint32_t x0, x1, ei, ek;
    uint8_t ecs;
} OrcEnc;

/* ===================================================================== */
/* default allocator, csc_default_alloc.cpp:5-17                          */
static void *def_alloc(void *p, size_t n) { (void)p; return malloc(n); }
static void def_free(void *p, void *a) { (void)p; free(a); }
static ISzAlloc g_default_alloc = {def_alloc, def_free};

/* ===================================================================== */
/* MemIO::WriteBlock, csc_memio.cpp:83-108                               */
static int write_block(OrcEnc *e, uint8_t *buf, uint32_t size, int rc1bc0)
{
    uint8_t fb = (uint8_t)(rc1bc0 << 7);
    if (size == e->bsize) fb |= (1 << 6);
    if (e->os->Write(e->os, &fb, 1) != 1) return -1;
    if (size != e->bsize) {
        uint8_t sb[3];
        sb[0] = (size >> 16) & 0xff; sb[1] = (size >> 8) & 0xff; sb[2] = size & 0xff;
        if (e->os->Write(e->os, sb, 3) != 3) return -1;
    }
    if (size && e->os->Write(e->os, buf, size) != size) return -1;
    return (int)size;
}

/* ===================================================================== */
/* Coder, csc_coder.cpp                                                  */
static void coder_reset_state(OrcEnc *e) /* csc_coder.cpp:9-16,65-73 */
{
    e->rc_low = 0; e->rc_range = 0xFFFFFFFFu; e->rc_cachesize = 1; e->rc_cache = 0;
    e->rc_size = e->bc_size = 0; e->bc_curbits = e->bc_curval = 0;
}

/* Coder::RC_ShiftLow, csc_coder.cpp:89-112 */
static void rc_shift_low(OrcEnc *e)
{
    if ((uint32_t)e->rc_low < 0xFF000000u || (int32_t)(e->rc_low >> 32) != 0) {
        uint8_t temp = e->rc_cache;
        do {
            e->rc_buf[e->rc_size++] = (uint8_t)(temp + (uint8_t)(e->rc_low >> 32));
            if (e->rc_size == e->bsize) {
                e->outsize += e->rc_size;
                if (write_block(e, e->rc_buf, e->bsize, 1) != (int)e->bsize)
                    longjmp(e->on_error, -WRITE_ERROR);
                e->rc_size = 0;
            }
            temp = 0xFF;
        } while (--e->rc_cachesize != 0);
        e->rc_cache = (uint8_t)((uint32_t)e->rc_low >> 24);
    }
    e->rc_cachesize++;
    e->rc_low = (uint64_t)((uint32_t)e->rc_low << 8);
}

/* EncodeBit macro, csc_coder.h:67-81 */
static inline void enc_bit(OrcEnc *e, uint32_t v, uint32_t *p)
{
    uint32_t bound = (e->rc_range >> 12) * *p;
    if (v) {
        e->rc_range = bound;
        *p += (0xFFF - *p) >> 5;
    } else {
        e->rc_low += bound;
        e->rc_range -= bound;
        *p -= *p >> 5;
    }
    if (e->rc_range < (1u << 24)) {
        e->rc_range <<= 8;
        rc_shift_low(e);
    }
}

/* BCWCheckBound, csc_coder.h:7-16 */
static void bc_check_bound(OrcEnc *e)
{
    if (e->bc_size == e->bsize) {
        e->outsize += e->bc_size;
        if (write_block(e, e->bc_buf, e->bsize, 0) != (int)e->bsize)
            longjmp(e->on_error, -WRITE_ERROR);
        e->bc_size = 0;
    }
}

/* Coder::EncDirect16, csc_coder.cpp:76-87 */
static void enc_direct16(OrcEnc *e, uint32_t val, uint32_t len)
{
    e->bc_curval = (e->bc_curval << len) | val;
    e->bc_curbits += len;
    while (e->bc_curbits >= 8) {
        e->bc_buf[e->bc_size++] = (e->bc_curval >> (e->bc_curbits - 8)) & 0xFF;
        bc_check_bound(e);
        e->bc_curbits -= 8;
    }
}

/* EncodeDirect macro, csc_coder.h:83-88 */
static void enc_direct(OrcEnc *e, uint32_t v, uint32_t l)
{
    if (l <= 16) enc_direct16(e, v, l);
    else { enc_direct16(e, v >> 16, l - 16); enc_direct16(e, v & 0xFFFF, 16); }
}

/* Coder::Flush, csc_coder.cpp:40-74.  The byte at rc_buf[rc_size] is NOT
 * stored (stale content of the persistent buffer). */
static void coder_flush(OrcEnc *e)
{
    for (int i = 0; i < 5; i++) rc_shift_low(e);
    e->rc_size++;
    for (int i = 0; i < 2; i++) {
        if (i == 1) e->bc_buf[e->bc_size++] = 0;
        else e->bc_buf[e->bc_size++] = (e->bc_curval << (8 - e->bc_curbits)) & 0xFF;
        bc_check_bound(e);
    }
    e->outsize += e->rc_size + e->bc_size;
    if (write_block(e, e->rc_buf, e->rc_size, 1) != (int)e->rc_size
        || write_block(e, e->bc_buf, e->bc_size, 0) != (int)e->bc_size)
        longjmp(e->on_error, -WRITE_ERROR);
    coder_reset_state(e);
}

/* ===================================================================== */
/* Model, csc_model.cpp                                                  */
static const uint32_t kDistTable[33] = { /* csc_model.cpp:45-55 */
    0, 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 2049, 4097, 8193,
    16385, 32769, 65537, 131073, 262145, 524289, 1048577, 2097153,
    4194305, 8388609, 16777217, 33554433, 67108865, 134217729, 268435457,
    536870913, 1073741825,
};
static const uint32_t kRev16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};

static void fill_p2bits(uint32_t t[512]) /* csc_model.cpp:68-70: log(float) is the float overload in C++ */
{
    for (int i = 0; i < 512; i++)
        t[i] = (uint32_t)(128 * logf((float)(i * 8 + 4) / 4096) / log(0.5));
}

static void fill_probs(uint32_t *p, int n) { for (int i = 0; i < n; i++) p[i] = 2048; }

static void model_reset(OrcEnc *e) /* csc_model.cpp:88-111 */
{
    e->alloc->Free(e->alloc, e->p_delta);
    e->p_delta = NULL;
    fill_probs(e->p_state, 64 * 3);
    fill_probs(e->p_lit, 256 * 256);
    fill_probs(e->p_repdist, 64 * 3);
    fill_probs(e->p_dist, 8 + 16 * 2 + 32 * 4);
    fill_probs(e->p_len_slot, 2);
    fill_probs(e->p_len_x1, 8);
    fill_probs(e->p_len_x2, 8);
    fill_probs(e->p_len_x3, 128);
    fill_probs(e->p_dist_extra, 29 * 16);
    e->p_longlen = 2048; e->p_rle_flag = 2048;
    e->state = 0; e->ctx = 0; e->lp_rebuild_int = 0;
}

/* price of coding bit v under probability p, FEncodeBit csc_model.cpp:161-167 */
static inline uint32_t bit_price(const OrcEnc *e, uint32_t v, uint32_t p)
{
    return v ? e->p_2_bits[p >> 3] : e->p_2_bits[(4096 - p) >> 3];
}

/* Model::encode_matchlen_1, csc_model.cpp:113-145 */
static void encode_matchlen_1(OrcEnc *e, uint32_t len)
{
    uint32_t *p, c;
    if (len < 16) {
        if (len < 8) {
            enc_bit(e, 0, &e->p_len_slot[0]);
            p = e->p_len_x1;
        } else {
            enc_bit(e, 1, &e->p_len_slot[0]);
            enc_bit(e, 0, &e->p_len_slot[1]);
            len -= 8;
            p = e->p_len_x2;
        }
        c = len | 0x08;
        do { enc_bit(e, (c >> 2) & 1, &p[c >> 3]); c <<= 1; } while (c < 0x40);
    } else {
        enc_bit(e, 1, &e->p_len_slot[0]);
        enc_bit(e, 1, &e->p_len_slot[1]);
        len -= 16;
        p = e->p_len_x3;
        c = len | 0x80;
        do { enc_bit(e, (c >> 6) & 1, &p[c >> 7]); c <<= 1; } while (c < 0x4000);
    }
}

/* Model::encode_matchlen_2, csc_model.cpp:147-159 */
static void encode_matchlen_2(OrcEnc *e, uint32_t len)
{
    if (len >= 143) {
        encode_matchlen_1(e, 143);
        len -= 143;
        while (len >= 143) { len -= 143; enc_bit(e, 0, &e->p_longlen); }
        enc_bit(e, 1, &e->p_longlen);
    }
    encode_matchlen_1(e, len);
}

/* literal bits under an order-1 context row, shared by csc_model.cpp:176-183,431-441,452-459,504-509 */
static void encode_byte_tree(OrcEnc *e, uint32_t *row, uint32_t c)
{
    c |= 0x100;
    do { enc_bit(e, (c >> 7) & 1, &row[c >> 8]); c <<= 1; } while (c < 0x10000);
}

/* Model::EncodeLiteral, csc_model.cpp:169-183 */
static void encode_literal(OrcEnc *e, uint32_t c)
{
    enc_bit(e, 0, &e->p_state[e->state * 3 + 0]);
    e->state = (e->state * 4) & 0x3F;
    uint32_t *row = &e->p_lit[e->ctx * 256];
    e->ctx = c;
    encode_byte_tree(e, row, c);
}

/* Model::GetLiteralPrice, csc_model.cpp:185-196 */
static uint32_t literal_price(const OrcEnc *e, uint32_t fstate, uint32_t fctx, uint32_t c)
{
    uint32_t ret = bit_price(e, 0, e->p_state[fstate * 3 + 0]);
    const uint32_t *row = &e->p_lit[fctx * 256];
    c |= 0x100;
    do { ret += bit_price(e, (c >> 7) & 1, row[c >> 8]); c <<= 1; } while (c < 0x10000);
    return ret;
}

/* Model::EncodeRep0Len1, csc_model.cpp:198-207 */
static void encode_rep0len1(OrcEnc *e)
{
    enc_bit(e, 1, &e->p_state[e->state * 3 + 0]);
    enc_bit(e, 0, &e->p_state[e->state * 3 + 1]);
    enc_bit(e, 0, &e->p_state[e->state * 3 + 2]);
    e->ctx = 0;
    e->state = (e->state * 4 + 2) & 0x3F;
}

/* Model::GetRep0Len1Price, csc_model.cpp:209-216 */
static uint32_t rep0len1_price(const OrcEnc *e, uint32_t fs)
{
    return bit_price(e, 1, e->p_state[fs * 3 + 0]) + bit_price(e, 0, e->p_state[fs * 3 + 1])
         + bit_price(e, 0, e->p_state[fs * 3 + 2]);
}

/* Model::EncodeRepDistMatch, csc_model.cpp:218-232 */
static void encode_rep_match(OrcEnc *e, uint32_t rep_idx, uint32_t match_len)
{
    enc_bit(e, 1, &e->p_state[e->state * 3 + 0]);
    enc_bit(e, 0, &e->p_state[e->state * 3 + 1]);
    enc_bit(e, 1, &e->p_state[e->state * 3 + 2]);
    uint32_t i = 1, j;
    j = (rep_idx >> 1) & 1; enc_bit(e, j, &e->p_repdist[e->state * 3 + i - 1]); i += i + j;
    j = rep_idx & 1;        enc_bit(e, j, &e->p_repdist[e->state * 3 + i - 1]);
    encode_matchlen_2(e, match_len);
    e->state = (e->state * 4 + 3) & 0x3F;
}

/* Model::len_price_rebuild, csc_model.cpp:234-270 */
static void len_price_rebuild(OrcEnc *e)
{
    for (int i = 0; i < 32; i++) {
        uint32_t ret = 0, len = (uint32_t)i, c;
        const uint32_t *p;
        if (len < 16) {
            if (len < 8) {
                ret += bit_price(e, 0, e->p_len_slot[0]);
                p = e->p_len_x1;
            } else {
                ret += bit_price(e, 1, e->p_len_slot[0]);
                ret += bit_price(e, 0, e->p_len_slot[1]);
                len -= 8;
                p = e->p_len_x2;
            }
            c = len | 0x08;
            do { ret += bit_price(e, (c >> 2) & 1, p[c >> 3]); c <<= 1; } while (c < 0x40);
        } else {
            ret += bit_price(e, 1, e->p_len_slot[0]);
            ret += bit_price(e, 1, e->p_len_slot[1]);
            len -= 16;
            p = e->p_len_x3;
            c = len | 0x80;
            do { ret += bit_price(e, (c >> 6) & 1, p[c >> 7]); c <<= 1; } while (c < 0x4000);
        }
        e->len_price[i] = ret;
    }
    e->lp_rebuild_int = 4096;
}

/* Model::GetRepDistPrice, csc_model.cpp:273-284 */
static uint32_t rep_dist_price(const OrcEnc *e, uint32_t fs, uint32_t rep_idx)
{
    uint32_t ret = bit_price(e, 1, e->p_state[fs * 3 + 0]) + bit_price(e, 0, e->p_state[fs * 3 + 1])
                 + bit_price(e, 1, e->p_state[fs * 3 + 2]);
    uint32_t i = 1, j;
    j = (rep_idx >> 1) & 1; ret += bit_price(e, j, e->p_repdist[fs * 3 + i - 1]); i += i + j;
    j = rep_idx & 1;        ret += bit_price(e, j, e->p_repdist[fs * 3 + i - 1]);
    return ret;
}

/* Model::GetMatchLenPrice, csc_model.cpp:286-299 */
static uint32_t match_len_price(OrcEnc *e, uint32_t match_len)
{
    if (match_len >= 32) return 128 * 6;
    if (e->lp_rebuild_int-- == 0) len_price_rebuild(e);
    return e->len_price[match_len];
}

/* slot search shared by EncodeMatch / GetMatchDistPrice, csc_model.cpp:328-337,375-384 */
static uint32_t dist_slot(uint32_t dist)
{
    uint32_t l = 0, r = 32;
    while (l + 1 < r) {
        uint32_t mid = l + (r - l) / 2;
        if (kDistTable[mid] > dist) r = mid;
        else if (kDistTable[mid] < dist) l = mid;
        else l = r = mid;
    }
    return l;
}

/* Model::EncodeMatch, csc_model.cpp:301-366 */
static void encode_match(OrcEnc *e, uint32_t dist, uint32_t len)
{
    enc_bit(e, 1, &e->p_state[e->state * 3 + 0]);
    enc_bit(e, 1, &e->p_state[e->state * 3 + 1]);
    encode_matchlen_2(e, len);
    uint32_t pdist_pos, sbits;
    switch (len) {
    case 0: pdist_pos = 0; sbits = 3; break;
    case 1: case 2: pdist_pos = 16 * (len - 1) + 8; sbits = 4; break;
    case 3: case 4: case 5: pdist_pos = 32 * (len - 3) + 8 + 16 * 2; sbits = 5; break;
    default: pdist_pos = 32 * 3 + 8 + 16 * 2; sbits = 5; break;
    }
    uint32_t slot = dist_slot(dist), c = slot | (1u << sbits);
    uint32_t extra_bits = slot > 2 ? slot - 2 : 0;
    uint32_t *p = e->p_dist + pdist_pos;
    do { enc_bit(e, (c >> (sbits - 1)) & 1, &p[c >> sbits]); c <<= 1; } while (c < (1u << (sbits * 2)));
    if (extra_bits) {
        uint32_t extra_len = dist - (1u << extra_bits) - 1;
        if (extra_bits > 4) enc_direct(e, extra_len >> 4, extra_bits - 4);
        c = kRev16[extra_len & 0x0F] | 0x10;
        p = &e->p_dist_extra[(extra_bits - 1) * 16];
        do { enc_bit(e, (c >> 3) & 1, &p[c >> 4]); c <<= 1; } while (c < (1u << 8));
    }
    e->state = (e->state * 4 + 1) & 0x3F;
}

/* Model::GetMatchDistPrice, csc_model.cpp:368-387 */
static uint32_t match_dist_price(const OrcEnc *e, uint32_t fs, uint32_t dist)
{
    uint32_t ret = bit_price(e, 1, e->p_state[fs * 3 + 0]) + bit_price(e, 1, e->p_state[fs * 3 + 1]);
    uint32_t l = dist_slot(dist);
    return ret + (l > 2 ? l + 2 : 2) * 128;
}

/* Model::EncodeInt, csc_model.cpp:389-414 */
static void encode_int(OrcEnc *e, uint32_t num)
{
    uint32_t tmp = num, slot = 0;
    while (tmp) { tmp >>= 1; slot++; }
    if (slot) slot--;
    enc_direct(e, slot, 5);
    if (slot == 0) enc_direct(e, num, 1);
    else enc_direct(e, num - (1u << slot), slot);
}

/* Model::CompressLiterals, csc_model.cpp:448-461 */
static void compress_literals(OrcEnc *e, const uint8_t *src, uint32_t size)
{
    encode_int(e, size);
    for (uint32_t i = 0; i < size; i++) {
        uint32_t c = src[i];
        uint32_t *row = &e->p_lit[e->ctx * 256];
        e->ctx = c;
        encode_byte_tree(e, row, c);
    }
}

/* Model::CompressBad, csc_model.cpp:463-469 */
static void compress_bad(OrcEnc *e, const uint8_t *src, uint32_t size)
{
    encode_int(e, size);
    for (uint32_t i = 0; i < size; i++) enc_direct16(e, src[i], 8);
}

/* Model::CompressRLE, csc_model.cpp:471-513 */
static void compress_rle(OrcEnc *e, const uint8_t *src, uint32_t size)
{
    uint32_t i, j, len, sctx = 0;
    encode_int(e, size);
    if (e->p_delta == NULL) {
        e->p_delta = (uint32_t *)e->alloc->Alloc(e->alloc, 256 * 256 * 4);
        fill_probs(e->p_delta, 256 * 256);
    }
    for (i = 0; i < size;) {
        if (i > 0 && size - i > 3 && src[i - 1] == src[i] && src[i] == src[i + 1] && src[i] == src[i + 2]) {
            j = i + 3; len = 3;
            while (j < size && src[j] == src[j - 1]) { len++; j++; }
            if (len > 10) {
                sctx = src[j - 1];
                len -= 11;
                enc_bit(e, 1, &e->p_rle_flag);
                encode_matchlen_2(e, len);
                i = j;
                continue;
            }
        }
        enc_bit(e, 0, &e->p_rle_flag);
        encode_byte_tree(e, &e->p_delta[sctx * 256], src[i]);
        sctx = src[i];
        i++;
    }
}

/* ===================================================================== */
/* MatchFinder, csc_mf.cpp                                               */
static inline uint32_t hash2(const uint8_t *p) /* csc_mf.cpp:23-28 */
{
    uint16_t v; memcpy(&v, p, 2);
    return (v * 65521u) & 0x3FFF;
}
static inline uint32_t hash3(const uint8_t *p) /* csc_mf.cpp:30-33 */
{
    return ((uint32_t)p[0] << 8) ^ ((uint32_t)p[1] << 5) ^ p[2];
}
static inline uint32_t hash6(const uint8_t *p, uint32_t bits) /* csc_mf.cpp:35-42 */
{
    uint32_t v; uint16_t v2;
    memcpy(&v, p, 4); memcpy(&v2, p + 4, 2);
    return ((v ^ ((uint32_t)v2 << 13)) * 2654435761u) >> (32 - bits);
}

/* MatchFinder::Init, csc_mf.cpp:45-106.  The table block gets 32 spare words:
 * the reference indexes bt_nodes_[bt_size_*2 .. +1] after a SlidePos that left
 * bt_pos_ == bt_size_ (csc_mf.cpp:201 vs :405); its own slack (:69) absorbs it. */
static int mf_init(OrcEnc *e, uint32_t bt_size, uint32_t bt_bits, uint32_t ht_width, uint32_t ht_bits)
{
    e->vld_rge = e->wnd_size - MIN_BLOCK - 4;
    e->pos = e->vld_rge;
    e->bt_pos = 0;
    e->ht_bits = ht_bits; e->ht_width = ht_width; e->bt_bits = bt_bits; e->bt_size = bt_size;
    if (!e->bt_bits || !e->bt_size) e->bt_bits = e->bt_size = 0;
    if (!e->ht_bits || !e->ht_width) e->ht_bits = e->ht_width = 0;
    e->mf_size = (uint64_t)HT2_SIZE + HT3_SIZE + ((uint64_t)1 << e->ht_bits) * e->ht_width;
    if (e->bt_bits) e->mf_size += ((uint64_t)1 << e->bt_bits) + (uint64_t)e->bt_size * 2;
    e->mfbuf = (uint32_t *)e->alloc->Alloc(e->alloc, sizeof(uint32_t) * (e->mf_size + 32));
    if (!e->mfbuf) return -1;
    memset(e->mfbuf, 0, sizeof(uint32_t) * (e->mf_size + 32));
    uint64_t cpos = 0;
    e->ht2 = e->mfbuf + cpos; cpos += HT2_SIZE;
    e->ht3 = e->mfbuf + cpos; cpos += HT3_SIZE;
    if (e->ht_bits && e->ht_width) { e->ht6 = e->mfbuf + cpos; cpos += (uint64_t)e->ht_width << e->ht_bits; }
    else e->ht6 = NULL;
    if (e->bt_bits) {
        e->bt_head = e->mfbuf + cpos; cpos += (uint64_t)1 << e->bt_bits;
        e->bt_nodes = e->mfbuf + cpos;
    } else e->bt_head = NULL;
    return 0;
}

/* MatchFinder::normalize, csc_mf.cpp:108-114 */
static void mf_normalize(OrcEnc *e)
{
    uint32_t diff = e->pos - e->vld_rge + 1;
    for (uint64_t i = 0; i < e->mf_size; i++)
        e->mfbuf[i] = e->mfbuf[i] > diff ? e->mfbuf[i] - diff : 0;
    e->pos -= diff;
}

/* MatchFinder::SetArg, csc_mf.cpp:121-127 */
static void mf_set_arg(OrcEnc *e, uint32_t bt_cyc, uint32_t ht_cyc, uint32_t ht_low, uint32_t good_len)
{
    e->bt_cyc = bt_cyc; e->ht_cyc = ht_cyc; e->ht_low = ht_low; e->good_len = good_len;
}

static inline uint32_t wrap_back(const OrcEnc *e, uint32_t wpos, uint32_t dist)
{
    return wpos >= dist ? wpos - dist : wpos + e->wnd_size - dist;
}

/* MatchFinder::SlidePos, csc_mf.cpp:134-206 */
static void mf_slide_pos(OrcEnc *e, uint32_t wnd_pos, uint32_t len, uint32_t limit)
{
    uint32_t h6, lasth6 = 0;
    for (uint32_t i = 1; i < len;) {
        uint32_t wpos = wnd_pos + i;
        if (e->pos >= 0xFFFFFFF0u) mf_normalize(e);
        e->ht2[hash2(e->wnd + wpos)] = e->pos;
        e->ht3[hash3(e->wnd + wpos)] = e->pos;

        if (i + 128 < len) { i += 4; e->pos += 4; e->bt_pos += 4; continue; }

        if (e->ht_width) {
            h6 = hash6(e->wnd + wpos, e->ht_bits);
            uint32_t *b = e->ht6 + (size_t)h6 * e->ht_width;
            if (h6 != lasth6) {
                uint32_t cands = UMIN(e->ht_width, e->ht_cyc);
                for (uint32_t j = cands - 1; j > 0; j--) b[j] = b[j - 1];
            }
            b[0] = e->pos;
            lasth6 = h6;
        }

        if (!e->bt_head) { e->pos++; i++; continue; }
        uint32_t hbt = hash6(e->wnd + wpos, e->bt_bits);
        if (e->bt_pos >= e->bt_size) e->bt_pos -= e->bt_size;
        uint32_t dist = e->pos - e->bt_head[hbt];
        uint32_t *l = &e->bt_nodes[(size_t)e->bt_pos * 2], *r = &e->bt_nodes[(size_t)e->bt_pos * 2 + 1];
        uint32_t lenl = 0, lenr = 0;
        for (uint32_t cyc = 0;; cyc++) {
            if (cyc >= e->bt_cyc || dist >= e->bt_size || dist >= e->vld_rge) { *l = *r = 0; break; }
            uint32_t cmp_pos = wrap_back(e, wpos, dist);
            uint32_t clen = UMIN(lenl, lenr);
            uint32_t climit = UMIN(limit - i, e->wnd_size - cmp_pos);
            if (clen >= climit) { *l = *r = 0; break; }
            uint32_t bt_npos = e->bt_pos >= dist ? e->bt_pos - dist : e->bt_pos + e->bt_size - dist;
            uint32_t *tlast = &e->bt_nodes[(size_t)bt_npos * 2];
            const uint8_t *pcur = e->wnd + wpos, *pmatch = e->wnd + cmp_pos;
            if (pcur[clen] == pmatch[clen]) {
                uint32_t climit2 = UMIN(e->good_len, climit);
                clen++;
                while (clen < climit2 && pcur[clen] == pmatch[clen]) clen++;
                if (clen >= e->good_len) { *l = tlast[0]; *r = tlast[1]; break; }
                else if (clen >= climit2) { *l = *r = 0; break; }
            }
            if (pmatch[clen] < pcur[clen]) {
                *l = e->pos - dist;
                dist = e->pos - *(l = &tlast[1]);
                lenl = clen;
            } else {
                *r = e->pos - dist;
                dist = e->pos - *(r = &tlast[0]);
                lenr = clen;
            }
        }
        e->bt_head[hbt] = e->pos;
        e->bt_pos++;
        e->pos++;
        i++;
    }
}

/* MatchFinder::SlidePosFast, csc_mf.cpp:208-241 */
static void mf_slide_pos_fast(OrcEnc *e, uint32_t wnd_pos, uint32_t len)
{
    for (uint32_t i = 0; i < len;) {
        uint32_t wpos = wnd_pos + i;
        if (e->pos >= 0xFFFFFFF0u) mf_normalize(e);
        uint32_t h = hash2(e->wnd + wpos);
        if (h % 16) {
            i++; e->pos++;
            if (++e->bt_pos >= e->bt_size) e->bt_pos -= e->bt_size;
            continue;
        }
        if (e->ht_width) {
            h = hash6(e->wnd + wpos, e->ht_bits);
            uint32_t *b = e->ht6 + (size_t)h * e->ht_width;
            for (uint32_t k = e->ht_width - 1; k > 0; k--) b[k] = b[k - 1];
            b[0] = e->pos;
        }
        if (e->bt_head) {
            h = hash6(e->wnd + wpos, e->bt_bits);
            e->bt_nodes[(size_t)e->bt_pos * 2] = e->bt_nodes[(size_t)e->bt_pos * 2 + 1] = 0;
            e->bt_head[h] = e->pos;
            if (++e->bt_pos >= e->bt_size) e->bt_pos -= e->bt_size;
        }
        i++; e->pos++;
    }
}

static const uint32_t kBound[7] = {0, 0, 64, 1024, 16 * KB, 256 * KB, 4 * MB}; /* csc_mf.cpp:245 */

/* common prefix length of the current position and a window candidate, capped at climit */
static inline uint32_t prefix_len(const uint8_t *pcur, const uint8_t *pmatch, uint32_t climit)
{
    uint32_t n = 0;
    while (n < climit && pcur[n] == pmatch[n]) n++;
    return n;
}

/* MatchFinder::find_match, csc_mf.cpp:243-495 */
static uint32_t mf_find_match(OrcEnc *e, MFUnit *ret, const uint32_t *rep_dist, uint32_t wpos, uint32_t limit)
{
    const uint8_t *pcur = e->wnd + wpos;
    uint32_t h2 = hash2(pcur), h3 = hash3(pcur), h6 = 0, hbt = 0;
    uint32_t minlen = 1, cnt = 0, dist = 0;
    if (e->ht_width) h6 = hash6(pcur, e->ht_bits);
    if (e->bt_head) hbt = hash6(pcur, e->bt_bits);

#define PUSH_CAND(L, D) do { ret[cnt].len = (L); ret[cnt].dist = (D); if (cnt + 2 < MF_CAND_LIMIT) cnt++; } while (0)

    /* rep distances, :266-299 */
    for (uint32_t i = 0; i < 4; i++) {
        if (rep_dist[i] >= e->vld_rge) continue;
        uint32_t cmp_pos = wrap_back(e, wpos, rep_dist[i]);
        uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
        const uint8_t *pmatch = e->wnd + cmp_pos;
        if (minlen >= climit || pmatch[minlen] != pcur[minlen]) continue;
        uint32_t match_len = prefix_len(pcur, pmatch, climit);
        if (match_len && i == 0) PUSH_CAND(1, 1);   /* rep0len1 */
        if (match_len > minlen) {
            minlen = match_len;
            PUSH_CAND(match_len, 1 + i);
            if (match_len >= e->good_len) { dist = 0xFFFFFFFFu; break; }
        }
    }

    if (e->ht_low) {
        /* HT2 :303-332 (strict `wpos > dist`) and HT3 :334-363 */
        for (int t = 0; t < 2; t++) {
            uint32_t entry = t == 0 ? e->ht2[h2] : e->ht3[h3];
            if (!(e->pos - entry > dist)) continue;
            dist = e->pos - entry;
            if (dist >= e->vld_rge) continue;
            uint32_t cmp_pos = t == 0 ? (wpos > dist ? wpos - dist : wpos + e->wnd_size - dist)
                                      : wrap_back(e, wpos, dist);
            uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
            const uint8_t *pmatch = e->wnd + cmp_pos;
            if (minlen >= climit || pmatch[minlen] != pcur[minlen]) continue;
            uint32_t match_len = prefix_len(pcur, pmatch, climit);
            if (match_len > minlen) {
                minlen = match_len;
                if (match_len <= 6 && dist >= kBound[match_len]) continue;
                PUSH_CAND(match_len, 4 + dist);
                if (match_len >= e->good_len) dist = 0xFFFFFFFFu;
            }
        }
        e->ht2[h2] = e->pos;
        e->ht3[h3] = e->pos;
    }

    if (e->bt_head) {
        /* :373-402 head candidate beyond the tree range */
        dist = e->pos - e->bt_head[hbt];
        uint32_t *l = &e->bt_nodes[(size_t)e->bt_pos * 2], *r = &e->bt_nodes[(size_t)e->bt_pos * 2 + 1];
        if (dist >= e->bt_size && dist < e->vld_rge) {
            uint32_t cmp_pos = wrap_back(e, wpos, dist);
            uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
            const uint8_t *pmatch = e->wnd + cmp_pos;
            if (!(minlen >= climit || pmatch[minlen] != pcur[minlen])) {
                uint32_t match_len = prefix_len(pcur, pmatch, climit);
                if (match_len > minlen) {
                    minlen = match_len;
                    if (!(match_len <= 6 && dist >= kBound[match_len])) {
                        PUSH_CAND(match_len, 4 + dist);
                        if (match_len >= e->good_len) dist = 0xFFFFFFFFu;
                    }
                }
            }
        }
        /* :404-451 descend + insert */
        uint32_t lenl = 0, lenr = 0;
        for (uint32_t cyc = 0;; cyc++) {
            if (cyc >= e->bt_cyc || dist >= e->bt_size || dist >= e->vld_rge) { *l = *r = 0; break; }
            uint32_t cmp_pos = wrap_back(e, wpos, dist);
            uint32_t clen = UMIN(lenl, lenr);
            uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
            if (clen >= climit) { *l = *r = 0; break; }
            uint32_t bt_npos = e->bt_pos >= dist ? e->bt_pos - dist : e->bt_pos + e->bt_size - dist;
            uint32_t *tlast = &e->bt_nodes[(size_t)bt_npos * 2];
            const uint8_t *pmatch = e->wnd + cmp_pos;
            if (pcur[clen] == pmatch[clen]) {
                clen++;
                while (clen < climit && pcur[clen] == pmatch[clen]) clen++;
                if (clen > minlen) {
                    minlen = clen;
                    if (clen > 6 || dist < kBound[clen]) PUSH_CAND(clen, 4 + dist);
                }
                if (clen >= e->good_len) { *l = tlast[0]; *r = tlast[1]; dist = 0xFFFFFFFFu; break; }
                else if (clen >= climit) { *l = *r = 0; break; }
            }
            if (pmatch[clen] < pcur[clen]) {
                *l = e->pos - dist;
                dist = e->pos - *(l = &tlast[1]);
                lenl = clen;
            } else {
                *r = e->pos - dist;
                dist = e->pos - *(r = &tlast[0]);
                lenr = clen;
            }
        }
        e->bt_head[hbt] = e->pos;
        if (++e->bt_pos >= e->bt_size) e->bt_pos -= e->bt_size;
    }

    /* HT6 bucket, :453-491 */
    uint32_t *b = e->ht6 + (size_t)h6 * e->ht_width;
    uint32_t cands = UMIN(e->ht_width, e->ht_cyc);
    for (uint32_t i = 0; i < cands; i++) {
        if (e->pos - b[i] <= dist) continue;
        dist = e->pos - b[i];
        if (dist >= e->vld_rge) continue;
        uint32_t cmp_pos = wrap_back(e, wpos, dist);
        uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
        const uint8_t *pmatch = e->wnd + cmp_pos;
        if (minlen >= climit || pmatch[minlen] != pcur[minlen]) continue;
        uint32_t match_len = prefix_len(pcur, pmatch, climit);
        if (match_len > minlen) {
            minlen = match_len;
            if (match_len <= 6 && dist >= kBound[match_len]) continue;
            PUSH_CAND(match_len, 4 + dist);
            if (match_len >= e->good_len) { dist = 0xFFFFFFFFu; break; }
        }
    }
    if (e->ht_width) {
        for (uint32_t i = cands - 1; i > 0; i--) b[i] = b[i - 1];
        b[0] = e->pos;
    }
#undef PUSH_CAND
    if (++e->pos >= 0xFFFFFFF0u) mf_normalize(e);
    return cnt;
}

/* the lazy parser's preference predicate, csc_mf.cpp:508-516 and :570-582 */
static int second_better(MFUnit u1, MFUnit u2)
{
    static const uint32_t cof[] = {0, 4, 8, 12};
    return u2.len > 1 && (
        (u2.len > u1.len + 3)
        || (u2.len > u1.len && u2.dist <= 4)
        || (u2.len + 2 > u1.len && u2.dist <= 4 && u1.dist > 4)
        || (u2.len >= u1.len && (u2.dist >> cof[u2.len - u1.len]) <= u1.dist)
        || (u2.len < u1.len && u2.len + 2 >= u1.len && u1.dist > 4
            && (u1.dist >> cof[u1.len - u2.len]) > u2.dist));
}

/* MatchFinder::FindMatch, csc_mf.cpp:497-524 */
static MFUnit mf_find_best(OrcEnc *e, const uint32_t *rep_dist, uint32_t wnd_pos, uint32_t limit)
{
    e->mfcand[0].len = 1; e->mfcand[0].dist = 0;
    uint32_t n = mf_find_match(e, e->mfcand + 1, rep_dist, wnd_pos, limit);
    uint32_t best = 0;
    for (uint32_t i = 1; i <= n; i++) {
        if (!best) { best = i; continue; }
        if (second_better(e->mfcand[best], e->mfcand[i])) best = i;
    }
    return e->mfcand[best];
}

/* MatchFinder::TestFind, csc_mf.cpp:526-568 (bucket index lacks +i) */
static int mf_test_find(OrcEnc *e, uint32_t wpos, const uint8_t *src, uint32_t limit)
{
    uint32_t dists[9] = {e->wnd_size, e->wnd_size};
    uint32_t depth = 0;
    uint32_t h = hash2(src);
    if (h % 16) return 0;
    if (e->ht_width) {
        h = hash6(src, e->ht_bits);
        for (uint32_t i = 0; i < e->ht_width && i < 8; i++)
            dists[depth++] = e->pos - e->ht6[(size_t)h * e->ht_width];
    }
    if (e->bt_head) {
        h = hash6(src, e->bt_bits);
        dists[depth++] = e->pos - e->bt_head[h];
    }
    for (uint32_t i = 0; i < depth; i++) {
        uint32_t dist = dists[i];
        if (dist >= e->vld_rge) continue;
        uint32_t cmp_pos = wrap_back(e, wpos, dist);
        uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
        if (prefix_len(src, e->wnd + cmp_pos, climit) > 18) return 1;
    }
    return 0;
}

/* MatchFinder::FindMatchWithPrice, csc_mf.cpp:584-625 */
static void mf_find_priced(OrcEnc *e, uint32_t state, MFUnit *ret, const uint32_t *rep_dist,
                           uint32_t wnd_pos, uint32_t limit)
{
    e->mfcand[0].len = 1; e->mfcand[0].dist = 0;
    uint32_t n = mf_find_match(e, e->mfcand + 1, rep_dist, wnd_pos, limit);
    ret[0] = e->mfcand[n];
    if (ret[0].len >= e->good_len) return;
    ret[1].dist = 0;
    uint32_t lpos = 1;
    for (uint32_t i = 1; i <= n; i++) {
        uint32_t distprice, rdist;
        if (e->mfcand[i].len == 1 && e->mfcand[i].dist == 1) {
            ret[1].len = rep0len1_price(e, state);
            ret[1].dist = 1;
            continue;
        } else if (e->mfcand[i].dist <= 4) {
            distprice = rep_dist_price(e, state, e->mfcand[i].dist - 1);
            rdist = 0;
        } else {
            distprice = match_dist_price(e, state, e->mfcand[i].dist - 5);
            rdist = e->mfcand[i].dist - 4;
        }
        while (lpos < e->mfcand[i].len) {
            lpos++;
            if (lpos <= 6 && rdist >= kBound[lpos]) { ret[lpos].dist = 0; continue; }
            ret[lpos].dist = e->mfcand[i].dist;
            ret[lpos].len = distprice + match_len_price(e, lpos - 2);
        }
    }
}

/* ===================================================================== */
/* LZ, csc_lz.cpp                                                        */

/* LZ::encode_nonlit, csc_lz.cpp:127-154 */
static void lz_encode_nonlit(OrcEnc *e, MFUnit u)
{
    if (u.dist <= 4) {
        if (u.len == 1 && u.dist == 1) encode_rep0len1(e);
        else {
            encode_rep_match(e, u.dist - 1, u.len - 2);
            uint32_t d = e->rep_dist[u.dist - 1];
            for (uint32_t k = u.dist - 1; k > 0; k--) e->rep_dist[k] = e->rep_dist[k - 1];
            e->rep_dist[0] = d;
        }
    } else {
        encode_match(e, u.dist - 5, u.len - 2);
        e->rep_dist[3] = e->rep_dist[2]; e->rep_dist[2] = e->rep_dist[1];
        e->rep_dist[1] = e->rep_dist[0]; e->rep_dist[0] = u.dist - 4;
    }
}

/* LZ::compress_normal, csc_lz.cpp:156-199 */
static void lz_compress_normal(OrcEnc *e, uint32_t size, int lazy)
{
    MFUnit u1 = {0, 0}, u2;
    int got_u1 = 0;
    for (uint32_t i = 0; i < size;) {
        if (!got_u1) u1 = mf_find_best(e, e->rep_dist, e->wnd_curpos, size - i);
        if (u1.len == 1 || !lazy || u1.len >= e->lz_good_len) {
            if (u1.dist == 0) encode_literal(e, e->wnd[e->wnd_curpos]);
            else lz_encode_nonlit(e, u1);
            mf_slide_pos(e, e->wnd_curpos, u1.len, size - i);
            i += u1.len; e->wnd_curpos += u1.len;
            if (u1.dist) e->ctx = e->wnd[e->wnd_curpos - 1];
            got_u1 = 0;
            continue;
        }
        u2 = mf_find_best(e, e->rep_dist, e->wnd_curpos + 1, size - i - 1);
        if (second_better(u1, u2)) {
            encode_literal(e, e->wnd[e->wnd_curpos]);
            mf_slide_pos(e, e->wnd_curpos, 1, size - i - 1);
            i++; e->wnd_curpos++;
            u1 = u2;
            got_u1 = 1;
        } else {
            lz_encode_nonlit(e, u1);
            mf_slide_pos(e, e->wnd_curpos + 1, u1.len - 1, size - i - 1);
            i += u1.len; e->wnd_curpos += u1.len;
            e->ctx = e->wnd[e->wnd_curpos - 1];
            got_u1 = 0;
        }
    }
}

/* LZ::ap_backward, csc_lz.cpp:335-362 */
static void lz_ap_backward(OrcEnc *e, int end)
{
    APUnit *ap = e->ap;
    for (int i = end; i;) {
        ap[ap[i].back_pos].next_pos = i;
        i = ap[i].back_pos;
    }
    for (int i = 0; i != end;) {
        uint32_t next = (uint32_t)ap[i].next_pos;
        if (ap[next].dist == 0) {
            encode_literal(e, ap[i].lit);
        } else if (ap[next].dist <= 4) {
            if (next - i == 1 && ap[next].dist == 1) encode_rep0len1(e);
            else encode_rep_match(e, ap[next].dist - 1, next - i - 2);
            e->ctx = ap[next - 1].lit;
        } else {
            encode_match(e, ap[next].dist - 5, next - i - 2);
            e->ctx = ap[next - 1].lit;
        }
        i = (int)next;
    }
    memcpy(e->rep_dist, ap[end].rep_dist, sizeof(e->rep_dist));
}

/* LZ::compress_advanced, csc_lz.cpp:207-333 */
static void lz_compress_advanced(OrcEnc *e, uint32_t size)
{
    APUnit *ap = e->ap;
    MFUnit *appt = e->appt;
    uint32_t apend = 0, apcur = 0;
    for (uint32_t i = 0; i < size;) {
        mf_find_priced(e, e->state, appt, e->rep_dist, e->wnd_curpos, size - i);
        if (appt[0].dist == 0) {
            encode_literal(e, e->wnd[e->wnd_curpos]);
            mf_slide_pos(e, e->wnd_curpos, 1, size - i);
            i++; e->wnd_curpos++;
            continue;
        }
        apcur = 0; apend = 1;
        ap[0].price = 0; ap[0].back_pos = 0;
        memcpy(ap[0].rep_dist, e->rep_dist, sizeof(e->rep_dist));
        ap[0].state = e->state;
        uint32_t aplimit = UMIN((uint32_t)AP_LIMIT, size - i);
        for (;;) {
            ap[apcur].lit = e->wnd[e->wnd_curpos];
            if (apcur) { /* fix cur state, :231-268 */
                int l = ap[apcur].back_pos;
                memcpy(ap[apcur].rep_dist, ap[l].rep_dist, sizeof(ap[l].rep_dist));
                if (ap[apcur].dist == 0) {
                    ap[apcur].state = (ap[l].state * 4) & 0x3F;
                } else if (ap[apcur].dist <= 4) {
                    uint32_t len = apcur - (uint32_t)l;
                    if (len == 1 && ap[apcur].dist == 1)
                        ap[apcur].state = (ap[l].state * 4 + 2) & 0x3F;
                    else {
                        ap[apcur].state = (ap[l].state * 4 + 3) & 0x3F;
                        uint32_t k = ap[apcur].dist - 1, tmp = ap[apcur].rep_dist[k];
                        if (k >= 1) { /* dist==1 leaves the array as is, :251-258 */
                            for (; k > 0; k--) ap[apcur].rep_dist[k] = ap[apcur].rep_dist[k - 1];
                            ap[apcur].rep_dist[0] = tmp;
                        }
                    }
                } else {
                    ap[apcur].state = (ap[l].state * 4 + 1) & 0x3F;
                    ap[apcur].rep_dist[0] = ap[apcur].dist - 4;
                    ap[apcur].rep_dist[1] = ap[l].rep_dist[0];
                    ap[apcur].rep_dist[2] = ap[l].rep_dist[1];
                    ap[apcur].rep_dist[3] = ap[l].rep_dist[2];
                }
                if (apcur < aplimit)
                    mf_find_priced(e, ap[apcur].state, appt, ap[apcur].rep_dist, e->wnd_curpos, size - i - apcur);
            }
            if (apcur == aplimit) { /* :271-275 */
                lz_ap_backward(e, (int)apcur);
                i += apcur;
                break;
            }
            if (appt[0].len == 1 && apcur + 1 == apend) { /* :277-285 */
                lz_ap_backward(e, (int)apcur);
                encode_literal(e, ap[apcur].lit);
                i += apcur;
                mf_slide_pos(e, e->wnd_curpos, 1, size - i);
                e->wnd_curpos++;
                i++;
                break;
            }
            if (apcur + 1 >= apend) ap[apend++].price = 0xFFFFFFFFu;
            if (appt[0].len >= e->lz_good_len || (appt[0].len > 1 && appt[0].len + apcur >= aplimit)) { /* :290-299 */
                lz_ap_backward(e, (int)apcur);
                i += apcur;
                lz_encode_nonlit(e, appt[0]);
                mf_slide_pos(e, e->wnd_curpos, appt[0].len, size - i);
                i += appt[0].len;
                e->wnd_curpos += appt[0].len;
                e->ctx = e->wnd[e->wnd_curpos - 1];
                break;
            }
            uint32_t lit_ctx = e->wnd_curpos ? e->wnd[e->wnd_curpos - 1] : 0;
            uint32_t cprice = literal_price(e, ap[apcur].state, lit_ctx, e->wnd[e->wnd_curpos]);
            if (cprice + ap[apcur].price < ap[apcur + 1].price) {
                ap[apcur + 1].dist = 0;
                ap[apcur + 1].back_pos = (int)apcur;
                ap[apcur + 1].price = cprice + ap[apcur].price;
            }
            if (appt[1].dist && appt[1].len + ap[apcur].price < ap[apcur + 1].price) {
                ap[apcur + 1].dist = 1;
                ap[apcur + 1].back_pos = (int)apcur;
                ap[apcur + 1].price = appt[1].len + ap[apcur].price;
            }
            uint32_t len = appt[0].len;
            while (apcur + len >= apend) ap[apend++].price = 0xFFFFFFFFu;
            while (len > 1) {
                if (appt[len].dist && appt[len].len + ap[apcur].price < ap[apcur + len].price) {
                    ap[apcur + len].dist = appt[len].dist;
                    ap[apcur + len].back_pos = (int)apcur;
                    ap[apcur + len].price = appt[len].len + ap[apcur].price;
                }
                len--;
            }
            apcur++;
            mf_slide_pos(e, e->wnd_curpos, 1, size - i - apcur);
            e->wnd_curpos++;
        }
    }
}

/* Test hook (NULL in liborc.so): tests/model/m3_model.c includes this file and points it at its own arrangement of
 * compress_advanced -- the one the HIP kernels use -- to prove that arrangement equal to the one above. */
static void (*orc_adv_hook)(OrcEnc *e, uint32_t size);
/* the same for compress_normal (tests/model/hp_model.c: the lazy levels' arrangement) */
static void (*orc_norm_hook)(OrcEnc *e, uint32_t size, int lazy);

/* LZ::EncodeNormal, csc_lz.cpp:61-100 */
static void lz_encode_normal(OrcEnc *e, const uint8_t *src, uint32_t size, uint32_t lz_mode)
{
    for (uint32_t i = 0; i < size;) {
        uint32_t cur = UMIN(e->wnd_size - e->wnd_curpos, size - i);
        cur = UMIN(cur, MIN_BLOCK);
        memcpy(e->wnd + e->wnd_curpos, src + i, cur);
        if (lz_mode == 1 || lz_mode == 2) { if (orc_norm_hook) orc_norm_hook(e, cur, lz_mode == 2); else lz_compress_normal(e, cur, lz_mode == 2); }
        else if (lz_mode == 3) { if (orc_adv_hook) orc_adv_hook(e, cur); else lz_compress_advanced(e, cur); }
        else if (lz_mode == 5) {
            mf_set_arg(e, 1, 1, 0, e->lz_good_len);
            mf_slide_pos_fast(e, e->wnd_curpos, cur);   /* compress_mf_skip :201-205 */
            e->wnd_curpos += cur;
            mf_set_arg(e, e->lz_bt_cyc, e->lz_ht_cyc, 1, e->lz_good_len);
        } else {
            exit(0); /* the reference prints "Error!" and exits, :86-88 */
        }
        if (e->wnd_curpos >= e->wnd_size) e->wnd_curpos = 0;
        i += cur;
    }
    if (lz_mode != 5) encode_match(e, 64, 0);
}

/* LZ::IsDuplicateBlock, csc_lz.cpp:102-112 */
static int lz_is_duplicate_block(OrcEnc *e, const uint8_t *src, uint32_t size)
{
    for (uint32_t i = 0; i < size; i++)
        if (mf_test_find(e, e->wnd_curpos, src + i, size - i)) return 1;
    return 0;
}

/* LZ::Init + Reset, csc_lz.cpp:10-52 */
static int lz_init(OrcEnc *e)
{
    const CSCProps *p = &e->props;
    e->wnd_size = (uint32_t)p->dict_size;
    if (e->wnd_size < 32 * KB) e->wnd_size = 32 * KB;
    if (e->wnd_size > 1024 * MB) e->wnd_size = 1024 * MB;
    e->wnd = (uint8_t *)e->alloc->Alloc(e->alloc, (size_t)e->wnd_size + 8);
    if (!e->wnd) return -1;
    if (mf_init(e, p->bt_size, p->bt_hash_bits, p->hash_width, p->hash_bits)) {
        e->alloc->Free(e->alloc, e->wnd);
        return -1;
    }
    e->lz_good_len = p->good_len; e->lz_bt_cyc = p->bt_cyc; e->lz_ht_cyc = p->hash_width;
    mf_set_arg(e, e->lz_bt_cyc, e->lz_ht_cyc, 1, e->lz_good_len);
    e->appt = (MFUnit *)e->alloc->Alloc(e->alloc, sizeof(MFUnit) * (e->lz_good_len + 1));
    e->ap = (APUnit *)e->alloc->Alloc(e->alloc, sizeof(APUnit) * (AP_LIMIT + 1));
    e->wnd_curpos = 0;
    e->rep_dist[0] = e->rep_dist[1] = e->rep_dist[2] = e->rep_dist[3] = e->wnd_size;
    memset(e->wnd, 0, (size_t)e->wnd_size + 8);
    model_reset(e);
    return 0;
}

/* ===================================================================== */
/* Analyzer, csc_analyzer.cpp                                            */
static void fill_logtable(uint32_t *t) /* csc_analyzer.cpp:9-15 */
{
    for (uint32_t i = 0; i < (MIN_BLOCK >> 4); i++)
        t[i] = (uint32_t)((double)100 * log((double)i * 16 + 8) / log((double)2));
    t[MIN_BLOCK >> 4] = (uint32_t)((double)100 * log((double)MIN_BLOCK) / log((double)2));
}

/* Analyzer::get_channel_idx, csc_analyzer.cpp:122-164 */
static int32_t an_channel_idx(const uint8_t *src, uint32_t size)
{
    static const uint32_t d[5] = {1, 2, 3, 4, 8};
    uint32_t same[5] = {0}, succ[5] = {0};
    for (uint32_t i = 0; i + 16 < size; i++)
        for (int k = 0; k < 5; k++) {
            same[k] += (src[i] == src[i + d[k]]);
            succ[k] += (uint32_t)abs((int)src[i] - (int)src[i + d[k]]);
        }
    uint32_t max_same = same[0], min_same = same[0], max_succ = succ[0], min_succ = succ[0], best = 0;
    for (uint32_t k = 0; k < 5; k++) {
        if (same[k] < min_same) min_same = same[k];
        if (same[k] > max_same) max_same = same[k];
        if (succ[k] > max_succ) max_succ = succ[k];
        if (succ[k] < min_succ) { min_succ = succ[k]; best = k; }
    }
    (void)max_same;
    if (((max_succ > succ[best] * 4) || (max_succ > succ[best] + 40 * size))
        && (same[best] > min_same * 3)
        && (same[0] < 0.3 * size))
        return (int32_t)best;
    return -1;
}

/* Analyzer::GetDltBpb, csc_analyzer.cpp:166-182 */
static uint32_t an_dlt_bpb(const uint32_t *lt, const uint8_t *src, uint32_t size, uint32_t chn)
{
    uint32_t freq[256] = {0};
    uint8_t prev = 0;
    for (uint32_t i = 0; i < chn; i++)
        for (uint32_t j = i; j < size; j += chn) {
            freq[(uint8_t)(src[j] - prev)]++;
            prev = src[j];
        }
    uint32_t bpb = size * lt[size >> 4];
    for (uint32_t i = 0; i < 256; i++) bpb -= freq[i] * lt[freq[i] >> 4];
    return bpb / size;
}

/* Analyzer::Analyze, csc_analyzer.cpp:184-239 */
static uint32_t an_analyze(const uint32_t *lt, const uint8_t *src, uint32_t size, uint32_t *bpb)
{
    uint32_t freq[256] = {0}, freq80[2] = {0};
    if (size > MIN_BLOCK) size = MIN_BLOCK;
    if (size < 512) return DT_SKIP;
    for (uint32_t i = 0; i < size; i++) freq[src[i]]++;
    uint32_t diff_num = 0, entropy = size * lt[size >> 4];
    for (uint32_t i = 0; i < 256; i++) {
        entropy -= freq[i] * lt[freq[i] >> 4];
        diff_num += (freq[i] > 0);
        freq80[i >> 7] += freq[i];
    }
    *bpb = entropy / size;
    uint32_t avg = size >> 8, alpha = 0;
    for (uint32_t i = 'a'; i <= 'z'; i++) alpha += freq[i];

    if (freq80[1] < (size >> 3)
        && (freq[' '] + freq['\n'] + freq[':'] + freq['.'] + freq['/'] > (size >> 4))
        && (freq['a'] + freq['e'] + freq['t'] > (size >> 4))
        && entropy > 300 * size && alpha > (size / 3))
        return DT_ENGTXT;
    if (freq[0x8b] > avg && freq[0x00] > avg * 2 && freq[0xE8] > 6) return DT_EXE;
    if (entropy > (log((double)diff_num - 2) / log((double)2) - 0.6) * 100.0 * size && diff_num < 16 && diff_num >= 6)
        return DT_ENTROPY;
    if (entropy < 400 * size && diff_num < 200) return DT_NORMAL;
    int32_t idx = an_channel_idx(src, size);
    if (idx != -1) return DT_DLT + (uint32_t)idx;
    if (entropy > 795 * size) return DT_BAD;
    else if (entropy > 780 * size) return DT_FAST;
    return DT_NORMAL;
}

/* ===================================================================== */
/* Filters, csc_filters.cpp                                              */
static const char kWords[123][8] = { /* csc_filters.cpp:8-38 (the data the format is defined by) */
    "",
    "ac","ad","ai","al","am","an","ar","as","at","ea","ec","ed","ee","el","en","er","es","et","id","ie",
    "ig","il","in","io","is","it","of","ol","on","oo","or","os","ou","ow","ul","un","ur","us","ba","be",
    "ca","ce","co","ch","de","di","ge","gh","ha","he","hi","ho","ra","re","ri","ro","rs","la","le","li",
    "lo","ld","ll","ly","se","si","so","sh","ss","st","ma","me","mi","ne","nc","nd","ng","nt","pa","pe",
    "ta","te","ti","to","th","tr","wa","ve",
    "all","and","but","dow","for","had","hav","her","him","his","man","mor","not","now","one","out",
    "she","the","was","wer","whi","whe","wit","you","any","are",
    "that","said","with","have","this","from","were","tion",
};

/* Filters::MakeWordTree, csc_filters.cpp:87-111 */
static void flt_make_trie(TrieNode *trie)
{
    uint32_t nodes = 1;
    uint8_t sym = 0x82;
    memset(trie, 0, sizeof(TrieNode) * 300);
    for (uint32_t i = 1; i < 123; i++) {
        uint32_t pos = 0;
        for (uint32_t j = 0; kWords[i][j] != 0; j++) {
            uint32_t idx = (uint32_t)(kWords[i][j] - 'a');
            if (trie[pos].next[idx]) pos = trie[pos].next[idx];
            else { trie[pos].next[idx] = nodes; pos = nodes; nodes++; }
        }
        trie[pos].symbol = sym++;
    }
}

static void flt_need_swap(OrcEnc *e, uint32_t size) /* csc_filters.cpp:141-147 etc. */
{
    if (e->swap_size < size) {
        if (e->swap_size > 0) e->alloc->Free(e->alloc, e->swap_buf);
        e->swap_buf = (uint8_t *)e->alloc->Alloc(e->alloc, size);
        e->swap_size = size;
    }
}

/* Filters::Forward_Delta, csc_filters.cpp:132-164 */
static void delta_forward(uint8_t *src, const uint8_t *copy, uint32_t size, uint32_t chn)
{
    uint32_t dst = 0;
    uint8_t prev = 0;
    for (uint32_t i = 0; i < chn; i++)
        for (uint32_t j = i; j < size; j += chn) {
            src[dst++] = (uint8_t)(copy[j] - prev);
            prev = copy[j];
        }
}
static void flt_forward_delta(OrcEnc *e, uint8_t *src, uint32_t size, uint32_t chn)
{
    if (size < 512) return;
    flt_need_swap(e, size);
    memcpy(e->swap_buf, src, size);
    delta_forward(src, e->swap_buf, size, chn);
}

/* Filters::Foward_Dict core, csc_filters.cpp:256-335; dst capacity = cap */
static uint32_t dict_forward(const TrieNode *trie, uint8_t *src, uint8_t *dst, uint32_t size, uint32_t cap)
{
    uint32_t i, dst_size = 0;
    for (i = 0; i < size - 5;) {
        if (dst_size > cap - 16) return 0;
        if (src[i] >= 'a' && src[i] <= 'z') {
            uint32_t sym = 0, longest = 0, pos = 0;
            for (uint32_t j = 0;;) {
                uint32_t idx = (uint32_t)src[i + j] - 'a';
                if (idx > 25 || trie[pos].next[idx] == 0) break;
                pos = trie[pos].next[idx];
                j++;
                if (trie[pos].symbol) { sym = trie[pos].symbol; longest = j; }
            }
            if (sym) { dst[dst_size++] = (uint8_t)sym; i += longest; continue; }
            dst[dst_size++] = src[i];
            i++;
        } else {
            if (src[i] >= 0x82) { dst[dst_size++] = 254; dst[dst_size++] = src[i]; }
            else dst[dst_size++] = src[i];
            i++;
        }
    }
    for (; i < size; i++) {
        if (src[i] >= 0x82) { dst[dst_size++] = 254; dst[dst_size++] = src[i]; }
        else dst[dst_size++] = src[i];
    }
    if (dst_size > size * 0.82) return 0;
    memset(dst + dst_size, 0x20, size - dst_size);
    memcpy(src, dst, size);
    return 1;
}
static uint32_t flt_forward_dict(OrcEnc *e, uint8_t *src, uint32_t size)
{
    if (size < 16384) return 0;
    flt_need_swap(e, size);
    return dict_forward(e->trie, src, e->swap_buf, size, e->swap_size);
}

/* Shelwien's E8/E9 state machine, csc_filters.cpp:508-598 */
static void e89_init(OrcEnc *e) { e->ecs = 0xFF; e->x0 = e->x1 = 0; e->ei = 0; e->ek = 5; }
static int32_t e89_cache_byte(OrcEnc *e, int32_t c)
{
    int32_t d = (e->ecs & 0x80) ? -1 : (int32_t)(uint8_t)e->x1;
    e->x1 >>= 8; e->x1 |= (e->x0 << 24);
    e->x0 >>= 8; e->x0 |= ((uint32_t)c << 24);
    e->ecs <<= 1; e->ei++;
    return d;
}
static uint32_t e89_xswap(uint32_t x)
{
    x <<= 7;
    return (x >> 24) | ((uint32_t)(uint8_t)(x >> 16) << 8) | ((uint32_t)(uint8_t)(x >> 8) << 16)
         | ((uint32_t)(uint8_t)x << (24 - 7));
}
static int32_t e89_forward(OrcEnc *e, int32_t c)
{
    if (e->ei >= e->ek) {
        if ((e->x1 & 0xFE000000u) == 0xE8000000u) {
            e->ek = e->ei + 4;
            uint32_t x = e->x0 - 0xFF000000u;
            if (x < 0x02000000u) {
                x = (x + e->ei) & 0x01FFFFFFu;
                x = e89_xswap(x);
                e->x0 = x + 0xFF000000u;
            }
        }
    }
    return e89_cache_byte(e, c);
}
static int32_t e89_flush(OrcEnc *e)
{
    if (e->ecs != 0xFF) {
        while (e->ecs & 0x80) { e89_cache_byte(e, 0); ++e->ecs; }
        int32_t d = e89_cache_byte(e, 0); ++e->ecs;
        return d;
    }
    e89_init(e);
    return -1;
}
static void flt_forward_e89(OrcEnc *e, uint8_t *src, uint32_t size) /* csc_filters.cpp:588-598 */
{
    uint32_t i, j;
    int32_t c;
    e89_init(e);
    for (i = 0, j = 0; i < size; i++) {
        c = e89_forward(e, src[i]);
        if (c >= 0) src[j++] = (uint8_t)c;
    }
    while ((c = e89_flush(e)) >= 0) src[j++] = (uint8_t)c;
}

/* ===================================================================== */
/* CSCEncoder, csc_encoder_main.cpp                                      */

/* CSCEncoder::compress_block, csc_encoder_main.cpp:35-83 */
static void enc_compress_block(OrcEnc *e, uint8_t *src, uint32_t size, uint32_t type)
{
    if (size == 0) return;
    uint32_t mode = e->props.lz_mode;
    if (type == DT_NORMAL) {
        encode_int(e, type);
        lz_encode_normal(e, src, size, mode);
    } else if (type == DT_EXE) {
        encode_int(e, type);
        flt_forward_e89(e, src, size);
        lz_encode_normal(e, src, size, mode);
    } else if (type == DT_ENGTXT) {
        if (flt_forward_dict(e, src, size)) { encode_int(e, type); encode_int(e, size); }
        else encode_int(e, DT_NORMAL);
        lz_encode_normal(e, src, size, mode);
    } else if (type == DT_FAST) {
        encode_int(e, DT_NORMAL);
        lz_encode_normal(e, src, size, mode);
    } else if (type == DT_BAD) {
        encode_int(e, type);
        lz_encode_normal(e, src, size, 5);
        compress_bad(e, src, size);
    } else if (type == DT_ENTROPY) {
        encode_int(e, type);
        lz_encode_normal(e, src, size, 5);
        compress_literals(e, src, size);
    } else if (type >= DT_DLT && type < DT_DLT + 5) {
        uint32_t chn = kDltIndex[type - DT_DLT];
        encode_int(e, type);
        lz_encode_normal(e, src, size, 5);
        flt_forward_delta(e, src, size, chn);
        compress_rle(e, src, size);
    }
}

/* CSCEncoder::Compress, csc_encoder_main.cpp:85-147 */
static void enc_compress(OrcEnc *e, uint8_t *src, uint32_t size)
{
    uint32_t last_type = DT_NORMAL, this_type, last_begin = 0, last_size = 0, bpb = 0;
    int use_filters = (e->props.DLTFilter + e->props.EXEFilter + e->props.TXTFilter) != 0;
    for (uint32_t i = 0; i < size;) {
        uint32_t cur = UMIN(MIN_BLOCK, size - i);
        this_type = use_filters ? an_analyze(e->log_table, src + i, cur, &bpb) : DT_NORMAL;
        if (this_type == DT_SKIP) this_type = last_type;
        if (this_type != DT_NORMAL) {
            if (this_type == DT_EXE && e->props.EXEFilter == 0) this_type = DT_NORMAL;
            else if (this_type == DT_ENGTXT && e->props.TXTFilter == 0) this_type = DT_NORMAL;
            else if (this_type >= DT_DLT && e->props.DLTFilter == 0) this_type = DT_NORMAL;
        }
        if (this_type >= DT_DLT
            && an_dlt_bpb(e->log_table, src + i, cur, kDltIndex[this_type - DT_DLT]) >= bpb * 0.95)
            this_type = DT_NORMAL;
        if (this_type >= DT_NO_LZ && lz_is_duplicate_block(e, src + i, cur)) this_type = DT_NORMAL;
        if (last_type != this_type || last_size + cur > e->props.raw_blocksize) {
            if (last_size) {
                enc_compress_block(e, src + last_begin, last_size, last_type);
                encode_int(e, 0);
            }
            last_begin = i;
            last_size = 0;
        }
        last_type = this_type;
        last_size += cur;
        i += cur;
    }
    if (last_size) {
        enc_compress_block(e, src + last_begin, last_size, last_type);
        encode_int(e, 1);
        coder_flush(e);
    }
}

/* ===================================================================== */
/* public API, csc_enc.cpp                                               */

void CSCEncProps_Init(CSCProps *p, uint32_t dict_size, int level) /* csc_enc.cpp:16-97 */
{
    dict_size += 10 * KB;
    if (dict_size < 32 * KB) dict_size = 32 * KB;
    if (dict_size > 1024 * MB) dict_size = 1024 * MB;
    p->dict_size = dict_size;
    if (level < 1) level = 1;
    if (level > 5) level = 5;
    p->DLTFilter = 1; p->TXTFilter = 1; p->EXEFilter = 1;
    p->csc_blocksize = 64 * KB;
    p->raw_blocksize = 2 * MB;
    uint32_t hbits;
    if (dict_size < MB) hbits = 19;
    else if (dict_size <= 4 * MB) hbits = 20;
    else if (dict_size <= 16 * MB) hbits = 21;
    else if (dict_size <= 64 * MB) hbits = 22;
    else if (dict_size <= 256 * MB) hbits = 23;
    else hbits = 24;
    while (((uint32_t)1 << hbits) > dict_size) hbits--;
    if (dict_size <= 16 * MB) p->bt_size = dict_size;
    else if (dict_size <= 64 * MB) p->bt_size = (dict_size - 16 * MB) / 2 + 16 * MB;
    else if (dict_size <= 256 * MB) p->bt_size = (dict_size - 64 * MB) / 4 + 40 * MB;
    else p->bt_size = (dict_size - 256 * MB) / 8 + 88 * MB;
    p->good_len = 32;
    p->hash_bits = (uint8_t)hbits;
    p->bt_hash_bits = (uint8_t)(hbits + 1);
    switch (level) {
    case 1: p->hash_width = 1; p->lz_mode = 2; p->bt_size = 0; p->hash_bits++; break;
    case 2: p->hash_width = 8; p->lz_mode = 2; p->bt_size = 0; p->good_len = 24; p->hash_bits--; break;
    case 3: p->hash_width = 2; p->lz_mode = 3; p->bt_size = 0; p->good_len = 16; p->hash_bits++; break;
    case 4: p->hash_width = 8; p->lz_mode = 3; p->bt_size = 0; p->good_len = 24; p->hash_bits--; break;
    case 5: p->lz_mode = 3; p->good_len = 48; p->bt_cyc = 32; p->hash_width = 0; break;
    }
    if (p->bt_size == p->dict_size) p->hash_width = 0;
}

uint64_t CSCEnc_EstMemUsage(const CSCProps *p) /* csc_enc.cpp:99-112 */
{
    uint64_t ret = 0;
    ret += p->dict_size;
    ret += p->csc_blocksize * 2;
    if (p->bt_size) ret += (uint64_t)(int64_t)(((1 << p->bt_hash_bits) + 2 * p->bt_size)) * sizeof(uint32_t);
    if (p->hash_width) ret += (uint64_t)(int64_t)(p->hash_width * (1 << p->hash_bits)) * sizeof(uint32_t);
    ret += 80 * KB * sizeof(uint32_t);
    ret += 256 * 256 * sizeof(uint32_t) * 2;
    ret += 2 * MB;
    return ret;
}

void CSCEnc_WriteProperties(const CSCProps *props, uint8_t *s, int full) /* csc_enc.cpp:145-158 */
{
    (void)full;
    s[0] = (props->dict_size >> 24) & 0xff; s[1] = (props->dict_size >> 16) & 0xff;
    s[2] = (props->dict_size >> 8) & 0xff;  s[3] = props->dict_size & 0xff;
    s[4] = (props->csc_blocksize >> 16) & 0xff; s[5] = (props->csc_blocksize >> 8) & 0xff;
    s[6] = props->csc_blocksize & 0xff;
    s[7] = (props->raw_blocksize >> 16) & 0xff; s[8] = (props->raw_blocksize >> 8) & 0xff;
    s[9] = props->raw_blocksize & 0xff;
}

static void enc_free_all(OrcEnc *e)
{
    ISzAlloc *a = e->alloc;
    a->Free(a, e->rc_buf); a->Free(a, e->bc_buf);
    a->Free(a, e->wnd); a->Free(a, e->mfbuf); a->Free(a, e->appt); a->Free(a, e->ap);
    a->Free(a, e->p_lit); a->Free(a, e->p_delta);
    if (e->swap_size > 0) a->Free(a, e->swap_buf);
    a->Free(a, e);
}

/* Test hook (tests/test_gpu_parity.py::test_pos_renormalisation...): start the match finder's position counter somewhere else.
 * MatchFinder::Init leaves pos_ = vld_rge_ over zeroed tables (csc_mf.cpp:56-57,81); any larger start is the same situation --
 * every table entry reads as out of range -- and brings normalize() (csc_mf.cpp:108-114, pos_ >= 0xFFFFFFF0) within reach of
 * a test instead of 3.2 GB into a stream.  Call right after CSCEnc_Create. */
void orc_debug_set_pos(CSCEncHandle h, uint32_t pos) { ((OrcEnc *)h)->pos = pos; }
uint32_t orc_debug_get_pos(CSCEncHandle h) { return ((OrcEnc *)h)->pos; }

CSCEncHandle CSCEnc_Create(const CSCProps *props, ISeqOutStream *os, ISzAlloc *alloc) /* csc_enc.cpp:114-133 */
{
    if (alloc == NULL) alloc = &g_default_alloc;
    OrcEnc *e = (OrcEnc *)alloc->Alloc(alloc, sizeof(OrcEnc));
    if (!e) return NULL;
    memset(e, 0, sizeof(*e));
    e->alloc = alloc; e->os = os; e->props = *props;
    e->bsize = props->csc_blocksize;
    /* CSCEncoder::Init, csc_encoder_main.cpp:5-33 */
    fill_logtable(e->log_table);
    flt_make_trie(e->trie);
    coder_reset_state(e);
    e->outsize = 0;
    e->rc_buf = (uint8_t *)alloc->Alloc(alloc, e->bsize);
    e->bc_buf = (uint8_t *)alloc->Alloc(alloc, e->bsize);
    fill_p2bits(e->p_2_bits);
    e->p_lit = (uint32_t *)alloc->Alloc(alloc, 256 * 256 * sizeof(uint32_t));
    if (!e->rc_buf || !e->bc_buf || !e->p_lit || lz_init(e) < 0) { enc_free_all(e); return NULL; }
    return e;
}

void CSCEnc_Destroy(CSCEncHandle h) { enc_free_all((OrcEnc *)h); } /* csc_enc.cpp:135-143 */

/* try { Compress } catch (int), csc_enc.cpp:175-179 */
static int enc_try_compress(OrcEnc *e, uint8_t *buf, uint32_t size)
{
    int code = setjmp(e->on_error);
    if (code != 0) return -code;
    enc_compress(e, buf, size);
    return 0;
}

int CSCEnc_Encode(CSCEncHandle h, ISeqInStream *is, ICompressProgress *progress) /* csc_enc.cpp:160-191 */
{
    OrcEnc *e = (OrcEnc *)h;
    int ret = 0;
    uint8_t *buf = (uint8_t *)e->alloc->Alloc(e->alloc, e->props.raw_blocksize);
    uint64_t insize = 0;
    for (;;) {
        size_t size = e->props.raw_blocksize;
        ret = is->Read(is, buf, &size);
        if (ret >= 0 && size) {
            insize += size;
            ret = enc_try_compress(e, buf, (uint32_t)size);
            if (progress)
                progress->Progress(progress, insize, (uint64_t)(e->outsize + e->rc_size + e->bc_size));
        } else if (ret < 0) {
            ret = READ_ERROR;
        }
        if (ret < 0 || size == 0) break;
    }
    e->alloc->Free(e->alloc, buf);
    return ret;
}

int CSCEnc_Encode_Flush(CSCEncHandle h) /* csc_enc.cpp:193-203 */
{
    OrcEnc *e = (OrcEnc *)h;
    int code = setjmp(e->on_error);
    if (code != 0) return -code;
    encode_int(e, SIG_EOF);
    coder_flush(e);
    return 0;
}

/* ===================================================================== */
/* probes for intermediate goldens                                       */
uint32_t orc_analyze_block(const uint8_t *src, uint32_t size, uint32_t *bpb)
{
    uint32_t lt[(MIN_BLOCK >> 4) + 1];
    fill_logtable(lt);
    return an_analyze(lt, src, size, bpb);
}
uint32_t orc_dlt_bpb(const uint8_t *src, uint32_t size, uint32_t chn)
{
    uint32_t lt[(MIN_BLOCK >> 4) + 1];
    fill_logtable(lt);
    return an_dlt_bpb(lt, src, size, chn);
}
void orc_forward_e89(uint8_t *buf, uint32_t size)
{
    OrcEnc *e = (OrcEnc *)calloc(1, sizeof(OrcEnc));
    flt_forward_e89(e, buf, size);
    free(e);
}
uint32_t orc_forward_dict(uint8_t *buf, uint32_t size)
{
    if (size < 16384) return 0;
    TrieNode *trie = (TrieNode *)calloc(300, sizeof(TrieNode));
    uint8_t *tmp = (uint8_t *)malloc((size_t)size + 32);
    flt_make_trie(trie);
    uint32_t r = dict_forward(trie, buf, tmp, size, size);
    free(tmp); free(trie);
    return r;
}
void orc_forward_delta(uint8_t *buf, uint32_t size, uint32_t chn)
{
    if (size < 512) return;
    uint8_t *tmp = (uint8_t *)malloc(size);
    memcpy(tmp, buf, size);
    delta_forward(buf, tmp, size, chn);
    free(tmp);
}
void orc_tables(uint32_t p2bits[512], uint32_t logtab[513])
{
    fill_p2bits(p2bits);
    fill_logtable(logtab);
}
