/*
 * oracle/orc_decoder.c -- CPU restatement of the libcsc DECODE path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see orc_encoder.c header).  Follows
 * /root/reference/src/libcsc/csc_dec.cpp (decoder core + CSCDec_* API),
 * csc_memio.cpp:5-81 (ReadBlock) and the inverse filters of
 * csc_filters.cpp:337-399,560-610.  Parity pinned against oracle/_ref by
 * tests/test_oracle_vs_ref.py (both decoders fed the same streams, including
 * corrupted and truncated ones).
 */
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

#include "orc_api.h"

#define KB 1024u
#define MB 1048576u
#define MIN_BLOCK (8u * KB)
#define UMIN(a, b) ((a) < (b) ? (a) : (b))

enum { DT_NORMAL = 1, DT_ENGTXT = 2, DT_EXE = 3, DT_ENTROPY = 7, DT_BAD = 8, SIG_EOF = 9, DT_DLT = 0x10 };
static const uint32_t kDltIndex[5] = {1, 2, 3, 4, 8};

static const uint32_t kDistTable[33] = { /* csc_dec.cpp:44-54 */
    0, 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513, 1025, 2049, 4097, 8193,
    16385, 32769, 65537, 131073, 262145, 524289, 1048577, 2097153,
    4194305, 8388609, 16777217, 33554433, 67108865, 134217729, 268435457,
    536870913, 1073741825,
};
static const uint32_t kRev16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};

static const char kWords[123][8] = { /* csc_filters.cpp:8-38 */
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
#define MAX_SYMBOL (0x82 + 122) /* Filters::maxSymbol after MakeWordTree, csc_filters.cpp:92-110 */

typedef struct DataBlock { /* csc_memio.h:14-18 */
    struct DataBlock *next;
    uint32_t size;
    uint8_t buf[1];
} DataBlock;

typedef struct OrcDec {
    ISzAlloc *alloc;
    ISeqInStream *is;
    jmp_buf on_error;
    uint32_t bsize, raw_blocksize;
    DataBlock *rc_blocks, *bc_blocks;

    uint8_t *rc_buf, *bc_buf;
    uint32_t rc_bufsize, bc_bufsize;   /* set by the first ReadBlock, csc_dec.cpp:336 */
    uint32_t rc_range, rc_code;
    uint32_t bc_curbits, bc_curval;
    uint32_t rc_pos, bc_pos;            /* prc_/pbc_ as offsets */
    uint32_t rc_size, bc_size;
    int64_t outsize;

    uint32_t p_rle_flag;
    uint32_t *p_lit, *p_delta;
    uint32_t p_repdist[64 * 4];
    uint32_t p_dist[8 + 16 * 2 + 32 * 4];
    uint32_t p_dist_extra[29 * 16];
    uint32_t p_len_slot[2], p_len_x1[8], p_len_x2[8], p_len_x3[128];
    uint32_t p_longlen, ctx;
    uint32_t p_state[64 * 3];
    uint32_t state;

    uint32_t rep_dist[4];
    uint32_t wnd_size;
    uint8_t *wnd;
    uint32_t wnd_curpos;

    uint8_t *swap_buf;
    uint32_t swap_size;
    uint32_t x0, x1, ei, ek;
    uint8_t ecs;
} OrcDec;

static void *def_alloc(void *p, size_t n) { (void)p; return malloc(n); }
static void def_free(void *p, void *a) { (void)p; free(a); }
static ISzAlloc g_default_alloc = {def_alloc, def_free};

/* MemIO::ReadBlock, csc_memio.cpp:5-81 */
static int read_block(OrcDec *d, uint8_t *buffer, uint32_t *size, int rc1bc0)
{
    DataBlock **blist = rc1bc0 ? &d->rc_blocks : &d->bc_blocks;
    if (*blist) {
        *size = (*blist)->size;
        memcpy(buffer, (*blist)->buf, *size);
        DataBlock *tb = (*blist)->next;
        d->alloc->Free(d->alloc, *blist);
        *blist = tb;
        return (int)*size;
    }
    for (;;) {
        uint8_t fb;
        size_t iosize = 1;
        d->is->Read(d->is, &fb, &iosize);
        if (iosize != 1) return -1;
        uint32_t cur;
        if ((fb >> 6) & 1) cur = d->bsize;
        else {
            uint8_t sb[3];
            iosize = 3;
            d->is->Read(d->is, sb, &iosize);
            if (iosize != 3) return -1;
            cur = ((uint32_t)sb[0] << 16) + ((uint32_t)sb[1] << 8) + sb[2];
        }
        if (!cur || cur > d->bsize) return -1;
        iosize = cur;
        *size = cur;
        if (((fb >> 7) & 1) == rc1bc0) {
            d->is->Read(d->is, buffer, &iosize);
            if (iosize != cur) return -1;
            break;
        }
        DataBlock *nb = (DataBlock *)d->alloc->Alloc(d->alloc, sizeof(DataBlock) + cur);
        nb->size = cur;
        nb->next = NULL;
        d->is->Read(d->is, nb->buf, &iosize);
        if (iosize != cur) { d->alloc->Free(d->alloc, nb); return -1; }
        DataBlock **tail = rc1bc0 ? &d->bc_blocks : &d->rc_blocks;
        while (*tail) tail = &(*tail)->next;
        *tail = nb;
    }
    return (int)*size;
}

/* DecodeBit macro, csc_dec.cpp:10-35 (v = 2v + bit) */
static inline uint32_t dec_bit(OrcDec *d, uint32_t v, uint32_t *p)
{
    if (d->rc_range < (1u << 24)) {
        d->rc_range <<= 8;
        d->rc_code = (d->rc_code << 8) + d->rc_buf[d->rc_pos++];
        d->rc_size++;
        if (d->rc_size >= d->rc_bufsize) {
            d->outsize += d->rc_size;
            if (read_block(d, d->rc_buf, &d->rc_bufsize, 1) < 0) longjmp(d->on_error, -READ_ERROR);
            d->rc_size = 0;
            d->rc_pos = 0;
        }
    }
    uint32_t bound = (d->rc_range >> 12) * *p;
    if (d->rc_code < bound) {
        d->rc_range = bound;
        *p += (0xFFF - *p) >> 5;
        return v + v + 1;
    }
    d->rc_range -= bound;
    d->rc_code -= bound;
    *p -= *p >> 5;
    return v + v;
}

/* coder_decode_direct, csc_dec.cpp:65-88 */
static uint32_t dec_direct16(OrcDec *d, uint32_t len)
{
    while (d->bc_curbits < len) {
        d->bc_curval = (d->bc_curval << 8) | d->bc_buf[d->bc_pos++];
        d->bc_size++;
        if (d->bc_size >= d->bc_bufsize) {
            d->outsize += d->bc_size;
            if (read_block(d, d->bc_buf, &d->bc_bufsize, 0) < 0) longjmp(d->on_error, -READ_ERROR);
            d->bc_size = 0;
            d->bc_pos = 0;
        }
        d->bc_curbits += 8;
    }
    uint32_t r = (d->bc_curval >> (d->bc_curbits - len)) & ((1u << len) - 1);
    d->bc_curbits -= len;
    return r;
}
static uint32_t dec_direct(OrcDec *d, uint32_t l) /* DecodeDirect, csc_dec.cpp:37-42 */
{
    if (l <= 16) return dec_direct16(d, l);
    uint32_t v = dec_direct16(d, l - 16) << 16;
    return v | dec_direct16(d, 16);
}

static uint32_t decode_int(OrcDec *d) /* csc_dec.cpp:90-97 */
{
    uint32_t slot = dec_direct(d, 5);
    uint32_t num = dec_direct(d, slot == 0 ? 1 : slot);
    if (slot) num += (1u << slot);
    return num;
}

static uint32_t decode_byte_tree(OrcDec *d, uint32_t *row)
{
    uint32_t c = 1;
    do { c = dec_bit(d, c, &row[c]); } while (c < 0x100);
    return c & 0xFF;
}

static uint32_t decode_matchlen_1(OrcDec *d) /* csc_dec.cpp:187-220 */
{
    uint32_t v, lenbase, *p, i = 1;
    v = dec_bit(d, 0, &d->p_len_slot[0]);
    if (v == 0) { p = d->p_len_x1; lenbase = 0; }
    else {
        v = dec_bit(d, 0, &d->p_len_slot[1]);
        if (v == 0) { p = d->p_len_x2; lenbase = 8; }
        else { p = d->p_len_x3; lenbase = 16; }
    }
    if (lenbase == 16) {
        do { i = dec_bit(d, i, &p[i]); } while (i < 0x80);
        return lenbase + (i & 0x7F);
    }
    do { i = dec_bit(d, i, &p[i]); } while (i < 0x08);
    return lenbase + (i & 0x07);
}

static uint32_t decode_matchlen_2(OrcDec *d) /* csc_dec.cpp:222-234 */
{
    uint32_t len = decode_matchlen_1(d);
    if (len == 143) {
        for (;; len += 143) {
            if (dec_bit(d, 0, &d->p_longlen)) break;
        }
        return len + decode_matchlen_1(d);
    }
    return len;
}

static void decode_match(OrcDec *d, uint32_t *dist, uint32_t *len) /* csc_dec.cpp:236-283 */
{
    *len = decode_matchlen_2(d);
    uint32_t pdist_pos, sbits;
    switch (*len) {
    case 0: pdist_pos = 0; sbits = 3; break;
    case 1: case 2: pdist_pos = 16 * (*len - 1) + 8; sbits = 4; break;
    case 3: case 4: case 5: pdist_pos = 32 * (*len - 3) + 8 + 16 * 2; sbits = 5; break;
    default: pdist_pos = 32 * 3 + 8 + 16 * 2; sbits = 5; break;
    }
    uint32_t *p = d->p_dist + pdist_pos, i = 1;
    do { i = dec_bit(d, i, &p[i]); } while (i < (1u << sbits));
    uint32_t slot = i & ((1u << sbits) - 1);
    if (slot <= 2) *dist = slot;
    else {
        uint32_t ebits = slot - 2, elen = 0;
        if (ebits > 4) elen = dec_direct(d, ebits - 4);
        i = 1;
        p = &d->p_dist_extra[(ebits - 1) * 16];
        do { i = dec_bit(d, i, &p[i]); } while (i < 0x10);
        *dist = kDistTable[slot] + (elen << 4) + kRev16[i & 0x0F];
    }
    d->state = (d->state * 4 + 1) & 0x3F;
}

/* copy with the bounds checks of csc_dec.cpp:508-518,545-555 */
static void lz_copy_match(OrcDec *d, uint32_t dist, uint32_t len, uint32_t *i, uint32_t limit)
{
    uint32_t cpy_pos = d->wnd_curpos >= dist ? d->wnd_curpos - dist : d->wnd_curpos + d->wnd_size - dist;
    if (cpy_pos >= d->wnd_size || cpy_pos + len > d->wnd_size || len + *i > limit
        || d->wnd_curpos + len > d->wnd_size)
        longjmp(d->on_error, -DECODE_ERROR);
    uint8_t *dst = d->wnd + d->wnd_curpos, *src = d->wnd + cpy_pos;
    *i += len;
    d->wnd_curpos += len;
    while (len--) *dst++ = *src++;
    d->ctx = d->wnd[d->wnd_curpos - 1];
}

/* CSCDecoder::lz_decode, csc_dec.cpp:476-571 */
static int lz_decode(OrcDec *d, uint8_t *dst, uint32_t *size, uint32_t limit)
{
    uint32_t copied_size = 0, copied_wndpos = d->wnd_curpos, i;
    for (i = 0; i <= limit;) {
        if (dec_bit(d, 0, &d->p_state[d->state * 3 + 0]) == 0) {
            uint32_t c = decode_byte_tree(d, &d->p_lit[d->ctx * 256]); /* decode_literal :155-167 */
            d->ctx = c;
            d->state = (d->state * 4) & 0x3F;
            d->wnd[d->wnd_curpos++] = (uint8_t)c;
            i++;
        } else if (dec_bit(d, 0, &d->p_state[d->state * 3 + 1]) == 1) {
            uint32_t dist, len;
            decode_match(d, &dist, &len);
            if (len == 0 && dist == 64) break;
            dist++; len += 2;
            d->rep_dist[3] = d->rep_dist[2]; d->rep_dist[2] = d->rep_dist[1];
            d->rep_dist[1] = d->rep_dist[0]; d->rep_dist[0] = dist;
            lz_copy_match(d, dist, len, &i, limit);
        } else if (dec_bit(d, 0, &d->p_state[d->state * 3 + 2]) == 0) {
            d->state = (d->state * 4 + 2) & 0x3F; /* decode_1byte_match :289-293 */
            d->ctx = 0;
            uint32_t cpy_pos = d->wnd_curpos > d->rep_dist[0] ? d->wnd_curpos - d->rep_dist[0]
                                                               : d->wnd_curpos + d->wnd_size - d->rep_dist[0];
            d->wnd[d->wnd_curpos] = d->wnd[cpy_pos];
            d->wnd_curpos++;
            i++;
            d->ctx = d->wnd[d->wnd_curpos - 1];
        } else {
            uint32_t k = 1, len; /* decode_repdist_match :295-305 */
            do { k = dec_bit(d, k, &d->p_repdist[d->state * 3 + k - 1]); } while (k < 0x4);
            uint32_t idx = k & 0x3;
            len = decode_matchlen_2(d);
            d->state = (d->state * 4 + 3) & 0x3F;
            len += 2;
            if (len + i > limit) longjmp(d->on_error, -DECODE_ERROR);
            uint32_t dist = d->rep_dist[idx];
            for (uint32_t j = idx; j > 0; j--) d->rep_dist[j] = d->rep_dist[j - 1];
            d->rep_dist[0] = dist;
            lz_copy_match(d, dist, len, &i, limit);
        }
        if (d->wnd_curpos > d->wnd_size) longjmp(d->on_error, -DECODE_ERROR);
        else if (d->wnd_curpos == d->wnd_size) {
            d->wnd_curpos = 0;
            memcpy(dst + copied_size, d->wnd + copied_wndpos, i - copied_size);
            copied_wndpos = 0;
            copied_size = i;
        }
    }
    *size = i;
    memcpy(dst + copied_size, d->wnd + copied_wndpos, *size - copied_size);
    return 0;
}

static void lz_copy2dict(OrcDec *d, const uint8_t *src, uint32_t size) /* csc_dec.cpp:573-584 */
{
    for (uint32_t i = 0; i < size;) {
        uint32_t cur = UMIN(d->wnd_size - d->wnd_curpos, size - i);
        cur = UMIN(cur, MIN_BLOCK);
        memcpy(d->wnd + d->wnd_curpos, src + i, cur);
        d->wnd_curpos += cur;
        d->wnd_curpos = d->wnd_curpos >= d->wnd_size ? 0 : d->wnd_curpos;
        i += cur;
    }
}

static int decode_bad(OrcDec *d, uint8_t *dst, uint32_t *size, uint32_t max) /* csc_dec.cpp:98-108 */
{
    *size = decode_int(d);
    if (*size > max) return -1;
    for (uint32_t i = 0; i < *size; i++) dst[i] = (uint8_t)dec_direct16(d, 8);
    return 0;
}

static int decode_literals(OrcDec *d, uint8_t *dst, uint32_t *size, uint32_t max) /* csc_dec.cpp:169-185 */
{
    *size = decode_int(d);
    if (*size > max) return -1;
    for (uint32_t i = 0; i < *size; i++) {
        d->ctx = decode_byte_tree(d, &d->p_lit[d->ctx * 256]);
        dst[i] = (uint8_t)d->ctx;
    }
    return 0;
}

static int decode_rle(OrcDec *d, uint8_t *dst, uint32_t *size, uint32_t max) /* csc_dec.cpp:110-153 */
{
    uint32_t sctx = 0, i;
    if (d->p_delta == NULL) {
        d->p_delta = (uint32_t *)d->alloc->Alloc(d->alloc, 256 * 256 * sizeof(uint32_t));
        for (i = 0; i < 256 * 256; i++) d->p_delta[i] = 2048;
    }
    *size = decode_int(d);
    if (*size > max) return -1;
    for (i = 0; i < *size;) {
        if (dec_bit(d, 0, &d->p_rle_flag) == 0) {
            dst[i] = (uint8_t)decode_byte_tree(d, &d->p_delta[sctx * 256]);
            sctx = dst[i];
            i++;
        } else {
            uint32_t len = decode_matchlen_2(d) + 11;
            if (i == 0) return -1;
            while (len-- > 0 && i < *size) { dst[i] = dst[i - 1]; i++; }
            sctx = dst[i - 1];
        }
    }
    return 0;
}

/* ---- inverse filters ---- */
static void need_swap(OrcDec *d, uint32_t size)
{
    if (d->swap_size < size) {
        if (d->swap_size > 0) d->alloc->Free(d->alloc, d->swap_buf);
        d->swap_buf = (uint8_t *)d->alloc->Alloc(d->alloc, size);
        d->swap_size = size;
    }
}

static void dict_inverse(uint8_t *src, uint8_t *dst, uint32_t size) /* csc_filters.cpp:337-369 */
{
    uint32_t i = 0, j, dst_pos = 0;
    while (dst_pos < size) {
        if (src[i] >= 0x82 && src[i] < MAX_SYMBOL) {
            uint32_t idx = (uint32_t)src[i] - 0x82 + 1; /* wordIndex[sym] = i, csc_filters.cpp:107 */
            for (j = 0; kWords[idx][j] && dst_pos < size; j++) dst[dst_pos++] = (uint8_t)kWords[idx][j];
        } else if (src[i] == 254 && (i + 1 < size && src[i + 1] >= 0x82)) {
            i++;
            dst[dst_pos++] = src[i];
        } else {
            dst[dst_pos++] = src[i];
        }
        i++;
    }
    memcpy(src, dst, size);
}

static void delta_inverse(uint8_t *src, const uint8_t *copy, uint32_t size, uint32_t chn) /* csc_filters.cpp:371-399 */
{
    uint32_t dst_pos = 0, prev = 0;
    for (uint32_t i = 0; i < chn; i++)
        for (uint32_t j = i; j < size; j += chn) {
            src[j] = (uint8_t)(copy[dst_pos++] + prev);
            prev = src[j];
        }
}

/* E8/E9 inverse, csc_filters.cpp:508-524,533-537,560-586,600-610 */
static void e89_init(OrcDec *d) { d->ecs = 0xFF; d->x0 = d->x1 = 0; d->ei = 0; d->ek = 5; }
static int32_t e89_cache_byte(OrcDec *d, int32_t c)
{
    int32_t r = (d->ecs & 0x80) ? -1 : (int32_t)(uint8_t)d->x1;
    d->x1 >>= 8; d->x1 |= (d->x0 << 24);
    d->x0 >>= 8; d->x0 |= ((uint32_t)c << 24);
    d->ecs <<= 1; d->ei++;
    return r;
}
static uint32_t e89_yswap(uint32_t x)
{
    x = ((uint32_t)(uint8_t)(x >> 24) << 7) | ((uint32_t)(uint8_t)(x >> 16) << 8)
      | ((uint32_t)(uint8_t)(x >> 8) << 16) | (x << 24);
    return x >> 7;
}
static int32_t e89_inverse(OrcDec *d, int32_t c)
{
    if (d->ei >= d->ek) {
        if ((d->x1 & 0xFE000000u) == 0xE8000000u) {
            d->ek = d->ei + 4;
            uint32_t x = d->x0 - 0xFF000000u;
            if (x < 0x02000000u) {
                x = e89_yswap(x);
                x = (x - d->ei) & 0x01FFFFFFu;
                d->x0 = x + 0xFF000000u;
            }
        }
    }
    return e89_cache_byte(d, c);
}
static int32_t e89_flush(OrcDec *d)
{
    if (d->ecs != 0xFF) {
        while (d->ecs & 0x80) { e89_cache_byte(d, 0); ++d->ecs; }
        int32_t r = e89_cache_byte(d, 0); ++d->ecs;
        return r;
    }
    e89_init(d);
    return -1;
}
static void flt_inverse_e89(OrcDec *d, uint8_t *src, uint32_t size)
{
    uint32_t i, j;
    int32_t c;
    e89_init(d);
    for (i = 0, j = 0; i < size; i++) {
        c = e89_inverse(d, src[i]);
        if (c >= 0) src[j++] = (uint8_t)c;
    }
    while ((c = e89_flush(d)) >= 0) src[j++] = (uint8_t)c;
}

/* re-prime the arithmetic decoder from the next RC+BC blocks, csc_dec.cpp:336-345,657-680 */
static int dec_prime(OrcDec *d)
{
    d->rc_range = 0xFFFFFFFFu; d->rc_code = 0;
    d->rc_size = d->bc_size = 0; d->bc_curbits = d->bc_curval = 0;
    d->rc_pos = d->bc_pos = 0;
    if (read_block(d, d->rc_buf, &d->rc_bufsize, 1) < 0 || read_block(d, d->bc_buf, &d->bc_bufsize, 0) < 0)
        return -1;
    d->rc_code = ((uint32_t)d->rc_buf[1] << 24) | ((uint32_t)d->rc_buf[2] << 16)
               | ((uint32_t)d->rc_buf[3] << 8) | d->rc_buf[4];
    d->rc_pos += 5;
    d->rc_size += 5;
    return 0;
}

/* CSCDecoder::Decompress, csc_dec.cpp:586-682 */
static int dec_decompress(OrcDec *d, uint8_t *dst, uint32_t *size, uint32_t max)
{
    int ret = 0;
    uint32_t type = decode_int(d);
    switch (type) {
    case DT_NORMAL:
        ret = lz_decode(d, dst, size, max);
        break;
    case DT_EXE:
        ret = lz_decode(d, dst, size, max);
        flt_inverse_e89(d, dst, *size);
        break;
    case DT_ENGTXT:
        *size = decode_int(d);
        ret = lz_decode(d, dst, size, max);
        need_swap(d, *size);
        dict_inverse(dst, d->swap_buf, *size);
        break;
    case DT_BAD:
        ret = decode_bad(d, dst, size, max);
        if (ret < 0) return ret;
        lz_copy2dict(d, dst, *size);
        break;
    case DT_ENTROPY:
        ret = decode_literals(d, dst, size, max);
        if (ret < 0) return ret;
        lz_copy2dict(d, dst, *size);
        break;
    case SIG_EOF:
        *size = 0;
        break;
    default:
        if (type >= DT_DLT && type < DT_DLT + 5) {
            uint32_t chn = kDltIndex[type - DT_DLT];
            ret = decode_rle(d, dst, size, max);
            if (ret < 0) return ret;
            if (*size >= 512) {
                need_swap(d, *size);
                memcpy(d->swap_buf, dst, *size);
                delta_inverse(dst, d->swap_buf, *size, chn);
            }
            lz_copy2dict(d, dst, *size);
        } else {
            longjmp(d->on_error, -DECODE_ERROR);
        }
        break;
    }
    if (decode_int(d) == 1) {
        d->outsize += d->bc_size + d->rc_size;
        if (dec_prime(d) < 0) return -1;
    }
    return ret;
}

/* ---- public API, csc_dec.cpp:684-777 ---- */
void CSCDec_ReadProperties(CSCProps *props, uint8_t *s) /* csc_dec.cpp:733-738 */
{
    props->dict_size = ((uint32_t)s[0] << 24) + ((uint32_t)s[1] << 16) + ((uint32_t)s[2] << 8) + s[3];
    props->csc_blocksize = ((uint32_t)s[4] << 16) + ((uint32_t)s[5] << 8) + s[6];
    props->raw_blocksize = ((uint32_t)s[7] << 16) + ((uint32_t)s[8] << 8) + s[9];
}

void CSCDec_Destroy(CSCDecHandle h) /* csc_dec.cpp:722-731 + Destroy :389-405 + MemIO::Destroy */
{
    OrcDec *d = (OrcDec *)h;
    ISzAlloc *a = d->alloc;
    a->Free(a, d->p_lit); a->Free(a, d->p_delta); a->Free(a, d->wnd);
    a->Free(a, d->rc_buf); a->Free(a, d->bc_buf);
    if (d->swap_size > 0) a->Free(a, d->swap_buf);
    while (d->rc_blocks) { DataBlock *n = d->rc_blocks->next; a->Free(a, d->rc_blocks); d->rc_blocks = n; }
    while (d->bc_blocks) { DataBlock *n = d->bc_blocks->next; a->Free(a, d->bc_blocks); d->bc_blocks = n; }
    a->Free(a, d);
}

CSCDecHandle CSCDec_Create(const CSCProps *props, ISeqInStream *is, ISzAlloc *alloc) /* csc_dec.cpp:692-720 */
{
    if (alloc == NULL) alloc = &g_default_alloc;
    if (props->dict_size > 1024 * MB) return NULL;
    if (props->dict_size < 32 * KB) return NULL;
    OrcDec *d = (OrcDec *)alloc->Alloc(alloc, sizeof(OrcDec));
    if (!d) return NULL;
    memset(d, 0, sizeof(*d));
    d->alloc = alloc; d->is = is;
    d->bsize = props->csc_blocksize;
    d->raw_blocksize = props->raw_blocksize;
    /* CSCDecoder::Init, csc_dec.cpp:309-387 */
    d->rc_buf = (uint8_t *)alloc->Alloc(alloc, d->bsize);
    d->bc_buf = (uint8_t *)alloc->Alloc(alloc, d->bsize);
    if (!d->rc_buf || !d->bc_buf || dec_prime(d) < 0) { CSCDec_Destroy(d); return NULL; }
    d->p_lit = (uint32_t *)alloc->Alloc(alloc, 256 * 256 * sizeof(uint32_t));
    if (!d->p_lit) { CSCDec_Destroy(d); return NULL; }
#define FILL(P, K) do { for (int i_ = 0; i_ < (K); i_++) (P)[i_] = 2048; } while (0)
    FILL(d->p_state, 64 * 3); FILL(d->p_lit, 256 * 256); FILL(d->p_repdist, 64 * 3);
    FILL(d->p_dist, 8 + 16 * 2 + 32 * 4);
    FILL(d->p_len_slot, 2); FILL(d->p_len_x1, 8); FILL(d->p_len_x2, 8); FILL(d->p_len_x3, 128);
    FILL(d->p_dist_extra, 29 * 16);
#undef FILL
    d->p_longlen = 2048; d->p_rle_flag = 2048; d->state = 0; d->ctx = 0;
    d->wnd_size = (uint32_t)props->dict_size;
    d->wnd = (uint8_t *)alloc->Alloc(alloc, (size_t)d->wnd_size + 8);
    if (!d->wnd) { CSCDec_Destroy(d); return NULL; }
    d->wnd_curpos = 0;
    d->rep_dist[0] = d->rep_dist[1] = d->rep_dist[2] = d->rep_dist[3] = 0;
    return d;
}

static int dec_try_decompress(OrcDec *d, uint8_t *buf, uint32_t *size)
{
    int code = setjmp(d->on_error);
    if (code != 0) return -code;
    return dec_decompress(d, buf, size, d->raw_blocksize);
}

int CSCDec_Decode(CSCDecHandle h, ISeqOutStream *os, ICompressProgress *progress) /* csc_dec.cpp:740-777 */
{
    OrcDec *d = (OrcDec *)h;
    int ret = 0;
    uint8_t *buf = (uint8_t *)d->alloc->Alloc(d->alloc, d->raw_blocksize);
    uint64_t outsize = 0;
    for (;;) {
        /* `size` is read uninitialised by the reference when Decompress throws before
         * setting it (csc_dec.cpp:746-763); the oracle pins it to 0 in that case. */
        uint32_t size = 0;
        ret = dec_try_decompress(d, buf, &size);
        if (ret == 0) outsize += size;
        if (progress) progress->Progress(progress, (uint64_t)(d->outsize + d->rc_size + d->bc_size), outsize);
        if (size == 0 || ret < 0) break;
        size_t wrote = os->Write(os, buf, size);
        if (wrote == CSC_WRITE_ABORT) break;
        else if (wrote < size) { ret = WRITE_ERROR; break; }
    }
    d->alloc->Free(d->alloc, buf);
    return ret;
}

/* ---- probes ---- */
void orc_inverse_e89(uint8_t *buf, uint32_t size)
{
    OrcDec *d = (OrcDec *)calloc(1, sizeof(OrcDec));
    flt_inverse_e89(d, buf, size);
    free(d);
}
void orc_inverse_dict(uint8_t *buf, uint32_t size)
{
    uint8_t *tmp = (uint8_t *)malloc((size_t)size + 8);
    dict_inverse(buf, tmp, size);
    free(tmp);
}
void orc_inverse_delta(uint8_t *buf, uint32_t size, uint32_t chn)
{
    if (size < 512) return;
    uint8_t *tmp = (uint8_t *)malloc(size);
    memcpy(tmp, buf, size);
    delta_inverse(buf, tmp, size, chn);
    free(tmp);
}
