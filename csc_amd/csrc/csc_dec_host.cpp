// csc_dec_host.cpp -- CSCDec_* of libcsc_mi355x.so (reference: src/libcsc/csc_dec.cpp).
//
// Round-1 status (DESIGN.md "decode"): decoding one libcsc stream is a bit-serial chain with no
// bulk byte work to spread over lanes, and its input arrives through on-demand ISeqInStream
// reads that only the calling thread may issue (csc_memio.cpp:5-81).  This round serves the
// decode half of the boundary from the host thread that owns the callbacks; a device-resident
// decoder (one wavefront per task, host feeding RC/BC blocks on demand) is a section-8(f) "next" row.
// It is NOT a fallback for anything: there is no HIP decoder it could stand in for.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/csc_mi355x.h"

namespace {

constexpr uint32_t KB = 1024u, MB = 1048576u, MIN_BLOCK = 8u * KB;
enum : uint32_t { DT_NORMAL = 1, DT_ENGTXT = 2, DT_EXE = 3, DT_ENTROPY = 7, DT_BAD = 8, SIG_EOF = 9, DT_DLT = 0x10 };
const uint32_t kDltIndex[5] = {1, 2, 3, 4, 8};

const char *const kWords[122] = {   // csc_filters.cpp:8-38; symbol 0x82+i expands to kWords[i]
    "ac","ad","ai","al","am","an","ar","as","at","ea","ec","ed","ee","el","en","er","es","et","id","ie",
    "ig","il","in","io","is","it","of","ol","on","oo","or","os","ou","ow","ul","un","ur","us","ba","be",
    "ca","ce","co","ch","de","di","ge","gh","ha","he","hi","ho","ra","re","ri","ro","rs","la","le","li",
    "lo","ld","ll","ly","se","si","so","sh","ss","st","ma","me","mi","ne","nc","nd","ng","nt","pa","pe",
    "ta","te","ti","to","th","tr","wa","ve",
    "all","and","but","dow","for","had","hav","her","him","his","man","mor","not","now","one","out",
    "she","the","was","wer","whi","whe","wit","you","any","are",
    "that","said","with","have","this","from","were","tion",
};

void *def_alloc(void *, size_t n) { return malloc(n); }
void def_free(void *, void *a) { free(a); }
ISzAlloc g_default_alloc = {def_alloc, def_free};

struct Fail { int code; };   // internal only; caught inside CSCDec_Decode / CSCDec_Create

struct Block { Block *next; uint32_t size; uint8_t data[1]; };

struct Decoder {
    ISzAlloc *alloc;
    ISeqInStream *is;
    uint32_t bsize, raw_blocksize;
    Block *queue[2];                 // [1] = RC blocks read ahead, [0] = BC blocks (csc_memio.h:20-21)

    // arithmetic decoder
    uint8_t *buf[2];                 // [1] RC, [0] BC
    uint32_t fill[2], rd[2];         // bytes valid in buf / bytes consumed
    uint32_t range, code, bc_bits, bc_val;
    int64_t consumed;                // GetCompressedSize

    // model (csc_dec.cpp:434-456)
    uint32_t *p_lit, *p_delta;
    uint32_t p_state[192], p_repdist[256], p_dist[168], p_dist_extra[464];
    uint32_t p_len_slot[2], p_len_x1[8], p_len_x2[8], p_len_x3[128];
    uint32_t p_longlen, p_rle_flag, ctx, state;

    // dictionary
    uint32_t rep[4], wnd_size, wnd_pos;
    uint8_t *wnd, *swap;
    uint32_t swap_size;

    // ---- MemIO::ReadBlock, csc_memio.cpp:5-81 ----
    int read_block(int kind)
    {
        if (Block *b = queue[kind]) {
            fill[kind] = b->size;
            memcpy(buf[kind], b->data, b->size);
            queue[kind] = b->next;
            alloc->Free(alloc, b);
            return 0;
        }
        for (;;) {
            uint8_t fb, sb[3];
            size_t n = 1;
            is->Read(is, &fb, &n);
            if (n != 1) return -1;
            uint32_t cur = bsize;
            if (!((fb >> 6) & 1)) {
                n = 3;
                is->Read(is, sb, &n);
                if (n != 3) return -1;
                cur = ((uint32_t)sb[0] << 16) + ((uint32_t)sb[1] << 8) + sb[2];
            }
            if (!cur || cur > bsize) return -1;
            n = cur;
            if ((int)((fb >> 7) & 1) == kind) {
                fill[kind] = cur;
                is->Read(is, buf[kind], &n);
                return n == cur ? 0 : -1;
            }
            Block *nb = (Block *)alloc->Alloc(alloc, sizeof(Block) + cur);
            if (!nb) return -1;
            nb->size = cur; nb->next = nullptr;
            is->Read(is, nb->data, &n);
            if (n != cur) { alloc->Free(alloc, nb); return -1; }
            Block **tail = &queue[kind ^ 1];
            while (*tail) tail = &(*tail)->next;
            *tail = nb;
        }
    }

    inline uint32_t next_byte(int kind)
    {
        uint32_t b = buf[kind][rd[kind]++];
        if (rd[kind] >= fill[kind]) {           // refill as soon as the block is exhausted (csc_dec.cpp:14-21,70-76)
            consumed += rd[kind];
            if (read_block(kind) < 0) throw Fail{READ_ERROR};
            rd[kind] = 0;
        }
        return b;
    }

    inline uint32_t bit(uint32_t v, uint32_t &p)   // DecodeBit, csc_dec.cpp:10-35
    {
        if (range < (1u << 24)) { range <<= 8; code = (code << 8) + next_byte(1); }
        uint32_t bound = (range >> 12) * p;
        if (code < bound) { range = bound; p += (0xFFF - p) >> 5; return v + v + 1; }
        range -= bound; code -= bound; p -= p >> 5;
        return v + v;
    }
    uint32_t direct16(uint32_t len)                // coder_decode_direct, csc_dec.cpp:65-88
    {
        while (bc_bits < len) { bc_val = (bc_val << 8) | next_byte(0); bc_bits += 8; }
        uint32_t r = (bc_val >> (bc_bits - len)) & ((1u << len) - 1);
        bc_bits -= len;
        return r;
    }
    uint32_t direct(uint32_t l) { return l <= 16 ? direct16(l) : ((direct16(l - 16) << 16) | direct16(16)); }
    uint32_t get_int()                             // decode_int, csc_dec.cpp:90-97
    {
        uint32_t slot = direct(5), num = direct(slot == 0 ? 1 : slot);
        return slot ? num + (1u << slot) : num;
    }
    uint32_t byte_tree(uint32_t *row) { uint32_t c = 1; do { c = bit(c, row[c]); } while (c < 0x100); return c & 0xFF; }

    uint32_t matchlen_1()                          // csc_dec.cpp:187-220
    {
        uint32_t *p, base, i = 1;
        if (bit(0, p_len_slot[0]) == 0) { p = p_len_x1; base = 0; }
        else if (bit(0, p_len_slot[1]) == 0) { p = p_len_x2; base = 8; }
        else { p = p_len_x3; base = 16; }
        uint32_t top = base == 16 ? 0x80 : 0x08;
        do { i = bit(i, p[i]); } while (i < top);
        return base + (i & (top - 1));
    }
    uint32_t matchlen_2()                          // csc_dec.cpp:222-234
    {
        uint32_t len = matchlen_1();
        if (len != 143) return len;
        while (!bit(0, p_longlen)) len += 143;
        return len + matchlen_1();
    }
    void match(uint32_t &dist, uint32_t &len)      // decode_match, csc_dec.cpp:236-283
    {
        static const uint32_t rev16[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
        len = matchlen_2();
        uint32_t pos, sbits;
        if (len == 0) { pos = 0; sbits = 3; }
        else if (len <= 2) { pos = 16 * (len - 1) + 8; sbits = 4; }
        else if (len <= 5) { pos = 32 * (len - 3) + 8 + 32; sbits = 5; }
        else { pos = 32 * 3 + 8 + 32; sbits = 5; }
        uint32_t *p = p_dist + pos, i = 1;
        do { i = bit(i, p[i]); } while (i < (1u << sbits));
        uint32_t slot = i & ((1u << sbits) - 1);
        if (slot <= 2) dist = slot;
        else {
            uint32_t ebits = slot - 2, elen = ebits > 4 ? direct(ebits - 4) : 0;
            i = 1;
            p = &p_dist_extra[(ebits - 1) * 16];
            do { i = bit(i, p[i]); } while (i < 0x10);
            dist = ((1u << ebits) + 1) + (elen << 4) + rev16[i & 0x0F];   // dist_table_[slot] = 2^(slot-2)+1
        }
        state = (state * 4 + 1) & 0x3F;
    }

    void copy_match(uint32_t dist, uint32_t len, uint32_t &i, uint32_t limit)   // csc_dec.cpp:506-518,543-555
    {
        uint32_t from = wnd_pos >= dist ? wnd_pos - dist : wnd_pos + wnd_size - dist;
        if (from >= wnd_size || from + len > wnd_size || len + i > limit || wnd_pos + len > wnd_size)
            throw Fail{DECODE_ERROR};
        uint8_t *d = wnd + wnd_pos, *s = wnd + from;
        i += len;
        wnd_pos += len;
        while (len--) *d++ = *s++;
        ctx = wnd[wnd_pos - 1];
    }

    void lz_decode(uint8_t *dst, uint32_t *size, uint32_t limit)   // csc_dec.cpp:476-571
    {
        uint32_t copied = 0, copied_from = wnd_pos, i;
        for (i = 0; i <= limit;) {
            if (bit(0, p_state[state * 3]) == 0) {
                uint32_t c = byte_tree(&p_lit[ctx * 256]);
                ctx = c;
                state = (state * 4) & 0x3F;
                wnd[wnd_pos++] = (uint8_t)c;
                i++;
            } else if (bit(0, p_state[state * 3 + 1]) == 1) {
                uint32_t dist, len;
                match(dist, len);
                if (len == 0 && dist == 64) break;
                dist++; len += 2;
                rep[3] = rep[2]; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = dist;
                copy_match(dist, len, i, limit);
            } else if (bit(0, p_state[state * 3 + 2]) == 0) {
                state = (state * 4 + 2) & 0x3F;
                uint32_t from = wnd_pos > rep[0] ? wnd_pos - rep[0] : wnd_pos + wnd_size - rep[0];
                if (from > wnd_size) throw Fail{DECODE_ERROR};   // the reference reads out of bounds here on bad input
                wnd[wnd_pos] = wnd[from];
                wnd_pos++;
                i++;
                ctx = wnd[wnd_pos - 1];
            } else {
                uint32_t k = 1;
                do { k = bit(k, p_repdist[state * 3 + k - 1]); } while (k < 4);
                uint32_t idx = k & 3, len = matchlen_2() + 2;
                state = (state * 4 + 3) & 0x3F;
                if (len + i > limit) throw Fail{DECODE_ERROR};
                uint32_t dist = rep[idx];
                for (uint32_t j = idx; j > 0; j--) rep[j] = rep[j - 1];
                rep[0] = dist;
                copy_match(dist, len, i, limit);
            }
            if (wnd_pos > wnd_size) throw Fail{DECODE_ERROR};
            if (wnd_pos == wnd_size) {
                wnd_pos = 0;
                memcpy(dst + copied, wnd + copied_from, i - copied);
                copied_from = 0;
                copied = i;
            }
        }
        *size = i;
        memcpy(dst + copied, wnd + copied_from, *size - copied);
    }

    void to_dict(const uint8_t *src, uint32_t size)   // lz_copy2dict, csc_dec.cpp:573-584
    {
        for (uint32_t i = 0; i < size;) {
            uint32_t cur = wnd_size - wnd_pos < size - i ? wnd_size - wnd_pos : size - i;
            if (cur > MIN_BLOCK) cur = MIN_BLOCK;
            memcpy(wnd + wnd_pos, src + i, cur);
            wnd_pos += cur;
            if (wnd_pos >= wnd_size) wnd_pos = 0;
            i += cur;
        }
    }

    uint8_t *scratch(uint32_t size)
    {
        if (swap_size < size) {
            if (swap_size) alloc->Free(alloc, swap);
            swap = (uint8_t *)alloc->Alloc(alloc, size);
            swap_size = size;
        }
        return swap;
    }

    // ---- inverse filters ----
    void inverse_dict(uint8_t *src, uint32_t size)    // csc_filters.cpp:337-369
    {
        uint8_t *dst = scratch(size);
        uint32_t i = 0, o = 0;
        while (o < size) {
            uint8_t b = src[i];
            if (b >= 0x82 && b < 0x82 + 122) {
                for (const char *w = kWords[b - 0x82]; *w && o < size; w++) dst[o++] = (uint8_t)*w;
            } else if (b == 254 && i + 1 < size && src[i + 1] >= 0x82) {
                dst[o++] = src[++i];
            } else {
                dst[o++] = b;
            }
            i++;
        }
        memcpy(src, dst, size);
    }
    void inverse_delta(uint8_t *src, uint32_t size, uint32_t chn)   // csc_filters.cpp:371-399
    {
        if (size < 512) return;
        uint8_t *copy = scratch(size);
        memcpy(copy, src, size);
        uint32_t o = 0, prev = 0;
        for (uint32_t c = 0; c < chn; c++)
            for (uint32_t j = c; j < size; j += chn) { src[j] = (uint8_t)(copy[o++] + prev); prev = src[j]; }
    }
    // Inverse_E89, csc_filters.cpp:533-537,560-575,600-610 -- in the unrolled form of the 8-byte delay
    // line (see flt_forward_e89 in csc_kernels_blocks.inc): operands are rewritten in place.
    static void inverse_e89(uint8_t *b, uint32_t size)
    {
        for (uint32_t j = 0; j + 5 < size;) {
            if ((b[j] & 0xFE) != 0xE8) { j++; continue; }
            uint32_t x0 = (uint32_t)b[j + 1] | ((uint32_t)b[j + 2] << 8) | ((uint32_t)b[j + 3] << 16) | ((uint32_t)b[j + 4] << 24);
            uint32_t x = x0 - 0xFF000000u;
            if (x < 0x02000000u) {
                x = ((x >> 24) << 7) | (((x >> 16) & 0xFF) << 8) | (((x >> 8) & 0xFF) << 16) | (x << 24);   // E89yswap
                x >>= 7;
                x = (x - (j + 5)) & 0x01FFFFFFu;
                x += 0xFF000000u;
                b[j + 1] = (uint8_t)x; b[j + 2] = (uint8_t)(x >> 8); b[j + 3] = (uint8_t)(x >> 16); b[j + 4] = (uint8_t)(x >> 24);
            }
            j += 4;
        }
    }

    int prime()   // csc_dec.cpp:336-345, 657-680
    {
        range = 0xFFFFFFFFu; code = 0; bc_bits = bc_val = 0;
        rd[0] = rd[1] = 0;
        if (read_block(1) < 0 || read_block(0) < 0) return -1;
        code = ((uint32_t)buf[1][1] << 24) | ((uint32_t)buf[1][2] << 16) | ((uint32_t)buf[1][3] << 8) | buf[1][4];
        rd[1] = 5;
        return 0;
    }

    // CSCDecoder::Decompress, csc_dec.cpp:586-682
    int decompress(uint8_t *dst, uint32_t *size, uint32_t max)
    {
        uint32_t type = get_int();
        switch (type) {
        case DT_NORMAL: lz_decode(dst, size, max); break;
        case DT_EXE: lz_decode(dst, size, max); inverse_e89(dst, *size); break;
        case DT_ENGTXT: *size = get_int(); lz_decode(dst, size, max); inverse_dict(dst, *size); break;
        case DT_BAD:
            *size = get_int();
            if (*size > max) return -1;
            for (uint32_t i = 0; i < *size; i++) dst[i] = (uint8_t)direct16(8);
            to_dict(dst, *size);
            break;
        case DT_ENTROPY:
            *size = get_int();
            if (*size > max) return -1;
            for (uint32_t i = 0; i < *size; i++) { ctx = byte_tree(&p_lit[ctx * 256]); dst[i] = (uint8_t)ctx; }
            to_dict(dst, *size);
            break;
        case SIG_EOF: *size = 0; break;
        default: {
            if (!(type >= DT_DLT && type < DT_DLT + 5)) throw Fail{DECODE_ERROR};
            if (!p_delta) {   // decode_rle, csc_dec.cpp:110-153
                p_delta = (uint32_t *)alloc->Alloc(alloc, 256 * 256 * sizeof(uint32_t));
                for (uint32_t i = 0; i < 256 * 256; i++) p_delta[i] = 2048;
            }
            *size = get_int();
            if (*size > max) return -1;
            uint32_t sctx = 0;
            for (uint32_t i = 0; i < *size;) {
                if (bit(0, p_rle_flag) == 0) {
                    dst[i] = (uint8_t)byte_tree(&p_delta[sctx * 256]);
                    sctx = dst[i++];
                } else {
                    uint32_t len = matchlen_2() + 11;
                    if (i == 0) return -1;
                    while (len-- > 0 && i < *size) { dst[i] = dst[i - 1]; i++; }
                    sctx = dst[i - 1];
                }
            }
            inverse_delta(dst, *size, kDltIndex[type - DT_DLT]);
            to_dict(dst, *size);
        } break;
        }
        if (get_int() == 1) {
            consumed += rd[0] + rd[1];
            if (prime() < 0) return -1;
        }
        return 0;
    }
};

struct DecInstance {
    uint32_t magic;
    ISzAlloc *alloc;
    Decoder *d;
    uint8_t *out;
};
constexpr uint32_t kMagicDec = 0x43534344;   // "CSCD"

void destroy(DecInstance *x)
{
    ISzAlloc *a = x->alloc;
    if (Decoder *d = x->d) {
        a->Free(a, d->p_lit); a->Free(a, d->p_delta); a->Free(a, d->wnd);
        a->Free(a, d->buf[0]); a->Free(a, d->buf[1]);
        if (d->swap_size) a->Free(a, d->swap);
        for (int k = 0; k < 2; k++)
            while (d->queue[k]) { Block *n = d->queue[k]->next; a->Free(a, d->queue[k]); d->queue[k] = n; }
        a->Free(a, d);
    }
    a->Free(a, x->out);
    x->magic = 0;
    a->Free(a, x);
}

}  // namespace

extern "C" {

void CSCDec_ReadProperties(CSCProps *props, uint8_t *s)   // csc_dec.cpp:733-738
{
    props->dict_size = ((uint32_t)s[0] << 24) + ((uint32_t)s[1] << 16) + ((uint32_t)s[2] << 8) + s[3];
    props->csc_blocksize = ((uint32_t)s[4] << 16) + ((uint32_t)s[5] << 8) + s[6];
    props->raw_blocksize = ((uint32_t)s[7] << 16) + ((uint32_t)s[8] << 8) + s[9];
}

CSCDecHandle CSCDec_Create(const CSCProps *props, ISeqInStream *instream, ISzAlloc *alloc)   // csc_dec.cpp:692-720
{
    if (alloc == NULL) alloc = &g_default_alloc;
    if (props->dict_size > 1024 * MB || props->dict_size < 32 * KB) return NULL;
    if (props->csc_blocksize == 0 || props->raw_blocksize == 0) return NULL;
    DecInstance *x = (DecInstance *)alloc->Alloc(alloc, sizeof(DecInstance));
    if (!x) return NULL;
    memset(x, 0, sizeof(*x));
    x->magic = kMagicDec; x->alloc = alloc;
    Decoder *d = x->d = (Decoder *)alloc->Alloc(alloc, sizeof(Decoder));
    if (!d) { destroy(x); return NULL; }
    memset(d, 0, sizeof(*d));
    d->alloc = alloc; d->is = instream;
    d->bsize = props->csc_blocksize; d->raw_blocksize = props->raw_blocksize;
    d->buf[0] = (uint8_t *)alloc->Alloc(alloc, d->bsize);
    d->buf[1] = (uint8_t *)alloc->Alloc(alloc, d->bsize);
    x->out = (uint8_t *)alloc->Alloc(alloc, d->raw_blocksize);
    if (!d->buf[0] || !d->buf[1] || !x->out || d->prime() < 0) { destroy(x); return NULL; }   // Init reads already (:336-345)
    d->p_lit = (uint32_t *)alloc->Alloc(alloc, 256 * 256 * sizeof(uint32_t));
    d->wnd_size = (uint32_t)props->dict_size;
    d->wnd = (uint8_t *)alloc->Alloc(alloc, (size_t)d->wnd_size + 8);
    if (!d->p_lit || !d->wnd) { destroy(x); return NULL; }
    auto fill = [](uint32_t *p, int n) { for (int i = 0; i < n; i++) p[i] = 2048; };
    fill(d->p_state, 192); fill(d->p_lit, 256 * 256); fill(d->p_repdist, 192); fill(d->p_dist, 168);
    fill(d->p_len_slot, 2); fill(d->p_len_x1, 8); fill(d->p_len_x2, 8); fill(d->p_len_x3, 128);
    fill(d->p_dist_extra, 464);
    d->p_longlen = d->p_rle_flag = 2048;
    return (CSCDecHandle)x;
}

void CSCDec_Destroy(CSCDecHandle p)   // csc_dec.cpp:722-731
{
    DecInstance *x = (DecInstance *)p;
    if (x && x->magic == kMagicDec) destroy(x);
}

int CSCDec_Decode(CSCDecHandle p, ISeqOutStream *os, ICompressProgress *progress)   // csc_dec.cpp:740-777
{
    DecInstance *x = (DecInstance *)p;
    Decoder *d = x->d;
    int ret = 0;
    uint64_t outsize = 0;
    for (;;) {
        uint32_t size = 0;
        try {
            ret = d->decompress(x->out, &size, d->raw_blocksize);
        } catch (const Fail &f) {
            ret = f.code;
        } catch (...) {
            ret = -1;
        }
        if (ret == 0) outsize += size;
        if (progress) progress->Progress(progress, (uint64_t)(d->consumed + d->rd[0] + d->rd[1]), outsize);
        if (size == 0 || ret < 0) break;
        size_t wrote = os->Write(os, x->out, size);
        if (wrote == CSC_WRITE_ABORT) break;
        if (wrote < size) { ret = WRITE_ERROR; break; }
    }
    return ret;
}

}  // extern "C"
