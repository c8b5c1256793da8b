/*
 * tests/model/hp_model.c -- CPU model of the ARRANGEMENT the HIP kernels give the hash-bucket match finder under the lazy parser
 * (levels 1 / 2: csc_lz.cpp:156-199 over csc_mf.cpp:243-495 with ht_width <= 8; csc_amd/csrc/csc_kernels_hp.inc).  TEST
 * INFRASTRUCTURE: it includes the oracle's encoder, replaces only compress_normal through the oracle's test hook, and must produce
 * the oracle's bytes (tests/test_hp_model.py).  What it proves before any of it runs on a GPU:
 *
 *  1. SPECULATIVE INSERTER.  Every position of a sub-block is inserted ahead of the parser with find_match's rule (HT2 / HT3
 *     overwrite, bucket shifted by one with the position in front; csc_mf.cpp:365-366,487-491), 64 positions per batch: all table
 *     words gathered first, positions of a batch with the same key resolved from the batch itself, the last position of a key
 *     stores.  The parser's SlidePos (:134-206) inserts the same words EXCEPT (a) the long-match skip (:145) and (b) the same-hash
 *     no-shift rule (:150-156); a slid range that contains neither needs nothing, one that does is taken back (the records hold
 *     every word as the position found it) and replayed exactly.
 *  2. RECORDS.  What a position's hash candidates are -- distances as the position found the tables, which of them find_match
 *     would look at (strictly increasing distance, :301-363,453-485), their match lengths capped at good_len -- depends on
 *     neither the parse nor the model: the inserter leaves it per position; the parser compares the four rep distances, runs the
 *     acceptance over rep + record slots and extends a capped length against the window on demand.
 */
#include <stdio.h>
#include "../../oracle/orc_encoder.c"

#define HP_R 256
#define HP_W 8            /* bucket slots a record can hold */

typedef struct {
    uint32_t d2, d3, bd[HP_W];             /* distances as the position found the tables */
    uint32_t l2, l3, bl[HP_W];             /* match lengths capped at min(good_len, climit); 0 where not looked at */
    uint32_t vm;                           /* which ones find_match looks at: bit 0 HT2, 1 HT3, 2.. bucket */
    uint8_t hflag;                         /* SlidePos would not shift the bucket here (h6 == 0 or == h6 of the position before) */
} HpRec;

static struct {
    HpRec rec[HP_R];
    uint32_t sb0, pos0, size, head, carry_h6;
    int la;
    unsigned long long n_batches, n_find, n_slides, n_dev, n_undo_pos, n_exact_pos, n_noevent, n_extend, n_pipe_sb, n_fallback_sb, n_samekey, n_len129;
} H;

static void hpm_die(const char *what) { fprintf(stderr, "hp_model: %s\n", what); abort(); }

static uint32_t hpm_prefix_from(const uint8_t *a, const uint8_t *b, uint32_t start, uint32_t lim)
{
    uint32_t n = start;
    while (n < lim && a[n] == b[n]) n++;
    return n > lim ? lim : n;
}

/* ---- the inserter: positions [i0, i0 + n) of the open sub-block, n <= 64, find_match's rule ---- */
static void hpm_batch(OrcEnc *e, uint32_t i0, uint32_t n)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    uint32_t h2[64], h3[64], h6[64], o2[64], o3[64], ob[64][HP_W];
    H.n_batches++;
    for (uint32_t k = 0; k < n; k++) {             /* ONE gather, before any store of this batch */
        const uint8_t *p = e->wnd + H.sb0 + i0 + k;
        h2[k] = hash2(p); h3[k] = hash3(p); h6[k] = hash6(p, e->ht_bits);
        o2[k] = e->ht2[h2[k]]; o3[k] = e->ht3[h3[k]];
        for (uint32_t t = 0; t < W; t++) ob[k][t] = e->ht6[(size_t)h6[k] * e->ht_width + t];
    }
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t pos = H.pos0 + i0 + k, wpos = H.sb0 + i0 + k, limit = H.size - (i0 + k);
        HpRec *R = &H.rec[(i0 + k) % HP_R];
        uint32_t e2 = o2[k], e3 = o3[k], eb[HP_W];
        int later2 = 0, later3 = 0, later6 = 0;
        for (uint32_t t = 0; t < W; t++) eb[t] = ob[k][t];
        for (uint32_t j = 0; j < n; j++) {
            if (j < k) {
                if (h2[j] == h2[k]) e2 = H.pos0 + i0 + j;
                if (h3[j] == h3[k]) e3 = H.pos0 + i0 + j;
                if (h6[j] == h6[k]) { for (uint32_t t = W - 1; t > 0; t--) eb[t] = eb[t - 1]; eb[0] = H.pos0 + i0 + j; H.n_samekey++; }
            } else if (j > k) { later2 |= h2[j] == h2[k]; later3 |= h3[j] == h3[k]; later6 |= h6[j] == h6[k]; }
        }
        if (!later2) e->ht2[h2[k]] = pos;
        if (!later3) e->ht3[h3[k]] = pos;
        if (!later6) {
            uint32_t *b = e->ht6 + (size_t)h6[k] * e->ht_width;
            for (uint32_t t = W - 1; t > 0; t--) b[t] = eb[t - 1];
            b[0] = pos;
        }
        R->d2 = pos - e2; R->d3 = pos - e3;
        for (uint32_t t = 0; t < W; t++) R->bd[t] = pos - eb[t];
        R->hflag = (h6[k] == 0 || h6[k] == (k ? h6[k - 1] : H.carry_h6)) ? 1 : 0;
        /* which candidates find_match looks at (strictly increasing distance), and their capped lengths */
        const uint8_t *pcur = e->wnd + wpos;
        uint32_t M = 0;
        R->vm = 0; R->l2 = R->l3 = 0;
        if (R->d2 > M) { M = R->d2; if (R->d2 < e->vld_rge) R->vm |= 1u; }
        if (R->d3 > M) { M = R->d3; if (R->d3 < e->vld_rge) R->vm |= 2u; }
        for (uint32_t t = 0; t < W; t++) { R->bl[t] = 0; if (R->bd[t] <= M) continue; M = R->bd[t]; if (R->bd[t] < e->vld_rge) R->vm |= 4u << t; }
        if (R->vm & 1u) {
            uint32_t cp = wpos > R->d2 ? wpos - R->d2 : wpos + e->wnd_size - R->d2;       /* strict, csc_mf.cpp:306 */
            R->l2 = hpm_prefix_from(pcur, e->wnd + cp, 0, UMIN(UMIN(limit, e->wnd_size - cp), e->good_len));
        }
        if (R->vm & 2u) {
            uint32_t cp = wrap_back(e, wpos, R->d3);
            R->l3 = hpm_prefix_from(pcur, e->wnd + cp, 0, UMIN(UMIN(limit, e->wnd_size - cp), e->good_len));
        }
        for (uint32_t t = 0; t < W; t++) if (R->vm & (4u << t)) {
            uint32_t cp = wrap_back(e, wpos, R->bd[t]);
            R->bl[t] = hpm_prefix_from(pcur, e->wnd + cp, 0, UMIN(UMIN(limit, e->wnd_size - cp), e->good_len));
        }
    }
    H.carry_h6 = h6[n - 1];
}

/* take back the speculative inserts of positions [a, head), newest first: every word as the position found it */
static void hpm_undo(OrcEnc *e, uint32_t a)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    for (uint32_t p = H.head; p-- > a;) {
        const HpRec *R = &H.rec[p % HP_R];
        const uint8_t *q = e->wnd + H.sb0 + p;
        const uint32_t pos = H.pos0 + p;
        uint32_t *b = e->ht6 + (size_t)hash6(q, e->ht_bits) * e->ht_width;
        for (uint32_t t = 0; t < W; t++) b[t] = pos - R->bd[t];
        e->ht2[hash2(q)] = pos - R->d2;
        e->ht3[hash3(q)] = pos - R->d3;
        H.n_undo_pos++;
    }
    H.head = a;
}

/* MatchFinder::SlidePos (csc_mf.cpp:134-206, hash tables only) over positions s + start_i .. s + len - 1 of the sub-block, resumed
 * inside a call: lasth6 = the hash of the last position the call has inserted (0 at its start) */
static void hpm_slide_exact(OrcEnc *e, uint32_t s, uint32_t len, uint32_t start_i, uint32_t lasth6)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    for (uint32_t i = start_i; i < len;) {
        const uint8_t *q = e->wnd + H.sb0 + s + i;
        const uint32_t pos = H.pos0 + s + i;
        e->ht2[hash2(q)] = pos;
        e->ht3[hash3(q)] = pos;
        if (i + 128 < len) { i += 4; continue; }
        uint32_t h6 = hash6(q, e->ht_bits);
        uint32_t *b = e->ht6 + (size_t)h6 * e->ht_width;
        if (h6 != lasth6) for (uint32_t j = W - 1; j > 0; j--) b[j] = b[j - 1];
        b[0] = pos;
        lasth6 = h6;
        i++;
        H.n_exact_pos++;
    }
}

static void hpm_advance(OrcEnc *e, uint32_t upto)
{
    if (upto > H.size) upto = H.size;
    while (H.head < upto) {
        uint32_t n = UMIN(64u, H.size - H.head);
        if (H.head == 0) H.carry_h6 = 0xFFFFFFFFu;
        hpm_batch(e, H.head, n);
        H.head += n;
    }
}

/* the parser coded `len` bytes at sub-block offset s and calls SlidePos(s, len): positions s + 1 .. s + len - 1 */
static void hpm_slide(OrcEnc *e, uint32_t s, uint32_t len)
{
    if (len <= 1) return;
    H.n_slides++;
    const uint32_t a = s + 1, b = s + len;
    if (len > 129) H.n_len129++;
    if (H.head <= a) {                    /* nothing speculated in the range: SlidePos itself */
        if (H.head < a) hpm_die("inserter behind a match start");
        hpm_slide_exact(e, s, len, 1, 0);
        H.head = b; H.carry_h6 = 0xFFFFFFFFu;
        return;
    }
    const uint32_t hi = UMIN(H.head, b);
    int dev = len > 129;
    for (uint32_t q = a; q < hi && !dev; q++) dev = H.rec[q % HP_R].hflag;
    if (!dev) {
        if (hi < b) {                     /* the rest of the range: SlidePos, continuing behind the last position inserted */
            hpm_slide_exact(e, s, len, hi - s, hash6(e->wnd + H.sb0 + hi - 1, e->ht_bits));
            H.head = b; H.carry_h6 = 0xFFFFFFFFu;
        } else H.n_noevent++;             /* the speculative inserts ARE SlidePos's */
        return;
    }
    H.n_dev++;
    hpm_undo(e, a);
    hpm_slide_exact(e, s, len, 1, 0);
    H.head = b; H.carry_h6 = 0xFFFFFFFFu;
}

/* find_match (csc_mf.cpp:243-495, hash tables only) over the record of sub-block offset p */
static uint32_t hpm_find_match(OrcEnc *e, MFUnit *ret, const uint32_t *rep_dist, uint32_t p)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    const uint32_t wpos = H.sb0 + p, limit = H.size - p;
    const uint8_t *pcur = e->wnd + wpos;
    uint32_t minlen = 1, cnt = 0, dist = 0;
    H.n_find++;
    hpm_advance(e, p + 1 + (uint32_t)H.la);
    if (H.head <= p) hpm_die("record not there");
    const HpRec *R = &H.rec[p % HP_R];
#define PUSH_CAND(L, D) do { ret[cnt].len = (L); ret[cnt].dist = (D); if (cnt + 2 < MF_CAND_LIMIT) cnt++; } while (0)
    for (uint32_t i = 0; i < 4; i++) {
        if (rep_dist[i] >= e->vld_rge) continue;
        uint32_t cmp_pos = wrap_back(e, wpos, rep_dist[i]);
        uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
        uint32_t match_len = prefix_len(pcur, e->wnd + cmp_pos, climit);
        if (i == 0 && match_len >= 2) PUSH_CAND(1, 1);
        if (match_len > minlen) {
            minlen = match_len;
            PUSH_CAND(match_len, 1 + i);
            if (match_len >= e->good_len) { dist = 0xFFFFFFFFu; break; }
        }
    }
    /* the record's slots in find_match's order; `dist` only matters as "everything behind a good_len hit is not looked at" */
    for (uint32_t t = 0; t < 2 + W && dist != 0xFFFFFFFFu; t++) {
        if (!(R->vm & (1u << t))) continue;
        const uint32_t d = t == 0 ? R->d2 : t == 1 ? R->d3 : R->bd[t - 2];
        uint32_t ml = t == 0 ? R->l2 : t == 1 ? R->l3 : R->bl[t - 2];
        if (ml >= e->good_len) {             /* capped: the real length */
            uint32_t cp = t == 0 ? (wpos > d ? wpos - d : wpos + e->wnd_size - d) : wrap_back(e, wpos, d);
            ml = hpm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
            H.n_extend++;
        }
        if (ml > minlen) {
            minlen = ml;
            if (ml <= 6 && d >= kBound[ml]) continue;
            PUSH_CAND(ml, 4 + d);
            if (ml >= e->good_len) dist = 0xFFFFFFFFu;
        }
    }
#undef PUSH_CAND
    return cnt;
}

static MFUnit hpm_find_best(OrcEnc *e, const uint32_t *rep_dist, uint32_t p)
{
    e->mfcand[0].len = 1; e->mfcand[0].dist = 0;
    uint32_t n = hpm_find_match(e, e->mfcand + 1, rep_dist, p);
    uint32_t best = 0;
    for (uint32_t i = 1; i <= n; i++) {
        if (!best) { best = i; continue; }
        if (second_better(e->mfcand[best], e->mfcand[i])) best = i;
    }
    return e->mfcand[best];
}

/* LZ::compress_normal (csc_lz.cpp:156-199) over records; i = sub-block offset of wnd_curpos */
static void hpm_norm_pipe(OrcEnc *e, uint32_t size, int lazy)
{
    MFUnit u1 = {0, 0}, u2;
    int got_u1 = 0;
    H.sb0 = e->wnd_curpos; H.pos0 = e->pos; H.size = size; H.head = 0; H.carry_h6 = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < size;) {
        if (!got_u1) u1 = hpm_find_best(e, e->rep_dist, i);
        if (u1.len == 1 || !lazy || u1.len >= e->lz_good_len) {
            if (u1.dist == 0) encode_literal(e, e->wnd[e->wnd_curpos]);
            else lz_encode_nonlit(e, u1);
            hpm_slide(e, i, u1.len);
            i += u1.len; e->wnd_curpos += u1.len;
            if (u1.dist) e->ctx = e->wnd[e->wnd_curpos - 1];
            got_u1 = 0;
            continue;
        }
        u2 = hpm_find_best(e, e->rep_dist, i + 1);
        if (second_better(u1, u2)) {
            encode_literal(e, e->wnd[e->wnd_curpos]);
            i++; e->wnd_curpos++;
            u1 = u2;
            got_u1 = 1;
        } else {
            lz_encode_nonlit(e, u1);
            hpm_slide(e, i + 1, u1.len - 1);
            i += u1.len; e->wnd_curpos += u1.len;
            e->ctx = e->wnd[e->wnd_curpos - 1];
            got_u1 = 0;
        }
    }
    hpm_advance(e, size);               /* (a literal at the very end: nothing behind it was asked for) */
    if (H.head != size) hpm_die("inserter did not finish the sub-block");
    e->pos = H.pos0 + size;
}

static void hpm_norm(OrcEnc *e, uint32_t size, int lazy)
{
    const int ok = !e->bt_head && e->ht_width && UMIN(e->ht_width, e->ht_cyc) <= HP_W && e->ht_low && e->good_len <= 255 && e->pos < 0xFFFF0000u;
    if (!ok) { H.n_fallback_sb++; lz_compress_normal(e, size, lazy); return; }
    H.n_pipe_sb++;
    hpm_norm_pipe(e, size, lazy);
}

__attribute__((destructor)) static void hpm_stats(void)
{
    if (!getenv("HPM_STATS")) return;
    fprintf(stderr, "hp_model: sub-blocks pipe %llu fallback %llu, batches %llu, finds %llu, slides %llu (longer than 129: %llu), no event needed %llu, deviations %llu, "
            "positions undone %llu, exact slide positions %llu, same-key pairs %llu, extensions %llu\n",
            H.n_pipe_sb, H.n_fallback_sb, H.n_batches, H.n_find, H.n_slides, H.n_len129, H.n_noevent, H.n_dev, H.n_undo_pos, H.n_exact_pos, H.n_samekey, H.n_extend);
}
__attribute__((constructor)) static void hpm_install(void)
{
    const char *s;
    H.la = (s = getenv("HPM_LA")) ? atoi(s) : 64;
    orc_norm_hook = hpm_norm;
}
