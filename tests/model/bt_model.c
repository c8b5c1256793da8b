/*
 * tests/model/bt_model.c -- CPU model of the ARRANGEMENT the HIP kernels give the binary-tree match finder (level 5,
 * csc_mf.cpp:159-203,368-451) under LZ::compress_advanced (csc_amd/csrc/csc_kernels_bt.inc).  TEST INFRASTRUCTURE: it includes
 * the oracle's encoder, replaces only compress_advanced through the oracle's test hook, and must produce the oracle's bytes
 * (tests/test_bt_model.py).  What it proves before any of it runs on a GPU:
 *
 *  1. THE TREE INSERT OF A POSITION DOES NOT DEPEND ON THE PARSE.  find_match's descent (csc_mf.cpp:404-451) and SlidePos's
 *     (:159-203) leave the same tree: the compare cap (good_len vs limit) only changes what is REPORTED, never where the descent
 *     goes or what it stores.  So an INSERTER runs ahead of the parser with SlidePos's rule (compare capped at good_len) and
 *     writes, per position, a RECORD: the HT2 / HT3 / far-head distances with their capped match lengths and the (length, distance)
 *     pairs of the descent steps that raised the running maximum.  The one parse dependence is the long-match skip (:145): a match
 *     longer than 129 is reported to the inserter, which takes back what it inserted beyond the match's first byte (an undo log
 *     of every word it overwrote) and replays the range with the skip rule.
 *  2. 64 POSITIONS AT ONCE.  The trees of different hash values share no live node, so the descents of a batch of 64 positions
 *     run interleaved, one step of every lane per round; positions of one batch with the same hash run one after the other.
 *     HT2 / HT3 / tree heads are gathered once per batch, same-key positions are resolved from the batch itself.
 *  3. find_match's ACCEPTANCE OVER A RECORD: rep candidates (the only ones that need the parse) first, then HT2 / HT3 / far head
 *     from the record with the reference's distance gating, then the descent's pairs filtered by `length > minlen`; a capped
 *     length (= good_len) is extended against the window on demand.
 *  4. (round 5) THE HASH SIDE OF THE ACCEPTANCE DOES NOT NEED THE PARSE EITHER.  minlen only grows, and a candidate is accepted iff its
 *     length exceeds the running minlen; so with m0 = minlen after the four rep candidates, "accepted under m0" == "accepted under
 *     m0 = 1, and longer than m0".  The inserter therefore runs the HT2 / HT3 / far-head / descent part of find_match once with
 *     minlen = 1 and leaves the pushed candidates H' (strictly increasing lengths) in the record; the parser's list is
 *     [rep part] + the suffix of H' longer than m0 -- no per-candidate work.  Exactness of the caps: lengths in a record stop at
 *     good_len.  (a) no rep reached good_len and no considered HT2 / HT3 / far candidate is capped: every comparison the reference makes
 *     is between numbers below the cap, except for the descent's terminal pair (length == good_len, last of H'), which is extended on
 *     demand.  (b) otherwise (`amb`, or a rep of good_len: the reference sets dist = 0xFFFFFFFF and skips HT2 / HT3): every later
 *     candidate needs a real length above good_len, so of the descent only the terminal pair (`dcap`) can still count -- the general
 *     walk over rep / HT2 / HT3 / far / that one pair decides.  btm_find_match_fast is this; it is asserted equal to the full
 *     walk over the raw pairs (btm_find_match) at EVERY call.  Needs good_len > 6 (the bound[] rule must see real lengths).
 */
#include <stdio.h>
#include "../../oracle/orc_encoder.c"

#define BT_R 256          /* record ring (positions) */
#define BT_MAXE 32        /* descent steps (bt_cyc <= 32 in every configuration CSCEncProps_Init produces) */
#define BT_ULOG 40

typedef struct {
    uint32_t d2, d3, dfar, dhead;          /* distances: HT2, HT3, far head (0 = none), tree head as this position saw it */
    uint32_t l2, l3, lfar;                 /* match lengths capped at min(good_len, climit) */
    uint32_t n;                            /* descent pairs */
    uint32_t el[BT_MAXE], ed[BT_MAXE];
    uint32_t un;                           /* undo log: tree words this insert overwrote */
    uint32_t uslot[BT_ULOG], uold[BT_ULOG];
    /* round 5: the hash side of find_match's acceptance, done by the inserter (point 4 of the header) */
    uint32_t hn, hl[BT_MAXE + 3], hd[BT_MAXE + 3];   /* H': what find_match pushes from HT2 / HT3 / far head / the descent when no rep candidate counts */
    uint32_t m3;                           /* minlen after the HT2 / HT3 / far-head stage under that assumption (capped lengths) */
    uint32_t amb;                          /* a CONSIDERED HT2 / HT3 / far-head candidate reached the compare cap: its real length decides what follows */
    uint32_t dcap;                         /* distance of the descent's pair that reached good_len (0 = none): the one pair whose real length can exceed its record */
} BtRec;

static struct {
    BtRec rec[BT_R];
    uint32_t sb0, pos0, btpos0, size, head;
    int la, pipe;
    unsigned long long n_batches, n_events, n_undo_pos, n_fallback_sb, n_pipe_sb, n_samehash, n_extend, n_rounds, n_steps, n_find, n_shadow, n_blocked, n_fast, n_slow;
    unsigned long long h_rounds[12], h_chain[12], h_steps[12], n_chainsteps;
} B;

static void btm_die(const char *what) { fprintf(stderr, "bt_model: %s\n", what); abort(); }

/* min(common prefix, lim) given that the first `start` bytes are known equal */
static uint32_t btm_prefix_from(const uint8_t *a, const uint8_t *b, uint32_t start, uint32_t lim)
{
    uint32_t n = start;
    while (n < lim && a[n] == b[n]) n++;
    return n > lim ? lim : n;
}

/* ---- the inserter: positions [i0, i0 + n) of the open sub-block, n <= 64 ---- */
static void btm_batch(OrcEnc *e, uint32_t i0, uint32_t n)
{
    uint32_t h2[64], h3[64], hb[64], o2[64], o3[64], oh[64];
    uint32_t dist[64], l[64], r[64], lold[64], rold[64], lenl[64], lenr[64], cyc[64], runmax[64];
    int active[64], waiting[64], prevlane[64];
    /* the tree ring reuses the slot of position x for position x + bt_size: a descent of this batch that reaches a candidate whose
     * slot belongs to a LATER position of the batch (distance > bt_size - 64) must see -- and modify -- that node as it was before
     * the batch, whatever the later position has stored meanwhile: SHADOW copies of the batch's own slots, taken before any store */
    uint32_t sh[64][2];
    int lsh[64], rsh[64];
    B.n_batches++;
    for (uint32_t k = 0; k < n; k++) { sh[k][0] = e->bt_nodes[(size_t)(B.btpos0 + i0 + k) * 2]; sh[k][1] = e->bt_nodes[(size_t)(B.btpos0 + i0 + k) * 2 + 1]; }
    /* hashes + ONE gather, before any store of this batch */
    for (uint32_t k = 0; k < n; k++) {
        const uint8_t *p = e->wnd + B.sb0 + i0 + k;
        h2[k] = hash2(p); h3[k] = hash3(p); hb[k] = hash6(p, e->bt_bits);
        o2[k] = e->ht2[h2[k]]; o3[k] = e->ht3[h3[k]]; oh[k] = e->bt_head[hb[k]];
    }
    /* same-key positions inside the batch */
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t pos = B.pos0 + i0 + k;
        BtRec *R = &B.rec[(i0 + k) % BT_R];
        uint32_t e2 = o2[k], e3 = o3[k], eh = oh[k];
        int later2 = 0, later3 = 0, laterb = 0;
        prevlane[k] = -1;
        for (uint32_t j = 0; j < n; j++) {
            if (j < k) {
                if (h2[j] == h2[k]) e2 = B.pos0 + i0 + j;
                if (h3[j] == h3[k]) e3 = B.pos0 + i0 + j;
                if (hb[j] == hb[k]) { eh = B.pos0 + i0 + j; prevlane[k] = (int)j; }
            } else if (j > k) {
                later2 |= h2[j] == h2[k]; later3 |= h3[j] == h3[k]; laterb |= hb[j] == hb[k];
            }
        }
        R->d2 = pos - e2; R->d3 = pos - e3; R->dhead = pos - eh;
        if (!later2) e->ht2[h2[k]] = pos;
        if (!later3) e->ht3[h3[k]] = pos;
        if (!laterb) e->bt_head[hb[k]] = pos;
        if (prevlane[k] >= 0) B.n_samehash++;
    }
    /* HT2 / HT3 / far-head match lengths, capped at good_len (the parser extends a capped one on demand) */
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t wpos = B.sb0 + i0 + k, limit = B.size - (i0 + k);
        BtRec *R = &B.rec[(i0 + k) % BT_R];
        const uint8_t *pcur = e->wnd + wpos;
        R->l2 = R->l3 = R->lfar = 0; R->dfar = 0;
        if (R->d2 < e->vld_rge) {
            uint32_t cp = wpos > R->d2 ? wpos - R->d2 : wpos + e->wnd_size - R->d2;       /* strict, csc_mf.cpp:306 */
            uint32_t cl = UMIN(limit, e->wnd_size - cp);
            R->l2 = btm_prefix_from(pcur, e->wnd + cp, 0, UMIN(cl, e->good_len));
        }
        if (R->d3 < e->vld_rge) {
            uint32_t cp = wrap_back(e, wpos, R->d3);
            uint32_t cl = UMIN(limit, e->wnd_size - cp);
            R->l3 = btm_prefix_from(pcur, e->wnd + cp, 0, UMIN(cl, e->good_len));
        }
        if (R->dhead >= e->bt_size && R->dhead < e->vld_rge) {
            uint32_t cp = wrap_back(e, wpos, R->dhead);
            uint32_t cl = UMIN(limit, e->wnd_size - cp);
            R->dfar = R->dhead;
            R->lfar = btm_prefix_from(pcur, e->wnd + cp, 0, UMIN(cl, e->good_len));
        }
    }
    /* the descents, one step of every running lane per round */
    for (uint32_t k = 0; k < n; k++) {
        BtRec *R = &B.rec[(i0 + k) % BT_R];
        const uint32_t btp = B.btpos0 + i0 + k;
        dist[k] = R->dhead; l[k] = btp * 2; r[k] = btp * 2 + 1; lold[k] = rold[k] = 0; lenl[k] = lenr[k] = 0; cyc[k] = 0; runmax[k] = 1;
        R->n = 0; R->un = 0; lsh[k] = rsh[k] = -1;
        waiting[k] = !B.pipe && prevlane[k] >= 0;
        active[k] = !waiting[k];
    }
    {   /* statistics: longest same-hash chain of the batch */
        uint32_t depth[64], mx = 0;
        for (uint32_t k = 0; k < n; k++) { depth[k] = prevlane[k] >= 0 ? depth[prevlane[k]] + 1 : 0; if (depth[k] > mx) mx = depth[k]; }
        uint32_t b = 0; while ((1u << b) <= mx && b < 11) b++;
        B.h_chain[b]++;
    }
    uint32_t rounds_here = 0;
    for (;;) {
        int any = 0;
        for (uint32_t k = 0; k < n; k++) if (active[k] || waiting[k]) any = 1;
        if (!any) break;
        B.n_rounds++; rounds_here++;
        /* a waiting lane starts once the previous lane of its hash is done */
        for (uint32_t k = 0; k < n; k++)
            if (waiting[k] && !active[prevlane[k]] && !waiting[prevlane[k]]) { waiting[k] = 0; active[k] = 2; }   /* 2: starts next round */
        /* PIPELINED CHAINS.  A lane does not wait for the earlier lanes of its hash to finish: a descent rewrites the tree top-down and
         * only ever stores into its two open slots (l, r); everything above them on its path is final, everything off its path is
         * untouched.  So a later position of the same tree may follow right behind -- it only must not read a node one of whose two
         * child slots is an open slot of an unfinished earlier position of its hash (snapshot taken when the round starts). */
        uint32_t hl2[64], hr2[64]; int unf[64];
        for (uint32_t k = 0; k < n; k++) { hl2[k] = l[k]; hr2[k] = r[k]; unf[k] = active[k] != 0; }
        for (uint32_t k = 0; k < n; k++) {
            if (active[k] == 2) { active[k] = 1; continue; }
            if (!active[k]) continue;
            BtRec *R = &B.rec[(i0 + k) % BT_R];
            const uint32_t wpos = B.sb0 + i0 + k, limit = B.size - (i0 + k), pos = B.pos0 + i0 + k, btp = B.btpos0 + i0 + k;
            uint32_t *nodes = e->bt_nodes;
#define BT_STORE(slot, old, val) do { if (R->un >= BT_ULOG) btm_die("undo log overflow"); R->uslot[R->un] = (slot); R->uold[R->un] = (old); R->un++; nodes[slot] = (val); } while (0)
#define BT_STORE_L(val) do { if (lsh[k] >= 0) sh[lsh[k]][1] = (val); else BT_STORE(l[k], lold[k], val); } while (0)
#define BT_STORE_R(val) do { if (rsh[k] >= 0) sh[rsh[k]][0] = (val); else BT_STORE(r[k], rold[k], val); } while (0)
            if (cyc[k] >= e->bt_cyc || dist[k] >= e->bt_size || dist[k] >= e->vld_rge) {
                BT_STORE_L(0); BT_STORE_R(0); active[k] = 0; continue;
            }
            const uint32_t cp = wrap_back(e, wpos, dist[k]);
            uint32_t clen = UMIN(lenl[k], lenr[k]);
            const uint32_t climit = UMIN(limit, e->wnd_size - cp);
            if (clen >= climit) { BT_STORE_L(0); BT_STORE_R(0); active[k] = 0; continue; }
            const uint32_t npos = btp >= dist[k] ? btp - dist[k] : btp + e->bt_size - dist[k];
            if (B.pipe) {
                /* which child slot of the node this step needs is known from the window bytes alone: the one it will move to
                 * (both when the step ends the descent by copying the node's children, :185, :433) */
                const uint32_t c2 = UMIN(e->good_len, climit);
                const uint32_t f = btm_prefix_from(e->wnd + wpos, e->wnd + cp, clen, c2);
                const int both = f >= e->good_len || B.pipe == 1;
                const int ends0 = f > clen && f < e->good_len && f >= c2;       /* ends with zero children: needs neither */
                const uint32_t need = e->wnd[cp + f] < e->wnd[wpos + f] ? npos * 2 + 1 : npos * 2;
                int blocked = 0;
                for (uint32_t j = 0; j < k; j++) {
                    if (!unf[j] || hb[j] != hb[k]) continue;
                    if (both) { if (hl2[j] >> 1 == npos || hr2[j] >> 1 == npos) blocked = 1; }
                    else if (!ends0 && (hl2[j] == need || hr2[j] == need)) blocked = 1;
                }
                if (blocked) { B.n_blocked++; continue; }
            }
            const uint32_t rel = npos - (B.btpos0 + i0);
            const int shadowed = rel < n && rel > k;
            if (shadowed) B.n_shadow++;
            const uint32_t t0 = shadowed ? sh[rel][0] : nodes[(size_t)npos * 2], t1 = shadowed ? sh[rel][1] : nodes[(size_t)npos * 2 + 1];
            const uint32_t climit2 = UMIN(e->good_len, climit);
            const uint8_t *pcur = e->wnd + wpos, *pm = e->wnd + cp;
            const uint32_t full = btm_prefix_from(pcur, pm, clen, climit2);
            B.n_steps++;
            if (full > clen) {
                clen = full;
                if (clen > runmax[k]) {
                    runmax[k] = clen;
                    if (clen > 6 || dist[k] < kBound[clen]) {      /* :429: a pair the bound[] rule drops is never pushed */
                        if (R->n >= BT_MAXE) btm_die("record overflow");
                        R->el[R->n] = clen; R->ed[R->n] = dist[k]; R->n++;
                    }
                }
                if (clen >= e->good_len) { BT_STORE_L(t0); BT_STORE_R(t1); active[k] = 0; continue; }
                else if (clen >= climit2) { BT_STORE_L(0); BT_STORE_R(0); active[k] = 0; continue; }
            }
            if (pm[clen] < pcur[clen]) {
                BT_STORE_L(pos - dist[k]);
                l[k] = npos * 2 + 1; lold[k] = t1; lsh[k] = shadowed ? (int)rel : -1; dist[k] = pos - t1; lenl[k] = clen;
            } else {
                BT_STORE_R(pos - dist[k]);
                r[k] = npos * 2; rold[k] = t0; rsh[k] = shadowed ? (int)rel : -1; dist[k] = pos - t0; lenr[k] = clen;
            }
            cyc[k]++;
        }
    }
    /* H' (header, point 4): find_match's walk over HT2 / HT3 / far head / pairs with minlen = 1 and no rep kill, on the capped lengths */
    for (uint32_t k = 0; k < n; k++) {
        BtRec *R = &B.rec[(i0 + k) % BT_R];
        uint32_t m = 1, kill = 0;
        R->hn = 0; R->amb = 0; R->dcap = 0;
#define H_PUSH(L, D) do { R->hl[R->hn] = (L); R->hd[R->hn] = (D); R->hn++; } while (0)
        if (R->d2 < e->vld_rge) {                                   /* :297-301 (dist = 0: always looked at) */
            if (R->l2 >= e->good_len) R->amb = 1;
            if (R->l2 > m) { m = R->l2; if (!(R->l2 <= 6 && R->d2 >= kBound[R->l2])) { H_PUSH(R->l2, R->d2); if (R->l2 >= e->good_len) kill = 1; } }
        }
        if (!kill && R->d3 > R->d2 && R->d3 < e->vld_rge) {         /* :334: only behind HT2's distance */
            if (R->l3 >= e->good_len) R->amb = 1;
            if (R->l3 > m) { m = R->l3; if (!(R->l3 <= 6 && R->d3 >= kBound[R->l3])) H_PUSH(R->l3, R->d3); }
        }
        if (R->dfar) {
            if (R->lfar >= e->good_len) R->amb = 1;
            if (R->lfar > m) { m = R->lfar; if (!(R->lfar <= 6 && R->dfar >= kBound[R->lfar])) H_PUSH(R->lfar, R->dfar); }
        }
        R->m3 = m;
        for (uint32_t j = 0; j < R->n; j++) {
            if (R->el[j] > m) H_PUSH(R->el[j], R->ed[j]);           /* (raw pairs increase strictly: a suffix) */
            if (R->el[j] >= e->good_len) { if (j + 1 != R->n) btm_die("a capped pair that is not the terminal one"); R->dcap = R->ed[j]; }
        }
#undef H_PUSH
    }
    { uint32_t b = 0; while ((1u << b) <= rounds_here && b < 11) b++; B.h_rounds[b]++; }
    for (uint32_t k = 0; k < n; k++) { uint32_t st = cyc[k], b = 0; while ((1u << b) <= st && b < 11) b++; B.h_steps[b]++; if (prevlane[k] >= 0) B.n_chainsteps += st; }
}

/* take back the inserts of positions [a, head), newest first */
static void btm_undo(OrcEnc *e, uint32_t a)
{
    for (uint32_t p = B.head; p-- > a;) {
        BtRec *R = &B.rec[p % BT_R];
        const uint8_t *q = e->wnd + B.sb0 + p;
        const uint32_t pos = B.pos0 + p;
        for (uint32_t k = R->un; k-- > 0;) e->bt_nodes[R->uslot[k]] = R->uold[k];
        e->bt_head[hash6(q, e->bt_bits)] = pos - R->dhead;
        e->ht2[hash2(q)] = pos - R->d2;
        e->ht3[hash3(q)] = pos - R->d3;
        B.n_undo_pos++;
    }
    B.head = a;
}

static void btm_advance(OrcEnc *e, uint32_t upto)
{
    if (upto > B.size) upto = B.size;
    while (B.head < upto) {
        uint32_t n = UMIN(64u, B.size - B.head);
        if (B.head + n > upto + 63) btm_die("advance");
        btm_batch(e, B.head, n);
        B.head += n;
    }
}

/* the parser coded a match of `len` bytes at sub-block offset s: SlidePos(s, len), csc_mf.cpp:134-206 */
static void btm_slide(OrcEnc *e, uint32_t s, uint32_t len)
{
    if (len <= 129) return;                 /* no skip phase: the speculative inserts ARE SlidePos's */
    B.n_events++;
    if (B.head > s + 1) btm_undo(e, s + 1);
    else if (B.head < s + 1) btm_die("inserter behind a match start");
    uint32_t i = 1;
    for (; i + 128 < len; i += 4) {         /* :145: HT2 / HT3 only, every fourth position */
        const uint8_t *q = e->wnd + B.sb0 + s + i;
        e->ht2[hash2(q)] = B.pos0 + s + i;
        e->ht3[hash3(q)] = B.pos0 + s + i;
    }
    B.head = s + i;                          /* positions from here on get the whole insert */
}

/* find_match over the record of sub-block offset p (csc_mf.cpp:243-495 for bt_head != 0, ht_width == 0) */
static uint32_t btm_find_match(OrcEnc *e, MFUnit *ret, const uint32_t *rep_dist, uint32_t p)
{
    const uint32_t wpos = B.sb0 + p, limit = B.size - p;
    const uint8_t *pcur = e->wnd + wpos;
    uint32_t minlen = 1, cnt = 0, dist = 0;
    B.n_find++;
    btm_advance(e, p + 1 + (uint32_t)B.la);
    if (B.head <= p) btm_die("record not there");
    const BtRec *R = &B.rec[p % BT_R];
#define PUSH_CAND(L, D) do { ret[cnt].len = (L); ret[cnt].dist = (D); if (cnt + 2 < MF_CAND_LIMIT) cnt++; } while (0)
    for (uint32_t i = 0; i < 4; i++) {
        if (rep_dist[i] >= e->vld_rge) continue;
        uint32_t cmp_pos = wrap_back(e, wpos, rep_dist[i]);
        uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
        uint32_t match_len = prefix_len(pcur, e->wnd + cmp_pos, climit);
        if (i == 0 && match_len >= 2) PUSH_CAND(1, 1);          /* the reference's test is pmatch[1] == pcur[1] after a prefix >= 1: length >= 2 */
        if (match_len > minlen) {
            minlen = match_len;
            PUSH_CAND(match_len, 1 + i);
            if (match_len >= e->good_len) { dist = 0xFFFFFFFFu; break; }
        }
    }
    for (int t = 0; t < 2; t++) {
        const uint32_t d = t == 0 ? R->d2 : R->d3;
        uint32_t ml = t == 0 ? R->l2 : R->l3;
        if (!(d > dist)) continue;
        dist = d;
        if (d >= e->vld_rge) continue;
        if (ml >= e->good_len) {             /* capped: the real length */
            uint32_t cp = t == 0 ? (wpos > d ? wpos - d : wpos + e->wnd_size - d) : wrap_back(e, wpos, d);
            ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
            B.n_extend++;
        }
        if (ml > minlen) {
            minlen = ml;
            if (ml <= 6 && d >= kBound[ml]) continue;
            PUSH_CAND(ml, 4 + d);
            if (ml >= e->good_len) dist = 0xFFFFFFFFu;
        }
    }
    if (R->dfar) {
        uint32_t ml = R->lfar;
        if (ml >= e->good_len) {
            uint32_t cp = wrap_back(e, wpos, R->dfar);
            ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
            B.n_extend++;
        }
        if (ml > minlen) {
            minlen = ml;
            if (!(ml <= 6 && R->dfar >= kBound[ml])) PUSH_CAND(ml, 4 + R->dfar);
        }
    }
    for (uint32_t k = 0; k < R->n; k++) {
        uint32_t ml = R->el[k];
        if (ml >= e->good_len) {
            uint32_t cp = wrap_back(e, wpos, R->ed[k]);
            ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
            B.n_extend++;
        }
        if (ml > minlen) { minlen = ml; PUSH_CAND(ml, 4 + R->ed[k]); }
    }
#undef PUSH_CAND
    return cnt;
}

/* the parser's side of point 4: rep candidates here, the hash side from H' */
static uint32_t btm_find_match_fast(OrcEnc *e, MFUnit *ret, const uint32_t *rep_dist, uint32_t p)
{
    const uint32_t wpos = B.sb0 + p, limit = B.size - p;
    const uint8_t *pcur = e->wnd + wpos;
    const BtRec *R = &B.rec[p % BT_R];
    uint32_t minlen = 1, cnt = 0, kill = 0;
#define PUSH_CAND(L, D) do { ret[cnt].len = (L); ret[cnt].dist = (D); if (cnt + 2 < MF_CAND_LIMIT) cnt++; } while (0)
    for (uint32_t i = 0; i < 4; i++) {
        if (rep_dist[i] >= e->vld_rge) continue;
        uint32_t cmp_pos = wrap_back(e, wpos, rep_dist[i]);
        uint32_t match_len = prefix_len(pcur, e->wnd + cmp_pos, UMIN(limit, e->wnd_size - cmp_pos));
        if (i == 0 && match_len >= 2) PUSH_CAND(1, 1);
        if (match_len > minlen) {
            minlen = match_len;
            PUSH_CAND(match_len, 1 + i);
            if (match_len >= e->good_len) { kill = 1; break; }
        }
    }
    if (!kill && !R->amb) {
        /* (a): the suffix of H' that is longer than the reps' minlen; its last entry may be the capped terminal pair */
        B.n_fast++;
        for (uint32_t j = 0; j < R->hn; j++) {
            uint32_t ml = R->hl[j];
            if (ml <= minlen) continue;
            if (ml >= e->good_len) {
                if (j + 1 != R->hn || R->hd[j] != R->dcap) btm_die("a capped entry of H' that is not the terminal pair");
                uint32_t cp = wrap_back(e, wpos, R->hd[j]);
                ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
            }
            PUSH_CAND(ml, 4 + R->hd[j]);
        }
        return cnt;
    }
    /* (b): the general walk over HT2 / HT3 / far head and the one pair that can still count */
    B.n_slow++;
    uint32_t dist = kill ? 0xFFFFFFFFu : 0;
    for (int t = 0; t < 2; t++) {
        const uint32_t d = t == 0 ? R->d2 : R->d3;
        uint32_t ml = t == 0 ? R->l2 : R->l3;
        if (!(d > dist)) continue;
        dist = d;
        if (d >= e->vld_rge) continue;
        if (ml >= e->good_len) {
            uint32_t cp = t == 0 ? (wpos > d ? wpos - d : wpos + e->wnd_size - d) : wrap_back(e, wpos, d);
            ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
        }
        if (ml > minlen) {
            minlen = ml;
            if (ml <= 6 && d >= kBound[ml]) continue;
            PUSH_CAND(ml, 4 + d);
            if (ml >= e->good_len) dist = 0xFFFFFFFFu;
        }
    }
    if (R->dfar) {
        uint32_t ml = R->lfar;
        if (ml >= e->good_len) {
            uint32_t cp = wrap_back(e, wpos, R->dfar);
            ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
        }
        if (ml > minlen) {
            minlen = ml;
            if (!(ml <= 6 && R->dfar >= kBound[ml])) PUSH_CAND(ml, 4 + R->dfar);
        }
    }
    if (R->dcap) {
        uint32_t cp = wrap_back(e, wpos, R->dcap);
        uint32_t ml = btm_prefix_from(pcur, e->wnd + cp, e->good_len, UMIN(limit, e->wnd_size - cp));
        if (ml > minlen) { minlen = ml; PUSH_CAND(ml, 4 + R->dcap); }
    }
#undef PUSH_CAND
    return cnt;
}

static void btm_find_priced(OrcEnc *e, uint32_t state, MFUnit *ret, const uint32_t *rep_dist, uint32_t p)
{
    e->mfcand[0].len = 1; e->mfcand[0].dist = 0;
    uint32_t n = btm_find_match(e, e->mfcand + 1, rep_dist, p);
    if (e->good_len > 6) {
        /* point 4 of the header: the record's pre-accepted hash list gives the same candidates, at every call */
        MFUnit alt[MF_CAND_LIMIT + 2];
        uint32_t n2 = btm_find_match_fast(e, alt, rep_dist, p);
        if (n2 != n) btm_die("fast acceptance: candidate count differs");
        for (uint32_t i = 0; i < n; i++) if (alt[i].len != e->mfcand[1 + i].len || alt[i].dist != e->mfcand[1 + i].dist) btm_die("fast acceptance: candidate differs");
    }
    ret[0] = e->mfcand[n];
    if (ret[0].len >= e->good_len) return;
    ret[1].dist = 0;
    uint32_t lpos = 1;
    for (uint32_t i = 1; i <= n; i++) {
        uint32_t distprice, rdist;
        if (e->mfcand[i].len == 1 && e->mfcand[i].dist == 1) { ret[1].len = rep0len1_price(e, state); ret[1].dist = 1; continue; }
        else if (e->mfcand[i].dist <= 4) { distprice = rep_dist_price(e, state, e->mfcand[i].dist - 1); rdist = 0; }
        else { distprice = match_dist_price(e, state, e->mfcand[i].dist - 5); rdist = e->mfcand[i].dist - 4; }
        while (lpos < e->mfcand[i].len) {
            lpos++;
            if (lpos <= 6 && rdist >= kBound[lpos]) { ret[lpos].dist = 0; continue; }
            ret[lpos].dist = e->mfcand[i].dist;
            ret[lpos].len = distprice + match_len_price(e, lpos - 2);
        }
    }
}

/* LZ::compress_advanced (csc_lz.cpp:207-333) over records; positions are sub-block offsets */
static void btm_adv_pipe(OrcEnc *e, uint32_t size)
{
    APUnit *ap = e->ap;
    MFUnit *appt = e->appt;
    uint32_t apend = 0, apcur = 0;
    B.sb0 = e->wnd_curpos; B.pos0 = e->pos; B.btpos0 = e->bt_pos; B.size = size; B.head = 0;
    for (uint32_t i = 0; i < size;) {
        btm_find_priced(e, e->state, appt, e->rep_dist, i);
        if (appt[0].dist == 0) {
            encode_literal(e, e->wnd[e->wnd_curpos]);
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
            if (apcur) {
                int l = ap[apcur].back_pos;
                memcpy(ap[apcur].rep_dist, ap[l].rep_dist, sizeof(ap[l].rep_dist));
                if (ap[apcur].dist == 0) ap[apcur].state = (ap[l].state * 4) & 0x3F;
                else if (ap[apcur].dist <= 4) {
                    uint32_t len = apcur - (uint32_t)l;
                    if (len == 1 && ap[apcur].dist == 1) ap[apcur].state = (ap[l].state * 4 + 2) & 0x3F;
                    else {
                        ap[apcur].state = (ap[l].state * 4 + 3) & 0x3F;
                        uint32_t k = ap[apcur].dist - 1, tmp = ap[apcur].rep_dist[k];
                        if (k >= 1) { for (; k > 0; k--) ap[apcur].rep_dist[k] = ap[apcur].rep_dist[k - 1]; ap[apcur].rep_dist[0] = tmp; }
                    }
                } else {
                    ap[apcur].state = (ap[l].state * 4 + 1) & 0x3F;
                    ap[apcur].rep_dist[0] = ap[apcur].dist - 4;
                    ap[apcur].rep_dist[1] = ap[l].rep_dist[0]; ap[apcur].rep_dist[2] = ap[l].rep_dist[1]; ap[apcur].rep_dist[3] = ap[l].rep_dist[2];
                }
                if (apcur < aplimit) btm_find_priced(e, ap[apcur].state, appt, ap[apcur].rep_dist, i + apcur);
            }
            if (apcur == aplimit) { lz_ap_backward(e, (int)apcur); i += apcur; break; }
            if (appt[0].len == 1 && apcur + 1 == apend) {
                lz_ap_backward(e, (int)apcur);
                encode_literal(e, ap[apcur].lit);
                i += apcur;
                e->wnd_curpos++;
                i++;
                break;
            }
            if (apcur + 1 >= apend) ap[apend++].price = 0xFFFFFFFFu;
            if (appt[0].len >= e->lz_good_len || (appt[0].len > 1 && appt[0].len + apcur >= aplimit)) {
                lz_ap_backward(e, (int)apcur);
                i += apcur;
                lz_encode_nonlit(e, appt[0]);
                btm_slide(e, i, appt[0].len);
                i += appt[0].len;
                e->wnd_curpos += appt[0].len;
                e->ctx = e->wnd[e->wnd_curpos - 1];
                break;
            }
            uint32_t lit_ctx = e->wnd_curpos ? e->wnd[e->wnd_curpos - 1] : 0;
            uint32_t cprice = literal_price(e, ap[apcur].state, lit_ctx, e->wnd[e->wnd_curpos]);
            if (cprice + ap[apcur].price < ap[apcur + 1].price) { ap[apcur + 1].dist = 0; ap[apcur + 1].back_pos = (int)apcur; ap[apcur + 1].price = cprice + ap[apcur].price; }
            if (appt[1].dist && appt[1].len + ap[apcur].price < ap[apcur + 1].price) { ap[apcur + 1].dist = 1; ap[apcur + 1].back_pos = (int)apcur; ap[apcur + 1].price = appt[1].len + ap[apcur].price; }
            uint32_t len = appt[0].len;
            while (apcur + len >= apend) ap[apend++].price = 0xFFFFFFFFu;
            while (len > 1) {
                if (appt[len].dist && appt[len].len + ap[apcur].price < ap[apcur + len].price) {
                    ap[apcur + len].dist = appt[len].dist; ap[apcur + len].back_pos = (int)apcur; ap[apcur + len].price = appt[len].len + ap[apcur].price;
                }
                len--;
            }
            apcur++;
            e->wnd_curpos++;
        }
    }
    /* the sub-block is closed: every position is in the tables */
    btm_advance(e, size);
    if (B.head != size) btm_die("inserter did not finish the sub-block");
    e->pos = B.pos0 + size;
    e->bt_pos = B.btpos0 + size;
}

static void btm_adv(OrcEnc *e, uint32_t size)
{
    /* the pipeline form covers the configuration level 5 produces; corner cases keep the one-wavefront form (as in the kernels):
     * the tree ring about to wrap (bt_pos_ may stand at bt_size_, csc_mf.cpp:201 vs :405), pos_ about to be renormalised */
    const int ok = e->bt_head && !e->ht_width && e->ht_low && e->bt_cyc <= BT_MAXE && e->good_len <= 255
                   && e->bt_pos + size + 8 < e->bt_size && e->pos < 0xFFFF0000u;
    if (!ok) { B.n_fallback_sb++; lz_compress_advanced(e, size); return; }
    B.n_pipe_sb++;
    btm_adv_pipe(e, size);
}

__attribute__((destructor)) static void btm_stats(void)
{
    if (!getenv("BTM_STATS")) return;
    fprintf(stderr, "bt_model: sub-blocks pipe %llu fallback %llu, batches %llu, rounds %llu, steps %llu, finds %llu, same-hash lanes %llu, long-match events %llu, positions undone %llu, extensions %llu, shadowed steps %llu\n",
            B.n_pipe_sb, B.n_fallback_sb, B.n_batches, B.n_rounds, B.n_steps, B.n_find, B.n_samehash, B.n_events, B.n_undo_pos, B.n_extend, B.n_shadow);
    fprintf(stderr, "bt_model: histograms (bucket b: value < 2^b): rounds/batch"); for (int i = 0; i < 12; i++) fprintf(stderr, " %llu", B.h_rounds[i]);
    fprintf(stderr, " | longest chain/batch"); for (int i = 0; i < 12; i++) fprintf(stderr, " %llu", B.h_chain[i]);
    fprintf(stderr, " | steps/lane"); for (int i = 0; i < 12; i++) fprintf(stderr, " %llu", B.h_steps[i]);
    fprintf(stderr, " | steps of chained lanes %llu | blocked lane-rounds %llu | acceptance from H' %llu, general walk %llu\n", B.n_chainsteps, B.n_blocked, B.n_fast, B.n_slow);
}
__attribute__((constructor)) static void btm_install(void)
{
    const char *s;
    B.la = (s = getenv("BTM_LA")) ? atoi(s) : 64;
    B.pipe = (s = getenv("BTM_PIPE")) ? atoi(s) : 2;   /* 2 = the kernel's rule (only the child slot the step needs), 1 = both child slots, 0 = chains one after the other */
    orc_adv_hook = btm_adv;
}
