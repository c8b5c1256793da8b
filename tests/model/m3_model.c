/*
 * tests/model/m3_model.c -- CPU model of the ARRANGEMENT the HIP kernels give LZ::compress_advanced on the hash-table
 * levels (csc_amd/csrc/csc_kernels_dp4.inc).  TEST INFRASTRUCTURE: it includes the oracle's encoder, replaces only
 * compress_advanced (csc_lz.cpp:207-333) through the oracle's test hook, and must produce the oracle's bytes
 * (tests/test_m3_model.py).  What it proves, before any of it runs on a GPU:
 *
 *  1. SPECULATIVE MATCH-FINDER PRE-PASS.  Every position of a sub-block is inserted ahead of the parser with find_match's
 *     rule (HT2/HT3 overwrite, bucket shift; csc_mf.cpp:365-366,487-491), 64 positions per batch: all table words are
 *     gathered first, same-key positions inside the batch are resolved in registers, one store per touched word.  The
 *     parser's SlidePos (csc_mf.cpp:134-206) inserts the same words EXCEPT (a) the long-match skip (:145) and (b) the
 *     same-hash no-shift rule (:150-156); a slid range that contains neither needs nothing, one that does is undone
 *     (the gathered words are the undo log) and replayed exactly.
 *  2. PARSE-INDEPENDENT HASH CANDIDATES.  Distance, full match length, bound[] verdict and the distance-gating of
 *     HT2 -> HT3 -> bucket depend on the tables and the window only; find_match's sequential acceptance (:266-485) is
 *     then a prefix maximum over the 4 rep slots + the hash slots.
 *  3. REP MATCH LENGTHS FROM EQUALITY MASKS.  For a distance d, bit t of mask(base) says wnd[base+t] == wnd[base+t-d];
 *     the match length of a rep candidate at any node is a count of trailing ones, capped like the reference caps it
 *     (limit, window end).  A pushed hash candidate brings its mask along, so the rep lengths after a match need no load.
 *  4. THE DP IN A RING RELATIVE TO THE CURRENT NODE (lane l = node k + l), labels carrying coder state and four entry
 *     ids instead of being derived at visit time (:231-268), relaxation in node order with the reference's strict `<`.
 */
#include <stdio.h>
#include "../../oracle/orc_encoder.c"

#define M3_NS 12          /* hash slots: 0 HT2, 1 HT3, 2.. bucket */
#define M3_RING 32
#define M3_INF 0xFFFFFFFFu

typedef struct {
    uint32_t dist, ent;   /* candidate distance (pos - ent); ent = table word before this position's insert */
    uint32_t ml;          /* full common-prefix length under the reference's caps */
    int eid;              /* entry id (mask) when pushable, else -1 */
    uint8_t cons, drop;
    uint32_t dcost;       /* (slot > 2 ? slot + 2 : 2) * 128 of GetMatchDistPrice */
} M3Slot;
typedef struct {
    M3Slot s[M3_NS];
    uint32_t h2, h3, h6;
    uint8_t hdev;
    /* find_match's acceptance over the hash slots ALONE (minlen still 1) and FindMatchWithPrice's table for it, as the compare
     * wavefronts precompute them: pm_h = pushed hash slots, a0l_h = the longest, len_owner[l] = hash slot + 1 that owns length l */
    uint32_t pm_h, a0l_h;
    uint8_t len_owner[M3_RING + 1];         /* h6 == h6 of the previous position, or h6 == 0: SlidePos would not shift here */
} M3Rec;
typedef struct { uint32_t dist, base, cov_end; uint64_t mask; int fwd; uint32_t req_at; } M3Ent;

static struct {
    M3Rec *rec;           /* one per position of the sub-block */
    M3Ent *ent; int nent, cap_ent;
    int rid[4];           /* entry ids of LZ::rep_dist_ */
    uint32_t sb0, pos0, size, ih, ch;
    int la, refresh_delay, refresh_at;
    /* statistics */
    unsigned long long n_nodes, n_windows, n_slide, n_dev, n_undo_pos, n_slow, n_refresh, n_batches, n_exact_pos, n_direct_lit, n_split_checked;
    unsigned long long n_len_gt129, n_hdev_events, n_repnodes, n_h0, n_h1, n_h2, n_h3p, n_a0long;
} M;

static void m3_die(const char *what) { fprintf(stderr, "m3_model: %s\n", what); abort(); }

static int m3_new_ent(uint32_t dist, uint32_t base, uint64_t mask, uint32_t cov_end)
{
    if (M.nent == M.cap_ent) { M.cap_ent = M.cap_ent ? M.cap_ent * 2 : 4096; M.ent = (M3Ent *)realloc(M.ent, sizeof(M3Ent) * (size_t)M.cap_ent); }
    M3Ent *x = &M.ent[M.nent];
    x->dist = dist; x->base = base; x->mask = mask; x->cov_end = cov_end; x->fwd = -1; x->req_at = 0xFFFFFFFFu;
    return M.nent++;
}

/* bit t: wnd[base + t] == wnd[base + t - d].  The mask CARRIES the reference's caps (csc_mf.cpp:268-269): no bit at or beyond the
 * sub-block end (limit), and a mask based in front of position == distance (source wrapped to the window's end) stops there
 * (wnd_size - cmp_pos) -- so a query is a count of trailing ones and nothing else. */
static uint64_t m3_eq_mask(const OrcEnc *e, uint32_t base, uint32_t d, uint32_t sb_end)
{
    uint64_t m = 0;
    for (uint32_t t = 0; t < 64 && base + t < sb_end; t++) {
        uint32_t q = base + t;
        if (base < d && q >= d) break;
        uint32_t src = q >= d ? q - d : q + e->wnd_size - d;
        if (e->wnd[q] == e->wnd[src]) m |= 1ull << t;
    }
    return m;
}

/* match length of distance-entry x at window position wpos under find_match's caps (:267-279); *slow = the mask did not cover it */
static uint32_t m3_rep_len(OrcEnc *e, const M3Ent *x, uint32_t wpos, uint32_t limit, int *slow)
{
    uint32_t d = x->dist;
    uint32_t cmp_pos = wrap_back(e, wpos, d);
    uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
    uint32_t direct = prefix_len(e->wnd + wpos, e->wnd + cmp_pos, climit);
    uint32_t off = wpos - x->base;
    *slow = 0;
    if (wpos < x->base || off >= 64) { *slow = 1; return direct; }
    uint64_t v = ~(x->mask >> off);
    uint32_t run = v ? (uint32_t)__builtin_ctzll(v) : 64;
    if (run > 64 - off) run = 64 - off;
    uint32_t ml = run;                                                     /* no cap here: the mask carries them */
    if (run >= 64 - off) { *slow = 1; return direct; }                     /* reached the end of the mask: extend by loads */
    if (x->base < d && d <= wpos) { *slow = 1; return direct; }            /* mask based in front of position == distance */
    if (ml != direct) { fprintf(stderr, "mask %u direct %u off %u d %u wpos %u base %u climit %u\n", ml, direct, off, d, wpos, x->base, climit); m3_die("eq-mask length differs from the direct compare"); }
    return ml;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* inserter: positions [i0, i0 + n) of the sub-block, n <= 64, find_match's insert rule, batch form                  */
static void m3_insert_batch(OrcEnc *e, uint32_t i0, uint32_t n)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    uint32_t o2[64], o3[64], ob[64][M3_NS];
    M.n_batches++;
    for (uint32_t l = 0; l < n; l++) {            /* hashes + ONE gather per lane, before any store */
        M3Rec *r = &M.rec[i0 + l];
        const uint8_t *p = e->wnd + M.sb0 + i0 + l;
        r->h2 = hash2(p); r->h3 = hash3(p); r->h6 = hash6(p, e->ht_bits);
        o2[l] = e->ht2[r->h2]; o3[l] = e->ht3[r->h3];
        for (uint32_t w = 0; w < W; w++) ob[l][w] = e->ht6[(size_t)r->h6 * e->ht_width + w];
    }
    for (uint32_t l = 0; l < n; l++) {            /* same-key lanes inside the batch, resolved "in registers" */
        M3Rec *r = &M.rec[i0 + l];
        const uint32_t pos = M.pos0 + i0 + l;
        uint32_t e2 = o2[l], e3 = o3[l], eb[M3_NS + 64];
        uint32_t nb = 0;
        for (int j = (int)l - 1; j >= 0; j--) if (M.rec[i0 + j].h2 == r->h2) { e2 = M.pos0 + i0 + (uint32_t)j; break; }
        for (int j = (int)l - 1; j >= 0; j--) if (M.rec[i0 + j].h3 == r->h3) { e3 = M.pos0 + i0 + (uint32_t)j; break; }
        for (int j = (int)l - 1; j >= 0 && nb < W; j--) if (M.rec[i0 + j].h6 == r->h6) eb[nb++] = M.pos0 + i0 + (uint32_t)j;
        for (uint32_t w = 0; w < W && nb < W; w++) eb[nb++] = ob[l][w];
        r->s[0].ent = e2; r->s[0].dist = pos - e2;
        r->s[1].ent = e3; r->s[1].dist = pos - e3;
        for (uint32_t w = 0; w < W; w++) { r->s[2 + w].ent = eb[w]; r->s[2 + w].dist = pos - eb[w]; }
        {
            const uint32_t prev_h6 = (i0 + l) ? M.rec[i0 + l - 1].h6 : 0xFFFFFFFFu;
            r->hdev = (r->h6 == 0 || r->h6 == prev_h6) ? 1 : 0;
        }
    }
    for (uint32_t l = 0; l < n; l++) {            /* one store per touched word: the LAST lane of a key writes */
        M3Rec *r = &M.rec[i0 + l];
        const uint32_t pos = M.pos0 + i0 + l;
        int last2 = 1, last3 = 1, last6 = 1;
        for (uint32_t j = l + 1; j < n; j++) {
            if (M.rec[i0 + j].h2 == r->h2) last2 = 0;
            if (M.rec[i0 + j].h3 == r->h3) last3 = 0;
            if (M.rec[i0 + j].h6 == r->h6) last6 = 0;
        }
        if (last2) e->ht2[r->h2] = pos;
        if (last3) e->ht3[r->h3] = pos;
        if (last6 && W) {
            uint32_t *b = e->ht6 + (size_t)r->h6 * e->ht_width;
            for (uint32_t w = W - 1; w > 0; w--) b[w] = r->s[2 + w - 1].ent;
            b[0] = pos;
        }
    }
}

/* compare stage for position i: which hash slots find_match would look at, their full lengths, bound[] verdicts, masks */
static void m3_compare(OrcEnc *e, uint32_t i)
{
    M3Rec *r = &M.rec[i];
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    const uint32_t wpos = M.sb0 + i, limit = M.size - i, vld = e->vld_rge, sb_end = M.sb0 + M.size;
    uint32_t Mx = 0;                                   /* running distance of the HT chain (:301-363, :453-485), no good_len hit */
    for (uint32_t s = 0; s < 2 + W; s++) {
        M3Slot *q = &r->s[s];
        q->cons = 0; q->drop = 0; q->ml = 0; q->eid = -1; q->dcost = 0;
        if (q->dist <= Mx) continue;
        Mx = q->dist;
        if (q->dist >= vld) continue;
        q->cons = 1;
        uint32_t cmp_pos;
        if (s == 0) cmp_pos = wpos > q->dist ? wpos - q->dist : wpos + e->wnd_size - q->dist;     /* HT2: strict, :306 */
        else cmp_pos = wrap_back(e, wpos, q->dist);
        uint32_t climit = UMIN(limit, e->wnd_size - cmp_pos);
        q->ml = prefix_len(e->wnd + wpos, e->wnd + cmp_pos, climit);
        q->drop = (q->ml <= 6 && q->dist >= kBound[q->ml]) ? 1 : 0;
        if (q->ml >= 2 && !q->drop) {
            q->eid = m3_new_ent(q->dist, wpos, m3_eq_mask(e, wpos, q->dist, sb_end), sb_end);
            uint32_t l = dist_slot(q->dist - 1);
            q->dcost = (l > 2 ? l + 2 : 2) * 128;
        }
    }
    /* hash-only acceptance (csc_mf.cpp:301-363,453-485 with minlen 1) and its length table (:600-624) */
    {
        uint32_t premax = 1, prevL = 1;
        r->pm_h = 0; r->a0l_h = 1;
        memset(r->len_owner, 0, sizeof(r->len_owner));
        for (uint32_t s = 0; s < 2 + W; s++) {
            const M3Slot *q = &r->s[s];
            const uint32_t Ls = q->cons ? q->ml : 0;
            if (q->cons && Ls > premax && !q->drop) {
                r->pm_h |= 1u << s;
                for (uint32_t l = prevL + 1; l <= Ls && l <= M3_RING; l++)
                    if (!(l <= 6 && q->dist >= kBound[l])) r->len_owner[l] = (uint8_t)(s + 1);
                prevL = Ls; r->a0l_h = Ls;
                if (Ls >= e->good_len) break;
            }
            if (Ls > premax) premax = Ls;
        }
    }
}

static void m3_ensure(OrcEnc *e, uint32_t i_master)
{
    uint32_t target = UMIN(M.size, i_master + (uint32_t)M.la);
    if (target <= i_master) target = UMIN(M.size, i_master + 1);
    while (M.ih < target) {
        uint32_t n = UMIN(64u, M.size - M.ih);
        m3_insert_batch(e, M.ih, n);
        M.ih += n;
    }
    while (M.ch < M.ih) { m3_compare(e, M.ch); M.ch++; }
}

/* SlidePos (csc_mf.cpp:134-206, hash tables only) for sub-block positions [from, s + len), the match starting at s */
static void m3_exact_slide(OrcEnc *e, uint32_t s, uint32_t len, uint32_t from, uint32_t lasth6)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    for (uint32_t i = from - s; i < len;) {
        const uint8_t *p = e->wnd + M.sb0 + s + i;
        const uint32_t pos = M.pos0 + s + i;
        e->ht2[hash2(p)] = pos;
        e->ht3[hash3(p)] = pos;
        if (i + 128 < len) { i += 4; continue; }
        if (e->ht_width) {
            uint32_t h6 = hash6(p, e->ht_bits);
            uint32_t *b = e->ht6 + (size_t)h6 * e->ht_width;
            if (h6 != lasth6) for (uint32_t j = W - 1; j > 0; j--) b[j] = b[j - 1];
            b[0] = pos;
            lasth6 = h6;
        }
        i++;
        M.n_exact_pos++;
    }
}

/* take back the speculative inserts of positions [a, ih): 64 at a time from the top, the FIRST position of a key in a batch
 * restores the word (its gathered value is the oldest) */
static void m3_undo(OrcEnc *e, uint32_t a)
{
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    uint32_t hi = M.ih;
    while (hi > a) {
        uint32_t lo = hi - a > 64 ? hi - 64 : a;
        for (uint32_t q = lo; q < hi; q++) {
            M3Rec *r = &M.rec[q];
            int f2 = 1, f3 = 1, f6 = 1;
            for (uint32_t j = lo; j < q; j++) {
                if (M.rec[j].h2 == r->h2) f2 = 0;
                if (M.rec[j].h3 == r->h3) f3 = 0;
                if (M.rec[j].h6 == r->h6) f6 = 0;
            }
            if (f2) e->ht2[r->h2] = r->s[0].ent;
            if (f3) e->ht3[r->h3] = r->s[1].ent;
            if (f6) for (uint32_t w = 0; w < W; w++) e->ht6[(size_t)r->h6 * e->ht_width + w] = r->s[2 + w].ent;
            M.n_undo_pos++;
        }
        hi = lo;
    }
}

/* the parser slid over [s + 1, s + len) instead of searching there */
static void m3_slide_event(OrcEnc *e, uint32_t s, uint32_t len)
{
    if (len <= 1) return;
    const uint32_t a = s + 1, b = s + len;
    M.n_slide++;
    if (len > 129) M.n_len_gt129++;
    if (M.ih <= a) {                                   /* nothing speculated there yet */
        m3_exact_slide(e, s, len, a, 0);
        M.ih = b; M.ch = b;
        return;
    }
    const uint32_t hi = UMIN(M.ih, b);
    int dev = len > 129;
    for (uint32_t q = a; q < hi && !dev; q++) if (M.rec[q].hdev) { dev = 1; M.n_hdev_events++; }
    if (!dev) {
        if (hi < b) {                                  /* the rest of the range: SlidePos itself, continuing its lasth6 */
            m3_exact_slide(e, s, len, hi, M.rec[hi - 1].h6);
            M.ih = b; M.ch = b;
        }
        return;                                        /* the speculative inserts ARE SlidePos's */
    }
    M.n_dev++;
    m3_undo(e, a);
    m3_exact_slide(e, s, len, a, 0);
    M.ih = b; M.ch = b;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* the parser                                                                                                        */
typedef struct { uint32_t price, dist, back, state; int id[4]; } M3Lab;
typedef struct { uint32_t price, dist, back; } M3Shadow;

static void m3_perm_ids(int *out, const int *in, uint32_t code, int newid)
{
    /* encode_nonlit's rep bookkeeping (csc_lz.cpp:127-154) on ids: code 1..4 = rep index + 1, else a new distance */
    if (code <= 4) {
        uint32_t k = code - 1;
        out[0] = in[k];
        for (uint32_t j = 1, t = 0; j < 4; t++) { if (t == k) continue; out[j++] = in[t]; }
    } else { out[0] = newid; out[1] = in[0]; out[2] = in[1]; out[3] = in[2]; }
}

/* ---- the way out's literal trees on a wavefront of their own (csc_kernels_dp4.inc: d6_literal / d6_trees / d6_join) ----
 * The kernel codes a literal's flag when the packet's turn comes and leaves its eight tree decisions to another wavefront, which
 * takes whatever records are there, up to eight at a time: one gather of the batch's cells, same-cell literals chained in registers
 * (the later one starts from the earlier one's updated probability, the last one stores), one scatter.  It waits for the trees only
 * where p_lit is read (a window's literal prices) and at the sub-block's end.  Shadowed here: the oracle codes the literal at once; a
 * second copy of p_lit takes the deferred batches (batch sizes 1..8 from a seeded generator: how many records a batch finds is a
 * matter of timing on the GPU, the result must not depend on it), and at every join every decision's probability and every touched
 * cell must equal the oracle's.  That is the whole argument: only literals touch p_lit, so their updates commute with every other
 * packet's. */
#define M3_LT_MAX 4096
static struct { uint32_t ctx, sym, pold[8]; } m3_lt[M3_LT_MAX];
static uint32_t m3_lt_n, m3_lt_seed = 12345u;
static uint32_t *m3_sh_plit;
static unsigned long long m3_lt_batches, m3_lt_chained, m3_lt_joins, m3_lt_max_pending;

static uint32_t m3_p_update(uint32_t bit, uint32_t p) { return bit ? p + ((0xFFF - p) >> 5) : p - (p >> 5); }

static void m3_join(OrcEnc *e)
{
    if (!m3_lt_n) return;
    m3_lt_joins++;
    if (m3_lt_n > m3_lt_max_pending) m3_lt_max_pending = m3_lt_n;
    for (uint32_t done = 0; done < m3_lt_n;) {
        m3_lt_seed = m3_lt_seed * 1664525u + 1013904223u;
        const uint32_t n = UMIN(m3_lt_n - done, 1u + ((m3_lt_seed >> 24) & 7u));
        uint32_t cell[8][8], bit[8][8], pold[8][8], pnew[8][8];
        for (uint32_t i = 0; i < n; i++) for (uint32_t k = 0; k < 8; k++) {                 /* the gather: every lane before any update */
            const uint32_t cc = m3_lt[done + i].sym | 0x100u;
            cell[i][k] = m3_lt[done + i].ctx * 256u + (cc >> (8u - k)); bit[i][k] = (cc >> (7u - k)) & 1u;
            pold[i][k] = m3_sh_plit[cell[i][k]];
        }
        for (uint32_t i = 0; i < n; i++) for (uint32_t k = 0; k < 8; k++) {                 /* chains in literal order; same cell => same level */
            int prev = -1;
            for (int j = (int)i - 1; j >= 0 && prev < 0; j--) if (cell[j][k] == cell[i][k]) prev = j;
            if (prev >= 0) { pold[i][k] = pnew[prev][k]; m3_lt_chained++; }
            pnew[i][k] = m3_p_update(bit[i][k], pold[i][k]);
            if (pold[i][k] != m3_lt[done + i].pold[k]) m3_die("a deferred tree decision saw another probability than the oracle's");
        }
        for (uint32_t i = 0; i < n; i++) for (uint32_t k = 0; k < 8; k++) {                 /* the scatter: the last literal on a cell stores */
            int later = 0;
            for (uint32_t j = i + 1; j < n; j++) if (cell[j][k] == cell[i][k]) later = 1;
            if (!later) m3_sh_plit[cell[i][k]] = pnew[i][k];
        }
        for (uint32_t i = 0; i < n; i++) for (uint32_t k = 0; k < 8; k++)
            if (cell[i][k] / 256u != m3_lt[done + i].ctx) m3_die("tree cell outside its context row");
        done += n;
        m3_lt_batches++;
    }
    for (uint32_t i = 0; i < m3_lt_n; i++) for (uint32_t k = 0; k < 8; k++) {
        const uint32_t c = m3_lt[i].ctx * 256u + ((m3_lt[i].sym | 0x100u) >> (8u - k));
        if (m3_sh_plit[c] != e->p_lit[c]) m3_die("p_lit after the deferred batches differs from the oracle's");
    }
    m3_lt_n = 0;
}

static void m3_literal(OrcEnc *e, uint32_t c)
{
    if (m3_lt_n == M3_LT_MAX) m3_join(e);
    const uint32_t cc = c | 0x100u;
    m3_lt[m3_lt_n].ctx = e->ctx; m3_lt[m3_lt_n].sym = c;
    for (uint32_t k = 0; k < 8; k++) m3_lt[m3_lt_n].pold[k] = e->p_lit[e->ctx * 256u + (cc >> (8u - k))];
    m3_lt_n++;
    encode_literal(e, c);
}

static void m3_adv(OrcEnc *e, uint32_t size)
{
    /* nothing is pending between sub-blocks; other block types may have coded literals meanwhile */
    if (!m3_sh_plit) m3_sh_plit = (uint32_t *)malloc(256 * 256 * sizeof(uint32_t));
    memcpy(m3_sh_plit, e->p_lit, 256 * 256 * sizeof(uint32_t));
    const uint32_t W = UMIN(e->ht_width, e->ht_cyc);
    if (e->bt_head || !e->ht_width || W > M3_NS - 2 || e->lz_good_len > M3_RING || e->lz_good_len < 2 || size == 0
        || e->pos >= 0xFFFF0000u) { lz_compress_advanced(e, size); return; }
    const uint32_t NS = 2 + W, vld = e->vld_rge;
    M.sb0 = e->wnd_curpos; M.pos0 = e->pos; M.size = size; M.ih = 0; M.ch = 0;
    const uint32_t sb_end = M.sb0 + size;
    M.rec = (M3Rec *)calloc(size, sizeof(M3Rec));
    M.nent = 0;
    /* the stream's rep distances get fresh masks at every sub-block start */
    for (int r = 0; r < 4; r++) M.rid[r] = m3_new_ent(e->rep_dist[r], M.sb0, e->rep_dist[r] < vld ? m3_eq_mask(e, M.sb0, e->rep_dist[r], sb_end) : 0, sb_end);
    static uint32_t fin_dist[AP_LIMIT + 2], fin_back[AP_LIMIT + 2];

    for (uint32_t i = 0; i < size;) {
        const uint32_t w0 = i, aplimit = UMIN((uint32_t)AP_LIMIT, size - i);
        /* per coder state, fixed while the window is open (nothing is coded inside it) */
        uint32_t lit_flag[64], p1flag[64], match_base[64], rep_price[64][4];
        for (uint32_t s = 0; s < 64; s++) {
            lit_flag[s] = bit_price(e, 0, e->p_state[s * 3]);
            p1flag[s] = rep0len1_price(e, s);
            match_base[s] = bit_price(e, 1, e->p_state[s * 3]) + bit_price(e, 1, e->p_state[s * 3 + 1]);
            for (uint32_t r = 0; r < 4; r++) rep_price[s][r] = rep_dist_price(e, s, r);
        }
        M3Lab ring[M3_RING + 1];
        for (int l = 0; l <= M3_RING; l++) ring[l].price = M3_INF;
        ring[0].price = 0; ring[0].dist = 0; ring[0].back = 0; ring[0].state = e->state;
        memcpy(ring[0].id, M.rid, sizeof(M.rid));
        uint32_t reach = 0, k = 0, a0l = 1, a0code = 0, exit_kind = 0;
        int a0id = -1;
        M3Lab cur;
        /* the spine + edges split of the kernel (csc_kernels_dp4.inc, d5_spine / d5_edges), shadowed: the match edges of the even
         * and of the odd nodes go into a ring each (strict `<` in node order inside a ring); node t's label = the better of the
         * two rings' entries by (price, source node), then node t - 1's length-1 edge if strictly better.  Asserted equal to the
         * label of the one ring at every node. */
        static M3Shadow sh[2][AP_LIMIT + M3_RING + 2], own[AP_LIMIT + M3_RING + 2];
        for (uint32_t t = 0; t < aplimit + M3_RING + 2; t++) { sh[0][t].price = sh[1][t].price = own[t].price = M3_INF; }
        M.n_windows++;
        for (;; k++) {
            const uint32_t p = w0 + k, wpos = M.sb0 + p, limit = size - p;
            cur = ring[0];
            fin_dist[k] = cur.dist; fin_back[k] = cur.back;
            if (k >= 1) {
                const M3Shadow *A = &sh[k & 1][k], *B = &sh[(k - 1) & 1][k];
                M3Shadow m = *A;
                if (B->price < m.price || (B->price == m.price && B->price != M3_INF && B->back < m.back)) m = *B;
                if (own[k].price < m.price) m = own[k];
                if (m.price != cur.price || m.dist != cur.dist || m.back != cur.back) m3_die("spine/edge merge differs from the single ring");
                M.n_split_checked++;
            }
            M.n_nodes++;
            if (k == aplimit) { exit_kind = 1; break; }                       /* :271 */
            m3_ensure(e, p);
            const M3Rec *r = &M.rec[p];
            /* ---- candidates: 4 rep slots from masks, hash slots from the record ---- */
            uint32_t L[4 + M3_NS], code[4 + M3_NS], dprice[4 + M3_NS], cdist[4 + M3_NS];
            int cid[4 + M3_NS];
            uint8_t cons[4 + M3_NS], drop[4 + M3_NS];
            for (uint32_t s = 0; s < 4; s++) {
                /* a refreshed mask that has arrived replaces the id in the label (and so in everything relaxed from it) */
                if (M.ent[cur.id[s]].fwd >= 0 && p >= M.ent[cur.id[s]].req_at + (uint32_t)M.refresh_delay) cur.id[s] = M.ent[cur.id[s]].fwd;
                M3Ent *x = &M.ent[cur.id[s]];
                cons[s] = x->dist < vld; drop[s] = 0; L[s] = 0; code[s] = 1 + s; cid[s] = cur.id[s]; cdist[s] = 0;
                dprice[s] = rep_price[cur.state][s];
                if (cons[s]) {
                    int slow;
                    L[s] = m3_rep_len(e, x, wpos, limit, &slow);
                    if (slow) M.n_slow++;
                    if (wpos - x->base > (uint32_t)M.refresh_at && x->fwd < 0 && x->cov_end > wpos) {   /* ask for a mask based here */
                        x->fwd = m3_new_ent(x->dist, wpos, m3_eq_mask(e, wpos, x->dist, sb_end), sb_end);
                        x = &M.ent[cur.id[s]];     /* (the table may have moved) */
                        x->req_at = p;
                        M.n_refresh++;
                    }
                }
            }
            for (uint32_t s = 0; s < NS; s++) {
                const M3Slot *q = &r->s[s];
                cons[4 + s] = q->cons; drop[4 + s] = q->drop; L[4 + s] = q->cons ? q->ml : 0;
                code[4 + s] = 4 + q->dist; cid[4 + s] = q->eid; cdist[4 + s] = q->dist;
                dprice[4 + s] = match_base[cur.state] + q->dcost;
            }
            /* ---- find_match's acceptance as a prefix maximum (:266-485) ---- */
            uint32_t pm = 0, premax = 1, has1 = (cons[0] && L[0] >= 2) ? 1u : 0u, n = 0;
            for (uint32_t s = 0; s < 4 + NS; s++) {
                const uint32_t Ls = cons[s] ? L[s] : 0;
                if (cons[s] && Ls > premax && !drop[s]) pm |= 1u << s;
                if (Ls > premax) premax = Ls;
                if ((pm >> s) & 1u) if (Ls >= e->good_len) break;            /* everything after the first good_len hit is ignored */
            }
            n = (uint32_t)__builtin_popcount(pm) + has1;
            {
                /* what the kernel relies on: (a) no rep length >= 2 => the node's acceptance IS the hash slots' own; (b) otherwise the
                 * hash candidates pushed are the hash-only ones longer than the longest rep length Rmax, and every length above Rmax
                 * keeps the owner the hash-only table gives it (lengths up to Rmax belong to rep candidates) */
                uint32_t Rmax = 1;
                for (uint32_t s2 = 0; s2 < 4; s2++) if (cons[s2] && L[s2] > Rmax) Rmax = L[s2];
                int repkill = 0;
                for (uint32_t s2 = 0; s2 < 4; s2++) if (((pm >> s2) & 1u) && L[s2] >= e->good_len) repkill = 1;
                if (Rmax < 2 && (pm >> 4) != r->pm_h) m3_die("hash-only acceptance differs without rep candidates");
                if (!repkill) {
                    uint32_t expect = 0;
                    for (uint32_t s2 = 0; s2 < NS; s2++) if (((r->pm_h >> s2) & 1u) && r->s[s2].ml > Rmax) expect |= 1u << s2;
                    if ((pm >> 4) != expect) m3_die("pushed hash candidates are not the hash-only ones above the longest rep length");
                }
            }
            { uint32_t rm = 0; for (uint32_t s2 = 0; s2 < 4; s2++) if (cons[s2] && L[s2] > rm) rm = L[s2];
              if (rm >= 2) M.n_repnodes++; else { int nh = __builtin_popcount(pm >> 4); if (nh == 0) M.n_h0++; else if (nh == 1) M.n_h1++; else if (nh == 2) M.n_h2++; else M.n_h3p++; } }
            a0l = 1; a0code = 0; a0id = -1;
            if (pm) { uint32_t top = 31u - (uint32_t)__builtin_clz(pm); a0l = L[top]; a0code = code[top]; a0id = cid[top]; }
            else if (has1) { a0l = 1; a0code = 1; }
            if (k == 0 && n == 0) { exit_kind = 4; break; }                    /* :213: no candidate at all -> plain literal */
            /* ---- FindMatchWithPrice's table (:600-624): length l in "lane" l ---- */
            uint32_t lane_code[M3_RING + 1] = {0}, lane_price[M3_RING + 1] = {0};
            int lane_id[M3_RING + 1];
            if (a0l < e->good_len) {
                uint32_t prevL = 1, nt = 0;
                uint32_t own[M3_RING + 1];
                for (uint32_t s = 0; s < 4 + NS; s++) {
                    if (!((pm >> s) & 1u)) continue;
                    for (uint32_t l = prevL + 1; l <= L[s]; l++) {
                        own[l] = s;
                        if (s >= 4 && l <= M3_RING && !(l <= 6 && cdist[s] >= kBound[l]) && r->len_owner[l] != s - 4 + 1) m3_die("length owner differs from the hash-only table");
                        const uint32_t rdist = s < 4 ? 0 : cdist[s];
                        lane_code[l] = (l <= 6 && rdist >= kBound[l]) ? 0 : code[s];
                        lane_id[l] = cid[s];
                        if (lane_code[l]) nt++;                                  /* GetMatchLenPrice ticks (l - 2 < 32 always here) */
                    }
                    prevL = L[s];
                }
                /* the 4097-call refresh of the length prices (csc_model.cpp:286-299), calls numbered in length order */
                uint32_t old_lp[32], trigger = 0xFFFFFFFFu;
                if (nt && e->lp_rebuild_int < nt) {
                    memcpy(old_lp, e->len_price, sizeof(old_lp));
                    trigger = e->lp_rebuild_int;
                    len_price_rebuild(e);
                    e->lp_rebuild_int = 4096 - (nt - trigger - 1);
                } else e->lp_rebuild_int -= nt;
                uint32_t idx = 0;
                for (uint32_t l = 2; l <= a0l; l++) {
                    if (!lane_code[l]) continue;
                    const uint32_t lp = (trigger != 0xFFFFFFFFu && idx < trigger) ? old_lp[l - 2] : e->len_price[l - 2];
                    lane_price[l] = dprice[own[l]] + lp;
                    idx++;
                }
            }
            /* ---- ways out (:277, :290) ---- */
            if (a0l == 1 && reach == k) { exit_kind = 2; break; }
            if (a0l >= e->lz_good_len || (a0l > 1 && a0l + k >= aplimit)) { exit_kind = 3; break; }
            if (k + a0l > reach) reach = k + a0l;
            /* ---- relaxation: lane 1 literal then rep0len1 (:301-313), lanes 2.. the match lengths (:315-326) ---- */
            {
                const uint32_t lit_ctx = wpos ? e->wnd[wpos - 1] : 0;
                m3_join(e);                                                     /* (a window's literal prices read p_lit) */
                const uint32_t tree = literal_price(e, cur.state, lit_ctx, e->wnd[wpos]) - lit_flag[cur.state];
                uint32_t c1 = tree + lit_flag[cur.state] + cur.price, k1 = 0;
                if (has1 && p1flag[cur.state] + cur.price < c1) { c1 = p1flag[cur.state] + cur.price; k1 = 1; }
                own[k + 1].price = c1; own[k + 1].dist = k1; own[k + 1].back = k;
                if (c1 < ring[1].price) {
                    ring[1].price = c1; ring[1].dist = k1; ring[1].back = k;
                    ring[1].state = (cur.state * 4 + (k1 ? 2u : 0u)) & 0x3F;
                    memcpy(ring[1].id, cur.id, sizeof(cur.id));
                }
            }
            for (uint32_t l = 2; l <= a0l; l++) {
                if (!lane_code[l]) continue;
                const uint32_t np = lane_price[l] + cur.price;
                if (np < sh[k & 1][k + l].price) { sh[k & 1][k + l].price = np; sh[k & 1][k + l].dist = lane_code[l]; sh[k & 1][k + l].back = k; }
                if (np < ring[l].price) {
                    ring[l].price = np; ring[l].dist = lane_code[l]; ring[l].back = k;
                    ring[l].state = (cur.state * 4 + (lane_code[l] <= 4 ? 3u : 1u)) & 0x3F;
                    m3_perm_ids(ring[l].id, cur.id, lane_code[l], lane_id[l]);
                }
            }
            /* ---- the ring moves on by one node ---- */
            memmove(&ring[0], &ring[1], sizeof(M3Lab) * M3_RING);
            ring[M3_RING].price = M3_INF;
        }
        /* ---- way out: back-trace over the log (csc_lz.cpp:335-362), then what the exit itself codes ---- */
        const uint32_t end = k;
        if (exit_kind == 4) {
            m3_literal(e, e->wnd[M.sb0 + w0]);
            M.n_direct_lit++;
            i++;
            continue;
        }
        {
            static uint32_t nxt[AP_LIMIT + 2];
            for (uint32_t t = end; t;) { nxt[fin_back[t]] = t; t = fin_back[t]; }
            for (uint32_t t = 0; t != end;) {
                const uint32_t next = nxt[t], nd = fin_dist[next];
                if (nd == 0) m3_literal(e, e->wnd[M.sb0 + w0 + t]);
                else if (nd <= 4) {
                    if (next - t == 1 && nd == 1) encode_rep0len1(e);
                    else encode_rep_match(e, nd - 1, next - t - 2);
                    e->ctx = e->wnd[M.sb0 + w0 + next - 1];
                } else {
                    encode_match(e, nd - 5, next - t - 2);
                    e->ctx = e->wnd[M.sb0 + w0 + next - 1];
                }
                t = next;
            }
            /* rep_dist_ = ap[end].rep_dist (:358-361): the ids of the exit node's label, distances from their entries */
            memcpy(M.rid, cur.id, sizeof(cur.id));
            for (int r = 0; r < 4; r++) e->rep_dist[r] = M.ent[M.rid[r]].dist;
        }
        i += end;
        if (exit_kind == 2) {
            m3_literal(e, e->wnd[M.sb0 + i]);
            i++;
        } else if (exit_kind == 3) {
            MFUnit u; u.len = a0l; u.dist = a0code;
            int nid[4];
            if (!(a0l == 1 && a0code == 1)) { m3_perm_ids(nid, M.rid, a0code, a0id); memcpy(M.rid, nid, sizeof(nid)); }
            lz_encode_nonlit(e, u);
            for (int r = 0; r < 4; r++) if (e->rep_dist[r] != M.ent[M.rid[r]].dist) m3_die("rep ids and rep distances disagree");
            m3_slide_event(e, i, a0l);
            i += a0l;
            e->ctx = e->wnd[M.sb0 + i - 1];
        }
    }
    m3_join(e);
    /* whatever the parser never reached was inserted speculatively beyond the sub-block?  No: the pre-pass stops at `size` */
    if (M.ih < size) m3_ensure(e, size - 1);
    if (M.ih != size) m3_die("inserter did not end at the sub-block end");
    e->pos = M.pos0 + size;
    e->wnd_curpos = M.sb0 + size;
    free(M.rec); M.rec = NULL;
}

static void m3_stats_atexit(void)
{
    if (!getenv("M3_STATS")) return;
    fprintf(stderr, "m3_model: nodes %llu windows %llu direct-literals %llu | slide events %llu (len>129: %llu, same-hash: %llu) deviations %llu undone positions %llu exact positions %llu | "
            "batches %llu | mask refreshes %llu slow-path rep compares %llu | nodes with a rep length >= 2: %llu, else hash candidates pushed 0/1/2/3+: %llu/%llu/%llu/%llu | spine/edge merges checked %llu\n",
            M.n_nodes, M.n_windows, M.n_direct_lit, M.n_slide, M.n_len_gt129, M.n_hdev_events, M.n_dev, M.n_undo_pos, M.n_exact_pos, M.n_batches, M.n_refresh, M.n_slow, M.n_repnodes, M.n_h0, M.n_h1, M.n_h2, M.n_h3p, M.n_split_checked);
    fprintf(stderr, "m3_model: literal trees: joins %llu, deferred batches %llu, chained decisions %llu, most records pending at a join %llu\n",
            m3_lt_joins, m3_lt_batches, m3_lt_chained, m3_lt_max_pending);
}

__attribute__((constructor)) static void m3_install(void)
{
    const char *s;
    M.la = (s = getenv("M3_LA")) ? atoi(s) : 192;
    M.refresh_delay = (s = getenv("M3_REFRESH_DELAY")) ? atoi(s) : 6;
    M.refresh_at = (s = getenv("M3_REFRESH_AT")) ? atoi(s) : 16;
    if (M.la < 1) M.la = 1;
    orc_adv_hook = m3_adv;
    atexit(m3_stats_atexit);
}
