/*
 * corpus.c -- seeded synthetic stand-ins for the corpora BASELINE.json names (enwik8/9,
 * silesia.tar, EXE/text/delta mixes); none of the real files exist in this environment.
 *
 * Deterministic and random-access: content is produced in independent 64 KiB pages, page k of
 * stream (kind, seed) depending only on (kind, seed, k), so any byte range -- e.g. task slice r
 * of the archiver's -p split (csarc.cpp:532-543) -- can be produced without the bytes before it.
 * PRNG: xoshiro256** seeded through splitmix64 (never libc rand).
 *
 * Kinds:
 *   0 T  enwik-like: XML-ish wiki pages, Zipf(1.1) words from a fixed 4096-word vocabulary, links,
 *        entities, a pool of recurring template sentences (long-distance repeats)
 *   1 X  EXE-like: x86-ish opcode soup with E8 rel32 calls to recurring targets, 8B/00-heavy
 *   2 D  delta tables: interleaved little-endian random-walk columns (2/4/8-byte records)
 *   3 R  uniform random bytes
 *   4 E  8-symbol uniform alphabet
 *   5 S  "silesia-like": T 25 %, X 25 %, D 10 %, low-entropy structured 38 %, R/E 2 % (by page)
 *   6 M  config-5 mix: X | T | D in alternating 64 MiB segments
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PAGE 65536u

typedef struct { uint64_t s[4]; } Rng;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void rng_seed(Rng *r, uint64_t seed)
{
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(Rng *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline uint32_t rng_below(Rng *r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static inline double rng_unit(Rng *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* ---------------------------------------------------------------- vocabulary */
#define NWORDS 4096
#define NTEMPL 256
static char g_words[NWORDS][16];
static uint8_t g_wlen[NWORDS];
static double g_cdf[NWORDS];
static char g_templ[NTEMPL][200];
static uint16_t g_tlen[NTEMPL];
static int g_ready = 0;

static const char *const kCommon[] = {
    "the","of","and","in","to","a","is","was","for","as","on","with","by","that","it","from","at","his","an","he",
    "which","are","this","be","or","were","also","has","had","its","not","but","first","one","their","have","new",
    "after","who","they","two","her","been","other","when","there","all","during","into","time","she","may","more",
    "school","years","city","over","only","world","would","where","later","most","these","about","up","state",
    "between","national","such","united","states","war","under","then","made","some","born","three","year","known",
    "used","university","part","than","became","american","history","many","south","county","north","people","out",
    "season","however","can","while","work","john","early","before","team","being","family","both","population",
    "through","area","life","government","second","well","including","film","music","so","name","several","century",
    "series","since","four","house","company","until","called","international","member","against","district","group",
    "river","album","began","following","general","number","west","high","played","end","because","league","east",
    "college","church","same","each","no","public","party","no","them","home","game","him","along","although",
    "that","said","with","have","this","from","were","tion","station","nation","information","section","population",
};

static void build_vocab(void)
{
    if (g_ready) return;
    Rng r;
    rng_seed(&r, 0xC5C0FFEEull);
    static const char *const on[] = {"b","c","d","f","g","h","j","k","l","m","n","p","r","s","t","v","w","st","tr","ch","sh","th","pl","gr","br","cr","fl","pr","qu","sp"};
    static const char *const nu[] = {"a","e","i","o","u","a","e","i","o","ea","ou","ai","ee","oo","ie","io"};
    static const char *const co[] = {"","","n","r","s","t","l","m","d","ng","nt","st","rd","ll","ss","ck"};
    int ncommon = (int)(sizeof(kCommon) / sizeof(kCommon[0]));
    for (int i = 0; i < NWORDS; i++) {
        char *w = g_words[i];
        if (i < ncommon) {
            strncpy(w, kCommon[i], 15);
            w[15] = 0;
        } else {
            int syl = 1 + (int)rng_below(&r, 3) + (i > 1500 ? 1 : 0);
            w[0] = 0;
            for (int s = 0; s < syl; s++) {
                char tmp[16];
                snprintf(tmp, sizeof(tmp), "%s%s%s", on[rng_below(&r, 30)], nu[rng_below(&r, 16)], co[rng_below(&r, 16)]);
                if (strlen(w) + strlen(tmp) < 15) strcat(w, tmp);
            }
        }
        g_wlen[i] = (uint8_t)strlen(w);
    }
    double tot = 0;
    for (int i = 0; i < NWORDS; i++) { tot += 1.0 / pow((double)(i + 1), 1.1); g_cdf[i] = tot; }
    for (int i = 0; i < NWORDS; i++) g_cdf[i] /= tot;
    /* recurring template sentences */
    for (int t = 0; t < NTEMPL; t++) {
        char *p = g_templ[t];
        int len = 0, nw = 6 + (int)rng_below(&r, 18);
        for (int k = 0; k < nw && len < 180; k++) {
            double u = rng_unit(&r);
            int lo = 0, hi = NWORDS - 1;
            while (lo < hi) { int mid = (lo + hi) / 2; if (g_cdf[mid] < u) lo = mid + 1; else hi = mid; }
            memcpy(p + len, g_words[lo], g_wlen[lo]);
            len += g_wlen[lo];
            p[len++] = ' ';
        }
        p[len - 1] = '.';
        g_tlen[t] = (uint16_t)len;
    }
    g_ready = 1;
}

static inline int zipf_word(Rng *r)
{
    double u = rng_unit(r);
    int lo = 0, hi = NWORDS - 1;
    while (lo < hi) { int mid = (lo + hi) / 2; if (g_cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { uint8_t *p; uint32_t n, cap; } Out;
static inline void put(Out *o, const char *s, uint32_t len)
{
    if (o->n + len > o->cap) len = o->cap - o->n;
    memcpy(o->p + o->n, s, len);
    o->n += len;
}
static inline void puts_(Out *o, const char *s) { put(o, s, (uint32_t)strlen(s)); }
static void put_word(Out *o, Rng *r, int cap)
{
    int w = zipf_word(r);
    char tmp[16];
    memcpy(tmp, g_words[w], g_wlen[w]);
    if (cap && tmp[0] >= 'a' && tmp[0] <= 'z') tmp[0] = (char)(tmp[0] - 32);
    put(o, tmp, g_wlen[w]);
}

/* ---------------------------------------------------------------- page generators */
static void page_text(uint8_t *out, Rng *r, uint64_t page_no)
{
    Out o = {out, 0, PAGE};
    char num[64];
    while (o.n < PAGE) {
        puts_(&o, "  <page>\n    <title>");
        put_word(&o, r, 1); puts_(&o, " "); put_word(&o, r, 1);
        puts_(&o, "</title>\n    <id>");
        snprintf(num, sizeof(num), "%u", (unsigned)(page_no * 37 + rng_below(r, 100000)));
        puts_(&o, num);
        puts_(&o, "</id>\n    <revision>\n      <timestamp>20");
        snprintf(num, sizeof(num), "%02u-%02u-%02uT%02u:%02u:%02uZ", 2 + rng_below(r, 5), 1 + rng_below(r, 12), 1 + rng_below(r, 28),
                 rng_below(r, 24), rng_below(r, 60), rng_below(r, 60));
        puts_(&o, num);
        puts_(&o, "</timestamp>\n      <text xml:space=\"preserve\">");
        int paras = 2 + (int)rng_below(r, 8);
        for (int p = 0; p < paras && o.n < PAGE; p++) {
            uint32_t kind = rng_below(r, 100);
            if (kind < 8) { puts_(&o, "\n== "); put_word(&o, r, 1); puts_(&o, " "); put_word(&o, r, 0); puts_(&o, " ==\n"); }
            int sents = 1 + (int)rng_below(r, 7);
            uint32_t line = 0;
            for (int s = 0; s < sents && o.n < PAGE; s++) {
                if (rng_below(r, 100) < 6) {   /* recurring template sentence: long-distance repeat */
                    int t = (int)rng_below(r, NTEMPL);
                    put(&o, g_templ[t], g_tlen[t]);
                    puts_(&o, " ");
                    line += g_tlen[t];
                    continue;
                }
                int nw = 5 + (int)rng_below(r, 20);
                for (int k = 0; k < nw && o.n < PAGE; k++) {
                    uint32_t m = rng_below(r, 1000);
                    uint32_t before = o.n;
                    if (m < 18) { puts_(&o, "[["); put_word(&o, r, 1); if (m < 8) { puts_(&o, "|"); put_word(&o, r, 0); } puts_(&o, "]]"); }
                    else if (m < 22) puts_(&o, "&amp;");
                    else if (m < 28) { puts_(&o, "&quot;"); put_word(&o, r, 0); puts_(&o, "&quot;"); }
                    else if (m < 33) { puts_(&o, "''"); put_word(&o, r, 0); puts_(&o, "''"); }
                    else if (m < 40) { snprintf(num, sizeof(num), "%u", 1 + rng_below(r, 2100)); puts_(&o, num); }
                    else put_word(&o, r, k == 0);
                    line += o.n - before + 1;
                    if (k + 1 < nw) {
                        if (m % 17 == 3) puts_(&o, ", ");
                        else if (line > 60 + (m % 60)) { puts_(&o, "\n"); line = 0; }
                        else puts_(&o, " ");
                    }
                }
                puts_(&o, rng_below(r, 10) ? ". " : ".\n");
            }
            puts_(&o, "\n\n");
            if (kind >= 92) {
                int items = 2 + (int)rng_below(r, 5);
                for (int k = 0; k < items; k++) { puts_(&o, "* [["); put_word(&o, r, 1); puts_(&o, "]] - "); put_word(&o, r, 0); puts_(&o, " "); put_word(&o, r, 0); puts_(&o, "\n"); }
            }
            if (kind >= 97) { puts_(&o, "{{"); put_word(&o, r, 1); puts_(&o, "|"); put_word(&o, r, 0); puts_(&o, "="); put_word(&o, r, 0); puts_(&o, "}}\n"); }
        }
        puts_(&o, "</text>\n    </revision>\n  </page>\n");
    }
}

static void page_exe(uint8_t *out, Rng *r, uint64_t page_no, uint64_t seed)
{
    /* recurring absolute call targets shared by the whole stream */
    Rng tr;
    rng_seed(&tr, seed ^ 0xE8E8E8E8ull);
    static uint32_t targets[4096];
    for (int i = 0; i < 4096; i++) targets[i] = (uint32_t)(rng_next(&tr) % (64u << 20));
    uint32_t n = 0;
    uint64_t base = page_no * PAGE;
    static const uint8_t regs[8] = {0x45, 0x4D, 0x55, 0x5D, 0x44, 0x4C, 0x54, 0x05};
    while (n < PAGE) {
        uint32_t m = rng_below(r, 100);
        uint8_t tmp[8];
        uint32_t len;
        if (m < 14) {   /* call rel32 */
            uint32_t t = targets[zipf_word(r) & 4095];
            uint32_t rel = t - (uint32_t)(base + n + 5);
            tmp[0] = 0xE8; tmp[1] = (uint8_t)rel; tmp[2] = (uint8_t)(rel >> 8); tmp[3] = (uint8_t)(rel >> 16); tmp[4] = (uint8_t)(rel >> 24);
            len = 5;
        } else if (m < 40) { tmp[0] = 0x8B; tmp[1] = regs[rng_below(r, 8)]; tmp[2] = (uint8_t)(rng_below(r, 16) * 4 - 32); len = 3; }
        else if (m < 52) { tmp[0] = 0x89; tmp[1] = regs[rng_below(r, 8)]; tmp[2] = (uint8_t)(rng_below(r, 16) * 4); len = 3; }
        else if (m < 60) { tmp[0] = 0x00; tmp[1] = 0x00; len = 2; }
        else if (m < 68) { tmp[0] = 0x83; tmp[1] = 0xC4; tmp[2] = (uint8_t)(rng_below(r, 8) * 4); len = 3; }
        else if (m < 74) { tmp[0] = 0x0F; tmp[1] = (uint8_t)(0x84 + rng_below(r, 2)); tmp[2] = (uint8_t)rng_below(r, 200); tmp[3] = 0; tmp[4] = 0; tmp[5] = 0; len = 6; }
        else if (m < 80) { tmp[0] = 0x50 + (uint8_t)rng_below(r, 8); len = 1; }
        else if (m < 85) { tmp[0] = 0xC3; tmp[1] = 0x90; tmp[2] = 0x90; len = 3; }
        else if (m < 92) { tmp[0] = 0xB8 + (uint8_t)rng_below(r, 8); tmp[1] = (uint8_t)rng_below(r, 64); tmp[2] = 0; tmp[3] = 0; tmp[4] = 0; len = 5; }
        else { tmp[0] = (uint8_t)rng_next(r); tmp[1] = (uint8_t)rng_next(r); len = 2; }
        if (n + len > PAGE) len = PAGE - n;
        memcpy(out + n, tmp, len);
        n += len;
    }
}

static void page_delta(uint8_t *out, Rng *r, uint64_t page_no)
{
    /* record width cycles with the page number: 2, 4, 8 bytes; columns random-walk (sigma ~ 3) */
    static const uint32_t widths[3] = {2, 4, 8};
    uint32_t w = widths[page_no % 3];
    int32_t col[8];
    for (uint32_t c = 0; c < w; c++) col[c] = (int32_t)rng_below(r, 256);
    for (uint32_t i = 0; i < PAGE; i += w)
        for (uint32_t c = 0; c < w && i + c < PAGE; c++) {
            int32_t step = (int32_t)rng_below(r, 7) - 3 + (int32_t)(rng_below(r, 16) == 0 ? (int32_t)rng_below(r, 9) - 4 : 0);
            col[c] += step;
            out[i + c] = (uint8_t)col[c];
        }
}

static void page_random(uint8_t *out, Rng *r)
{
    for (uint32_t i = 0; i < PAGE; i += 8) { uint64_t v = rng_next(r); memcpy(out + i, &v, 8); }
}
static void page_entropy8(uint8_t *out, Rng *r)
{
    static const uint8_t sym[8] = {3, 50, 99, 140, 200, 7, 64, 250};
    for (uint32_t i = 0; i < PAGE; i++) out[i] = sym[rng_below(r, 8)];
}
static void page_structured(uint8_t *out, Rng *r, uint64_t page_no)
{
    /* low-entropy structured records: fixed 32-byte rows with slowly varying fields */
    uint32_t id = (uint32_t)(page_no * 2048);
    for (uint32_t i = 0; i < PAGE; i += 32) {
        uint8_t row[32];
        memset(row, 0, 32);
        row[0] = 'R'; row[1] = 'C';
        memcpy(row + 2, &id, 4); id++;
        uint32_t v = rng_below(r, 16); memcpy(row + 8, &v, 4);
        row[12] = (uint8_t)('A' + rng_below(r, 6));
        uint32_t f = 1000 + rng_below(r, 64); memcpy(row + 16, &f, 4);
        row[31] = '\n';
        memcpy(out + i, row, 32);
    }
}

static void gen_page(int kind, uint64_t seed, uint64_t page_no, uint8_t *out)
{
    Rng r;
    rng_seed(&r, seed * 0x9E3779B97F4A7C15ull + page_no * 0xD1B54A32D192ED03ull + (uint64_t)kind);
    switch (kind) {
    case 0: page_text(out, &r, page_no); break;
    case 1: page_exe(out, &r, page_no, seed); break;
    case 2: page_delta(out, &r, page_no); break;
    case 3: page_random(out, &r); break;
    case 4: page_entropy8(out, &r); break;
    case 5: {
        uint32_t m = (uint32_t)((page_no * 2654435761ull >> 7) % 100);
        if (m < 25) page_text(out, &r, page_no);
        else if (m < 50) page_exe(out, &r, page_no, seed);
        else if (m < 60) page_delta(out, &r, page_no);
        else if (m < 98) page_structured(out, &r, page_no);
        else if (m < 99) page_random(out, &r);
        else page_entropy8(out, &r);
    } break;
    case 6: {
        uint64_t seg = (page_no * PAGE) / (64ull << 20);
        int k = (int)(seg % 3);
        if (k == 0) page_exe(out, &r, page_no, seed);
        else if (k == 1) page_text(out, &r, page_no);
        else page_delta(out, &r, page_no);
    } break;
    default: memset(out, 0, PAGE);
    }
}

/* Fill out[0..n) with bytes [offset, offset+n) of stream (kind, seed). */
void csc_corpus_fill(int kind, uint64_t seed, uint64_t offset, uint8_t *out, uint64_t n)
{
    build_vocab();
    uint8_t *page = (uint8_t *)malloc(PAGE);
    uint64_t done = 0;
    while (done < n) {
        uint64_t pos = offset + done, page_no = pos / PAGE, in = pos % PAGE;
        uint64_t take = PAGE - in;
        if (take > n - done) take = n - done;
        gen_page(kind, seed, page_no, page);
        memcpy(out + done, page + in, take);
        done += take;
    }
    free(page);
}
