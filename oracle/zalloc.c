/*
 * oracle/zalloc.c -- deterministic ISzAlloc implementations for the tests.
 * TEST INFRASTRUCTURE.  The reference leaves one byte per coder flush
 * un-stored (csc_coder.cpp:46-47), so goldens are only reproducible with an
 * allocator that hands out memory of known content.
 */
#include <stdlib.h>
#include <string.h>
#include "orc_api.h"

static void *zero_alloc_fn(void *p, size_t n) { (void)p; return calloc(1, n ? n : 1); }
static void any_free_fn(void *p, void *a) { (void)p; free(a); }
static ISzAlloc g_zero = {zero_alloc_fn, any_free_fn};

static void *aa_alloc_fn(void *p, size_t n)
{
    (void)p;
    void *m = malloc(n ? n : 1);
    if (m) memset(m, 0xAA, n);
    return m;
}
static ISzAlloc g_aa = {aa_alloc_fn, any_free_fn};

ISzAlloc *orc_zero_alloc(void) { return &g_zero; }
ISzAlloc *orc_aa_alloc(void) { return &g_aa; }
