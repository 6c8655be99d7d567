/*
 * huf.c -- oracle (test infrastructure only): histogram + Huffman coder of the reference.
 * Restates src/utils.rs:4-16 (count_freqs) and src/huf.rs (build / Enc / Dec / BinTrie /
 * serialisation).  Item order = ascending symbol key, ties between equally rare subtrees by one fixed
 * rule (both deviation D1: the reference's own order is a HashMap's, random per process).
 */
#include "cniic_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- utils.rs:4-16 count_freqs ---------------- */

static void radix_sort_u32(uint32_t *a, uint32_t *tmp, uint64_t n) {
    /* 3 passes x 11 bits */
    for (int pass = 0; pass < 3; pass++) {
        uint64_t cnt[2049];
        memset(cnt, 0, sizeof cnt);
        int sh = pass * 11;
        for (uint64_t i = 0; i < n; i++) cnt[((a[i] >> sh) & 2047) + 1]++;
        for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
        for (uint64_t i = 0; i < n; i++) tmp[cnt[(a[i] >> sh) & 2047]++] = a[i];
        uint32_t *t = a; a = tmp; tmp = t;
    }
    /* 3 swaps: result is in the buffer originally called tmp; copy back */
    memcpy(tmp, a, n * sizeof(uint32_t));
}

int orc_count_freqs(const uint32_t *syms, uint64_t n, uint32_t *keys, uint64_t *counts,
                    uint64_t cap, uint64_t *n_unique) {
    *n_unique = 0;
    if (n == 0) return ORC_OK;
    uint32_t *s = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *t = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!s || !t) { free(s); free(t); return ORC_ERR_NOMEM; }
    memcpy(s, syms, n * sizeof(uint32_t));
    radix_sort_u32(s, t, n);
    uint64_t u = 0;
    int rc = ORC_OK;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && s[j] == s[i]) j++;
        if (u >= cap) { rc = ORC_ERR_CAPACITY; break; }
        keys[u] = s[i];
        counts[u] = j - i;   /* HashMap<T,u64> entry: +1 per occurrence */
        u++;
        i = j;
    }
    free(s); free(t);
    *n_unique = u;
    return rc;
}

/* ---------------- huf.rs:58-117 build ---------------- */
/*
 * The reference pops the two rarest subtrees off a BinaryHeap and pushes their parent (left = first popped,
 * right = second popped, huf.rs:100-110).  WHICH of several equally rare subtrees comes first is decided by the
 * heap's array mechanics applied to the iteration order of a std HashMap (huf.rs:30-31, 96) -- random per process:
 * the reference itself does not produce the same tree twice, and nothing depends on it (every such tree has the
 * same cost, huf.rs:22-43 serialises whichever it built).  The oracle and the product therefore fix ONE rule
 * (deviation D1, like the ascending-key item order):
 *     the rarest subtree first; among equally rare ones a leaf before a branch, leaves by ascending symbol key,
 *     branches in the order they were made.
 * With leaves sorted by (count, key) and branches appended as they are made (their counts never decrease), that is the
 * classic two-queue merge: no heap at all.
 */
typedef struct {
    uint64_t freq;
    uint32_t node;
} suffix_t; /* huf.rs:63-66 */

static int suffix_cmp(const void *pa, const void *pb) {
    const suffix_t *a = (const suffix_t *)pa, *b = (const suffix_t *)pb;
    if (a->freq != b->freq) return a->freq < b->freq ? -1 : 1;
    return a->node < b->node ? -1 : a->node > b->node ? 1 : 0;
}

/* BinTrie (huf.rs:167-171) as arrays: nodes 0..n-1 are leaves (symbol index = node id),
 * nodes n..2n-2 are branches with left[]/right[] children. */
typedef struct {
    uint64_t nleaf;
    uint32_t *left, *right; /* indexed by node - nleaf */
    uint32_t root;
} trie_t;

static int trie_build(const uint64_t *counts, uint64_t n, trie_t *t) {
    if (n == 0 || n > 0x7fffffffu) return ORC_ERR_BAD_ARG; /* huf.rs:99 assert!(len > 0) */
    t->nleaf = n;
    t->left = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    t->right = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    suffix_t *leaf = (suffix_t *)malloc(n * sizeof(suffix_t));
    uint64_t *bfreq = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!t->left || !t->right || !leaf || !bfreq) { free(t->left); free(t->right); free(leaf); free(bfreq); return ORC_ERR_NOMEM; }
    for (uint64_t i = 0; i < n; i++) { leaf[i].freq = counts[i]; leaf[i].node = (uint32_t)i; }
    qsort(leaf, n, sizeof(suffix_t), suffix_cmp); /* (count, key): node ids are ascending keys */
    uint64_t li = 0, bi = 0, made = 0;
    while (made + 1 < n) { /* huf.rs:100-110 */
        suffix_t two[2];
        for (int k = 0; k < 2; k++) {
            if (li < n && (bi >= made || leaf[li].freq <= bfreq[bi])) two[k] = leaf[li++];
            else { two[k].freq = bfreq[bi]; two[k].node = (uint32_t)(n + bi); bi++; }
        }
        t->left[made] = two[0].node;
        t->right[made] = two[1].node;
        bfreq[made] = two[0].freq + two[1].freq;
        made++;
    }
    t->root = n > 1 ? (uint32_t)(n + made - 1) : 0;
    free(leaf);
    free(bfreq);
    return ORC_OK;
}

static void trie_free(trie_t *t) { free(t->left); free(t->right); }

/* BinTrieIter (huf.rs:209-292): leaves left-to-right with their bit strings (0 = left).
 * Iterative DFS; code bits collected MSB-first into a u64 (depth <= 64 asserted by caller). */
static int trie_codes(const trie_t *t, uint32_t *lens, uint64_t *codes) {
    uint64_t n = t->nleaf;
    /* explicit stack of (node, len, code) */
    typedef struct { uint32_t node; uint32_t len; uint64_t code; } fr_t;
    size_t cap = 256, sp = 0;
    fr_t *st = (fr_t *)malloc(cap * sizeof(fr_t));
    if (!st) return ORC_ERR_NOMEM;
    st[sp++] = (fr_t){ t->root, 0, 0 };
    int rc = ORC_OK;
    while (sp) {
        fr_t f = st[--sp];
        if (f.node < n) {
            lens[f.node] = f.len;
            if (codes) codes[f.node] = f.code;
            if (f.len > 64) rc = ORC_ERR_BAD_ARG;
            continue;
        }
        if (sp + 2 > cap) {
            cap *= 2;
            fr_t *ns = (fr_t *)realloc(st, cap * sizeof(fr_t));
            if (!ns) { free(st); return ORC_ERR_NOMEM; }
            st = ns;
        }
        st[sp++] = (fr_t){ t->right[f.node - n], f.len + 1, (f.code << 1) | 1 };
        st[sp++] = (fr_t){ t->left[f.node - n], f.len + 1, (f.code << 1) };
    }
    free(st);
    return rc;
}

int orc_huf_build(const uint64_t *counts, uint64_t n, uint32_t *lens, uint64_t *codes) {
    trie_t t;
    int rc = trie_build(counts, n, &t);
    if (rc) return rc;
    rc = trie_codes(&t, lens, codes);
    trie_free(&t);
    return rc;
}

/* ---------------- symbol wire format ---------------- */

static int sym_size(int kind) {
    switch (kind) {
    case ORC_SYM_CHAR: return 1;    /* ser.rs:129-135 */
    case ORC_SYM_RGB: return 11;    /* ser.rs:210-214 */
    case ORC_SYM_SIGNED: return 6;  /* hilbertc.rs:561-565 -> ser.rs:188-195 */
    }
    return -1;
}

static int sym_ser(int kind, uint32_t key, orc_buf *b) {
    switch (kind) {
    case ORC_SYM_CHAR: return orc_ser_u8(b, (uint8_t)key);
    case ORC_SYM_RGB: {
        uint8_t c[3] = { (uint8_t)(key >> 16), (uint8_t)(key >> 8), (uint8_t)key };
        return orc_ser_rgb(b, c);
    }
    case ORC_SYM_SIGNED: {
        for (int i = 0; i < 3; i++) {
            int16_t v = (int16_t)((int)((key >> (18 - 9 * i)) & 511) - 255);
            int rc = orc_ser_i16(b, v);
            if (rc) return rc;
        }
        return ORC_OK;
    }
    }
    return ORC_ERR_BAD_ARG;
}

static int sym_de(int kind, orc_rd *r, uint32_t *key) {
    switch (kind) {
    case ORC_SYM_CHAR: {
        uint8_t v;
        if (orc_de_u8(r, &v)) return ORC_ERR_DECODE;
        *key = v;
        return ORC_OK;
    }
    case ORC_SYM_RGB: {
        uint8_t c[3];
        if (orc_de_rgb(r, c)) return ORC_ERR_DECODE;
        *key = ((uint32_t)c[0] << 16) | ((uint32_t)c[1] << 8) | c[2];
        return ORC_OK;
    }
    case ORC_SYM_SIGNED: {
        uint32_t k = 0;
        for (int i = 0; i < 3; i++) {
            int16_t v;
            if (orc_de_i16(r, &v)) return ORC_ERR_DECODE;
            /* any i16 deserialises; values outside [-255,255] cannot be packed -> reject */
            if (v < -255 || v > 255) return ORC_ERR_DECODE;
            k = (k << 9) | (uint32_t)(v + 255);
        }
        *key = k;
        return ORC_OK;
    }
    }
    return ORC_ERR_BAD_ARG;
}

/* huf.rs:305-321 BinTrie::serialize, pre-order; iterative to survive deep tries */
static int trie_serialize(const trie_t *t, int kind, const uint32_t *keys, orc_buf *out) {
    uint64_t n = t->nleaf;
    size_t cap = 256, sp = 0;
    uint32_t *st = (uint32_t *)malloc(cap * sizeof(uint32_t));
    if (!st) return ORC_ERR_NOMEM;
    st[sp++] = t->root;
    int rc = ORC_OK;
    while (sp && !rc) {
        uint32_t nd = st[--sp];
        if (nd < n) {
            rc = orc_ser_u8(out, 0); /* SER_ENUM_LEAF huf.rs:296 */
            if (!rc) rc = sym_ser(kind, keys[nd], out);
        } else {
            rc = orc_ser_u8(out, 1); /* SER_ENUM_BRANCH huf.rs:297 */
            if (sp + 2 > cap) {
                cap *= 2;
                uint32_t *ns = (uint32_t *)realloc(st, cap * sizeof(uint32_t));
                if (!ns) { free(st); return ORC_ERR_NOMEM; }
                st = ns;
            }
            st[sp++] = t->right[nd - n];
            st[sp++] = t->left[nd - n];
        }
    }
    free(st);
    return rc;
}

static int64_t key_find(const uint32_t *keys, uint64_t n, uint32_t k) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    return (lo < n && keys[lo] == k) ? (int64_t)lo : -1;
}

/* huf.rs:22-43 */
int orc_huf_encode_all(int sym_kind, const uint32_t *syms, uint64_t n, orc_buf *out) {
    if (sym_size(sym_kind) < 0) return ORC_ERR_BAD_ARG;
    if (n == 0) return ORC_ERR_BAD_ARG; /* huf.rs:99 would panic on an empty heap */
    uint32_t *keys = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint64_t *counts = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!keys || !counts) { free(keys); free(counts); return ORC_ERR_NOMEM; }
    uint64_t u = 0;
    int rc = orc_count_freqs(syms, n, keys, counts, n, &u); /* 1. huf.rs:30 */
    trie_t t;
    memset(&t, 0, sizeof t);
    uint32_t *lens = NULL;
    uint64_t *codes = NULL;
    if (!rc) rc = trie_build(counts, u, &t);                  /* huf.rs:31 */
    if (!rc) {
        lens = (uint32_t *)malloc(u * sizeof(uint32_t));
        codes = (uint64_t *)malloc(u * sizeof(uint64_t));
        if (!lens || !codes) rc = ORC_ERR_NOMEM;
    }
    if (!rc) rc = trie_codes(&t, lens, codes);                /* Enc::from(&Dec) huf.rs:125-135 */
    if (!rc) rc = trie_serialize(&t, sym_kind, keys, out);    /* 2. huf.rs:34 */
    if (!rc) {                                                /* 3. huf.rs:37-41 */
        orc_bitw w;
        orc_bitw_init(&w, out);
        for (uint64_t i = 0; i < n && !rc; i++) {
            int64_t idx = key_find(keys, u, syms[i]);
            if (idx < 0) { rc = ORC_ERR_BAD_ARG; break; }
            uint32_t len = lens[idx];
            if (len == 0) continue;                           /* huf.rs:140-142 */
            uint64_t code = codes[idx];
            /* BitArray::from_slice (bit.rs:123-149) then write_arr (bit.rs:164-178) */
            uint8_t full[8];
            size_t nfull = len / 8;
            uint8_t pc = (uint8_t)(len % 8);
            for (size_t k = 0; k < nfull; k++)
                full[k] = (uint8_t)(code >> (len - 8 * (k + 1)));
            uint8_t partial = (uint8_t)(code & orc_bit_mask(pc));
            rc = orc_bitw_code(&w, full, nfull, partial, pc);
        }
        if (!rc) rc = orc_bitw_pad_and_flush(&w);
    }
    trie_free(&t);
    free(lens); free(codes); free(keys); free(counts);
    return rc;
}

int orc_huf_size(int sym_kind, const uint64_t *counts, uint64_t n, uint64_t *nbytes) {
    int S = sym_size(sym_kind);
    if (S < 0 || n == 0) return ORC_ERR_BAD_ARG;
    uint32_t *lens = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!lens) return ORC_ERR_NOMEM;
    int rc = orc_huf_build(counts, n, lens, NULL);
    if (!rc) {
        uint64_t bits = 0;
        for (uint64_t i = 0; i < n; i++) bits += counts[i] * lens[i];
        *nbytes = n * (uint64_t)(1 + S) + (n - 1) + (bits + 7) / 8;
    }
    free(lens);
    return rc;
}

/* ---------------- decode: huf.rs:46-53, 323-348, 187-206, 366-374 ---------------- */

typedef struct {
    uint32_t *left, *right; /* per branch */
    uint32_t *leaf_key;     /* per leaf   */
    uint8_t  *is_leaf;      /* per node   */
    uint32_t *slot;         /* per node: index into leaf_key or left/right */
    size_t nnodes, nleaf, nbranch, cap;
} dtrie_t;

static int dtrie_grow(dtrie_t *d) {
    size_t nc = d->cap ? d->cap * 2 : 64;
    d->left = (uint32_t *)realloc(d->left, nc * sizeof(uint32_t));
    d->right = (uint32_t *)realloc(d->right, nc * sizeof(uint32_t));
    d->leaf_key = (uint32_t *)realloc(d->leaf_key, nc * sizeof(uint32_t));
    d->is_leaf = (uint8_t *)realloc(d->is_leaf, nc);
    d->slot = (uint32_t *)realloc(d->slot, nc * sizeof(uint32_t));
    if (!d->left || !d->right || !d->leaf_key || !d->is_leaf || !d->slot) return ORC_ERR_NOMEM;
    d->cap = nc;
    return ORC_OK;
}

/* pre-order deserialise without recursion: a stack of branches waiting for children */
static int dtrie_read(int kind, orc_rd *in, dtrie_t *d, uint32_t *root) {
    typedef struct { uint32_t node; int filled; } pend_t;
    size_t pcap = 64, psp = 0;
    pend_t *pend = (pend_t *)malloc(pcap * sizeof(pend_t));
    if (!pend) return ORC_ERR_NOMEM;
    int rc = ORC_OK;
    int have_root = 0;
    for (;;) {
        uint8_t tag;
        if (orc_de_u8(in, &tag)) { rc = ORC_ERR_DECODE; break; }
        if (d->nnodes + 1 > d->cap && (rc = dtrie_grow(d))) break;
        uint32_t id = (uint32_t)d->nnodes++;
        if (tag == 0) {
            uint32_t key;
            if ((rc = sym_de(kind, in, &key))) break;
            d->is_leaf[id] = 1;
            d->slot[id] = (uint32_t)d->nleaf;
            d->leaf_key[d->nleaf++] = key;
        } else if (tag == 1) {
            d->is_leaf[id] = 0;
            d->slot[id] = (uint32_t)d->nbranch++;
        } else { rc = ORC_ERR_DECODE; break; } /* huf.rs:343-345 */
        /* attach to parent */
        if (!have_root) { *root = id; have_root = 1; }
        else {
            pend_t *p = &pend[psp - 1];
            if (p->filled == 0) { d->left[d->slot[p->node]] = id; p->filled = 1; }
            else { d->right[d->slot[p->node]] = id; psp--; }
        }
        if (tag == 1) {
            if (psp + 1 > pcap) {
                pcap *= 2;
                pend_t *np = (pend_t *)realloc(pend, pcap * sizeof(pend_t));
                if (!np) { rc = ORC_ERR_NOMEM; break; }
                pend = np;
            }
            pend[psp++] = (pend_t){ id, 0 };
        }
        if (psp == 0) break; /* tree complete */
    }
    free(pend);
    return rc;
}

int orc_huf_decode_all(int sym_kind, orc_rd *in, uint32_t *syms, uint64_t nsyms) {
    if (sym_size(sym_kind) < 0) return ORC_ERR_BAD_ARG;
    dtrie_t d;
    memset(&d, 0, sizeof d);
    uint32_t root = 0;
    int rc = dtrie_read(sym_kind, in, &d, &root);
    if (!rc) {
        /* bit_reader (bit.rs:256-259): MSB-first bits of the remaining bytes */
        uint64_t bitpos = 0;
        const uint8_t *p = in->p + in->pos;
        uint64_t nbits = (uint64_t)(in->n - in->pos) * 8;
        for (uint64_t i = 0; i < nsyms; i++) {
            uint32_t nd = root;
            while (!d.is_leaf[nd]) { /* BinTrie::lookup huf.rs:187-206 */
                if (bitpos >= nbits) { rc = ORC_ERR_DECODE; break; } /* EOF -> None */
                int bit = (p[bitpos >> 3] >> (7 - (bitpos & 7))) & 1;
                bitpos++;
                nd = bit ? d.right[d.slot[nd]] : d.left[d.slot[nd]];
            }
            if (rc) break;
            syms[i] = d.leaf_key[d.slot[nd]];
        }
        in->pos += (size_t)((bitpos + 7) / 8);
    }
    free(d.left); free(d.right); free(d.leaf_key); free(d.is_leaf); free(d.slot);
    return rc;
}
