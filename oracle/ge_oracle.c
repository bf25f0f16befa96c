/*
 * ge_oracle.c -- CPU restatement of the Phaken/graph-embeddings hot path.
 * TEST INFRASTRUCTURE ONLY; see ge_oracle.h for the rules and for the
 * "parity unpinned" statement.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -pthread (oracle/Makefile).
 * float/double mixing follows the Java expressions token by token; every
 * narrowing `(float)` below corresponds to a Java implicit compound-assignment
 * narrowing or an explicit cast in the cited line.
 */
#include "ge_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* java.util.Random (JDK 8 source semantics; public algorithm)               */
/* ------------------------------------------------------------------------- */
#define JR_MULT 0x5DEECE66DULL
#define JR_ADD  0xBULL
#define JR_MASK ((1ULL << 48) - 1)

void geo_jrand_init(geo_jrand *r, int64_t seed) {
    r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;       /* initialScramble */
}

int32_t geo_jrand_next(geo_jrand *r, int bits) {
    r->seed = (r->seed * JR_MULT + JR_ADD) & JR_MASK;
    return (int32_t)(uint32_t)(r->seed >> (48 - bits));   /* (int)(seed >>> (48-bits)) */
}

int32_t geo_jrand_next_int(geo_jrand *r) { return geo_jrand_next(r, 32); }

int32_t geo_jrand_next_int_bound(geo_jrand *r, int32_t bound) {
    /* Random.nextInt(int bound); used via ExtendedRandom.uniform(int),
     * J/util/rnd/ExtendedRandom.java:53-56 */
    int32_t rr = geo_jrand_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) {
        rr = (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    } else {
        int32_t u = rr;
        for (;;) {
            rr = u % bound;
            /* u - r + m < 0 with Java int wrap-around */
            int32_t t = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
            if (t >= 0) break;
            u = geo_jrand_next(r, 31);
        }
    }
    return rr;
}

float geo_jrand_next_float(geo_jrand *r) {
    return (float)geo_jrand_next(r, 24) / (float)(1 << 24);
}

void geo_jrand_shuffle(geo_jrand *r, int32_t *a, int32_t n) {
    /* J/util/rnd/ExtendedRandom.java:398-407 */
    for (int32_t i = 0; i < n; i++) {
        int32_t k = i + geo_jrand_next_int_bound(r, n - i);
        int32_t t = a[i]; a[i] = a[k]; a[k] = t;
    }
}

/* ------------------------------------------------------------------------- */
/* java.util.HashMap<Integer,Float> (JDK 8) iteration-order emulation         */
/* (BCV extends HashMap, J/bca/util/BCV.java:14; its order is the COO order,  */
/* J/bca/BookmarkColoring.java:99-103).  Restated from the JDK 8 source:      */
/*  - putVal appends at the bin tail; a 9th node in one bin calls treeifyBin; */
/*  - merge() resizes BEFORE the lookup, links a new key at the bin HEAD and  */
/*    calls treeifyBin when the bin already held >= 7 nodes (8th node);       */
/*  - treeifyBin with table length < 64 only calls resize();                  */
/*  - at length >= 64 the bin becomes a red-black tree of TreeNodes ordered   */
/*    by hash; iteration still follows `next`: the tree ROOT is moved to the  */
/*    front (moveRootToFront), a later key is linked behind its tree PARENT   */
/*    (putTreeVal); resize splits a tree bin into lo/hi lists in `next`       */
/*    order and untreeifies lists of <= 6 nodes (TreeNode.split);             */
/*  - remove() of a tree node unlinks it from the chain, may untreeify a      */
/*    small tree, and moves the new root to the front (removeTreeNode).       */
/* Integer keys: hash = k ^ (k >>> 16) is a bijection, so two keys never tie  */
/* on the hash and the tree order is the signed order of the hashes.          */
/* ------------------------------------------------------------------------- */
#define JHM_TREEIFY_THRESHOLD   8
#define JHM_UNTREEIFY_THRESHOLD 6
#define JHM_MIN_TREEIFY_CAPACITY 64

typedef struct {
    int32_t key; float val;
    int32_t next;                          /* Node.next */
    int32_t prev, parent, left, right;     /* TreeNode links (valid while the node's bin is a tree) */
    uint8_t red;
} jnode;
typedef struct {
    jnode   *nodes;
    int32_t  n_nodes, cap_nodes;
    int32_t *head;
    uint8_t *tree;       /* tree[b] != 0: the nodes of bin b are TreeNodes */
    int32_t  tabcap;     /* 0 = table == null */
    int32_t  threshold;
    int32_t  size;
} jhm;

static inline uint32_t jhm_hash(int32_t key) {
    uint32_t h = (uint32_t)key;            /* Integer.hashCode() == value */
    return h ^ (h >> 16);                  /* HashMap.hash() */
}
static inline int32_t jhm_shash(const jhm *m, int32_t e) { return (int32_t)jhm_hash(m->nodes[e].key); }  /* `int hash`, signed compares */

static void jhm_init(jhm *m) { memset(m, 0, sizeof(*m)); }
static void jhm_free(jhm *m) { free(m->nodes); free(m->head); free(m->tree); jhm_init(m); }
static void jhm_clear(jhm *m) {
    m->n_nodes = 0; m->size = 0; m->tabcap = 0; m->threshold = 0;
}

/* ---- TreeNode: rotations and rebalancing (HashMap.TreeNode.rotateLeft/Right, balanceInsertion, balanceDeletion) ---- */
#define N(e) (m->nodes[e])
static int32_t jt_rotate_left(jhm *m, int32_t root, int32_t p) {
    int32_t r, pp, rl;
    if (p >= 0 && (r = N(p).right) >= 0) {
        if ((rl = N(p).right = N(r).left) >= 0) N(rl).parent = p;
        if ((pp = N(r).parent = N(p).parent) < 0) { root = r; N(r).red = 0; }
        else if (N(pp).left == p) N(pp).left = r;
        else N(pp).right = r;
        N(r).left = p;
        N(p).parent = r;
    }
    return root;
}
static int32_t jt_rotate_right(jhm *m, int32_t root, int32_t p) {
    int32_t l, pp, lr;
    if (p >= 0 && (l = N(p).left) >= 0) {
        if ((lr = N(p).left = N(l).right) >= 0) N(lr).parent = p;
        if ((pp = N(l).parent = N(p).parent) < 0) { root = l; N(l).red = 0; }
        else if (N(pp).right == p) N(pp).right = l;
        else N(pp).left = l;
        N(l).right = p;
        N(p).parent = l;
    }
    return root;
}
static int32_t jt_balance_insertion(jhm *m, int32_t root, int32_t x) {
    N(x).red = 1;
    for (;;) {
        int32_t xp, xpp, xppl, xppr;
        if ((xp = N(x).parent) < 0) { N(x).red = 0; return x; }
        else if (!N(xp).red || (xpp = N(xp).parent) < 0) return root;
        if (xp == (xppl = N(xpp).left)) {
            if ((xppr = N(xpp).right) >= 0 && N(xppr).red) {
                N(xppr).red = 0; N(xp).red = 0; N(xpp).red = 1; x = xpp;
            } else {
                if (x == N(xp).right) {
                    root = jt_rotate_left(m, root, x = xp);
                    xpp = (xp = N(x).parent) < 0 ? -1 : N(xp).parent;
                }
                if (xp >= 0) {
                    N(xp).red = 0;
                    if (xpp >= 0) { N(xpp).red = 1; root = jt_rotate_right(m, root, xpp); }
                }
            }
        } else {
            if (xppl >= 0 && N(xppl).red) {
                N(xppl).red = 0; N(xp).red = 0; N(xpp).red = 1; x = xpp;
            } else {
                if (x == N(xp).left) {
                    root = jt_rotate_right(m, root, x = xp);
                    xpp = (xp = N(x).parent) < 0 ? -1 : N(xp).parent;
                }
                if (xp >= 0) {
                    N(xp).red = 0;
                    if (xpp >= 0) { N(xpp).red = 1; root = jt_rotate_left(m, root, xpp); }
                }
            }
        }
    }
}
static int32_t jt_balance_deletion(jhm *m, int32_t root, int32_t x) {
    for (;;) {
        int32_t xp, xpl, xpr;
        if (x < 0 || x == root) return root;
        else if ((xp = N(x).parent) < 0) { N(x).red = 0; return x; }
        else if (N(x).red) { N(x).red = 0; return root; }
        else if ((xpl = N(xp).left) == x) {
            if ((xpr = N(xp).right) >= 0 && N(xpr).red) {
                N(xpr).red = 0; N(xp).red = 1;
                root = jt_rotate_left(m, root, xp);
                xpr = (xp = N(x).parent) < 0 ? -1 : N(xp).right;
            }
            if (xpr < 0) x = xp;
            else {
                int32_t sl = N(xpr).left, sr = N(xpr).right;
                if ((sr < 0 || !N(sr).red) && (sl < 0 || !N(sl).red)) { N(xpr).red = 1; x = xp; }
                else {
                    if (sr < 0 || !N(sr).red) {
                        if (sl >= 0) N(sl).red = 0;
                        N(xpr).red = 1;
                        root = jt_rotate_right(m, root, xpr);
                        xpr = (xp = N(x).parent) < 0 ? -1 : N(xp).right;
                    }
                    if (xpr >= 0) {
                        N(xpr).red = (xp < 0) ? 0 : N(xp).red;
                        if ((sr = N(xpr).right) >= 0) N(sr).red = 0;
                    }
                    if (xp >= 0) { N(xp).red = 0; root = jt_rotate_left(m, root, xp); }
                    x = root;
                }
            }
        } else {   /* symmetric */
            if (xpl >= 0 && N(xpl).red) {
                N(xpl).red = 0; N(xp).red = 1;
                root = jt_rotate_right(m, root, xp);
                xpl = (xp = N(x).parent) < 0 ? -1 : N(xp).left;
            }
            if (xpl < 0) x = xp;
            else {
                int32_t sl = N(xpl).left, sr = N(xpl).right;
                if ((sl < 0 || !N(sl).red) && (sr < 0 || !N(sr).red)) { N(xpl).red = 1; x = xp; }
                else {
                    if (sl < 0 || !N(sl).red) {
                        if (sr >= 0) N(sr).red = 0;
                        N(xpl).red = 1;
                        root = jt_rotate_left(m, root, xpl);
                        xpl = (xp = N(x).parent) < 0 ? -1 : N(xp).left;
                    }
                    if (xpl >= 0) {
                        N(xpl).red = (xp < 0) ? 0 : N(xp).red;
                        if ((sl = N(xpl).left) >= 0) N(sl).red = 0;
                    }
                    if (xp >= 0) { N(xp).red = 0; root = jt_rotate_right(m, root, xp); }
                    x = root;
                }
            }
        }
    }
}
static int32_t jt_root(const jhm *m, int32_t e) { while (N(e).parent >= 0) e = N(e).parent; return e; }

/* TreeNode.moveRootToFront: the root becomes the bin's first node; the other nodes keep their `next` order */
static void jt_move_root_to_front(jhm *m, int32_t *tab, int32_t tabcap, int32_t root) {
    if (root < 0 || tabcap <= 0) return;
    const int32_t index = (int32_t)(jhm_hash(N(root).key) & (uint32_t)(tabcap - 1));
    const int32_t first = tab[index];
    if (root != first) {
        int32_t rn;
        tab[index] = root;
        const int32_t rp = N(root).prev;
        if ((rn = N(root).next) >= 0) N(rn).prev = rp;
        if (rp >= 0) N(rp).next = rn;
        if (first >= 0) N(first).prev = root;
        N(root).next = first;
        N(root).prev = -1;
    }
}
/* TreeNode.treeify: build the tree over the chain that starts at `first` (in chain order), then moveRootToFront */
static void jt_treeify(jhm *m, int32_t *tab, int32_t tabcap, int32_t first) {
    int32_t root = -1;
    for (int32_t x = first, next; x >= 0; x = next) {
        next = N(x).next;
        N(x).left = N(x).right = -1;
        if (root < 0) { N(x).parent = -1; N(x).red = 0; root = x; }
        else {
            const int32_t h = jhm_shash(m, x);
            for (int32_t p = root;;) {
                const int32_t ph = jhm_shash(m, p);
                const int dir = (ph > h) ? -1 : 1;        /* distinct Integer keys never share a hash */
                const int32_t xp = p;
                if ((p = (dir <= 0) ? N(p).left : N(p).right) < 0) {
                    N(x).parent = xp;
                    if (dir <= 0) N(xp).left = x; else N(xp).right = x;
                    root = jt_balance_insertion(m, root, x);
                    break;
                }
            }
        }
    }
    jt_move_root_to_front(m, tab, tabcap, root);
}
/* TreeNode.find from the root */
static int32_t jt_find(const jhm *m, int32_t root, int32_t key) {
    const int32_t h = (int32_t)jhm_hash(key);
    int32_t p = root;
    while (p >= 0) {
        const int32_t ph = jhm_shash(m, p);
        if (ph > h) p = N(p).left;
        else if (ph < h) p = N(p).right;
        else return p;                                     /* equal hash <=> equal Integer */
    }
    return -1;
}

static int32_t jhm_new_node(jhm *m, int32_t key, float val) {
    if (m->n_nodes == m->cap_nodes) {
        m->cap_nodes = m->cap_nodes ? m->cap_nodes * 2 : 64;
        m->nodes = (jnode *)realloc(m->nodes, sizeof(jnode) * (size_t)m->cap_nodes);
    }
    int32_t id = m->n_nodes++;
    m->nodes[id].key = key; m->nodes[id].val = val; m->nodes[id].next = -1;
    m->nodes[id].prev = m->nodes[id].parent = m->nodes[id].left = m->nodes[id].right = -1; m->nodes[id].red = 0;
    return id;
}

/* TreeNode.putTreeVal for a key known to be absent: the new node is linked behind its tree parent */
static void jt_put_tree_val(jhm *m, int32_t b, int32_t key, float val) {
    const int32_t h = (int32_t)jhm_hash(key);
    const int32_t root = jt_root(m, m->head[b]);
    for (int32_t p = root;;) {
        const int dir = (jhm_shash(m, p) > h) ? -1 : 1;
        const int32_t xp = p;
        if ((p = (dir <= 0) ? N(p).left : N(p).right) < 0) {
            const int32_t x = jhm_new_node(m, key, val);      /* may move m->nodes: no cached pointers */
            const int32_t xpn = N(xp).next;
            N(x).next = xpn;
            if (dir <= 0) N(xp).left = x; else N(xp).right = x;
            N(xp).next = x;
            N(x).parent = N(x).prev = xp;
            if (xpn >= 0) N(xpn).prev = x;
            jt_move_root_to_front(m, m->head, m->tabcap, jt_balance_insertion(m, root, x));
            return;
        }
    }
}

/* HashMap.resize(): a list bin splits into lo/hi lists, relative order kept; a tree bin goes through TreeNode.split */
static void jhm_resize(jhm *m) {
    const int32_t oldcap = m->tabcap;
    const int32_t newcap = oldcap ? oldcap * 2 : 16;
    int32_t *nh = (int32_t *)malloc(sizeof(int32_t) * (size_t)newcap);
    uint8_t *ntree = (uint8_t *)calloc((size_t)newcap, 1);
    for (int32_t b = 0; b < newcap; b++) nh[b] = -1;
    for (int32_t j = 0; j < oldcap; j++) {
        int32_t e = m->head[j];
        if (e < 0) continue;
        if (N(e).next < 0) { nh[jhm_hash(N(e).key) & (uint32_t)(newcap - 1)] = e; continue; }
        int32_t lo_h = -1, lo_t = -1, hi_h = -1, hi_t = -1, lc = 0, hc = 0;
        const int is_tree = m->tree[j];
        for (int32_t nx; e >= 0; e = nx) {
            nx = N(e).next;
            N(e).next = -1;
            if ((jhm_hash(N(e).key) & (uint32_t)oldcap) == 0) {
                if ((N(e).prev = lo_t) < 0) lo_h = e; else N(lo_t).next = e;
                lo_t = e; ++lc;
            } else {
                if ((N(e).prev = hi_t) < 0) hi_h = e; else N(hi_t).next = e;
                hi_t = e; ++hc;
            }
        }
        if (!is_tree) { nh[j] = lo_h; nh[j + oldcap] = hi_h; continue; }
        /* TreeNode.split */
        if (lo_h >= 0) {
            nh[j] = lo_h;
            if (lc > JHM_UNTREEIFY_THRESHOLD) {        /* else untreeify: plain nodes in this order */
                ntree[j] = 1;
                if (hi_h >= 0) jt_treeify(m, nh, newcap, lo_h);   /* (else is already treeified) */
            }
        }
        if (hi_h >= 0) {
            nh[j + oldcap] = hi_h;
            if (hc > JHM_UNTREEIFY_THRESHOLD) {
                ntree[j + oldcap] = 1;
                if (lo_h >= 0) jt_treeify(m, nh, newcap, hi_h);
            }
        }
    }
    free(m->head); free(m->tree);
    m->head = nh; m->tree = ntree;
    m->tabcap = newcap;
    m->threshold = (newcap / 4) * 3;       /* 0.75 * cap, exact for cap >= 16 */
}

/* HashMap.treeifyBin: below 64 buckets it only resizes */
static void jhm_treeify_bin(jhm *m, int32_t key) {
    if (m->tabcap < JHM_MIN_TREEIFY_CAPACITY) { jhm_resize(m); return; }
    const int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    int32_t e = m->head[b], tl = -1;
    if (e < 0) return;
    for (; e >= 0; e = N(e).next) { N(e).prev = tl; tl = e; }        /* replacementTreeNode keeps the order */
    m->tree[b] = 1;
    jt_treeify(m, m->head, m->tabcap, m->head[b]);
}

static int32_t jhm_find(const jhm *m, int32_t key) {
    if (!m->tabcap) return -1;
    const int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    int32_t e = m->head[b];
    if (e < 0) return -1;
    if (m->tree[b]) return jt_find(m, jt_root(m, e), key);
    while (e >= 0) { if (m->nodes[e].key == key) return e; e = m->nodes[e].next; }
    return -1;
}

/* BCV.add(int,float): super.put(key, getOrDefault(key, 0f) + value)  J/bca/util/BCV.java:35-37.
 * HashMap.putVal: a new key is appended at the bin TAIL (binCount >= 7 there, i.e. it is the 9th node: treeifyBin),
 * or goes through putTreeVal; then resize when ++size > threshold. */
static void jhm_bcv_add(jhm *m, int32_t key, float value) {
    int32_t e = jhm_find(m, key);
    if (e >= 0) { m->nodes[e].val = m->nodes[e].val + value; return; }
    if (!m->tabcap) jhm_resize(m);
    const int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    if (m->head[b] < 0) { const int32_t id = jhm_new_node(m, key, 0.0f + value); m->head[b] = id; }
    else if (m->tree[b]) jt_put_tree_val(m, b, key, 0.0f + value);
    else {
        const int32_t id = jhm_new_node(m, key, 0.0f + value);
        int32_t t = m->head[b], bin_count = 0;
        while (m->nodes[t].next >= 0) { t = m->nodes[t].next; ++bin_count; }
        m->nodes[t].next = id;
        if (bin_count >= JHM_TREEIFY_THRESHOLD - 1) jhm_treeify_bin(m, key);
    }
    if (++m->size > m->threshold) jhm_resize(m);
}

/* HashMap.merge(key, value, Float::sum) (JDK 8): resize BEFORE the lookup when size > threshold; a new key goes
 * through putTreeVal in a tree bin, else it is linked at the bin HEAD and treeifyBin runs when the bin already
 * held >= 7 nodes (binCount counts every node of the bin here); no resize afterwards.
 * Called from BCV.merge, J/bca/util/BCV.java:105-107. */
static void jhm_merge_sum(jhm *m, int32_t key, float value) {
    if (m->size > m->threshold || !m->tabcap) jhm_resize(m);
    int32_t e = jhm_find(m, key);
    if (e >= 0) { m->nodes[e].val = m->nodes[e].val + value; return; }  /* Float.sum(old, value) */
    const int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    if (m->head[b] >= 0 && m->tree[b]) jt_put_tree_val(m, b, key, value);
    else {
        int32_t bin_count = 0;
        for (int32_t t = m->head[b]; t >= 0; t = m->nodes[t].next) ++bin_count;
        const int32_t id = jhm_new_node(m, key, value);
        m->nodes[id].next = m->head[b];
        m->head[b] = id;
        if (bin_count >= JHM_TREEIFY_THRESHOLD - 1) jhm_treeify_bin(m, key);
    }
    ++m->size;
}

/* TreeNode.removeTreeNode(map, tab, movable = true) */
static void jt_remove_tree_node(jhm *m, int32_t b, int32_t p) {
    int32_t first = m->head[b], root = first, rl;
    const int32_t succ = N(p).next, pred = N(p).prev;
    if (pred < 0) m->head[b] = first = succ; else N(pred).next = succ;
    if (succ >= 0) N(succ).prev = pred;
    if (first < 0) { m->tree[b] = 0; return; }
    if (N(root).parent >= 0) root = jt_root(m, root);
    if (root < 0 || N(root).right < 0 || (rl = N(root).left) < 0 || N(rl).left < 0) {
        m->tree[b] = 0;                                   /* too small: untreeify, chain order kept */
        return;
    }
    const int32_t pl = N(p).left, pr = N(p).right;
    int32_t replacement;
    if (pl >= 0 && pr >= 0) {
        int32_t s = pr, sl;
        while ((sl = N(s).left) >= 0) s = sl;            /* find successor */
        const uint8_t c = N(s).red; N(s).red = N(p).red; N(p).red = c;   /* swap colors */
        const int32_t sr = N(s).right;
        const int32_t pp = N(p).parent;
        if (s == pr) { N(p).parent = s; N(s).right = p; }
        else {
            const int32_t sp = N(s).parent;
            if ((N(p).parent = sp) >= 0) { if (s == N(sp).left) N(sp).left = p; else N(sp).right = p; }
            if ((N(s).right = pr) >= 0) N(pr).parent = s;
        }
        N(p).left = -1;
        if ((N(p).right = sr) >= 0) N(sr).parent = p;
        if ((N(s).left = pl) >= 0) N(pl).parent = s;
        if ((N(s).parent = pp) < 0) root = s;
        else if (p == N(pp).left) N(pp).left = s;
        else N(pp).right = s;
        replacement = sr >= 0 ? sr : p;
    }
    else if (pl >= 0) replacement = pl;
    else if (pr >= 0) replacement = pr;
    else replacement = p;
    if (replacement != p) {
        const int32_t pp = N(replacement).parent = N(p).parent;
        if (pp < 0) root = replacement;
        else if (p == N(pp).left) N(pp).left = replacement;
        else N(pp).right = replacement;
        N(p).left = N(p).right = N(p).parent = -1;
    }
    const int32_t r = N(p).red ? root : jt_balance_deletion(m, root, replacement);
    if (replacement == p) {                              /* detach */
        const int32_t pp = N(p).parent;
        N(p).parent = -1;
        if (pp >= 0) {
            if (p == N(pp).left) N(pp).left = -1;
            else if (p == N(pp).right) N(pp).right = -1;
        }
    }
    jt_move_root_to_front(m, m->head, m->tabcap, r);
}

/* HashMap.remove(key) -> removeNode */
static void jhm_remove(jhm *m, int32_t key) {
    if (!m->tabcap) return;
    const int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    if (m->head[b] >= 0 && m->tree[b]) {
        const int32_t p = jt_find(m, jt_root(m, m->head[b]), key);
        if (p >= 0) { jt_remove_tree_node(m, b, p); --m->size; }
        return;
    }
    int32_t e = m->head[b], prev = -1;
    while (e >= 0) {
        if (m->nodes[e].key == key) {
            if (prev < 0) m->head[b] = m->nodes[e].next; else m->nodes[prev].next = m->nodes[e].next;
            --m->size;
            return;
        }
        prev = e; e = m->nodes[e].next;
    }
}
#undef N

/* iteration: bins ascending, list order */
#define JHM_FOREACH(m, e) \
    for (int32_t _b = 0; _b < (m)->tabcap; _b++) \
        for (int32_t e = (m)->head[_b]; e >= 0; e = (m)->nodes[e].next)

/* Float.compare */
static int jfloat_compare(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    int32_t ia, ib;
    if (a != a) ia = 0x7fc00000; else memcpy(&ia, &a, 4);   /* floatToIntBits canonical NaN */
    if (b != b) ib = 0x7fc00000; else memcpy(&ib, &b, 4);
    return ia == ib ? 0 : (ia < ib ? -1 : 1);
}

/* BCV.max(): values().stream().max(Float::compareTo).orElse(1f)  J/bca/util/BCV.java:82-84 */
static float bcv_max(const jhm *m) {
    int have = 0; float best = 1.0f;
    JHM_FOREACH(m, e) {
        float v = m->nodes[e].val;
        if (!have) { best = v; have = 1; }
        else best = (jfloat_compare(best, v) >= 0) ? best : v;   /* BinaryOperator.maxBy */
    }
    return best;
}
/* BCV.min(): orElse(0f)  J/bca/util/BCV.java:75-77 */
static float bcv_min(const jhm *m) {
    int have = 0; float best = 0.0f;
    JHM_FOREACH(m, e) {
        float v = m->nodes[e].val;
        if (!have) { best = v; have = 1; }
        else best = (jfloat_compare(best, v) <= 0) ? best : v;   /* BinaryOperator.minBy */
    }
    return best;
}
/* BCV.sum(): reduce(Float::sum).orElse(0f), sequential stream = left fold in iteration order */
static float bcv_sum(const jhm *m) {
    int have = 0; float s = 0.0f;
    JHM_FOREACH(m, e) {
        if (!have) { s = m->nodes[e].val; have = 1; } else s = s + m->nodes[e].val;
    }
    return s;
}
/* BCV.toUnity  J/bca/util/BCV.java:64-70 */
static void bcv_to_unity(jhm *m, int32_t root) {
    jhm_remove(m, root);
    const float sum = bcv_sum(m);
    JHM_FOREACH(m, e) m->nodes[e].val = m->nodes[e].val / sum - 1e-6f;
}
/* BCV.toCounts + scale  J/bca/util/BCV.java:52-59,89-91 */
static void bcv_to_counts(jhm *m, int32_t root) {
    const float aMax = bcv_max(m), aMin = bcv_min(m);
    const float mx = 1000.0f, mn = 1.0f;
    JHM_FOREACH(m, e) m->nodes[e].val = (m->nodes[e].val / ((aMax - aMin) / (mx - mn))) + mn;
    jhm_remove(m, root);
}

/* ------------------------------------------------------------------------- */
/* BCA jobs                                                                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t V;
    const int64_t *out_ptr; const int32_t *out_idx; const float *out_w;
    const int64_t *in_ptr;  const int32_t *in_idx;  const float *in_w;
    double alpha, epsilon;
    /* TreeMap<Integer,PaintedNode> stand-in: dense paint + membership + min-heap of ids */
    double  *paint;
    uint8_t *in_tree;
    int32_t *heap; int32_t heap_n, heap_cap;
} bca_ctx;

static void heap_push(bca_ctx *c, int32_t id) {
    if (c->heap_n == c->heap_cap) {
        c->heap_cap = c->heap_cap ? c->heap_cap * 2 : 256;
        c->heap = (int32_t *)realloc(c->heap, sizeof(int32_t) * (size_t)c->heap_cap);
    }
    int32_t i = c->heap_n++;
    while (i > 0) {
        int32_t p = (i - 1) / 2;
        if (c->heap[p] <= id) break;
        c->heap[i] = c->heap[p]; i = p;
    }
    c->heap[i] = id;
}
static int32_t heap_pop(bca_ctx *c) {
    int32_t top = c->heap[0];
    int32_t last = c->heap[--c->heap_n];
    int32_t i = 0, n = c->heap_n;
    for (;;) {
        int32_t l = 2 * i + 1, r = l + 1, s = i; int32_t sv = last;
        if (l < n && c->heap[l] < sv) { s = l; sv = c->heap[l]; }
        if (r < n && c->heap[r] < sv) { s = r; sv = c->heap[r]; }
        if (s == i) break;
        c->heap[i] = c->heap[s]; i = s;
    }
    if (n > 0) c->heap[i] = last;
    return top;
}

/* nodeTree.containsKey / get().addPaint / put(new PaintedNode)
 * J/bca/jobs/DirectedWeighted.java:89-96 */
static inline void tree_add(bca_ctx *c, int32_t node, double p) {
    if (c->in_tree[node]) c->paint[node] += p;
    else { c->paint[node] = p; c->in_tree[node] = 1; heap_push(c, node); }
}

/* DirectedWeighted.doWork  J/bca/jobs/DirectedWeighted.java:31-101 */
static void dowork_directed(bca_ctx *c, int32_t bookmark, int reverse, jhm *bcv) {
    const double alpha = c->alpha, epsilon = c->epsilon;
    const int64_t *ptr = reverse ? c->in_ptr : c->out_ptr;
    const int32_t *idx = reverse ? c->in_idx : c->out_idx;
    const float   *w   = reverse ? c->in_w   : c->out_w;
    c->heap_n = 0;
    tree_add(c, bookmark, 1.0);                               /* :39 */
    while (c->heap_n > 0) {
        const int32_t focus = heap_pop(c);                    /* pollFirstEntry :48 */
        c->in_tree[focus] = 0;
        const double wet = c->paint[focus];
        jhm_bcv_add(bcv, focus, (float)(alpha * wet));        /* :53 */
        if (wet < epsilon) continue;                          /* :56 */
        const int64_t b = ptr[focus], e = ptr[focus + 1];
        if (e == b) continue;                                 /* :66 */
        double total = 0;
        for (int64_t k = b; k < e; k++) total += w[k];        /* :69-75 */
        if (total == 0) continue;                             /* :77 */
        for (int64_t k = b; k < e; k++) {
            const float weight = w[k];
            const double p = (1 - alpha) * wet * (weight / total);   /* :82 */
            if (p < epsilon) continue;                        /* :85 */
            tree_add(c, idx[k], p);
        }
    }
}

/* UndirectedWeighted.doWork  J/bca/jobs/UndirectedWeighted.java:31-114 */
static void dowork_undirected(bca_ctx *c, int32_t bookmark, jhm *bcv) {
    const double alpha = c->alpha, epsilon = c->epsilon;
    c->heap_n = 0;
    tree_add(c, bookmark, 1.0);
    while (c->heap_n > 0) {
        const int32_t focus = heap_pop(c);
        c->in_tree[focus] = 0;
        const double wet = c->paint[focus];
        jhm_bcv_add(bcv, focus, (float)(alpha * wet));        /* :55 */
        if (wet < epsilon) continue;                          /* :58 */
        double total = 0;
        for (int64_t k = c->out_ptr[focus]; k < c->out_ptr[focus + 1]; k++) total += c->out_w[k];  /* :63-67 */
        for (int64_t k = c->in_ptr[focus];  k < c->in_ptr[focus + 1];  k++) total += c->in_w[k];   /* :69-73 */
        /* no total == 0 guard in the undirected version */
        for (int64_t k = c->out_ptr[focus]; k < c->out_ptr[focus + 1]; k++) {       /* :75-93 */
            const float weight = c->out_w[k];
            const double p = (1 - alpha) * wet * (weight / total);
            if (p < epsilon) continue;
            tree_add(c, c->out_idx[k], p);
        }
        for (int64_t k = c->in_ptr[focus]; k < c->in_ptr[focus + 1]; k++) {         /* :95-112 */
            const float weight = c->in_w[k];
            const double p = (1 - alpha) * wet * (weight / total);
            if (p < epsilon) continue;
            tree_add(c, c->in_idx[k], p);
        }
    }
}

/* BCAJob.call (J/bca/util/BCAJob.java:31-36) + normalisation
 * (J/bca/BookmarkColoring.java:81-91) */
static void bca_job(bca_ctx *c, int32_t bookmark, int directed, int normalize, jhm *bcv, jhm *rev) {
    jhm_clear(bcv);
    if (directed) {
        dowork_directed(c, bookmark, 0, bcv);
        jhm_clear(rev);
        dowork_directed(c, bookmark, 1, rev);     /* DirectedWeighted passes reverse=true, :23 */
        JHM_FOREACH(rev, e) jhm_merge_sum(bcv, rev->nodes[e].key, rev->nodes[e].val);
    } else {
        dowork_undirected(c, bookmark, bcv);
    }
    if (normalize == GEO_NORM_UNITY) bcv_to_unity(bcv, bookmark);
    else if (normalize == GEO_NORM_COUNTS) bcv_to_counts(bcv, bookmark);
}

static int bca_ctx_init(bca_ctx *c, int32_t V,
                        const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                        const int64_t *in_ptr, const int32_t *in_idx, const float *in_w,
                        double alpha, double epsilon) {
    memset(c, 0, sizeof(*c));
    c->V = V; c->out_ptr = out_ptr; c->out_idx = out_idx; c->out_w = out_w;
    c->in_ptr = in_ptr; c->in_idx = in_idx; c->in_w = in_w;
    c->alpha = alpha; c->epsilon = epsilon;
    c->paint = (double *)calloc((size_t)(V > 0 ? V : 1), sizeof(double));
    c->in_tree = (uint8_t *)calloc((size_t)(V > 0 ? V : 1), 1);
    return (c->paint && c->in_tree) ? 0 : -1;
}
static void bca_ctx_free(bca_ctx *c) { free(c->paint); free(c->in_tree); free(c->heap); }

/* Math.max(double,double) */
static double jmath_max(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0) { return signbit(a) ? b : a; }
    return (a >= b) ? a : b;
}

int geo_bca_build(int32_t V,
                  const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                  const int64_t *in_ptr, const int32_t *in_idx, const float *in_w,
                  double alpha, double epsilon, int directed, int normalize,
                  geo_coo *res) {
    bca_ctx c;
    if (bca_ctx_init(&c, V, out_ptr, out_idx, out_w, in_ptr, in_idx, in_w, alpha, epsilon)) return -1;
    jhm bcv, rev; jhm_init(&bcv); jhm_init(&rev);
    memset(res, 0, sizeof(*res));
    res->V = V;
    int64_t cap = (int64_t)V * 8 + 16;
    res->I = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    res->J = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    res->X = (float *)malloc(sizeof(float) * (size_t)cap);
    res->row_ptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)V + 1));
    res->max = 0;
    /* bookmarks in ascending id = completion order with threads: 1 */
    for (int32_t b = 0; b < V; b++) {
        res->row_ptr[b] = res->nnz;
        bca_job(&c, b, directed, normalize, &bcv, &rev);
        res->max = jmath_max(res->max, (double)bcv_max(&bcv));          /* :97 */
        if (res->nnz + bcv.size > cap) {
            while (res->nnz + bcv.size > cap) cap *= 2;
            res->I = (int32_t *)realloc(res->I, sizeof(int32_t) * (size_t)cap);
            res->J = (int32_t *)realloc(res->J, sizeof(int32_t) * (size_t)cap);
            res->X = (float *)realloc(res->X, sizeof(float) * (size_t)cap);
        }
        JHM_FOREACH(&bcv, e) {                                          /* :99-103 */
            res->I[res->nnz] = b; res->J[res->nnz] = bcv.nodes[e].key; res->X[res->nnz] = bcv.nodes[e].val;
            res->nnz++;
        }
    }
    res->row_ptr[V] = res->nnz;
    jhm_free(&bcv); jhm_free(&rev); bca_ctx_free(&c);
    return 0;
}

void geo_coo_free(geo_coo *c) {
    free(c->I); free(c->J); free(c->X); free(c->row_ptr);
    memset(c, 0, sizeof(*c));
}

int64_t geo_bca_single(int32_t V,
                  const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                  const int64_t *in_ptr, const int32_t *in_idx, const float *in_w,
                  double alpha, double epsilon, int directed, int normalize,
                  int32_t bookmark, int32_t *keys, float *vals, int64_t cap) {
    bca_ctx c;
    if (bca_ctx_init(&c, V, out_ptr, out_idx, out_w, in_ptr, in_idx, in_w, alpha, epsilon)) return -1;
    jhm bcv, rev; jhm_init(&bcv); jhm_init(&rev);
    bca_job(&c, bookmark, directed, normalize, &bcv, &rev);
    int64_t n = 0;
    JHM_FOREACH(&bcv, e) { if (n < cap) { keys[n] = bcv.nodes[e].key; vals[n] = bcv.nodes[e].val; } n++; }
    jhm_free(&bcv); jhm_free(&rev); bca_ctx_free(&c);
    return n;
}

/* Replays a sequence of map operations on one java.util.HashMap<Integer,Float> and returns its iteration order
 * (known-answer tests of the order emulation).  op 0 = BCV.add(key, 1f) (HashMap.put), 1 = merge(key, 1f, Float::sum),
 * 2 = remove(key).  *table_len = table.length afterwards, tree_bins = number of treeified bins. */
int64_t geo_hashmap_replay(int64_t n_ops, const int32_t *ops, const int32_t *keys,
                           int32_t *out_keys, int64_t cap, int32_t *table_len, int32_t *tree_bins) {
    jhm m; jhm_init(&m);
    for (int64_t k = 0; k < n_ops; k++) {
        if (ops[k] == 0) jhm_bcv_add(&m, keys[k], 1.0f);
        else if (ops[k] == 1) jhm_merge_sum(&m, keys[k], 1.0f);
        else jhm_remove(&m, keys[k]);
    }
    int64_t n = 0;
    JHM_FOREACH(&m, e) { if (n < cap) out_keys[n] = m.nodes[e].key; n++; }
    if (table_len) *table_len = m.tabcap;
    if (tree_bins) { int32_t t = 0; for (int32_t b = 0; b < m.tabcap; b++) t += (m.head[b] >= 0 && m.tree[b]); *tree_bins = t; }
    jhm_free(&m);
    return n;
}

/* ------------------------------------------------------------------------- */
/* GloVe / pGloVe cost + AdaGrad                                              */
/* ------------------------------------------------------------------------- */
struct geo_glove {
    int32_t V, D; int64_t N; int threads; int cost_kind; double xmax;
    int32_t *I, *J; float *X;
    float *focus, *context, *fbias, *cbias;
    float *gsf, *gsc, *gsfb, *gscb;      /* Adagrad.gradSq*  or  Adam/AMSGrad.M1* */
    float *m2f, *m2c, *m2fb, *m2cb;      /* Adam/AMSGrad.M2* (unused by Adagrad)  */
    int32_t *perm;
    geo_jrand rng;
    int opt_kind;
    int iteration;                       /* argument of createJob(id, iteration) for the next epoch */
};

static const float LEARNING_RATE = 0.05f;     /* J/opt/Optimizer.java:26 */

/* GloveCost / PGloveCost  J/opt/GloveCost.java:7-20, J/opt/PGloveCost.java:7-20 */
static inline float inner_cost(int kind, int32_t D, const float *foc, const float *ctx,
                               float fb, float cb, float Xij) {
    float ic = 0;
    for (int32_t d = 0; d < D; d++) ic += foc[d] * ctx[d];
    if (kind == GEO_COST_GLOVE)
        ic = (float)((double)ic + ((double)(fb + cb) - log((double)Xij)));
    else
        ic = (float)((double)ic + ((double)(fb + cb) - log((double)(Xij / (1 - Xij)))));
    return ic;
}
static inline float weighted_cost(int kind, double xmax, float ic, float Xij) {
    if (kind == GEO_COST_GLOVE)
        return ((double)Xij > xmax) ? ic : (float)pow((double)Xij / xmax, 0.75) * ic;
    return Xij * ic;
}

/* Adagrad.createJob body for one nonzero  J/opt/grad/Adagrad.java:51-93 */
static inline void adagrad_update(int kind, double xmax, int32_t D, int32_t bu, int32_t bv, float Xij,
                                  float *focus, float *context, float *fbias, float *cbias,
                                  float *gsf, float *gsc, float *gsfb, float *gscb, float *cost) {
    float *foc = focus + (int64_t)bu * D, *ctx = context + (int64_t)bv * D;
    float *g1s = gsf + (int64_t)bu * D,  *g2s = gsc + (int64_t)bv * D;
    const float ic = inner_cost(kind, D, foc, ctx, fbias[bu], cbias[bv], Xij);
    float wc = weighted_cost(kind, xmax, ic, Xij);
    *cost = (float)((double)*cost + 0.5 * wc * ic);                      /* :60 */
    for (int32_t d = 0; d < D; d++) {
        const float grad1 = wc * ctx[d];                                  /* :73 */
        const float grad2 = wc * foc[d];                                  /* :74 */
        foc[d] = (float)((double)foc[d] - grad1 / sqrt((double)g1s[d]) * LEARNING_RATE);  /* :76 */
        ctx[d] = (float)((double)ctx[d] - grad2 / sqrt((double)g2s[d]) * LEARNING_RATE);  /* :77 */
        g1s[d] += grad1 * grad1;                                          /* :79 */
        g2s[d] += grad2 * grad2;                                          /* :80 */
    }
    fbias[bu] = (float)((double)fbias[bu] - wc / sqrt((double)gsfb[bu]));  /* :88 (no lr) */
    cbias[bv] = (float)((double)cbias[bv] - wc / sqrt((double)gscb[bv]));  /* :89 */
    wc *= wc;                                                              /* :90 */
    gsfb[bu] += wc;                                                        /* :92 */
    gscb[bv] += wc;                                                        /* :93 */
}

/* FastMath.max(float,float) of commons-math 2.x */
static inline float fastmath_maxf(float a, float b) { return (a <= b) ? b : ((a + b) != (a + b) ? NAN : a); }

static const float ADAM_BETA1 = 0.9f, ADAM_BETA2 = 0.999f, ADAM_EPS = 1e-7f;    /* J/opt/grad/Adam.java:45-53 */

/* Adam.createJob's per-job constant  J/opt/grad/Adam.java:84 */
static double adam_correction(int iteration) {
    return LEARNING_RATE * sqrt(1 - pow(ADAM_BETA2, iteration + 1)) / (1 - pow(ADAM_BETA1, iteration + 1));
}

/* Adam.createJob body (J/opt/grad/Adam.java:86-146) and AMSGrad.createJob body (J/opt/grad/AMSGrad.java:100-160)
 * for one nonzero.  m1* live in the gs* arrays of the shared state, m2* in the m2* arrays. */
static inline void adam_update(int ams, double correction, int kind, double xmax, int32_t D, int32_t bu, int32_t bv, float Xij,
                               float *focus, float *context, float *fbias, float *cbias,
                               float *M1f, float *M1c, float *M1fb, float *M1cb,
                               float *M2f, float *M2c, float *M2fb, float *M2cb, float *cost) {
    float *foc = focus + (int64_t)bu * D, *ctx = context + (int64_t)bv * D;
    float *m1f = M1f + (int64_t)bu * D, *m1c = M1c + (int64_t)bv * D;
    float *m2f = M2f + (int64_t)bu * D, *m2c = M2c + (int64_t)bv * D;
    const float beta1 = ADAM_BETA1, beta2 = ADAM_BETA2, epsilon = ADAM_EPS;
    const float ic = inner_cost(kind, D, foc, ctx, fbias[bu], cbias[bv], Xij);
    const float wc = weighted_cost(kind, xmax, ic, Xij);
    *cost = (float)((double)*cost + 0.5 * wc * ic);
    for (int32_t d = 0; d < D; d++) {
        const float grad_u = wc * ctx[d];
        const float grad_v = wc * foc[d];
        const float m1 = beta1 * m1f[d] + (1 - beta1) * grad_u;
        const float m2 = beta1 * m1c[d] + (1 - beta1) * grad_v;
        float v1, v2;
        if (!ams) {
            v1 = beta2 * m2f[d] + (1 - beta2) * (grad_u * grad_u);
            v2 = beta2 * m2c[d] + (1 - beta2) * (grad_v * grad_v);
            foc[d] = (float)((double)foc[d] - correction * m1 / (sqrt((double)v1) + epsilon));      /* Adam.java:118 */
            ctx[d] = (float)((double)ctx[d] - correction * m2 / (sqrt((double)v2) + epsilon));
        } else {
            v1 = fastmath_maxf(m2f[d], beta2 * m2f[d] + (1 - beta2) * (grad_u * grad_u));          /* AMSGrad.java:129-130 */
            v2 = fastmath_maxf(m2c[d], beta2 * m2c[d] + (1 - beta2) * (grad_v * grad_v));
            foc[d] = (float)((double)foc[d] - LEARNING_RATE / (sqrt((double)v1) + epsilon) * m1);   /* AMSGrad.java:133 */
            ctx[d] = (float)((double)ctx[d] - LEARNING_RATE / (sqrt((double)v2) + epsilon) * m2);
        }
        m1f[d] = m1; m1c[d] = m2; m2f[d] = v1; m2c[d] = v2;
    }
    const float m1 = beta1 * M1fb[bu] + (1 - beta1) * wc;
    const float m2 = beta1 * M1cb[bv] + (1 - beta1) * wc;
    float v1, v2;
    if (!ams) {
        v1 = beta2 * M2fb[bu] + (1 - beta2) * (wc * wc);
        v2 = beta2 * M2cb[bv] + (1 - beta2) * (wc * wc);
        fbias[bu] = (float)((double)fbias[bu] - correction * m1 / (sqrt((double)v1) + epsilon));
        cbias[bv] = (float)((double)cbias[bv] - correction * m2 / (sqrt((double)v2) + epsilon));
    } else {
        v1 = fastmath_maxf(M2fb[bu], beta2 * M2fb[bu] + (1 - beta2) * (wc * wc));
        v2 = fastmath_maxf(M2cb[bv], beta2 * M2cb[bv] + (1 - beta2) * (wc * wc));
        fbias[bu] = (float)((double)fbias[bu] - LEARNING_RATE / (sqrt((double)v1) + epsilon) * m1);
        cbias[bv] = (float)((double)cbias[bv] - LEARNING_RATE / (sqrt((double)v2) + epsilon) * m2);
    }
    M1fb[bu] = m1; M1cb[bv] = m2; M2fb[bu] = v1; M2cb[bv] = v2;
}

float geo_opt_job(int opt_kind, int iteration, int32_t D, int64_t n, const int32_t *I, const int32_t *J, const float *X,
                  double xmax, int cost_kind,
                  float *focus, float *context, float *fbias, float *cbias,
                  float *s1f, float *s1c, float *s1fb, float *s1cb,
                  float *s2f, float *s2c, float *s2fb, float *s2cb) {
    float cost = 0;
    if (opt_kind == GEO_OPT_ADAGRAD) {
        for (int64_t k = 0; k < n; k++)
            adagrad_update(cost_kind, xmax, D, I[k], J[k], X[k], focus, context, fbias, cbias, s1f, s1c, s1fb, s1cb, &cost);
    } else {
        const double correction = adam_correction(iteration);
        for (int64_t k = 0; k < n; k++)
            adam_update(opt_kind == GEO_OPT_AMSGRAD, correction, cost_kind, xmax, D, I[k], J[k], X[k], focus, context, fbias, cbias,
                        s1f, s1c, s1fb, s1cb, s2f, s2c, s2fb, s2cb, &cost);
    }
    return cost;
}

float geo_adagrad_job(int32_t D, int64_t n, const int32_t *I, const int32_t *J, const float *X,
                      double xmax, int cost_kind,
                      float *focus, float *context, float *fbias, float *cbias,
                      float *gsf, float *gsc, float *gsfb, float *gscb) {
    float cost = 0;
    for (int64_t k = 0; k < n; k++)
        adagrad_update(cost_kind, xmax, D, I[k], J[k], X[k], focus, context, fbias, cbias,
                       gsf, gsc, gsfb, gscb, &cost);
    return cost;
}

geo_glove *geo_glove_create(int32_t V, int32_t D, int64_t N,
                            const int32_t *I, const int32_t *J, const float *X,
                            double xmax, int cost_kind, int64_t seed, int threads) {
    return geo_glove_create_opt(V, D, N, I, J, X, xmax, cost_kind, seed, threads, GEO_OPT_ADAGRAD);
}

geo_glove *geo_glove_create_opt(int32_t V, int32_t D, int64_t N,
                            const int32_t *I, const int32_t *J, const float *X,
                            double xmax, int cost_kind, int64_t seed, int threads, int opt_kind) {
    geo_glove *g = (geo_glove *)calloc(1, sizeof(*g));
    if (!g) return NULL;
    g->opt_kind = opt_kind;
    g->V = V; g->D = D; g->N = N; g->threads = threads < 1 ? 1 : threads;
    g->cost_kind = cost_kind; g->xmax = xmax;
    size_t vd = (size_t)V * (size_t)D;
    g->I = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    g->J = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    g->X = (float *)malloc(sizeof(float) * (size_t)(N ? N : 1));
    g->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    g->focus = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->context = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->gsf = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->gsc = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->fbias = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->cbias = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->gsfb = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->gscb = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->m2f = (float *)calloc((vd ? vd : 1), sizeof(float));
    g->m2c = (float *)calloc((vd ? vd : 1), sizeof(float));
    g->m2fb = (float *)calloc((size_t)(V ? V : 1), sizeof(float));
    g->m2cb = (float *)calloc((size_t)(V ? V : 1), sizeof(float));
    memcpy(g->I, I, sizeof(int32_t) * (size_t)N);
    memcpy(g->J, J, sizeof(int32_t) * (size_t)N);
    memcpy(g->X, X, sizeof(float) * (size_t)N);
    geo_jrand_init(&g->rng, seed);
    /* Optimizer ctor  J/opt/Optimizer.java:50-57 */
    for (int32_t i = 0; i < V; i++) {
        g->fbias[i] = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
        g->cbias[i] = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
        for (int32_t d = 0; d < D; d++) {
            g->focus[(size_t)i * D + d]   = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
            g->context[(size_t)i * D + d] = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
        }
    }
    /* Adagrad ctor  J/opt/grad/Adagrad.java:27-33: gradSq = 1;  Adam / AMSGrad ctors: moments = 0 (new float[]) */
    const float init1 = opt_kind == GEO_OPT_ADAGRAD ? 1.0f : 0.0f;
    for (size_t k = 0; k < vd; k++) g->gsf[k] = g->gsc[k] = init1;
    for (int32_t i = 0; i < V; i++) g->gsfb[i] = g->gscb[i] = init1;
    /* Permutation ctor  J/util/rnd/Permutation.java:11-15 */
    for (int64_t k = 0; k < N; k++) g->perm[k] = (int32_t)k;
    return g;
}

void geo_glove_destroy(geo_glove *g) {
    if (!g) return;
    free(g->I); free(g->J); free(g->X); free(g->perm);
    free(g->focus); free(g->context); free(g->gsf); free(g->gsc);
    free(g->fbias); free(g->cbias); free(g->gsfb); free(g->gscb);
    free(g->m2f); free(g->m2c); free(g->m2fb); free(g->m2cb);
    free(g);
}

typedef struct { geo_glove *g; int id; float cost; } job_arg;

/* Adagrad.createJob(id, iteration)  J/opt/grad/Adagrad.java:42-98 */
static void *run_job(void *p) {
    job_arg *a = (job_arg *)p;
    geo_glove *g = a->g;
    const int T = g->threads;
    const int64_t per = g->N / T;
    const int64_t offset = per * a->id;                                   /* :47 */
    const int64_t lines = (a->id == T - 1) ? per + g->N % T : per;        /* Optimizer.java:59-63 */
    float cost = 0;
    const double correction = adam_correction(g->iteration);
    for (int64_t i = 0; i < lines; i++) {
        const int32_t p2 = g->perm[i + offset];                           /* BookmarkColoring.java:127-137 */
        if (g->opt_kind == GEO_OPT_ADAGRAD)
            adagrad_update(g->cost_kind, g->xmax, g->D, g->I[p2], g->J[p2], g->X[p2],
                           g->focus, g->context, g->fbias, g->cbias, g->gsf, g->gsc, g->gsfb, g->gscb, &cost);
        else
            adam_update(g->opt_kind == GEO_OPT_AMSGRAD, correction, g->cost_kind, g->xmax, g->D, g->I[p2], g->J[p2], g->X[p2],
                        g->focus, g->context, g->fbias, g->cbias, g->gsf, g->gsc, g->gsfb, g->gscb,
                        g->m2f, g->m2c, g->m2fb, g->m2cb, &cost);
    }
    a->cost = cost;
    return NULL;
}

double geo_glove_epoch_noshuffle(geo_glove *g, int race) {
    const int T = g->threads;
    job_arg *args = (job_arg *)calloc((size_t)T, sizeof(job_arg));
    double local = 0;
    if (race && T > 1) {
        pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
        for (int t = 0; t < T; t++) { args[t].g = g; args[t].id = t; pthread_create(&th[t], NULL, run_job, &args[t]); }
        for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); local += args[t].cost; }
        free(th);
    } else {
        for (int t = 0; t < T; t++) { args[t].g = g; args[t].id = t; run_job(&args[t]); local += args[t].cost; }
    }
    free(args);
    g->iteration++;
    return g->N ? local / (double)g->N : local / 0.0;                     /* Optimizer.java:96 */
}

double geo_glove_epoch(geo_glove *g, int race) {
    geo_jrand_shuffle(&g->rng, g->perm, (int32_t)g->N);                   /* Optimizer.java:79 */
    return geo_glove_epoch_noshuffle(g, race);
}

int geo_glove_optimize(geo_glove *g, int maxiter, double tolerance,
                       double *history, double *final_cost, int race) {
    /* Optimizer.optimize  J/opt/Optimizer.java:66-120 */
    double prev = 0, fin = 0; int it;
    for (it = 0; it < maxiter; it++) {
        double local = geo_glove_epoch(g, race);
        if (history) history[it] = local;
        double diff = fabs(prev - local);
        prev = local;
        if (diff <= tolerance) { fin = local; it++; break; }
    }
    if (final_cost) *final_cost = fin;
    return it;
}

void geo_glove_extract(const geo_glove *g, double *out) {
    /* J/opt/Optimizer.java:129-140: float add, float /2, widened on store */
    size_t vd = (size_t)g->V * (size_t)g->D;
    for (size_t k = 0; k < vd; k++) out[k] = (g->focus[k] + g->context[k]) / 2;
}

float   *geo_glove_focus(geo_glove *g)       { return g->focus; }
float   *geo_glove_context(geo_glove *g)     { return g->context; }
float   *geo_glove_fbias(geo_glove *g)       { return g->fbias; }
float   *geo_glove_cbias(geo_glove *g)       { return g->cbias; }
float   *geo_glove_gsq_focus(geo_glove *g)   { return g->gsf; }
float   *geo_glove_gsq_context(geo_glove *g) { return g->gsc; }
float   *geo_glove_gsq_fbias(geo_glove *g)   { return g->gsfb; }
float   *geo_glove_gsq_cbias(geo_glove *g)   { return g->gscb; }
float   *geo_glove_m2_focus(geo_glove *g)    { return g->m2f; }
float   *geo_glove_m2_context(geo_glove *g)  { return g->m2c; }
float   *geo_glove_m2_fbias(geo_glove *g)    { return g->m2fb; }
float   *geo_glove_m2_cbias(geo_glove *g)    { return g->m2cb; }
void     geo_glove_set_iteration(geo_glove *g, int it) { g->iteration = it; }
int32_t *geo_glove_perm(geo_glove *g)        { return g->perm; }
uint64_t geo_glove_rng_state(const geo_glove *g) { return g->rng.seed; }

/* ------------------------------------------------------------------------- */
/* String.format("%11.6E")                                                    */
/* ------------------------------------------------------------------------- */
int geo_format_11_6E(double v, char *buf, int buflen) {
    if (v != v) return snprintf(buf, (size_t)buflen, "%11s", "NaN");
    if (isinf(v)) return snprintf(buf, (size_t)buflen, "%11s", v > 0 ? "Infinity" : "-Infinity");
    char tmp[64];
    int neg = signbit(v) ? 1 : 0;
    double a = fabs(v);
    /* shortest decimal that round-trips (what FloatingDecimal hands the Formatter) */
    int prec;
    for (prec = 0; prec <= 17; prec++) {
        snprintf(tmp, sizeof tmp, "%.*e", prec, a);
        if (strtod(tmp, NULL) == a) break;
    }
    /* tmp = d.ddddde[+-]XX ; collect digits + exponent */
    char digits[32]; int nd = 0; int ex = 0;
    char *ep = strchr(tmp, 'e');
    ex = atoi(ep + 1);
    for (char *p = tmp; p < ep; p++) if (*p >= '0' && *p <= '9') digits[nd++] = *p;
    /* round HALF_UP to 7 significant digits */
    int want = 7;
    int d7[8];
    for (int k = 0; k < want; k++) d7[k] = (k < nd) ? digits[k] - '0' : 0;
    if (nd > want && digits[want] >= '5') {
        int k = want - 1;
        while (k >= 0) { if (++d7[k] < 10) break; d7[k] = 0; k--; }
        if (k < 0) { for (int q = want - 1; q > 0; q--) d7[q] = d7[q - 1]; d7[0] = 1; ex += 1; }
    }
    if (a == 0.0) ex = 0;
    char body[40];
    int n = snprintf(body, sizeof body, "%s%d.%d%d%d%d%d%dE%c%02d", neg ? "-" : "",
                     d7[0], d7[1], d7[2], d7[3], d7[4], d7[5], d7[6],
                     ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
    (void)n;
    return snprintf(buf, (size_t)buflen, "%11s", body);
}
