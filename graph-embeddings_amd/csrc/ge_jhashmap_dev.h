// ge_jhashmap_dev.h -- exact iteration order of a java.util.HashMap<Integer,Float> (JDK 8), device side.
//
// BCV extends HashMap (J/bca/util/BCV.java:14) and its iteration order is the COO order of a row
// (J/bca/BookmarkColoring.java:99-103).  k_bca ranks a row's keys by (bin, class, sequence), which is the order
// of a map whose bins are plain lists and whose table only grows by load factor.  Two JDK mechanisms break that
// and are replayed here, sequentially, by ONE lane, for the rare rows that meet them (bca.hip: bins_may_treeify):
//   * treeifyBin with table.length < 64 (MIN_TREEIFY_CAPACITY) only calls resize(): a bin that receives its 9th
//     node through putVal -- or its 8th through merge(), whose binCount counts every node -- doubles the table early;
//   * at table.length >= 64 such a bin becomes a red-black tree of TreeNodes ordered by hash.  Iteration still
//     follows `next`: the tree ROOT is moved to the front of the bin (moveRootToFront), a later key is linked behind
//     its tree PARENT (putTreeVal), resize() splits a tree bin in `next` order and turns lists of <= 6 nodes back
//     into plain bins (TreeNode.split / untreeify), remove() may untreeify a small tree or move a new root forward.
// Keys are distinct Integers: hash = k ^ (k >>> 16) is a bijection, so tree order = signed order of the hashes and
// no tieBreakOrder is ever needed.  Nodes are the row's entries e = 0..n-1 (index into BcaWork.touched).
#pragma once
#include <stdint.h>

namespace gejm {

struct Map {
    // per entry
    int32_t *key, *next, *prev, *parent, *left, *right, *red;
    // per bin (capacity slots each)
    int32_t *head, *tree;
    int32_t cap, thr, size, max_cap;
    bool overflow;
};

__device__ __forceinline__ int32_t jhash(int32_t k) { const uint32_t h = (uint32_t)k; return (int32_t)(h ^ (h >> 16)); }

__device__ inline void reset(Map &m) { m.cap = 0; m.thr = 0; m.size = 0; m.overflow = false; }

__device__ inline int32_t rotate_left(Map &m, int32_t root, int32_t p) {
    int32_t r, pp, rl;
    if (p >= 0 && (r = m.right[p]) >= 0) {
        if ((rl = m.right[p] = m.left[r]) >= 0) m.parent[rl] = p;
        if ((pp = m.parent[r] = m.parent[p]) < 0) { root = r; m.red[r] = 0; }
        else if (m.left[pp] == p) m.left[pp] = r;
        else m.right[pp] = r;
        m.left[r] = p;
        m.parent[p] = r;
    }
    return root;
}
__device__ inline int32_t rotate_right(Map &m, int32_t root, int32_t p) {
    int32_t l, pp, lr;
    if (p >= 0 && (l = m.left[p]) >= 0) {
        if ((lr = m.left[p] = m.right[l]) >= 0) m.parent[lr] = p;
        if ((pp = m.parent[l] = m.parent[p]) < 0) { root = l; m.red[l] = 0; }
        else if (m.right[pp] == p) m.right[pp] = l;
        else m.left[pp] = l;
        m.right[l] = p;
        m.parent[p] = l;
    }
    return root;
}
// TreeNode.balanceInsertion
__device__ inline int32_t balance_insertion(Map &m, int32_t root, int32_t x) {
    m.red[x] = 1;
    for (;;) {
        int32_t xp = m.parent[x], xpp, xppl, xppr;
        if (xp < 0) { m.red[x] = 0; return x; }
        if (!m.red[xp] || (xpp = m.parent[xp]) < 0) return root;
        if (xp == (xppl = m.left[xpp])) {
            if ((xppr = m.right[xpp]) >= 0 && m.red[xppr]) { m.red[xppr] = 0; m.red[xp] = 0; m.red[xpp] = 1; x = xpp; }
            else {
                if (x == m.right[xp]) {
                    x = xp; root = rotate_left(m, root, x);
                    xp = m.parent[x]; xpp = xp < 0 ? -1 : m.parent[xp];
                }
                if (xp >= 0) {
                    m.red[xp] = 0;
                    if (xpp >= 0) { m.red[xpp] = 1; root = rotate_right(m, root, xpp); }
                }
            }
        } else {
            if (xppl >= 0 && m.red[xppl]) { m.red[xppl] = 0; m.red[xp] = 0; m.red[xpp] = 1; x = xpp; }
            else {
                if (x == m.left[xp]) {
                    x = xp; root = rotate_right(m, root, x);
                    xp = m.parent[x]; xpp = xp < 0 ? -1 : m.parent[xp];
                }
                if (xp >= 0) {
                    m.red[xp] = 0;
                    if (xpp >= 0) { m.red[xpp] = 1; root = rotate_left(m, root, xpp); }
                }
            }
        }
    }
}
// TreeNode.balanceDeletion
__device__ inline int32_t balance_deletion(Map &m, int32_t root, int32_t x) {
    for (;;) {
        int32_t xp, xpl, xpr;
        if (x < 0 || x == root) return root;
        if ((xp = m.parent[x]) < 0) { m.red[x] = 0; return x; }
        if (m.red[x]) { m.red[x] = 0; return root; }
        if ((xpl = m.left[xp]) == x) {
            if ((xpr = m.right[xp]) >= 0 && m.red[xpr]) {
                m.red[xpr] = 0; m.red[xp] = 1;
                root = rotate_left(m, root, xp);
                xp = m.parent[x]; xpr = xp < 0 ? -1 : m.right[xp];
            }
            if (xpr < 0) x = xp;
            else {
                int32_t sl = m.left[xpr], sr = m.right[xpr];
                if ((sr < 0 || !m.red[sr]) && (sl < 0 || !m.red[sl])) { m.red[xpr] = 1; x = xp; }
                else {
                    if (sr < 0 || !m.red[sr]) {
                        if (sl >= 0) m.red[sl] = 0;
                        m.red[xpr] = 1;
                        root = rotate_right(m, root, xpr);
                        xp = m.parent[x]; xpr = xp < 0 ? -1 : m.right[xp];
                    }
                    if (xpr >= 0) {
                        m.red[xpr] = xp < 0 ? 0 : m.red[xp];
                        if ((sr = m.right[xpr]) >= 0) m.red[sr] = 0;
                    }
                    if (xp >= 0) { m.red[xp] = 0; root = rotate_left(m, root, xp); }
                    x = root;
                }
            }
        } else {
            if (xpl >= 0 && m.red[xpl]) {
                m.red[xpl] = 0; m.red[xp] = 1;
                root = rotate_right(m, root, xp);
                xp = m.parent[x]; xpl = xp < 0 ? -1 : m.left[xp];
            }
            if (xpl < 0) x = xp;
            else {
                int32_t sl = m.left[xpl], sr = m.right[xpl];
                if ((sl < 0 || !m.red[sl]) && (sr < 0 || !m.red[sr])) { m.red[xpl] = 1; x = xp; }
                else {
                    if (sl < 0 || !m.red[sl]) {
                        if (sr >= 0) m.red[sr] = 0;
                        m.red[xpl] = 1;
                        root = rotate_left(m, root, xpl);
                        xp = m.parent[x]; xpl = xp < 0 ? -1 : m.left[xp];
                    }
                    if (xpl >= 0) {
                        m.red[xpl] = xp < 0 ? 0 : m.red[xp];
                        if ((sl = m.left[xpl]) >= 0) m.red[sl] = 0;
                    }
                    if (xp >= 0) { m.red[xp] = 0; root = rotate_right(m, root, xp); }
                    x = root;
                }
            }
        }
    }
}
__device__ inline int32_t root_of(const Map &m, int32_t e) { while (m.parent[e] >= 0) e = m.parent[e]; return e; }

// TreeNode.moveRootToFront
__device__ inline void move_root_to_front(Map &m, int32_t cap, int32_t root) {
    if (root < 0) return;
    const int32_t index = jhash(m.key[root]) & (cap - 1);
    const int32_t first = m.head[index];
    if (root != first) {
        m.head[index] = root;
        const int32_t rp = m.prev[root], rn = m.next[root];
        if (rn >= 0) m.prev[rn] = rp;
        if (rp >= 0) m.next[rp] = rn;
        if (first >= 0) m.prev[first] = root;
        m.next[root] = first;
        m.prev[root] = -1;
    }
}
// TreeNode.treeify over the chain that starts at `first`
__device__ inline void treeify(Map &m, int32_t cap, int32_t first) {
    int32_t root = -1;
    for (int32_t x = first, nx; x >= 0; x = nx) {
        nx = m.next[x];
        m.left[x] = m.right[x] = -1;
        if (root < 0) { m.parent[x] = -1; m.red[x] = 0; root = x; continue; }
        const int32_t h = jhash(m.key[x]);
        for (int32_t p = root;;) {
            const bool go_left = jhash(m.key[p]) > h;
            const int32_t xp = p;
            p = go_left ? m.left[p] : m.right[p];
            if (p < 0) {
                m.parent[x] = xp;
                if (go_left) m.left[xp] = x; else m.right[xp] = x;
                root = balance_insertion(m, root, x);
                break;
            }
        }
    }
    move_root_to_front(m, cap, root);
}

// HashMap.resize, in place: old bin j splits into j (lo) and j + oldcap (hi), relative order kept
__device__ inline void resize(Map &m) {
    const int32_t oldcap = m.cap, newcap = oldcap ? oldcap * 2 : 16;
    if (newcap > m.max_cap) { m.overflow = true; return; }
    for (int32_t b = oldcap; b < newcap; ++b) { m.head[b] = -1; m.tree[b] = 0; }
    m.cap = newcap;
    m.thr = (newcap / 4) * 3;
    for (int32_t j = 0; j < oldcap; ++j) {
        int32_t e = m.head[j];
        if (e < 0) continue;
        const bool was_tree = m.tree[j] != 0;
        int32_t lo_h = -1, lo_t = -1, hi_h = -1, hi_t = -1, lc = 0, hc = 0;
        for (int32_t nx; e >= 0; e = nx) {
            nx = m.next[e];
            m.next[e] = -1;
            if ((jhash(m.key[e]) & oldcap) == 0) {
                if ((m.prev[e] = lo_t) < 0) lo_h = e; else m.next[lo_t] = e;
                lo_t = e; ++lc;
            } else {
                if ((m.prev[e] = hi_t) < 0) hi_h = e; else m.next[hi_t] = e;
                hi_t = e; ++hc;
            }
        }
        m.head[j] = lo_h; m.head[j + oldcap] = hi_h;
        m.tree[j] = 0; m.tree[j + oldcap] = 0;
        if (!was_tree) continue;
        // TreeNode.split: <= 6 nodes untreeify (a plain bin in this order); a side that kept every node is still a valid tree
        if (lo_h >= 0 && lc > 6) { m.tree[j] = 1; if (hi_h >= 0) treeify(m, newcap, lo_h); }
        if (hi_h >= 0 && hc > 6) { m.tree[j + oldcap] = 1; if (lo_h >= 0) treeify(m, newcap, hi_h); }
    }
}

// HashMap.treeifyBin for the bin of entry e
__device__ inline void treeify_bin(Map &m, int32_t e) {
    if (m.cap < 64) { resize(m); return; }
    const int32_t b = jhash(m.key[e]) & (m.cap - 1);
    int32_t tl = -1;
    for (int32_t q = m.head[b]; q >= 0; q = m.next[q]) { m.prev[q] = tl; tl = q; }
    m.tree[b] = 1;
    treeify(m, m.cap, m.head[b]);
}

// TreeNode.putTreeVal for an absent key: the node is linked behind its tree parent
__device__ inline void put_tree_val(Map &m, int32_t b, int32_t x) {
    const int32_t h = jhash(m.key[x]);
    const int32_t root = root_of(m, m.head[b]);
    for (int32_t p = root;;) {
        const bool go_left = jhash(m.key[p]) > h;
        const int32_t xp = p;
        p = go_left ? m.left[p] : m.right[p];
        if (p < 0) {
            const int32_t xpn = m.next[xp];
            m.next[x] = xpn;
            m.left[x] = m.right[x] = -1;
            if (go_left) m.left[xp] = x; else m.right[xp] = x;
            m.next[xp] = x;
            m.parent[x] = m.prev[x] = xp;
            if (xpn >= 0) m.prev[xpn] = x;
            move_root_to_front(m, m.cap, balance_insertion(m, root, x));
            return;
        }
    }
}

// HashMap.putVal of an absent key (BCV.add, J/bca/util/BCV.java:35-37): bin tail; the 9th node calls treeifyBin
__device__ inline void put_new(Map &m, int32_t e) {
    if (!m.cap) resize(m);
    if (m.overflow) return;
    const int32_t b = jhash(m.key[e]) & (m.cap - 1);
    m.next[e] = -1; m.prev[e] = -1; m.parent[e] = -1; m.left[e] = -1; m.right[e] = -1; m.red[e] = 0;
    if (m.head[b] < 0) m.head[b] = e;
    else if (m.tree[b]) put_tree_val(m, b, e);
    else {
        int32_t t = m.head[b], bin_count = 0;
        while (m.next[t] >= 0) { t = m.next[t]; ++bin_count; }
        m.next[t] = e;
        if (bin_count >= 7) treeify_bin(m, e);
    }
    if (++m.size > m.thr) resize(m);
}
// HashMap.merge (BCV.merge, J/bca/util/BCV.java:105-107): resize BEFORE the lookup; an absent key goes to the bin HEAD,
// the 8th node calls treeifyBin; no resize afterwards.  `absent` = false: only the resize check runs.
__device__ inline void merge(Map &m, int32_t e, bool absent) {
    if (m.size > m.thr || !m.cap) resize(m);
    if (!absent || m.overflow) return;
    const int32_t b = jhash(m.key[e]) & (m.cap - 1);
    m.next[e] = -1; m.prev[e] = -1; m.parent[e] = -1; m.left[e] = -1; m.right[e] = -1; m.red[e] = 0;
    if (m.head[b] >= 0 && m.tree[b]) put_tree_val(m, b, e);
    else {
        int32_t bin_count = 0;
        for (int32_t t = m.head[b]; t >= 0; t = m.next[t]) ++bin_count;
        m.next[e] = m.head[b];
        m.head[b] = e;
        if (bin_count >= 7) treeify_bin(m, e);
    }
    ++m.size;
}

// HashMap.remove of a present key -> removeNode (-> TreeNode.removeTreeNode, movable = true)
__device__ inline void remove(Map &m, int32_t p) {
    const int32_t b = jhash(m.key[p]) & (m.cap - 1);
    --m.size;
    if (!m.tree[b]) {
        if (m.head[b] == p) m.head[b] = m.next[p];
        else { int32_t q = m.head[b]; while (m.next[q] != p) q = m.next[q]; m.next[q] = m.next[p]; }
        return;
    }
    int32_t first = m.head[b], root = first, rl;
    const int32_t succ = m.next[p], pred = m.prev[p];
    if (pred < 0) m.head[b] = first = succ; else m.next[pred] = succ;
    if (succ >= 0) m.prev[succ] = pred;
    if (first < 0) { m.tree[b] = 0; return; }
    if (m.parent[root] >= 0) root = root_of(m, root);
    if (m.right[root] < 0 || (rl = m.left[root]) < 0 || m.left[rl] < 0) { m.tree[b] = 0; return; }   // too small: untreeify
    const int32_t pl = m.left[p], pr = m.right[p];
    int32_t replacement;
    if (pl >= 0 && pr >= 0) {
        int32_t s = pr, sl;
        while ((sl = m.left[s]) >= 0) s = sl;
        const int32_t c = m.red[s]; m.red[s] = m.red[p]; m.red[p] = c;
        const int32_t sr = m.right[s], pp = m.parent[p];
        if (s == pr) { m.parent[p] = s; m.right[s] = p; }
        else {
            const int32_t sp = m.parent[s];
            if ((m.parent[p] = sp) >= 0) { if (s == m.left[sp]) m.left[sp] = p; else m.right[sp] = p; }
            if ((m.right[s] = pr) >= 0) m.parent[pr] = s;
        }
        m.left[p] = -1;
        if ((m.right[p] = sr) >= 0) m.parent[sr] = p;
        if ((m.left[s] = pl) >= 0) m.parent[pl] = s;
        if ((m.parent[s] = pp) < 0) root = s;
        else if (p == m.left[pp]) m.left[pp] = s;
        else m.right[pp] = s;
        replacement = sr >= 0 ? sr : p;
    }
    else if (pl >= 0) replacement = pl;
    else if (pr >= 0) replacement = pr;
    else replacement = p;
    if (replacement != p) {
        const int32_t pp = m.parent[replacement] = m.parent[p];
        if (pp < 0) root = replacement;
        else if (p == m.left[pp]) m.left[pp] = replacement;
        else m.right[pp] = replacement;
        m.left[p] = m.right[p] = m.parent[p] = -1;
    }
    const int32_t r = m.red[p] ? root : balance_deletion(m, root, replacement);
    if (replacement == p) {
        const int32_t pp = m.parent[p];
        m.parent[p] = -1;
        if (pp >= 0) {
            if (p == m.left[pp]) m.left[pp] = -1;
            else if (p == m.right[pp]) m.right[pp] = -1;
        }
    }
    move_root_to_front(m, m.cap, r);
}

}  // namespace gejm
