"""Static check of the hand-counted `s_waitcnt vmcnt(N)` of k_adagrad_runs (csrc/glove.hip, "The wait is counted by hand").

The kernel requests the streamed rows of the NEXT nonzero with `buffer_load ... lds` into one of two LDS images and, one step
later, waits for that image with vmcnt(N_AFTER): vector-memory instructions complete in order, so the wait is long enough only if,
on EVERY path through the code, at least N_AFTER vector-memory instructions are issued behind the request it waits for.  This
module disassembles a code object (llvm-objdump), builds the control-flow graph of one kernel and computes, by a min-over-paths
dataflow, how many vector-memory instructions have been issued behind the request in use (a request is N_DMA loads into LDS;
the most recent N_DMA are the request for the other image, issued just before the wait; the ones before those are the request
the wait is for).  A compiler change that drops, predicates away
or moves one of the counted instructions -- or puts a spill (scratch_*) into the walk -- makes that minimum fall below N.
"""
import re

INF = 1 << 20
_VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")
_BRANCH = re.compile(r"^s_(cbranch_\w+|branch)$")


def parse_kernels(asm_text):
    """{symbol: [(addr, mnemonic, operands)]} from `llvm-objdump -d` output."""
    out, cur = {}, None
    for line in asm_text.splitlines():
        m = re.match(r"^[0-9a-f]{16} <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or "//" not in line:
            continue
        code, _, tail = line.partition("//")
        am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        parts = code.split(None, 1)
        if not am or not parts:
            continue
        cur.append((int(am.group(1), 16), parts[0], parts[1].strip() if len(parts) > 1 else ""))
    return out


def _vmcnt(operands):
    m = re.search(r"vmcnt\((\d+)\)", operands)
    return int(m.group(1)) if m else None


def analyse(instrs, group):
    """Min-over-paths state BEFORE every instruction: a tuple k of group + 1 counts, k[j] = vector-memory instructions issued behind
    the (j+1)-th most recent `... lds` load (INF: that load is known complete, or does not exist).  A request is `group` loads into
    LDS; at a wait the image in use was requested by the loads numbered group+1 .. 2*group from the most recent backwards (the
    most recent `group` are the request for the other image), so the wait vmcnt(N) covers it iff N <= k[group] on every path.
    Returns (states, problems): problems lists branch targets outside the kernel (none expected)."""
    index = {a: i for i, (a, _, _) in enumerate(instrs)}
    n = len(instrs)
    succ = [[] for _ in range(n)]
    problems = []
    for i, (addr, mn, ops) in enumerate(instrs):
        nxt = i + 1 if i + 1 < n else None
        if _BRANCH.match(mn):
            off = int(ops.split()[0])
            if off >= 1 << 15:
                off -= 1 << 16
            size = (instrs[i + 1][0] - addr) if nxt is not None else 4
            tgt = index.get(addr + size + 4 * off)
            if tgt is None:
                problems.append("branch at %#x leaves the kernel" % addr)
            else:
                succ[i].append(tgt)
            if mn != "s_branch" and nxt is not None:
                succ[i].append(nxt)
        elif mn == "s_endpgm":
            pass
        elif mn.startswith("s_setpc") or mn.startswith("s_swappc"):
            problems.append("indirect jump at %#x" % addr)
        elif nxt is not None:
            succ[i].append(nxt)
    state = [None] * n
    state[0] = (INF,) * (group + 1)
    work = [0]
    while work:
        i = work.pop()
        k = state[i]
        _, mn, ops = instrs[i]
        if _VMEM.match(mn):
            if re.search(r"\blds\b", ops):
                k = (0,) + tuple(min(x + 1, INF) for x in k[:-1])
            else:
                k = tuple(min(x + 1, INF) for x in k)
        elif mn == "s_waitcnt":
            w = _vmcnt(ops)
            if w is not None:
                k = tuple(INF if w <= x else x for x in k)
        for j in succ[i]:
            old = state[j]
            merged = k if old is None else tuple(min(a, b) for a, b in zip(old, k))
            if merged != old:
                state[j] = merged
                work.append(j)
    return state, problems


def template_args(symbol):
    """(VW, NCH, OPT, EMB16, FAT) from the mangled name `...k_adagrad_runsILi4ELi1ELi0ELb0ELb1EE...`."""
    m = re.search(r"k_adagrad_runsILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E", symbol)
    return (int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "1", m.group(5) == "1") if m else None


def n_tab_dma(vw, nch, opt, emb16, fat):
    """(N_TAB, N_DMA) of the instance, as glove.hip computes them: the stores of a step and the loads of one request."""
    mom = 1 if opt != 0 else 0
    n_q = 1 if vw == 4 else vw
    img_r_chunks = (nch + 1) // 2 if emb16 else nch
    n_tab = nch * (2 + mom) + (0 if fat else 2 + mom)
    n_dma = (img_r_chunks + nch * (1 + mom)) * n_q + (0 if fat else 2 + mom)
    return n_tab, n_dma


def n_after(vw, nch, opt, emb16, fat):
    """N_AFTER = N_TAB + N_DMA: the immediate of the hand-counted wait."""
    return sum(n_tab_dma(vw, nch, opt, emb16, fat))
