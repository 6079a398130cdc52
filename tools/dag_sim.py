"""Design aid for the persistent `update` kernel (csrc/tgp_kernels_dag.hip): builds the static task list of the
128-tile Cholesky + inverse DAG, checks it (every product exactly once, dependencies earlier in the list) and runs a
discrete-event simulation of the dispatch rule (chain workgroup + bulk workgroups popping the list in order and
spinning on flags) under a simple cost model, to compare orderings and burst lengths before any GPU time is spent.

usage: python tools/dag_sim.py [NB=32] [burst=4] [workers=255]
"""
import heapq
import sys
from collections import defaultdict

US_PRODUCT = 16.0   # one 128^3 product on one CU (13.7 us at the MFMA peak)
US_TASK = 6.0       # per bulk task: dequeue, flags, C tile in/out, pipeline fill
US_TRI = 9.0        # triangular 128^3 product (T / E tasks)
US_LEAF, US_SYRK, US_TRMM, US_CHAIN_MISC = 26.0, 8.0, 8.0, 4.0


def build(NB, burst):
    """-> list of tasks in priority order.  task = dict(kind, out, deps=[flag...], sets=flag, cost)"""
    tasks = []
    # flags: ("L", i, k) tile L(i,k) final; ("Wd", j); ("P", i, j, k1) partial of tile (i,j) complete through column k1-1
    # ("W", i, c) final inverse tile; ("V", i, c, k1)
    def bursts(lo, hi):
        """partition [lo, hi) into bursts: long ones first, the LAST ones short (they are the urgent ones)"""
        out = []
        k = lo
        while k < hi:
            left = hi - k
            if left <= 2:
                n = 1
            elif left <= burst + 1:
                n = left - 2 if left - 2 >= 1 else 1
            else:
                n = burst
            out.append((k, k + n))
            k += n
        return out

    for j in range(NB):
        for i in range(j, NB):
            # columns the bulk contributes to tile (i, j): diagonal tile: k <= j-2 (the chain adds k = j-1 itself)
            hi = j - 1 if i == j else j
            prev = None
            for (k0, k1) in bursts(0, max(hi, 0)):
                deps = [("L", i, k1 - 1), ("L", j, k1 - 1)]
                if prev is not None:
                    deps.append(prev)
                fl = ("P", i, j, k1)
                tasks.append(dict(kind="G", out=(i, j), k=(k0, k1), deps=deps, sets=fl,
                                  cost=US_TASK + US_PRODUCT * (k1 - k0) * (0.56 if i == j else 1.0), ready_step=k1 - 1,
                                  need_step=j if i == j else (j - 0.5 if i == j + 1 else j)))
                prev = fl
            if i >= j + 2:  # L(i,j) = P(i,j) W_jj^T
                deps = [("Wd", j)] + ([prev] if prev else [])
                tasks.append(dict(kind="T", out=(i, j), deps=deps, sets=("L", i, j), cost=US_TASK + US_TRI,
                                  ready_step=j, need_step=j + 1))
    # inverse: V(i,c) = sum_{k=c}^{i-1} L(i,k) W(k,c)   (W(c,c) = Wd(c)),  W(i,c) = -W_ii V(i,c)
    for i in range(1, NB):
        for c in range(i):
            prev = None
            for (k0, k1) in bursts(c, i):
                deps = [("L", i, k1 - 1)]
                deps.append(("Wd", c) if k1 - 1 == c else ("W", k1 - 1, c))
                if k0 == c and k1 - 1 > c:
                    deps.append(("Wd", c))
                if prev is not None:
                    deps.append(prev)
                fl = ("V", i, c, k1)
                tasks.append(dict(kind="X", out=(i, c), k=(k0, k1), deps=deps, sets=fl,
                                  cost=US_TASK + US_PRODUCT * (k1 - k0), ready_step=k1 - 1 + 0.5, need_step=i))
                prev = fl
            tasks.append(dict(kind="E", out=(i, c), deps=[("Wd", i), prev], sets=("W", i, c), cost=US_TASK + US_TRI,
                              ready_step=i, need_step=i + 1))
    return tasks


def add_chain(tasks, NB):
    last = {}
    for t in tasks:
        if t["kind"] == "G":
            last[t["out"]] = t["sets"]
    chain = []
    for j in range(NB):
        deps = [last[(j, j)]] if (j, j) in last else []
        if j > 0:
            deps.append(("L", j, j - 1))
        chain.append(dict(kind="CA", out=(j, j), deps=deps, sets=("Wd", j), also=[("L", j, j)],
                          cost=(US_SYRK if j else 0.0) + US_LEAF + US_CHAIN_MISC / 2, ready_step=j - 0.4, need_step=-1))
        if j + 1 < NB:
            deps = [("Wd", j)] + ([last[(j + 1, j)]] if (j + 1, j) in last else [])
            chain.append(dict(kind="CB", out=(j + 1, j), deps=deps, sets=("L", j + 1, j), cost=US_TRMM + US_CHAIN_MISC / 2,
                              ready_step=j, need_step=-1))
    return tasks + chain


def topo_order(tasks, policy):
    """Kahn's algorithm; among the tasks whose producers are all placed, the smallest key goes first."""
    if policy == "need":
        key = lambda t: (t["need_step"], t["ready_step"])
    elif policy == "ready":
        key = lambda t: (t["ready_step"], t["need_step"])
    else:
        key = lambda t: (0.5 * (t["ready_step"] + t["need_step"]), t["need_step"])
    producer = {}
    for n, t in enumerate(tasks):
        producer[t["sets"]] = n
        for f in t.get("also", []):
            producer[f] = n
    indeg = [0] * len(tasks)
    users = defaultdict(list)
    for n, t in enumerate(tasks):
        for d in t["deps"]:
            if d not in producer:
                raise RuntimeError(f"no producer for {d} (needed by {t['kind']} {t['out']})")
            indeg[n] += 1
            users[producer[d]].append(n)
    heap = [(key(t), n) for n, t in enumerate(tasks) if indeg[n] == 0]
    heapq.heapify(heap)
    out = []
    while heap:
        _, n = heapq.heappop(heap)
        out.append(tasks[n])
        for u in users[n]:
            indeg[u] -= 1
            if indeg[u] == 0:
                heapq.heappush(heap, (key(tasks[u]), u))
    assert len(out) == len(tasks), "cycle"
    return out


def simulate(NB, burst, workers, policy, verbose=False):
    tasks = topo_order(add_chain(build(NB, burst), NB), policy)
    done = {}
    free = [(0.0, w) for w in range(workers)]
    heapq.heapify(free)
    chain_free = 0.0
    busy = stall = chain_wait = 0.0
    for t in tasks:
        ready = max([done[d] for d in t["deps"]] + [0.0])
        if t["kind"] in ("CA", "CB"):
            start = max(chain_free, ready)
            chain_wait += max(0.0, ready - chain_free)
            end = start + t["cost"]
            chain_free = end
        else:
            wfree, w = heapq.heappop(free)
            start = max(wfree, ready)
            stall += max(0.0, ready - wfree)
            end = start + t["cost"]
            busy += t["cost"]
            heapq.heappush(free, (end, w))
        done[t["sets"]] = end
        for f in t.get("also", []):
            done[f] = end
    makespan = max(done.values())
    nb = sum(1 for t in tasks if t["kind"] not in ("CA", "CB"))
    return dict(tasks=nb, makespan=makespan, chain_end=chain_free, chain_wait=chain_wait, bulk_busy=busy, stall=stall,
                util=busy / (workers * makespan))


def check_coverage(NB, burst):
    tasks = build(NB, burst)
    seen = defaultdict(int)
    for t in tasks:
        if t["kind"] == "G":
            for k in range(*t["k"]):
                seen[("G",) + t["out"] + (k,)] += 1
        if t["kind"] == "X":
            for k in range(*t["k"]):
                seen[("X",) + t["out"] + (k,)] += 1
    for j in range(NB):
        for i in range(j, NB):
            hi = j - 1 if i == j else j
            for k in range(max(hi, 0)):
                assert seen[("G", i, j, k)] == 1, (i, j, k)
    for i in range(1, NB):
        for c in range(i):
            for k in range(c, i):
                assert seen[("X", i, c, k)] == 1, (i, c, k)
    return len(tasks)


if __name__ == "__main__":
    NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    burst = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 255
    print("tasks", check_coverage(NB, burst))
    for policy in ("need", "ready", "mid"):
        try:
            r = simulate(NB, burst, workers, policy)
            print(policy, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
        except RuntimeError as e:
            print(policy, "INVALID:", e)
