"""Back-of-envelope schedule simulator for 'continued chains' (DESIGN §9): T tiles x NC chunk units dealt to G resident
workers as contiguous ranges; a worker runs the HEAD of its last tile first, then whole tiles, then the TAIL of its first
tile (continuing from the accumulators the previous worker wrote).  Each CU holds G/256 workers that share one matrix pipe
(processor sharing); a job has a latency-only prologue / epilogue during which the pipe is free for the other worker.
usage: python tools/experiments/sk_sim.py [T NC]"""
import sys

CH = 2.19      # us per 8-channel chunk of a 128-pixel x 64-channel tile (72 MFMAs x 64 cycles x 4 ... at 2.1 GHz)
PRO, EPI, SPILL = 2.0, 1.0, 0.4


def jobs_for(i, G, T, NC):
    U = T * NC
    u0, u1 = i * U // G, (i + 1) * U // G
    if u1 <= u0:
        return []
    ta, ca = divmod(u0, NC)
    tb, cb = divmod(u1 - 1, NC)
    cb += 1
    if ta == tb:
        return [dict(tile=ta, c0=ca, c1=cb)]
    out = []
    head = dict(tile=tb, c0=0, c1=cb)
    tail = dict(tile=ta, c0=ca, c1=NC)
    if cb < NC:
        out.append(head)
    for t in range(ta + (1 if ca > 0 else 0), tb + (1 if cb == NC else 0)):
        out.append(dict(tile=t, c0=0, c1=NC))
    if ca > 0:
        out.append(tail)
    return out


def simulate(worker_jobs, per_cu, NC):
    """worker_jobs: list (per worker) of job dicts.  worker w sits on CU w % 256."""
    n = len(worker_jobs)
    done_at = {}                      # (tile, c1) -> time the partial / tile was finished
    state = [dict(j=0, phase="idle", left=0.0) for _ in range(n)]
    t = 0.0
    cus = {}
    for w in range(n):
        cus.setdefault(w % 256, []).append(w)
    finish = [0.0] * n
    active = set(w for w in range(n) if worker_jobs[w])
    dt = 0.05
    while active:
        for cu, ws in cus.items():
            comp = [w for w in ws if w in active and state[w]["phase"] == "compute"]
            for w in ws:
                if w not in active:
                    continue
                s = state[w]
                job = worker_jobs[w][s["j"]]
                if s["phase"] == "idle":
                    need = (job["tile"], job["c0"])
                    if job["c0"] == 0 or need in done_at and done_at[need] <= t:
                        s["phase"], s["left"] = "pro", PRO + (SPILL if job["c0"] else 0.0)
                elif s["phase"] == "pro":
                    s["left"] -= dt
                    if s["left"] <= 0:
                        s["phase"], s["left"] = "compute", (job["c1"] - job["c0"]) * CH
                elif s["phase"] == "compute":
                    s["left"] -= dt / len(comp)
                    if s["left"] <= 0:
                        s["phase"], s["left"] = "epi", (EPI if job["c1"] == NC else SPILL)
                elif s["phase"] == "epi":
                    s["left"] -= dt
                    if s["left"] <= 0:
                        done_at[(job["tile"], job["c1"])] = t
                        s["j"] += 1
                        s["phase"] = "idle"
                        if s["j"] == len(worker_jobs[w]):
                            active.discard(w)
                            finish[w] = t
        t += dt
        if t > 1e4:
            raise SystemExit("stuck")
    return max(finish)


def main():
    T, NC = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 8)
    now = simulate([[dict(tile=i, c0=0, c1=NC)] for i in range(T)], None, NC)
    print("T=%d NC=%d  ideal %.1f us   one block per tile: %.1f us" % (T, NC, T * NC * CH / 256, now))
    for G in (256, 512, 768):
        wj = [jobs_for(i, G, T, NC) for i in range(G)]
        assert sum(j["c1"] - j["c0"] for w in wj for j in w) == T * NC
        print("  G=%d continued chains: %.1f us" % (G, simulate(wj, None, NC)))
    # pairs: CU c's range of T*NC/256 units; worker c takes its partial tiles (head first, tail last), worker c+256 its whole tiles
    base = [jobs_for(i, 256, T, NC) for i in range(256)]
    part = [[j for j in w if not (j["c0"] == 0 and j["c1"] == NC)] for w in base]
    whole = [[j for j in w if j["c0"] == 0 and j["c1"] == NC] for w in base]
    print("  256 ranges, partial tiles on worker c, whole tiles on worker c+256: %.1f us" % simulate(part + whole, None, NC))
    # the same, the tail handed to the whole-tile worker (it runs last there)
    part2 = [[j for j in w if j["c0"] == 0] for w in part]
    whole2 = [wh + [j for j in pa if j["c0"] > 0] for wh, pa in zip(whole, part)]
    print("  ... head on worker c, whole tiles then tail on worker c+256:        %.1f us" % simulate(part2 + whole2, None, NC))


main()
