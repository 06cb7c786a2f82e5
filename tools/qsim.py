#!/usr/bin/env python3
"""Development aid: discrete-event model of the step engine's queues (256 envs, run / play / other batches whose time does not depend on how many lanes hold an env).
Calibrated with the batch times of tools/timing4.py; reproduces the measured sensitivities to wave counts."""
import heapq, random, sys
def sim(T_run=6600, T_claim=1200, T_play=79000, T_other=30000, n_waves=7, n_serve=4, N=256, steps=400, p_play=0.08, p_other=0.06,
        th=64, seed=1, lane_cost=0.0, policy="longest"):
    rnd = random.Random(seed)
    q = {0: [], 1: [], 2: []}   # run, play, other
    for e in range(N): q[0].append(e)
    t_env = [0]*N
    done = 0
    now = 0.0
    # events: (time, wave, cls, items)
    ev = []
    idle = set(range(n_waves))
    busy = 0
    wave_busy_time = [0.0]*n_waves
    batches = {0:0,1:0,2:0}; items_tot = {0:0,1:0,2:0}
    def pick(w):
        can = w >= n_waves - n_serve
        nr, np_, no = len(q[0]), (len(q[1]) if can else 0), (len(q[2]) if can else 0)
        if np_ >= th: return 1
        if no >= th: return 2
        if nr >= th: return 0
        if nr or np_ or no:
            if policy == "service_first":
                if np_ and np_ >= no: return 1
                if no: return 2
                return 0 if nr else -1
            if nr and nr >= np_ and nr >= no: return 0
            if np_ >= 1 and np_ >= no: return 1
            if no >= 1: return 2
            if nr: return 0
        return -1
    def dispatch():
        nonlocal busy
        for w in sorted(idle):
            c = pick(w)
            if c < 0: continue
            items = q[c][:64]; del q[c][:64]
            T = (T_run, T_play, T_other)[c] + T_claim + lane_cost*len(items)
            heapq.heappush(ev, (now + T, w, c, items))
            idle.discard(w); busy += 1
            wave_busy_time[w] += T
            batches[c] += 1; items_tot[c] += len(items)
    dispatch()
    while done < N:
        now, w, c, items = heapq.heappop(ev)
        for e in items:
            if c == 0:
                r = rnd.random()
                if r < p_play: q[1].append(e); continue
                if r < p_play + p_other: q[2].append(e); continue
            t_env[e] += 1
            if t_env[e] >= steps: done += 1
            else: q[0].append(e)
        idle.add(w); busy -= 1
        dispatch()
    util = [b/now for b in wave_busy_time]
    return now/steps, batches, {k: items_tot[k]/max(1,batches[k]) for k in batches}, util
if __name__ == "__main__":
    base = sim()
    print("base cycles/step %.0f" % base[0], {k: round(v/400,2) for k,v in base[1].items()}, {k: round(v,1) for k,v in base[2].items()}, [round(u,2) for u in base[3]])
    for name, kw in [("claim 300", dict(T_claim=300)), ("run 5000", dict(T_run=5000)), ("run 4000 claim 300", dict(T_run=4000, T_claim=300)),
                     ("play 69000", dict(T_play=69000)), ("play 60000", dict(T_play=60000)), ("play 50000 other 22000", dict(T_play=50000, T_other=22000)),
                     ("other 22000", dict(T_other=22000)), ("8 waves", dict(n_waves=8)), ("4 waves", dict(n_waves=4)), ("3 serve", dict(n_serve=3)), ("2 serve", dict(n_serve=2)),
                     ("5 serve", dict(n_serve=5)), ("7 serve", dict(n_serve=7)), ("service first", dict(policy="service_first")), ("th 32", dict(th=32))]:
        r = sim(**kw)
        print("%-24s cycles/step %.0f (%+.1f%%)" % (name, r[0], (base[0]/r[0]-1)*100), {k: round(v/400,2) for k,v in r[1].items()}, {k: round(v,1) for k,v in r[2].items()}, [round(u,2) for u in r[3]])
