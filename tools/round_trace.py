#!/usr/bin/env python
"""Where do the repair rounds of the headline workload go?  (GPU box; LQRRT_TRACE=2 dumps every fused round's state)

    python tools/round_trace.py [attempts]        -> stdout: rounds per wave by cause, re-steers per sample, horizon moves

For every wave of the measurement window: the horizon (first goal hit among the current records) after every round, how many
samples re-steered more than once, how many re-steers were thrown away (the record was replaced again before the wave
converged, or the sample lay beyond the committed prefix), and the length of the longest chain of in-wave parents at the end
(the number of rounds the wave NEEDED) against the rounds it took."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(attempts):
    sys.path.insert(0, ROOT)
    sys.argv = ["bench"]
    import bench
    prob, eng = bench.build_problem("cfg4", 10000 + attempts, 1024, 0)
    eng.extend(1024, until_size=9500)
    sys.stderr.write("[window]\n")
    st = eng.extend(1024, max_attempts=attempts)
    sys.stderr.write("[end] %r\n" % (st.as_dict(),))


def main():
    attempts = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    if os.environ.get("ROUND_TRACE_CHILD"):
        return child(attempts)
    env = dict(os.environ, LQRRT_TRACE="2", ROUND_TRACE_CHILD="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), str(attempts)], env=env, capture_output=True, text=True)
    err = p.stderr
    if "[window]" not in err:
        print(err[-3000:]); sys.exit(1)
    err = err.split("[window]", 1)[1]
    waves = []          # each: dict(W, rounds=[state list], C, hit)
    cur = None
    for line in err.splitlines():
        m = re.match(r"\[roundstate N=(\d+) W=(\d+) r=(\d+)\](.*)", line)
        if m:
            N, W, r = int(m.group(1)), int(m.group(2)), int(m.group(3))
            st = [tuple(int(v) for v in tok.split(":")) for tok in m.group(4).split()]
            if r == 0:
                cur = dict(N=N, W=W, rounds=[]); waves.append(cur)
            cur["rounds"].append(st)
            continue
        m = re.match(r"\[wave N=(\d+) W=(\d+)\] commit C=(\d+) acc=(\d+) hit=(\d+) rounds=(\d+)", line)
        if m and cur is not None:
            cur["C"], cur["hit"] = int(m.group(3)), int(m.group(5))
    waves = [w for w in waves if "C" in w]
    tot_att = sum(w["C"] for w in waves)
    s = 1024.0 / max(1, tot_att)
    n_rounds = sum(len(w["rounds"]) - 1 for w in waves)      # the last dumped round is the converged one
    needed = 0; resteers = 0; wasted_replaced = 0; wasted_beyond = 0; hz_fwd = 0; hz_back = 0; multi = 0
    rounds_after_hz_move = 0; tail_rounds = 0; list_hist = {}
    for w in waves:
        W, C = w["W"], w["C"]
        def horizon(st):
            for t, (par, ln, hit, ch, sl) in enumerate(st):
                if ln > 0 and hit: return t
            return W - 1
        hz = [horizon(st) for st in w["rounds"]]
        for a, b in zip(hz, hz[1:]):
            if b > a: hz_fwd += 1
            if b < a: hz_back += 1
        # first round after which the horizon never changed again
        last_move = 0
        for i in range(1, len(hz)):
            if hz[i] != hz[i - 1]: last_move = i
        rounds_after_hz_move += max(0, len(w["rounds"]) - 1 - last_move)
        cnt = [0] * W
        for st in w["rounds"]:
            k = 0
            for t, (par, ln, hit, ch, sl) in enumerate(st):
                if ch: cnt[t] += 1; k += 1
            if k: list_hist[min(k, 16)] = list_hist.get(min(k, 16), 0) + 1
            if 0 < k <= 2: tail_rounds += 1
        resteers += sum(cnt)
        wasted_beyond += sum(cnt[C:])
        wasted_replaced += sum(max(0, c - 1) for c in cnt[:C])
        multi += sum(1 for c in cnt[:C] if c > 1)
        # depth of the final in-wave parent chains within the committed prefix
        fin = w["rounds"][-1]
        depth = [0] * W
        for t in range(C):
            par = fin[t][0]
            if par < 0: depth[t] = depth[~par] + 1
        needed += max(depth[:C]) if C else 0
    print("waves %d, attempts %d; per 1024 attempts: waves %.1f, repair rounds %.1f, needed by the final parent chains %.1f"
          % (len(waves), tot_att, len(waves) * s, n_rounds * s, needed * s))
    print("re-steers %.1f per 1024: thrown away because replaced again %.1f (%d samples re-steered more than once), beyond the committed prefix %.1f"
          % (resteers * s, wasted_replaced * s, multi, wasted_beyond * s))
    print("horizon moved forward %d times, backward %d times (per 1024: %.1f / %.1f); rounds run after its last move %.1f per 1024"
          % (hz_fwd, hz_back, hz_fwd * s, hz_back * s, rounds_after_hz_move * s))
    print("rounds with 1-2 re-steers: %.1f per 1024; rounds by number of re-steers (16 = 16 or more): %s"
          % (tail_rounds * s, " ".join("%d:%.1f" % (k, v * s) for k, v in sorted(list_hist.items()))))


if __name__ == "__main__":
    main()
