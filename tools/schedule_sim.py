"""List-scheduling bound for W-mix on 1024 wave slots: 1872 jobs of 57 us (exit after one ADMM iteration) and 2224 of
95 us (100 iterations) -- the per-instance times of the fused kernel (device stamps).  Greedy assignment in random order,
hardest-first (LPT: needs knowledge the solver does not have before solving) and the perfect packing.  No GPU needed.
    python tools/schedule_sim.py"""
import heapq
import random


def makespan(jobs, slots=1024):
    h = [0.0] * slots
    heapq.heapify(h)
    for j in jobs:
        heapq.heappush(h, heapq.heappop(h) + j)
    return max(h)


if __name__ == "__main__":
    easy, hard = 57.0, 95.0
    jobs = [easy] * 1872 + [hard] * 2224
    rnd = []
    for s in range(20):
        random.seed(s)
        j = jobs[:]
        random.shuffle(j)
        rnd.append(makespan(j))
    print(f"perfect packing {sum(jobs) / 1024:.1f} us | greedy, random order {sum(rnd) / len(rnd):.1f} us (min {min(rnd):.1f}, max {max(rnd):.1f}) | "
          f"greedy, hardest first {makespan(sorted(jobs, reverse=True)):.1f} us | + ~25 us of launch / first-round load burst each")
