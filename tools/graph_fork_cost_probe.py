"""What does a fork / join cost inside a replayed hipGraph?  (the question behind profiles/ab/r05j and r05k: parallel branches that
lose although the branch's work is small.)

A chain of CHAIN dependent small kernels (an in-place add on 256 KiB: ~3 us of work, the dependent-launch floor) is captured
  (a) as it is,
  (b) with a fork every EVERY-th kernel: one kernel of the same kind on a second stream, joined before the next main kernel,
  (c) the same forks, all joined once at the end of the chain,
  (d) the side kernels appended to the main chain instead (the same number of launches, no fork),
and replayed; the difference per fork is printed.  python tools/graph_fork_cost_probe.py [CHAIN] [EVERY]"""
import sys
import time

import torch

CHAIN = int(sys.argv[1]) if len(sys.argv) > 1 else 200
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = 'cuda:0'
x = torch.zeros(65536, device=dev)
y = torch.zeros(65536, device=dev)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()


def chain(mode):
    cur = torch.cuda.current_stream()
    forks = 0
    for i in range(CHAIN):
        x.add_(1.0)
        if i % EVERY == EVERY - 1:
            forks += 1
            if mode == 'inline':
                y.add_(1.0)
            elif mode in ('join_next', 'join_end'):
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    y.add_(1.0)
                if mode == 'join_next':
                    cur.wait_stream(side)
    if mode == 'join_end':
        cur.wait_stream(side)
    return forks


def replay_us(mode, n=200):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        forks = chain(mode)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            chain(mode)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, forks


base, _ = replay_us('plain')
print(f'chain of {CHAIN} dependent small kernels: {base:8.1f} us per replay = {base / CHAIN:.2f} us per kernel')
for mode, what in (('inline', 'side kernels appended to the main chain (no fork)'),
                   ('join_next', 'fork, side kernel, join before the next main kernel'),
                   ('join_end', 'fork, side kernel, ONE join at the end of the chain')):
    t, forks = replay_us(mode)
    print(f'{what:58s}: {t:8.1f} us per replay, {forks} side kernels: {(t - base) / forks:+.2f} us each')
