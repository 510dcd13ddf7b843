#!/usr/bin/env python3
"""GPU box: how does hipGraph replay schedule a two-stream capture?  Synthetic fork/join patterns of spin kernels
(torch.cuda._sleep), timed with events: a replay that takes max(main, side) overlaps perfectly, main + side is serial.

    python tools/graph_sched_probe.py
"""
import sys
import torch

dev = torch.device("cuda:0")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
CYC = int(sys.argv[1]) if len(sys.argv) > 1 else 40000


def spin(n=1):
    torch.cuda._sleep(CYC * n)


def fork():
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)


def join():
    torch.cuda.current_stream().wait_stream(side)


def time_graph(build, reps=20):
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build(False)           # warm-up eager
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build(True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def only_main(n):
    def b(_):
        for _ in range(n):
            spin()
    return b


def pat_long_side(n_main, n_side):
    """side chain forked at the root, joined at the end (query GRU beside the STN head)"""
    def b(_):
        spin()
        fork()
        with torch.cuda.stream(side):
            for _ in range(n_side):
                spin()
        for _ in range(n_main):
            spin()
        join()
        spin()
    return b


def pat_alternating(n, side_first, delay=0, side_per=1, main_per=2):
    """n ops; op i: `main_per` main kernels, then `side_per` side kernels that depend on the op's FIRST main kernel.
    side_first: the side kernels are issued right inside the op (capture order: side before the next op's main kernels);
    else they are issued `delay`+1 ops later."""
    def b(_):
        pending = []
        spin()
        for i in range(n):
            ev = torch.cuda.Event()
            spin()
            ev.record(torch.cuda.current_stream())
            for _ in range(main_per - 1):
                spin()

            def launch(ev=ev):
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    for _ in range(side_per):
                        spin()
            if side_first:
                launch()
            else:
                pending.append(launch)
                while len(pending) > delay + 1:
                    pending.pop(0)()
        for l in pending:
            l()
        join()
        spin()
    return b


def pat_real(side_first, n_root=6, n_side=50, n_main=70, n_tail=130):
    """root chain -> fork -> [side chain | main chain] -> join -> long tail on the main stream (the training step's forward)"""
    def b(_):
        for _ in range(n_root):
            spin()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())

        def launch():
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for _ in range(n_side):
                    spin()
        if side_first:
            launch()
        for _ in range(n_main):
            spin()
        if not side_first:
            launch()
        join()
        for _ in range(n_tail):
            spin()
    return b


def pat_staged(n_stages, n_main, n_side, lag=1):
    """n_stages times: main lane of stage k (n_main kernels) beside the side lane of stage k - lag (n_side kernels), forked at the
    start of the stage and joined at its end -- short parallel branches instead of one long side chain."""
    def b(_):
        spin()
        for k in range(n_stages + lag):
            if k >= lag:
                fork()
                with torch.cuda.stream(side):
                    for _ in range(n_side):
                        spin()
            if k < n_stages:
                for _ in range(n_main):
                    spin()
            if k >= lag:
                join()
        spin()
    return b


unit = time_graph(only_main(50)) / 50
print("spin kernel in a single-stream graph: %.2f us each (incl. boundary)" % unit)
for name, build, n_main, n_side in [
        # (n_main, n_side) given as (ideal units, -(serial units)) for the staged patterns: the last side lane has nothing beside it
        ("8 stages: main 45 | side 30 of the previous stage, fork/join per stage", pat_staged(8, 45, 30), 8 * 45 + 30 + 2, -(8 * 75 + 2)),
        ("3 stages: main 150 | side 80 of the previous stage", pat_staged(3, 150, 80), 3 * 150 + 80 + 2, -(3 * 230 + 2)),
        ("root 6 -> side 50 | main 70 -> join -> 130, side captured first", pat_real(True), 206, 50),
        ("root 6 -> side 50 | main 70 -> join -> 130, main captured first", pat_real(False), 206, 50),
        ("long side chain 30 beside main 30", pat_long_side(30, 30), 32, 30),
        ("long side chain 60 beside main 30", pat_long_side(30, 60), 32, 60),
        ("alternating x20, side issued first", pat_alternating(20, True), 42, 20),
        ("alternating x20, main first (side one op late)", pat_alternating(20, False, 0), 42, 20),
        ("alternating x20, main first (side two ops late)", pat_alternating(20, False, 1), 42, 20),
        ("alternating x20, side 3 per op, side first", pat_alternating(20, True, 0, 3, 3), 62, 60),
        ("alternating x20, side 3 per op, main first", pat_alternating(20, False, 0, 3, 3), 62, 60),
        ("alternating x20, side 3 per op, all side at the end", pat_alternating(20, False, 100, 3, 3), 62, 60)]:
    t = time_graph(build)
    ideal, serial = (n_main, -n_side) if n_side < 0 else (max(n_main, n_side), n_main + n_side)
    print("%-55s %8.1f us   ideal %7.1f   serial %7.1f   (in units: %.1f)" % (name, t, ideal * unit, serial * unit, t / unit))


# ---- two single-stream graphs on two streams: does the LAUNCH order matter? -------------------------------------------------
def chain_graph(n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(n):
            spin()
    return g


cap = torch.cuda.Stream()
A, C = chain_graph(100, cap), chain_graph(60, cap)
Bg = chain_graph(60, side)
for order in ("A, B(side, after A), C", "A, C, B(side, after A)"):
    def run():
        A.replay()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        if order.startswith("A, B"):
            with torch.cuda.stream(side):
                side.wait_event(ev)
                Bg.replay()
            C.replay()
        else:
            C.replay()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                Bg.replay()
        torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    host = (time.perf_counter() - t0) / 10 * 1e6
    e1.record()
    torch.cuda.synchronize()
    print("graphs %-28s %8.1f us per round (host issue %7.1f us)   ideal %7.1f   serial %7.1f" % (
        order, e0.elapsed_time(e1) / 10 * 1e3, host, 160 * unit, 220 * unit))


# ---- a graph on the main stream beside EAGER launches on the side stream ----------------------------------------------------
def run_mixed():
    A.replay()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    C.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for _ in range(60):
            spin()
    torch.cuda.current_stream().wait_stream(side)


for _ in range(3):
    run_mixed()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run_mixed()
e1.record()
torch.cuda.synchronize()
print("graph A, graph C on main; 60 EAGER kernels on side after A: %8.1f us per round   ideal %7.1f   serial %7.1f" % (
    e0.elapsed_time(e1) / 10 * 1e3, 160 * unit, 220 * unit))
