"""hipGraph-timed launches of the linear-Gaussian propagation kernels (K8 / K9 / K10) at the BASELINE
shapes on the MI355X (parity lives in tests/test_gpu_linear_gaussian.py).

    python tools/lgbench.py [--shapes c4,c2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesmc_amd  # noqa: E402
from aesmc_amd import _kernels  # noqa: E402

SHAPES = {"c4": (1024, 4096, 10, 10), "c2": (256, 1024, 10, 10), "c4s": (128, 4096, 10, 10),
          "d16": (1024, 4096, 16, 16), "d4": (1024, 4096, 4, 3), "d2": (1024, 4096, 2, 2), "d6": (1024, 4096, 6, 6),
          "d8": (1024, 4096, 8, 8), "d12": (1024, 4096, 12, 12), "d1": (4096, 8192, 1, 1),
          "k64": (16384, 64, 10, 10), "k16": (65536, 16, 10, 10)}


def operands(B, K, dx, dy, dtype, device, seed=0):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *shape: torch.randn(*shape, generator=gen, dtype=torch.float64).to(device=device, dtype=dtype)
    eye = torch.eye(dx, dtype=torch.float64)
    A = (0.9 * eye + 0.01 * torch.randn(dx, dx, generator=gen, dtype=torch.float64)).to(device=device, dtype=dtype)
    Q = (0.45 * eye + 0.01 * torch.randn(dx, dx, generator=gen, dtype=torch.float64)).to(device=device, dtype=dtype)
    C = (torch.randn(dy, dx, generator=gen, dtype=torch.float64) * 0.3).to(device=device, dtype=dtype)
    return {"x_prev": r(B, K, dx), "x": r(B, K, dx), "eps": r(B, K, dx), "y": r(B, dy), "A": A, "Q": Q, "C": C,
            "off_q": r(B, dx), "off_g": r(dy),
            "s_p": torch.tensor(1.0, dtype=dtype, device=device), "s_g": torch.tensor(0.5, dtype=dtype, device=device),
            "s_q": torch.tensor(0.7, dtype=dtype, device=device)}


def graph_time(fn, repeats=10):
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(repeats):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    begin.record()
    for _ in range(3):
        graph.replay()
    end.record()
    torch.cuda.synchronize()
    return begin.elapsed_time(end) * 1e3 / (3 * repeats)   # us


def bench(name, dtype=torch.float32):
    B, K, dx, dy = SHAPES[name]
    k = _kernels.get()
    device = torch.device("cuda:0")
    sets = [operands(B, K, dx, dy, dtype, device, seed=s) for s in range(4 if B * K > 1 << 21 else 12)]
    esz = 4 if dtype == torch.float32 else 8
    out = {}
    state = {"i": 0}

    def nxt():
        state["i"] = (state["i"] + 1) % len(sets)
        return sets[state["i"]]

    def run(label, fn, nbytes):
        us = graph_time(fn)
        out[label] = {"us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 8e6, 3)}

    N = B * K
    run("K8 particle_affine", lambda: (lambda o: k.particle_affine(o["x_prev"], o["Q"], o["off_q"]))(nxt()),
        esz * N * (dx + dx))
    run("torch matmul + add", lambda: (lambda o: o["x_prev"] @ o["Q"].t() + o["off_q"].unsqueeze(1))(nxt()),
        esz * N * (dx + dx))
    run("K9 affine_rsample", lambda: (lambda o: k.affine_rsample(o["x_prev"], o["Q"], o["off_q"], o["eps"], o["s_q"]))(nxt()),
        esz * N * 3 * dx)
    run("K10 affine_logweight",
        lambda: (lambda o: k.affine_logweight(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]),
                                              (o["Q"], o["off_q"]), (o["s_p"], o["s_g"], o["s_q"])))(nxt()),
        esz * N * (2 * dx + 1))
    outs = [torch.empty_like(o["x"]) for o in sets]

    def k15():
        i = (state["i"] + 1) % len(sets)
        state["i"] = i
        o = sets[i]
        return k.affine_propagate(o["x_prev"], o["eps"], o["y"], (o["A"], None), (o["C"], o["off_g"]),
                                  (o["Q"], o["off_q"]), (o["s_p"], o["s_g"], o["s_q"]), out_x=outs[i])
    run("K15 affine_propagate", k15, esz * N * (3 * dx + 1))
    run("K11 particle_affine_backward", lambda: (lambda o: k.particle_affine_backward(o["eps"], o["x_prev"], o["Q"], True, True, True))(nxt()),
        esz * N * 3 * dx)
    lws = [k.affine_logweight(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]), (o["Q"], o["off_q"]),
                              (o["s_p"], o["s_g"], o["s_q"])) for o in sets]
    lses = [k.logweight_lse(lw, None, None, want_lw=False)[1] for lw in lws]
    need = [True, True, False, True, False, True, False, True, True, False, False, False]

    def k12():
        i = (state["i"] + 1) % len(sets)
        state["i"] = i
        o = sets[i]
        return k.affine_logweight_backward(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]),
                                           (o["Q"], o["off_q"]), (o["s_p"], o["s_g"], o["s_q"]), need, lw=lws[i],
                                           lse=lses[i], grad_lse=torch.ones_like(lses[i]))
    run("K12 affine_logweight_backward", k12, esz * N * (4 * dx + 1))
    need14 = list(need)
    need14[1] = False

    def k14():
        i = (state["i"] + 1) % len(sets)
        state["i"] = i
        o = sets[i]
        return k.affine_step_backward(o["x_prev"], o["x"], o["y"], (o["A"], None), (o["C"], o["off_g"]),
                                      (o["Q"], o["off_q"]), (o["s_p"], o["s_g"], o["s_q"]), need14, lws[i], lses[i],
                                      grad_lse=torch.ones_like(lses[i]), grad_x=o["eps"])
    run("K14 affine_step_backward", k14, esz * N * (4 * dx + 1))
    if os.environ.get("LGBENCH_VARIANTS"):
        full = list(need14)
        for slots, label in (((8,), "K14 without offset_q"), ((8, 9, 10, 11), "K14 without offset_q, scales")):
            need14[:] = full
            for slot in slots:
                need14[slot] = False
            run(label, k14, esz * N * (4 * dx + 1))
        need14[:] = full
    return out


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--shapes", default="c4,c2,c4s")
    args = parser.parse_args()
    for name in args.shapes.split(","):
        print(name, json.dumps(bench(name)), flush=True)


if __name__ == "__main__":
    main()
