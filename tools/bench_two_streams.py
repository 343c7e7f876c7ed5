#!/usr/bin/env python3
"""Does the chip run the batched kernels of two half-batches side by side faster than one whole batch?  Two models (own
workspaces), two streams, 500 000 limb rays each, against one model with 1 000 000 -- the question behind overlapping
the latency-bound tracer / radiance update of one sub-chunk with the issue-bound look-up of another."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import bench
from jurassic_hip import lib

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
case = bench.build_case("limb_1e6", bench.workload_rays("limb_1e6", np.arange(N)))
nd = case.ctl.nd


def setup(lo, hi):
    m = lib.Model(case.ctl, case.lib_tables())
    m.set_atm(case.atm)
    n = hi - lo
    g = torch.from_numpy(np.ascontiguousarray(case.geom[lo:hi].T)).to(dev)
    rad = torch.zeros((n, nd), dtype=torch.float64, device=dev)
    tau, tp = torch.zeros_like(rad), torch.zeros((3, n), dtype=torch.float64, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    m.set_workspace_budget(60 << 30)
    m.reserve(n)
    return dict(m=m, n=n, g=g, rad=rad, tau=tau, tp=tp, st=st, s=torch.cuda.Stream())


def run(parts, steps=5, stagger=0):
    def once():
        for k, p in enumerate(parts):
            with torch.cuda.stream(p["s"]):
                if stagger and k:
                    torch.cuda._sleep(int(stagger * k))
                p["rad"].zero_()
                p["m"].formod_device(p["n"], p["g"].data_ptr(), p["rad"].data_ptr(), p["tau"].data_ptr(), p["tp"].data_ptr(), 0,
                                     p["st"].data_ptr(), p["s"].cuda_stream)
    once(); once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


whole = setup(0, N)
t_whole = run([whole])
halves = [setup(0, N // 2), setup(N // 2, N)]
t_serial = run([halves[0]]) + run([halves[1]])
t_both = run(halves)
# calibrate the spin kernel, then start the second half while the first is in its look-up kernel
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(100_000_000); torch.cuda.synchronize()
cycles_per_ms = 100_000_000 / ((time.perf_counter() - t0) * 1e3)
staggered = {str(ms): round(run(halves, stagger=ms * cycles_per_ms), 2) for ms in (4, 8, 15, 25)}
ok = torch.equal(torch.cat([halves[0]["rad"], halves[1]["rad"]]), whole["rad"])
print(json.dumps({"rays": N, "ms_one_batch": round(t_whole, 2), "ms_two_halves_one_after_the_other": round(t_serial, 2),
                  "ms_two_halves_on_two_streams": round(t_both, 2),
                  "ms_two_streams_second_started_x_ms_later": staggered, "same_bits": bool(ok)}))
