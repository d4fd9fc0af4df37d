"""Per-simulation-step latency of the closed loop at the Diamond shape: separate
compute_RO_state + observer.update against the fused sekf_step_projected, through Python and through the bare C ABI.
Usage (GPU box): python tools/step_latency.py"""
import io
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'soft-robot-control_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    torch.cuda.init()
    import bench
    import workloads as wl
    from sofacontrol_amd import _lib
    from sofacontrol_amd.mor.pod import POD
    _lib.set_device(0)
    w = wl.diamond_c2()
    rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
    with contextlib.redirect_stdout(io.StringIO()):
        tp, gm = bench.build_model(w)
    print(json.dumps(bench.closed_loop_latency(w, rom, tp), indent=1))


if __name__ == '__main__':
    main()
