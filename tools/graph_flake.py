"""Diagnostic: eager vs HIP-graph replay of six cfg1 train steps, repeated; prints the largest relative difference per
repetition (tests/test_step_gpu.py::test_hip_graph_replay_equals_eager bounds it by 1e-5)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from test_step_gpu import make, rel
from downgan_amd import synthetic

for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    res = {}
    for graphs in (False, True):
        eng, *_, xc, xf = make(4, 16, 16, 2, 16, "f32")
        if graphs:
            eng.enable_graphs(xc, xf)
        out = []
        for step in range(6):
            alpha = torch.from_numpy(synthetic.alpha(4, step)).cuda()
            ran_g = eng.train_step(xc, xf, alpha)
            out.append(eng.read_scalars(ran_g))
        res[graphs] = out
    worst = [max((rel(a[k], b[k]), k) for k in a) for a, b in zip(res[False], res[True])]
    print(rep, " ".join(f"{w:.1e}:{k}" for w, k in worst), flush=True)
