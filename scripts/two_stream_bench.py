"""Scratch: aggregate NN throughput of two nets (own weights, own stream) replaying their graphs concurrently."""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nn_cases
from crazyara_amd import rise_config as ro
from crazyara_amd.neuralnetapi import HipAPI
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = ro.rise_v2_config(19)
sd = ro.make_state_dict(cfg, seed=1)
d = nn_cases.export_case(tempfile.mkdtemp(), "b", cfg, sd)
nets = [HipAPI(0, B, d, "float16") for _ in range(2)]
for n in nets: n.time_forward(10)
N = 200
t0 = time.perf_counter(); ms1 = nets[0].time_forward(N) / N; t1 = time.perf_counter()
print(f"one stream : {ms1:.4f} ms/forward  {B/ms1*1e3:.0f} evals/s")
res = [0, 0]
def run(i): res[i] = nets[i].time_forward(N) / N
ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
wall = time.perf_counter() - t0
print(f"two streams: {res[0]:.4f} / {res[1]:.4f} ms/forward each, wall {wall*1e3:.1f} ms for {2*N} forwards -> {2*N*B/wall:.0f} evals/s")
for n in nets: n.close()
