import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crazyara_amd import rise_config
from oracle import rise_oracle as ro
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch default threads", torch.get_num_threads())
cfg = rise_config.rise_v2_config(19); sd = rise_config.make_state_dict(cfg, 1)
x = (torch.rand(64, 34, 8, 8) < 0.1).float()
for t in (8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    ro.predict(cfg, sd, x[:8])
    t0 = time.perf_counter(); ro.predict(cfg, sd, x); el = time.perf_counter() - t0
    print(f"threads {t}: {64/el:.1f} evals/s")
