"""Library baseline on the same GPU: the headline network through stock PyTorch-ROCm (MIOpen / rocBLAS kernels), fp16.

The reference's GPU back end is TensorRT, which does not exist on this hardware; the path a user gets "for free" on an MI355X is the
reference's own PyTorch module on PyTorch-ROCm.  This script times that (BN folded into the convolutions, channels_last or NCHW, half
precision, eager and replayed from one captured HIP graph) on the workload of bench.py, so the hand-written kernels are compared
with something measured on the same box.  Development tool: not part of bench.py, the tests or the product path.

usage: python scripts/torch_rocm_baseline.py [blocks=19] [batch=256]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

from crazyara_amd import rise_config

EPS = 1e-5


def fold(sd, conv, bn):
    w = sd[conv + ".weight"].double()
    g, b, m, v = (sd[f"{bn}.{s}"].double() for s in ("weight", "bias", "running_mean", "running_var"))
    s = g / torch.sqrt(v + EPS)
    return (w * s.view(-1, 1, 1, 1)).float(), (b - m * s).float()


class Net:
    """RiseV3 forward (rise_mobile_v3.py:143-184, builder_util.py) on folded weights; mobile bottleneck blocks, ca_se gates, policy map."""

    def __init__(self, cfg, sd, dtype, memory_format):
        dev = "cuda"
        self.cfg = cfg
        self.fmt = memory_format

        def put(t):
            t = t.to(dev, dtype)
            return t.contiguous(memory_format=memory_format) if t.dim() == 4 else t

        def cb(conv, bn):
            w, b = fold(sd, conv, bn)
            return put(w), put(b)

        self.stem = cb("body_spatial.0.body.0", "body_spatial.0.body.1")
        self.blocks = []
        for i, (k, se) in enumerate(zip(cfg.kernels, cfg.se_types)):
            p = f"body_spatial.{i + 1}"
            gate = (put(sd[p + ".se.fc.0.weight"]), put(sd[p + ".se.fc.2.weight"])) if se == "ca_se" else None
            self.blocks.append((gate, cb(p + ".body.0", p + ".body.1"), cb(p + ".body.3", p + ".body.4"), cb(p + ".body.6", p + ".body.7"), k))
        self.p0 = cb("policy_head.body.0", "policy_head.body.1")
        self.p1 = put(sd["policy_head.body.3.weight"])
        self.v0 = cb("value_head.body.0", "value_head.body.1")
        self.fc = [put(sd[f"value_head.body_final.{i}.{s}"]) for i in (0, 2) for s in ("weight", "bias")]

    def __call__(self, x):
        x = F.relu(F.conv2d(x, *self.stem, padding=1))
        for gate, e, d, p, k in self.blocks:
            if gate is not None:
                y = x.mean((2, 3))
                y = F.hardsigmoid(F.linear(F.relu(F.linear(y, gate[0])), gate[1]))
                x = x * y[:, :, None, None]
            y = F.relu(F.conv2d(x, *e))
            y = F.relu(F.conv2d(y, *d, padding=k // 2, groups=y.shape[1]))
            x = x + F.conv2d(y, *p)
        pol = F.conv2d(F.relu(F.conv2d(x, *self.p0, padding=1)), self.p1, padding=1).flatten(1)
        v = F.relu(F.conv2d(x, *self.v0)).flatten(1)
        v = torch.tanh(F.linear(F.relu(F.linear(v, self.fc[0], self.fc[1])), self.fc[2], self.fc[3]))
        return v, torch.softmax(pol.float(), 1)


def timed(fn, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = rise_config.rise_v2_config(blocks, 34, 81)
    sd = rise_config.make_state_dict(cfg, seed=2024, stress=True)
    out = {"workload": f"crazyhouse RISEv2-{blocks}, batch {batch}, fp16, BN folded", "torch": torch.__version__, "runs": []}
    torch.backends.cudnn.benchmark = True                                     # MIOpen find mode
    for name, fmt in (("nchw", torch.contiguous_format), ("channels_last", torch.channels_last)):
        net = Net(cfg, sd, torch.float16, fmt)
        x = torch.randn(batch, 34, 8, 8, device="cuda", dtype=torch.float16).contiguous(memory_format=fmt)
        with torch.no_grad():
            for _ in range(5):
                net(x)
            eager = timed(lambda: net(x), 30)
            graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                net(x)
                with torch.cuda.graph(graph, stream=s):
                    net(x)
            torch.cuda.current_stream().wait_stream(s)
            replay = timed(graph.replay, 50)
        out["runs"].append({"layout": name, "eager_ms": round(eager * 1e3, 3), "eager_evals_per_sec": round(batch / eager),
                            "graph_ms": round(replay * 1e3, 3), "graph_evals_per_sec": round(batch / replay)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
