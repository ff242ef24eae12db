"""Generates tests/golden/onnx/*.onnx: the REFERENCE PyTorch modules exported by torch's own ONNX exporter.

Run in the build container only (needs /root/reference):  python oracle/make_onnx_fixtures.py
Follows export_to_onnx (DeepCrazyhouse/src/training/trainer_agent_pytorch.py:588-633): input "data", outputs "value_out",
"policy_out" [, "auxiliary_out", "wdl_out", "plys_to_end_out"], dynamic batch axis or a "-bsize-<B>" file, "-v<maj>.<min>" in the name.
The reference then runs onnx-simplifier, which is absent here; the importer is written for both forms.

torch's TorchScript exporter serialises the graph in C++; its last step imports the `onnx` package only to splice onnxscript
functions in (none are used here), so that step is replaced by the identity.
"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import make_golden
    import onnx_cases
    from crazyara_amd.rise_config import make_state_dict
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, *a, **k: proto
    RiseV3 = make_golden.import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden", "onnx")
    os.makedirs(out_dir, exist_ok=True)
    only = sys.argv[1:]
    for name in onnx_cases.CASES:
        if only and name not in only:
            continue
        cfg, seed, fname, batch, stress = onnx_cases.unpack(name)
        model = make_golden.reference_model(RiseV3, cfg)
        sd = make_state_dict(cfg, seed=seed, stress=stress)
        if cfg.conv_block == "a0_res_block":
            sd = {("body." + k[len("body_spatial."):] if k.startswith("body_spatial.") else k): v for k, v in sd.items()}
        model.load_state_dict(sd, strict=True)
        outputs = ["value_out", "policy_out"] + (["auxiliary_out", "wdl_out", "plys_to_end_out"] if cfg.use_wdl else [])
        dyn = None if batch else {n: {0: "batch_size"} for n in ["data"] + outputs}
        x = torch.zeros(batch or 1, cfg.nb_input_channels, 8, 8)
        path = os.path.join(out_dir, fname)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(model, x, path, input_names=["data"], output_names=outputs, dynamic_axes=dyn, dynamo=False)
        print(f"{name}: {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
