#!/usr/bin/env python3
"""Child of tools/bench_support.measure_traffic, run under `rocprofv3 --kernel-trace --pmc <counter>`: launches the
projection kernel of the fused pipeline's first pass (codes + row statistics, csrc/project.hip) for every configuration
in argv[1] (JSON list of {"tag", "grid", "frames", "u8"[, "mode": "max" | "derive_slice" | "slice"]}), `reps` times each, and prints the launch order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfgs = json.loads(sys.argv[1])
    import torch
    import radar_ml_amd as rml
    from radar_ml_amd import _lib
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    ctx = _lib.context(dev)
    reps = 3
    st = torch.cuda.current_stream(dev).cuda_stream
    for c in cfgs:
        X, Y, Z = c["grid"]
        B = int(c["frames"])
        D = rml.feature_len(X, Y, Z)
        # zeros are enough: the counters see bytes, not values (and k_synth* must not be mistaken for a projection)
        V = torch.zeros((B, X, Y, Z), dtype=torch.uint8 if c["u8"] else torch.float32, device=dev)
        ldq = ((D + 127) // 128 * 128)
        ldq = ldq if (ldq // 128) % 2 else ldq + 128          # the model's code-row stride (rml_svm::Dq)
        q = torch.empty((B, ldq), dtype=torch.uint8, device=dev)
        isum = torch.empty(B, dtype=torch.int32, device=dev)
        isq = torch.empty(B, dtype=torch.int64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        mode = c.get("mode", "max")
        ijk = torch.tensor([[X // 2, Y // 2, Z // 2]], dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
        torch.cuda.synchronize()
        for _ in range(reps):
            if mode == "derive_slice":          # k_derive_slice: derive + slices, codes + statistics (the pipeline's first pass)
                _lib.check(lib.rml_derive_slice(ctx, V.data_ptr(), 1 if c["u8"] else 0, B, X, Y, Z, 1, ijk.data_ptr(), None, 255.0, 7, None, 0,
                                                q.data_ptr(), ldq, isum.data_ptr(), isq.data_ptr(), flags.data_ptr(), st), "rml_derive_slice")
            elif mode == "slice":               # k_slice_rows at a given voxel
                _lib.check(lib.rml_project(ctx, V.data_ptr(), 1 if c["u8"] else 0, B, X, Y, Z, 1, ijk.data_ptr(), 255.0, 7, None, 0, q.data_ptr(),
                                           ldq, isum.data_ptr(), isq.data_ptr(), flags.data_ptr(), st), "rml_project")
            else:
                _lib.check(lib.rml_project(ctx, V.data_ptr(), 1 if c["u8"] else 0, B, X, Y, Z, 0, None, 255.0, 7, None, 0, q.data_ptr(), ldq,
                                           isum.data_ptr(), isq.data_ptr(), None if c["u8"] else flags.data_ptr(), st), "rml_project")
        torch.cuda.synchronize()
        del V, q
        torch.cuda.empty_cache()
    print("PMC_CHILD_ORDER " + json.dumps({"tags": [c["tag"] for c in cfgs], "reps": reps}))


if __name__ == "__main__":
    main()
