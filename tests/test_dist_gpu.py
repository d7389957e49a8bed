"""The N > 1 control flow of bench.py on a real GPU: two ranks on ONE device (RML_BENCH_ONE_DEVICE=1, gloo staging the
collectives through the host -- RCCL refuses two ranks per device), against the single-process run of the same global batch.
Frames shard by global frame index, so the gathered labels of the 2-rank job must be the single-process labels; the SGAN
replicas must stay identical through the flat-bucket all-reduce with HIP-graph replay.  This is the multi-GPU evidence a
1-GPU box can give; the RCCL run itself is the driver's SCALE bench."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    """bench.py prints {"doc": verbose rows} on one line and, LAST, the contract line: return the contract object with the
    verbose rows attached under "doc".  The contract line must stand alone under 4 KB (what the driver parses)."""
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) >= 2, "expected a doc line and a contract line in:\n" + out[-2000:]
    last = lines[-1]
    assert len(last) < 4096, len(last)
    line = json.loads(last)
    assert "doc" not in line and list(line.keys())[-1] == "summary"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    doc = json.loads(lines[-2])
    assert list(doc.keys()) == ["doc"]
    line["doc"] = doc["doc"]
    return line


def test_two_rank_bench_on_one_device_matches_the_single_process_run():
    per_rank = 1024
    common = ["--steps", "2", "--warmup", "1", "--train", "1500", "--no-cpu", "--no-pmc", "--no-u8", "--parity", "256",
              "--general-frames", "512", "--dnn-frames", "512", "--dnn-parity", "64"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RML_BENCH_ONE_DEVICE", None)
    # bench.py lets MIOpen time its convolution solvers (tune_convolutions); the picks land in MIOpen's user find-db, which later
    # processes on the box read: keep this test's picks to itself, so that the numerics tests of the suite see MIOpen's defaults
    import tempfile
    env["MIOPEN_USER_DB_PATH"] = tempfile.mkdtemp(prefix="rml_miopen_db_")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", str(2 * per_rank),
                          "--walabot-frames", str(2 * per_rank)] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=1500)
    assert one.returncode == 0, one.stderr.decode()[-3000:]
    l1 = _line(one.stdout.decode())
    env2 = dict(env, RML_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env2.pop(k, None)
    # no launcher: `python bench.py --gpus 2` starts its own two ranks (torch.distributed.run on 127.0.0.1)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", str(per_rank),
                          "--walabot-frames", str(per_rank)] + common, cwd=ROOT, env=env2, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=1500)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    l2 = _line(two.stdout.decode())
    _check_pair(l1, l2, per_rank)
    assert l2["config"]["collective_backend"] == "gloo"


def test_eight_rank_bench_on_one_device_dry_run():
    """N = 8 without the hardware: `python bench.py --gpus 8` starting its own eight ranks on ONE device (gloo), against the
    single-process run of the same global batch -- so that the driver's first 8-GPU run is not also the first 8-rank run of the
    control flow (model broadcast, eight slabs, label all-gather, general rows, CNN row, SGAN replicas with graph replay and the
    flat-bucket all-reduce).  No rate is asked of it."""
    per_rank, world = 512, 8
    common = ["--steps", "1", "--warmup", "1", "--train", "1500", "--no-cpu", "--no-pmc", "--no-u8", "--no-slice", "--parity", "128",
              "--general-frames", "256", "--dnn-frames", "256", "--dnn-parity", "32", "--dnn-train-steps", "40"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RML_BENCH_ONE_DEVICE", None)
    import tempfile
    env["MIOPEN_USER_DB_PATH"] = tempfile.mkdtemp(prefix="rml_miopen_db_")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", str(world * per_rank),
                          "--walabot-frames", str(world * 2 * per_rank)] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=1500)
    assert one.returncode == 0, one.stderr.decode()[-3000:]
    l1 = _line(one.stdout.decode())
    env8 = dict(env, RML_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env8.pop(k, None)
    eight = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--frames", str(per_rank),
                            "--walabot-frames", str(2 * per_rank)] + common, cwd=ROOT, env=env8, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=2400)
    assert eight.returncode == 0, eight.stderr.decode()[-3000:]
    l8 = _line(eight.stdout.decode())
    assert l8["n_gpus"] == world and l8["config"]["collective_backend"] == "gloo" and l8["scaling"] == "weak"
    assert l8["config"]["global_frames"] == world * per_rank == l1["config"]["global_frames"]
    assert l8["labels_crc32"] == l1["labels_crc32"]                                                     # eight slabs, gathered in frame order
    assert l8["doc"]["walabot_grid"]["labels_crc32"] == l1["doc"]["walabot_grid"]["labels_crc32"]
    assert l8["summary"]["parity_gate"] == "pass" and l1["summary"]["parity_gate"] == "pass"
    sg = l8["doc"]["sgan_train_step"]
    assert "error" not in sg, sg
    assert sg["replicas_identical"] is True and sg["n_gpus"] == world
    assert l8["doc"]["dnn_forward"]["parity"]["label_mismatch"] == 0
    assert l8["roofline"].get("traffic") is None


def _check_pair(l1, l2, per_rank):
    """l1: the single-process line of the global batch, l2: the two-rank line (_line attaches the verbose rows under "doc")."""
    assert l2["n_gpus"] == 2 and l2["config"]["global_frames"] == 2 * per_rank == l1["config"]["global_frames"]
    # every rank classified its slab of the same global batch: gathered labels == single-process labels, both grids
    assert l2["labels_crc32"] == l1["labels_crc32"]
    assert l2["doc"]["walabot_grid"]["labels_crc32"] == l1["doc"]["walabot_grid"]["labels_crc32"]
    for ln in (l1, l2):
        d = ln["doc"]
        assert d["parity"]["label_calib_mismatch"] == 0 and d["parity"]["label_vote_mismatch"] == 0
        assert d["parity"]["dec_ovo_max_abs_err"] <= 1e-5
        assert d["general_rows"]["parity"]["label_calib_mismatch"] == 0 and d["general_rows"]["parity"]["dec_ovo_max_abs_err"] <= 1e-5
        # the reference-faithful rows (derive -> slice -> SVM in one pass; slices at given voxels): parity against the oracle
        for g in (d, d["walabot_grid"]):
            for key in ("derive_slice_svm", "slice_mode"):
                par = g["slice_rows"][key]["parity"]
                assert par["label_calib_mismatch"] == 0 and par["label_vote_mismatch"] == 0 and par["dec_ovo_max_abs_err"] <= 1e-5
            assert g["slice_rows"]["derive_slice_svm"]["parity"]["derived_target_mismatch"] == 0
        assert ln["summary"]["parity_gate"] == "pass"
    sg = l2["doc"]["sgan_train_step"]
    assert "error" not in sg, sg
    assert sg["replicas_identical"] is True and sg["hip_graph"] is True and sg["n_gpus"] == 2
    assert l2["roofline"].get("traffic") is None and "traffic_note" in l2["doc"]["roofline"]
    assert l1["doc"]["dnn_forward"]["parity"]["label_mismatch"] == 0 and l2["doc"]["dnn_forward"]["parity"]["label_mismatch"] == 0


def test_two_rank_bench_over_rccl_when_two_devices_are_visible():
    """The same comparison on TWO devices over RCCL (backend "nccl"): the all_gather_into_tensor of the labels and the flat-bucket
    gradient all-reduce beside HIP-graph replay on a real communicator.  Arms itself where the box has >= 2 GPUs (the build
    boxes have one: skipped there), so that the first multi-GPU run of the driver is not also the first RCCL run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices (RCCL refuses two ranks on one)")
    per_rank = 1024
    common = ["--steps", "2", "--warmup", "1", "--train", "1500", "--no-cpu", "--no-pmc", "--no-u8", "--parity", "256",
              "--general-frames", "512", "--dnn-frames", "512", "--dnn-parity", "64"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RML_BENCH_ONE_DEVICE", None)
    import tempfile
    env["MIOPEN_USER_DB_PATH"] = tempfile.mkdtemp(prefix="rml_miopen_db_")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", str(2 * per_rank),
                          "--walabot-frames", str(2 * per_rank)] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=1500)
    assert one.returncode == 0, one.stderr.decode()[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", str(per_rank),
                          "--walabot-frames", str(per_rank)] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=1500)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    l1, l2 = _line(one.stdout.decode()), _line(two.stdout.decode())
    _check_pair(l1, l2, per_rank)
    assert l2["config"]["collective_backend"] == "nccl"           # = RCCL on ROCm

