"""N>1 path on CPU: world_size-2 gloo processes exercise the frame sharding and the label all-gather."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_classify(vol):
    # a deterministic per-frame "label" that depends only on the frame's content
    return (vol.reshape(vol.shape[0], -1).sum(dim=1).to(torch.int64) % 3).to(torch.int32)


def _make(lo, hi):
    g = torch.Generator().manual_seed(1234)
    allv = torch.randint(0, 255, (37, 2, 3, 4), generator=g).float()      # same global data set on every rank
    return allv[lo:hi]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from radar_ml_amd import dist as rd
    got = rd.classify_sharded(_fake_classify, n_frames, _make)
    probs = rd.gather_labels(_make(*rd.shard_range(n_frames, rank, world)).reshape(-1, 24)[:, :3].contiguous(), n_frames)
    q.put((rank, got.numpy(), probs.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [36, 37])
def test_two_rank_gloo_sharding_matches_single_process(n_frames):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_classify(_make(0, n_frames)).numpy()
    wantp = _make(0, n_frames).reshape(-1, 24)[:, :3].numpy()
    for rank, got, probs in res:
        np.testing.assert_array_equal(got, want)          # every rank holds all labels, in frame order
        np.testing.assert_array_equal(probs, wantp)


def test_eight_rank_gloo_sharding_with_ragged_slabs():
    """The world size the driver's scaling run ends at: 8 ranks, 37 frames (slabs of 5 and 4), labels and a (frames, 3) array gathered
    in frame order on every rank."""
    world, n_frames = 8, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _fake_classify(_make(0, n_frames)).numpy()
    wantp = _make(0, n_frames).reshape(-1, 24)[:, :3].numpy()
    assert sorted(r for r, _, _ in res) == list(range(world))
    for rank, got, probs in res:
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(probs, wantp)


def test_shard_range_partitions():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from radar_ml_amd import dist as rd
    for n in (0, 1, 7, 8, 65536, 262144, 1000003):
        for world in (1, 2, 4, 8):
            spans = [rd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    t = torch.arange(5)
    assert rd.gather_labels(t) is t                       # no process group: identity


def _bench_sgan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import radar_ml_amd  # noqa: F401
    import bench
    a = types.SimpleNamespace(seed=1234, steps=2, warmup=1)
    env = {"dev": torch.device("cpu"), "rank": rank, "world": world}
    row = bench.run_sgan(a, env, n=6, hw=16, steps=2)          # bench.py's own multi-rank leg of configs[4], small and on CPU
    q.put((rank, row))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_sgan_leg_runs_data_parallel_on_two_gloo_ranks():
    """bench.py's run_sgan -- the N > 1 leg of BASELINE configs[4] -- under world size 2: rank 0 reports the whole-job rate,
    the replicas end bit-identical after the all-reduced updates."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_sgan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    row = res[0]
    assert row["n_gpus"] == 2 and row["global_batch"] == 12 and row["replicas_identical"] is True
    assert row["value"] > 0 and "all-reduce" in row["parallelism"]


def test_bench_helpers_resolve_every_name_they_call():
    """tools/bench_support.py is only exercised on the GPU box (cpu_baseline / roofline.traffic legs of bench.py); a missing helper
    there turns into an ``error`` field of the JSON line instead of a failure.  Check statically that every module-level function
    a helper calls exists (round 3 lost two of them to an editing accident and the bench line lost its CPU baseline)."""
    import ast
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ("tools/bench_support.py", "bench.py"):
        src = open(os.path.join(root, rel)).read()
        tree = ast.parse(src)
        defined = {n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
        imported = set()
        for n in ast.walk(tree):
            if isinstance(n, ast.Import):
                imported |= {(a.asname or a.name).split(".")[0] for a in n.names}
            elif isinstance(n, ast.ImportFrom):
                imported |= {a.asname or a.name for a in n.names}
        import builtins
        known = defined | imported | set(dir(builtins))
        for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
            local = {a.arg for a in fn.args.args + fn.args.kwonlyargs} | {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}
            local |= {n.name for n in ast.walk(fn) if isinstance(n, (ast.FunctionDef, ast.Lambda)) and hasattr(n, "name")}
            local |= {a.arg for n in ast.walk(fn) if isinstance(n, (ast.FunctionDef, ast.Lambda)) for a in n.args.args}
            for call in [n for n in ast.walk(fn) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)]:
                assert call.func.id in known | local, "%s: %s() calls undefined %s()" % (rel, fn.name, call.func.id)
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_support
    for name in ("reference_libs_baseline", "reference_libs_process_pool", "measure_traffic"):
        assert callable(getattr(bench_support, name))


def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus 8` on a box with fewer devices must fail loudly instead of printing an n_gpus = 1 line; a launcher whose
    WORLD_SIZE disagrees with --gpus is refused the same way (neither reaches the GPU work)."""
    import torch
    n = max(torch.cuda.device_count(), 1) + 1
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RML_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"refusing" in r.stderr and b'"n_gpus"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], cwd=ROOT,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"WORLD_SIZE=1" in r.stderr and b'"n_gpus"' not in r.stdout
