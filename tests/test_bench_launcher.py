"""CPU tier: `python bench.py --gpus N` launches its own ranks (bench.self_launch) — the driver's N = 1 command is the plain form
`python bench.py --gpus 1 --steps K --warmup W`, so the N > 1 form must work without torch.distributed.run too (the reference's
only multi-GPU script, train_hybrid_progressive_gan.py:62-68, needs no launcher either).  The launcher is exercised here on a
stand-in rank script (the real ranks need a GPU: tests/test_gpu_bench_dp.py runs them)."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

from conftest import ROOT

import bench


def _script(tmp_path, body):
    path = tmp_path / "rank.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def _launch(tmp_path, body, gpus, argv=()):
    code = ("import sys, json; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(%d, %r, script=%r))" % (ROOT, gpus, list(argv), _script(tmp_path, body)))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))


def test_launcher_gives_every_rank_the_torchrun_environment_and_passes_rank0s_line_through(tmp_path):
    res = _launch(tmp_path, """
        import json, os, sys
        import torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
        dist.init_process_group("gloo", rank=rank, world_size=world)      # the rendezvous the ranks of bench.py make
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        print("rank %d says hello on stderr" % rank, file=sys.stderr)
        print(json.dumps({"n_gpus": world, "sum": float(t), "argv": sys.argv[1:]}))       # every rank prints: only rank 0's is passed on
        dist.destroy_process_group()
    """, 3, ["--gpus", "3", "--steps", "2"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"n_gpus": 3, "sum": 6.0, "argv": ["--gpus", "3", "--steps", "2"]}
    for r in range(3):
        assert "[rank %d] rank %d says hello on stderr" % (r, r) in res.stderr


def test_launcher_returns_the_failing_ranks_code_and_stops_the_others(tmp_path):
    t0 = time.time()
    res = _launch(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            print("rank 1 gives up", file=sys.stderr)
            sys.exit(7)
        time.sleep(600)          # the other ranks would sit in a collective for ever
    """, 3)
    assert res.returncode == 7
    assert time.time() - t0 < 120
    assert "[rank 1] rank 1 gives up" in res.stderr and "rank 1 exited with code 7" in res.stderr


def test_a_world_size_that_contradicts_gpus_is_refused_with_both_launch_forms_named(tmp_path):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode != 0
    assert "python bench.py --gpus N" in res.stderr and "torch.distributed.run" in res.stderr


def test_cpu_baseline_fills_all_cores_with_pinned_replicas(monkeypatch):
    """bench.py's cpu_baseline: one process stops scaling at ~32 threads, so ALL host cores are filled with
    concurrent replicas of the same 5 + 1 step; the aggregate is steps summed over the replicas / the slowest replica's time.  Here:
    two replicas of two threads each on a batch of 2 (the replicas import only torch and oracle/, never the GPU library)."""
    import torch
    import torch.nn as nn
    import shapegan_amd.model.gan as G
    monkeypatch.setattr(nn.Module, "cuda", lambda self, *a, **k: self)       # CPU tier: the shells' constructors call .cuda()
    monkeypatch.setattr(G, "default_device", torch.device("cpu"))
    torch.manual_seed(0)
    g, c = G.Generator(), G.Discriminator()
    g_state = {k: v.detach().clone() for k, v in g.state_dict().items()}
    c_state = {k: v.detach().clone() for k, v in c.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    reals = [torch.rand(2, 32, 32, 32, generator=gen) * 2 - 1 for _ in range(5)]
    zs = [torch.randn(2, 128, generator=gen) for _ in range(5)]
    zg = torch.randn(2, 128, generator=gen)
    agg = bench._cpu_replicas(2, 2, 1, reals, zs, zg, g_state, c_state)
    assert agg is not None and agg["replicas"] == 2 and agg["steps_each"] == 1
    assert agg["steps_per_s"] > 0 and abs(agg["steps_per_s"] - 2 * 1 / agg["slowest_s"]) < 1e-2 * agg["steps_per_s"]
