"""GPU tier: the driver's multi-GPU bench command, rehearsed end to end on the ONE GPU a test box has.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...

with SG_DIST_BACKEND=gloo both ranks sit on cuda:0 (RCCL cannot form a communicator with two ranks on one device; gloo carries
the same flat-buffer exchange through the host).  Everything bench.py does for N > 1 runs: rendezvous, per-rank data, GradBucket
arm / tail exchange from inside backward / finish, the MAX-reduce of the elapsed time, the `comm` record, the replica digests, the
closing barrier and teardown — for every --config, i.e. every BASELINE workload incl. the two the 8-GPU run is defined on
(train_hybrid_progressive_gan.py:62-68 is the reference's one multi-GPU site)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_bench(config, extra_env, steps=2, warmup=1, timeout=900, plain=False, scaling=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"SG_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "4"})
    env.update(extra_env)
    tail = ["bench.py", "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--no-extras", "--config", config]
    if scaling:
        tail += ["--scaling", scaling]
    if plain:      # the driver's N = 1 command shape with N = 2: bench.py launches its own ranks
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, "bench.py --gpus 2 --config %s failed (%d):\n%s\n%s" % (config, res.returncode, res.stdout[-2000:], res.stderr[-4000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly ONE JSON line, got %d:\n%s" % (len(lines), res.stdout[-2000:])
    return json.loads(lines[0]), res.stderr


def _check_line(line, config, steps=2, warmup=1, scaling="weak"):
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["warmup"] == warmup
    assert line["config"]["parallelism"] == "dp2" and line["scaling"] == scaling and line["higher_is_better"] is True
    assert line["value"] > 0 and line["ms_per_step"] > 0
    per_step = {"wgan": 1.0, "hybrid_progressive": 1.0, "hybrid_wgan": 1.0, "sdf": 0.2}[config]
    reference_batch = {"wgan": 64, "hybrid_progressive": 16, "hybrid_wgan": 8, "sdf": 200000}[config]
    if scaling == "weak":
        # whole-job aggregate over both ranks: value = 2 * steps * units / elapsed; every rank runs the reference's batch
        assert abs(line["value"] - 2 * per_step / (line["ms_per_step"] * 1e-3)) <= 2e-3 * line["value"]
        assert line["config"]["per_gpu_batch"] == reference_batch and line["config"]["global_batch"] == 2 * reference_batch
    else:
        # strong: the two ranks share ONE reference step (64 -> 32 samples, 16 -> 8 shapes, 200 000 -> 100 000 points per GPU)
        assert abs(line["value"] - per_step / (line["ms_per_step"] * 1e-3)) <= 2e-3 * line["value"]
        assert line["config"]["per_gpu_batch"] == reference_batch // 2 and line["config"]["global_batch"] == reference_batch
    comm = line["comm"]
    assert comm["world"] == 2 and comm["allreduce_bytes_per_step"] > 1000
    assert comm["replicas_bit_identical"] is True, "rank 0 / rank 1 parameters differ after the run: %r" % (comm,)
    # what the exchange cost the compute stream / the host, measured in the timed region, and which RCCL the C-ABI library binds
    assert comm["exposed_ms_per_step"] >= 0 and comm["host_wait_ms_per_step"] >= 0 and comm["exchanges_waited_per_step"] >= 1
    assert comm["rccl_header_version"].count(".") == 2 and comm["rccl_runtime_version"].count(".") == 2
    assert comm["rccl_header_version"].split(".")[0] == comm["rccl_runtime_version"].split(".")[0]
    assert comm["rccl_path"].endswith("librccl.so") or "loader default" in comm["rccl_path"]
    return comm


@pytest.mark.parametrize("config", ["wgan", "hybrid_progressive", "hybrid_wgan", "sdf"])
def test_bench_two_ranks_end_to_end_on_one_gpu(config):
    line, _ = _run_bench(config, {})
    comm = _check_line(line, config)
    assert comm["transport"] == "torch-gloo" and comm["reason"] == "backend is not nccl"


@pytest.mark.parametrize("config", ["wgan", "hybrid_progressive", "hybrid_wgan", "sdf"])
def test_bench_plain_command_launches_its_own_ranks(config):
    """`python bench.py --gpus 2 --config <c>` with no launcher and no WORLD_SIZE: the same line as the torchrun form."""
    line, err = _run_bench(config, {}, plain=True)
    comm = _check_line(line, config)
    assert comm["transport"] == "torch-gloo"
    assert "[rank 0]" in err or "[rank 1]" in err or err == ""       # rank output, when there is any, carries its rank


@pytest.mark.parametrize("config", ["wgan", "hybrid_progressive", "hybrid_wgan", "sdf"])
def test_bench_strong_scaling_splits_the_reference_batch(config):
    """`--scaling strong` ("weak scaling as the headline, strong scaling alongside"): the reference's batch is split over
    the ranks, `value` counts ONE reference step per step time, the replicas stay bit-identical."""
    line, _ = _run_bench(config, {}, plain=True, scaling="strong")
    _check_line(line, config, scaling="strong")


def test_native_exchange_refuses_two_ranks_on_one_device_loudly_and_uniformly():
    """The failure the negotiation was written for, provoked on one GPU: both ranks try to build the C-ABI RCCL communicator on
    cuda:0, RCCL rejects it ("duplicate GPU") on every rank, all ranks agree to fall back to torch.distributed and say so on
    stderr — no hang, no mixed transport — and the run completes with identical replicas."""
    line, err = _run_bench("wgan", {"SG_NATIVE_ALLREDUCE": "force", "SG_COMM_INIT_TIMEOUT": "90"})
    comm = _check_line(line, "wgan")
    assert comm["transport"] == "torch-gloo" and comm["reason"].startswith("fallback: ")
    assert err.count("FALLING BACK to torch.distributed all-reduce") == 2, err[-3000:]      # one message per rank
