"""CPU tier: the host-side mirror of the reference's `model` / `util` surface, and the C-ABI library's exports.
No kernel is launched here (there is no GPU in this tier)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, summarize
from oracle import ref_import

import shapegan_amd.lib as L
from shapegan_amd.model import CHECKPOINT_PATH, LATENT_CODE_SIZE, LATENT_CODES_FILENAME, MODEL_PATH, SavableModule
from shapegan_amd.model.autoencoder import Autoencoder
from shapegan_amd.model.gan import Discriminator, Generator
from shapegan_amd.model.progressive_gan import Discriminator as ProgressiveDiscriminator
from shapegan_amd.model.progressive_gan import RESOLUTIONS
from shapegan_amd.model.sdf_net import SDFNet


def test_library_exports_every_declared_symbol():
    """libshapegan_hip.so loads without a GPU and exports exactly what include/shapegan_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "shapegan_hip.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", header, flags=re.I))
    assert len(declared) >= 30
    lib, comm = ctypes.CDLL(L.LIB_PATH), ctypes.CDLL(L.COMM_PATH)
    for name in sorted(declared):
        home = comm if name in L.COMM_SIGNATURES else lib       # the RCCL exchange lives in libshapegan_comm.so
        assert hasattr(home, name), "missing export: " + name
    assert declared == set(L.SIGNATURES) | set(L.COMM_SIGNATURES), (declared ^ (set(L.SIGNATURES) | set(L.COMM_SIGNATURES)))
    abi = int(re.search(r"#define SG_ABI_VERSION (\d+)", header).group(1))
    assert L.load().sg_abi_version() == abi == 8
    # every place that pins the ABI number agrees with the header (the driver's build check runs __graft_entry__.build())
    assert "sg_abi_version() == %d" % abi in open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "sg_abi_version() != %d" % abi in open(os.path.join(ROOT, "shapegan_amd", "lib.py")).read()
    assert L.load_comm().sg_allreduce_unique_id_bytes() == 128


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "_hip", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libshapegan_hip.so")
    with pytest.raises(RuntimeError, match="libshapegan_hip.so is missing"):
        L.load()


def test_no_fallback_between_libraries(monkeypatch):
    """Dispatch is by tensor device only and never falls back: CPU tensors need the C++ twin (a missing twin raises — no
    eager-PyTorch path), the HIP library is required for every GPU tensor and size query, and one call may not mix devices."""
    monkeypatch.setattr(L, "_cpu", None)
    monkeypatch.setattr(L, "CPU_PATH", "/nonexistent/libshapegan_cpu.so")
    d = Discriminator()
    if not next(d.parameters()).is_cuda:
        with pytest.raises(RuntimeError, match="libshapegan_cpu.so is missing"):
            d(torch.zeros(2, 32, 32, 32))
    monkeypatch.undo()
    L._DeviceOfCall.kind = None
    L._note(True)
    with pytest.raises(RuntimeError, match="both GPU and CPU tensors"):
        L.ptr(torch.zeros(4))          # a CPU tensor joining a call that already saw a GPU tensor
    assert L._DeviceOfCall.kind is None


def test_constants_and_filenames():
    assert (MODEL_PATH, LATENT_CODE_SIZE) == ("models", 128)
    assert CHECKPOINT_PATH == os.path.join("models", "checkpoints")
    assert LATENT_CODES_FILENAME == os.path.join("models", "sdf_net_latent_codes.to")
    g = Generator()
    assert g.filename == "generator.to" and g.get_filename() == os.path.join("models", "generator.to")
    g.filename = "wgan-generator.to"  # train_wgan.py:27 mutates it
    assert g.get_filename(epoch=20) == os.path.join("models", "checkpoints", "wgan-generator-epoch-00020.to")
    assert g.get_filename(epoch=3, filename="sdf_net_latent_codes.to").endswith("sdf_net_latent_codes-epoch-00003.to")
    assert Autoencoder(is_variational=True).filename == "variational-autoencoder-128.to"
    assert Autoencoder(is_variational=False).filename == "autoencoder-128.to"
    p = ProgressiveDiscriminator()
    assert p.filename == "hybrid_progressive_gan_discriminator_0.to"
    p.set_iteration(3)
    assert p.filename == "hybrid_progressive_gan_discriminator_3.to" and p.iteration == 3
    assert RESOLUTIONS == [8, 16, 32, 64]
    assert SDFNet().filename == "sdf_net.to"
    assert isinstance(g, SavableModule) and Discriminator().use_sigmoid is True


def test_state_dict_contract(golden_modules):
    """Keys, shapes and seed-for-seed initial values equal the reference's (fixture 'init' summaries)."""
    cases = [("generator", 11, Generator), ("discriminator", 12, Discriminator),
             ("autoencoder", 14, lambda: Autoencoder(is_variational=False)),
             ("vae", 15, lambda: Autoencoder(is_variational=True)),
             ("progressive_it2_fade03", 22, ProgressiveDiscriminator), ("sdfnet_L128", 30, SDFNet),
             ("sdfnet_L256", 30, lambda: SDFNet(latent_code_size=256))]
    for tag, seed, build in cases:
        torch.manual_seed(seed)
        sd = build().state_dict()
        init = golden_modules.sub(tag + "/init")
        assert set(sd) == set(init), tag
        for k, v in sd.items():
            np.testing.assert_array_equal(summarize(v.float()), init[k], err_msg=tag + ":" + k)
    sd = ProgressiveDiscriminator().state_dict()
    assert len(sd) == 20 and "optional_layers.2.0.weight" in sd and "optional_layer_2.0.weight" in sd
    assert len(Autoencoder().state_dict()) == 69 and "encoder.vae-bn.running_mean" in Autoencoder().state_dict()
    assert len(Generator().state_dict()) == 23 and len(Discriminator().state_dict()) == 8


def test_save_load_roundtrip(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(1)
    a = Discriminator()
    a.filename = "wgan-critic.to"
    a.save()
    a.save(epoch=7)
    assert os.path.exists("models/wgan-critic.to") and os.path.exists("models/checkpoints/wgan-critic-epoch-00007.to")
    b = Discriminator()
    b.filename = "wgan-critic.to"
    b.load()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    # strict=False like the reference: a checkpoint with foreign keys still loads
    sd = torch.load("models/wgan-critic.to")
    sd["unrelated"] = torch.zeros(1)
    torch.save(sd, "models/wgan-critic.to")
    b.load()


def test_chairs_checkpoint_loads(chairs_state):
    net = SDFNet()
    res = net.load_state_dict(chairs_state, strict=False)
    assert not res.missing_keys and not res.unexpected_keys


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present")
def test_voxel_grid_bit_exact_vs_reference():
    from shapegan_amd.util import get_voxel_coordinates
    ref = ref_import.load()
    for r in (8, 16, 32, 64):
        a, b = ref.get_voxel_coordinates(r), get_voxel_coordinates(r)
        assert a.dtype == b.dtype and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_voxel_grid_ordering():
    from shapegan_amd.util import get_voxel_coordinates
    p = get_voxel_coordinates(32)
    assert p.shape == (32768, 3) and p.dtype == np.float32
    np.testing.assert_allclose(p[1], [-1, -1, -0.93548], atol=1e-5)   # z fastest
    np.testing.assert_allclose(p[32], [-1, -0.93548, -1], atol=1e-5)
    np.testing.assert_allclose(p[-1], [1, 1, 1])


def test_flat_optimizer_views_on_cpu():
    """optim._Flat re-points parameters/grads at flat buffers without changing values, shapes or identity."""
    from shapegan_amd import optim
    torch.manual_seed(0)
    d = ProgressiveDiscriminator()
    before = {k: v.clone() for k, v in d.state_dict().items()}
    ids = [id(p) for p in d.parameters()]
    f = optim._Flat(d.parameters())
    assert [id(p) for p in d.parameters()] == ids
    for k, v in d.state_dict().items():
        assert torch.equal(v, before[k])
    assert f.coherent()
    f.zero_grad()  # torch's set_to_none semantics; every slice is open for one direct gradient write
    assert not f.coherent() and all(p.grad is None for p in d.parameters())
    w = d.head[1].weight
    wi = [i for i, p in enumerate(f.params) if p is w][0]
    with torch.no_grad():                              # a plain backward runs with grad mode off
        assert L.grad_destination(w, w.shape) is None  # not inside lib.backward(): gradients stay ordinary tensors
        L._direct_write_depth += 1                     # what lib.backward(loss) does around loss.backward()
        try:
            dst = L.grad_destination(w, w.shape)           # what a weight-gradient kernel asks for
            assert dst is not None and dst.data_ptr() == f.grad.data_ptr() + 4 * f.offsets[wi]
            assert L.grad_destination(w, w.shape) is None  # a second contribution in the same cycle must not clobber the first
            f.zero_grad()
            with torch.enable_grad():
                assert L.grad_destination(w, w.shape) is None  # create_graph backward: gradients must stay ordinary tensors
        finally:
            L._direct_write_depth -= 1
    # gradients that arrived as ordinary tensors (stock autograd, or None) are pulled into the flat buffer on demand
    for i, p in enumerate(d.parameters()):
        p.grad = None if i == 0 else torch.full_like(p, float(i))
    f.adopt_grads()
    # the parameter without a gradient contributes zeros to the wire but stays grad-less (step() skips it, as torch.optim does)
    assert f.params[0].grad is None and float(f.grad[:f.params[0].numel()].abs().sum()) == 0.0
    assert all(f.grad_view_ok(i) for i in range(1, len(f.params)))
    assert float(f.params[3].grad.mean()) == 3.0
    f.adopt_grads(attach_missing=True)
    assert f.coherent()


# ---- input pipeline: host-side index work is bit-exact ---------------------------------------
class _Ints(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


@pytest.mark.parametrize("workers", [0, 2])
def test_loader_index_order_matches_torch_dataloader(workers):
    """ResidentVoxels / VoxelStream visit items in the order DataLoader(shuffle=True) would under the same global seed,
    and leave the global RNG in the same state."""
    from shapegan_amd.datasets import loader_index_order
    for n, seed in ((37, 0), (1000, 123)):
        torch.manual_seed(seed)
        want = torch.cat([b for b in torch.utils.data.DataLoader(_Ints(n), shuffle=True, batch_size=8, num_workers=workers)])
        after_ref = torch.rand(1)
        torch.manual_seed(seed)
        got = loader_index_order(n, True)
        after = torch.rand(1)
        assert torch.equal(got, want) and torch.equal(after, after_ref)
    assert torch.equal(loader_index_order(5, False), torch.arange(5))
    torch.manual_seed(9)
    list(torch.utils.data.DataLoader(_Ints(5), shuffle=False, batch_size=2))
    after_ref = torch.rand(1)
    torch.manual_seed(9)
    loader_index_order(5, False)
    assert torch.equal(torch.rand(1), after_ref)


def test_create_batches_matches_reference_fixture(golden_steps_f2):
    """train_sdf_autodecoder.py:55-69, fixture produced by executing the reference's own function source."""
    from shapegan_amd.datasets import create_batches
    g = golden_steps_f2
    for case, batch in enumerate((64, 50, 512)):
        signs = g["batches/%d/signs" % case]
        np.random.seed(700 + case)
        got = list(create_batches(signs, batch))
        assert [len(b) for b in got] == list(g["batches/%d/sizes" % case])
        flat = np.concatenate(got)
        assert np.array_equal(flat, g["batches/%d/flat" % case])
        assert abs(int(signs[flat].sum()) * 2 - flat.size) == 0          # balanced signs


def test_voxel_dataset_items_match_reference_fixture(golden_steps_f2, tmp_path):
    """datasets.py:16-23 on real .npy files: clamp, optional rescale, NaN / inf / boundary values; glob is sorted."""
    from shapegan_amd.datasets import VoxelDataset
    g = golden_steps_f2
    raw = g["vox/raw"]
    for i, c in enumerate("cadbe"):
        np.save(str(tmp_path / ("%s.npy" % c)), raw[i])
    ordered = [str(tmp_path / ("%s.npy" % c)) for c in "cadbe"]
    ds = VoxelDataset(ordered)
    assert len(ds) == 5
    for i in range(5):
        assert np.array_equal(ds[i].numpy(), g["vox/rescaled"][i], equal_nan=True)
    ds.rescale_sdf = False
    for i in range(5):
        assert np.array_equal(ds[i].numpy(), g["vox/clamped"][i], equal_nan=True)
    assert np.array_equal(VoxelDataset(ordered, clamp=None)[1].numpy(), raw[1], equal_nan=True)
    assert [os.path.basename(f) for f in VoxelDataset.glob(str(tmp_path) + "/**.npy").files] == ["a.npy", "b.npy", "c.npy", "d.npy", "e.npy"]
    with pytest.raises(Exception):
        VoxelDataset.glob(str(tmp_path) + "/nothing/**.npy")
    split = tmp_path / "train.txt"
    split.write_text("a\nmissing\nd\n")
    assert [os.path.basename(f) for f in VoxelDataset.from_split(str(tmp_path) + "/{:s}.npy", str(split)).files] == ["a.npy", "d.npy"]


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present (GPU box)")
def test_point_dataset_matches_reference(tmp_path):
    """datasets.py:53-92 against the real class: same files, same np.random draws, same tensors."""
    import importlib.util
    from shapegan_amd.datasets import PointDataset
    spec = importlib.util.spec_from_file_location("ref_datasets", os.path.join(ref_import.REFERENCE_ROOT, "datasets.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.RandomState(1)
    for kind in ("uniform", "surface"):
        os.makedirs(str(tmp_path / kind))
        for name in ("s0", "s1", "s2"):
            np.save(str(tmp_path / kind / (name + ".npy")), rng.rand(500, 4).astype(np.float32))
    (tmp_path / "train.txt").write_text("s0\ns1\ns2\n")
    theirs, ours = ref.PointDataset.from_split(str(tmp_path), "train", 64), PointDataset.from_split(str(tmp_path), "train", 64)
    assert len(theirs) == len(ours) == 3 and theirs.filenames == ours.filenames
    np.random.seed(11)
    want = [theirs[i] for i in (2, 0, 1)]
    np.random.seed(11)
    got = [ours[i] for i in (2, 0, 1)]
    for a, b in zip(want, got):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and b[0].shape == (64, 4)


def test_sdfnet_backward_tile_layout_is_the_documented_function_of_n():
    """include/shapegan_hip.h: 64-point tiles; the last round of 512 tiles is cut differently when it is between 1/2 and 3/4
    full (256 more 64-point tiles, the rest 32-point tiles) or at most half full and not the only round (all 32-point tiles).
    Host code of the library, callable without a GPU."""
    lib = L.load()
    for n in (1, 63, 64, 65, 16384, 16385, 16416, 20000, 24576, 24577, 32768, 32769, 33100, 200000, 49152, 49153, 50000, 57344,
              57345, 65536, 262144, 1000003):
        tiles = (n + 63) // 64
        rem = tiles % 512
        full = tiles - rem
        if 256 < rem <= 384:
            big = full + 256
            small = (n - 64 * big + 31) // 32
            starts = [64 * t for t in range(big)] + [min(64 * big + 32 * t, n) for t in range(small + 1)]
        elif full == 0 or rem == 0 or rem > 384:
            starts = [min(64 * t, n) for t in range(tiles + 1)]
        else:
            small = (n - 64 * full + 31) // 32
            starts = [64 * t for t in range(full)] + [min(64 * full + 32 * t, n) for t in range(small + 1)]
        blocks = lib.sg_sdfnet_bwd_blocks(n)
        assert blocks == len(starts) - 1, n
        probe = sorted(set(list(range(min(blocks + 1, 5))) + list(range(max(0, blocks - 40), blocks + 1)) + [blocks // 2]))
        for t in probe:
            assert lib.sg_sdfnet_bwd_tile_start(n, t) == starts[t], (n, t)
        assert starts[-1] == n and all(b > a for a, b in zip(starts, starts[1:]))


def test_parameter_epochs_are_per_optimizer_buffer():
    """lib.param_epoch_of: an optimizer's step moves the epoch of the parameters in ITS flat buffer only (the critic's five updates
    per WGAN unit must not invalidate weight images of the generator); unscoped events (graph replay, load_state_dict,
    clip_weights) move everybody's; an epoch value is never handed out twice."""
    import torch
    from shapegan_amd import optim
    a, b = torch.nn.Parameter(torch.randn(10)), torch.nn.Parameter(torch.randn(7))
    c = torch.nn.Parameter(torch.randn(3))            # in no optimizer
    oa, ob = optim.RMSprop([a], lr=0.1), optim.Adam([b], lr=0.1)
    ea, eb, ec = L.param_epoch_of(a), L.param_epoch_of(b), L.param_epoch_of(c)
    L.bump_param_epoch(oa.f.range)
    assert L.param_epoch_of(a) > ea and L.param_epoch_of(b) == eb and L.param_epoch_of(c) == ec
    assert L.param_epoch_of(a, b) == L.param_epoch_of(a)
    ea = L.param_epoch_of(a)
    L.bump_param_epoch(ob.f.range)
    assert L.param_epoch_of(a) == ea and L.param_epoch_of(b) > ea
    L.bump_param_epoch()
    e = L.param_epoch_of(c)
    assert e > eb and L.param_epoch_of(a) == e and L.param_epoch_of(b) == e
    del oa, ob
    import gc
    gc.collect()
    assert all(r[0] != a.data_ptr() for r in L._PARAM_RANGES)


def test_kept_weight_images_follow_object_version_epoch_and_shape():
    """ops._KeptWeightImages (the host side of sg_conv3d_k4s2p1_dgrad_keep): an image is reused only for the same tensor object at
    the same version / parameter epoch / shapes; at most `cap` weights are remembered.  (The GPU tier checks the results.)"""
    import torch
    from shapegan_amd import ops, optim
    kept = ops._KeptWeightImages(cap=3)
    w = torch.nn.Parameter(torch.randn(8, 4, 4, 4, 4))
    ws, unchanged = kept.get(w, 1000, (2, 4))
    assert not unchanged and ws.numel() >= 1000
    ws2, unchanged = kept.get(w, 1000, (2, 4))
    assert unchanged and ws2 is ws
    assert kept.get(w, 1000, (3, 4))[1] is False          # other shapes
    assert kept.get(w, 1000, (3, 4))[1] is True
    with torch.no_grad():
        w.mul_(2.0)                                       # tensor version
    assert kept.get(w, 1000, (3, 4))[1] is False
    opt = optim.RMSprop([w], lr=0.1)                      # re-points w.data into a flat buffer: another address
    assert kept.get(w, 1000, (3, 4))[1] is False
    assert kept.get(w, 1000, (3, 4))[1] is True
    L.bump_param_epoch(opt.f.range)                       # what opt.step() does
    assert kept.get(w, 1000, (3, 4))[1] is False
    assert kept.get(w, 5000, (3, 4))[1] is False          # a larger workspace is a new buffer
    L.bump_param_epoch()                                  # graph replay / load_state_dict / clip_weights
    assert kept.get(w, 5000, (3, 4))[1] is False
    others = [torch.nn.Parameter(torch.randn(4)) for _ in range(3)]
    for o in others:
        kept.get(o, 100, ())
    assert len(kept.entries) == 3 and kept.get(w, 5000, (3, 4))[1] is False      # w was the oldest: dropped


def test_wgrad_act_path_is_refused_outside_its_limits(monkeypatch):
    """ConvFwd.backward takes the fused weight-gradient + activation-backward kernel only when
    sg_conv3d_k4s2p1_wgrad_act will really serve the call — the shape is eligible (incl. the 32-bit buffer ranges the kernel
    itself checks) AND its scratch fits under the workspace cap — and the two-pass path otherwise, instead of raising.  (Since
    round 3 the kernel reads the grid in place: the scratch is the 8.5 MB of partial tiles, whatever the batch.)  Host code only."""
    from shapegan_amd import ops
    lib = L.load()

    def served(batch, r):
        x = torch.empty(batch, 1, r, r, r, device="meta")
        w = torch.empty(64, 1, 4, 4, 4, device="meta")
        y = torch.empty(batch, 64, r // 2, r // 2, r // 2, device="meta")
        return ops._wgrad_act_served(x, w, y, L.ACT_LEAKY)
    assert served(128, 32) and served(16, 64) and served(240, 64)      # BASELINE shapes and the large batch ADVICE r2 named
    assert lib.sg_conv3d_k4s2p1_wgrad_workspace_bytes(240, 1, 64, 32, 32, 32) <= 16 << 20
    assert not served(256, 64)                          # 256 x 64 x 32^3 x 4 B = 2 GiB of dy: outside the 32-bit buffer range
    assert not served(2048, 32)
    assert lib.sg_conv3d_k4s2p1_wgrad_act_eligible(2048, 1, 64, 16, 16, 16, L.ACT_LEAKY) == 0
    assert not served(4, 32)                            # too small: the generic path is as good
    monkeypatch.setattr(ops, "_WGRAD_WS_CAP", 1 << 20)  # a cap below the scratch: refused, not attempted
    assert not served(128, 32)


def test_call_device_state_is_thread_local_and_cleared_on_rejection():
    """ADVICE r2 / VERDICT r2 weak #6b: the device of the call being assembled is per thread, and a rejected argument leaves
    nothing behind for the next call."""
    import threading
    L.reset_call_state()
    L._note(True)                                       # this thread is half-way through assembling a GPU call
    seen = {}

    def other():
        seen["start"] = L._DeviceOfCall.kind            # must not see the main thread's state
        L.ptr(torch.zeros(4))
        seen["after"] = L._DeviceOfCall.kind
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == {"start": None, "after": "cpu"}
    assert L._DeviceOfCall.kind == "cuda"
    L.reset_call_state()
    L.ptr(torch.zeros(4))
    with pytest.raises(RuntimeError, match="contiguous"):
        L.ptr(torch.zeros(4, 4).t())
    assert L._DeviceOfCall.kind is None


def test_native_exchange_negotiation_falls_back_loudly_and_uniformly():
    """VERDICT r2 next #4d: the C-ABI RCCL exchange is the default under the nccl backend, but it must come up on EVERY rank and
    pass a verification exchange, or every rank closes what it built and falls back to torch.distributed — with a message, never
    with a hang or a mixed transport.  The negotiation logic with fake communicators (no GPU / RCCL in this tier)."""
    from shapegan_amd import parallel

    class FakeComm(object):
        closed = False

        def close(self):
            self.closed = True

    def run(make, verify, others_ok=(True, True)):
        logs, votes = [], iter(others_ok)
        comm, reason = parallel.negotiate_native(make, verify, agree=lambda ok: ok and next(votes), log=logs.append)
        return comm, reason, logs

    made = []

    def make_ok():
        made.append(FakeComm())
        return made[-1]

    def make_fail():
        raise RuntimeError("ncclCommInitRank did not return within 120 s")

    comm, reason, logs = run(make_ok, lambda c: True)
    assert comm is made[-1] and reason == "" and not logs and not comm.closed
    comm, reason, logs = run(make_fail, lambda c: True)
    assert comm is None and "did not return" in reason and len(logs) == 1 and "FALLING BACK" in logs[0]
    comm, reason, logs = run(make_ok, lambda c: True, others_ok=(False, True))       # another rank failed to initialise
    assert comm is None and made[-1].closed and "another rank" in reason and "FALLING BACK" in logs[0]
    comm, reason, logs = run(make_ok, lambda c: False)                                # wrong sums through the new communicator
    assert comm is None and made[-1].closed and "disagreed" in reason
    comm, reason, logs = run(make_ok, lambda c: (_ for _ in ()).throw(RuntimeError("launch failed")))
    assert comm is None and made[-1].closed and "launch failed" in reason
    comm, reason, logs = run(make_ok, lambda c: True, others_ok=(True, False))        # another rank failed the verification
    assert comm is None and made[-1].closed and "another rank failed the verification" in reason
    # single process: nothing to negotiate, nothing loaded
    assert parallel.native_comm() is None and parallel.TRANSPORT["name"] == "none"


def test_optimizer_skips_parameters_without_gradient_and_merges_runs():
    """a parameter whose gradient is None is skipped by step() as in torch.optim — also after adopt_grads() (the
    data-parallel exchange zero-fills its slice for the wire but leaves p.grad None, so Adam's moments do not keep moving it) —
    and the remaining parameters are updated in maximal contiguous runs.  CPU twin, against torch.optim.Adam / RMSprop."""
    from shapegan_amd import optim
    L.load_cpu()
    torch.manual_seed(0)
    shapes = [(5, 3), (7,), (4, 4), (2,), (9,)]
    for name, mine, theirs in (("adam", lambda ps: optim.Adam(ps, lr=1e-2), lambda ps: torch.optim.Adam(ps, lr=1e-2)),
                               ("rmsprop", lambda ps: optim.RMSprop(ps, lr=1e-2), lambda ps: torch.optim.RMSprop(ps, lr=1e-2))):
        ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        a, b = mine(ps), theirs(qs)
        for step, missing in enumerate(((), (2,), (2,), (0, 4), ())):
            a.zero_grad()
            b.zero_grad(set_to_none=True)
            gs = [torch.randn(s) for s in shapes]
            for i, (p, q, g) in enumerate(zip(ps, qs, gs)):
                if i in missing:
                    continue
                q.grad = g.clone()
                o = a.f.offsets[i]
                a.f.grad[o:o + g.numel()].copy_(g.reshape(-1))
                p.grad = a.f.grad[o:o + g.numel()].view(p.shape) if (step + i) % 3 else g.clone()   # slice view or a stray tensor
            if step == 2:
                a.f.adopt_grads()                # what GradBucket.finish() does before the exchange
                assert ps[2].grad is None
            segs = a._segments()
            if step in (1, 2) and all(a.f.grad_view_ok(i) for i in (0, 1, 3, 4)):
                assert len(segs) == 2            # [0, 1] and [3, 4]: one launch per run
            a.step()
            b.step()
            for i, (p, q) in enumerate(zip(ps, qs)):
                torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-5, atol=1e-6, msg=lambda m: "%s step %d param %d: %s" % (name, step, i, m))
    with pytest.raises(RuntimeError, match="every parameter needs a gradient"):
        ps = [torch.nn.Parameter(torch.randn(3)) for _ in range(2)]
        opt = optim.Adam(ps, lr=1e-3, capturable=True)
        opt.zero_grad()
        ps[0].grad = torch.randn(3)
        opt.step()


def test_adam_keeps_per_parameter_step_counters_when_all_grads_are_flat_views():
    """with every gradient a flat-buffer view (the GPU direct-write case, and the state after every
    GradBucket.finish() -> adopt_grads()) but DIFFERENT per-parameter step counters (a parameter had no gradient on earlier
    steps), the update must still use each parameter's own bias-correction step: one segment per run of equal counters, not one
    launch with steps[0].  CPU twin, against torch.optim.Adam."""
    from shapegan_amd import optim
    L.load_cpu()
    torch.manual_seed(3)
    shapes = [(6,), (5, 2), (3,)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a, b = optim.Adam(ps, lr=1e-2), torch.optim.Adam(qs, lr=1e-2)
    for step in range(6):
        a.zero_grad()
        b.zero_grad(set_to_none=True)
        for i, (p, q, s) in enumerate(zip(ps, qs, shapes)):
            if i == 0 and step < 3:
                continue                         # parameter 0 joins late
            g = torch.randn(s)
            q.grad = g.clone()
            p.grad = g.clone()                   # a stray tensor; adopt_grads() moves it into the flat slice
        a.f.adopt_grads()
        if step >= 3:
            assert a.f.coherent()
            assert len(a._segments(keys=[st + 1 for st in a.steps])) == 2      # [param 0] and [params 1, 2]
        a.step()
        b.step()
        for i, (p, q) in enumerate(zip(ps, qs)):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-5, atol=1e-6,
                                       msg=lambda m: "step %d param %d: %s" % (step, i, m))
    assert a.steps == [3, 6, 6]


def test_autograd_grad_returns_ordinary_tensors_not_flat_slices():
    """torch.autograd.grad(loss, params) hands gradients to the caller, who may keep them; they must not alias the
    optimizer's flat gradient buffer (the next backward writes there).  Direct slice writes happen only inside lib.backward()."""
    from shapegan_amd import optim
    from shapegan_amd.model.gan import Discriminator
    L.load_cpu()
    torch.manual_seed(1)
    d = Discriminator()
    if next(d.parameters()).is_cuda:
        pytest.skip("CPU-twin check")
    opt = optim.RMSprop(d.parameters(), lr=1e-4)
    x = torch.rand(2, 32, 32, 32) * 2 - 1
    opt.zero_grad()
    grads = torch.autograd.grad(d(x).mean(), list(d.parameters()))
    lo, hi = opt.f.grad.data_ptr(), opt.f.grad.data_ptr() + 4 * opt.f.total
    assert all(not (lo <= g.data_ptr() < hi) for g in grads)
    kept = [g.clone() for g in grads]
    opt.zero_grad()
    L.backward(d(x * 0.5).mean())
    assert opt.f.coherent()                       # the trainers' backward wrote the slices in place
    for g, k in zip(grads, kept):
        assert torch.equal(g, k)                  # and left the tensors the caller kept alone
    opt.zero_grad()
    d(x * 0.25).mean().backward()                 # a plain backward: ordinary tensors, copied in on demand by step() / the exchange
    assert all(not (lo <= p.grad.data_ptr() < hi) for p in d.parameters())
    opt.step()


def test_grouped_generator_pass_refuses_an_unserved_last_layer_before_touching_batchnorm_buffers():
    """run_stack_groups decided "is the last ConvTranspose3d(C -> 1) served" AFTER the producers and the grouped
    BatchNorms had run — a generator variant with C > 64 crashed with its running statistics already advanced for every group.  The
    shape is planned from the module attributes and refused before anything is launched; the caller then evaluates group by
    group."""
    import torch.nn as nn
    from shapegan_amd.model import stack
    torch.manual_seed(0)
    layers = nn.Sequential(nn.ConvTranspose3d(8, 96, 4, 2, 1), nn.BatchNorm3d(96), nn.LeakyReLU(0.2),
                           nn.ConvTranspose3d(96, 1, 4, 2, 1), nn.Tanh())
    assert stack._shape_after([(layers[0], layers[1], (1, 0.2))], (6, 8, 4, 4, 4)) == (6, 96, 8, 8, 8)
    gen = nn.Sequential(nn.ConvTranspose3d(128, 256, 4, 1), nn.BatchNorm3d(256), nn.LeakyReLU(0.2), nn.ConvTranspose3d(256, 128, 4, 2, 1))
    assert stack._shape_after([(gen[0], gen[1], None), (gen[3], None, None)], (5, 128, 1, 1, 1)) == (5, 128, 8, 8, 8)
    x = torch.randn(6, 8, 4, 4, 4)
    outs = [torch.empty(2, 1, 16, 16, 16) for _ in range(3)]
    with torch.no_grad():
        assert stack.run_stack_groups(layers, x, 3, outs) is False       # 96 channels: not served by sg_convT3d_k4s2p1_to1_pre
    assert int(layers[1].num_batches_tracked) == 0 and torch.equal(layers[1].running_mean, torch.zeros(96))
