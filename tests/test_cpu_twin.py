"""CPU tier: libshapegan_cpu.so, the plain-C++ twin of the C ABI (SURVEY.md 8b `*_cpu`, BASELINE configs[0] "on CPU").

The twin is exercised through the SAME Python shells and the SAME parity checks as the HIP kernels: the bodies of the GPU-tier
tests (tests/test_gpu_modules.py, tests/test_gpu_ops.py) are re-run here with `.cuda()` turned into a no-op, so every tensor
stays on the CPU and shapegan_amd.lib dispatches each call to `sg_<name>_cpu`.  What is compared against is unchanged: the
fixtures made from the real reference and the CPU oracle."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn as nn

import shapegan_amd.lib as L
import test_gpu_modules as M
import test_gpu_ops as OPS
import test_gpu_losses as LOSS
import test_gpu_fullsize as FULL


@pytest.fixture()
def on_cpu(monkeypatch):
    """Tensors / modules asked to move to the GPU stay where they are: the test body then runs on the CPU twin."""
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self.clone())   # a copy, like a device transfer
    monkeypatch.setattr(nn.Module, "cuda", lambda self, *a, **k: self)
    import shapegan_amd.util as U
    monkeypatch.setattr(U, "device", torch.device("cpu"))
    for mod in ("shapegan_amd.model.gan", "shapegan_amd.model.autoencoder"):
        import importlib
        monkeypatch.setattr(importlib.import_module(mod), "default_device", torch.device("cpu"))
    for mod in (M, OPS, LOSS):
        monkeypatch.setattr(mod, "DEV", "cpu")
    twin = L.load_cpu()
    yield twin


def test_twin_exports_every_entry_point():
    """`sg_<name>_cpu` exists for every entry point of the header except the size queries / forced-kernel test hooks."""
    lib = ctypes.CDLL(L.CPU_PATH)
    twins = [n for n in L.SIGNATURES if n not in L.NO_TWIN]
    assert len(twins) >= 50
    for n in twins:
        assert hasattr(lib, n + "_cpu"), "missing twin: " + n
    for n in L.NO_TWIN:
        assert n.endswith("_impl") or n.endswith("_workspace_bytes") or n.endswith("_workspace_bytes_for") or n in (
            "sg_abi_version", "sg_last_error", "sg_sdfnet_packed_floats", "sg_sdfnet_acts_floats", "sg_sdfnet_bwd_blocks", "sg_sdfnet_bwd_tile_start", "sg_sdf_batch_sort_max_shapes",
            "sg_sdfgen_acts_floats", "sg_sdfgen_packed_norm_offset", "sg_sdfgen_bwd_blocks", "sg_pointnet_packed_floats",
            "sg_conv3d_k4s2p1_wgrad_act_eligible", "sg_convT3d_k4s2p1_to1_pre_eligible", "sg_conv3d_k4s2p1_wgrad_dy_image",
            "sg_conv3d_k4s2p1_image_layout")


def test_dispatch_is_by_tensor_device_only(on_cpu):
    from shapegan_amd import ops
    x = torch.randn(1, 2, 4, 4, 4)
    w = torch.randn(3, 2, 4, 4, 4)
    y = ops.conv_fwd_raw(x, w, None)
    torch.testing.assert_close(y, torch.nn.functional.conv3d(x, w, None, stride=2, padding=1), rtol=1e-5, atol=1e-5)
    # a size query has no tensor: it is host code of the HIP library and does not consume / need a device
    assert L.load().sg_conv3d_k4s2p1_fwd_workspace_bytes(1, 2, 3, 2, 2, 2) >= 0
    # a compute entry called without any tensor argument naming its device refuses to guess
    with pytest.raises(RuntimeError, match="without a tensor argument"):
        L.load().sg_clamp(0, 4, -1.0, 1.0, None)


# ---- kernel families (bodies from tests/test_gpu_ops.py) ---------------------------------------------------------------------
@pytest.mark.parametrize("N,Ci,Co,R", [(2, 3, 5, 8), (1, 1, 4, 6), (3, 8, 1, 4), (1, 2, 2, 2), (2, 24, 48, 8)])
def test_conv3d(on_cpu, N, Ci, Co, R):
    OPS.test_conv3d_fwd_dgrad_wgrad(N, Ci, Co, R)


@pytest.mark.parametrize("N,Ci,Co,R", [(2, 16, 8, 4), (3, 8, 1, 8), (1, 5, 3, 3)])
def test_conv_transpose3d(on_cpu, N, Ci, Co, R):
    OPS.test_conv_transpose3d(N, Ci, Co, R)


def test_conv_wgrad_through_activation(on_cpu):
    OPS.test_conv_wgrad_through_activation(2, 5, 4, 1)
    OPS.test_conv_wgrad_through_activation(1, 3, 2, 2)


def test_conv_wgrad_act_fallback(on_cpu, monkeypatch):
    OPS.test_conv_wgrad_act_falls_back_when_scratch_exceeds_cap(monkeypatch)


def test_data_writes_are_seen(on_cpu, monkeypatch):
    OPS.test_a_write_through_data_is_never_served_a_stale_weight_image(monkeypatch)


def test_from_sdf_zero_channels(on_cpu):
    OPS.test_conv_from_sdf_zero_channels()


@pytest.mark.parametrize("M_,N_,K_", [(64, 128, 256), (4, 128, 256), (5, 7, 3)])
def test_linear(on_cpu, M_, N_, K_):
    OPS.test_linear_fwd_bwd(M_, N_, K_)


def test_gemm_double_backward(on_cpu):
    OPS.test_gemm_double_backward()


@pytest.mark.parametrize("N,C,S", [(4, 8, 64), (4, 256, 1), (3, 7, 27)])
def test_batchnorm(on_cpu, N, C, S):
    OPS.test_batchnorm_train_fwd_bwd(N, C, S)


def test_activations_and_reductions(on_cpu):
    OPS.test_activations()
    OPS.test_mean_reduction()
    OPS.test_gather_scatter_rows_bit_exact()


def test_optimizers_match_torch(on_cpu):
    OPS.test_rmsprop_adam_clamp_match_torch()


@pytest.mark.parametrize("N,latent", [(1, 128), (63, 128), (777, 256), (130, 16)])
def test_sdfnet_points(on_cpu, N, latent):
    OPS.test_sdfnet_points_mode(N, latent)


@pytest.mark.parametrize("S,pps", [(3, 512), (1, 37)])
def test_sdfnet_shapes(on_cpu, S, pps):
    OPS.test_sdfnet_shapes_mode(S, pps)


@pytest.mark.parametrize("S,N", [(5, 700), (3, 64), (7, 33100)])   # 33100: 512 tiles of 64 + 11 of 32 in the partial-sum layout
def test_sdfnet_segments(on_cpu, S, N):
    OPS.test_sdfnet_segments_mode(S, N)


@pytest.mark.parametrize("S,pc,N", [(64, 200, 20000), (5, 40, 700), (300, 7, 1000), (4097, 3, 9000)])
def test_sdf_batch_sort(on_cpu, S, pc, N):
    OPS.test_sdf_batch_sort(S, pc, N)


def test_layernorm_segmax_colsum(on_cpu):
    OPS.test_layernorm_act(300, 256, 100, True, 2)
    OPS.test_layernorm_act(64, 64, 64, False, 0)
    OPS.test_segmax_and_adjoints(3, 50, 64)
    OPS.test_segmax_nan_and_inf_follow_torch_max(3, 50, 512)
    OPS.test_segmax_nan_and_inf_follow_torch_max(2, 1000, 130)
    OPS.test_colsum_tall()


def test_losses_and_blends(on_cpu):
    LOSS.test_weighted_l1_matches_reconstruction_loss((3, 7, 5))
    LOSS.test_kld_matches_reference()
    for shape in ((4, 32, 32, 32), (3, 7, 5), (1,), (2049,)):
        LOSS.test_voxel_difference_bit_exact(shape)
    LOSS.test_mean_sq_plain_and_row_weighted()
    LOSS.test_deepsdf_loss_is_the_sum_of_its_two_ops_bit_for_bit(20000, 64, 128, True)
    LOSS.test_deepsdf_loss_is_the_sum_of_its_two_ops_bit_for_bit(5000, 5000, 16, False)
    LOSS.test_deepsdf_loss_is_the_sum_of_its_two_ops_bit_for_bit(1, 1, 1, True)
    LOSS.test_lerp_rows_bit_exact()
    LOSS.test_mean_difference_matches_torch(128, 64)
    LOSS.test_mean_difference_matches_torch(7, 0)
    LOSS.test_gradient_penalty_value_and_gradient(5, (7, 3))
    LOSS.test_subsample2_bit_exact_and_adjoint()
    LOSS.test_fade_blend_first_and_second_order(1, 0.3)
    LOSS.test_scatter_max_ragged(1000, 7, 64, True)
    LOSS.test_scatter_max_ragged(10, 12, 5, False)
    LOSS.test_discriminator_clip_weights()


@pytest.mark.parametrize("late_polls", [0, 2])
def test_bad_batch_index_update_is_guarded(on_cpu, late_polls, monkeypatch):
    LOSS.test_bad_batch_index_leaves_the_state_before_the_bad_batch(False, late_polls, monkeypatch)


# ---- modules and training steps (bodies from tests/test_gpu_modules.py: reference-made fixtures + CPU oracle) ----------------
def test_bad_index_words_per_trainer(on_cpu):
    LOSS.test_bad_batch_index_words_belong_to_one_trainer()


def test_generator_and_discriminator(on_cpu, golden_modules):
    M.test_generator(golden_modules)
    M.test_generator_writes_into_a_given_tensor_without_grad_mode()
    M.test_discriminator(golden_modules)


def test_autoencoders(on_cpu, golden_modules, monkeypatch):
    M.test_autoencoder_classic(golden_modules)
    M.test_vae_forward(golden_modules, monkeypatch)


@pytest.mark.parametrize("it,fade", [(0, 1.0), (1, 0.4), (2, 0.3)])
def test_progressive_discriminator(on_cpu, golden_modules, it, fade):
    M.test_progressive_discriminator(golden_modules, it, fade)


def test_sdfnet_module_and_gradient_penalty(on_cpu, golden_modules):
    M.test_sdfnet_module(golden_modules, 128)
    M.test_gradient_penalty_double_backward(golden_modules)


def test_training_trajectories(on_cpu, golden_steps, monkeypatch):
    """train_wgan.py, train_autoencoder.py (configs[0]'s loop body), train_sdf_autodecoder.py, train_hybrid_wgan.py steps
    on the native optimizers and modules, all on the CPU twin, against the reference-made trajectories."""
    M.test_wgan_trajectory(golden_steps, monkeypatch)
    M.test_autoencoder_trajectory(golden_steps)
    M.test_sdf_autodecoder_trajectory(golden_steps)
    M.test_hybrid_wgan_trajectory(golden_steps)


def test_point_gan_family_on_the_fused_generator(on_cpu, golden_steps_f4):
    """SURVEY.md 8f rank 4 on the twin: SDFGenerator(128, 256, 8) runs the LayerNorm form of the fused MLP entry points
    (sg_sdfgen_pack / _fwd / _bwd / _bwd_finish, sg_gemm_nt_batched_lnrelu) — against the reference-made fixtures, the oracles and
    the module's own layer-by-layer path."""
    assert L.load().sg_sdfgen_packed_norm_offset(0) == 809216 and L.load().sg_sdfgen_packed_norm_offset(1) == 811008   # (the twin's copy)
    M.test_point_gan_modules(golden_steps_f4)
    M.test_point_gan_trajectory(golden_steps_f4)
    M.test_sdf_generator_fused_vs_layerwise_and_oracle()
    M.test_gemm_nt_lnrelu_matches_torch()
    M.test_point_gan_sparse_max_adjoint_matches_dense_and_oracle()
    for b, p in ((3, 1056), (1, 32), (5, 64)):
        M.test_pointnet_select_matches_layerwise(b, p)
    M.test_rowdot_family_matches_torch_to_second_order()
    for args in ((6, 512, 1280, 200), (2, 100, 300, 300), (1, 1, 5, 1)):
        M.test_gather_rows_grouped_and_its_deterministic_adjoint(*args)


def test_adam_step_together(on_cpu):
    M.test_adam_step_together_equals_separate_steps()


def test_two_threads_drive_two_modules_concurrently(on_cpu):
    """The DataParallel shape (train_hybrid_progressive_gan.py:62-68: replicas called from one Python thread each; SURVEY.md 8b
    "safe under DataParallel-style multi-thread calls"): two threads run forward + backward of native modules at the same time,
    sharing the SDFNet's pack cache object the way shallow replica copies do.  ctypes releases the GIL inside every library
    call, so dispatcher state, workspaces and caches really are used concurrently.  Results must equal the serial run bit for
    bit (the twin is deterministic for a fixed thread count)."""
    import copy
    import threading
    from shapegan_amd.model.gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    torch.manual_seed(3)
    net, disc = SDFNet(device="cpu"), Discriminator()
    replica = copy.copy(net)                             # nn.DataParallel's replicate(): shallow copy sharing __dict__ entries
    replica._parameters = {k: v for k, v in net._parameters.items()}
    replica._modules = dict(net._modules)
    pts = [torch.rand(700, 3) * 2 - 1 for _ in range(2)]
    lat = [torch.randn(700, 128) * 0.1 for _ in range(2)]
    vox = torch.rand(3, 32, 32, 32) * 2 - 1

    def sdf_job(module, i, out):
        p = pts[i].clone().requires_grad_(True)
        for _ in range(4):
            y = module(p, lat[i])
            (g,) = torch.autograd.grad(y.sum(), p)
        out["sdf%d" % i] = (y.detach().clone(), g.clone())

    def disc_job(out):
        x = vox.clone().requires_grad_(True)
        for _ in range(3):
            y = disc(x)
            (g,) = torch.autograd.grad(y.sum(), x)
        out["disc"] = (y.detach().clone(), g.clone())

    serial = {}
    sdf_job(net, 0, serial)
    sdf_job(replica, 1, serial)
    disc_job(serial)
    for _ in range(3):
        got, errors = {}, []

        def guarded(fn, *a):
            try:
                fn(*a)
            except Exception as e:       # noqa: BLE001 — reported below
                errors.append(e)
        threads = [threading.Thread(target=guarded, args=(sdf_job, net, 0, got)),
                   threading.Thread(target=guarded, args=(sdf_job, replica, 1, got)),
                   threading.Thread(target=guarded, args=(disc_job, got))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for k in serial:
            assert torch.equal(got[k][0], serial[k][0]) and torch.equal(got[k][1], serial[k][1]), k


def test_wgan_headline_step_at_cpu_size(on_cpu):
    FULL.test_wgan_headline_step_vs_oracle(batch=4)


def test_config3_and_config4_step_checks_at_cpu_size(on_cpu):
    """The bodies of the BASELINE-size configs[3] / configs[4] oracle checks (tests/test_gpu_fullsize.py) at sizes the twin
    finishes in seconds: iteration 1 (16^3) with fade-in, batch 3; hybrid WGAN at batch 1."""
    FULL.hybrid_progressive_case(1, 3, 0.5, 2000)
    FULL.hybrid_wgan_case(1)


# ---- round 4: the critic's tail, the classic-GAN losses, the VAE reparameterisation on the twin -----------------------------------
@pytest.mark.parametrize("N,C,act", [(6, 16, 1), (17, 8, 0), (3, 4, 2)])
def test_head_dot_cpu(on_cpu, N, C, act):
    LOSS.test_head_dot_forward_and_backward(N, C, act)


def test_conv_head_node_cpu(on_cpu):
    LOSS.test_conv_head_node_matches_the_two_layer_composition()


def test_bce_neg_mean_log_and_reparam_cpu(on_cpu):
    for n in (64, 1):
        LOSS.test_bce_and_neg_mean_log_match_torch(n)
    LOSS.test_vae_reparameterisation_matches_torch()


@pytest.mark.parametrize("batch", [5])
def test_generator_fused_inference_cpu(on_cpu, batch):
    M.test_generator_fused_inference_matches_the_unfused_form(batch)


def test_conv_transpose3d_to_one_channel_with_input_transform_cpu(on_cpu):
    OPS.test_conv_transpose3d_to_one_channel_streaming_kernel_random_shapes()


def test_grouped_generator_pass_cpu(on_cpu):
    M.test_generator_forward_groups_equals_separate_evaluations(3, 5)
    M.test_wgan_step_with_grouped_generator_pass_equals_the_updates_one_by_one(n_units=1)
    M.test_wgan_step_on_real_batches_delivered_into_the_trainers_slots()
