"""Two ranks sharing ONE MI355X (gloo transport staged through the host; the kernels, the sharded-table ops, the
gradient buckets and both fused-Adam arenas are the product code): the data-parallel + row-sharded run must reproduce
the single-process run on the global batch."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, F, H, D, BL, S = 20, 80, 32, 16, 24, 37  # S odd: ragged last shard


def _data(world):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(world * BL, T, F, generator=g)
    idx = torch.randint(0, S, (world * BL,), generator=g)
    idx[1] = idx[BL]  # the same sequence on both ranks
    ns = torch.randint(20, 200, (world * BL,), generator=g)
    e2, e1 = torch.randn(world * BL, D, generator=g), torch.randn(world * BL, D, generator=g)
    return x, idx, ns, e2, e1


def _build():
    from fhvae import FHVAE

    torch.manual_seed(11)
    return FHVAE(T * F, [H, H], [H, H], D, D, [H, H], num_seqs=S, reference_compat=False).cuda()


def _worker(rank, world, port, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-scalablefhvae_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dist_shard import DistributedFHVAE

    x, idx, ns, e2, e1 = _data(world)
    sl = slice(rank * BL, (rank + 1) * BL)
    m = _build()
    runner = DistributedFHVAE(m, lr=1e-3, betas=(0.95, 0.999))
    fwd = m.forward
    m.forward = lambda *a, **k: fwd(*a, eps=(e2[sl].cuda(), e1[sl].cuda()), **k)
    losses = [runner.train_step(x[sl].cuda(), idx[sl].cuda(), ns[sl].cuda(), alpha=10.0)[0].item() for _ in range(3)]
    full = runner.state_dict()  # collective: gathers the table shards and their Adam moments (checkpoint path)
    ret[rank] = dict(losses=losses, rows=(runner.sh.row0, runner.sh.row1), shard=runner.shard.detach().cpu(),
                     w=m.z2_pre_encoder.lstm.weight_hh_l1.detach().cpu(), wd=m.pre_decoder.lstm.weight_ih_l0.detach().cpu(),
                     status=runner.check_status(), table=full["state_dict"]["mu2_table"].cpu(),
                     table_m=full["optimizer"]["state"][0]["exp_avg"].cpu(),
                     w_m=full["optimizer"]["state"][full["param_names"].index("z2_pre_encoder.lstm.weight_hh_l1")]["exp_avg"].cpu(),
                     step=float(full["optimizer"]["state"][1]["step"]), names=full["param_names"])
    # re-shard: a fresh runner loaded from the gathered state continues identically
    m2 = _build()
    r2 = DistributedFHVAE(m2, lr=1e-3, betas=(0.95, 0.999))
    r2.load_state_dict(full)
    f2 = m2.forward
    m2.forward = lambda *a, **k: f2(*a, eps=(e2[sl].cuda(), e1[sl].cuda()), **k)
    la = runner.train_step(x[sl].cuda(), idx[sl].cuda(), ns[sl].cuda(), alpha=10.0)[0].item()
    lb = r2.train_step(x[sl].cuda(), idx[sl].cuda(), ns[sl].cuda(), alpha=10.0)[0].item()
    ret[rank] = dict(ret[rank], resumed=(la, lb), resumed_shard_equal=bool(torch.allclose(runner.shard, r2.shard, rtol=1e-4, atol=1e-5)))
    dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_process_global_batch():
    from hip_optim import FusedAdam
    from train_model import loss_function

    world = 2
    x, idx, ns, e2, e1 = _data(world)
    m = _build()
    opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
    ref_losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m(x.cuda(), idx, S, ns, eps=(e2, e1))
        loss = loss_function(out[0], out[1], 10.0)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, 29700 + os.getpid() % 200, ret), nprocs=world, join=True)
    assert ret[0]["rows"] == (0, 19) and ret[1]["rows"] == (19, 37)
    for k in range(3):  # global loss = mean of the two local losses (equal local batch sizes)
        got = 0.5 * (ret[0]["losses"][k] + ret[1]["losses"][k])
        assert abs(got - ref_losses[k]) <= 2e-4 * abs(ref_losses[k]), (k, got, ref_losses[k])
    table = m.mu2_table.detach().cpu()
    for r in range(world):
        a, b = ret[r]["rows"]
        # 3 Adam steps of lr 1e-3 move a weight by <= 3e-3; Adam's g/sqrt(v) amplifies the f32 summation-order
        # differences between the split and the global batch on near-zero gradients: allow 1e-4 absolute (3 % of a step)
        torch.testing.assert_close(ret[r]["shard"], table[a:b], rtol=2e-4, atol=1e-4)
        torch.testing.assert_close(ret[r]["w"], m.z2_pre_encoder.lstm.weight_hh_l1.detach().cpu(), rtol=2e-4, atol=1e-4)
        torch.testing.assert_close(ret[r]["wd"], m.pre_decoder.lstm.weight_ih_l0.detach().cpu(), rtol=2e-4, atol=1e-4)
        assert ret[r]["status"] == 0
        # checkpoint path: the gathered table is the concatenation of the shards; layout = the single-GPU optimizer's
        assert torch.equal(ret[r]["table"][a:b], ret[r]["shard"]) and ret[r]["table"].shape == (S, D)
        la, lb = ret[r]["resumed"]
        assert la == lb and ret[r]["resumed_shard_equal"], ret[r]["resumed"]
    assert torch.equal(ret[0]["table"], ret[1]["table"])
    sd1 = opt.state_dict()
    names1 = [n for n, _ in m.named_parameters()]
    assert ret[0]["names"] == names1 and ret[0]["step"] == 3.0 == float(sd1["state"][0]["step"])
    torch.testing.assert_close(ret[0]["table_m"], sd1["state"][names1.index("mu2_table")]["exp_avg"].cpu(), rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(ret[0]["w_m"], sd1["state"][names1.index("z2_pre_encoder.lstm.weight_hh_l1")]["exp_avg"].cpu(),
                               rtol=2e-3, atol=1e-7)


def _worker_tiny_table(rank, world, port, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-scalablefhvae_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dist_shard import DistributedFHVAE
    from fhvae import FHVAE

    torch.manual_seed(11)
    S2 = 2  # fewer rows than ranks: the last rank's shard is empty
    m = FHVAE(T * F, [H, H], [H, H], D, D, [H, H], num_seqs=S2, reference_compat=False).cuda()
    runner = DistributedFHVAE(m, lr=1e-3, betas=(0.95, 0.999))
    g = torch.Generator().manual_seed(100 + rank)
    x, idx, ns = torch.randn(8, T, F, generator=g).cuda(), torch.randint(0, S2, (8,), generator=g).cuda(), torch.randint(20, 200, (8,), generator=g).cuda()
    losses = [runner.train_step(x, idx, ns, alpha=10.0)[0].item() for _ in range(2)]
    ret[rank] = dict(losses=losses, rows=(runner.sh.row0, runner.sh.row1), table=runner.gather_table().cpu(), status=runner.check_status())
    dist.destroy_process_group()


def test_empty_shard_rank_takes_part():
    """3 ranks, a 2-row table: rank 2 owns no rows (FH_CHECK_POS(S) / n = 0 crashed it)."""
    world = 3
    ret = mp.Manager().dict()
    mp.spawn(_worker_tiny_table, args=(world, 29900 + os.getpid() % 90, ret), nprocs=world, join=True)
    assert [ret[r]["rows"] for r in range(world)] == [(0, 1), (1, 2), (2, 2)]
    for r in range(world):
        assert all(np.isfinite(ret[r]["losses"])) and ret[r]["status"] == 0 and ret[r]["table"].shape == (2, D)
        assert torch.equal(ret[r]["table"], ret[0]["table"])


# ---- the bf16 compute mode on two ranks: bf16 step cells (the persistent kernels need the whole GPU: FHVAE_NO_CLUSTER), the heads
# on the projection kernel with the stacked weights from the nets' operand-cast launch, K5 on the split-operand bf16 MFMA kernels
# over row shards merged across ranks, sign * CE riding in the kernels, deferred weight gradients + the early all-reduce hook
H2, D2, BL2, S2B = 128, 32, 32, 2048


def _build_bf16():
    from fhvae import FHVAE

    torch.manual_seed(13)
    return FHVAE(T * F, [H2, H2], [H2, H2], D2, D2, [H2, H2], num_seqs=S2B, reference_compat=False, compute_dtype="bf16").cuda()


def _data_bf16(world):
    g = torch.Generator().manual_seed(9)
    n = world * BL2
    return (torch.randn(n, T, F, generator=g), torch.randint(0, S2B, (n,), generator=g), torch.randint(20, 200, (n,), generator=g),
            torch.randn(n, D2, generator=g), torch.randn(n, D2, generator=g))


def _worker_bf16(rank, world, port, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-scalablefhvae_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FHVAE_NO_CLUSTER="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hip_binding as hb
    from dist_shard import DistributedFHVAE

    x, idx, ns, e2, e1 = _data_bf16(world)
    sl = slice(rank * BL2, (rank + 1) * BL2)
    m = _build_bf16()
    runner = DistributedFHVAE(m, lr=1e-3, betas=(0.95, 0.999))
    assert runner.overlap and runner.sh.backend.lp
    fwd = m.forward
    m.forward = lambda *a, **k: fwd(*a, eps=(e2[sl].cuda(), e1[sl].cuda()), **k)
    used0 = hb.PAIR_SIDE["used"]
    losses = [runner.train_step(x[sl].cuda(), idx[sl].cuda(), ns[sl].cuda(), alpha=10.0)[0].item() for _ in range(3)]
    ret[rank] = dict(losses=losses, shard=runner.shard.detach().cpu(), rows=(runner.sh.row0, runner.sh.row1),
                     w=m.z1_pre_encoder.lstm.weight_hh_l1.detach().cpu(), wh=m.dec_gauss_layer.mulayer.weight.detach().cpu(),
                     status=runner.check_status(), pair_used=hb.PAIR_SIDE["used"] - used0)
    dist.destroy_process_group()


def test_two_ranks_bf16_mode_match_single_process_global_batch():
    from hip_optim import FusedAdam
    from train_model import loss_function

    world = 2
    x, idx, ns, e2, e1 = _data_bf16(world)
    old = os.environ.get("FHVAE_NO_CLUSTER")
    os.environ["FHVAE_NO_CLUSTER"] = "1"
    try:
        m = _build_bf16()
        opt = FusedAdam(m.parameters(), lr=1e-3, betas=(0.95, 0.999))
        ref_losses = []
        for _ in range(3):
            opt.zero_grad()
            out = m(x.cuda(), idx, S2B, ns, eps=(e2, e1))
            loss = loss_function(out[0], out[1], 10.0)
            loss.backward()
            opt.step()
            ref_losses.append(loss.item())
        ret = mp.Manager().dict()
        mp.spawn(_worker_bf16, args=(world, 29500 + os.getpid() % 150, ret), nprocs=world, join=True)
    finally:
        if old is None:
            os.environ.pop("FHVAE_NO_CLUSTER", None)
        else:
            os.environ["FHVAE_NO_CLUSTER"] = old
    assert ret[0]["rows"] == (0, 1024) and ret[1]["rows"] == (1024, 2048)
    for k in range(3):  # same bf16 kernels on a split batch: operand rounding is per element, the f32 sums differ in order only
        got = 0.5 * (ret[0]["losses"][k] + ret[1]["losses"][k])
        assert abs(got - ref_losses[k]) <= 5e-4 * abs(ref_losses[k]), (k, got, ref_losses[k])
    table = m.mu2_table.detach().cpu()
    for r in range(world):
        a, b = ret[r]["rows"]
        assert ret[r]["status"] == 0 and ret[r]["pair_used"] == 3   # the per-frame head took the lower bound's ready-made operand every step
        torch.testing.assert_close(ret[r]["shard"], table[a:b], rtol=1e-3, atol=3e-4)
        torch.testing.assert_close(ret[r]["w"], m.z1_pre_encoder.lstm.weight_hh_l1.detach().cpu(), rtol=1e-3, atol=3e-4)
        torch.testing.assert_close(ret[r]["wh"], m.dec_gauss_layer.mulayer.weight.detach().cpu(), rtol=1e-3, atol=3e-4)
    assert torch.equal(ret[0]["w"], ret[1]["w"]) and torch.equal(ret[0]["wh"], ret[1]["wh"])   # replicas stay bit-identical
